// gwtf_prior.hip -- the global prior flow on the shape latent: the whole GlobalRNVPDecoder (reference
// lib/networks/decoders.py:7-38: n_flows RealNVPFlowCouple = 2 n_flows elementary RealNVPFlow, flows.py:163-243), forward and
// backward, eval- and train-mode BatchNorm, from ONE C call per direction.
//
// The work is per SHAPE: B rows (<= 128 in every shipped configuration on one GPU; more: row blocks) of G latents through 14 flows of two (B x G/2)(G/2 x F) -> BN -> Swish ->
// (B x F)(F x G/2) MLPs -- a few MFLOP, a chain of 28 dependent layers.  The reference runs it as ~100 library launches forward and
// ~200 backward.  Rounds 1-3 walked the whole chain in one workgroup on one compute unit (1.15 ms forward, 2.75 - 3.72 ms backward:
// hidden behind the decoders on a side stream at 64 shapes per rank, the critical path at <= 16).  Round 4 (below): one launch per
// LAYER with the layer's columns dealt over workgroups -- 2 launches per flow forward, 4 backward, 0.30 + 0.60 ms at G = 512.
//
// Raw parameter arena, per elementary flow j (module order, reference flows.py:174-190), branch 0 = mu, 1 = logvar:
//   W0[F][Gk] | bn.weight[F] | bn.bias[F] | bn.running_mean[F] | bn.running_var[F] | W1[Gw][F] | b1[Gw]
// Gw / Gk = warped / kept latents of the flow (pattern 0: even / odd, pattern 1: first / second half; flows.py:219-243).
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "gwtf_rows.h"
#include "../../include/gwtf.h"

namespace {

constexpr int kMaxFlows = 64, kMaxRows = 1 << 16, kMaxG = 512, kMaxF = 128;   // rows: any batch, walked 128 at a time

struct Geom {          // one elementary flow
  int Gw, Gk, wstride, woff, kstride, koff;
  size_t raw;          // offset of the flow's record in the raw arena
};
struct Plan {           // kernel argument: small on purpose -- a per-flow table indexed by the loop counter made the compiler copy
  int n2, B, G, F, mode; // the whole argument block to scratch memory (2.4 KB per lane, every field access a scratch load)
  float eps;
};

__host__ __device__ inline size_t branch_floats(int F, int Gw, int Gk) { return (size_t)F * Gk + 4 * (size_t)F + (size_t)Gw * F + Gw; }

// geometry of elementary flow j (reference flows.py:219-243: couples alternate between pattern 0 = even / odd latents and
// pattern 1 = first / second half); the record offset is the sum of the preceding records (wave-uniform integer work)
__host__ __device__ inline Geom geom_at(int j, int G, int F) {
  Geom g;
  size_t off = 0;
  for (int i = 0; i <= j; ++i) {
    const int pattern = (i / 2) % 2, second = i % 2;
    if (pattern == 0) {
      g.wstride = g.kstride = 2;
      g.woff = second; g.koff = 1 - second;
      g.Gw = second ? G / 2 : (G + 1) / 2;
    } else {
      g.wstride = g.kstride = 1;
      g.Gw = second ? G - G / 2 : G / 2;
      g.woff = second ? G / 2 : 0;
      g.koff = second ? 0 : G / 2;
    }
    g.Gk = G - g.Gw;
    g.raw = off;
    off += 2 * branch_floats(F, g.Gw, g.Gk);
  }
  return g;
}

struct Branch {        // pointers into one branch record
  const float *W0, *gamma, *beta, *rm, *rv, *W1, *b1;
};
__device__ inline Branch branch_of(const float* raw, const Geom& ge, int F, int x) {
  const float* p = raw + ge.raw + (size_t)x * branch_floats(F, ge.Gw, ge.Gk);
  Branch b;
  b.W0 = p; p += (size_t)F * ge.Gk;
  b.gamma = p; b.beta = p + F; b.rm = p + 2 * F; b.rv = p + 3 * F; p += 4 * (size_t)F;
  b.W1 = p; p += (size_t)ge.Gw * F;
  b.b1 = p;
  return b;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ROUND 4: one launch per LAYER, the columns of the layer dealt over workgroups (as gwtf_heads.hip does for the Gaussian heads).
// The single-workgroup version above this comment in the history walked all 14 flows on ONE compute unit: 1.15 ms forward + 2.75 ms
// (G = 128) / 3.72 ms (G = 512) backward -- beside the decoders on a side stream that hides it at 64 shapes per rank, but the
// critical path of a data-parallel step at <= 16 shapes per rank (bench.py also.train_step.ae_shard).  BatchNorm's batch statistics
// are per COLUMN, so a layer splits over column blocks with no exchange inside it; only the layer boundaries (a contraction over all
// columns of the previous layer) need the launch boundary.  Per flow:
//   forward   hidden_kernel  (2 ceil(F/16) workgroups: 16 hidden columns each, both branches)  kept . W0^T -> BatchNorm -> Swish
//             out_kernel     (ceil(Gw/16) workgroups: 16 warped latents each)                   h . W1^T + b1 -> log / exp -> affine map,
//                                                                                             kept latents copied through
//   backward  hidden_kernel  (recompute h, xhat, statistics)
//             out_bwd_kernel (ceil(Gw/16)): affine map backward -> dO block, dW1, db1, the warped part of the flowing gradient
//             hid_bwd_kernel (2 ceil(F/16)): dH = dO . W1 -> Swish / BatchNorm backward -> dgamma, dbeta, dHpre block, dW0
//             kept_bwd_kernel(ceil(Gk/16)): dkept = dHpre . W0 (both branches) + the kept part of the flowing gradient
// Every product on the exact-fp32 MFMA with the latent rows on M (gwtf_rows.h), operands straight from L2.
// ---------------------------------------------------------------------------------------------------------------------------------
using namespace gwtf_rows;

// acc[m] += A(row 16 (wave + 4 m) + c16, k) * B(col c16, k) over k < K.  arow[m] / brow: the lane's operand rows; AV / BV: k is the
// contiguous index (16-byte loads) else strided by ask / bsk.  Rounds of four 16-k steps; A is zeroed beyond K (B is clamped).
template <int MTW, bool AV, bool BV>
__device__ __forceinline__ void tile_product(f32x4 (&acc)[MTW], const float* const (&arow)[MTW], size_t ask, const float* brow, size_t bsk,
                                             int K, int q) {
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += 64) {
    f32x4 a[4][MTW], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      b[s] = BV ? load4v(brow, ks, K) : load4s(brow, ks, K, bsk);
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[s][m] = AV ? load4v(arow[m], ks, K) : load4s(arow[m], ks, K, ask);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (k0 + 16 * s >= K) break;                 // wave-uniform
      const int ks = k0 + 16 * s + 4 * q;
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[s][m] = zero_from(a[s][m], ks, K);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][m][t], b[s][t], acc[m], 0, 0, 0);
    }
  }
}

// Row blocks: a workgroup holds 64 MTW latent rows in its accumulators (MTW = 1: B <= 64, else 2).  A longer batch (the gathered rows
// of a large data-parallel group) is walked in blocks of 128 rows: whatever must wait for a column total over ALL rows (BatchNorm
// statistics, their backward sums) is stashed in the element's own output slot by the lane that owns it and read back by the same
// lane after the total -- no exchange, no extra barrier; weight gradients add up block by block in their own slots (one owner
// each).  With one block (B <= 128: every shipped configuration on one GPU) nothing is stashed and the code path is the
// register-resident one the timings in the header were taken on.

// hidden layer of one flow: block (x = branch, ft = tile of 16 hidden columns).  H / XH [B][2][F], ST [2][2][F] = statistics used,
// bn_stats_j [2][2][F] = batch statistics (train) or null
template <int MTW>
__global__ __launch_bounds__(256) void hidden_kernel(const Plan P, const Geom ge, const float* __restrict__ raw, const float* __restrict__ gin,
                                                     float* __restrict__ H, float* __restrict__ XH, float* __restrict__ ST,
                                                     float* __restrict__ bn_stats_j, int training) {
  constexpr int ROWS = 64 * MTW;
  __shared__ float s_red[4][16];
  const int B = P.B, F = P.F, G = P.G, FT = (F + 15) / 16;
  const int NRB = (B + ROWS - 1) / ROWS;
  const int x = blockIdx.x / FT, ft = blockIdx.x - x * FT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  const Branch br = branch_of(raw, ge, F, x);
  const int f = 16 * ft + c16, fc = min(f, F - 1);
  f32x4 acc[MTW];
  const float* brow = br.W0 + (size_t)fc * ge.Gk;
  auto product = [&](int r0) {
    const float* arow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      arow[m] = gin + (size_t)min(r0 + 16 * (wave + 4 * m) + c16, B - 1) * G + ge.koff;
    }
    if (ge.kstride == 1) tile_product<MTW, true, true>(acc, arow, 1, brow, 1, ge.Gk, q);
    else tile_product<MTW, false, true>(acc, arow, (size_t)ge.kstride, brow, 1, ge.Gk, q);
  };
  auto slot = [&](int b) { return (size_t)b * 2 * F + (size_t)x * F + f; };
  auto reload = [&](int r0) {                  // the pre-activations this lane stashed
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
        acc[m][r] = (b < B && f < F) ? H[slot(b)] : 0.f;
      }
  };
  float mean, var;
  if (training) {
    float s1[1] = {0.f};
#pragma unroll 1
    for (int rb = 0; rb < NRB; ++rb) {
      const int r0 = rb * ROWS;
      product(r0);
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
          s1[0] += b < B ? acc[m][r] : 0.f;
          if (NRB > 1 && b < B && f < F) H[slot(b)] = acc[m][r];
        }
    }
    column_totals<1>(s1, s_red, wave, c16, q);
    mean = s1[0] / (float)B;
    float s2[1] = {0.f};
#pragma unroll 1
    for (int rb = 0; rb < NRB; ++rb) {
      const int r0 = rb * ROWS;
      if (NRB > 1) reload(r0);
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[m][r] - mean;
          s2[0] += r0 + 16 * (wave + 4 * m) + 4 * q + r < B ? d * d : 0.f;
        }
    }
    column_totals<1>(s2, s_red, wave, c16, q);
    var = s2[0] / (float)B;
  } else {
    mean = br.rm[fc];
    var = br.rv[fc];
  }
  const bool fon = f < F;      // (no early exit: every lane supplies operand ROWS to the products of the loop below)
  if (fon && wave == 0 && q == 0) {
    if (ST) { ST[(x * 2 + 0) * F + f] = mean; ST[(x * 2 + 1) * F + f] = var; }
    if (training && bn_stats_j) { bn_stats_j[(x * 2 + 0) * F + f] = mean; bn_stats_j[(x * 2 + 1) * F + f] = var; }
  }
  const float isd = 1.0f / sqrtf(var + GWTF_BN_EPS), ga = br.gamma[fc], be = br.beta[fc];
#pragma unroll 1
  for (int rb = 0; rb < NRB; ++rb) {
    const int r0 = rb * ROWS;
    if (!training) product(r0);
    else if (NRB > 1) reload(r0);
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
        if (b < B && fon) {
          const float xh = (acc[m][r] - mean) * isd;
          const size_t o = slot(b);
          if (XH) XH[o] = xh;
          H[o] = swish(fmaf(xh, ga, be));
        }
      }
  }
}

// output layer + affine map of one flow: block = 16 warped latents; every block also copies its share of the kept latents through
template <int MTW>
__global__ __launch_bounds__(256) void out_kernel(const Plan P, const Geom ge, const float* __restrict__ raw, const float* __restrict__ gin,
                                                  const float* __restrict__ H, float* __restrict__ gout, float* __restrict__ mo,
                                                  float* __restrict__ lo) {
  constexpr int ROWS = 64 * MTW;
  const int B = P.B, F = P.F, G = P.G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  const Branch bm = branch_of(raw, ge, F, 0), bl = branch_of(raw, ge, F, 1);
  const int w = 16 * blockIdx.x + c16, wc = min(w, ge.Gw - 1);
#pragma unroll 1
  for (int r0 = 0; r0 < B; r0 += ROWS) {
    f32x4 am[MTW], al[MTW];
    const float* hm[MTW];
    const float* hl[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      am[m] = al[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      hm[m] = H + (size_t)min(r0 + 16 * (wave + 4 * m) + c16, B - 1) * 2 * F;
      hl[m] = hm[m] + F;
    }
    tile_product<MTW, true, true>(am, hm, 1, bm.W1 + (size_t)wc * F, 1, F, q);
    tile_product<MTW, true, true>(al, hl, 1, bl.W1 + (size_t)wc * F, 1, F, q);
    if (w < ge.Gw) {
      const int col = ge.woff + ge.wstride * w;
      const float b1m = bm.b1[w], b1l = bl.b1[w];
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
          if (b < B) {
            const size_t o = (size_t)b * G + col;
            const float x0 = gin[o], mu = am[m][r] + b1m;
            const float lv = logf(P.eps + expf(al[m][r] + b1l));                                                // flows.py:198-201
            gout[o] = P.mode == GWTF_MODE_DIRECT ? expf(0.5f * lv) * x0 + mu : expf(-0.5f * lv) * (x0 - mu);   // :206-209
            mo[o] = mu;
            lo[o] = lv;
          }
        }
    }
  }
  // kept latents pass through (mu = logvar = 0 there: exp(0) * g + 0)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < B * ge.Gk; t += gridDim.x * 256) {
    const int b = t / ge.Gk, kk = t - b * ge.Gk;
    const size_t o = (size_t)b * G + ge.koff + ge.kstride * kk;
    gout[o] = gin[o];
    mo[o] = 0.f;
    lo[o] = 0.f;
  }
}

// backward through the affine map and the output layer's weights: block = 16 warped latents.  dO [B][2][Gwmax]
template <int MTW>
__global__ __launch_bounds__(256) void out_bwd_kernel(const Plan P, const Geom ge, int j, const float* __restrict__ gout,
                                                      const float* __restrict__ mu_j, const float* __restrict__ lv_j,
                                                      const float* __restrict__ Gcur, const float* __restrict__ Ggs,
                                                      const float* __restrict__ Glvs, const float* __restrict__ H,
                                                      float* __restrict__ dO, float* __restrict__ Gnext, float* __restrict__ gr, int Gwmax) {
  constexpr int ROWS = 64 * MTW;
  __shared__ float s_do[2][ROWS][17];
  const int B = P.B, F = P.F, G = P.G, FT = (F + 15) / 16;
  const size_t BG = (size_t)B * G, BF = branch_floats(F, ge.Gw, ge.Gk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  const int w0 = 16 * blockIdx.x;
  float sdo = 0.f;                                 // db1 of (branch, latent) = thread < 32, summed over the row blocks
#pragma unroll 1
  for (int r0 = 0; r0 < B; r0 += ROWS) {
    const int nb = min(ROWS, B - r0);
    if (r0) __syncthreads();                       // the previous block's products have read s_do
    for (int t = threadIdx.x; t < ROWS * 16; t += 256) {
      const int bl = t >> 4, b = r0 + bl, wl = t & 15, w = w0 + wl;
      float dmu = 0.f, dol = 0.f;
      if (b < B && w < ge.Gw) {
        const size_t o = (size_t)b * G + ge.woff + ge.wstride * w;
        const float go = Gcur[o] + (Ggs ? Ggs[(size_t)j * BG + o] : 0.f);
        const float lv = lv_j[o];
        float dx, dlv;
        if (P.mode == GWTF_MODE_DIRECT) {          // y = e^{lv/2} x + mu
          dx = go * expf(0.5f * lv);
          dmu = go;
          dlv = go * 0.5f * (gout[o] - mu_j[o]);
        } else {                                    // y = e^{-lv/2} (x - mu)
          const float e = expf(-0.5f * lv);
          dx = go * e;
          dmu = -go * e;
          dlv = -0.5f * go * gout[o];
        }
        if (Glvs) dlv += Glvs[(size_t)j * BG + o];
        dol = dlv * (1.0f - P.eps * expf(-lv));     // lv = log(eps + e^o): dlv/do = e^o / (eps + e^o)
        dO[((size_t)b * 2 + 0) * Gwmax + w] = dmu;
        dO[((size_t)b * 2 + 1) * Gwmax + w] = dol;
        Gnext[o] = dx;
      }
      s_do[0][bl][wl] = dmu;
      s_do[1][bl][wl] = dol;
    }
    __syncthreads();
    if (threadIdx.x < 32) {                         // db1
      const int x = threadIdx.x >> 4, wl = threadIdx.x & 15;
      for (int bl = 0; bl < nb; ++bl) sdo += s_do[x][bl][wl];
    }
    // dW1_x[w][f] (+)= sum_b dO_x[b][w] H_x[b][f]: 2 x FT output tiles dealt over the waves
    const int KB = (nb + 15) / 16 * 16;
#pragma unroll 1
    for (int id = wave; id < 2 * FT; id += 4) {
      const int x = id / FT, nt = id - x * FT;
      f32x4 dw = {0.f, 0.f, 0.f, 0.f};
      const int f = 16 * nt + c16;
      const float* hcol = H + (size_t)r0 * 2 * F + (size_t)x * F + min(f, F - 1);
#pragma unroll 2
      for (int k0 = 0; k0 < KB; k0 += 16) {
        const f32x4 bv = load4s(hcol, k0 + 4 * q, nb, (size_t)2 * F);
#pragma unroll
        for (int t = 0; t < 4; ++t) dw = __builtin_amdgcn_mfma_f32_16x16x4f32(s_do[x][k0 + 4 * q + t][c16], bv[t], dw, 0, 0, 0);
      }
      float* dW1 = gr + (size_t)x * BF + (size_t)F * ge.Gk + 4 * (size_t)F;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int w = w0 + 4 * q + r;
        if (w < ge.Gw && f < F) dW1[(size_t)w * F + f] = r0 ? dW1[(size_t)w * F + f] + dw[r] : dw[r];
      }
    }
  }
  if (threadIdx.x < 32) {
    const int x = threadIdx.x >> 4, wl = threadIdx.x & 15;
    if (w0 + wl < ge.Gw) gr[(size_t)x * BF + (size_t)F * ge.Gk + 4 * (size_t)F + (size_t)ge.Gw * F + w0 + wl] = sdo;
  }
}

// dH = dO . W1 -> through Swish and BatchNorm -> dgamma, dbeta, dHpre block (DHP [B][2][F]) and dW0: block (x, ft)
template <int MTW>
__global__ __launch_bounds__(256) void hid_bwd_kernel(const Plan P, const Geom ge, const float* __restrict__ raw, const float* __restrict__ xin,
                                                      const float* __restrict__ dO, const float* __restrict__ XH, const float* __restrict__ ST,
                                                      float* __restrict__ DHP, float* __restrict__ gr, int Gwmax, int training) {
  constexpr int ROWS = 64 * MTW;
  __shared__ float s_dy[ROWS][17];
  __shared__ float s_red[4][16];
  const int B = P.B, F = P.F, G = P.G, FT = (F + 15) / 16;
  const int NRB = (B + ROWS - 1) / ROWS;
  const size_t BF = branch_floats(F, ge.Gw, ge.Gk);
  const int x = blockIdx.x / FT, ft = blockIdx.x - x * FT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  const Branch br = branch_of(raw, ge, F, x);
  const int f = 16 * ft + c16, fc = min(f, F - 1);
  const float ga = br.gamma[fc], be = br.beta[fc];
  const float isd = 1.0f / sqrtf(ST[(x * 2 + 1) * F + fc] + GWTF_BN_EPS);
  auto slot = [&](int b) { return (size_t)b * 2 * F + (size_t)x * F + f; };
  f32x4 acc[MTW];
  float xh[MTW][4], sdy[1] = {0.f}, sdyx[1] = {0.f};
  // pass A: dH through the Swish per row block; the column sums over ALL rows; with several blocks dH is stashed in DHP's slot
#pragma unroll 1
  for (int rb = 0; rb < NRB; ++rb) {
    const int r0 = rb * ROWS;
    const float* arow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      arow[m] = dO + ((size_t)min(r0 + 16 * (wave + 4 * m) + c16, B - 1) * 2 + x) * Gwmax;
    }
    tile_product<MTW, true, false>(acc, arow, 1, br.W1 + fc, (size_t)F, ge.Gw, q);
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
        const bool on = b < B && f < F;
        const float xn = on ? XH[slot(b)] : 0.f;
        const float hb = fmaf(xn, ga, be);
        const float sg = 1.0f / (1.0f + expf(-hb));
        const float dh = on ? acc[m][r] * (sg * (1.0f + hb * (1.0f - sg))) : 0.f;
        xh[m][r] = xn;
        acc[m][r] = dh;
        sdy[0] += dh;
        sdyx[0] = fmaf(dh, xn, sdyx[0]);
        if (NRB > 1 && on) DHP[slot(b)] = dh;
      }
  }
  column_totals<1>(sdy, s_red, wave, c16, q);
  column_totals<1>(sdyx, s_red, wave, c16, q);
  if (wave == 0 && q == 0 && f < F) {
    float* gb = gr + (size_t)x * BF + (size_t)F * ge.Gk;
    gb[f] = sdyx[0];            // d gamma
    gb[F + f] = sdy[0];         // d beta
  }
  const float m1 = training ? sdy[0] / (float)B : 0.f, m2 = training ? sdyx[0] / (float)B : 0.f;   // eval: the statistics are constants
  // pass B: through the BatchNorm, the dHpre block, dW0 (+)= per row block
  const int GkT = (ge.Gk + 15) / 16;
#pragma unroll 1
  for (int rb = 0; rb < NRB; ++rb) {
    const int r0 = rb * ROWS, nb = min(ROWS, B - r0);
    if (rb) __syncthreads();                         // the previous block's products have read s_dy
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bl = 16 * (wave + 4 * m) + 4 * q + r, b = r0 + bl;
        const bool on = b < B && f < F;
        float dh = acc[m][r], xn = xh[m][r];
        if (NRB > 1) {
          dh = on ? DHP[slot(b)] : 0.f;
          xn = on ? XH[slot(b)] : 0.f;
        }
        const float dy = on ? ga * isd * (dh - m1 - xn * m2) : 0.f;
        s_dy[bl][c16] = dy;
        if (on) DHP[slot(b)] = dy;
      }
    __syncthreads();
    // dW0_x[f][kk] (+)= sum_b dHpre[b][f] kept[b][kk]: ceil(Gk / 16) output tiles dealt over the waves
    const int KB = (nb + 15) / 16 * 16;
#pragma unroll 1
    for (int nt = wave; nt < GkT; nt += 4) {
      f32x4 dw = {0.f, 0.f, 0.f, 0.f};
      const int kk = 16 * nt + c16;
      const float* kcol = xin + (size_t)r0 * G + ge.koff + (size_t)ge.kstride * min(kk, ge.Gk - 1);
#pragma unroll 2
      for (int k0 = 0; k0 < KB; k0 += 16) {
        const f32x4 bv = load4s(kcol, k0 + 4 * q, nb, (size_t)G);
#pragma unroll
        for (int t = 0; t < 4; ++t) dw = __builtin_amdgcn_mfma_f32_16x16x4f32(s_dy[k0 + 4 * q + t][c16], bv[t], dw, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fo = 16 * ft + 4 * q + r;
        if (fo < F && kk < ge.Gk) {
          float* d = gr + (size_t)x * BF + (size_t)fo * ge.Gk + kk;
          *d = rb ? *d + dw[r] : dw[r];
        }
      }
    }
  }
}

// dkept = dHpre . W0 (both branches summed) + the flowing gradient on the kept latents: block = 16 kept latents
template <int MTW>
__global__ __launch_bounds__(256) void kept_bwd_kernel(const Plan P, const Geom ge, int j, const float* __restrict__ raw,
                                                       const float* __restrict__ DHP, const float* __restrict__ Gcur,
                                                       const float* __restrict__ Ggs, float* __restrict__ Gnext) {
  constexpr int ROWS = 64 * MTW;
  const int B = P.B, F = P.F, G = P.G;
  const size_t BG = (size_t)B * G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  const int kk = 16 * blockIdx.x + c16, kc = min(kk, ge.Gk - 1);
#pragma unroll 1
  for (int r0 = 0; r0 < B; r0 += ROWS) {
    f32x4 acc[MTW];
    const float* arow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int x = 0; x < 2; ++x) {
      const Branch br = branch_of(raw, ge, F, x);
#pragma unroll
      for (int m = 0; m < MTW; ++m) arow[m] = DHP + (size_t)min(r0 + 16 * (wave + 4 * m) + c16, B - 1) * 2 * F + (size_t)x * F;
      tile_product<MTW, true, false>(acc, arow, 1, br.W0 + kc, (size_t)ge.Gk, F, q);
    }
    if (kk < ge.Gk) {
      const int col = ge.koff + ge.kstride * kk;
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
          if (b < B) {
            const size_t o = (size_t)b * G + col;
            Gnext[o] = Gcur[o] + (Ggs ? Ggs[(size_t)j * BG + o] : 0.f) + acc[m][r];
          }
        }
    }
  }
}

bool make_plan(Plan& P, int n_flows, int B, int G, int F, int mode, float eps) {
  if (n_flows <= 0 || 2 * n_flows > kMaxFlows || B <= 0 || B > kMaxRows || G < 2 || G > kMaxG || F <= 0 || F > kMaxF) return false;
  P.n2 = 2 * n_flows; P.B = B; P.G = G; P.F = F; P.mode = mode; P.eps = eps;
  return true;
}

}  // namespace

extern "C" size_t gwtf_prior_raw_floats(int n_flows, int G, int F) {
  Plan P;
  if (!make_plan(P, n_flows, 1, G, F, GWTF_MODE_DIRECT, 0.f)) return 0;
  const Geom l = geom_at(P.n2 - 1, G, F);
  return l.raw + 2 * branch_floats(F, l.Gw, l.Gk);
}

extern "C" size_t gwtf_prior_raw_offset(int n_flows, int G, int F, int j) {
  Plan P;
  if (!make_plan(P, n_flows, 1, G, F, GWTF_MODE_DIRECT, 0.f) || j < 0 || j >= P.n2) return 0;
  return geom_at(j, G, F).raw;
}

extern "C" size_t gwtf_prior_workspace_floats(int B, int G, int F) {
  // backward: dO [B][2][Gwmax] | H [B][2][F] | XH [B][2][F] | DHP [B][2][F] | ST [2][2][F] | GA [B][G] | GB [B][G]; the forward uses H only
  const size_t Gwmax = (size_t)(G + 1) / 2;
  return (size_t)B * 2 * Gwmax + 3 * (size_t)B * 2 * F + 4 * (size_t)F + 2 * (size_t)B * G;
}

#define GWTF_PRIOR_LAUNCH(KERNEL, GRID, ...)                                                                  \
  do {                                                                                                        \
    if (B > 64) hipLaunchKernelGGL((KERNEL<2>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);                   \
    else hipLaunchKernelGGL((KERNEL<1>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);                          \
  } while (0)

extern "C" int gwtf_prior_forward(const float* g, const float* raw, float* gs, float* mus, float* logvars, float* workspace,
                                  float* bn_stats, int n_flows, int B, int G, int F, float eps, int mode, int training,
                                  void* stream) {
  Plan P;
  if (!g || !raw || !gs || !mus || !logvars || !workspace || (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE) ||
      !make_plan(P, n_flows, B, G, F, mode, eps) || (training && B < 2))
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t BG = (size_t)B * G;
  const int FT = (F + 15) / 16;
  float* H = workspace;
  const float* cur = g;
  for (int step = 0; step < P.n2; ++step) {
    const int j = mode == GWTF_MODE_DIRECT ? step : P.n2 - 1 - step;
    const Geom ge = geom_at(j, G, F);
    float* stats_j = (training && bn_stats) ? bn_stats + (size_t)j * 4 * F : nullptr;
    GWTF_PRIOR_LAUNCH(hidden_kernel, 2 * FT, P, ge, raw, cur, H, static_cast<float*>(nullptr), static_cast<float*>(nullptr), stats_j, training);
    GWTF_PRIOR_LAUNCH(out_kernel, (ge.Gw + 15) / 16, P, ge, raw, cur, static_cast<const float*>(H), gs + j * BG, mus + j * BG, logvars + j * BG);
    cur = gs + j * BG;
  }
  return (int)hipGetLastError();
}

extern "C" int gwtf_prior_backward(const float* g, const float* raw, const float* gs, const float* mus, const float* logvars,
                                   const float* g_gs, const float* g_logvars, float* workspace, float* g_raw, float* g_g,
                                   int n_flows, int B, int G, int F, float eps, int mode, int training, void* stream) {
  Plan P;
  if (!g || !raw || !gs || !mus || !logvars || !workspace || !g_raw || !g_g ||
      (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE) || !make_plan(P, n_flows, B, G, F, mode, eps) || (training && B < 2))
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int Gwmax = (G + 1) / 2, FT = (F + 15) / 16;
  const size_t BG = (size_t)B * G;
  float* dO = workspace;
  float* H = dO + (size_t)B * 2 * Gwmax;
  float* XH = H + (size_t)B * 2 * F;
  float* DHP = XH + (size_t)B * 2 * F;
  float* ST = DHP + (size_t)B * 2 * F;
  float* Gcur = ST + 4 * F;
  float* Gnext = Gcur + BG;
  hipError_t e = hipMemsetAsync(Gcur, 0, BG * sizeof(float), st);      // everything enters through the list slots
  if (e != hipSuccess) return (int)e;
  for (int step = 0; step < P.n2; ++step) {
    const int j = mode == GWTF_MODE_DIRECT ? P.n2 - 1 - step : step;       // reverse of the forward's processing order
    const Geom ge = geom_at(j, G, F);
    const int jprev = mode == GWTF_MODE_DIRECT ? j - 1 : j + 1;           // the flow whose output this flow read
    const float* xin = (jprev < 0 || jprev >= P.n2) ? g : gs + (size_t)jprev * BG;
    float* gr = g_raw + ge.raw;
    GWTF_PRIOR_LAUNCH(hidden_kernel, 2 * FT, P, ge, raw, xin, H, XH, ST, static_cast<float*>(nullptr), training);
    GWTF_PRIOR_LAUNCH(out_bwd_kernel, (ge.Gw + 15) / 16, P, ge, j, gs + (size_t)j * BG, mus + (size_t)j * BG, logvars + (size_t)j * BG,
                      static_cast<const float*>(Gcur), g_gs, g_logvars, static_cast<const float*>(H), dO, Gnext, gr, Gwmax);
    GWTF_PRIOR_LAUNCH(hid_bwd_kernel, 2 * FT, P, ge, raw, xin, static_cast<const float*>(dO), static_cast<const float*>(XH),
                      static_cast<const float*>(ST), DHP, gr, Gwmax, training);
    GWTF_PRIOR_LAUNCH(kept_bwd_kernel, (ge.Gk + 15) / 16, P, ge, j, raw, static_cast<const float*>(DHP), static_cast<const float*>(Gcur), g_gs,
                      Gnext);
    float* tmp = Gcur; Gcur = Gnext; Gnext = tmp;
  }
  e = hipMemcpyAsync(g_g, Gcur, BG * sizeof(float), hipMemcpyDeviceToDevice, st);
  return e != hipSuccess ? (int)e : (int)hipGetLastError();
}
