// gwtf_film_train.hip -- the FiLM conditioning heads of every coupling under autograd: forward + backward in HIP.
//
// Reference: CondRealNVPFlow3D's T_*_0_cond_w / T_*_0_cond_b (lib/networks/flows.py:33-45, 68-80, used at :100-101, 105-106):
//     Linear(G -> f, no bias) -> BatchNorm1d(f) over the B latent rows -> Swish -> Linear(f -> f) ,   a = eps + exp(head_w(g)),  b = head_b(g)
// A decoder stack has 4 such heads per coupling (2 branches x {scale, shift}); a mixture of K stacks of C couplings H = 4 K C of them
// (528 for the airplane config), all reading the same latent rows.  model.train() (batch statistics over the rows) and the
// differentiable eval path (running statistics) ran them as two batched library products around one BatchNorm + Swish kernel plus
// autograd's slicing glue (~20 launches forward + backward, 0.55 ms of a 12 ms training step).  Here:
//
//   film_heads_fwd_kernel   one workgroup per HEAD walks the whole chain with its intermediates in registers / LDS: layer 0 on the
//                           exact-fp32 MFMA (rows = latent rows, columns = the head's f features, operands straight from L2),
//                           column statistics (two passes, like torch), BatchNorm + Swish in the accumulators' own layout, one LDS
//                           transpose, layer 1, exp / bias epilogue written straight into the pipeline's film_raw record
//                           [B][KC][2][2][FP].  Parameters are read IN PLACE from the raw arena (gwtf_layout.h GwtfRaw).
//   film_heads_bwd_kernel   one workgroup per head: d(out) -> dL1, db1, d(hn) -> BatchNorm / Swish backward (batch-statistic terms
//                           in train mode) -> dgamma, dbeta, d(hraw) -> dL0, each parameter gradient written IN PLACE into the flat
//                           arena gradient (every slot has exactly one writer).
//   film_heads_dg_kernel    dL/dg = sum over ALL heads' columns of d(hraw) . L0 -- the one contraction that crosses heads: a split-K
//                           product over (column tile of G) x (slice of heads), partials summed by the caller (deterministic).
//
// An earlier fused attempt (docs/LOG.md round 3: 650 us per replay) chained generic one-workgroup GEMM pieces that handed their
// intermediates over through global memory; nothing here leaves the compute unit between the layers.
// Exact fp32 products (v_mfma_f32_16x16x4_f32): the work is tiny (0.6 MFLOP per head and layer), latency is what counts.
// Limits: f <= 96; any number of latent rows B_all (all ranks' rows when data parallel), walked 128 at a time.
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "gwtf_rows.h"
#include "../../include/gwtf.h"

namespace {

using namespace gwtf_rows;

struct HeadPtrs { const float *L0, *bn, *L1, *b1; size_t off_L0, off_bn, off_L1, off_b1; };
__device__ __forceinline__ HeadPtrs head_of(const float* raw, int h, int f, int G) {
  const GwtfRaw R(f, G);
  const int kc = h >> 2, br = (h >> 1) & 1, wh = h & 1;
  const size_t base = (size_t)kc * R.coupling_size() + (size_t)br * R.branch_size();
  HeadPtrs p;
  p.off_L0 = base + R.film_l0(wh); p.off_bn = base + R.film_bn(wh); p.off_L1 = base + R.film_l1(wh); p.off_b1 = base + R.film_l1b(wh);
  p.L0 = raw + p.off_L0; p.bn = raw + p.off_bn; p.L1 = raw + p.off_L1; p.b1 = raw + p.off_b1;
  return p;
}

// NT = FP / 16 column tiles of a head, MTW = row tiles per wave (1: B_all <= 64, else 2); 4 waves, wave w owns row tiles w, w + 4 of a
// block of 64 MTW rows.  More than 128 rows (the gathered rows of a large data-parallel group): the rows are walked in blocks; what
// must wait for a column total over ALL rows is stashed in the element's own output slot (hraw forward, dhraw backward) by the lane
// that owns it and read back by the same lane, and the parameter gradients add up block by block in their own slots.  With one block
// nothing is stashed: the register-resident path the timings above were taken on.
template <int NT, int MTW>
__global__ __launch_bounds__(256) void film_heads_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ g,
                                                             const float* __restrict__ poison, float* __restrict__ hraw,
                                                             float* __restrict__ hn, float* __restrict__ stats,
                                                             float* __restrict__ film_raw, int KC, int f, int G, int Ball, int row0,
                                                             int B, float eps, int training) {
  constexpr int FP = 16 * NT, PITCH = FP + 4, ROWS = 64 * MTW;
  __shared__ __align__(16) float s_hn[ROWS * PITCH];
  __shared__ float s_red[4][FP];
  const int h = blockIdx.x, kc = h >> 2, br = (h >> 1) & 1, wh = h & 1, H = 4 * KC;
  const int NRB = (Ball + ROWS - 1) / ROWS;
  const HeadPtrs P = head_of(raw, h, f, G);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;

  // ---- layer 0: hraw[b][j] = sum_k g[b][k] L0[j][k] ------------------------------------------------------------------
  f32x4 acc[MTW][NT];
  const float* brow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) brow[nt] = P.L0 + (size_t)min(16 * nt + c16, f - 1) * G;
  auto layer0 = [&](int r0) {
    const float* arow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      arow[m] = g + (size_t)min(r0 + 16 * (wave + 4 * m) + c16, Ball - 1) * G;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int k0 = 0; k0 < G; k0 += 64) {           // four 16-k steps per round: every load of a round is issued before its first MFMA
      f32x4 a[4][MTW], b[4][NT];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ks = k0 + 16 * s + 4 * q;
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[s][m] = load4v(arow[m], ks, G);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[s][nt] = load4v(brow[nt], ks, G);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int ks = k0 + 16 * s + 4 * q;
        if (k0 + 16 * s >= G) break;                 // wave-uniform
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[s][m] = zero_from(a[s][m], ks, G);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][m][t], b[s][nt][t], acc[m][nt], 0, 0, 0);
      }
    }
  };
  // accumulator (m, nt)[r] = hraw of row r0 + 16 (wave + 4 m) + 4 q + r, column 16 nt + c16
  auto hslot = [&](int b, int j) { return ((size_t)b * H + h) * f + j; };
  auto reload = [&](int r0) {
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r, j = 16 * nt + c16;
          acc[m][nt][r] = (b < Ball && j < f) ? hraw[hslot(b, j)] : 0.f;
        }
  };

  // ---- BatchNorm statistics over the rows (two passes, as torch) or the running statistics ----------------------------
  float mean[NT], rstd[NT], var[NT];
  if (training) {
    float s1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s1[nt] = 0.f;
#pragma unroll 1
    for (int rb = 0; rb < NRB; ++rb) {
      const int r0 = rb * ROWS;
      layer0(r0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r, j = 16 * nt + c16;
            s1[nt] += b < Ball ? acc[m][nt][r] : 0.f;
            if (NRB > 1 && b < Ball && j < f) hraw[hslot(b, j)] = acc[m][nt][r];
          }
    }
    column_totals<NT>(s1, s_red, wave, c16, q);
    float s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      mean[nt] = s1[nt] / (float)Ball;
      s2[nt] = 0.f;
    }
#pragma unroll 1
    for (int rb = 0; rb < NRB; ++rb) {
      const int r0 = rb * ROWS;
      if (NRB > 1) reload(r0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc[m][nt][r] - mean[nt];
            s2[nt] += r0 + 16 * (wave + 4 * m) + 4 * q + r < Ball ? d * d : 0.f;
          }
    }
    column_totals<NT>(s2, s_red, wave, c16, q);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) var[nt] = s2[nt] / (float)Ball;
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = min(16 * nt + c16, f - 1);
      mean[nt] = P.bn[2 * f + j];
      var[nt] = P.bn[3 * f + j];
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) rstd[nt] = 1.0f / sqrtf(var[nt] + GWTF_BN_EPS);
  if (wave == 0 && q == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = 16 * nt + c16;
      if (j < f) {
        stats[(size_t)h * f + j] = mean[nt];
        stats[(size_t)(H + h) * f + j] = var[nt];
        stats[(size_t)(2 * H + h) * f + j] = rstd[nt];
      }
    }
  }

  const float* lrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) lrow[nt] = P.L1 + (size_t)min(16 * nt + c16, f - 1) * f;
  const float pz = poison ? poison[kc * 2 + br] : 0.f;
#pragma unroll 1
  for (int rb = 0; rb < NRB; ++rb) {
    const int r0 = rb * ROWS;
    if (!training) layer0(r0);
    else if (NRB > 1) reload(r0);
    if (rb) __syncthreads();                         // the previous block's second layer has read s_hn
    // ---- BatchNorm + Swish in place; hraw / hn to memory (the backward's inputs), hn to LDS for the second layer ---------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = 16 * nt + c16;
      const bool jon = j < f;
      const float ga = P.bn[min(j, f - 1)], be = P.bn[f + min(j, f - 1)];
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int bl = 16 * (wave + 4 * m) + 4 * q + r, b = r0 + bl;
          const float x = acc[m][nt][r];
          const float y = swish(fmaf((x - mean[nt]) * rstd[nt], ga, be));
          s_hn[bl * PITCH + j] = jon ? y : 0.f;              // zero columns beyond f: they are the second layer's K padding
          if (jon && b < Ball) {
            const size_t o = hslot(b, j);
            hraw[o] = x;
            hn[o] = y;
          }
        }
    }
    __syncthreads();

    // ---- layer 1: o[b][j] = sum_i hn[b][i] L1[j][i] + b1[j] -------------------------------------------------------------
    f32x4 acc2[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc2[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < FP; k0 += 16) {
      f32x4 a[MTW], b[NT];
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[m] = *reinterpret_cast<const f32x4*>(&s_hn[(16 * (wave + 4 * m) + c16) * PITCH + k0 + 4 * q]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = load4v(lrow[nt], k0 + 4 * q, f);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc2[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t], b[nt][t], acc2[m][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int j = 16 * nt + c16;
      if (j >= f) continue;
      const float bias = P.b1[j];
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
          if (b >= row0 && b < row0 + B) {
            const float o = acc2[m][nt][r] + bias;
            film_raw[((((size_t)(b - row0) * KC + kc) * 2 + br) * 2 + wh) * FP + j] = wh == 0 ? eps + expf(o) + pz : o;
          }
        }
    }
  }
}

template <int NT, int MTW>
__global__ __launch_bounds__(256) void film_heads_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ g,
                                                             const float* __restrict__ hraw, const float* __restrict__ hn,
                                                             const float* __restrict__ stats, const float* __restrict__ film_raw,
                                                             const float* __restrict__ g_film_raw, float* __restrict__ g_raw,
                                                             float* __restrict__ dhraw, int KC, int f, int G, int Ball, int row0,
                                                             int B, float eps, int training) {
  constexpr int FP = 16 * NT, PITCH = FP + 4, ROWS = 64 * MTW;
  __shared__ __align__(16) float s_do[ROWS * PITCH];
  __shared__ __align__(16) float s_dh[ROWS * PITCH];
  __shared__ float s_red[4][FP];
  const int h = blockIdx.x, kc = h >> 2, br = (h >> 1) & 1, wh = h & 1, H = 4 * KC;
  const int NRB = (Ball + ROWS - 1) / ROWS;
  const HeadPtrs P = head_of(raw, h, f, G);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  auto hslot = [&](int b, int j) { return ((size_t)b * H + h) * f + j; };

  float ga[NT], be[NT], mu[NT], rs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int i = min(16 * nt + c16, f - 1);
    ga[nt] = P.bn[i];
    be[nt] = P.bn[f + i];
    mu[nt] = training ? stats[(size_t)h * f + i] : P.bn[2 * f + i];
    rs[nt] = training ? stats[(size_t)(2 * H + h) * f + i] : 1.0f / sqrtf(P.bn[3 * f + i] + GWTF_BN_EPS);
  }
  const int nsl = 256 / FP < 4 ? 256 / FP : 4, dcol = threadIdx.x % FP, dsl = threadIdx.x / FP;
  float db1 = 0.f;                                 // this thread's share of db1[dcol] over the row blocks
  f32x4 acc[MTW][NT];
  float xh[MTW][NT][4];
  float sb[NT], sg[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) sb[nt] = sg[nt] = 0.f;

  // ---- pass A, per row block: d(out) -> db1, d(hn), dL1 (+)=, Swish backward, the column sums of the BatchNorm backward ------------
#pragma unroll 1
  for (int rb = 0; rb < NRB; ++rb) {
    const int r0 = rb * ROWS, nb = min(ROWS, Ball - r0);
    const int KB = (nb + 15) / 16 * 16;                  // contraction length over the block's rows, in whole k steps
    if (rb) __syncthreads();                             // the previous block's products have read s_do
    // d(out): the upstream gradient of this head's rows (zero outside this rank's rows), through the exp of the scale head
    for (int idx = threadIdx.x; idx < ROWS * FP; idx += 256) {
      const int bl = idx / FP, j = idx - bl * FP, b = r0 + bl;
      float v = 0.f;
      if (b >= row0 && b < row0 + B && j < f) {
        const size_t o = ((((size_t)(b - row0) * KC + kc) * 2 + br) * 2 + wh) * FP + j;
        v = g_film_raw[o];
        if (wh == 0) v *= film_raw[o] - eps;              // a = eps + exp(o): d a / d o = a - eps
      }
      s_do[bl * PITCH + j] = v;
    }
    __syncthreads();
    if (dsl < nsl)                                       // db1[j] = sum_b d(out)[b][j]
      for (int bl = dsl; bl < nb; bl += nsl) db1 += s_do[bl * PITCH + dcol];

    // d(hn)[b][i] = sum_j d(out)[b][j] L1[j][i]
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < FP; k0 += 16) {
      f32x4 a[MTW], b[NT];
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[m] = *reinterpret_cast<const f32x4*>(&s_do[(16 * (wave + 4 * m) + c16) * PITCH + k0 + 4 * q]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = load4s(P.L1 + min(16 * nt + c16, f - 1), k0 + 4 * q, f, (size_t)f);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t], b[nt][t], acc[m][nt], 0, 0, 0);
    }

    // dL1[j][i] (+)= sum_b d(out)[b][j] hn[b][i]: the NT x NT output tiles dealt over the waves
#pragma unroll 1
    for (int id = wave; id < NT * NT; id += 4) {
      const int mt = id / NT, nt = id - mt * NT;
      f32x4 dw = {0.f, 0.f, 0.f, 0.f};
      const float* hcol = hn + hslot(r0, min(16 * nt + c16, f - 1));
#pragma unroll 2
      for (int k0 = 0; k0 < KB; k0 += 16) {
        const f32x4 bv = load4s(hcol, k0 + 4 * q, nb, (size_t)H * f);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          dw = __builtin_amdgcn_mfma_f32_16x16x4f32(s_do[(k0 + 4 * q + t) * PITCH + 16 * mt + c16], bv[t], dw, 0, 0, 0);
      }
      const int i = 16 * nt + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mt + 4 * q + r;
        if (j < f && i < f) {
          float* d = g_raw + P.off_L1 + (size_t)j * f + i;
          *d = rb ? *d + dw[r] : dw[r];
        }
      }
    }

    // Swish backward in the accumulators' layout (lane: column i = 16 nt + c16, rows r0 + 16 (wave + 4 m) + 4 q + r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool ion = 16 * nt + c16 < f;
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
          const bool on = ion && b < Ball;
          const float x = on ? hraw[hslot(b, 16 * nt + c16)] : 0.f;
          const float xn = (x - mu[nt]) * rs[nt];
          const float hb = fmaf(xn, ga[nt], be[nt]);
          const float sgm = 1.0f / (1.0f + expf(-hb));
          const float dh = on ? acc[m][nt][r] * (sgm * (1.0f + hb * (1.0f - sgm))) : 0.f;
          xh[m][nt][r] = xn;
          acc[m][nt][r] = dh;
          sb[nt] += dh;
          sg[nt] = fmaf(dh, xn, sg[nt]);
          if (NRB > 1 && on) dhraw[hslot(b, 16 * nt + c16)] = dh;
        }
    }
  }
  __syncthreads();
  if (dsl < nsl) s_red[dsl][dcol] = db1;
  __syncthreads();
  if (threadIdx.x < f) {
    float s = 0.f;
    for (int u = 0; u < nsl; ++u) s += s_red[u][threadIdx.x];
    g_raw[P.off_b1 + threadIdx.x] = s;
  }
  __syncthreads();
  column_totals<NT>(sb, s_red, wave, c16, q);
  column_totals<NT>(sg, s_red, wave, c16, q);
  if (wave == 0 && q == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int i = 16 * nt + c16;
      if (i < f) {
        g_raw[P.off_bn + i] = sg[nt];          // d gamma
        g_raw[P.off_bn + f + i] = sb[nt];      // d beta
      }
    }
  }

  // ---- pass B, per row block: BatchNorm backward -> d(hraw), dL0 (+)= ---------------------------------------------------------
  const int GT = (G + 15) / 16;
#pragma unroll 1
  for (int rb = 0; rb < NRB; ++rb) {
    const int r0 = rb * ROWS, nb = min(ROWS, Ball - r0);
    const int KB = (nb + 15) / 16 * 16;
    if (rb) __syncthreads();                             // the previous block's products have read s_dh
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int i = 16 * nt + c16;
      const float k = ga[nt] * rs[nt], mb = sb[nt] / (float)Ball, mg = sg[nt] / (float)Ball;
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int bl = 16 * (wave + 4 * m) + 4 * q + r, b = r0 + bl;
          const bool on = i < f && b < Ball;
          float dh = acc[m][nt][r], xn = xh[m][nt][r];
          if (NRB > 1) {
            dh = on ? dhraw[hslot(b, i)] : 0.f;
            xn = on ? (hraw[hslot(b, i)] - mu[nt]) * rs[nt] : 0.f;
          }
          const float dx = training ? k * (dh - mb - xn * mg) : k * dh;
          s_dh[bl * PITCH + i] = on ? dx : 0.f;
          if (on) dhraw[hslot(b, i)] = dx;
        }
    }
    __syncthreads();

    // dL0[j][k] (+)= sum_b d(hraw)[b][j] g[b][k]: NT x ceil(G / 16) output tiles dealt over the waves
#pragma unroll 1
    for (int id = wave; id < NT * GT; id += 4) {
      const int mt = id / GT, nt = id - mt * GT;
      f32x4 dw = {0.f, 0.f, 0.f, 0.f};
      const float* gcol = g + (size_t)r0 * G + min(16 * nt + c16, G - 1);
#pragma unroll 2
      for (int k0 = 0; k0 < KB; k0 += 16) {
        const f32x4 bv = load4s(gcol, k0 + 4 * q, nb, (size_t)G);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          dw = __builtin_amdgcn_mfma_f32_16x16x4f32(s_dh[(k0 + 4 * q + t) * PITCH + 16 * mt + c16], bv[t], dw, 0, 0, 0);
      }
      const int kcol = 16 * nt + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * mt + 4 * q + r;
        if (j < f && kcol < G) {
          float* d = g_raw + P.off_L0 + (size_t)j * G + kcol;
          *d = rb ? *d + dw[r] : dw[r];
        }
      }
    }
  }
}

// dg_part[s][b][k] = sum over the heads h = s, s + S, ... and their columns j of d(hraw)[b][h][j] L0_h[j][k]
// grid (ceil(G / 16), S); wave w owns row tiles w (, w + 4) of each block of 64 MTW rows
template <int MTW>
__global__ __launch_bounds__(256) void film_heads_dg_kernel(const float* __restrict__ raw, const float* __restrict__ dhraw,
                                                            float* __restrict__ dg_part, int KC, int f, int G, int Ball) {
  constexpr int ROWS = 64 * MTW;
  const int H = 4 * KC, S = gridDim.y, s = blockIdx.y, nt = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, q = lane >> 4;
  const int FPK = (f + 15) / 16 * 16;
  const int kcol = min(16 * nt + c16, G - 1);
#pragma unroll 1
  for (int r0 = 0; r0 < Ball; r0 += ROWS) {
    f32x4 acc[MTW];
    const float* arow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      arow[m] = dhraw + (size_t)min(r0 + 16 * (wave + 4 * m) + c16, Ball - 1) * H * f;
    }
#pragma unroll 1
    for (int h = s; h < H; h += S) {
      const HeadPtrs P = head_of(raw, h, f, G);
      const float* bcol = P.L0 + kcol;
#pragma unroll 2
      for (int k0 = 0; k0 < FPK; k0 += 16) {
        const int ks = k0 + 4 * q;
        f32x4 a[MTW];
        const f32x4 bv = load4s(bcol, ks, f, (size_t)G);
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[m] = load4v(arow[m] + (size_t)h * f, ks, f);
#pragma unroll
        for (int m = 0; m < MTW; ++m) a[m] = zero_from(a[m], ks, f);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < MTW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t], bv[t], acc[m], 0, 0, 0);
      }
    }
    const int k = 16 * nt + c16;
    if (k < G) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = r0 + 16 * (wave + 4 * m) + 4 * q + r;
          if (b < Ball) dg_part[((size_t)s * Ball + b) * G + k] = acc[m][r];
        }
    }
  }
}

bool args_ok(int KC, int f, int G, int Ball, int row0, int B) {
  return KC > 0 && f > 0 && f <= GWTF_MAX_FP_TRAIN && G > 0 && Ball > 0 && Ball <= (1 << 16) && row0 >= 0 && B > 0 && row0 + B <= Ball;
}

}  // namespace

extern "C" int gwtf_film_heads_slices(int KC, int G) {
  const int GT = (G + 15) / 16, H = 4 * KC;
  int S = 512 / GT;
  if (S < 1) S = 1;
  return S < H ? S : H;
}

#define GWTF_FH_DISPATCH(KERNEL, ...)                                                                          \
  switch ((gwtf_padded_width(f) / 16) * 2 + (Ball > 64 ? 1 : 0)) {                                            \
    case 2: hipLaunchKernelGGL((KERNEL<1, 1>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 3: hipLaunchKernelGGL((KERNEL<1, 2>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 4: hipLaunchKernelGGL((KERNEL<2, 1>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 5: hipLaunchKernelGGL((KERNEL<2, 2>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 6: hipLaunchKernelGGL((KERNEL<3, 1>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 7: hipLaunchKernelGGL((KERNEL<3, 2>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 8: hipLaunchKernelGGL((KERNEL<4, 1>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 9: hipLaunchKernelGGL((KERNEL<4, 2>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;          \
    case 10: hipLaunchKernelGGL((KERNEL<5, 1>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;         \
    case 11: hipLaunchKernelGGL((KERNEL<5, 2>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;         \
    case 12: hipLaunchKernelGGL((KERNEL<6, 1>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;         \
    case 13: hipLaunchKernelGGL((KERNEL<6, 2>), dim3(4 * KC), dim3(256), 0, st, __VA_ARGS__); break;         \
    default: return GWTF_E_BADARG;                                                                              \
  }

extern "C" int gwtf_film_heads_forward(const float* raw, const float* g, const float* poison, float* hraw, float* hn, float* stats,
                                       float* film_raw, int KC, int f, int G, int Ball, int row0, int B, float eps, int training,
                                       void* stream) {
  if (!raw || !g || !hraw || !hn || !stats || !film_raw || !args_ok(KC, f, G, Ball, row0, B)) return GWTF_E_BADARG;
  if (training && Ball < 2) return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  GWTF_FH_DISPATCH(film_heads_fwd_kernel, raw, g, poison, hraw, hn, stats, film_raw, KC, f, G, Ball, row0, B, eps, training)
  return (int)hipGetLastError();
}

extern "C" int gwtf_film_heads_backward(const float* raw, const float* g, const float* hraw, const float* hn, const float* stats,
                                        const float* film_raw, const float* g_film_raw, float* g_raw, float* dhraw, float* dg_part,
                                        int KC, int f, int G, int Ball, int row0, int B, float eps, int training, void* stream) {
  if (!raw || !g || !hraw || !hn || !stats || !film_raw || !g_film_raw || !g_raw || !dhraw || !dg_part ||
      !args_ok(KC, f, G, Ball, row0, B))
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  GWTF_FH_DISPATCH(film_heads_bwd_kernel, raw, g, hraw, hn, stats, film_raw, g_film_raw, g_raw, dhraw, KC, f, G, Ball, row0, B, eps,
                   training)
  const dim3 grid((unsigned)((G + 15) / 16), (unsigned)gwtf_film_heads_slices(KC, G));
  if (Ball > 64)
    hipLaunchKernelGGL((film_heads_dg_kernel<2>), grid, dim3(256), 0, st, raw, dhraw, dg_part, KC, f, G, Ball);
  else
    hipLaunchKernelGGL((film_heads_dg_kernel<1>), grid, dim3(256), 0, st, raw, dhraw, dg_part, KC, f, G, Ball);
  return (int)hipGetLastError();
}
