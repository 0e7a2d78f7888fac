// gwtf_prior.hip -- the global prior flow on the shape latent: the whole GlobalRNVPDecoder (reference
// lib/networks/decoders.py:7-38: n_flows RealNVPFlowCouple = 2 n_flows elementary RealNVPFlow, flows.py:163-243) as ONE
// launch per direction, forward and backward, eval- and train-mode BatchNorm.
// Round 3: the products run on gwtf_gemm.h's gemm_direct (operands straight from L2 into the MFMA's registers, no LDS staging, no
// barrier inside a product) instead of the LDS-staged version below it in that header: see the timing note at gwtf_prior_forward.
//
// The work is per SHAPE: B <= 128 rows of G latents through 14 flows of two (B x G/2)(G/2 x F) -> BN -> Swish ->
// (B x F)(F x G/2) MLPs -- a few MFLOP, a chain of ~40 dependent steps.  The reference (and round 1 of this repo) runs it as
// ~100 library launches forward and ~200 backward; here one workgroup of 8 wavefronts walks the whole chain: every GEMM on
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate: no precision games for a latency-bound op), operands straight
// from L2 (the 1.8 MB of weights and the (B,G) activations never leave it), intermediates in a small global workspace,
// workgroup barriers + device-scope fences between the phases.  BatchNorm over the B rows (batch statistics in train mode)
// is a per-column loop.  Because the batch statistics couple all rows, the chain cannot be split over workgroups without
// grid-wide barriers -- and at B*G <= 64 K elements there is nothing to split.
//
// Raw parameter arena, per elementary flow j (module order, reference flows.py:174-190), branch 0 = mu, 1 = logvar:
//   W0[F][Gk] | bn.weight[F] | bn.bias[F] | bn.running_mean[F] | bn.running_var[F] | W1[Gw][F] | b1[Gw]
// Gw / Gk = warped / kept latents of the flow (pattern 0: even / odd, pattern 1: first / second half; flows.py:219-243).
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "gwtf_gemm.h"
#include "../../include/gwtf.h"

namespace {

using namespace gwtf_gemm;
constexpr int kMaxFlows = 64;

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

// Hidden layer of one flow, both branches: Hpre = kept . W0^T -> BatchNorm over the B rows -> Swish.
// hws [B][2][F]: hidden activations h;  xhat [B][2][F] (may be null): normalised pre-activations;  stat_used [2][2][F] =
// {mean, biased var} actually used (train: of the batch, also written to bn_stats_j when given; eval: running statistics).
template <bool TRAIN>
__device__ __forceinline__ void hidden_layer(const Plan& P, const Geom& ge, const float* __restrict__ raw, const float* __restrict__ gin,
                                             float* __restrict__ hws, float* __restrict__ xhat, float* __restrict__ stat_used,
                                             float* __restrict__ bn_stats_j, float* __restrict__ As, float* __restrict__ Bs) {
  const int B = P.B, F = P.F, G = P.G;
  for (int x = 0; x < 2; ++x) {
    const Branch br = branch_of(raw, ge, F, x);
    gemm_direct(B, F, ge.Gk, gin + ge.koff, G, ge.kstride, br.W0, ge.Gk, 1, hws + (size_t)x * F, 2 * F, false);
  }
  phase_sync();
  {
    const ColMap cm = col_map(2 * F);
    const int x = cm.col / F, f = cm.col % F;
    const Branch br = branch_of(raw, ge, F, x);
    auto at = [&](int b, int col) -> float& { return hws[(size_t)b * 2 * F + col]; };     // [b][x][f], col = x*F + f
    float mean, var;
    if (TRAIN) {
      mean = col_sum(cm, 2 * F, B, [&](int b, int col) { return at(b, col); }, As) / (float)B;
      var = col_sum(cm, 2 * F, B, [&](int b, int col) { const float d = at(b, col) - mean; return d * d; }, As) / (float)B;
      if (bn_stats_j && cm.on && cm.rg == 0) { bn_stats_j[(x * 2 + 0) * F + f] = mean; bn_stats_j[(x * 2 + 1) * F + f] = var; }
    } else {
      mean = br.rm[f];
      var = br.rv[f];
    }
    if (cm.on) {
      if (stat_used && cm.rg == 0) { stat_used[(x * 2 + 0) * F + f] = mean; stat_used[(x * 2 + 1) * F + f] = var; }
      const float isd = 1.0f / sqrtf(var + GWTF_BN_EPS), ga = br.gamma[f], be = br.beta[f];
      for (int b = cm.rg; b < B; b += cm.RG) {
        const float xh = (at(b, cm.col) - mean) * isd;
        if (xhat) xhat[(size_t)b * 2 * F + cm.col] = xh;
        at(b, cm.col) = swishf(fmaf(xh, ga, be));
      }
    }
  }
  phase_sync();
}

// forward workspace: H [B][2][F] | O [B][2][Gwmax]
template <bool TRAIN>
__global__ __launch_bounds__(kThreads) void prior_fwd_kernel(const Plan P, const float* __restrict__ g0, const float* __restrict__ raw,
                                                             float* __restrict__ gs, float* __restrict__ mus,
                                                             float* __restrict__ lvs, float* __restrict__ ws,
                                                             float* __restrict__ bn_stats) {
  __shared__ float As[kThreads];
  float* Bs = nullptr;
  const int B = P.B, G = P.G, F = P.F;
  const size_t BG = (size_t)B * G;
  const int Gwmax = (G + 1) / 2;
  float* hws = ws;
  float* O = hws + (size_t)B * 2 * F;
  const float* cur = g0;
  for (int step = 0; step < P.n2; ++step) {
    const int j = P.mode == GWTF_MODE_DIRECT ? step : P.n2 - 1 - step;
    const Geom ge = geom_at(j, G, F);
    hidden_layer<TRAIN>(P, ge, raw, cur, hws, nullptr, nullptr, bn_stats ? bn_stats + (size_t)j * 4 * F : nullptr, As, Bs);
    float* gout = gs + j * BG;
    float* mo = mus + j * BG;
    float* lo = lvs + j * BG;
    const Branch bm = branch_of(raw, ge, F, 0), bl = branch_of(raw, ge, F, 1);
    gemm_direct(B, ge.Gw, F, hws, 2 * F, 1, bm.W1, F, 1, O, 2 * Gwmax, false);
    gemm_direct(B, ge.Gw, F, hws + F, 2 * F, 1, bl.W1, F, 1, O + Gwmax, 2 * Gwmax, false);
    phase_sync();
    // the affine map on the warped latents; kept latents pass through (mu = logvar = 0 there: exp(0) * g + 0)
    for (int t = threadIdx.x; t < B * G; t += kThreads) {
      const int b = t / G, gi = t % G;
      const int rel = gi - ge.woff;
      const bool warped = ge.wstride == 2 ? (rel >= 0 && (rel & 1) == 0) : (rel >= 0 && rel < ge.Gw);
      const float x0 = cur[t];
      float y = x0, mu = 0.f, lv = 0.f;
      if (warped) {
        const int w = ge.wstride == 2 ? rel >> 1 : rel;
        mu = O[((size_t)b * 2 + 0) * Gwmax + w] + bm.b1[w];
        lv = logf(P.eps + expf(O[((size_t)b * 2 + 1) * Gwmax + w] + bl.b1[w]));                             // flows.py:198-201
        y = P.mode == GWTF_MODE_DIRECT ? expf(0.5f * lv) * x0 + mu : expf(-0.5f * lv) * (x0 - mu);          // :206-209
      }
      gout[t] = y;
      mo[t] = mu;
      lo[t] = lv;
    }
    phase_sync();
    cur = gout;
  }
}

// Backward of the whole stack.  Upstream gradients enter through every list slot: Ggs / Glvs [n2][B][G] (dL/d gs[j],
// dL/d logvars[j]; either may be null).  ws: dO [B][2][Gwmax] | H [B][2][F] | XH [B][2][F] | DH [B][2][F] | ST [2][2][F] |
// GA [B][G] | GB [B][G] | DK [B][Gwmax].  g_raw receives every parameter gradient (the BatchNorm buffers' slots stay zero);
// dg0 = dL/d input.
template <bool TRAIN>
__global__ __launch_bounds__(kThreads) void prior_bwd_kernel(const Plan P, const float* __restrict__ g0, const float* __restrict__ raw,
                                                             const float* __restrict__ gs, const float* __restrict__ mus,
                                                             const float* __restrict__ lvs, const float* __restrict__ Ggs,
                                                             const float* __restrict__ Glvs, float* __restrict__ ws,
                                                             float* __restrict__ g_raw, float* __restrict__ dg0, int Gwmax) {
  __shared__ float As[kThreads];
  float* Bs = nullptr;
  const int B = P.B, G = P.G, F = P.F;
  const size_t BG = (size_t)B * G;
  float* dO = ws;
  float* H = dO + (size_t)B * 2 * Gwmax;
  float* XH = H + (size_t)B * 2 * F;
  float* DH = XH + (size_t)B * 2 * F;
  float* ST = DH + (size_t)B * 2 * F;
  float* GA = ST + 4 * F;
  float* GB = GA + BG;
  float* DK = GB + BG;
  for (int t = threadIdx.x; t < (int)BG; t += kThreads) GA[t] = 0.f;      // everything enters through the list slots
  phase_sync();
  float* Gcur = GA;
  float* Gnext = GB;
#ifdef GWTF_DBG_PRIOR_STAMPS
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(g_raw + geom_at(P.n2 - 1, G, F).raw + (size_t)F * geom_at(P.n2 - 1, G, F).Gk + 2 * F);
  int n_st = 0;
#define GWTF_STAMP() do { if (step == 1 && threadIdx.x == 0) stamps[n_st++] = wall_clock64(); } while (0)
#else
#define GWTF_STAMP() do {} while (0)
#endif
  for (int step = 0; step < P.n2; ++step) {
    const int j = P.mode == GWTF_MODE_DIRECT ? P.n2 - 1 - step : step;       // reverse of the forward's processing order
    const Geom ge = geom_at(j, G, F);
    const int jprev = P.mode == GWTF_MODE_DIRECT ? j - 1 : j + 1;           // the flow whose output this flow read
    const float* xin = (jprev < 0 || jprev >= P.n2) ? g0 : gs + (size_t)jprev * BG;
    const float* gout = gs + (size_t)j * BG;
    const float* mu_j = mus + (size_t)j * BG;
    const float* lv_j = lvs + (size_t)j * BG;
    const Branch bm = branch_of(raw, ge, F, 0), bl = branch_of(raw, ge, F, 1);
    float* gr = g_raw + ge.raw;
    const size_t BF = branch_floats(F, ge.Gw, ge.Gk);
    GWTF_STAMP();
    // ---- B1: through the affine map: dO (both branches), the warped part of the next flowing gradient
    for (int t = threadIdx.x; t < B * ge.Gw; t += kThreads) {
      const int b = t / ge.Gw, w = t % ge.Gw;
      const size_t o = (size_t)b * G + ge.woff + ge.wstride * w;
      const float go = Gcur[o] + (Ggs ? Ggs[(size_t)j * BG + o] : 0.f);
      const float lv = lv_j[o];
      float dx, dmu, dlv;
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
      dO[((size_t)b * 2 + 0) * Gwmax + w] = dmu;
      dO[((size_t)b * 2 + 1) * Gwmax + w] = dlv * (1.0f - P.eps * expf(-lv));     // lv = log(eps + e^o): dlv/do = e^o / (eps + e^o)
      Gnext[o] = dx;
    }
    GWTF_STAMP();
    // ---- B2: recompute the hidden layer (H, XH, statistics used); its barriers also cover B1's stores
    hidden_layer<TRAIN>(P, ge, raw, xin, H, XH, ST, nullptr, As, Bs);
    GWTF_STAMP();
    // ---- B3: dH = dO . W1 ; dW1 = dO^T . H ; db1 = column sums of dO
    for (int x = 0; x < 2; ++x) {
      const float* W1 = x == 0 ? bm.W1 : bl.W1;
      gemm_direct(B, F, ge.Gw, dO + (size_t)x * Gwmax, 2 * Gwmax, 1, W1, 1, F, DH + (size_t)x * F, 2 * F, false);
      float* dW1 = gr + (size_t)x * BF + (size_t)F * ge.Gk + 4 * (size_t)F;
      for (int w0 = 0; w0 < ge.Gw; w0 += kMaxM) {
        const int wn = ge.Gw - w0 < kMaxM ? ge.Gw - w0 : kMaxM;
        gemm_direct(wn, F, B, dO + (size_t)x * Gwmax + w0, 1, 2 * Gwmax, H + (size_t)x * F, 1, 2 * F, dW1 + (size_t)w0 * F, F, false);
      }
      for (int w0 = 0; w0 < ge.Gw; w0 += 256) {
        const int wn = ge.Gw - w0 < 256 ? ge.Gw - w0 : 256;
        const ColMap cm = col_map(wn);
        const float sdo = col_sum(cm, wn, B, [&](int b, int col) { return dO[((size_t)b * 2 + x) * Gwmax + w0 + col]; }, As);
        if (cm.on && cm.rg == 0) gr[(size_t)x * BF + (size_t)F * ge.Gk + 4 * (size_t)F + (size_t)ge.Gw * F + w0 + cm.col] = sdo;
      }
    }
    phase_sync();
    GWTF_STAMP();
    // ---- B4: through Swish and BatchNorm: DH <- dL/dHpre; dgamma, dbeta
    {
      const ColMap cm = col_map(2 * F);
      const int x = cm.col / F, f = cm.col % F;
      const Branch br = x == 0 ? bm : bl;
      const float ga = br.gamma[f], be = br.beta[f];
      const float isd = 1.0f / sqrtf(ST[(x * 2 + 1) * F + f] + GWTF_BN_EPS);
      if (cm.on) {
        for (int b = cm.rg; b < B; b += cm.RG) {
          const size_t o = (size_t)b * 2 * F + cm.col;
          const float hb = fmaf(XH[o], ga, be);
          const float sg = 1.0f / (1.0f + expf(-hb));
          DH[o] *= sg * (1.0f + hb * (1.0f - sg));                       // d swish (own rows: read back by the same thread)
        }
      }
      const float sdy = col_sum(cm, 2 * F, B, [&](int b, int col) { return DH[(size_t)b * 2 * F + col]; }, As);
      const float sdyx = col_sum(cm, 2 * F, B, [&](int b, int col) { return DH[(size_t)b * 2 * F + col] * XH[(size_t)b * 2 * F + col]; }, As);
      if (cm.on) {
        if (cm.rg == 0) {
          float* gb = gr + (size_t)x * BF + (size_t)F * ge.Gk;
          gb[f] = sdyx;            // d gamma
          gb[F + f] = sdy;         // d beta
        }
        const float m1 = TRAIN ? sdy / (float)B : 0.f, m2 = TRAIN ? sdyx / (float)B : 0.f;
        for (int b = cm.rg; b < B; b += cm.RG) {
          const size_t o = (size_t)b * 2 * F + cm.col;
          DH[o] = ga * isd * (DH[o] - m1 - XH[o] * m2);                  // eval: m1 = m2 = 0 (statistics are constants)
        }
      }
    }
    phase_sync();
    GWTF_STAMP();
    // ---- B5: dkept = dHpre . W0 (both branches summed) ; dW0 = dHpre^T . kept
    for (int x = 0; x < 2; ++x) {
      const float* W0 = x == 0 ? bm.W0 : bl.W0;
      gemm_direct(B, ge.Gk, F, DH + (size_t)x * F, 2 * F, 1, W0, 1, ge.Gk, DK, ge.Gk, x == 1);   // same lane wrote x == 0
      gemm_direct(F, ge.Gk, B, DH + (size_t)x * F, 1, 2 * F, xin + ge.koff, ge.kstride, G, gr + (size_t)x * BF, ge.Gk, false);
    }
    phase_sync();
    for (int t = threadIdx.x; t < B * ge.Gk; t += kThreads) {
      const int b = t / ge.Gk, kk = t % ge.Gk;
      const size_t o = (size_t)b * G + ge.koff + ge.kstride * kk;
      Gnext[o] = Gcur[o] + (Ggs ? Ggs[(size_t)j * BG + o] : 0.f) + DK[(size_t)b * ge.Gk + kk];
    }
    phase_sync();
    GWTF_STAMP();
    float* tmp = Gcur; Gcur = Gnext; Gnext = tmp;
  }
  for (int t = threadIdx.x; t < (int)BG; t += kThreads) dg0[t] = Gcur[t];
}

bool make_plan(Plan& P, int n_flows, int B, int G, int F, int mode, float eps) {
  if (n_flows <= 0 || 2 * n_flows > kMaxFlows || B <= 0 || B > kMaxM || G < 2 || G > 2 * kMaxN || F <= 0 || F > kMaxM) return false;
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
  const size_t Gwmax = (size_t)(G + 1) / 2;
  return (size_t)B * 2 * Gwmax + 3 * (size_t)B * 2 * F + 4 * (size_t)F + 2 * (size_t)B * G + (size_t)B * Gwmax;
}

extern "C" int gwtf_prior_forward(const float* g, const float* raw, float* gs, float* mus, float* logvars, float* workspace,
                                  float* bn_stats, int n_flows, int B, int G, int F, float eps, int mode, int training,
                                  void* stream) {
  Plan P;
  if (!g || !raw || !gs || !mus || !logvars || !workspace || (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE) ||
      !make_plan(P, n_flows, B, G, F, mode, eps) || (training && B < 2))
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (training) hipLaunchKernelGGL(prior_fwd_kernel<true>, dim3(1), dim3(kThreads), 0, st, P, g, raw, gs, mus, logvars, workspace, bn_stats);
  else hipLaunchKernelGGL(prior_fwd_kernel<false>, dim3(1), dim3(kThreads), 0, st, P, g, raw, gs, mus, logvars, workspace, bn_stats);
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
  const int Gwmax = (G + 1) / 2;
  if (training) hipLaunchKernelGGL(prior_bwd_kernel<true>, dim3(1), dim3(kThreads), 0, st, P, g, raw, gs, mus, logvars, g_gs, g_logvars, workspace, g_raw, g_g, Gwmax);
  else hipLaunchKernelGGL(prior_bwd_kernel<false>, dim3(1), dim3(kThreads), 0, st, P, g, raw, gs, mus, logvars, g_gs, g_logvars, workspace, g_raw, g_g, Gwmax);
  return (int)hipGetLastError();
}
