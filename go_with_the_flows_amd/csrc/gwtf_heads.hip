// gwtf_heads.hip -- the per-SHAPE MLP heads around the point-flow decoder: FeatureEncoder / WeightsEncoder (reference
// lib/networks/encoders.py:31-89): n_layers x [Linear(no bias) -> BatchNorm1d -> Swish] then Linear(bias) heads (mu, logvar; the
// mixture-weight head ends in a log-softmax, :87-91).  In the training step they are g_posterior (512 -> 512 -> 128 + 128),
// p_prior (G -> G -> 3 + 3) and mixture_weights_encoder (3 x (G -> G) -> K): B <= 128 rows, a few MFLOP each, but ~120 small
// library launches per step (Linear, batch_norm, sigmoid, mul, ... and their autograd) on the critical path between the
// encoder and the decoders.
//
// ONE LAYER per launch, forward and backward:
//   forward : y = x W^T (+ bias) -> [BatchNorm over the B rows: batch statistics, running statistics updated `bn_updates` times,
//             or running statistics] -> [Swish | log-softmax].  A workgroup owns 64 output COLUMNS and all rows: everything
//             BatchNorm needs (column statistics) is local to it -- no grid-wide barrier; Dout / 64 workgroups.
//   backward: pass 1 (per block of 64 output columns): dL/dy through the activation and the BatchNorm (column sums again local),
//             dW = dy^T x, dbias, dgamma, dbeta;   pass 2 (per block of 64 INPUT columns): dL/dx = dy W.
// GEMMs: exact fp32 MFMA with operands straight from L2 (gemm_direct below); column sums: gwtf_gemm.h.  In a data-parallel run the caller gathers the rows of all
// ranks first (dist.gather_rows): the kernels always see the whole batch -- SyncBatchNorm semantics without a collective inside.
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "gwtf_gemm.h"
#include "../../include/gwtf.h"

namespace {

using namespace gwtf_gemm;
constexpr int kCols = 64;            // output (pass 1) / input (pass 2) columns per workgroup

struct HeadDims { int B, Din, Dout; };

// C (M x N) [+]= A (M x K) . B^T (N x K), M <= 128, N <= 64, for the whole workgroup, operands read STRAIGHT from global memory /
// L2 into the MFMA's registers (v_mfma_f32_16x16x4_f32: exact fp32 products) -- no LDS staging, no barriers: the staged version
// of gwtf_gemm.h (built for the prior flow's chain of tiny dependent products) spent ~6 us per 64-wide K chunk on scalar
// staging loads here (g_posterior forward + backward 550 us against 150 us for the library path; tools/diag/heads_time.py).
// A(i, k) at A + i*sai + k*sak, B(j, k) at Bm + j*sbj + k*sbk.  A K step covers 16 k values: lane (r = lane & 15, q = lane >> 4)
// holds k = k0 + 4q .. 4q+3 of row r -- ONE 16-byte load when k is the contiguous index (sak == 1, everything 16-byte aligned),
// four 4-byte loads otherwise (coalesced over the 16 rows when the row index is the contiguous one) -- and MFMA j of the step
// consumes element j: both operands use the same k order, which is all a contraction needs.
// Wave w owns column tile w & 3 and the row tiles (w >> 2), (w >> 2) + 2, ...: up to 4 accumulators.
// Loads carry NO lane-dependent control flow and no select between them (a select after a load is a wait for THAT load: the
// first version serialised its 12 loads per pass that way, 57 us per launch): rows beyond M / N are read from a clamped
// (valid) address and simply never stored -- a row of A or B only feeds its own output row / column -- and only the K tail,
// which does feed valid outputs, is zeroed, after all loads of the pass have been issued.
template <bool VEC>
__device__ __forceinline__ f32x4 load_k4_full(const float* __restrict__ rowp, long sk, int k) {
  f32x4 v;
  if (VEC) {
    v = *reinterpret_cast<const f32x4*>(rowp + k);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rowp[(long)(k + j) * sk];
  }
  return v;
}
__device__ __forceinline__ f32x4 load_k4_tail(const float* __restrict__ rowp, long sk, int k, int K) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = rowp[(long)min(k + j, K - 1) * sk];
  return v;
}
__device__ __forceinline__ f32x4 zero_tail(f32x4 v, int k, int K) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = k + j < K ? v[j] : 0.f;
  return v;
}
template <bool AVEC, bool BVEC, int CNT>
__device__ __forceinline__ void gemm_wave(int M, int N, int K, const float* __restrict__ A, long sai, long sak,
                                          const float* __restrict__ Bm, long sbj, long sbk, float* __restrict__ C, long ldc,
                                          bool accumulate, int nt, int m0, int r16, int q) {
  constexpr int U = 4;                                         // a pass = 4 steps of 16 k: one memory round trip per 64 k
  f32x4 acc[CNT];
  const float* arow[CNT];
#pragma unroll
  for (int u = 0; u < CNT; ++u) {
    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    arow[u] = A + (long)min(16 * (m0 + 2 * u) + r16, M - 1) * sai;
  }
  const float* brow = Bm + (long)min(16 * nt + r16, N - 1) * sbj;
  int k0 = 0;
  // (Measured and rejected: issuing pass p + 1's loads before pass p's MFMAs -- two passes of operands in registers -- made
  // every product SLOWER, 255 -> 344 us for the g_posterior module: hipcc serialises the doubled register set.)
#pragma unroll 1
  for (; k0 + 16 * U <= K; k0 += 16 * U) {
    f32x4 a[U][CNT], b[U];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      b[s] = load_k4_full<BVEC>(brow, sbk, ks);
#pragma unroll
      for (int u = 0; u < CNT; ++u) a[s][u] = load_k4_full<AVEC>(arow[u], sak, ks);
    }
#pragma unroll
    for (int s = 0; s < U; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < CNT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][u][j], b[s][j], acc[u], 0, 0, 0);
  }
  if (k0 < K) {                                                // the K tail (wave-uniform branch): clamped loads, then zeros
    f32x4 ta[U][CNT], tb[U];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      tb[s] = load_k4_tail(brow, sbk, ks, K);
#pragma unroll
      for (int u = 0; u < CNT; ++u) ta[s][u] = load_k4_tail(arow[u], sak, ks, K);
    }
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      tb[s] = zero_tail(tb[s], ks, K);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < CNT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][u][j], tb[s][j], acc[u], 0, 0, 0);
    }
  }
  const int n = 16 * nt + r16;
  if (n < N) {
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * (m0 + 2 * u) + 4 * q + r;
        if (m < M) {
          const long o = (long)m * ldc + n;
          C[o] = accumulate ? C[o] + acc[u][r] : acc[u][r];
        }
      }
    }
  }
}
template <bool AVEC, bool BVEC>
__device__ __forceinline__ void gemm_direct_t(int M, int N, int K, const float* __restrict__ A, long sai, long sak,
                                              const float* __restrict__ Bm, long sbj, long sbk, float* __restrict__ C, long ldc,
                                              bool accumulate) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
  const int MT = (M + 15) / 16, NT = (N + 15) / 16;
  const int nt = wave & 3, m0 = wave >> 2;                      // wave w: column tile w & 3, row tiles (w >> 2), (w >> 2) + 2, ...
  if (nt >= NT || m0 >= MT) return;
  switch ((MT - m0 + 1) / 2) {                                 // row tiles of this wave: wave-uniform
    case 1: gemm_wave<AVEC, BVEC, 1>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate, nt, m0, r16, q); break;
    case 2: gemm_wave<AVEC, BVEC, 2>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate, nt, m0, r16, q); break;
    case 3: gemm_wave<AVEC, BVEC, 3>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate, nt, m0, r16, q); break;
    default: gemm_wave<AVEC, BVEC, 4>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate, nt, m0, r16, q); break;
  }
}
__device__ __forceinline__ void gemm_direct(int M, int N, int K, const float* __restrict__ A, long sai, long sak,
                                            const float* __restrict__ Bm, long sbj, long sbk, float* __restrict__ C, long ldc,
                                            bool accumulate) {
  // 16-byte loads where k is the contiguous index and every row starts 16-byte aligned (wave-uniform decision)
  const bool avec = sak == 1 && (sai & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
  const bool bvec = sbk == 1 && (sbj & 3) == 0 && (reinterpret_cast<size_t>(Bm) & 15) == 0;
  if (avec && bvec) gemm_direct_t<true, true>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate);
  else if (avec) gemm_direct_t<true, false>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate);
  else gemm_direct_t<false, false>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate);
}

// s = BatchNorm(y) for a column given its statistics
__device__ __forceinline__ float bn_apply(float y, float mean, float rstd, float ga, float be) { return fmaf((y - mean) * rstd, ga, be); }

// ---- forward ----------------------------------------------------------------------------------------------------------------
// ypre [B][Dout]: the Linear's output (the BatchNorm input; kept for the backward), stats [3][Dout] = mean, biased var, rstd actually
// used, out [B][Dout].  bn_mode 0: no BatchNorm; 1: batch statistics (+ running update, momentum form, unbiased variance, applied
// bn_updates times: the reference evaluates p_prior once per mixture component on the same batch, models.py:169-193 inside
// flow_mixture.py:163-166); 2: running statistics.  act 0: none, 1: swish, 2: log-softmax over the Dout (<= 64) columns.
__global__ __launch_bounds__(kThreads) void head_fwd_kernel(HeadDims d, const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, long long* __restrict__ nbt, float momentum,
                                                            float bn_eps, int bn_mode, int bn_updates, int act,
                                                            float* __restrict__ ypre, float* __restrict__ stats,
                                                            float* __restrict__ out) {
  __shared__ float red[kThreads];
  const int B = d.B, Din = d.Din, Dout = d.Dout;
  const int c0 = blockIdx.x * kCols, nc = min(kCols, Dout - c0);
  gemm_direct(B, nc, Din, x, Din, 1, W + (size_t)c0 * Din, Din, 1, ypre + c0, Dout, false);
  phase_sync();
  const ColMap cm = col_map(kCols);
  const int col = c0 + cm.col;
  const bool live = cm.on && cm.col < nc;
  auto yat = [&](int b, int c) { return c < nc ? ypre[(size_t)b * Dout + c0 + c] + (bias ? bias[c0 + c] : 0.f) : 0.f; };
  float mean = 0.f, var = 1.f, rstd = 1.f, ga = 1.f, be = 0.f;
  if (bn_mode == 1) {
    mean = col_sum(cm, kCols, B, [&](int b, int c) { return yat(b, c); }, red) / (float)B;
    var = col_sum(cm, kCols, B, [&](int b, int c) { const float t = yat(b, c) - mean; return t * t; }, red) / (float)B;
  } else if (bn_mode == 2 && live) {
    mean = rmean[col];
    var = rvar[col];
  }
  if (bn_mode != 0 && live) {
    rstd = 1.0f / sqrtf(var + bn_eps);
    ga = gamma ? gamma[col] : 1.f;
    be = beta ? beta[col] : 0.f;
    if (cm.rg == 0) {
      stats[col] = mean; stats[Dout + col] = var; stats[2 * Dout + col] = rstd;
      if (bn_mode == 1 && rmean) {
        // running = (1 - m) running + m batch, `bn_updates` times with the same batch statistics: keep = (1 - m)^n
        const float unb = var * ((float)B / fmaxf((float)B - 1.0f, 1.0f));
        float keep = 1.0f;
        for (int i = 0; i < bn_updates; ++i) keep *= 1.0f - momentum;
        rmean[col] = keep * rmean[col] + (1.0f - keep) * mean;
        rvar[col] = keep * rvar[col] + (1.0f - keep) * unb;
      }
    }
  }
  if (bn_mode == 1 && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += bn_updates;
  if (act == 2) {
    // log-softmax over the columns of a row (all in this workgroup: Dout <= 64): thread t < B owns row t
    for (int b = threadIdx.x; b < B; b += kThreads) {
      float mx = -3.0e38f;
      for (int c = 0; c < nc; ++c) mx = fmaxf(mx, yat(b, c));
      float se = 0.f;
      for (int c = 0; c < nc; ++c) se += expf(yat(b, c) - mx);
      const float lse = mx + logf(se);
      for (int c = 0; c < nc; ++c) out[(size_t)b * Dout + c0 + c] = yat(b, c) - lse;
    }
    return;
  }
  if (live) {
    for (int b = cm.rg; b < B; b += cm.RG) {
      float s = yat(b, cm.col);
      if (bn_mode != 0) s = bn_apply(s, mean, rstd, ga, be);
      out[(size_t)b * Dout + col] = act == 1 ? swishf(s) : s;
    }
  }
}

// ---- backward, pass 1: per block of 64 output columns ---------------------------------------------------------------------------
// g_out [B][Dout] = dL/d out;  writes g_y [B][Dout] = dL/d(x W^T) (scratch for pass 2), g_W [Dout][Din], g_bias / g_gamma / g_beta
// [Dout] (each may be null).
__global__ __launch_bounds__(kThreads) void head_bwd1_kernel(HeadDims d, const float* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ ypre, const float* __restrict__ stats,
                                                             const float* __restrict__ out, const float* __restrict__ g_out,
                                                             int bn_mode, int act, float* __restrict__ g_y, float* __restrict__ g_W,
                                                             float* __restrict__ g_bias, float* __restrict__ g_gamma,
                                                             float* __restrict__ g_beta) {
  __shared__ float red[kThreads];
  const int B = d.B, Din = d.Din, Dout = d.Dout;
  const int c0 = blockIdx.x * kCols, nc = min(kCols, Dout - c0);
  const ColMap cm = col_map(kCols);
  const int col = c0 + cm.col;
  const bool live = cm.on && cm.col < nc;
  float mean = 0.f, rstd = 1.f, ga = 1.f, be = 0.f, bi = 0.f;
  if (live) {
    if (bn_mode != 0) { mean = stats[col]; rstd = stats[2 * Dout + col]; ga = gamma ? gamma[col] : 1.f; be = beta ? beta[col] : 0.f; }
    bi = bias ? bias[col] : 0.f;
  }
  if (act == 2) {
    // log-softmax: dL/dy = g - softmax(y) * sum_cols g, softmax = exp(out); rows are independent
    for (int b = threadIdx.x; b < B; b += kThreads) {
      float sg = 0.f;
      for (int c = 0; c < nc; ++c) sg += g_out[(size_t)b * Dout + c0 + c];
      for (int c = 0; c < nc; ++c) {
        const size_t o = (size_t)b * Dout + c0 + c;
        g_y[o] = g_out[o] - expf(out[o]) * sg;
      }
    }
    phase_sync();
  }
  // ds = dL/d(BatchNorm output) per element, recomputed wherever it is needed (two column sums + the final pass)
  auto ds_at = [&](int b, int c, float mean_c, float rstd_c, float ga_c, float be_c, float bi_c) {
    if (c >= nc) return 0.f;
    const size_t o = (size_t)b * Dout + c0 + c;
    if (act == 2) return g_y[o];
    const float g = g_out[o];
    if (act == 0) return g;
    float s = ypre[o] + bi_c;
    if (bn_mode != 0) s = bn_apply(s, mean_c, rstd_c, ga_c, be_c);
    const float sg = 1.0f / (1.0f + expf(-s));
    return g * sg * (1.0f + s * (1.0f - sg));                 // d swish / ds
  };
  float sum_ds = 0.f, sum_dsx = 0.f;
  if (bn_mode != 0 || g_bias) {
    sum_ds = col_sum(cm, kCols, B, [&](int b, int c) { return ds_at(b, c, mean, rstd, ga, be, bi); }, red);
    if (bn_mode != 0)
      sum_dsx = col_sum(cm, kCols, B, [&](int b, int c) {
        return c < nc ? ds_at(b, c, mean, rstd, ga, be, bi) * ((ypre[(size_t)b * Dout + c0 + c] + bi - mean) * rstd) : 0.f; }, red);
  }
  if (live && cm.rg == 0) {
    if (bn_mode != 0) {
      if (g_gamma) g_gamma[col] = sum_dsx;
      if (g_beta) g_beta[col] = sum_ds;
    }
    if (g_bias) g_bias[col] = bn_mode == 1 ? 0.f : (bn_mode == 2 ? ga * rstd * sum_ds : sum_ds);   // a bias before a batch-statistic BatchNorm has no gradient
  }
  if (live) {
    const float invB = 1.0f / (float)B;
    for (int b = cm.rg; b < B; b += cm.RG) {
      const size_t o = (size_t)b * Dout + col;
      float gy = ds_at(b, cm.col, mean, rstd, ga, be, bi);
      if (bn_mode == 1) {
        const float xh = (ypre[o] + bi - mean) * rstd;
        gy = ga * rstd * (gy - invB * sum_ds - xh * invB * sum_dsx);
      } else if (bn_mode == 2) {
        gy = ga * rstd * gy;
      }
      g_y[o] = gy;
    }
  }
  phase_sync();
  // dW[c0 .. c0 + nc)[:] = g_y[:, block]^T x : M = nc, K = B, N = Din in chunks of 64
  if (g_W)
    for (int n0 = 0; n0 < Din; n0 += kCols)
      gemm_direct(nc, min(kCols, Din - n0), B, g_y + c0, 1, Dout, x + n0, 1, Din, g_W + (size_t)c0 * Din + n0, Din, false);
}

// ---- backward, pass 2: per block of 64 input columns: g_x[:, block] (+)= g_y W[:, block] ------------------------------------------
__global__ __launch_bounds__(kThreads) void head_bwd2_kernel(HeadDims d, const float* __restrict__ W, const float* __restrict__ g_y,
                                                             float* __restrict__ g_x, int accumulate) {
  const int c0 = blockIdx.x * kCols, nc = min(kCols, d.Din - c0);
  gemm_direct(d.B, nc, d.Dout, g_y, d.Dout, 1, W + c0, 1, d.Din, g_x + c0, d.Din, accumulate != 0);
}

bool dims_ok(int B, int Din, int Dout) { return B >= 1 && B <= kMaxM && Din >= 1 && Dout >= 1 && Din <= 4096 && Dout <= 4096; }

}  // namespace

extern "C" int gwtf_head_layer_supported(int B, int Din, int Dout, int act) {
  return dims_ok(B, Din, Dout) && (act != 2 || Dout <= kCols) ? 1 : 0;
}

extern "C" int gwtf_head_layer_forward(const float* x, const float* W, const float* bias, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                       float bn_eps, int bn_mode, int bn_updates, int act, float* ypre, float* stats, float* out,
                                       int B, int Din, int Dout, void* stream) {
  if (!x || !W || !ypre || !out || !dims_ok(B, Din, Dout) || bn_mode < 0 || bn_mode > 2 || act < 0 || act > 2 ||
      (bn_mode != 0 && !stats) || (bn_mode == 2 && (!running_mean || !running_var)) || (act == 2 && Dout > kCols) ||
      (bn_mode == 1 && B < 2) || bn_updates < 1)
    return GWTF_E_BADARG;
  const HeadDims d = {B, Din, Dout};
  hipLaunchKernelGGL(head_fwd_kernel, dim3((Dout + kCols - 1) / kCols), dim3(kThreads), 0, (hipStream_t)stream, d, x, W, bias, gamma,
                     beta, running_mean, running_var, num_batches_tracked, momentum, bn_eps, bn_mode, bn_updates, act, ypre, stats,
                     out);
  return (int)hipGetLastError();
}

extern "C" int gwtf_head_layer_backward(const float* x, const float* W, const float* bias, const float* gamma, const float* beta,
                                        const float* ypre, const float* stats, const float* out, const float* g_out, int bn_mode,
                                        int act, float* g_y, float* g_x, int accumulate_g_x, float* g_W, float* g_bias,
                                        float* g_gamma, float* g_beta, int B, int Din, int Dout, void* stream) {
  if (!x || !W || !ypre || !out || !g_out || !g_y || !dims_ok(B, Din, Dout) || bn_mode < 0 || bn_mode > 2 || act < 0 || act > 2 ||
      (bn_mode != 0 && !stats) || (act == 2 && Dout > kCols))
    return GWTF_E_BADARG;
  const HeadDims d = {B, Din, Dout};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(head_bwd1_kernel, dim3((Dout + kCols - 1) / kCols), dim3(kThreads), 0, st, d, x, bias, gamma, beta, ypre, stats,
                     out, g_out, bn_mode, act, g_y, g_W, g_bias, g_gamma, g_beta);
  if (g_x)
    hipLaunchKernelGGL(head_bwd2_kernel, dim3((Din + kCols - 1) / kCols), dim3(kThreads), 0, st, d, W, g_y, g_x, accumulate_g_x);
  return (int)hipGetLastError();
}
