// gwtf_heads.hip -- the per-SHAPE MLP heads around the point-flow decoder: FeatureEncoder / WeightsEncoder (reference
// lib/networks/encoders.py:31-89): n_layers x [Linear(no bias) -> BatchNorm1d -> Swish] then Linear(bias) heads (mu, logvar; the
// mixture-weight head ends in a log-softmax, :87-91).  In the training step they are g_posterior (512 -> 512 -> 128 + 128),
// p_prior (G -> G -> 3 + 3) and mixture_weights_encoder (3 x (G -> G) -> K): B <= 128 rows, a few MFLOP each, but ~120 small
// library launches per step (Linear, batch_norm, sigmoid, mul, ... and their autograd) on the critical path between the
// encoder and the decoders.
//
// ONE LAYER per launch, forward and backward:
//   forward : y = x W^T (+ bias) -> [BatchNorm over the B rows: batch statistics, running statistics updated `bn_updates` times,
//             or running statistics] -> [Swish | log-softmax].  A workgroup owns 16 output COLUMNS and all rows: everything
//             BatchNorm needs (column statistics) is local to it -- no grid-wide barrier; Dout / 16 workgroups (the exact-fp32 MFMA makes wider blocks throughput-bound on one CU).
//   backward: pass 1 (per block of 16 output columns): dL/dy through the activation and the BatchNorm (column sums again local),
//             dW = dy^T x, dbias, dgamma, dbeta;   pass 2 (per block of 16 INPUT columns): dL/dx = dy W.
// GEMMs: exact fp32 MFMA with operands straight from L2 (gwtf_gemm.h: split-K over the waves for long contractions); column sums: gwtf_gemm.h.  In a data-parallel run the caller gathers the rows of all
// ranks first (dist.gather_rows): the kernels always see the whole batch -- SyncBatchNorm semantics without a collective inside.
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "gwtf_gemm.h"
#include "../../include/gwtf.h"

namespace {

using namespace gwtf_gemm;
constexpr int kNJ = 1, kCols = 16 * kNJ;   // output (pass 1) / input (pass 2) columns per workgroup: narrow, so that a 512-wide layer is 32 workgroups
constexpr int kSplitK = 128;         // contractions at least this long are cut over the eight waves (gemm_splitk)
constexpr int kKeep = 4;             // rows of its column a thread keeps in registers (batches up to kKeep * 32 = 128 rows)
constexpr int kGyPitch = kCols + 1;  // the dL/dy block in LDS for the weight gradient: [128 rows][17]
constexpr int kWorkFloats = 8 * 64 * kCols;      // split-K slabs (32 KB); the dL/dy block (8.5 KB) reuses them
static_assert(kWorkFloats >= kMaxM * kGyPitch, "the dL/dy block must fit the work area");

// C block = A . B^T: the long contractions split over the waves, the short ones one tile per wave
__device__ __forceinline__ void head_gemm(int M, int N, int K, const float* A, long sai, long sak, const float* Bm, long sbj, long sbk,
                                          float* C, long ldc, bool accumulate, float* work) {
  if (K >= kSplitK) {
    gemm_splitk<kNJ>(M, N, K, A, sai, sak, Bm, sbj, sbk, C, ldc, accumulate, work);      // walks the rows 64 at a time: any M
  } else {
    for (int m0 = 0; m0 < M; m0 += kMaxM)                                              // one pass per 128 rows (its accumulators' reach)
      gemm_direct(min(kMaxM, M - m0), N, K, A + (long)m0 * sai, sai, sak, Bm, sbj, sbk, C + (long)m0 * ldc, ldc, accumulate);
  }
}

struct HeadDims { int B, Din, Dout; };

// s = BatchNorm(y) for a column given its statistics
__device__ __forceinline__ float bn_apply(float y, float mean, float rstd, float ga, float be) { return fmaf((y - mean) * rstd, ga, be); }

// ---- forward ----------------------------------------------------------------------------------------------------------------
// ypre [B][Dout]: the Linear's output (the BatchNorm input; kept for the backward), stats [3][Dout] = mean, biased var, rstd actually
// used, out [B][Dout].  bn_mode 0: no BatchNorm; 1: batch statistics (+ running update, momentum form, unbiased variance, applied
// bn_updates times: the reference evaluates p_prior once per mixture component on the same batch, models.py:169-193 inside
// flow_mixture.py:163-166); 2: running statistics.  act 0: none, 1: swish, 2: log-softmax over the Dout (<= 16) columns.
__device__ __forceinline__ void head_fwd_block(int bid, HeadDims d, const float* __restrict__ x, const float* __restrict__ W,
                                               const float* __restrict__ bias, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float* __restrict__ rmean,
                                               float* __restrict__ rvar, long long* __restrict__ nbt, float momentum,
                                               float bn_eps, int bn_mode, int bn_updates, int act,
                                               float* __restrict__ ypre, float* __restrict__ stats,
                                               float* __restrict__ out) {
  __shared__ float red[kThreads];
  __shared__ float work[kWorkFloats];
  const int B = d.B, Din = d.Din, Dout = d.Dout;
  const int c0 = bid * kCols, nc = min(kCols, Dout - c0);
  head_gemm(B, nc, Din, x, Din, 1, W + (size_t)c0 * Din, Din, 1, ypre + c0, Dout, false, work);
  phase_sync();
  const ColMap cm = col_map(kCols);
  const int col = c0 + cm.col;
  const bool live = cm.on && cm.col < nc;
  auto yat = [&](int b, int c) { return c < nc ? ypre[(size_t)b * Dout + c0 + c] + (bias ? bias[c0 + c] : 0.f) : 0.f; };
  float mean = 0.f, var = 1.f, rstd = 1.f, ga = 1.f, be = 0.f;
  // up to kKeep rows per thread (B <= 128): the thread's y values stay in registers for the statistics and the output
  const bool keep = B <= kKeep * cm.RG;
  float yv[kKeep];
#pragma unroll
  for (int i = 0; i < kKeep; ++i) yv[i] = (keep && cm.on && cm.rg + i * cm.RG < B) ? yat(cm.rg + i * cm.RG, cm.col) : 0.f;
  if (bn_mode == 1 && keep) {
    mean = col_sum_kept(cm, kCols, B, yv, red) / (float)B;
    float dv[kKeep];
#pragma unroll
    for (int i = 0; i < kKeep; ++i) { const float t = yv[i] - mean; dv[i] = t * t; }
    var = col_sum_kept(cm, kCols, B, dv, red) / (float)B;
  } else if (bn_mode == 1) {
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
  if (bn_mode == 1 && nbt && bid == 0 && threadIdx.x == 0) *nbt += bn_updates;
  if (act == 2) {
    // log-softmax over the columns of a row (all in this workgroup: Dout <= kCols): thread t < B owns row t
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
  if (live && keep) {
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
      const int b = cm.rg + i * cm.RG;
      if (b < B) {
        float s = yv[i];
        if (bn_mode != 0) s = bn_apply(s, mean, rstd, ga, be);
        out[(size_t)b * Dout + col] = act == 1 ? swishf(s) : s;
      }
    }
  } else if (live) {
    for (int b = cm.rg; b < B; b += cm.RG) {
      float s = yat(b, cm.col);
      if (bn_mode != 0) s = bn_apply(s, mean, rstd, ga, be);
      out[(size_t)b * Dout + col] = act == 1 ? swishf(s) : s;
    }
  }
}

__global__ __launch_bounds__(kThreads) void head_fwd_kernel(HeadDims d, const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, long long* __restrict__ nbt, float momentum,
                                                            float bn_eps, int bn_mode, int bn_updates, int act,
                                                            float* __restrict__ ypre, float* __restrict__ stats,
                                                            float* __restrict__ out) {
  head_fwd_block(blockIdx.x, d, x, W, bias, gamma, beta, rmean, rvar, nbt, momentum, bn_eps, bn_mode, bn_updates, act, ypre, stats, out);
}

// TWO plain Linear heads on the same input (the mu / logvar heads of a FeatureEncoder, reference encoders.py:55-60) in one launch:
// blocks [0, ceil(Dout_a / 16)) are head a's column blocks, the rest head b's
struct HeadPair { const float* W[2]; const float* bias[2]; float* ypre[2]; float* out[2]; int Dout[2]; };
__global__ __launch_bounds__(kThreads) void head_fwd_pair_kernel(int B, int Din, const float* __restrict__ x, HeadPair hp) {
  const int na = (hp.Dout[0] + kCols - 1) / kCols;
  const int h = (int)blockIdx.x < na ? 0 : 1, bid = (int)blockIdx.x - (h ? na : 0);
  const HeadDims d = {B, Din, hp.Dout[h]};
  head_fwd_block(bid, d, x, hp.W[h], hp.bias[h], nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 1e-5f, 0, 1, 0, hp.ypre[h], nullptr,
                 hp.out[h]);
}

// ---- backward, pass 1: per block of 16 output columns ---------------------------------------------------------------------------
// g_out [B][Dout] = dL/d out;  writes g_y [B][Dout] = dL/d(x W^T) (scratch for pass 2), g_W [Dout][Din], g_bias / g_gamma / g_beta
// [Dout] (each may be null).
__device__ __forceinline__ void head_bwd1_block(int bid, HeadDims d, const float* __restrict__ x, const float* __restrict__ bias,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                const float* __restrict__ ypre, const float* __restrict__ stats,
                                                const float* __restrict__ out, const float* __restrict__ g_out,
                                                int bn_mode, int act, float* __restrict__ g_y, float* __restrict__ g_W,
                                                float* __restrict__ g_bias, float* __restrict__ g_gamma,
                                                float* __restrict__ g_beta) {
  __shared__ float red[kThreads];
  const int B = d.B, Din = d.Din, Dout = d.Dout;
  const int c0 = bid * kCols, nc = min(kCols, Dout - c0);
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
  __shared__ float work[kWorkFloats];
  lptr gyt = (lptr)work;                                      // dL/dy block [row][col], pitch 17, zero beyond the batch rows
  for (int t = threadIdx.x; t < kMaxM * kGyPitch; t += kThreads) gyt[t] = 0.f;
  __syncthreads();
  float sum_ds = 0.f, sum_dsx = 0.f;
  // (B <= 128) the thread's rows of ds and of the normalised input stay in registers: fetched and differentiated once
  const bool keep = B <= kKeep * cm.RG;
  float dsv[kKeep], xhv[kKeep];
#pragma unroll
  for (int i = 0; i < kKeep; ++i) {
    const int b = cm.rg + i * cm.RG;
    const bool have = keep && cm.on && b < B && cm.col < nc;
    dsv[i] = have ? ds_at(b, cm.col, mean, rstd, ga, be, bi) : 0.f;
    xhv[i] = (have && bn_mode != 0) ? (ypre[(size_t)b * Dout + c0 + cm.col] + bi - mean) * rstd : 0.f;
  }
  if ((bn_mode != 0 || g_bias) && keep) {
    sum_ds = col_sum_kept(cm, kCols, B, dsv, red);
    if (bn_mode != 0) {
      float pv[kKeep];
#pragma unroll
      for (int i = 0; i < kKeep; ++i) pv[i] = dsv[i] * xhv[i];
      sum_dsx = col_sum_kept(cm, kCols, B, pv, red);
    }
  } else if (bn_mode != 0 || g_bias) {
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
  // 128 rows at a time (the LDS block's reach): the weight gradient adds up over the row blocks of a longer batch
  const float invB = 1.0f / (float)B;
  for (int r0 = 0; r0 < B; r0 += kMaxM) {
    const int nb = min(kMaxM, B - r0);
    if (r0) __syncthreads();                                  // the previous block's product has read the LDS block
    if (live && keep) {           // (one row block: r0 == 0)
#pragma unroll
      for (int i = 0; i < kKeep; ++i) {
        const int b = cm.rg + i * cm.RG;
        if (b < B) {
          float gy = dsv[i];
          if (bn_mode == 1) gy = ga * rstd * (gy - invB * sum_ds - xhv[i] * invB * sum_dsx);
          else if (bn_mode == 2) gy = ga * rstd * gy;
          g_y[(size_t)b * Dout + col] = gy;
          gyt[b * kGyPitch + cm.col] = gy;
        }
      }
    } else if (live) {
      for (int b = r0 + cm.rg; b < r0 + nb; b += cm.RG) {
        const size_t o = (size_t)b * Dout + col;
        float gy = ds_at(b, cm.col, mean, rstd, ga, be, bi);
        if (bn_mode == 1) {
          const float xh = (ypre[o] + bi - mean) * rstd;
          gy = ga * rstd * (gy - invB * sum_ds - xh * invB * sum_dsx);
        } else if (bn_mode == 2) {
          gy = ga * rstd * gy;
        }
        g_y[o] = gy;
        gyt[(b - r0) * kGyPitch + cm.col] = gy;
      }
    }
    __syncthreads();
    // dW[c0 .. c0 + nc)[:] (+)= g_y[rows, block]^T x[rows] : M = nc, K = the block's rows, N = Din; the dL/dy operand from LDS (rows
    // beyond nb: stale but finite, they meet zeros), the waves cut Din
    if (g_W) gemm_nsplit_lds<kNJ>(nc, Din, nb, gyt, kGyPitch, (gptr_c)(x + (size_t)r0 * Din), Din, (gptr)(g_W + (size_t)c0 * Din), Din, r0 > 0);
  }
}

__global__ __launch_bounds__(kThreads) void head_bwd1_kernel(HeadDims d, const float* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ ypre, const float* __restrict__ stats,
                                                             const float* __restrict__ out, const float* __restrict__ g_out,
                                                             int bn_mode, int act, float* __restrict__ g_y, float* __restrict__ g_W,
                                                             float* __restrict__ g_bias, float* __restrict__ g_gamma,
                                                             float* __restrict__ g_beta) {
  head_bwd1_block(blockIdx.x, d, x, bias, gamma, beta, ypre, stats, out, g_out, bn_mode, act, g_y, g_W, g_bias, g_gamma, g_beta);
}

struct HeadPairBwd { const float* bias[2]; const float* ypre[2]; const float* out[2]; const float* g_out[2]; float* g_y[2]; float* g_W[2];
                     float* g_bias[2]; const float* W[2]; int Dout[2]; };
__global__ __launch_bounds__(kThreads) void head_bwd1_pair_kernel(int B, int Din, const float* __restrict__ x, HeadPairBwd hp) {
  const int na = (hp.Dout[0] + kCols - 1) / kCols;
  const int h = (int)blockIdx.x < na ? 0 : 1, bid = (int)blockIdx.x - (h ? na : 0);
  const HeadDims d = {B, Din, hp.Dout[h]};
  head_bwd1_block(bid, d, x, hp.bias[h], nullptr, nullptr, hp.ypre[h], nullptr, hp.out[h], hp.g_out[h], 0, 0, hp.g_y[h], hp.g_W[h],
                  hp.g_bias[h], nullptr, nullptr);
}

// ---- backward, pass 2: per block of 16 input columns: g_x[:, block] (+)= g_y W[:, block] ------------------------------------------
__global__ __launch_bounds__(kThreads) void head_bwd2_kernel(HeadDims d, const float* __restrict__ W, const float* __restrict__ g_y,
                                                             float* __restrict__ g_x, int accumulate) {
  const int c0 = blockIdx.x * kCols, nc = min(kCols, d.Din - c0);
  __shared__ float work[kWorkFloats];
  head_gemm(d.B, nc, d.Dout, g_y, d.Dout, 1, W + c0, 1, d.Din, g_x + c0, d.Din, accumulate != 0, work);
}

// the pair's input gradient: g_x[:, block] = g_ya Wa[:, block] + g_yb Wb[:, block] (the second product accumulates onto the first)
__global__ __launch_bounds__(kThreads) void head_bwd2_pair_kernel(int B, int Din, HeadPairBwd hp, float* __restrict__ g_x) {
  const int c0 = blockIdx.x * kCols, nc = min(kCols, Din - c0);
  __shared__ float work[kWorkFloats];
  head_gemm(B, nc, hp.Dout[0], hp.g_y[0], hp.Dout[0], 1, hp.W[0] + c0, 1, Din, g_x + c0, Din, false, work);
  phase_sync();
  head_gemm(B, nc, hp.Dout[1], hp.g_y[1], hp.Dout[1], 1, hp.W[1] + c0, 1, Din, g_x + c0, Din, true, work);
}

constexpr int kMaxRows = 1 << 16;   // any batch: the kernels walk it in blocks of 64 / 128 rows
bool dims_ok(int B, int Din, int Dout) { return B >= 1 && B <= kMaxRows && Din >= 1 && Dout >= 1 && Din <= 4096 && Dout <= 4096; }

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

// see include/gwtf.h
extern "C" int gwtf_head_pair_forward(const float* x, const float* Wa, const float* bias_a, const float* Wb, const float* bias_b,
                                      float* ypre_a, float* out_a, float* ypre_b, float* out_b, int B, int Din, int Dout_a, int Dout_b,
                                      void* stream) {
  if (!x || !Wa || !Wb || !ypre_a || !out_a || !ypre_b || !out_b || !dims_ok(B, Din, Dout_a) || !dims_ok(B, Din, Dout_b))
    return GWTF_E_BADARG;
  HeadPair hp = {{Wa, Wb}, {bias_a, bias_b}, {ypre_a, ypre_b}, {out_a, out_b}, {Dout_a, Dout_b}};
  const int blocks = (Dout_a + kCols - 1) / kCols + (Dout_b + kCols - 1) / kCols;
  hipLaunchKernelGGL(head_fwd_pair_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, B, Din, x, hp);
  return (int)hipGetLastError();
}

extern "C" int gwtf_head_pair_backward(const float* x, const float* Wa, const float* bias_a, const float* Wb, const float* bias_b,
                                       const float* ypre_a, const float* out_a, const float* ypre_b, const float* out_b,
                                       const float* g_out_a, const float* g_out_b, float* g_y_a, float* g_y_b, float* g_x, float* g_Wa,
                                       float* g_bias_a, float* g_Wb, float* g_bias_b, int B, int Din, int Dout_a, int Dout_b,
                                       void* stream) {
  if (!x || !Wa || !Wb || !ypre_a || !out_a || !ypre_b || !out_b || !g_out_a || !g_out_b || !g_y_a || !g_y_b ||
      !dims_ok(B, Din, Dout_a) || !dims_ok(B, Din, Dout_b))
    return GWTF_E_BADARG;
  HeadPairBwd hp = {{bias_a, bias_b}, {ypre_a, ypre_b}, {out_a, out_b}, {g_out_a, g_out_b}, {g_y_a, g_y_b}, {g_Wa, g_Wb},
                    {g_bias_a, g_bias_b}, {Wa, Wb}, {Dout_a, Dout_b}};
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (Dout_a + kCols - 1) / kCols + (Dout_b + kCols - 1) / kCols;
  hipLaunchKernelGGL(head_bwd1_pair_kernel, dim3(blocks), dim3(kThreads), 0, st, B, Din, x, hp);
  if (g_x) hipLaunchKernelGGL(head_bwd2_pair_kernel, dim3((Din + kCols - 1) / kCols), dim3(kThreads), 0, st, B, Din, hp, g_x);
  return (int)hipGetLastError();
}
