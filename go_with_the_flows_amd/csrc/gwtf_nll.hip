// gwtf_nll.hip -- Gaussian base log-density + mixture log-sum-exp + per-shape reduction.
//
// Reference: FlowMixtureNLL.forward, lib/networks/losses.py:88-137 (per-component body :112-122 =
// PointFlowNLL :11-20).  For shape i, point n, component k:
//     lp_k = -0.5 * sum_d [ lv0 + logdet + (z - mu0)^2 / exp(lv0) ] - 0.5 * 3 * log(2 pi) + log w_ik
//     nll_shape[i] = - sum_n logsumexp_k lp_k
// HBM-bound streaming kernel: 24*K bytes per point read once, coalesced along N; wavefront shuffle
// reduction, one LDS stage, one float atomic per workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/gwtf.h"

namespace {

constexpr int kMaxK = 64;
constexpr int kThreads = 256;
constexpr int kPtsPerBlock = 512;    // 2 points per thread: (N/512) x B workgroups (8 k for 64 x 2048) keep all 256 CUs streaming

__global__ __launch_bounds__(1024) void nll_kernel(const float* __restrict__ z, const float* __restrict__ logdet,
                                                       const float* __restrict__ mu0, const float* __restrict__ lv0,
                                                       const float* __restrict__ logits, float* __restrict__ point_lse,
                                                       float* __restrict__ nll_shape, int K, int B, int N) {
  __shared__ float s_logw[kMaxK];
  __shared__ float s_mu[kMaxK][3], s_lv[kMaxK][3], s_iv[kMaxK][3];
  __shared__ float s_part[16];
  const int b = blockIdx.y;
  const int ppb = gridDim.x == 1 ? N : kPtsPerBlock;      // one workgroup per shape: the whole cloud, plain store at the end
  if (threadIdx.x < 64) {
    // log w = log(exp(logit)) - logsumexp(logits)   (losses.py:101-104)
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logits[(size_t)b * K + k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(logits[(size_t)b * K + k] - m);
    const float lse = m + logf(s);
    for (int k = threadIdx.x; k < K; k += 64) s_logw[k] = logf(expf(logits[(size_t)b * K + k])) - lse;
  }
  for (int t = threadIdx.x; t < K * 3; t += blockDim.x) {
    const int k = t / 3, d = t % 3;
    const float l = lv0[((size_t)k * B + b) * 3 + d];
    s_mu[k][d] = mu0[((size_t)k * B + b) * 3 + d];
    s_lv[k][d] = l;
    s_iv[k][d] = 1.0f / expf(l);      // the points multiply by 1/exp(lv0): one division per (component, dim), not per point
  }
  __syncthreads();
  const float half_log2pi3 = 0.5f * 3.0f * 1.8378770664093453f;
  float local = 0.f;
  const int n_end = min(N, (int)(blockIdx.x + 1) * ppb);
  for (int n = blockIdx.x * ppb + threadIdx.x; n < n_end; n += blockDim.x) {
    float m = -INFINITY, s = 0.f;  // online log-sum-exp over components
    for (int k = 0; k < K; ++k) {
      float qsum = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t o = (((size_t)k * B + b) * 3 + d) * N + n;
        const float diff = z[o] - s_mu[k][d];
        qsum += (s_lv[k][d] + logdet[o]) + diff * diff * s_iv[k][d];
      }
      const float lp = -0.5f * qsum - half_log2pi3 + s_logw[k];
      if (lp > m || k == 0) {
        s = s * expf(m - lp) + 1.0f;
        m = lp;
      } else {
        s += expf(lp - m);
      }
    }
    const float lse = m + logf(s);
    if (point_lse) point_lse[(size_t)b * N + n] = lse;
    local += lse;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)blockDim.x / 64; ++w) t += s_part[w];
    if (gridDim.x == 1) nll_shape[b] = -t;
    else atomicAdd(&nll_shape[b], -t);
  }
}

// Backward of nll_kernel.  With r_k(n) = softmax_k(log w_k + log p_k(n)) (recomputed from the saved per-point lse):
//   d nll_b / d z     = +r_k (z - mu0)/exp(lv0)        d nll_b / d logdet = +0.5 r_k
//   d nll_b / d mu0   = -sum_n r_k (z - mu0)/exp(lv0)  d nll_b / d lv0    = sum_n r_k (0.5 - 0.5 (z - mu0)^2/exp(lv0))
//   d nll_b / d logit_j = -sum_n r_j(n) + N softmax_j(logits)
// all scaled by the upstream g_nll[b].  Streaming, HBM bound (reads 24K + 4, writes 24K bytes per point).
__global__ __launch_bounds__(kThreads) void nll_bwd_kernel(const float* __restrict__ z, const float* __restrict__ logdet,
                                                           const float* __restrict__ mu0, const float* __restrict__ lv0,
                                                           const float* __restrict__ logits,
                                                           const float* __restrict__ point_lse,
                                                           const float* __restrict__ g_nll, float* __restrict__ g_z,
                                                           float* __restrict__ g_ld, float* __restrict__ g_mu0,
                                                           float* __restrict__ g_lv0, float* __restrict__ g_logits, int K,
                                                           int B, int N) {
  __shared__ float s_logw[kMaxK], s_sm[kMaxK];
  __shared__ float s_mu[kMaxK][3], s_lv[kMaxK][3], s_iv[kMaxK][3];
  __shared__ float s_acc[kMaxK][7];   // per component: d mu0 (3), d lv0 (3), sum r
  const int b = blockIdx.y;
  if (threadIdx.x < 64) {
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logits[(size_t)b * K + k]);
    float sden = 0.f;
    for (int k = 0; k < K; ++k) sden += expf(logits[(size_t)b * K + k] - m);
    const float lse = m + logf(sden);
    for (int k = threadIdx.x; k < K; k += 64) {
      s_logw[k] = logf(expf(logits[(size_t)b * K + k])) - lse;
      s_sm[k] = expf(logits[(size_t)b * K + k] - lse);
    }
  }
  for (int t = threadIdx.x; t < K * 3; t += blockDim.x) {
    const int k = t / 3, d = t % 3;
    const float l = lv0[((size_t)k * B + b) * 3 + d];
    s_mu[k][d] = mu0[((size_t)k * B + b) * 3 + d];
    s_lv[k][d] = l;
    s_iv[k][d] = 1.0f / expf(l);      // the points multiply by 1/exp(lv0): one division per (component, dim), not per point
  }
  for (int t = threadIdx.x; t < K * 7; t += blockDim.x) (&s_acc[0][0])[t] = 0.f;
  __syncthreads();
  const float half_log2pi3 = 0.5f * 3.0f * 1.8378770664093453f;
  const float gb = g_nll[b];
  const int n_end = min(N, (int)(blockIdx.x + 1) * kPtsPerBlock);
  for (int k = 0; k < K; ++k) {
    float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = blockIdx.x * kPtsPerBlock + threadIdx.x; n < n_end; n += blockDim.x) {
      float diff[3], qsum = 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t o = (((size_t)k * B + b) * 3 + d) * N + n;
        diff[d] = z[o] - s_mu[k][d];
        qsum += (s_lv[k][d] + logdet[o]) + diff[d] * diff[d] * s_iv[k][d];
      }
      const float r = expf(-0.5f * qsum - half_log2pi3 + s_logw[k] - point_lse[(size_t)b * N + n]) * gb;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t o = (((size_t)k * B + b) * 3 + d) * N + n;
        const float w = diff[d] * s_iv[k][d];
        g_z[o] = r * w;
        g_ld[o] = 0.5f * r;
        a[d] -= r * w;
        a[3 + d] += r * (0.5f - 0.5f * diff[d] * w);
      }
      a[6] += r;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a[i] += __shfl_down(a[i], off);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < 7; ++i) atomicAdd(&s_acc[k][i], a[i]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < K * 7; t += blockDim.x) {
    const int k = t / 7, i = t % 7;
    const float v = s_acc[k][i];
    if (i < 3) atomicAdd(&g_mu0[((size_t)k * B + b) * 3 + i], v);
    else if (i < 6) atomicAdd(&g_lv0[((size_t)k * B + b) * 3 + (i - 3)], v);
    else {
      // -sum_n r_k(n) plus this block's share of N * softmax_k * g
      const int pts = n_end - blockIdx.x * kPtsPerBlock;
      atomicAdd(&g_logits[(size_t)b * K + k], -v + gb * (float)pts * s_sm[k]);
    }
  }
}

}  // namespace

extern "C" int gwtf_mixture_nll_backward(const float* z, const float* logdet, const float* mu0, const float* lv0,
                                         const float* logits, const float* point_lse, const float* g_nll, float* g_z,
                                         float* g_logdet, float* g_mu0, float* g_lv0, float* g_logits, int K, int B, int N,
                                         void* stream) {
  if (K <= 0 || K > kMaxK || B <= 0 || N <= 0 || !z || !logdet || !mu0 || !lv0 || !logits || !point_lse || !g_nll || !g_z ||
      !g_logdet || !g_mu0 || !g_lv0 || !g_logits)
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(g_mu0, 0, sizeof(float) * (size_t)K * B * 3, st);
  if (e == hipSuccess) e = hipMemsetAsync(g_lv0, 0, sizeof(float) * (size_t)K * B * 3, st);
  if (e == hipSuccess) e = hipMemsetAsync(g_logits, 0, sizeof(float) * (size_t)B * K, st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((N + kPtsPerBlock - 1) / kPtsPerBlock, B);
  hipLaunchKernelGGL(nll_bwd_kernel, grid, dim3(kThreads), 0, st, z, logdet, mu0, lv0, logits, point_lse, g_nll, g_z,
                     g_logdet, g_mu0, g_lv0, g_logits, K, B, N);
  return (int)hipGetLastError();
}

extern "C" int gwtf_mixture_nll(const float* z, const float* logdet, const float* mu0, const float* lv0,
                                const float* logits, float* point_lse, float* nll_shape, int K, int B, int N,
                                void* stream) {
  if (K <= 0 || K > kMaxK || B <= 0 || N <= 0 || !z || !logdet || !mu0 || !lv0 || !logits || !nll_shape)
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 4096 && B >= 16) {     // one 1024-thread workgroup per shape: no atomics, no memset node in front of the kernel
    hipLaunchKernelGGL(nll_kernel, dim3(1, B), dim3(1024), 0, st, z, logdet, mu0, lv0, logits, point_lse, nll_shape, K, B, N);
    return (int)hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(nll_shape, 0, sizeof(float) * (size_t)B, st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((N + kPtsPerBlock - 1) / kPtsPerBlock, B);
  hipLaunchKernelGGL(nll_kernel, grid, dim3(kThreads), 0, st, z, logdet, mu0, lv0, logits, point_lse, nll_shape, K, B, N);
  return (int)hipGetLastError();
}
