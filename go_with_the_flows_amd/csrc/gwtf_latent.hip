// gwtf_latent.hip -- the latent-space loss terms of the training step and their combination with the point NLL, two small launches forward and
// one backward (reference lib/networks/losses.py:24-33 GaussianFlowNLL, :36-41 GaussianEntropy, :159-170 Flow_Mixture_Loss.forward):
//     gnll = 0.5 ( sum_{b,j} [ lv0_j + sum_l flow_lv[l][b][j] + (z_bj - mu0_j)^2 / exp(lv0_j) ] / B + G log 2 pi )
//     gent = 0.5 ( G (1 + log 2 pi) + sum_{b,j} post_lv_bj / B )
//     pnll = sum_b nll_b / B            loss = pw pnll + gw gnll - ew gent
// As torch ops these are ~25 elementwise / reduction launches forward and ~30 backward on (B, G) tensors of a few thousand elements:
// 0.15 ms of a 9 ms step spent on launch latency.
#include <hip/hip_runtime.h>
#include "../../include/gwtf.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < kThreads / 64; ++w) s += red[w];      // every thread: the same order
  __syncthreads();
  return s;
}

// stage 1: one element per thread (the n2 flow_lv loads of a thread are independent and coalesced across the wave);
// part[blk] = {sum of the gnll integrand, sum of post_lv} of the block's 256 elements
__global__ __launch_bounds__(kThreads) void latent_part_kernel(const float* __restrict__ z, const float* __restrict__ mu0,
                                                               const float* __restrict__ lv0, const float* __restrict__ flow_lv,
                                                               const float* __restrict__ post_lv, float2* __restrict__ part, int B,
                                                               int G, int n2) {
  __shared__ float red[kThreads / 64];
  const int tid = threadIdx.x, n = B * G, e = blockIdx.x * kThreads + tid;
  float sg = 0.f, se = 0.f;
  if (e < n) {
    const int j = e % G;
    const float d = z[e] - mu0[j], l0 = lv0[j];
    float s0 = 0.f, s1 = 0.f;
    int l = 0;
    for (; l + 1 < n2; l += 2) {
      s0 += flow_lv[(size_t)l * n + e];
      s1 += flow_lv[(size_t)(l + 1) * n + e];
    }
    if (l < n2) s0 += flow_lv[(size_t)l * n + e];
    sg = l0 + d * d * __expf(-l0) + (s0 + s1);
    se = post_lv[e];
  }
  sg = block_sum(sg, red, tid);
  se = block_sum(se, red, tid);
  if (tid == 0) part[blockIdx.x] = make_float2(sg, se);
}

// stage 2 (one workgroup): the partials in block order + the per-shape point NLL -> out[4] = {loss, pnll, gnll, gent}
__global__ __launch_bounds__(kThreads) void latent_finish_kernel(const float* __restrict__ nll, const float2* __restrict__ part,
                                                                 float* __restrict__ out, int B, int G, int nb, float pw, float gw,
                                                                 float ew) {
  __shared__ float red[kThreads / 64];
  const int tid = threadIdx.x;
  float sg = 0.f, se = 0.f, sp = 0.f;
  for (int i = tid; i < nb; i += kThreads) {
    const float2 v = part[i];
    sg += v.x;
    se += v.y;
  }
  for (int b = tid; b < B; b += kThreads) sp += nll[b];
  sg = block_sum(sg, red, tid);
  se = block_sum(se, red, tid);
  sp = block_sum(sp, red, tid);
  if (tid == 0) {
    const float log2pi = 1.8378770664093453f;
    const float gnll = 0.5f * (sg / (float)B + log2pi * (float)G);
    const float gent = 0.5f * ((float)G * (1.0f + log2pi) + se / (float)B);
    const float pnll = sp / (float)B;
    out[0] = pw * pnll + gw * gnll - ew * gent;
    out[1] = pnll;
    out[2] = gnll;
    out[3] = gent;
  }
}

// g_out[4] = upstream of {loss, pnll, gnll, gent}.  Blocks [0, nb_e): elementwise part; the rest: the column sums of mu0 / lv0.
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ z,
                                                         const float* __restrict__ mu0, const float* __restrict__ lv0,
                                                         float* __restrict__ g_nll, float* __restrict__ g_z, float* __restrict__ g_mu0,
                                                         float* __restrict__ g_lv0, float* __restrict__ g_flow, float* __restrict__ g_post,
                                                         int B, int G, int n2, float pw, float gw, float ew, int nb_e) {
  const float gl = g_out[0];
  const float cp = (gl * pw + g_out[1]) / (float)B, cg = (gl * gw + g_out[2]) / (float)B, ce = (g_out[3] - gl * ew) / (float)B;
  const int n = B * G;
  if ((int)blockIdx.x < nb_e) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < B) g_nll[e] = cp;
    if (e >= n) return;
    const int j = e % G;
    g_z[e] = cg * (z[e] - mu0[j]) * __expf(-lv0[j]);
    g_post[e] = 0.5f * ce;
    const float c = 0.5f * cg;
    for (int l = 0; l < n2; ++l) g_flow[(size_t)l * n + e] = c;
    return;
  }
  const int j = ((int)blockIdx.x - nb_e) * 256 + threadIdx.x;
  if (j >= G) return;
  const float m = mu0[j], r = __expf(-lv0[j]);
  float sm = 0.f, sl = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = z[(size_t)b * G + j] - m;
    sm += d;
    sl = fmaf(d, d, sl);
  }
  g_mu0[j] = -cg * r * sm;
  g_lv0[j] = 0.5f * cg * ((float)B - r * sl);
}

}  // namespace

extern "C" int gwtf_latent_loss_workspace_floats(int B, int G) { return 2 * ((B * G + kThreads - 1) / kThreads); }

extern "C" int gwtf_latent_loss_forward(const float* nll, const float* z, const float* mu0, const float* lv0, const float* flow_lv,
                                        const float* post_lv, float* workspace, float* out4, int B, int G, int n2, float pw, float gw,
                                        float ew, void* stream) {
  if (!nll || !z || !mu0 || !lv0 || !flow_lv || !post_lv || !workspace || !out4 || B <= 0 || G <= 0 || n2 <= 0) return GWTF_E_BADARG;
  const int nb = (B * G + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(latent_part_kernel, dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, z, mu0, lv0, flow_lv, post_lv,
                     (float2*)workspace, B, G, n2);
  hipLaunchKernelGGL(latent_finish_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, nll, (const float2*)workspace, out4, B, G,
                     nb, pw, gw, ew);
  return (int)hipGetLastError();
}

extern "C" int gwtf_latent_loss_backward(const float* g_out4, const float* z, const float* mu0, const float* lv0, float* g_nll,
                                         float* g_z, float* g_mu0, float* g_lv0, float* g_flow_lv, float* g_post_lv, int B, int G,
                                         int n2, float pw, float gw, float ew, void* stream) {
  if (!g_out4 || !z || !mu0 || !lv0 || !g_nll || !g_z || !g_mu0 || !g_lv0 || !g_flow_lv || !g_post_lv || B <= 0 || G <= 0 || n2 <= 0)
    return GWTF_E_BADARG;
  const int nb_e = (B * G + 255) / 256, nb_c = (G + 255) / 256;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(nb_e + nb_c), dim3(256), 0, (hipStream_t)stream, g_out4, z, mu0, lv0, g_nll, g_z, g_mu0,
                     g_lv0, g_flow_lv, g_post_lv, B, G, n2, pw, gw, ew, nb_e);
  return (int)hipGetLastError();
}
