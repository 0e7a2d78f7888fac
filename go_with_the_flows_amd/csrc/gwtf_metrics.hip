// gwtf_metrics.hip -- structural losses between point sets: nearest-neighbour (Chamfer) distances and the approximate
// earth-mover matching.  These are the reference's ONLY native code (CUDA extension
// lib/metrics/pytorch_structural_losses/src/{nndistance.cu,approxmatch.cu}, bound in structural_loss.cpp:22-123 and
// pybind/bind.cpp:9-15); evaluate_ae.py cannot run on ROCm without them.  Same results (first-minimum tie rule, the
// nine-level auction schedule, clamp constants), different decomposition: the reference caps every launch at 32
// workgroups (one V100-era choice); here the grid covers (batch x tiles) and both directions run in one launch.
// Layout of the point sets is the reference's: [b][n][3] point-major.
#include <hip/hip_runtime.h>
#include "../../include/gwtf.h"

namespace {

constexpr int kTile = 1024;   // points of the "other" set staged in LDS per step (16 KiB as float4)

// dist[i][j] = min_k |a_j - b_k|^2, idx = first minimiser (reference nndistance.cu:2-124: strict '<' inside a tile,
// strict '>' across tiles).  grid = (query tiles, batch, 2 directions); 256 threads x 2 query points each.
__global__ __launch_bounds__(256) void nnd_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                  float* __restrict__ dist1, int* __restrict__ idx1,
                                                  float* __restrict__ dist2, int* __restrict__ idx2, int b, int n, int m) {
  __shared__ float4 buf[kTile];
  const bool rev = blockIdx.z == 1;
  const float* A = rev ? xyz2 : xyz1;
  const float* Bp = rev ? xyz1 : xyz2;
  const int na = rev ? m : n, nb = rev ? n : m;
  float* dist = rev ? dist2 : dist1;
  int* idx = rev ? idx2 : idx1;
  const int i = blockIdx.y;
  const int j0 = blockIdx.x * 512 + threadIdx.x, j1 = j0 + 256;
  if (blockIdx.x * 512 >= na) return;
  const bool v0 = j0 < na, v1 = j1 < na;
  const float ax0 = v0 ? A[((size_t)i * na + j0) * 3 + 0] : 0.f, ay0 = v0 ? A[((size_t)i * na + j0) * 3 + 1] : 0.f,
              az0 = v0 ? A[((size_t)i * na + j0) * 3 + 2] : 0.f;
  const float ax1 = v1 ? A[((size_t)i * na + j1) * 3 + 0] : 0.f, ay1 = v1 ? A[((size_t)i * na + j1) * 3 + 1] : 0.f,
              az1 = v1 ? A[((size_t)i * na + j1) * 3 + 2] : 0.f;
  float best0 = 3.4e38f, best1 = 3.4e38f;
  int bi0 = 0, bi1 = 0;
  for (int k0 = 0; k0 < nb; k0 += kTile) {
    const int cnt = min(kTile, nb - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
      const float* q = Bp + ((size_t)i * nb + k0 + t) * 3;
      buf[t] = make_float4(q[0], q[1], q[2], 0.f);
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < cnt; ++k) {
      const float4 q = buf[k];   // same address in every lane: LDS broadcast
      const float dx0 = q.x - ax0, dy0 = q.y - ay0, dz0 = q.z - az0;
      const float dx1 = q.x - ax1, dy1 = q.y - ay1, dz1 = q.z - az1;
      const float d0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;
      const float d1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
      if (d0 < best0) { best0 = d0; bi0 = k0 + k; }
      if (d1 < best1) { best1 = d1; bi1 = k0 + k; }
    }
  }
  if (v0) { dist[(size_t)i * na + j0] = best0; idx[(size_t)i * na + j0] = bi0; }
  if (v1) { dist[(size_t)i * na + j1] = best1; idx[(size_t)i * na + j1] = bi1; }
}

// reference nndistance.cu:129-148: grad_a[j] += 2 g_j (a_j - b_idx[j]); grad_b[idx[j]] -= the same.  z = direction.
__global__ __launch_bounds__(256) void nnd_grad_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                       const float* __restrict__ gd1, const int* __restrict__ idx1,
                                                       const float* __restrict__ gd2, const int* __restrict__ idx2,
                                                       float* __restrict__ g1, float* __restrict__ g2, int b, int n, int m) {
  const bool rev = blockIdx.z == 1;
  const float* A = rev ? xyz2 : xyz1;
  const float* Bp = rev ? xyz1 : xyz2;
  const int na = rev ? m : n, nb = rev ? n : m;
  const float* gd = rev ? gd2 : gd1;
  const int* idx = rev ? idx2 : idx1;
  float* ga = rev ? g2 : g1;
  float* gb = rev ? g1 : g2;
  const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= na) return;
  const int k = idx[(size_t)i * na + j];
  const float g = gd[(size_t)i * na + j] * 2.0f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float diff = g * (A[((size_t)i * na + j) * 3 + d] - Bp[((size_t)i * nb + k) * 3 + d]);
    atomicAdd(&ga[((size_t)i * na + j) * 3 + d], diff);
    atomicAdd(&gb[((size_t)i * nb + k) * 3 + d], -diff);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Approximate EMD matching (reference approxmatch.cu:3-182): nine temperature levels -4^j, j = 7..-1; per level
//   (L) ratioL[k] = remainL[k] / (1e-9 + sum_l e_kl remainR[l])
//   (R) sumr = remainR[l] sum_k e_kl ratioL[k];  ratioR[l] = min(remainR[l]/(sumr+1e-9), 1) remainR[l];
//       remainR[l] = max(0, remainR[l] - sumr)
//   (M) w = e_kl ratioL[k] ratioR[l];  match[l][k] += w;  remainL[k] = max(0, remainL[k] - sum_l w)
// with e_kl = exp(level |a_k - b_l|^2).  The reference runs the whole schedule inside one workgroup per cloud pair (at
// most 32 workgroups on the device); here every sweep is its own launch over (tiles of 256 points) x (cloud pairs), so a
// batch of 64 pairs of 2048 points puts 512 workgroups on the 256 CUs, and the stream order is the barrier between
// sweeps.  exp(level d) is evaluated as exp2((level log2 e) d) with the bare v_exp_f32 (results below 2^-126 flush to 0).
// temp: [b][2(n+m)] = remainL[n] | remainR[m] | ratioL[n] | ratioR[m]
constexpr int kEmdThreads = 256;
constexpr int kEmdTile = 1024;

struct EmdPair {
  const float* A; const float* Bp; float* M; float* remainL; float* remainR; float* ratioL; float* ratioR;
};
__device__ inline EmdPair emd_pair(const float* xyz1, const float* xyz2, float* match, float* temp, int i, int n, int m) {
  EmdPair p;
  p.A = xyz1 + (size_t)i * n * 3;
  p.Bp = xyz2 + (size_t)i * m * 3;
  p.M = match + (size_t)i * n * m;
  p.remainL = temp + (size_t)i * (n + m) * 2;
  p.remainR = p.remainL + n;
  p.ratioL = p.remainR + m;
  p.ratioR = p.ratioL + n;
  return p;
}

__global__ __launch_bounds__(256) void emd_init_kernel(float* __restrict__ temp, int n, int m) {
  float* remainL = temp + (size_t)blockIdx.y * (n + m) * 2;
  const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;   // integer ratios (:6-12)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) remainL[t] = multiL;
  else if (t < n + m) remainL[t] = multiR;
}

// sum over the OTHER set of exp2(lvl2 |p - q|^2) * weight(q), q staged through LDS as (x,y,z,weight); p per thread.
//   MODE 0: plain sum (sweeps L and R)
//   MODE 1: sweep M, materialising match[q][p] += w (w = e * wp * weight(q); `pitch` = row length of match); lanes
//           whose w is exactly 0 skip the read-modify-write -- at the sharp early levels that is almost all of them
//   MODE 2: sweep M without the matrix: also returns sum_q w * |p - q| in `cost` (ApproxMatch + MatchCost fused)
template <int MODE>
__device__ inline float emd_sweep(const float* __restrict__ other, const float* __restrict__ weight, int cnt_other,
                                  float px, float py, float pz, float lvl2, float4* buf, float wp, float* __restrict__ Mcol,
                                  int pitch, bool valid, float& cost) {
  float acc = 0.f, c = 0.f;
  for (int l0 = 0; l0 < cnt_other; l0 += kEmdTile) {
    const int cnt = min(kEmdTile, cnt_other - l0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += kEmdThreads)
      buf[t] = make_float4(other[(l0 + t) * 3 + 0], other[(l0 + t) * 3 + 1], other[(l0 + t) * 3 + 2], weight[l0 + t]);
    __syncthreads();
    if (MODE == 1) {
      if (valid) {
#pragma unroll 4
        for (int l = 0; l < cnt; ++l) {
          const float4 q = buf[l];
          const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
          const float w = __builtin_amdgcn_exp2f(lvl2 * (dx * dx + dy * dy + dz * dz)) * wp * q.w;
          if (w != 0.f) Mcol[(size_t)(l0 + l) * pitch] += w;
          acc += w;
        }
      }
    } else if (MODE == 2) {
#pragma unroll 4
      for (int l = 0; l < cnt; ++l) {
        const float4 q = buf[l];
        const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const float w = __builtin_amdgcn_exp2f(lvl2 * d2) * wp * q.w;
        c += w * __builtin_amdgcn_sqrtf(d2);
        acc += w;
      }
    } else {
#pragma unroll 4
      for (int l = 0; l < cnt; ++l) {
        const float4 q = buf[l];
        const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        acc += __builtin_amdgcn_exp2f(lvl2 * (dx * dx + dy * dy + dz * dz)) * q.w;
      }
    }
  }
  cost = c;
  return acc;
}

__global__ __launch_bounds__(kEmdThreads) void emd_left_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                               float* __restrict__ temp, int n, int m, float lvl2) {
  __shared__ float4 buf[kEmdTile];
  const EmdPair P = emd_pair(xyz1, xyz2, nullptr, temp, blockIdx.y, n, m);
  const int k = blockIdx.x * kEmdThreads + threadIdx.x;
  const bool v = k < n;
  const float x1 = v ? P.A[k * 3] : 0.f, y1 = v ? P.A[k * 3 + 1] : 0.f, z1 = v ? P.A[k * 3 + 2] : 0.f;
  float unused;
  const float s = emd_sweep<0>(P.Bp, P.remainR, m, x1, y1, z1, lvl2, buf, 0.f, nullptr, 0, v, unused);
  if (v) P.ratioL[k] = P.remainL[k] / (1e-9f + s);
}

__global__ __launch_bounds__(kEmdThreads) void emd_right_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                                float* __restrict__ temp, int n, int m, float lvl2) {
  __shared__ float4 buf[kEmdTile];
  const EmdPair P = emd_pair(xyz1, xyz2, nullptr, temp, blockIdx.y, n, m);
  const int l = blockIdx.x * kEmdThreads + threadIdx.x;
  const bool v = l < m;
  const float x2 = v ? P.Bp[l * 3] : 0.f, y2 = v ? P.Bp[l * 3 + 1] : 0.f, z2 = v ? P.Bp[l * 3 + 2] : 0.f;
  float unused;
  float sumr = emd_sweep<0>(P.A, P.ratioL, n, x2, y2, z2, lvl2, buf, 0.f, nullptr, 0, v, unused);
  if (v) {
    const float rr = P.remainR[l];
    sumr *= rr;
    P.ratioR[l] = fminf(rr / (sumr + 1e-9f), 1.0f) * rr;
    P.remainR[l] = fmaxf(0.0f, rr - sumr);
  }
}

template <bool FUSED>
__global__ __launch_bounds__(kEmdThreads) void emd_match_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                                float* __restrict__ match, float* __restrict__ temp,
                                                                float* __restrict__ out, int n, int m, float lvl2) {
  __shared__ float4 buf[kEmdTile];
  __shared__ float part[kEmdThreads / 64];
  const EmdPair P = emd_pair(xyz1, xyz2, match, temp, blockIdx.y, n, m);
  const int k = blockIdx.x * kEmdThreads + threadIdx.x;
  const bool v = k < n;
  const float x1 = v ? P.A[k * 3] : 0.f, y1 = v ? P.A[k * 3 + 1] : 0.f, z1 = v ? P.A[k * 3 + 2] : 0.f;
  const float rl = v ? P.ratioL[k] : 0.f;      // 0 for the padding lanes: they add nothing to the fused cost
  float c;
  const float s = emd_sweep<FUSED ? 2 : 1>(P.Bp, P.ratioR, m, x1, y1, z1, lvl2, buf, rl, FUSED ? nullptr : P.M + k, n, v, c);
  if (v) P.remainL[k] = fmaxf(0.0f, P.remainL[k] - s);
  if (FUSED) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < kEmdThreads / 64; ++w) t += part[w];
      atomicAdd(&out[blockIdx.y], t);
    }
  }
}

// out[i] = sum_{l,k} match[l][k] |a_k - b_l|   (reference approxmatch.cu:184-224).  grid = (column tiles, batch).
__global__ __launch_bounds__(256) void matchcost_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        const float* __restrict__ match, float* __restrict__ out, int b,
                                                        int n, int m) {
  __shared__ float part[4];
  const int i = blockIdx.y;
  const float* A = xyz1 + (size_t)i * n * 3;
  const float* Bp = xyz2 + (size_t)i * m * 3;
  const float* M = match + (size_t)i * n * m;
  float sub = 0.f;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const float x1 = A[k * 3], y1 = A[k * 3 + 1], z1 = A[k * 3 + 2];
    for (int l = 0; l < m; ++l) {
      const float dx = Bp[l * 3] - x1, dy = Bp[l * 3 + 1] - y1, dz = Bp[l * 3 + 2] - z1;   // wave-uniform loads
      sub += M[(size_t)l * n + k] * sqrtf(dx * dx + dy * dy + dz * dz);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sub += __shfl_down(sub, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sub;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[i], part[0] + part[1] + part[2] + part[3]);
}

// grad1[k] = sum_l match[l][k] (a_k - b_l) / max(|a_k - b_l|, 1e-10)      (reference :270-291)
__global__ __launch_bounds__(256) void matchcost_grad1_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                              const float* __restrict__ match, float* __restrict__ grad1,
                                                              int b, int n, int m) {
  const int i = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float* A = xyz1 + (size_t)i * n * 3;
  const float* Bp = xyz2 + (size_t)i * m * 3;
  const float* M = match + (size_t)i * n * m;
  const float x1 = A[k * 3], y1 = A[k * 3 + 1], z1 = A[k * 3 + 2];
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int l = 0; l < m; ++l) {
    const float dx = x1 - Bp[l * 3], dy = y1 - Bp[l * 3 + 1], dz = z1 - Bp[l * 3 + 2];
    const float d = M[(size_t)l * n + k] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
    gx += dx * d; gy += dy * d; gz += dz * d;
  }
  grad1[((size_t)i * n + k) * 3 + 0] = gx;
  grad1[((size_t)i * n + k) * 3 + 1] = gy;
  grad1[((size_t)i * n + k) * 3 + 2] = gz;
}

// grad2[l] = sum_k match[l][k] (b_l - a_k) / max(|b_l - a_k|, 1e-10)      (reference :229-269); one wave per l
__global__ __launch_bounds__(256) void matchcost_grad2_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                              const float* __restrict__ match, float* __restrict__ grad2,
                                                              int b, int n, int m) {
  const int i = blockIdx.y, l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (l >= m) return;
  const float* A = xyz1 + (size_t)i * n * 3;
  const float* Bp = xyz2 + (size_t)i * m * 3;
  const float* M = match + ((size_t)i * m + l) * n;
  const float x2 = Bp[l * 3], y2 = Bp[l * 3 + 1], z2 = Bp[l * 3 + 2];
  float gx = 0.f, gy = 0.f, gz = 0.f;
  for (int k = lane; k < n; k += 64) {
    const float dx = x2 - A[k * 3], dy = y2 - A[k * 3 + 1], dz = z2 - A[k * 3 + 2];
    const float d = M[k] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
    gx += dx * d; gy += dy * d; gz += dz * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gx += __shfl_down(gx, off); gy += __shfl_down(gy, off); gz += __shfl_down(gz, off);
  }
  if (lane == 0) {
    grad2[((size_t)i * m + l) * 3 + 0] = gx;
    grad2[((size_t)i * m + l) * 3 + 1] = gy;
    grad2[((size_t)i * m + l) * 3 + 2] = gz;
  }
}

}  // namespace

extern "C" int gwtf_nn_distance(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2,
                                int b, int n, int m, void* stream) {
  if (!xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2 || b <= 0 || n <= 0 || m <= 0) return GWTF_E_BADARG;
  const int tiles = (max(n, m) + 511) / 512;
  hipLaunchKernelGGL(nnd_kernel, dim3(tiles, b, 2), dim3(256), 0, (hipStream_t)stream, xyz1, xyz2, dist1, idx1, dist2, idx2, b,
                     n, m);
  return (int)hipGetLastError();
}

extern "C" int gwtf_nn_distance_grad(const float* xyz1, const float* xyz2, const float* grad_dist1, const int* idx1,
                                     const float* grad_dist2, const int* idx2, float* grad_xyz1, float* grad_xyz2, int b,
                                     int n, int m, void* stream) {
  if (!xyz1 || !xyz2 || !grad_dist1 || !idx1 || !grad_dist2 || !idx2 || !grad_xyz1 || !grad_xyz2 || b <= 0 || n <= 0 || m <= 0)
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3, st);
  if (e == hipSuccess) e = hipMemsetAsync(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nnd_grad_kernel, dim3((max(n, m) + 255) / 256, b, 2), dim3(256), 0, st, xyz1, xyz2, grad_dist1, idx1,
                     grad_dist2, idx2, grad_xyz1, grad_xyz2, b, n, m);
  return (int)hipGetLastError();
}

static int emd_schedule(const float* xyz1, const float* xyz2, float* match, float* temp, float* out, int b, int n, int m,
                        hipStream_t st) {
  hipError_t e = match ? hipMemsetAsync(match, 0, sizeof(float) * (size_t)b * n * m, st)
                       : hipMemsetAsync(out, 0, sizeof(float) * (size_t)b, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(emd_init_kernel, dim3((n + m + 255) / 256, b), dim3(256), 0, st, temp, n, m);
  const dim3 gl((n + kEmdThreads - 1) / kEmdThreads, b), gr((m + kEmdThreads - 1) / kEmdThreads, b);
  for (int j = 7; j > -2; --j) {   // approxmatch.cu:31-32
    const float lvl2 = -powf(4.0f, (float)j) * 1.4426950408889634f;
    hipLaunchKernelGGL(emd_left_kernel, gl, dim3(kEmdThreads), 0, st, xyz1, xyz2, temp, n, m, lvl2);
    hipLaunchKernelGGL(emd_right_kernel, gr, dim3(kEmdThreads), 0, st, xyz1, xyz2, temp, n, m, lvl2);
    if (match)
      hipLaunchKernelGGL(emd_match_kernel<false>, gl, dim3(kEmdThreads), 0, st, xyz1, xyz2, match, temp, out, n, m, lvl2);
    else
      hipLaunchKernelGGL(emd_match_kernel<true>, gl, dim3(kEmdThreads), 0, st, xyz1, xyz2, match, temp, out, n, m, lvl2);
  }
  return (int)hipGetLastError();
}

extern "C" int gwtf_approx_match(const float* xyz1, const float* xyz2, float* match, float* temp, int b, int n, int m,
                                 void* stream) {
  if (!xyz1 || !xyz2 || !match || !temp || b <= 0 || n <= 0 || m <= 0) return GWTF_E_BADARG;
  return emd_schedule(xyz1, xyz2, match, temp, nullptr, b, n, m, (hipStream_t)stream);
}

extern "C" int gwtf_emd_cost(const float* xyz1, const float* xyz2, float* temp, float* out, int b, int n, int m,
                             void* stream) {
  if (!xyz1 || !xyz2 || !temp || !out || b <= 0 || n <= 0 || m <= 0) return GWTF_E_BADARG;
  return emd_schedule(xyz1, xyz2, nullptr, temp, out, b, n, m, (hipStream_t)stream);
}

extern "C" int gwtf_match_cost(const float* xyz1, const float* xyz2, const float* match, float* out, int b, int n, int m,
                               void* stream) {
  if (!xyz1 || !xyz2 || !match || !out || b <= 0 || n <= 0 || m <= 0) return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)b, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(matchcost_kernel, dim3((n + 255) / 256, b), dim3(256), 0, st, xyz1, xyz2, match, out, b, n, m);
  return (int)hipGetLastError();
}

extern "C" int gwtf_match_cost_grad(const float* xyz1, const float* xyz2, const float* match, float* grad1, float* grad2,
                                    int b, int n, int m, void* stream) {
  if (!xyz1 || !xyz2 || !match || !grad1 || !grad2 || b <= 0 || n <= 0 || m <= 0) return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(matchcost_grad1_kernel, dim3((n + 255) / 256, b), dim3(256), 0, st, xyz1, xyz2, match, grad1, b, n, m);
  hipLaunchKernelGGL(matchcost_grad2_kernel, dim3((m + 3) / 4, b), dim3(256), 0, st, xyz1, xyz2, match, grad2, b, n, m);
  return (int)hipGetLastError();
}
