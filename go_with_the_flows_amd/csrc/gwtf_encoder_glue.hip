// gwtf_encoder_glue.hip -- the small dense algebra between the encoder's training kernels (csrc/gwtf_encoder_train.hip), as a handful of
// launches instead of ~60 torch operators per step (library GEMMs on one compute unit, 64-replica sums, element-wise chains):
//   gwtf_stat_compact             sum of the R replicas of a statistic slab [R][n] -> [n] (fixed order)
//   gwtf_enc_train_mform          top layer, M form: M = W_3^T diag(q_3) W_3 scaled by a power of two and packed as MFMA fragments,
//                                 mconst = {W_3^T r_3, 2^-k, 0, 0, 0}
//   gwtf_enc_train_dw3_finish     dW_3 = s (.) S + q (.) (W_3 G_2) + r (x) sum_p a_2
//   gwtf_enc_train_dw0_finish     dW_0 = s (.) sums + q (.) (W_0 Mxx) + r (x) m
// (reference: the autograd of lib/networks/encoders.py:11-60's Conv1d/BatchNorm1d/ReLU stack in train mode; the decomposition is
// DESIGN.md section 3.4's.)  All sums run in a fixed order: the same bits on every run.
#include <hip/hip_runtime.h>
#include "../../include/gwtf.h"

namespace {

// 64 outputs x 4 replica slices per workgroup: a thread's loads are independent (all in flight at once), the four slices are added
// in a fixed order through LDS
template <int RU>
__global__ __launch_bounds__(256) void stat_compact_any_kernel(const float* __restrict__ slab, float* __restrict__ out, int R, int n) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6, i = blockIdx.x * 64 + o;
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    if constexpr (RU > 0) {             // R == 4 RU: compile-time trip count
      float v[RU];
#pragma unroll
      for (int r = 0; r < RU; ++r) v[r] = slab[(size_t)(sl * RU + r) * n + i];
#pragma unroll
      for (int r = 0; r < RU; r += 2) { a0 += v[r]; a1 += r + 1 < RU ? v[r + 1] : 0.f; }
    } else {
      const int per = (R + 3) / 4, r0 = sl * per, r1 = min(R, r0 + per);
      for (int r = r0; r < r1; ++r) a0 += slab[(size_t)r * n + i];
    }
  }
  part[sl][o] = a0 + a1;
  __syncthreads();
  if (sl == 0 && i < n) out[i] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
}

constexpr int kT = 16, kKC = 64;      // output tile edge, contraction chunk

// one 16 x 16 output tile per workgroup: acc = sum_k A(row, k) B(k, col), operands staged through LDS 64 k at a time
template <typename FA, typename FB>
__device__ __forceinline__ float tile_dot(FA A, FB B, int K, int row0, int col0, float (*As)[kT + 1], float (*Bs)[kT + 1]) {
  const int tid = threadIdx.x, ty = tid / kT, tx = tid % kT;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += kKC) {
#pragma unroll
    for (int e = 0; e < kKC * kT / 256; ++e) {
      const int idx = e * 256 + tid, kk = idx / kT, t = idx % kT;
      As[kk][t] = A(row0 + t, k0 + kk);
      Bs[kk][t] = B(k0 + kk, col0 + t);
    }
    __syncthreads();
#pragma unroll 16
    for (int kk = 0; kk < kKC; ++kk) acc = fmaf(As[kk][ty], Bs[kk][tx], acc);
    __syncthreads();
  }
  return acc;
}

// blocks [0, (C3/16)^2): tiles of M = W^T diag(q) W with the tile's max |M| -> tmax[block]; the last C3/16 blocks: mconst[0..C3) = W^T r
__global__ __launch_bounds__(256) void enc_mform_kernel(const float* __restrict__ W, const float* __restrict__ q, const float* __restrict__ r,
                                                        float* __restrict__ M, float* __restrict__ tmax, float* __restrict__ mconst,
                                                        int C3, int C4) {
  __shared__ float As[kKC][kT + 1], Bs[kKC][kT + 1];
  __shared__ float red[4];
  const int nt = C3 / kT, tid = threadIdx.x;
  if ((int)blockIdx.x >= nt * nt) {
    // W^T r for 16 outputs: 16 slices of the contraction side by side, added in slice order
    const int o = tid % kT, sl = tid / kT, i = kT * ((int)blockIdx.x - nt * nt) + o, per = C4 / 16;
    float a = 0.f;
    for (int c = sl * per; c < (sl + 1) * per; ++c) a = fmaf(W[(size_t)c * C3 + i], r[c], a);
    As[sl][o] = a;
    __syncthreads();
    if (sl == 0) {
      float t = 0.f;
      for (int k = 0; k < 16; ++k) t += As[k][o];
      mconst[i] = t;
    }
    return;
  }
  const int by = blockIdx.x / nt, bx = blockIdx.x % nt;
  const float v = tile_dot([&](int i, int c) { return q[c] * W[(size_t)c * C3 + i]; }, [&](int c, int j) { return W[(size_t)c * C3 + j]; },
                           C4, kT * by, kT * bx, As, Bs);
  M[(size_t)(kT * by + tid / kT) * C3 + kT * bx + tid % kT] = v;
  float m = fabsf(v);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) tmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// M 2^k (k = 8 - floor(log2 max|M|): the operand scale that keeps the f16 split of M in range) -> hi/lo fragment images in the unit
// order of enc_train_pack_kernel (gwtf_encoder_train.hip:69); block 0 also writes mconst[C3..C3+4) = {2^-k, 0, 0, 0}
__global__ __launch_bounds__(256) void enc_mform_pack_kernel(const float* __restrict__ M, const float* __restrict__ tmax, int n_tiles,
                                                             float* __restrict__ units, float* __restrict__ mconst, int C3) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < n_tiles; i += 256) m = fmaxf(m, tmax[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-37f);
  const int e2 = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;       // floor(log2 m), exactly (m is a normal number)
  const float scale = ldexpf(1.0f, 8 - e2), inv = ldexpf(1.0f, e2 - 8);
  if (blockIdx.x == 0 && tid < 4) mconst[C3 + tid] = tid == 0 ? inv : 0.f;
  const int t = blockIdx.x * 256 + tid, KS = C3 / 32;
  if (t >= (C3 / 16) * KS * 2 * 64) return;
  const int lane = t & 63, part = (t >> 6) & 1, unit = t >> 7;
  const int mrow = unit / KS, ks = unit % KS;
  const int row = 16 * mrow + (lane & 15), qd = lane >> 4;
  _Float16 out[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 16 * (e >> 2) + 4 * qd + (e & 3);
    const float v = M[(size_t)row * C3 + k] * scale;
    const _Float16 hi = (_Float16)v;
    out[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
  }
  *reinterpret_cast<float4*>(units + (size_t)unit * 512 + part * 256 + lane * 4) = *reinterpret_cast<const float4*>(out);
}

__global__ __launch_bounds__(256) void enc_dw3_finish_kernel(const float* __restrict__ s, const float* __restrict__ q,
                                                             const float* __restrict__ r, const float* __restrict__ S,
                                                             const float* __restrict__ W, const float* __restrict__ gram,
                                                             const float* __restrict__ asum, float* __restrict__ dW, int C3, int C4) {
  __shared__ float As[kKC][kT + 1], Bs[kKC][kT + 1];
  const int nt = C3 / kT, by = blockIdx.x / nt, bx = blockIdx.x % nt, tid = threadIdx.x;
  const float wg = tile_dot([&](int c, int j) { return W[(size_t)c * C3 + j]; }, [&](int j, int i) { return gram[(size_t)j * C3 + i]; },
                            C3, kT * by, kT * bx, As, Bs);
  const int c = kT * by + tid / kT, i = kT * bx + tid % kT;
  dW[(size_t)c * C3 + i] = (s[c] * S[(size_t)c * C3 + i] + q[c] * wg) + r[c] * asum[i];
}

// layer 0 (3 -> C1): every sum its gradient needs exists already.  red5 [5][C1] rows 2..4 = sum_p dz x_e; mom12 = {sum x (3), sum x x^T
// upper triangle (6), ...}
__global__ __launch_bounds__(256) void enc_dw0_finish_kernel(const float* __restrict__ bconst, const float* __restrict__ red5,
                                                             const float* __restrict__ W0, const float* __restrict__ m,
                                                             float* __restrict__ dW0, int C1) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 3 * C1) return;
  const int c = t / 3, e = t % 3;
  const float mxx[3][3] = {{m[3], m[4], m[5]}, {m[4], m[6], m[7]}, {m[5], m[7], m[8]}};
  float wm = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) wm = fmaf(W0[c * 3 + d], mxx[d][e], wm);
  dW0[t] = (bconst[c] * red5[(size_t)(2 + e) * C1 + c] + bconst[C1 + c] * wm) + bconst[2 * C1 + c] * m[e];
}

}  // namespace

extern "C" int gwtf_stat_compact(const float* slab, float* out, int replicas, int n, void* stream) {
  if (!slab || !out || replicas <= 0 || n <= 0) return GWTF_E_BADARG;
  const dim3 grid((n + 63) / 64), block(256);
  if (replicas == 64) hipLaunchKernelGGL(stat_compact_any_kernel<16>, grid, block, 0, (hipStream_t)stream, slab, out, replicas, n);
  else hipLaunchKernelGGL(stat_compact_any_kernel<0>, grid, block, 0, (hipStream_t)stream, slab, out, replicas, n);
  return (int)hipGetLastError();
}

extern "C" size_t gwtf_enc_train_mform_workspace_floats(int C3) { return (size_t)C3 * C3 + (size_t)(C3 / kT) * (C3 / kT); }

extern "C" int gwtf_enc_train_mform(const float* W3, const float* bconst3, float* workspace, float* units_m, float* mconst, int C3, int C4,
                                    void* stream) {
  if (!W3 || !bconst3 || !workspace || !units_m || !mconst || C3 <= 0 || C4 <= 0 || C3 % 32 || C4 % kKC || C4 % 16) return GWTF_E_BADARG;
  const int nt = C3 / kT;
  float* M = workspace;
  float* tmax = workspace + (size_t)C3 * C3;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(enc_mform_kernel, dim3(nt * nt + nt), dim3(256), 0, st, W3, bconst3 + C4, bconst3 + 2 * C4, M, tmax, mconst, C3, C4);
  const int total = (C3 / 16) * (C3 / 32) * 2 * 64;
  hipLaunchKernelGGL(enc_mform_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, M, tmax, nt * nt, units_m, mconst, C3);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_dw3_finish(const float* bconst3, const float* S, const float* W3, const float* gram, const float* a2sum,
                                         float* dW3, int C3, int C4, void* stream) {
  if (!bconst3 || !S || !W3 || !gram || !a2sum || !dW3 || C3 <= 0 || C4 <= 0 || C3 % kKC || C4 % kT) return GWTF_E_BADARG;
  hipLaunchKernelGGL(enc_dw3_finish_kernel, dim3((C4 / kT) * (C3 / kT)), dim3(256), 0, (hipStream_t)stream, bconst3, bconst3 + C4,
                     bconst3 + 2 * C4, S, W3, gram, a2sum, dW3, C3, C4);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_dw0_finish(const float* bconst0, const float* red5, const float* W0, const float* mom12, float* dW0,
                                         int C1, void* stream) {
  if (!bconst0 || !red5 || !W0 || !mom12 || !dW0 || C1 <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(enc_dw0_finish_kernel, dim3((3 * C1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, bconst0, red5, W0, mom12, dW0,
                     C1);
  return (int)hipGetLastError();
}
