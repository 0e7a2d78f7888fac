// gwtf_encoder_train.hip -- PointNet cloud encoder under model.train(): batch-statistic BatchNorm forward, max-pool, and the
// whole backward, layer at a time.
// Reference: PointNetCloudEncoder (lib/networks/encoders.py:9-28: SharedDot -> BatchNorm1d -> ReLU, 3 -> 64 -> 128 -> 256 ->
// 512), the pooling of its caller (lib/networks/models.py:127-128: torch.max(features, dim=2)[0]) and what loss.backward()
// (lib/networks/training.py:54) computes for them; SyncBatchNorm (train_ae.py:152) = the same kernels with the statistic
// sums all-reduced by the host between a layer's kernel and its fold.
//
// Notation: layer l = 0..3, y_l = W_l a_{l-1} (pre-BatchNorm, a_{-1} = x), yhat_l = (y_l - mean_l) rstd_l,
// a_l = relu(gamma_l yhat_l + beta_l) = relu(s_l y_l + t_l), pooled[b][c] = max_n a_3[b][c][n].
// Batch statistics make every layer a global barrier (the mean / variance of y_l over all B N points must exist before
// a_l does), so the pass runs layer at a time with y_1, y_2, y_3 kept in HBM in the reference's own (B, C, N) layout
// (y_0 = W_0 x is three FMAs and is recomputed wherever it is needed).  288 GB of HBM make 0.5 GB of saved
// pre-activations a non-issue; what the library path (rocBLAS GEMM + MIOpen BatchNorm + elementwise kernels) pays on top is
// every intermediate written and read again between those calls, and fp32 matrix throughput: here each layer is ONE kernel
//   prologue  a_{l-1} = relu(s y + t) applied while loading the B operand (8 channels x 1 point per lane: 64-byte rows)
//   product   v_mfma_f32_16x16x32_f16, three-product split (W_hi a_hi + W_hi a_lo + W_lo a_hi), weights streamed
//             L2 -> LDS by LDS-DMA in 32-KiB chunks shared by 8 wavefronts (as gwtf_encoder.hip)
//   epilogue  y_l stored, sum y / sum y^2 per channel (DPP row sums -> LDS -> 64 global replicas), max |y|
// Backward, per layer from the top:  dy_l = rstd (gamma gm_l - m1 - yhat_l m2) = s gm_l + Q y_l + R  with per-channel
// constants from the two BatchNorm-backward sums (m1 = gamma mean(gm), m2 = gamma mean(gm yhat)), gm_l = dL/da_l masked by
// a_l > 0.  One kernel per layer computes dL/da_{l-1} = W_l^T dy_l (prologue builds dy_l from y_l and gm_l, same MFMA
// core with the transposed fragments) and, in its epilogue, masks it with a_{l-1} > 0, stores it, and accumulates the NEXT
// layer's two sums; the top layer's gm_3 is non-zero only at the arg-max point of each (shape, channel), which the
// prologue gets by comparing the point index.  The weight gradients dW_l = sum_p dy_l(p) a_{l-1}(p)^T contract over
// POINTS: both operands are read K-major straight from the (B, C, N) arrays (a lane = one channel x 8 consecutive points),
// every wavefront owns a 64 x 64 block of dW_l over one slice of the points and writes a partial that a second kernel sums in
// a fixed order.  Gradients have no natural scale, so every f16-split gradient operand is multiplied by a power of two
// chosen from a bound on |dy_l| (host-free: the bound comes from max |gm_l| and max |y_l| tracked by the producing
// kernels) and the result is scaled back: exact.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gwtf_device.h"

#ifndef GWTF_ENC_DBG
#define GWTF_ENC_DBG 0      // timing experiments only (tools/ab_build.sh): 1 no dA stores, 2 no epilogue loads, 3 no prologue loads, 4 no dW operand loads
#endif
using namespace gwtf_dev;

namespace {

constexpr int kChunk = 8192;        // floats per LDS chunk (32 KiB = 16 units of [hi|lo] 16x32 f16 fragments)
constexpr int kThreads = 512;
constexpr int kR = GWTF_STAT_REPLICAS;
constexpr int kC[5] = {3, 64, 128, 256, 512};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// LAYOUT OF THE STORED ACTIVATIONS (y_1, y_2 and the masked gradients dA_2, dA_1; internal to the pipeline, never seen by the caller):
// [B][N / 32][C][32] -- tiles of 32 points with all channels of a tile contiguous (C x 128 bytes), not the reference's (B, C, N).
// Every consumer here takes 32 points x ALL channels at a time (a wave of the forward / backward kernels, a k-step of the weight-
// gradient kernels): in (B, C, N) that is C separate 128-byte pieces 8 KB apart (every access another DRAM page: the weight-gradient
// kernels ran at 2 TB/s), in tiles it is ONE contiguous block.  N is padded to whole tiles in the allocation, plus one spare tile
// (gwtf_enc_train_act_floats).
// the same split in two: the point's offset inside channel 0 of its tile (computed once per point) + 32 floats per channel
// A point beyond the cloud (a lane of the last workgroup of a ragged N) maps into ONE spare tile behind the B shapes' tiles, so that
// loads and stores need no bounds branch (written as `if (n < N)` every access compiled to its own exec-masked block).
__device__ __forceinline__ size_t tix_point(int b, int C, int n, int N, int B) {
  const int NT = (N + 31) >> 5;
  return n < N ? ((size_t)b * NT + (n >> 5)) * C * 32 + (n & 31) : (size_t)B * NT * C * 32 + (n & 31);
}
__device__ __forceinline__ size_t tix(int b, int C, int ch, int n, int N) {
  const int NT = (N + 31) >> 5;
  return (((size_t)b * NT + (n >> 5)) * C + ch) * 32 + (n & 31);
}

// units [m][ks] of [hi | lo] x 64 lanes x 8 f16: A-operand row 16 m + (lane & 15), k-slot (ks, q, e) <-> contraction index
// 32 ks + 16 (e >> 2) + 4 q + (e & 3) (the map gwtf_encoder.hip uses).  transposed = 0: A[row][k] = W[row][k] (W [rows][kdim]);
// transposed = 1: A[row][k] = W[k][row] (W [kdim][rows]).
__global__ void enc_train_pack_kernel(const float* __restrict__ W, float* __restrict__ units, int rows, int kdim, int transposed) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int KS = kdim / 32;
  if (t >= (rows / 16) * KS * 2 * 64) return;
  const int lane = t & 63, part = (t >> 6) & 1, unit = t >> 7;
  const int m = unit / KS, ks = unit % KS;
  const int row = 16 * m + (lane & 15), q = lane >> 4;
  _Float16 out[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 16 * (e >> 2) + 4 * q + (e & 3);
    const float v = transposed ? W[(size_t)k * rows + row] : W[(size_t)row * kdim + k];
    const _Float16 hi = (_Float16)v;
    out[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
  }
  *reinterpret_cast<float4*>(units + (size_t)unit * 512 + part * 256 + lane * 4) = *reinterpret_cast<const float4*>(out);
}

// the six fragment images of layers 1..3 (forward + transposed) in ONE launch: blockIdx.y = job
struct PackJobs { const float* W[6]; float* units[6]; int rows[6], kdim[6], transposed[6]; };
__global__ void enc_train_pack_all_kernel(const PackJobs J) {
  const int j = blockIdx.y;
  const float* __restrict__ W = J.W[j];
  float* __restrict__ units = J.units[j];
  const int rows = J.rows[j], kdim = J.kdim[j], transposed = J.transposed[j];
  const int KS = kdim / 32, total = (rows / 16) * KS * 2 * 64;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int lane = t & 63, part = (t >> 6) & 1, unit = t >> 7;
    const int m = unit / KS, ks = unit % KS;
    const int row = 16 * m + (lane & 15), q = lane >> 4;
    _Float16 out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * ks + 16 * (e >> 2) + 4 * q + (e & 3);
      const float v = transposed ? W[(size_t)k * rows + row] : W[(size_t)row * kdim + k];
      const _Float16 hi = (_Float16)v;
      out[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
    }
    *reinterpret_cast<float4*>(units + (size_t)unit * 512 + part * 256 + lane * 4) = *reinterpret_cast<const float4*>(out);
  }
}

// sum x (3) and sum x x^T (6: xx xy xz yy yz zz) over all points -> mom[replica][12] (pre-zeroed)
__global__ __launch_bounds__(256) void enc_xmom_kernel(const float* __restrict__ x, float* __restrict__ mom, int B, int N) {
  __shared__ float part[4][9];
  const int b = blockIdx.y;
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) {
    const int n = blockIdx.x * 1024 + i * 256 + threadIdx.x;
    if (n < N) {
      const float a = x[((size_t)b * 3 + 0) * N + n], c = x[((size_t)b * 3 + 1) * N + n], d = x[((size_t)b * 3 + 2) * N + n];
      s[0] += a; s[1] += c; s[2] += d;
      s[3] = fmaf(a, a, s[3]); s[4] = fmaf(a, c, s[4]); s[5] = fmaf(a, d, s[5]);
      s[6] = fmaf(c, c, s[6]); s[7] = fmaf(c, d, s[7]); s[8] = fmaf(d, d, s[8]);
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int j = 0; j < 9; ++j) part[wave][j] = s[j];
  __syncthreads();
  if (threadIdx.x < 9)
    atomicAdd(&mom[((blockIdx.y * gridDim.x + blockIdx.x) % kR) * 12 + threadIdx.x],
              (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

__device__ __forceinline__ void bn_running_update(float* rm, float* rv, int c, double mean, double var, double n, float momentum) {
  if (rm) rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mean);
  if (rv) rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * var * (n > 1.0 ? n / (n - 1.0) : 1.0));
}

// layer 0: statistics of y_0 = W_0 x are exact functions of the moments of x.  mom12 = sums over all points (and ranks).
// aff [4][C0] = s, t, mean, rstd;  table0 [C0][4] = (s W_0 row, t)
__global__ void enc_fold0_kernel(const float* __restrict__ mom12, double n, const float* __restrict__ W0,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* rm, float* rv,
                                 float momentum, float* __restrict__ aff, float* __restrict__ table0, int C0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C0) return;
  const double w0 = W0[c * 3], w1 = W0[c * 3 + 1], w2 = W0[c * 3 + 2];
  const double mean = (w0 * mom12[0] + w1 * mom12[1] + w2 * mom12[2]) / n;
  const double e2 = (w0 * w0 * mom12[3] + 2 * w0 * w1 * mom12[4] + 2 * w0 * w2 * mom12[5] + w1 * w1 * mom12[6] +
                     2 * w1 * w2 * mom12[7] + w2 * w2 * mom12[8]) / n;
  double var = e2 - mean * mean;
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)GWTF_BN_EPS);
  const double s = gamma[c] * rstd, t = beta[c] - mean * s;
  bool poison = false;               // a non-finite input point: every channel of layer 0 is affected
  for (int i = 0; i < 9; ++i) poison |= gwtf_nonfinite(mom12[i]);
  poison |= gwtf_nonfinite((float)s) || gwtf_nonfinite((float)t);
  const float qnan = __builtin_bit_cast(float, 0x7fc00000);
  aff[c] = poison ? qnan : (float)s; aff[C0 + c] = poison ? qnan : (float)t;
  aff[2 * C0 + c] = poison ? qnan : (float)mean; aff[3 * C0 + c] = poison ? qnan : (float)rstd;
  table0[c * 4] = (float)(w0 * s); table0[c * 4 + 1] = (float)(w1 * s); table0[c * 4 + 2] = (float)(w2 * s); table0[c * 4 + 3] = (float)t;
  bn_running_update(rm, rv, c, mean, var, n, momentum);
}

// sums [2][C] = sum y, sum y^2 over all points (replicas and ranks already summed).  One workgroup.
// Non-finite values must reach the loss (the reference's isnan guard, training.py:43-46), but a ReLU compiled without NaN
// semantics turns a NaN activation into 0: the FOLDS carry the poison instead.  A layer whose statistics are not finite (a NaN
// anywhere in y_l shows in sum y) -- or whose input layer was already poisoned (aff_prev) -- gets NaN in every entry of its aff;
// the pooling kernel turns a non-finite aff into NaN codes.
__global__ __launch_bounds__(512) void enc_fold_kernel(const float* __restrict__ sums, int C, double n, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* rm, float* rv, float momentum,
                                                       float* __restrict__ aff, const float* __restrict__ aff_prev) {
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = aff_prev ? (int)gwtf_nonfinite(aff_prev[0]) : 0;
  __syncthreads();
  int bad = 0;
  for (int c = threadIdx.x; c < C; c += blockDim.x) bad |= (int)(gwtf_nonfinite(sums[c]) || gwtf_nonfinite(sums[C + c]));
  if (bad) atomicOr(&s_bad, 1);
  __syncthreads();
  const bool poison = s_bad != 0;
  const float qnan = __builtin_bit_cast(float, 0x7fc00000);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double mean = (double)sums[c] / n;
    double var = (double)sums[C + c] / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)GWTF_BN_EPS);
    const double s = gamma[c] * rstd;
    aff[c] = poison ? qnan : (float)s;
    aff[C + c] = poison ? qnan : (float)(beta[c] - mean * s);
    aff[2 * C + c] = poison ? qnan : (float)mean;
    aff[3 * C + c] = poison ? qnan : (float)rstd;
    bn_running_update(rm, rv, c, mean, var, n, momentum);
  }
}

// f16 hi/lo split of an accumulator-layout tile into elements 4*half..+3 of a B fragment; RELU: max(., 0) first
template <bool RELU>
__device__ __forceinline__ void split_into(const f32x4& a, f16x8& hi, f16x8& lo, int half) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x2 v = {RELU ? fmaxf(a[2 * p], 0.f) : a[2 * p], RELU ? fmaxf(a[2 * p + 1], 0.f) : a[2 * p + 1]};
    f16x2 h, l;
    split_pair(v, h, l);
    hi[4 * half + 2 * p] = h[0]; hi[4 * half + 2 * p + 1] = h[1];
    lo[4 * half + 2 * p] = l[0]; lo[4 * half + 2 * p + 1] = l[1];
  }
}

template <int KS, int NB>
__device__ __forceinline__ void tile_mfma(const float* __restrict__ unit0, int lane, const f16x8 (&bhi)[KS][NB],
                                          const f16x8 (&blo)[KS][NB], f32x4 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f16x8 ahi = *reinterpret_cast<const f16x8*>(unit0 + ks * 512 + lane * 4);
    const f16x8 alo = *reinterpret_cast<const f16x8*>(unit0 + ks * 512 + 256 + lane * 4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi[ks][nb], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo[ks][nb], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi[ks][nb], acc[nb], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
  return v;
}

// order-preserving map float -> unsigned (larger float <-> larger key), and the 64-bit arg-max key: ties go to the smaller n
__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned k) {
  return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ unsigned long long argmax_key(unsigned ordered, int n) {
  return ((unsigned long long)ordered << 32) | (unsigned)(0x7fffffff - n);
}

// ---- forward of one layer: y_out = W . relu(s in + t)  (FIRST: in = x, relu(W_0' x + t_0) from table0) -------------------
// LAST (the top layer): y is not stored.  Its BatchNorm + ReLU + max over points is monotone in y per channel, so the pooled
// code only needs max_n y and min_n y per (shape, channel) (which one: the sign of s, known after the fold): 64-bit keys
// (order-preserving bits of y | inverted point index) combined by integer max -- DPP row max to find the candidates, one
// exec-masked LDS atomic per (row, tile), one global atomic per (workgroup, channel).
template <int CIN, int COUT, bool FIRST, bool LAST, int NB>
__global__ __launch_bounds__(kThreads, NB == 1 ? 4 : 1) void enc_train_fwd_kernel(const float* __restrict__ in, const float* __restrict__ in_tab,
                                                                 const float* __restrict__ units, float* __restrict__ y_out,
                                                                 float* __restrict__ sums, float* __restrict__ ymax,
                                                                 unsigned long long* __restrict__ kmax,
                                                                 unsigned long long* __restrict__ kmin, int B, int N,
                                                                 const float* __restrict__ gamma_out) {
  constexpr int KS = CIN / 32, MT = COUT / 16, TM = 16 / KS, NCH = MT / TM;
  static_assert(16 % KS == 0 && MT % TM == 0, "whole chunks");
  __shared__ __attribute__((aligned(16))) float lds[2][kChunk];
  __shared__ __attribute__((aligned(16))) float tab[FIRST ? 4 * CIN : 2 * CIN];
  __shared__ float wsum[2][COUT];
  __shared__ float wmax[8];
  __shared__ unsigned long long wkey[LAST ? COUT : 1];
  __shared__ float wsgn[LAST ? COUT : 1];     // LAST: -1 where this layer's BatchNorm weight is negative (the pooled extreme is the MINIMUM of y there)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, n_wave0 = blockIdx.x * (128 * NB) + wave * 16 * NB;
  if (LAST)
    for (int t = tid; t < COUT; t += kThreads) {
      wkey[t] = 0ull;
      wsgn[t] = gamma_out[t] < 0.f ? -1.0f : 1.0f;
    }

  auto stage = [&](int buf, int g) {
#pragma unroll
    for (int i = 0; i < kChunk / 256 / 8; ++i) {
      const int piece = wave + 8 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(units + (size_t)g * kChunk + piece * 256 + lane * 4),
                                       (lds_void*)&lds[buf][piece * 256], 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int t = tid; t < (FIRST ? 4 * CIN : 2 * CIN); t += kThreads) tab[t] = in_tab[t];
  for (int t = tid; t < 2 * COUT; t += kThreads) (&wsum[0][0])[t] = 0.f;

  int n[NB];
  bool valid[NB];
  size_t tin[NB], tout[NB];          // the point's base offsets in the tiled input / output arrays
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    n[nb] = n_wave0 + 16 * nb + i16;
    valid[nb] = n[nb] < N;
    tin[nb] = tix_point(b, CIN, n[nb], N, B);
    tout[nb] = tix_point(b, COUT, n[nb], N, B);
  }
  __syncthreads();

  f16x8 bhi[KS][NB], blo[KS][NB];
  if (FIRST) {
    float px[NB], py[NB], pz[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      // (unconditional loads from a clamped address: `valid ? load : 0` compiles to an exec-masked branch per load -- docs/LOG.md)
      const int nc = min(n[nb], N - 1);
      px[nb] = in[((size_t)b * 3 + 0) * N + nc];
      py[nb] = in[((size_t)b * 3 + 1) * N + nc];
      pz[nb] = in[((size_t)b * 3 + 2) * N + nc];
    }
    const float4* t4 = reinterpret_cast<const float4*>(tab);
#pragma unroll
    for (int m = 0; m < CIN / 16; ++m) {
      float4 w[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = t4[16 * m + 4 * q + r];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = w[r].x * px[nb] + (w[r].y * py[nb] + (w[r].z * pz[nb] + w[r].w));
        split_into<true>(a, bhi[m >> 1][nb], blo[m >> 1][nb], m & 1);
      }
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int c0 = 32 * ks + 16 * half + 4 * q;
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&tab[c0]), t4 = *reinterpret_cast<const f32x4*>(&tab[CIN + c0]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x4 a;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = in[tin[nb] + (size_t)(c0 + r) * 32];     // (padded tiles: in bounds; points beyond N never leave the wave)
            a[r] = fmaf(s4[r], v, t4[r]);
          }
          split_into<true>(a, bhi[ks][nb], blo[ks][nb], half);
        }
      }
  }

  int g = 0;
  float amax = 0.f;
#pragma unroll 1
  for (int ci = 0; ci < NCH; ++ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (g + 1 < NCH) stage((g + 1) & 1, g + 1);
    const float* L = lds[g & 1];
    ++g;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int m = ci * TM + t;
      f32x4 acc[NB];
      tile_mfma<KS, NB>(L + t * KS * 512, lane, bhi, blo, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = 16 * m + 4 * q + r;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float v = valid[nb] ? acc[nb][r] : 0.f;
          if (!LAST) y_out[tout[nb] + (size_t)ch * 32] = v;      // (unconditional: beyond N it writes zeros into the tile's padding)
          s1 += v;
          s2 = fmaf(v, v, s2);
          amax = fmaxf(amax, fabsf(v));
        }
        s1 = row_sum16(s1);
        s2 = row_sum16(s2);
        if (i16 == 0) {
          atomicAdd(&wsum[0][ch], s1);
          atomicAdd(&wsum[1][ch], s2);
        }
        if (LAST) {
          // BatchNorm + ReLU + max over the points is monotone in y per channel, rising or falling with the sign of s = gamma rstd,
          // i.e. of the layer's BatchNorm weight, known before the pass: only ONE extreme per channel is tracked -- the maximum of
          // sgn y (the first version tracked maximum and minimum and let the fold choose: twice the select / DPP / atomic work in a
          // kernel whose epilogue, not its MFMAs, sets the time).  This lane's best of its NB points, then the row's; only the lanes
          // that hold it touch the LDS.
          const float big = 3.0e38f, sgn = wsgn[ch];
          float hi = -big;
          int nhi = 0;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float v = acc[nb][r] * sgn;
            if (valid[nb] && v > hi) { hi = v; nhi = n[nb]; }
          }
          const float rhi = row_max16(hi);
          if (hi == rhi && hi > -big) atomicMax(&wkey[ch], argmax_key(ordered_bits(hi), nhi));
        }
      }
    }
  }
  amax = wave_max(amax);
  if (lane == 0) wmax[wave] = amax;
  __syncthreads();
  float* dst = sums + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) % kR) * 2 * COUT;
  for (int t = tid; t < 2 * COUT; t += kThreads) atomicAdd(&dst[t], (&wsum[0][0])[t]);
  if (LAST)
    for (int t = tid; t < COUT; t += kThreads) {
      // maximum of y where gamma >= 0 (kmax), minimum where gamma < 0 (kmin; the key of -y is the key enc_pool_finalize_kernel
      // decodes for a minimum: ordered_bits(-y) == ~ordered_bits(y))
      unsigned long long* dstk = wsgn[t] < 0.f ? kmin : kmax;
      atomicMax(&dstk[(size_t)b * COUT + t], wkey[t]);
    }
  if (tid == 0) {
    float m = wmax[0];
    for (int w = 1; w < 8; ++w) m = fmaxf(m, wmax[w]);
    atomicMax(reinterpret_cast<int*>(ymax), __builtin_bit_cast(int, m));   // non-negative floats order like their bit patterns
  }
}

// ---- pooled[b][c] = max_n relu(s y_3 + t) = relu(s (s >= 0 ? max_n y_3 : min_n y_3) + t), its (first) arg-max, y_3 there ------
// A non-finite statistic or extreme (diverged weights, a NaN point) gives NaN, as torch's batch_norm + relu + max would.
__global__ void enc_pool_finalize_kernel(const unsigned long long* __restrict__ kmax, const unsigned long long* __restrict__ kmin,
                                         const float* __restrict__ aff, float* __restrict__ pooled, int* __restrict__ amax,
                                         float* __restrict__ ystar, int B, int C, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int c = i % C;
  const float s = aff[c], t = aff[C + c];
  // The forward kernel wrote ONE of the two keys per channel, chosen by the sign of the BatchNorm WEIGHT; the other is still the zero
  // it was initialised with.  Decide by what was written, not by the sign of s = gamma * rstd: a tiny negative gamma under a large
  // variance flushes s to -0 / +0 and `s < 0` then picks the key that never received a point.
  const bool use_max = kmin[i] == 0ull;
  const unsigned long long key = use_max ? kmax[i] : kmin[i];
  const unsigned ob = (unsigned)(key >> 32);
  const float y = from_ordered_bits(use_max ? ob : ~ob);
  const float a = fmaf(s, y, t);
  const bool bad = gwtf_nonfinite(a) || gwtf_nonfinite(s) || gwtf_nonfinite(t);
  pooled[i] = bad ? __builtin_bit_cast(float, 0x7fc00000) : fmaxf(a, 0.f);
  // a key that never received a point (every y of the row NaN) decodes to an index far outside the cloud: clamp, the backward
  // indexes per-point tables with it
  const int idx = 0x7fffffff - (int)(unsigned)(key & 0xffffffffu);
  amax[i] = key == 0ull ? 0 : min(max(idx, 0), N - 1);
  ystar[i] = y;
}

// ---- top of the backward: gp = g_pooled where the pooled activation is > 0; the two BatchNorm-backward sums of layer 3 -----
__global__ __launch_bounds__(256) void enc_top_kernel(const float* __restrict__ g_pooled, const float* __restrict__ pooled,
                                                      const float* __restrict__ ystar, const float* __restrict__ aff, float* __restrict__ gp,
                                                      float* __restrict__ sums, float* __restrict__ gmax, int B, int C) {
  // a workgroup = 32 channels x 8 slices of the shapes (a thread per channel walking all B shapes was B dependent round trips: 32 us
  // for 64 shapes); the slices' partial sums are combined in slice order: the same sums on every run
  __shared__ float s_db[8][32], s_dg[8][32], s_mx[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  float db = 0.f, dg = 0.f, mx = 0.f;
  if (c < C) {
    const float mean = aff[2 * C + c], rstd = aff[3 * C + c];
#pragma unroll 4
    for (int b = sl; b < B; b += 8) {
      const float g = pooled[b * C + c] > 0.f ? g_pooled[b * C + c] : 0.f;
      gp[b * C + c] = g;
      db += g;
      dg = fmaf(g, (ystar[b * C + c] - mean) * rstd, dg);
      mx = fmaxf(mx, fabsf(g));
    }
  }
  s_db[sl][cl] = db; s_dg[sl][cl] = dg; s_mx[sl][cl] = mx;
  __syncthreads();
  if (sl == 0 && c < C) {
#pragma unroll
    for (int u = 1; u < 8; ++u) { db += s_db[u][cl]; dg += s_dg[u][cl]; mx = fmaxf(mx, s_mx[u][cl]); }
    sums[c] = db;
    sums[C + c] = dg;
    atomicMax(reinterpret_cast<int*>(gmax), __builtin_bit_cast(int, mx));
  }
}

// ---- the arg-max rows of the top layer's backward -------------------------------------------------------------------------
// Per shape b: the distinct arg-max points get consecutive row numbers (slot_of[b][n] = row or -1), and
// extra[b][row][:] = sum over the channels c whose arg-max is that point of coef[b][c] W_3[c][:]   (coef = s_c gp[b][c]).
// Channels are visited in ascending order: deterministic sums.
// Two kernels: the row tables (this one: per shape the offsets [C + 1] and the channel list [C] of its rows), then one
// wavefront per row for the sums (enc_top_rows_kernel).
__global__ __launch_bounds__(256) void enc_top_scatter_kernel(const float* __restrict__ coef, const float* __restrict__ scale,
                                                              const int* __restrict__ amax,
                                                              int* __restrict__ slot_of, int* __restrict__ row_off,
                                                              int* __restrict__ row_list, int N, int C) {
  extern __shared__ int dyn[];                 // [N] row of a point | -1
  __shared__ int s_am[512], s_cnt[512], s_off[512], s_scan[256];
  __shared__ __align__(16) int s_list[512];
  __shared__ float s_cf[512];
  int* slot = dyn;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int n = tid; n < N; n += 256) slot[n] = -1;
  for (int c = tid; c < C; c += 256) {
    const int am = amax[(size_t)b * C + c];
    s_am[c] = min(max(am, 0), N - 1);           // defensive: the table below is indexed with it
    s_cf[c] = coef[(size_t)b * C + c] * (scale ? scale[c] : 1.0f);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256)
    if (s_cf[c] != 0.f) slot[s_am[c]] = 0;     // mark
  __syncthreads();
  // rows in ascending point order: each thread numbers a contiguous chunk of points
  const int per = (N + 255) / 256, n0 = tid * per, n1 = min(N, n0 + per);
  int mine = 0;
  for (int n = n0; n < n1; ++n) mine += slot[n] == 0;
  s_scan[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int row = s_scan[tid] - mine;
  for (int n = n0; n < n1; ++n) {
    const int r = slot[n] == 0 ? row++ : -1;
    slot[n] = r;
    slot_of[(size_t)b * N + n] = r;
  }
  const int rows = s_scan[255];
  __syncthreads();
  // channel lists per row, ascending channel order inside a row (the sums of enc_top_rows_kernel must not depend on the order LDS
  // atomics land in): count per row (integer atomics: order-free), offsets (scan), and each channel's place inside its row =
  // the number of SMALLER channels with the same row, counted by brute force over broadcast LDS reads (C^2 / 256 compares per
  // thread).  (The first version filled the lists by atomics and insertion-sorted every row with one thread: a shape whose
  // arg-max points coincide for ~60 channels -- one extreme point -- held the whole launch for 80 us.)
  for (int r = tid; r < C; r += 256) s_cnt[r] = 0;
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int r = s_cf[c] != 0.f ? slot[s_am[c]] : -1;
    s_list[c] = r;                                   // (s_list doubles as the channel -> row table until the fill below)
    if (r >= 0) atomicAdd(&s_cnt[r], 1);
  }
  __syncthreads();
  {
    const int a = 2 * tid < C ? s_cnt[2 * tid] : 0, c2 = 2 * tid + 1 < C ? s_cnt[2 * tid + 1] : 0;
    s_scan[tid] = a + c2;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int v = tid >= off ? s_scan[tid - off] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const int base = s_scan[tid] - (a + c2);
    if (2 * tid < C) s_off[2 * tid] = base;
    if (2 * tid + 1 < C) s_off[2 * tid + 1] = base + a;
  }
  __syncthreads();
  int place[2] = {0, 0}, myrow[2] = {-1, -1};
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + 256 * u < C) myrow[u] = s_list[tid + 256 * u];
  for (int c2 = 0; c2 < C; c2 += 4) {                // every lane reads the same words: LDS broadcast
    const int4 rr = *reinterpret_cast<const int4*>(&s_list[c2]);
    const int rv[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) place[u] += (c2 + e < tid + 256 * u && rv[e] == myrow[u]) ? 1 : 0;
  }
  __syncthreads();                                    // everyone has read the channel -> row table
  int dst[2] = {-1, -1};
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (myrow[u] >= 0) dst[u] = s_off[myrow[u]] + place[u];
  for (int c = tid; c < C; c += 256) s_list[c] = 0;   // entries beyond the number of hits are never read (row_off bounds them)
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (dst[u] >= 0) s_list[dst[u]] = tid + 256 * u;
  __syncthreads();
  // s_off is an exclusive prefix sum over ALL C entries (rows beyond `rows` have no hits): s_off[r + 1] ends row r
  for (int r = tid; r < C; r += 256) {
    row_off[(size_t)b * (C + 2) + r] = s_off[r];
    row_list[(size_t)b * C + r] = s_list[r];
  }
  if (tid == 0) {
    row_off[(size_t)b * (C + 2) + C] = s_off[C - 1] + s_cnt[C - 1];      // total number of hits
    row_off[(size_t)b * (C + 2) + C + 1] = rows;
  }
}

// extra[b][r][:] = sum over the row's channels (ascending) of coef[b][c] W_3[c][:]; one wavefront per (shape, row), a lane = 4 columns
__global__ __launch_bounds__(256) void enc_top_rows_kernel(const float* __restrict__ coef, const float* __restrict__ scale,
                                                           const int* __restrict__ row_off,
                                                           const int* __restrict__ row_list, const float* __restrict__ W3,
                                                           float* __restrict__ extra, int B, int C, int CP) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= B * C) return;
  const int b = w / C, r = w % C;
  const int* ro = row_off + (size_t)b * (C + 2);
  if (r >= ro[C + 1]) return;
  const int o = ro[r], end = ro[r + 1];
  for (int j = 4 * lane; j < CP; j += 256) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = o; k < end; ++k) {
      const int c = row_list[(size_t)b * C + k];
      const float g = coef[(size_t)b * C + c] * (scale ? scale[c] : 1.0f);
      const float4 wv = *reinterpret_cast<const float4*>(&W3[(size_t)c * CP + j]);
      a.x = fmaf(g, wv.x, a.x); a.y = fmaf(g, wv.y, a.y); a.z = fmaf(g, wv.z, a.z); a.w = fmaf(g, wv.w, a.w);
    }
    *reinterpret_cast<float4*>(&extra[((size_t)b * C + r) * CP + j]) = a;
  }
}

// ---- per-channel constants of dy = s gm + Q y + R and the power-of-two operand scale ---------------------------------------
// sums [2][C] = sum gm, sum gm yhat over ALL points (replicas and ranks summed).  bconst [3][C] + {up, down, -, -}
__global__ __launch_bounds__(512) void enc_bwd_consts_kernel(const float* __restrict__ sums, int C, double n,
                                                             const float* __restrict__ gamma, const float* __restrict__ aff,
                                                             const float* __restrict__ gmax, const float* __restrict__ ymax,
                                                             float* __restrict__ bconst) {
  __shared__ float red[512];
  float bound = 0.f;
  const float gm = gmax ? *gmax : 0.f, ym = ymax ? *ymax : 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double mean = aff[2 * C + c], rstd = aff[3 * C + c];
    const double m1 = (double)gamma[c] * sums[c] / n, m2 = (double)gamma[c] * sums[C + c] / n;
    const float s = aff[c], Q = (float)(-rstd * rstd * m2), R = (float)(-rstd * m1 + rstd * rstd * m2 * mean);
    bconst[c] = s;
    bconst[C + c] = Q;
    bconst[2 * C + c] = R;
    bound = fmaxf(bound, fabsf(s) * gm + fabsf(Q) * ym + fabsf(R));
  }
  red[threadIdx.x] = bound;
  __syncthreads();
  for (int off = 256; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // bound in [2^(e-127), 2^(e-126)): up = 2^(14 - (e - 126)) maps it below 2^15 (f16 max 65504)
    const int e = (__builtin_bit_cast(int, red[0]) >> 23) & 0xff;
    const bool ok = e >= 16 && e <= 240;
    bconst[3 * C] = ok ? __builtin_bit_cast(float, (127 + 14 - (e - 126)) << 23) : 1.0f;
    bconst[3 * C + 1] = ok ? __builtin_bit_cast(float, (127 - 14 + (e - 126)) << 23) : 1.0f;
    bconst[3 * C + 2] = 0.f;
    bconst[3 * C + 3] = 0.f;
  }
}

// ---- backward of one layer: dL/da_{l-1} = W_l^T dy_l, masked by a_{l-1} > 0; sums for the layer below -------------------------
// CIN = C_{l-1} (rows of the product), COUT = C_l (contraction).  BOTTOM: layer l-1 = 0 (y_0 from x, nothing stored, the extra
// sums sum gm_0 x_d for dW_0).
// TOP (layer 3, COUT == CIN): dy_3 = s gm_3 + Q y_3 + R with y_3 = W_3 a_2 makes the product linear in a_2,
//     dL/da_2 (p) = M a_2(p) + v + sum over {c : argmax(b, c) = p} s_c gp[b][c] W_3[c][:],   M = W_3^T diag(Q) W_3,  v = W_3^T R,
// a 256-wide contraction on a_2 (read from y_2: neither y_3 nor a 512-wide B fragment is needed) plus, for the <= 512 arg-max
// points of a shape, one precomputed row each (`extra` [B][ex_rows][CIN], `slot_of` [B][N] = row or -1; enc_top_scatter_kernel).
// The fragment images hold M times a power of two; bconst = v [CIN] followed by {down}.
template <int CIN, int COUT, int NB, bool TOP, bool BOTTOM>
__global__ __launch_bounds__(kThreads, NB == 1 ? 4 : 1) void enc_train_bwd_kernel(const float* __restrict__ y_l, const float* __restrict__ up_g,
                                                                 const float* __restrict__ gp, const int* __restrict__ amax,
                                                                 const float* __restrict__ bconst, const float* __restrict__ units,
                                                                 const float* __restrict__ y_prev, const float* __restrict__ aff_prev,
                                                                 const float* __restrict__ w0, float* __restrict__ dA_prev,
                                                                 float* __restrict__ sums, float* __restrict__ gmax_prev, int B,
                                                                 int N, int ex_rows, float* __restrict__ a2rows) {
  static_assert(!TOP || CIN == COUT, "the top layer contracts over its own input channels");
  constexpr int KS = COUT / 32, MT = CIN / 16, TM = 16 / KS, NCH = MT / TM, NS = BOTTOM ? 5 : (TOP ? 3 : 2);
  static_assert(16 % KS == 0 && MT % TM == 0, "whole chunks");
  __shared__ __attribute__((aligned(16))) float lds[2][kChunk];
  __shared__ __attribute__((aligned(16))) float bc[3 * COUT];
  __shared__ __attribute__((aligned(16))) float ap[4 * CIN];
  __shared__ __attribute__((aligned(16))) float w0s[BOTTOM ? CIN * 4 : 4];
  __shared__ float wsum[NS][CIN];
  __shared__ float wmax[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, n_wave0 = blockIdx.x * (128 * NB) + wave * 16 * NB;

  auto stage = [&](int buf, int g) {
#pragma unroll
    for (int i = 0; i < kChunk / 256 / 8; ++i) {
      const int piece = wave + 8 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(units + (size_t)g * kChunk + piece * 256 + lane * 4),
                                       (lds_void*)&lds[buf][piece * 256], 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int t = tid; t < (TOP ? CIN : 3 * COUT); t += kThreads) bc[t] = bconst[t];
  for (int t = tid; t < 4 * CIN; t += kThreads) ap[t] = aff_prev[t];
  if (BOTTOM)
    for (int t = tid; t < CIN; t += kThreads) {
      w0s[4 * t] = w0[3 * t]; w0s[4 * t + 1] = w0[3 * t + 1]; w0s[4 * t + 2] = w0[3 * t + 2]; w0s[4 * t + 3] = 0.f;
    }
  for (int t = tid; t < NS * CIN; t += kThreads) (&wsum[0][0])[t] = 0.f;
  const float up = TOP ? 1.0f : bconst[3 * COUT], down = TOP ? bconst[CIN] : bconst[3 * COUT + 1];

  int n[NB], slot[NB];
  bool valid[NB];
  size_t toff_l[NB], toff_p[NB];             // the point's base offsets in the tiled arrays of COUT (y_l, up_g) and CIN (y_prev, dA_prev) channels
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    n[nb] = n_wave0 + 16 * nb + i16;
    valid[nb] = n[nb] < N;
    toff_l[nb] = tix_point(b, COUT, n[nb], N, B);
    toff_p[nb] = tix_point(b, CIN, n[nb], N, B);
    slot[nb] = TOP ? amax[(size_t)b * N + min(n[nb], N - 1)] : -1;             // TOP: `amax` carries slot_of [B][N]
    if (!valid[nb]) slot[nb] = -1;
  }
  __syncthreads();

  // prologue: the B fragments for the lane's k-slots -- dy_l (scaled by `up`), or (TOP) a_{l-1} itself
  f16x8 bhi[KS][NB], blo[KS][NB];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int c0 = 32 * ks + 16 * half + 4 * q;
      if (TOP) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&ap[c0]), t4 = *reinterpret_cast<const f32x4*>(&ap[CIN + c0]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x4 a;
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = fmaf(s4[r], y_prev[toff_p[nb] + (size_t)(c0 + r) * 32], t4[r]);
          split_into<true>(a, bhi[ks][nb], blo[ks][nb], half);
          // the activations of the arg-max points, point-major: what the arg-max part of dW_3 contracts with (gwtf_enc_train_dw3).
          // Here they are in registers; gathered from the (B, C, N) array afterwards every VALUE costs a 128-byte line (72 us).
          if (a2rows && slot[nb] >= 0) {
            f32x4 ar;
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[r] = fmaxf(a[r], 0.f);
            *reinterpret_cast<f32x4*>(&a2rows[((size_t)b * ex_rows + slot[nb]) * CIN + c0]) = ar;
          }
        }
        continue;
      }
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(&bc[c0]) * up, q4 = *reinterpret_cast<const f32x4*>(&bc[COUT + c0]) * up,
                  r4 = *reinterpret_cast<const f32x4*>(&bc[2 * COUT + c0]) * up;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t idx = toff_l[nb] + (size_t)(c0 + r) * 32;
#if GWTF_ENC_DBG == 3
          const float yv = 1.0f + (float)idx * 1e-9f, gm = 0.5f;
#else
          const float yv = y_l[idx], gm = up_g[idx];          // (padded tiles: unconditional loads; masked below)
#endif
          d[r] = valid[nb] ? fmaf(s4[r], gm, fmaf(q4[r], yv, r4[r])) : 0.f;
        }
        split_into<false>(d, bhi[ks][nb], blo[ks][nb], half);
      }
    }

  float px[NB], py[NB], pz[NB];
  if (BOTTOM) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int nc = min(n[nb], N - 1);
      px[nb] = y_prev[((size_t)b * 3 + 0) * N + nc];
      py[nb] = y_prev[((size_t)b * 3 + 1) * N + nc];
      pz[nb] = y_prev[((size_t)b * 3 + 2) * N + nc];
    }
  }

  int g = 0;
  float gmx = 0.f;
  // The epilogue of a row tile needs y_prev at (row, point) -- the mask a_{l-1} > 0 and the BatchNorm-backward sum -- and, for
  // the top layer, the arg-max row of the point.  Loaded where they are used, every tile waited for its own round trip (38 - 47 us
  // of a 160 us kernel: docs/LOG.md, ablation); they are fetched one tile AHEAD instead, behind the previous tile's product.
  float yv_cur[4][NB], yv_nxt[4][NB];
  f32x4 ex_cur[NB], ex_nxt[NB];
  auto fetch_tile = [&](int m, float (&yv)[4][NB], f32x4 (&ex)[NB]) {
    const int j0 = 16 * m + 4 * q;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      ex[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (TOP && slot[nb] >= 0) ex[nb] = *reinterpret_cast<const f32x4*>(&gp[((size_t)b * ex_rows + slot[nb]) * CIN + j0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#if GWTF_ENC_DBG == 2
        yv[r][nb] = 1.0f;
#else
        yv[r][nb] = BOTTOM ? 0.f : y_prev[toff_p[nb] + (size_t)(j0 + r) * 32];
#endif
      }
    }
  };
  fetch_tile(0, yv_cur, ex_cur);
#pragma unroll 1
  for (int ci = 0; ci < NCH; ++ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (g + 1 < NCH) stage((g + 1) & 1, g + 1);
    const float* L = lds[g & 1];
    ++g;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int m = ci * TM + t, j0 = 16 * m + 4 * q;
      if (m + 1 < MT) fetch_tile(m + 1, yv_nxt, ex_nxt);
      f32x4 acc[NB];
      tile_mfma<KS, NB>(L + t * KS * 512, lane, bhi, blo, acc);
      const f32x4 sp = *reinterpret_cast<const f32x4*>(&ap[j0]), tp = *reinterpret_cast<const f32x4*>(&ap[CIN + j0]),
                  mp = *reinterpret_cast<const f32x4*>(&ap[2 * CIN + j0]), rp = *reinterpret_cast<const f32x4*>(&ap[3 * CIN + j0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + r;
        float sb = 0.f, sg = 0.f, sx0 = 0.f, sx1 = 0.f, sx2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float yv;
          if (BOTTOM) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&w0s[4 * j]);
            yv = w[0] * px[nb] + (w[1] * py[nb] + w[2] * pz[nb]);
          } else {
            yv = valid[nb] ? yv_cur[r][nb] : 0.f;       // (a select: the padding of the last tile is uninitialised memory)
          }
          const float pre = fmaf(sp[r], yv, tp[r]);
          const bool on = valid[nb] && pre > 0.f;
          const float gm = on ? (TOP ? fmaf(acc[nb][r], down, bc[j] + ex_cur[nb][r]) : acc[nb][r] * down) : 0.f;
          if (TOP) sx0 += on ? pre : 0.f;                  // sum_p a_{l-1}: the R term of dW_3 (gwtf_enc_train_dw3)
#if GWTF_ENC_DBG == 1
          if (!BOTTOM && valid[nb] && gm == 123.456f) dA_prev[toff_p[nb] + (size_t)j * 32] = gm;
#else
          if (!BOTTOM) dA_prev[toff_p[nb] + (size_t)j * 32] = gm;  // (unconditional: zeros into the padding beyond N)
#endif
          sb += gm;
          sg = fmaf(gm, (yv - mp[r]) * rp[r], sg);
          if (BOTTOM) {
            sx0 = fmaf(gm, px[nb], sx0);
            sx1 = fmaf(gm, py[nb], sx1);
            sx2 = fmaf(gm, pz[nb], sx2);
          }
          gmx = fmaxf(gmx, fabsf(gm));
        }
        sb = row_sum16(sb);
        sg = row_sum16(sg);
        if (BOTTOM) {
          sx0 = row_sum16(sx0);
          sx1 = row_sum16(sx1);
          sx2 = row_sum16(sx2);
        }
        if (TOP) sx0 = row_sum16(sx0);
        if (i16 == 0) {
          atomicAdd(&wsum[0][j], sb);
          atomicAdd(&wsum[1][j], sg);
          if (TOP) atomicAdd(&wsum[2][j], sx0);
          if (BOTTOM) {
            atomicAdd(&wsum[2][j], sx0);
            atomicAdd(&wsum[3][j], sx1);
            atomicAdd(&wsum[4][j], sx2);
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        ex_cur[nb] = ex_nxt[nb];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv_cur[r][nb] = yv_nxt[r][nb];
      }
    }
  }
  gmx = wave_max(gmx);
  if (lane == 0) wmax[wave] = gmx;
  __syncthreads();
  float* dst = sums + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) % kR) * NS * CIN;
  for (int t = tid; t < NS * CIN; t += kThreads) atomicAdd(&dst[t], (&wsum[0][0])[t]);
  if (tid == 0 && gmax_prev) {
    float m = wmax[0];
    for (int w = 1; w < 8; ++w) m = fmaxf(m, wmax[w]);
    atomicMax(reinterpret_cast<int*>(gmax_prev), __builtin_bit_cast(int, m));
  }
}

// ---- weight gradient dW_l[c][k] = sum_p dy_l[c](p) a_{l-1}[k](p): a wavefront = one 64 x 64 block over one slice of points ----
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 x = {v[2 * p], v[2 * p + 1]};
    f16x2 h, l;
    split_pair(x, h, l);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

// eight consecutive points of an operand row.  UNCONDITIONAL 16-byte loads: written as `n0 < p1 ? load : 0` hipcc turned each of
// the eight floats into its own exec-masked branch around a global_load_dword (168 dword loads and 160 branches per k-step in the
// layer-2 kernel, which then ran at 2 TB/s whatever the prefetch depth, layout or barrier).  The tiled activation arrays are padded to
// whole 32-point tiles, so a read inside a tile is always in bounds; what lies beyond the slice is zeroed by the producer's select.
__device__ __forceinline__ void load8(const float* row, int n0, int p1, float (&v)[8]) {
#if GWTF_ENC_DBG == 4
  const float t0 = (float)(n0 & 15) * 0.01f + 0.5f;
  const float4 a = make_float4(t0, t0 + 0.1f, t0 - 0.2f, t0), c = make_float4(-t0, t0, 0.3f, t0);
#else
  const float4 a = *reinterpret_cast<const float4*>(row + n0);
  const float4 c = *reinterpret_cast<const float4*>(row + n0 + 4);
#endif
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
}
// the same for the input clouds x (B, 3, N), which are NOT padded: the address is clamped into the row instead (N % 4 == 0, N >= 8)
__device__ __forceinline__ void load8_clamped(const float* row, int n0, int N, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(row + min(n0, N - 4));
  const float4 c = *reinterpret_cast<const float4*>(row + min(n0 + 4, N - 4));
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
}

// One workgroup = WM x WN wavefronts = the WHOLE (64 WM) x (64 WN) matrix over one slice of the points; a wavefront owns a
// 64 x 64 block.  Per k-step of 32 points every thread produces a few operand fragments (a fragment lane = one channel x 8
// consecutive points: two 16-byte loads per source array, the per-channel transform, the f16 hi/lo split) into a
// double-buffered LDS image laid out in MFMA fragment order, so each element is loaded, transformed and split ONCE per
// workgroup and every consumer read is one conflict-free ds_read_b128.  One barrier per k-step.
// AMODE 0: A = dy_l = s gm + Q y + R from (y_l, up_g), scaled by `up`;  AMODE 2: A = a_l = relu(s y_l + t) and B IS A (the Gram
// matrix of layer 3's input; bconst = that layer's aff).  FIRST: B = a_0 = relu(W_0' x + t_0) from x and table0.
// producer item `it` of the weight-gradient kernel -> (operand row, point group).  Consecutive lanes own consecutive 16-byte slots of
// the LDS fragment image (slot = 64 (row >> 4) 2 + (row & 15) + 16 q): a wave's ds_write_b128 covers one contiguous KiB.  (The first
// mapping, row = it >> 2, q = it & 3, put the four lanes of a quad 256 bytes apart -- one bank: 6-10 conflict cycles per LDS instruction.)
#ifndef GWTF_ENC_DWMAP
#define GWTF_ENC_DWMAP 1
#endif
__device__ __forceinline__ int dw_item_row(int it) { return GWTF_ENC_DWMAP ? (((it >> 6) << 4) | (it & 15)) : (it >> 2); }
__device__ __forceinline__ int dw_item_q(int it) { return GWTF_ENC_DWMAP ? ((it >> 4) & 3) : (it & 3); }

template <int WM, int WN, int AMODE, bool FIRST>
__global__ __launch_bounds__(64 * WM * WN) void enc_train_dw_kernel(const float* __restrict__ y_l, const float* __restrict__ up_g,
                                                                    const float* __restrict__ bconst,
                                                                    const float* __restrict__ y_prev,
                                                                    const float* __restrict__ tab_prev, float* __restrict__ partials,
                                                                    int B, int N, int nsl, int per) {
  constexpr int CA = 64 * WM, CB = 64 * WN, NT = 64 * WM * WN;
  constexpr bool SAME = AMODE == 2;
  constexpr int TA = CA / 16, TB = SAME ? 0 : CB / 16;          // 16-row fragment tiles in the LDS image
  constexpr int IA = CA * 4 / NT, IB = SAME ? 0 : CB * 4 / NT;   // producer items (row, q) per thread and k-step
  static_assert(CA * 4 % NT == 0 && (SAME || CB * 4 % NT == 0), "whole items per thread");
  __shared__ uint4 img[2][(TA + TB) * 2 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int sl = blockIdx.x, b = sl / nsl, part = sl % nsl;
  const int p0 = part * per, p1 = min(N, p0 + per);
  const float up = SAME ? 1.0f : bconst[3 * CA], down = SAME ? 1.0f : bconst[3 * CA + 1];

  // producer constants: item i of this thread is row (tid + NT i) >> 2, point group (tid + NT i) & 3
  float cA0[IA], cA1[IA], cA2[IA], cB0[IB > 0 ? IB : 1], cB1[IB > 0 ? IB : 1], cB2[IB > 0 ? IB : 1], cB3[IB > 0 ? IB : 1];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int row = dw_item_row(tid + NT * i);
    cA0[i] = bconst[row] * up;
    cA1[i] = bconst[CA + row] * up;
    cA2[i] = SAME ? 0.f : bconst[2 * CA + row] * up;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int row = dw_item_row(tid + NT * i);
    if (FIRST) {
      cB0[i] = tab_prev[4 * row]; cB1[i] = tab_prev[4 * row + 1]; cB2[i] = tab_prev[4 * row + 2]; cB3[i] = tab_prev[4 * row + 3];
    } else {
      cB0[i] = tab_prev[row]; cB1[i] = tab_prev[CB + row]; cB2[i] = cB3[i] = 0.f;
    }
  }

  // DEEP: two register sets -- the operands of the step after next are in flight while this step's products run.  Only together with
  // the LDS-only barrier below (a __syncthreads() waits for every outstanding global load: the second set would never be in flight
  // across it).  Where the registers allow it: layer 2 (222 VGPRs at two waves per SIMD); the Gram kernel sits at its 128-register cap,
  // layer 1 at 256.
  constexpr bool DEEP = !FIRST && !SAME;
  constexpr int SETS = DEEP ? 2 : 1;
  float rA[SETS][IA][8], rG[SETS][SAME ? 1 : IA][8], rB[SETS][IB > 0 ? IB : 1][FIRST ? 3 : 1][8];
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, DEEP ? 1 : 0>;
  auto fetch = [&](int p, auto SC) __attribute__((always_inline)) {
    constexpr int S = decltype(SC)::value;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int it = tid + NT * i, row = dw_item_row(it), n0 = p + 8 * dw_item_q(it);
      // (the pointer is biased by -n0: load8 adds it back; n0 is a multiple of 8 inside one 32-point tile)
      load8((SAME ? y_prev : y_l) + tix(b, CA, row, n0, N) - n0, n0, p1, rA[S][i]);
      if (!SAME) load8(up_g + tix(b, CA, row, n0, N) - n0, n0, p1, rG[S][i]);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int it = tid + NT * i, row = dw_item_row(it), n0 = p + 8 * dw_item_q(it);
      if (FIRST) {
#pragma unroll
        for (int d = 0; d < 3; ++d) load8_clamped(y_prev + ((size_t)b * 3 + d) * N, n0, N, rB[S][i][d]);
      } else {
        load8(y_prev + tix(b, CB, row, n0, N) - n0, n0, p1, rB[S][i][0]);
      }
    }
  };
  auto produce = [&](int buf, int p, auto SC) __attribute__((always_inline)) {
    constexpr int S = decltype(SC)::value;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int it = tid + NT * i, row = dw_item_row(it), qq = dw_item_q(it), n0 = p + 8 * qq;
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool on = n0 + (e & 4) < p1;
        if (SAME) d[e] = on ? fmaxf(fmaf(cA0[i], rA[S][i][e], cA1[i]), 0.f) : 0.f;
        else d[e] = on ? fmaf(cA0[i], rG[S][i][e], fmaf(cA1[i], rA[S][i][e], cA2[i])) : 0.f;
      }
      f16x8 hi, lo;
      split8(d, hi, lo);
      const int slot = ((row >> 4) * 2) * 64 + (row & 15) + 16 * qq;
      img[buf][slot] = __builtin_bit_cast(uint4, hi);
      img[buf][slot + 64] = __builtin_bit_cast(uint4, lo);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int it = tid + NT * i, row = dw_item_row(it), qq = dw_item_q(it), n0 = p + 8 * qq;
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool on = n0 + (e & 4) < p1;
        const float pre = FIRST ? cB0[i] * rB[S][i][0][e] + (cB1[i] * rB[S][i][FIRST ? 1 : 0][e] + (cB2[i] * rB[S][i][FIRST ? 2 : 0][e] + cB3[i]))
                                : fmaf(cB0[i], rB[S][i][0][e], cB1[i]);
        d[e] = on ? fmaxf(pre, 0.f) : 0.f;
      }
      f16x8 hi, lo;
      split8(d, hi, lo);
      const int slot = ((TA + (row >> 4)) * 2) * 64 + (row & 15) + 16 * qq;
      img[buf][slot] = __builtin_bit_cast(uint4, hi);
      img[buf][slot + 64] = __builtin_bit_cast(uint4, lo);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto products = [&](int buf) __attribute__((always_inline)) {
    f16x8 ahi[4], alo[4], bhi[4], blo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ahi[t] = __builtin_bit_cast(f16x8, img[buf][((4 * wm + t) * 2) * 64 + lane]);
      alo[t] = __builtin_bit_cast(f16x8, img[buf][((4 * wm + t) * 2 + 1) * 64 + lane]);
      const int tb = SAME ? 4 * wn + t : TA + 4 * wn + t;
      bhi[t] = __builtin_bit_cast(f16x8, img[buf][(tb * 2) * 64 + lane]);
      blo[t] = __builtin_bit_cast(f16x8, img[buf][(tb * 2 + 1) * 64 + lane]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[mt], bhi[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[mt], blo[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[mt], bhi[nt], acc[mt][nt], 0, 0, 0);
      }
  };
  const int nsteps = p0 < p1 ? (p1 - p0 + 31) / 32 : 0;
  auto point_of = [&](int st) { return p0 + 32 * st; };
  if (nsteps > 0) {
    fetch(point_of(0), S0{});
    produce(0, point_of(0), S0{});
  }
  if (DEEP && nsteps > 1) fetch(point_of(1), S1{});
  __syncthreads();
  int buf = 0;
  // one k-step: the image of step st is in img[buf]; the operands of step st + 1 are in flight in set SN (DEEP) or are fetched now;
  // DEEP fetches step st + 2 into the other set
  auto step = [&](int st, auto SN, auto SNN) __attribute__((always_inline)) {
    if (DEEP) {
      if (st + 2 < nsteps) fetch(point_of(st + 2), SNN);
    } else if (st + 1 < nsteps) {
      fetch(point_of(st + 1), SN);
    }
    products(buf);
    if (st + 1 < nsteps) produce(buf ^ 1, point_of(st + 1), SN);
    // An LDS-only barrier: __syncthreads() also waits for every outstanding GLOBAL load (its fence is s_waitcnt vmcnt(0)), i.e. for
    // the operands in flight for the coming steps.  Only the LDS image must be visible to the other waves here; the loads in flight
    // target this lane's own registers.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  };
  for (int st = 0; st < nsteps; st += 2) {
    step(st, S1{}, S0{});
    if (st + 1 < nsteps) step(st + 1, S0{}, S1{});
  }
  float* out = partials + (size_t)sl * CA * CB;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)(64 * wm + 16 * mt + 4 * q + r) * CB + 64 * wn + 16 * nt + i16] = acc[mt][nt][r] * down;
}

// S[c][k] = sum_b gp[b][c] a_2[k](b, amax[b][c]): the arg-max part of dW_3 (one workgroup per output channel c, a thread per k).
// a_2 at the arg-max points comes point-major from a2rows [B][C][CP] (row slot_of[b][amax[b][c]], written by the top layer's backward
// kernel from its registers): contiguous 1-KiB rows.  (Gathered from the (B, C, N) activations every value cost a 128-byte line:
// 500 MB of traffic, 72 us.)  The shapes with a non-zero gradient are compacted into LDS by the first wave, eight rows in flight per
// thread; ascending shapes: the same sum on every run.
__global__ __launch_bounds__(256) void enc_top_gather_kernel(const float* __restrict__ gp, const int* __restrict__ amax,
                                                             const int* __restrict__ slot_of, const float* __restrict__ a2rows,
                                                             float* __restrict__ S, int B, int N, int C, int CP) {
  __shared__ float s_g[64];
  __shared__ int s_row[64], s_n;
  const int c = blockIdx.x;
  for (int k0 = 0; k0 < CP; k0 += blockDim.x) {
    const int k = k0 + threadIdx.x, kc = min(k, CP - 1);
    float acc = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
      __syncthreads();
      if (threadIdx.x < 64) {
        const int b = b0 + threadIdx.x;
        const float g = b < B ? gp[(size_t)b * C + c] : 0.f;
        int row = -1;
        if (g != 0.f) row = slot_of[(size_t)b * N + min(max(amax[(size_t)b * C + c], 0), N - 1)];
        const bool on = g != 0.f && row >= 0;          // (a point with a gradient always has a row: enc_top_scatter_kernel)
        const unsigned long long live = __ballot(on);
        const int pos = __popcll(live & ((1ull << threadIdx.x) - 1ull));
        if (on) { s_g[pos] = g; s_row[pos] = b * C + row; }
        if (threadIdx.x == 0) s_n = __popcll(live);
      }
      __syncthreads();
      const int nb = s_n;
      for (int i0 = 0; i0 < nb; i0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = a2rows[(size_t)s_row[min(i0 + u, nb - 1)] * CP + kc];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i0 + u < nb) acc = fmaf(s_g[i0 + u], v[u], acc);
      }
    }
    if (k < CP) S[(size_t)c * CP + k] = acc;
  }
}

// out[e] = sum over the slices of partials[slice][e]: a workgroup = 32 elements x 8 groups of slices, eight loads in flight per thread,
// the groups combined in a fixed order (one thread per element walking 256 - 512 slices four at a time was 24 us of dependent loads)
__global__ __launch_bounds__(256) void enc_dw_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int slices,
                                                            int total) {
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, gq = threadIdx.x >> 5, e = blockIdx.x * 32 + el, ec = min(e, total - 1);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int p = gq;
  for (; p + 56 < slices; p += 64) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(p + 8 * u) * total + ec];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u & 3] += v[u];
  }
  for (; p < slices; p += 8) s[0] += partials[(size_t)p * total + ec];
  part[gq][el] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (gq == 0 && e < total) {
    float t = part[0][el];
#pragma unroll
    for (int u = 1; u < 8; ++u) t += part[u][el];
    out[e] = t;
  }
}

// points per wave / 16 of the backward kernels (layer 2, layer 1, the top layer's M form)
#ifndef GWTF_ENC_NB2
#define GWTF_ENC_NB2 1
#endif
#ifndef GWTF_ENC_NB1
#define GWTF_ENC_NB1 1
#endif
#ifndef GWTF_ENC_NBT
#define GWTF_ENC_NBT 2
#endif
constexpr int kNB2 = GWTF_ENC_NB2, kNB1 = GWTF_ENC_NB1, kNBT = GWTF_ENC_NBT;
// the same for the forward kernels (layers 1, 2, 3)
#ifndef GWTF_ENC_NF1
#define GWTF_ENC_NF1 2
#endif
#ifndef GWTF_ENC_NF2
#define GWTF_ENC_NF2 2
#endif
#ifndef GWTF_ENC_NF3
#define GWTF_ENC_NF3 2
#endif
constexpr int kNF1 = GWTF_ENC_NF1, kNF2 = GWTF_ENC_NF2, kNF3 = GWTF_ENC_NF3;
int dw_slices_per_shape(int layer) { return layer == 1 ? 8 : 4; }
int dw_per(int layer, int N) {
  const int nsl = dw_slices_per_shape(layer);
  return ((N + nsl - 1) / nsl + 31) / 32 * 32;
}

}  // namespace

extern "C" int gwtf_enc_train_supported(const int* widths, int n_widths) {
  if (!widths || n_widths != 5) return 0;
  for (int i = 0; i < 5; ++i)
    if (widths[i] != kC[i]) return 0;
  return 1;
}

// floats of one stored activation array of `channels` channels (tiles of 32 points: N padded to whole tiles)
extern "C" size_t gwtf_enc_train_act_floats(int B, int channels, int N) {
  if (B <= 0 || channels <= 0 || N <= 0) return 0;
  return ((size_t)B * ((N + 31) / 32) + 1) * channels * 32;      // + the spare tile of tix_point
}

extern "C" size_t gwtf_enc_train_units_floats(int layer) {
  if (layer < 1 || layer > 3) return 0;
  return (size_t)(kC[layer + 1] / 16) * (kC[layer] / 32) * 512;
}

extern "C" int gwtf_enc_train_pack(const float* W, float* units_fwd, float* units_bwd, int layer, void* stream) {
  if (!W || !units_fwd || !units_bwd || layer < 1 || layer > 3) return GWTF_E_BADARG;
  const int cin = kC[layer], cout = kC[layer + 1];
  const int total = (cout / 16) * (cin / 32) * 2 * 64;     // == (cin / 16) * (cout / 32) * 2 * 64
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(enc_train_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, W, units_fwd, cout, cin, 0);
  hipLaunchKernelGGL(enc_train_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, W, units_bwd, cin, cout, 1);
  return (int)hipGetLastError();
}

// all three layers' images from one launch (what the training step calls: six launches of gwtf_enc_train_pack otherwise)
extern "C" int gwtf_enc_train_pack_all(const float* W1, const float* W2, const float* W3, float* uf1, float* ub1, float* uf2, float* ub2,
                                       float* uf3, float* ub3, void* stream) {
  if (!W1 || !W2 || !W3 || !uf1 || !ub1 || !uf2 || !ub2 || !uf3 || !ub3) return GWTF_E_BADARG;
  PackJobs J;
  const float* Ws[3] = {W1, W2, W3};
  float* uf[3] = {uf1, uf2, uf3};
  float* ub[3] = {ub1, ub2, ub3};
  for (int l = 1; l <= 3; ++l) {
    const int cin = kC[l], cout = kC[l + 1];
    J.W[2 * (l - 1)] = Ws[l - 1]; J.units[2 * (l - 1)] = uf[l - 1]; J.rows[2 * (l - 1)] = cout; J.kdim[2 * (l - 1)] = cin; J.transposed[2 * (l - 1)] = 0;
    J.W[2 * l - 1] = Ws[l - 1]; J.units[2 * l - 1] = ub[l - 1]; J.rows[2 * l - 1] = cin; J.kdim[2 * l - 1] = cout; J.transposed[2 * l - 1] = 1;
  }
  hipLaunchKernelGGL(enc_train_pack_all_kernel, dim3(64, 6), dim3(256), 0, (hipStream_t)stream, J);
  return (int)hipGetLastError();
}

// a (rows x kdim) row-major matrix -> fragment images (rows % 16 == 0, kdim % 32 == 0): the top layer's M = W_3^T diag(Q) W_3
extern "C" int gwtf_enc_train_pack_matrix(const float* W, float* units, int rows, int kdim, void* stream) {
  if (!W || !units || rows <= 0 || kdim <= 0 || rows % 16 || kdim % 32) return GWTF_E_BADARG;
  const int total = (rows / 16) * (kdim / 32) * 2 * 64;
  hipLaunchKernelGGL(enc_train_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, units, rows, kdim, 0);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_xmoments(const float* x, float* mom, int B, int N, void* stream) {
  if (!x || !mom || B <= 0 || N <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(enc_xmom_kernel, dim3((N + 1023) / 1024, B), dim3(256), 0, (hipStream_t)stream, x, mom, B, N);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_fold0(const float* mom12, double n_total, const float* W0, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float* aff, float* table0,
                                    void* stream) {
  if (!mom12 || !W0 || !gamma || !beta || !aff || !table0 || n_total <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(enc_fold0_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mom12, n_total, W0, gamma, beta, running_mean,
                     running_var, momentum, aff, table0, kC[1]);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_fold(const float* sums, int layer, double n_total, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float* aff, const float* aff_prev,
                                   void* stream) {
  if (!sums || !gamma || !beta || !aff || !aff_prev || layer < 1 || layer > 3 || n_total <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(enc_fold_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, sums, kC[layer + 1], n_total, gamma, beta,
                     running_mean, running_var, momentum, aff, aff_prev);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_forward(int layer, const float* in, const float* in_tab, const float* units, float* y_out,
                                      float* sums, float* ymax, unsigned long long* kmax, unsigned long long* kmin,
                                      const float* gamma3, int B, int N, void* stream) {
  if (!in || !in_tab || !units || !sums || !ymax || B <= 0 || N <= 0 || layer < 1 || layer > 3) return GWTF_E_BADARG;
  if (layer < 3 ? !y_out : (!kmax || !kmin || !gamma3)) return GWTF_E_BADARG;
  const dim3 block(kThreads);
  auto grid = [&](int nb) { return dim3((N + 128 * nb - 1) / (128 * nb), B); };
  hipStream_t st = (hipStream_t)stream;
  if (layer == 1) hipLaunchKernelGGL((enc_train_fwd_kernel<64, 128, true, false, kNF1>), grid(kNF1), block, 0, st, in, in_tab, units, y_out, sums, ymax, kmax, kmin, B, N, gamma3);
  else if (layer == 2) hipLaunchKernelGGL((enc_train_fwd_kernel<128, 256, false, false, kNF2>), grid(kNF2), block, 0, st, in, in_tab, units, y_out, sums, ymax, kmax, kmin, B, N, gamma3);
  else hipLaunchKernelGGL((enc_train_fwd_kernel<256, 512, false, true, kNF3>), grid(kNF3), block, 0, st, in, in_tab, units, y_out, sums, ymax, kmax, kmin, B, N, gamma3);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_pool(const unsigned long long* kmax, const unsigned long long* kmin, const float* aff3, float* pooled,
                                   int* amax, float* ystar, int B, int N, void* stream) {
  if (!kmax || !kmin || !aff3 || !pooled || !amax || !ystar || B <= 0 || N <= 0) return GWTF_E_BADARG;
  const int C = kC[4];
  hipLaunchKernelGGL(enc_pool_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, kmax, kmin, aff3, pooled,
                     amax, ystar, B, C, N);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_top_scatter(const float* coef, const float* scale, const int* amax, const float* W3, float* extra,
                                          int* slot_of, int* tables, int B, int N, void* stream) {
  if (!coef || !amax || !W3 || !extra || !slot_of || !tables || B <= 0 || N <= 0 || (size_t)N * sizeof(int) > 48 * 1024)
    return GWTF_E_BADARG;
  const int C = kC[4];
  int* row_off = tables;                          // [B][C + 2]
  int* row_list = tables + (size_t)B * (C + 2);   // [B][C]
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(enc_top_scatter_kernel, dim3(B), dim3(256), (size_t)N * sizeof(int), st, coef, scale, amax, slot_of, row_off,
                     row_list,
                     N, C);
  hipLaunchKernelGGL(enc_top_rows_kernel, dim3((B * C + 3) / 4), dim3(256), 0, st, coef, scale, row_off, row_list, W3, extra, B, C,
                     kC[3]);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_top(const float* g_pooled, const float* pooled, const float* ystar, const float* aff3, float* gp,
                                  float* sums, float* gmax, int B, void* stream) {
  if (!g_pooled || !pooled || !ystar || !aff3 || !gp || !sums || !gmax || B <= 0) return GWTF_E_BADARG;
  const int C = kC[4];
  hipLaunchKernelGGL(enc_top_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, g_pooled, pooled, ystar, aff3, gp, sums,
                     gmax, B, C);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_bwd_consts(const float* sums, int layer, double n_total, const float* gamma, const float* aff,
                                         const float* gmax, const float* ymax, float* bconst, void* stream) {
  if (!sums || !gamma || !aff || !bconst || layer < 0 || layer > 3 || n_total <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(enc_bwd_consts_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, sums, kC[layer + 1], n_total, gamma, aff,
                     gmax, ymax, bconst);
  return (int)hipGetLastError();
}

extern "C" int gwtf_enc_train_backward(int layer, const float* y_l, const float* up_g, const float* bconst,
                                       const float* units_bwd, const float* y_prev, const float* aff_prev, const float* w0,
                                       float* dA_prev, float* sums, float* gmax_prev, int B, int N, void* stream) {
  if (!y_l || !up_g || !bconst || !units_bwd || !y_prev || !aff_prev || !sums || B <= 0 || N <= 0 || layer < 1 || layer > 2)
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(kThreads);
  if (layer == 2) {
    if (!dA_prev || !gmax_prev) return GWTF_E_BADARG;
    hipLaunchKernelGGL((enc_train_bwd_kernel<128, 256, kNB2, false, false>), dim3((N + 128 * kNB2 - 1) / (128 * kNB2), B), block, 0, st, y_l, up_g, nullptr, nullptr, bconst,
                       units_bwd, y_prev, aff_prev, w0, dA_prev, sums, gmax_prev, B, N, 0, nullptr);
  } else {
    if (!w0) return GWTF_E_BADARG;
    hipLaunchKernelGGL((enc_train_bwd_kernel<64, 128, kNB1, false, true>), dim3((N + 128 * kNB1 - 1) / (128 * kNB1), B), block, 0, st, y_l, up_g, nullptr, nullptr, bconst,
                       units_bwd, y_prev, aff_prev, w0, dA_prev, sums, gmax_prev, B, N, 0, nullptr);
  }
  return (int)hipGetLastError();
}

// layer 3 in the M form (see enc_train_bwd_kernel): units_m = fragment images of M * 2^k, mconst = v [256] | {2^-k},
// extra / slot_of from gwtf_enc_train_top_scatter; a2rows [B][512][256] (out): row slot_of[b][n] of shape b = a_2(b, :, n) for every
// arg-max point n (the other rows are not written) -- what gwtf_enc_train_dw3 contracts the top gradients with
extern "C" int gwtf_enc_train_backward_top(const float* y2, const float* aff2, const float* units_m, const float* mconst,
                                           const float* extra, const int* slot_of, float* dA2, float* sums, float* gmax2,
                                           float* a2rows, int B, int N, void* stream) {
  if (!y2 || !aff2 || !units_m || !mconst || !extra || !slot_of || !dA2 || !sums || !gmax2 || !a2rows || B <= 0 || N <= 0)
    return GWTF_E_BADARG;
  hipLaunchKernelGGL((enc_train_bwd_kernel<256, 256, kNBT, true, false>), dim3((N + 128 * kNBT - 1) / (128 * kNBT), B), dim3(kThreads), 0,
                     (hipStream_t)stream, nullptr, nullptr, extra, slot_of, mconst, units_m, y2, aff2, nullptr, dA2, sums, gmax2, B, N,
                     kC[4], a2rows);
  return (int)hipGetLastError();
}

extern "C" size_t gwtf_enc_train_dw_partial_floats(int layer, int B, int N) {
  if (layer < 1 || layer > 3 || B <= 0 || N <= 0) return 0;
  // layer 3 goes through the Gram matrix of a_2 (C[3] x C[3]); see gwtf_enc_train_dw3
  const int CA = layer == 3 ? kC[3] : kC[layer + 1];
  return (size_t)B * dw_slices_per_shape(layer) * CA * kC[layer];
}

extern "C" int gwtf_enc_train_dw(int layer, const float* y_l, const float* up_g, const float* bconst, const float* y_prev,
                                 const float* tab_prev, float* partials, float* dW, int B, int N, void* stream) {
  if (!y_l || !up_g || !bconst || !y_prev || !tab_prev || !partials || !dW || B <= 0 || N <= 0 || (N & 3) || layer < 1 || layer > 2)
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int CA = kC[layer + 1], CB = kC[layer], nsl = dw_slices_per_shape(layer), per = dw_per(layer, N);
  if (layer == 2)
    hipLaunchKernelGGL((enc_train_dw_kernel<4, 2, 0, false>), dim3(B * nsl), dim3(512), 0, st, y_l, up_g, bconst, y_prev, tab_prev, partials, B, N, nsl, per);
  else
    hipLaunchKernelGGL((enc_train_dw_kernel<2, 1, 0, true>), dim3(B * nsl), dim3(128), 0, st, y_l, up_g, bconst, y_prev, tab_prev, partials, B, N, nsl, per);
  const int total = CA * CB;
  hipLaunchKernelGGL(enc_dw_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, st, partials, dW, B * nsl, total);
  return (int)hipGetLastError();
}

// The top layer's weight gradient without a contraction over its 512 output channels: with dy_3 = s gm_3 + Q y_3 + R and
// y_3 = W_3 a_2,   dW_3 = s (.) S + Q (.) (W_3 G_2) + R (x) sum_p a_2,   S[c][:] = sum_b gp[b][c] a_2(b, amax[b][c]) (gm_3 is
// non-zero at the arg-max points only), G_2 = sum_p a_2 a_2^T.  This call leaves G_2 (256 x 256) and S (512 x 256); the caller
// finishes with one small library GEMM (W_3 G_2) -- half the matrix work of the direct product and no pass over y_3.
extern "C" int gwtf_enc_train_dw3(const float* gp, const int* amax, const int* slot_of, const float* a2rows, const float* y2,
                                  const float* aff2, float* partials, float* gram, float* S, int B, int N, void* stream) {
  if (!gp || !amax || !slot_of || !a2rows || !y2 || !aff2 || !partials || !gram || !S || B <= 0 || N <= 0 || (N & 3)) return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int C2 = kC[3], nsl = dw_slices_per_shape(3), per = dw_per(3, N);
  hipLaunchKernelGGL((enc_train_dw_kernel<4, 4, 2, false>), dim3(B * nsl), dim3(1024), 0, st, nullptr, nullptr, aff2, y2, aff2, partials,
                     B, N, nsl, per);
  hipLaunchKernelGGL(enc_dw_reduce_kernel, dim3((C2 * C2 + 31) / 32), dim3(256), 0, st, partials, gram, B * nsl, C2 * C2);
  hipLaunchKernelGGL(enc_top_gather_kernel, dim3(kC[4]), dim3(256), 0, st, gp, amax, slot_of, a2rows, S, B, N, kC[4], C2);
  return (int)hipGetLastError();
}
