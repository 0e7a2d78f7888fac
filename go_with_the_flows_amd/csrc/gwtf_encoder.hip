// gwtf_encoder.hip -- PointNet cloud encoder (eval BatchNorm) fused with the max-pool over points.
// Reference: PointNetCloudEncoder (lib/networks/encoders.py:9-28: SharedDot -> BatchNorm1d -> ReLU, 3 -> C0 -> C1 -> C2 ->
// C3) and the pooling its caller applies (lib/networks/models.py:127-128: torch.max(features, dim=2)[0]).
//
// One workgroup = 8 wavefronts = 256 points of one cloud; a wavefront owns 32 points (two 16-point column groups of the
// MFMA N axis) and carries them through ALL layers in registers: the C-layout output of one layer (lane (col, q) holds rows
// 4q..4q+3 of every 16-row tile) is, after ReLU and the f16 hi/lo split, directly the B operand of the next layer, because
// the weights are packed with the matching K permutation (k-slot (ks, q, e) <-> input feature 32ks + 16(e>>2) + 4q + (e&3)).
// Contractions run on v_mfma_f32_16x16x32_f16 with the three-product split (W_hi h_hi + W_hi h_lo + W_lo h_hi, fp32
// accumulate), as in the stack kernel (gwtf_device.h).  BatchNorm is folded into the weight rows and a bias (the bias is
// the accumulators' start value).  Weights (672 KiB for 64-128-256-512) stream L2 -> LDS by LDS-DMA in 32-KiB chunks
// shared by the 8 wavefronts, double-buffered against the MFMAs.  The last layer's output is never stored unless the
// caller asks for it: element-wise integer-max LDS atomics (every lane its own slot: conflict-free) combine the 8
// wavefronts, the 16 point-lanes of each feature are reduced once at the end, one global atomic per (workgroup, feature)
// -- values are >= 0 after ReLU, so integer max on the bit patterns is exact.
#include <hip/hip_runtime.h>
#include "gwtf_device.h"

using namespace gwtf_dev;

namespace {

constexpr int kChunk = 8192;        // floats per LDS chunk (32 KiB = 16 units of [hi|lo] 16x32 f16 fragments)
constexpr int kEncThreads = 512;
constexpr int kEncPoints = 256;

template <int C0, int C1, int C2, int C3>
struct Enc {
  static_assert(C0 % 32 == 0 && C1 % 32 == 0 && C2 % 32 == 0 && C3 % 16 == 0, "hidden widths: multiples of 32");
  static constexpr int KS1 = C0 / 32, KS2 = C1 / 32, KS3 = C2 / 32;
  static constexpr int MT1 = C1 / 16, MT2 = C2 / 16, MT3 = C3 / 16;
  static_assert(16 % KS1 == 0 && 16 % KS2 == 0 && 16 % KS3 == 0, "k-steps per tile must divide a chunk");
  static constexpr int TM1 = 16 / KS1, TM2 = 16 / KS2, TM3 = 16 / KS3;   // M-tiles per chunk
  static_assert(MT1 % TM1 == 0 && MT2 % TM2 == 0 && MT3 % TM3 == 0, "whole chunks per layer");
  static constexpr int NCH1 = MT1 / TM1, NCH2 = MT2 / TM2, NCH3 = MT3 / TM3;
  static constexpr int NCH = NCH1 + NCH2 + NCH3;
  static constexpr int NBIAS = C1 + C2 + C3;
  static constexpr int HEAD = (4 * C0 + NBIAS + 255) / 256 * 256;   // floats: layer-0 table, biases, pad
  static constexpr size_t PACKED = (size_t)HEAD + (size_t)NCH * kChunk;
};

// one workgroup per (layer unit (m, ks), part): 64 lanes x 8 f16
__global__ void enc_pack_layer_kernel(const float* __restrict__ W, const float* __restrict__ bn, float* __restrict__ chunks,
                                      float* __restrict__ bias, int Cout, int Cin) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int KS = Cin / 32;
  const int total = (Cout / 16) * KS * 2 * 64;
  if (t < Cout) {
    const float s = bn[t] / sqrtf(bn[3 * Cout + t] + GWTF_BN_EPS);
    bias[t] = bn[Cout + t] - bn[2 * Cout + t] * s;
  }
  if (t >= total) return;
  const int lane = t & 63, part = (t >> 6) & 1, unit = t >> 7;     // unit = m * KS + ks
  const int m = unit / KS, ks = unit % KS;
  const int row = 16 * m + (lane & 15), q = lane >> 4;
  const float s = bn[row] / sqrtf(bn[3 * Cout + row] + GWTF_BN_EPS);
  _Float16 out[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 16 * (e >> 2) + 4 * q + (e & 3);
    const float v = W[(size_t)row * Cin + k] * s;
    const _Float16 hi = (_Float16)v;
    out[e] = part == 0 ? hi : (_Float16)(v - (float)hi);
  }
  float4* dst = reinterpret_cast<float4*>(chunks + (size_t)unit * 512 + part * 256 + lane * 4);
  *dst = *reinterpret_cast<const float4*>(out);
}

__global__ void enc_pack_l0_kernel(const float* __restrict__ W, const float* __restrict__ bn, float4* __restrict__ tab, int C0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C0) return;
  const float s = bn[c] / sqrtf(bn[3 * C0 + c] + GWTF_BN_EPS);
  tab[c] = make_float4(W[c * 3] * s, W[c * 3 + 1] * s, W[c * 3 + 2] * s, bn[C0 + c] - bn[2 * C0 + c] * s);
}

// ReLU + split of one accumulator tile into elements 4*(m&1)..+3 of the next layer's B fragment
__device__ __forceinline__ void relu_split_into(const f32x4& a, f16x8& hi, f16x8& lo, int half) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x2 v = {fmaxf(a[2 * p], 0.f), fmaxf(a[2 * p + 1], 0.f)};
    f16x2 h, l;
    split_pair(v, h, l);
    hi[4 * half + 2 * p] = h[0]; hi[4 * half + 2 * p + 1] = h[1];
    lo[4 * half + 2 * p] = l[0]; lo[4 * half + 2 * p + 1] = l[1];
  }
}

template <int KS, int NB>
__device__ __forceinline__ void tile_mfma(const float* __restrict__ unit0, int lane, const f16x8 (&bhi)[KS][NB],
                                          const f16x8 (&blo)[KS][NB], const f32x4& init, f32x4 (&acc)[NB]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f16x8 ahi = *reinterpret_cast<const f16x8*>(unit0 + ks * 512 + lane * 4);
    const f16x8 alo = *reinterpret_cast<const f16x8*>(unit0 + ks * 512 + 256 + lane * 4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi[ks][nb], ks == 0 ? init : acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo[ks][nb], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi[ks][nb], acc[nb], 0, 0, 0);
    }
  }
}

template <int C0, int C1, int C2, int C3>
__global__ __launch_bounds__(kEncThreads) void encoder_kernel(const float* __restrict__ x, const float* __restrict__ packed,
                                                              float* __restrict__ feat, float* __restrict__ pooled, int B,
                                                              int N) {
  using E = Enc<C0, C1, C2, C3>;
  constexpr int NB = 2;
  __shared__ __attribute__((aligned(16))) float lds[2][kChunk];
  __shared__ __attribute__((aligned(16))) float head[E::HEAD];
  __shared__ int wgmax[C3 * 16];   // [tile][row r][lane]: element-wise max over the 8 wavefronts, reduced over lanes at the end
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, n_wave0 = blockIdx.x * kEncPoints + wave * 32;
  const float* chunks = packed + E::HEAD;

  auto stage = [&](int buf, int g) {
#pragma unroll
    for (int i = 0; i < kChunk / 256 / 8; ++i) {
      const int piece = wave + 8 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(chunks + (size_t)g * kChunk + piece * 256 + lane * 4),
                                       (lds_void*)&lds[buf][piece * 256], 16, 0, 0);
    }
  };
  stage(0, 0);
  for (int t = tid; t < E::HEAD; t += kEncThreads) head[t] = packed[t];
  for (int t = tid; t < C3 * 16; t += kEncThreads) wgmax[t] = 0;

  float px[NB], py[NB], pz[NB];
  bool valid[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n_wave0 + 16 * nb + i16;
    valid[nb] = n < N;
    px[nb] = valid[nb] ? x[((size_t)b * 3 + 0) * N + n] : 0.f;
    py[nb] = valid[nb] ? x[((size_t)b * 3 + 1) * N + n] : 0.f;
    pz[nb] = valid[nb] ? x[((size_t)b * 3 + 2) * N + n] : 0.f;
  }
  const bool ragged = n_wave0 + 32 > N;   // wave-uniform
  __syncthreads();   // head[] visible

  // layer 0 (3 -> C0) on the VALU, straight into the B fragments of layer 1
  f16x8 h0hi[E::KS1][NB], h0lo[E::KS1][NB];
  {
    const float4* tab = reinterpret_cast<const float4*>(head);
#pragma unroll
    for (int m = 0; m < C0 / 16; ++m) {
      float4 w[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = tab[16 * m + 4 * q + r];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = w[r].x * px[nb] + (w[r].y * py[nb] + (w[r].z * pz[nb] + w[r].w));
        relu_split_into(a, h0hi[m >> 1][nb], h0lo[m >> 1][nb], m & 1);
      }
    }
  }
  const float* bias1 = head + 4 * C0;
  const float* bias2 = bias1 + C1;
  const float* bias3 = bias2 + C2;

  int g = 0;   // running chunk index over all layers
  auto next_chunk = [&]() -> const float* {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (g + 1 < E::NCH) stage((g + 1) & 1, g + 1);
    const float* L = lds[g & 1];
    ++g;
    return L;
  };

  // layer 1: C0 -> C1
  f16x8 h1hi[E::KS2][NB], h1lo[E::KS2][NB];
#pragma unroll
  for (int ci = 0; ci < E::NCH1; ++ci) {
    const float* L = next_chunk();
#pragma unroll
    for (int t = 0; t < E::TM1; ++t) {
      const int m = ci * E::TM1 + t;
      const f32x4 init = *reinterpret_cast<const f32x4*>(bias1 + 16 * m + 4 * q);
      f32x4 acc[NB];
      tile_mfma<E::KS1, NB>(L + t * E::KS1 * 512, lane, h0hi, h0lo, init, acc);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_split_into(acc[nb], h1hi[m >> 1][nb], h1lo[m >> 1][nb], m & 1);
    }
  }
  // layer 2: C1 -> C2
  f16x8 h2hi[E::KS3][NB], h2lo[E::KS3][NB];
#pragma unroll
  for (int ci = 0; ci < E::NCH2; ++ci) {
    const float* L = next_chunk();
#pragma unroll
    for (int t = 0; t < E::TM2; ++t) {
      const int m = ci * E::TM2 + t;
      const f32x4 init = *reinterpret_cast<const f32x4*>(bias2 + 16 * m + 4 * q);
      f32x4 acc[NB];
      tile_mfma<E::KS2, NB>(L + t * E::KS2 * 512, lane, h1hi, h1lo, init, acc);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) relu_split_into(acc[nb], h2hi[m >> 1][nb], h2lo[m >> 1][nb], m & 1);
    }
  }
  // layer 3: C2 -> C3, pooled (and optionally stored)
#pragma unroll 1
  for (int ci = 0; ci < E::NCH3; ++ci) {
    const float* L = next_chunk();
#pragma unroll
    for (int t = 0; t < E::TM3; ++t) {
      const int m = ci * E::TM3 + t;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bias3 + 16 * m + 4 * q);
      f32x4 acc[NB];
      tile_mfma<E::KS3, NB>(L + t * E::KS3 * 512, lane, h2hi, h2lo, bias, acc);
      if (feat) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (valid[nb]) {
            const int n = n_wave0 + 16 * nb + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) feat[((size_t)b * C3 + 16 * m + 4 * q + r) * N + n] = fmaxf(acc[nb][r], 0.f);
          }
      }
      if (pooled) {
        if (ragged) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nb][r] = valid[nb] ? acc[nb][r] : 0.f;
        }
        // ReLU'd values are >= 0: integer max on the bit patterns is exact.  One conflict-free LDS atomic per (tile, row):
        // every lane owns its slot, the 16 point-lanes of a row are combined once at the end of the kernel.
#pragma unroll
        for (int r = 0; r < 4; ++r)
          atomicMax(&wgmax[(m * 4 + r) * 64 + lane], __builtin_bit_cast(int, fmaxf(fmaxf(acc[0][r], acc[1][r]), 0.f)));
      }
    }
  }
  if (pooled) {
    __syncthreads();
    for (int ft = tid; ft < C3; ft += kEncThreads) {
      const int m = ft >> 4, qq = (ft >> 2) & 3, r = ft & 3;
      const int4* src = reinterpret_cast<const int4*>(&wgmax[(m * 4 + r) * 64 + 16 * qq]);
      int v = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int4 w = src[i];
        v = max(max(v, max(w.x, w.y)), max(w.z, w.w));
      }
      atomicMax(reinterpret_cast<int*>(pooled) + (size_t)b * C3 + ft, v);
    }
  }
}

template <int C0, int C1, int C2, int C3>
int launch_encoder(const float* x, const float* packed, float* feat, float* pooled, int B, int N, hipStream_t st) {
  if (pooled) {
    hipError_t e = hipMemsetAsync(pooled, 0, sizeof(float) * (size_t)B * C3, st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((encoder_kernel<C0, C1, C2, C3>), dim3((N + kEncPoints - 1) / kEncPoints, B), dim3(kEncThreads), 0, st, x,
                     packed, feat, pooled, B, N);
  return (int)hipGetLastError();
}

template <int C0, int C1, int C2, int C3>
int pack_encoder(const float* raw, float* packed, hipStream_t st) {
  using E = Enc<C0, C1, C2, C3>;
  const int cin[4] = {3, C0, C1, C2}, cout[4] = {C0, C1, C2, C3};
  hipError_t e = hipMemsetAsync(packed, 0, sizeof(float) * E::HEAD, st);
  if (e != hipSuccess) return (int)e;
  const float* rec = raw;
  float* bias = packed + 4 * C0;
  float* chunks = packed + E::HEAD;
  for (int l = 0; l < 4; ++l) {
    const float* W = rec;
    const float* bn = rec + (size_t)cout[l] * cin[l];
    if (l == 0) {
      hipLaunchKernelGGL(enc_pack_l0_kernel, dim3((C0 + 63) / 64), dim3(64), 0, st, W, bn, reinterpret_cast<float4*>(packed), C0);
    } else {
      const int total = (cout[l] / 16) * (cin[l] / 32) * 2 * 64;
      hipLaunchKernelGGL(enc_pack_layer_kernel, dim3((total + 255) / 256), dim3(256), 0, st, W, bn, chunks, bias, cout[l], cin[l]);
      chunks += (size_t)(cout[l] / 16) * (cin[l] / 32) * 512;
      bias += cout[l];
    }
    rec += (size_t)cout[l] * cin[l] + 4 * (size_t)cout[l];
  }
  return (int)hipGetLastError();
}

// the shapes built today: every shipped config uses 3 -> 64 -> 128 -> 256 -> 512 (configs/*.yaml pc_enc_*); the small
// one keeps the tests' oracle runs short
#define GWTF_ENC_SHAPES(X) X(64, 128, 256, 512) X(64, 128, 64, 128)

bool shape_is(const int* w, int n, int c0, int c1, int c2, int c3) {
  return n == 5 && w[0] == 3 && w[1] == c0 && w[2] == c1 && w[3] == c2 && w[4] == c3;
}

}  // namespace

extern "C" size_t gwtf_encoder_raw_floats(const int* widths, int n_widths) {
  if (!widths || n_widths < 2) return 0;
  size_t t = 0;
  for (int l = 1; l < n_widths; ++l) t += (size_t)widths[l] * widths[l - 1] + 4 * (size_t)widths[l];
  return t;
}

extern "C" size_t gwtf_encoder_packed_floats(const int* widths, int n_widths) {
  if (!widths) return 0;
#define X(a, b, c, d) \
  if (shape_is(widths, n_widths, a, b, c, d)) return Enc<a, b, c, d>::PACKED;
  GWTF_ENC_SHAPES(X)
#undef X
  return 0;
}

extern "C" int gwtf_encoder_pack(const float* raw, float* packed, const int* widths, int n_widths, void* stream) {
  if (!raw || !packed || !widths) return GWTF_E_BADARG;
#define X(a, b, c, d) \
  if (shape_is(widths, n_widths, a, b, c, d)) return pack_encoder<a, b, c, d>(raw, packed, (hipStream_t)stream);
  GWTF_ENC_SHAPES(X)
#undef X
  return GWTF_E_UNSUPPORTED;
}

extern "C" int gwtf_encoder_forward(const float* x, const float* packed, float* features, float* pooled, int B, int N,
                                    const int* widths, int n_widths, void* stream) {
  if (!x || !packed || !widths || (!features && !pooled) || B <= 0 || N <= 0) return GWTF_E_BADARG;
#define X(a, b, c, d) \
  if (shape_is(widths, n_widths, a, b, c, d)) return launch_encoder<a, b, c, d>(x, packed, features, pooled, B, N, (hipStream_t)stream);
  GWTF_ENC_SHAPES(X)
#undef X
  return GWTF_E_UNSUPPORTED;
}
