// gwtf_film.hip -- per-shape FiLM conditioning vectors for every coupling of the stack.
//
// For coupling c and branch X in {logvar, mu} the reference evaluates two small MLPs on the latent g
// (lib/networks/flows.py:33-45 / 68-80, used at :100-101,105-106):
//     w = Linear_f->f( Swish( BN( Linear_G->f(g) ) ) ),   b = same structure, other weights
//     h <- (eps + exp(w)) * BN1(sd1(.)) + b
// This kernel produces, per (shape, coupling, branch, feature j), the float4 the fused stack kernel's
// epilogue consumes: { a, a*c1 + b, W2[0][j], W2[1][j] } with a = eps + exp(w) and c1 the (eval-mode)
// sd1_bn shift, plus the sd2 biases.  It is B rows of work per head -- latency, not throughput --
// so it is a plain VALU kernel: one workgroup per (coupling, branch), lane = output feature,
// each wave carries R shapes so every weight load is reused R times; g is staged through LDS.
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "../../include/gwtf.h"

namespace {

constexpr int kWaves = 4;
constexpr int kR = 8;               // shapes per wave per pass
constexpr int kBT = kWaves * kR;    // shapes per pass
constexpr int kMaxTrainB = 128;     // train mode keeps all B rows of both heads in LDS

template <bool TRAIN>
__global__ __launch_bounds__(256) void film_kernel(const float* __restrict__ g, const float* __restrict__ pf,
                                                   float* __restrict__ out, float* __restrict__ stats, int B, int G, int C,
                                                   int f, int FP, float eps) {
  extern __shared__ __align__(16) float smem[];
  const int c = blockIdx.x, br = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const GwtfPackF P(FP, G);
  const float* w = pf + ((size_t)c * 2 + br) * P.branch_size();
  const size_t FS = gwtf_film_out_size(FP);
  const bool act = lane < FP;
  const int hrows = TRAIN ? ((B + kBT - 1) / kBT) * kBT : kBT;  // rows >= B hold zeros
  float* gt = smem;                         // [kBT][G]
  float* hb = smem + (size_t)kBT * G;       // [hrows][2][FP]
  float* st = hb + (size_t)hrows * 2 * FP;  // [2][2][FP] effective scale / shift

  // effective BatchNorm scale/shift (eval: folded by the packer)
  if (!TRAIN) {
    for (int t = threadIdx.x; t < 2 * FP; t += blockDim.x) {
      const int which = t / FP, j = t % FP;
      st[which * 2 * FP + j] = w[P.s(which) + j];
      st[which * 2 * FP + FP + j] = w[P.t(which) + j];
    }
  }

  auto layer2 = [&](int b0, int rows_base) {
    // second Linear (f->f) + exp + fold, for the kR shapes of this wave starting at b0 + wave*kR
    float acc[2][kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      acc[0][r] = act ? w[P.l1b(0) + lane] : 0.f;
      acc[1][r] = act ? w[P.l1b(1) + lane] : 0.f;
    }
    for (int i = 0; i < FP; ++i) {
      const float w0 = act ? w[P.l1t(0) + (size_t)i * FP + lane] : 0.f;
      const float w1 = act ? w[P.l1t(1) + (size_t)i * FP + lane] : 0.f;
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int row = rows_base + wave * kR + r;
        acc[0][r] = fmaf(hb[((size_t)row * 2 + 0) * FP + i], w0, acc[0][r]);
        acc[1][r] = fmaf(hb[((size_t)row * 2 + 1) * FP + i], w1, acc[1][r]);
      }
    }
    if (act) {
      const float c1 = w[P.c1() + lane], w20 = w[P.w2() + lane], w21 = w[P.w2() + FP + lane];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int b = b0 + wave * kR + r;
        if (b < B) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (lane < f) {
            const float a = eps + expf(acc[0][r]);
            v = make_float4(a, fmaf(a, c1, acc[1][r]), w20, w21);
          }
          *reinterpret_cast<float4*>(out + ((size_t)b * C + c) * FS + (size_t)br * 4 * FP + 4 * lane) = v;
          if (lane < 2) out[((size_t)b * C + c) * FS + 8 * FP + 2 * br + lane] = w[P.b2() + lane];
        }
      }
    }
  };

  for (int b0 = 0; b0 < B; b0 += kBT) {
    __syncthreads();
    for (int t = threadIdx.x; t < kBT * G; t += blockDim.x) {
      const int r = t / G, i = t - r * G;
      gt[t] = (b0 + r < B) ? g[(size_t)(b0 + r) * G + i] : 0.f;
    }
    __syncthreads();
    // first Linear (G->f), both heads
    float acc[2][kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) acc[0][r] = acc[1][r] = 0.f;
    for (int i = 0; i < G; ++i) {
      const float w0 = act ? w[P.l0t(0) + (size_t)i * FP + lane] : 0.f;
      const float w1 = act ? w[P.l0t(1) + (size_t)i * FP + lane] : 0.f;
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const float gv = gt[(wave * kR + r) * G + i];
        acc[0][r] = fmaf(gv, w0, acc[0][r]);
        acc[1][r] = fmaf(gv, w1, acc[1][r]);
      }
    }
    const int rows_base = TRAIN ? b0 : 0;
    if (act) {
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int row = rows_base + wave * kR + r;
        if (TRAIN) {
          hb[((size_t)row * 2 + 0) * FP + lane] = acc[0][r];
          hb[((size_t)row * 2 + 1) * FP + lane] = acc[1][r];
        } else {
#pragma unroll
          for (int which = 0; which < 2; ++which) {
            const float h = fmaf(acc[which][r], st[which * 2 * FP + lane], st[which * 2 * FP + FP + lane]);
            hb[((size_t)row * 2 + which) * FP + lane] = h / (1.0f + expf(-h));
          }
        }
      }
    }
    if (!TRAIN) {
      __syncthreads();
      layer2(b0, 0);
    }
  }

  if (TRAIN) {
    __syncthreads();
    // batch statistics over the B rows (biased variance, two-pass), per head and feature
    for (int t = threadIdx.x; t < 2 * FP; t += blockDim.x) {
      const int which = t / FP, j = t % FP;
      float mean = 0.f, var = 0.f;
      for (int b = 0; b < B; ++b) mean += hb[((size_t)b * 2 + which) * FP + j];
      mean /= (float)B;
      for (int b = 0; b < B; ++b) {
        const float d = hb[((size_t)b * 2 + which) * FP + j] - mean;
        var = fmaf(d, d, var);
      }
      var /= (float)B;
      const float s = w[P.s(which) + j] / sqrtf(var + GWTF_BN_EPS);
      st[which * 2 * FP + j] = s;
      st[which * 2 * FP + FP + j] = w[P.t(which) + j] - mean * s;
      if (stats && j < f) {
        float* so = stats + ((((size_t)c * 2 + br) * 2 + which) * 2) * f;
        so[j] = mean;
        so[f + j] = var;
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < B * 2 * FP; t += blockDim.x) {
      const int j = t % FP, which = (t / FP) & 1;
      const float h = fmaf(hb[t], st[which * 2 * FP + j], st[which * 2 * FP + FP + j]);
      hb[t] = h / (1.0f + expf(-h));
    }
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += kBT) layer2(b0, b0);
  }
}

}  // namespace

extern "C" int gwtf_film_forward(const float* g, const float* packed_film, float* film_out, float* bn_stats_out, int B,
                                 int G, int C, int f, float eps, int training, void* stream) {
  if (B <= 0 || G <= 0 || C <= 0 || f <= 0 || f > GWTF_MAX_FP || !g || !packed_film || !film_out) return GWTF_E_BADARG;
  if (training && B > kMaxTrainB) return GWTF_E_BADARG;
  const int FP = gwtf_padded_width(f);
  const int hrows = training ? ((B + kBT - 1) / kBT) * kBT : kBT;
  const size_t smem = ((size_t)kBT * G + (size_t)hrows * 2 * FP + 4 * (size_t)FP) * sizeof(float);
  if (smem > 160 * 1024) return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (training) {
    e = hipFuncSetAttribute((const void*)film_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(film_kernel<true>, dim3(C, 2), dim3(256), smem, st, g, packed_film, film_out, bn_stats_out, B, G, C,
                       f, FP, eps);
  } else {
    e = hipFuncSetAttribute((const void*)film_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(film_kernel<false>, dim3(C, 2), dim3(256), smem, st, g, packed_film, film_out, bn_stats_out, B, G, C,
                       f, FP, eps);
  }
  return (int)hipGetLastError();
}
