// gwtf_film.hip -- per-shape FiLM conditioning vectors for every coupling of the stack.
//
// For coupling c and branch X in {logvar, mu} the reference evaluates two small MLPs on the latent g
// (lib/networks/flows.py:33-45 / 68-80, used at :100-101,105-106):
//     w = Linear_f->f( Swish( BN( Linear_G->f(g) ) ) ),   b = same structure, other weights
//     h <- (eps + exp(w)) * BN1(sd1(.)) + b
// This kernel produces, per (shape, coupling, branch, feature j), what the fused stack kernel consumes:
// c = c1 + b/a (start value of the sd1 accumulators), W2[0][j]*a, W2[1][j]*a, with a = eps + exp(w) > 0 and
// c1 the (eval-mode) sd1_bn shift -- relu(a*(y+c1)+b) = a*relu(y+c) -- plus the sd2 biases.  It is B rows of work per head -- latency, not throughput --
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
          float cv = 0.f, u0 = 0.f, u1 = 0.f;
          if (lane < f) {
            const float a = eps + expf(acc[0][r]);
            cv = c1 + acc[1][r] / a;
            u0 = w20 * a;
            u1 = w21 * a;
          }
          float* ob = out + ((size_t)b * C + c) * FS + (size_t)br * 3 * FP + lane;
          ob[0] = cv;
          ob[FP] = u0;
          ob[2 * FP] = u1;
          if (lane < 2) out[((size_t)b * C + c) * FS + 6 * FP + 2 * br + lane] = w[P.b2() + lane];
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


// ---------------------------------------------------------------------------------------------
// Eval-mode fast path.  No cross-shape dependency (BatchNorm is folded), so the grid also tiles B:
// one workgroup per (coupling, branch, tile of kBTe shapes).  Weights and the g tile are staged
// through LDS in G-chunks with coalesced float4 loads (the first version read them straight from
// L2 inside the dot-product loop and was latency-bound: 89 us for 66 workgroups); lane = output
// feature, each wave carries kRe shapes so a weight read from LDS feeds 2*kRe FMAs.
constexpr int kBTe = 16;            // shapes per workgroup
constexpr int kRe = kBTe / kWaves;  // shapes per wave
constexpr int kGC = 64;             // rows of L0T per LDS chunk

__global__ __launch_bounds__(256) void film_eval_kernel(const float* __restrict__ g, const float* __restrict__ pf,
                                                        float* __restrict__ out, int B, int G, int C, int f, int FP,
                                                        float eps) {
  extern __shared__ __align__(16) float smem[];
  const int c = blockIdx.x, br = blockIdx.y, b0 = blockIdx.z * kBTe;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const GwtfPackF P(FP, G);
  const float* w = pf + ((size_t)c * 2 + br) * P.branch_size();
  const size_t FS = gwtf_film_out_size(FP);
  const bool act = lane < FP;
  float* wl = smem;                                // [2][kGC][FP] weight chunk (also reused for L1T: [2][FP][FP])
  float* gt = wl + 2 * (size_t)(kGC > FP ? kGC : FP) * FP;  // [kBTe][kGC] latent chunk
  float* hb = gt + (size_t)kBTe * kGC;             // [kBTe][2][FP] hidden activations

  float acc[2][kRe];
#pragma unroll
  for (int r = 0; r < kRe; ++r) acc[0][r] = acc[1][r] = 0.f;
  for (int i0 = 0; i0 < G; i0 += kGC) {
    const int rows = min(kGC, G - i0);
    __syncthreads();
    // stage L0T rows [i0, i0+rows) of both heads: contiguous rows*FP floats each
    for (int which = 0; which < 2; ++which) {
      const float4* src = reinterpret_cast<const float4*>(w + P.l0t(which) + (size_t)i0 * FP);
      float4* dst = reinterpret_cast<float4*>(wl + (size_t)which * kGC * FP);
      for (int t = threadIdx.x; t < rows * FP / 4; t += blockDim.x) dst[t] = src[t];
    }
    for (int t = threadIdx.x; t < kBTe * kGC; t += blockDim.x) {
      const int r = t / kGC, i = t - r * kGC;
      gt[t] = (b0 + r < B && i < rows) ? g[(size_t)(b0 + r) * G + i0 + i] : 0.f;
    }
    __syncthreads();
    if (act) {
      for (int i = 0; i < rows; ++i) {
        const float w0 = wl[(size_t)i * FP + lane], w1 = wl[(size_t)(kGC + i) * FP + lane];
#pragma unroll
        for (int r = 0; r < kRe; ++r) {
          const float gv = gt[(wave * kRe + r) * kGC + i];
          acc[0][r] = fmaf(gv, w0, acc[0][r]);
          acc[1][r] = fmaf(gv, w1, acc[1][r]);
        }
      }
    }
  }
  __syncthreads();
  // BatchNorm (folded) + Swish -> hb ; stage L1T of both heads into wl
  if (act) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float s = w[P.s(which) + lane], t = w[P.t(which) + lane];
#pragma unroll
      for (int r = 0; r < kRe; ++r) {
        const float h = fmaf(acc[which][r], s, t);
        hb[((size_t)(wave * kRe + r) * 2 + which) * FP + lane] = h / (1.0f + expf(-h));
      }
    }
  }
  for (int which = 0; which < 2; ++which) {
    const float4* src = reinterpret_cast<const float4*>(w + P.l1t(which));
    float4* dst = reinterpret_cast<float4*>(wl + (size_t)which * FP * FP);
    for (int t = threadIdx.x; t < FP * FP / 4; t += blockDim.x) dst[t] = src[t];
  }
  __syncthreads();
  if (!act) return;
  float o[2][kRe];
#pragma unroll
  for (int r = 0; r < kRe; ++r) {
    o[0][r] = w[P.l1b(0) + lane];
    o[1][r] = w[P.l1b(1) + lane];
  }
  for (int i = 0; i < FP; ++i) {
    const float w0 = wl[(size_t)i * FP + lane], w1 = wl[(size_t)(FP + i) * FP + lane];
#pragma unroll
    for (int r = 0; r < kRe; ++r) {
      o[0][r] = fmaf(hb[((size_t)(wave * kRe + r) * 2 + 0) * FP + i], w0, o[0][r]);
      o[1][r] = fmaf(hb[((size_t)(wave * kRe + r) * 2 + 1) * FP + i], w1, o[1][r]);
    }
  }
  const float c1 = w[P.c1() + lane], w20 = w[P.w2() + lane], w21 = w[P.w2() + FP + lane];
#pragma unroll
  for (int r = 0; r < kRe; ++r) {
    const int b = b0 + wave * kRe + r;
    if (b < B) {
      float cv = 0.f, u0 = 0.f, u1 = 0.f;
      if (lane < f) {
        const float a = eps + expf(o[0][r]);
        cv = c1 + o[1][r] / a;
        u0 = w20 * a;
        u1 = w21 * a;
      }
      float* ob = out + ((size_t)b * C + c) * FS + (size_t)br * 3 * FP + lane;
      ob[0] = cv;
      ob[FP] = u0;
      ob[2 * FP] = u1;
      if (lane < 2) out[((size_t)b * C + c) * FS + 6 * FP + 2 * br + lane] = w[P.b2() + lane];
    }
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
    const int rowsmax = kGC > FP ? kGC : FP;
    const size_t smem_e = (2 * (size_t)rowsmax * FP + (size_t)kBTe * kGC + (size_t)kBTe * 2 * FP) * sizeof(float);
    hipLaunchKernelGGL(film_eval_kernel, dim3(C, 2, (B + kBTe - 1) / kBTe), dim3(256), smem_e, st, g, packed_film,
                       film_out, B, G, C, f, FP, eps);
  }
  return (int)hipGetLastError();
}
