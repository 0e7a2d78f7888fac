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
          if (TRAIN) {
            // raw FiLM scale / shift: sd1_bn's batch statistics are not known yet (gwtf_train_fold1 combines them)
            float* ob = out + ((((size_t)b * C + c) * 2 + br) * 2) * FP + lane;
            ob[0] = lane < f ? eps + expf(acc[0][r]) : 1.f;
            ob[FP] = lane < f ? acc[1][r] : 0.f;
          } else {
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
// one workgroup per (coupling, branch, tile of 16 shapes).  Both Linear layers of both heads run on
// v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain) with the 16 shapes on M:
//     layer 1:  H[16 x FP]  = g_tile[16 x G] . L0T[G x FP]      per head, K = G
//     layer 2:  O[16 x FP]  = swish(bn(H))   . L1T[FP x FP]     per head, K = FP
// Wave w owns feature block w (16 output features) of BOTH heads, so the final a = eps + exp(w-head),
// c = c1 + b-head / a is computed in-lane from its two accumulators.  The contraction index is free to be
// renamed, so k-slot (step 4*kg + t, quarter q) is mapped to column 16*kg + 4*q + t: a lane's A operands
// of four consecutive MFMAs are four consecutive floats.  (History: a VALU version with operands read from
// L2 inside the dot-product loop took 89 us for 66 workgroups; VALU from LDS 24 us; MFMA with LDS-staged
// operands and a barrier per chunk 15-18 us -- every phase was a dependent load->sync->compute step.)
constexpr int kBTe = 16;   // shapes per workgroup = one MFMA M tile

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The whole kernel is one dependent chain (load -> 2 GEMMs -> exp), so it is written for latency: all
// operands of a K chunk are fetched straight from L2 into registers with independent, unpredicated loads
// issued back to back (no LDS staging, no barrier; L0T is zero-padded to a multiple of 16 rows by the packer
// and out-of-range latent columns / shapes are clamped, their products land on zero weights or discarded
// rows), the layer-2 weights are prefetched before layer 1 starts, and the only LDS traffic is the 16 x FP
// transpose of the hidden activations between the two layers.
// kGCH = latent columns per register-resident chunk: 128 keeps a G = 128 head in ONE load round (shortest chain; 150 VGPRs,
// 3 workgroups per CU), 64 takes two rounds but 4-5 workgroups fit a CU -- chosen when the grid exceeds one round of the
// former (the K x C couplings of a mixture: 19.8 -> 17.1 us on the airplane grid of 1056 workgroups).
template <int MB, int kGCH>
__global__ __launch_bounds__(MB > 4 ? 512 : 256) void film_eval_kernel(const float* __restrict__ g, const float* __restrict__ pf,
                                                        float* __restrict__ out, int B, int G, int C, int f, float eps) {
  constexpr int FP = 16 * MB;
  __shared__ __align__(16) float hb[2][16][FP + 4];   // [head][shape][feature]
  const int c = blockIdx.x, br = blockIdx.y, b0 = blockIdx.z * kBTe;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, i16 = lane & 15;
  const GwtfPackF P(FP, G);
  const int GP = P.GP();
  const float* w = pf + ((size_t)c * 2 + br) * P.branch_size();
  const size_t FS = gwtf_film_out_size(FP);
  const bool own = wave < MB;                        // this wave's feature block exists
  const int ft = 16 * wave + i16;                    // output feature owned by this lane (both heads)
  const int brow = b0 + i16;                         // shape whose latent row this lane feeds to the A operand
  const float* grow = g + (size_t)(brow < B ? brow : B - 1) * G;
  const float* l0 = w + P.l0t(0) + 4 * ft;           // L0Q[col/4][ft][col%4] (gwtf_layout.h)
  const float* l1 = w + P.l0t(1) + 4 * ft;
  const bool gvec = (G & 3) == 0;                    // latent rows are 16-byte aligned and whole quads

  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4 w1[2][MB];                                   // layer-2 B operands (prefetched)
  if (own) {
#pragma unroll
    for (int kg = 0; kg < MB; ++kg)
#pragma unroll
      for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int t = 0; t < 4; ++t) w1[which][kg][t] = w[P.l1t(which) + ((size_t)(4 * kg + q) * FP + ft) * 4 + t];   // L1Q: one dwordx4
    for (int i0 = 0; i0 < GP; i0 += kGCH) {
      f32x4 a4[kGCH / 16], b4[2][kGCH / 16];
#pragma unroll
      for (int kg = 0; kg < kGCH / 16; ++kg) {
        if (i0 + 16 * kg < GP) {                     // wave-uniform
          const int col = i0 + 16 * kg + 4 * q;      // k-slot (4*kg + t, q) <-> latent column col + t
          if (gvec) {
            a4[kg] = *reinterpret_cast<const f32x4*>(grow + (col < G ? col : G - 4));
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) a4[kg][t] = grow[col + t < G ? col + t : G - 1];
          }
          b4[0][kg] = *reinterpret_cast<const f32x4*>(l0 + (size_t)col * FP);   // (col/4) * FP * 4
          b4[1][kg] = *reinterpret_cast<const f32x4*>(l1 + (size_t)col * FP);
        }
      }
#pragma unroll
      for (int kg = 0; kg < kGCH / 16; ++kg) {
        if (i0 + 16 * kg < GP) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[kg][t], b4[0][kg][t], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[kg][t], b4[1][kg][t], acc[1], 0, 0, 0);
          }
        }
      }
    }
    // BatchNorm (folded) + Swish; hidden activations -> LDS for the layer-2 A operand
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float sc = w[P.s(which) + ft], sh = w[P.t(which) + ft];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float h = fmaf(acc[which][r], sc, sh);
        hb[which][4 * q + r][ft] = h / (1.0f + expf(-h));
      }
    }
  }
  __syncthreads();
  if (!own) return;
  f32x4 o[2];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const float bias = w[P.l1b(which) + ft];
    o[which] = f32x4{bias, bias, bias, bias};
  }
#pragma unroll
  for (int kg = 0; kg < MB; ++kg) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(&hb[0][i16][16 * kg + 4 * q]);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(&hb[1][i16][16 * kg + 4 * q]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], w1[0][kg][t], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], w1[1][kg][t], o[1], 0, 0, 0);
    }
  }
  const float c1 = w[P.c1() + ft], w20 = w[P.w2() + ft], w21 = w[P.w2() + FP + ft];
  // NaN / Inf must reach the outputs (the reference aborts on a non-finite loss, training.py:43-46) although the stack
  // kernel's ReLU is a v_max that returns 0 for a NaN accumulator: a non-finite head output (non-finite latent or FiLM
  // weight) turns the whole record entry into NaN -- u = NaN survives the ReLU as relu(acc) * NaN -- and POISON (NaN when
  // any weight of the coupling is non-finite, gwtf_pack.hip) is added to the sd2 biases.  Bit tests: -fno-honor-nans.
  const float qnan = __builtin_bit_cast(float, 0x7fc00000u);
  const float poison = w[P.poison()];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = b0 + 4 * q + r;
    if (b < B) {
      float cv = 0.f, u0 = 0.f, u1 = 0.f;
      if (ft < f) {
        const float a = eps + expf(o[0][r]);
        cv = c1 + o[1][r] / a;
        u0 = w20 * a;
        u1 = w21 * a;
        if (gwtf_nonfinite(o[0][r]) || gwtf_nonfinite(o[1][r])) cv = u0 = u1 = qnan;
      }
      float* ob = out + ((size_t)b * C + c) * FS + (size_t)br * 3 * FP + ft;
      ob[0] = cv;
      ob[FP] = u0;
      ob[2 * FP] = u1;
      if (ft < 2) out[((size_t)b * C + c) * FS + 6 * FP + 2 * br + ft] = w[P.b2() + ft] + poison;
    }
  }
}

}  // namespace

extern "C" int gwtf_film_forward(const float* g, const float* packed_film, float* film_out, float* bn_stats_out, int B,
                                 int G, int C, int f, float eps, int training, void* stream) {
  if (B <= 0 || G <= 0 || C <= 0 || f <= 0 || f > GWTF_MAX_FP || !g || !packed_film || !film_out) return GWTF_E_BADARG;
  if (training && (B > kMaxTrainB || f > 64)) return GWTF_E_BADARG;     // train kernel: lane = feature (f <= 64); wider stacks use the torch FiLM graph
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
    const dim3 grid(C, 2, (B + kBTe - 1) / kBTe);
    const bool many = (long)grid.x * grid.y * grid.z > 768;   // more than one round of 256 CUs x 3 resident workgroups
    // one wavefront per block of 16 output features: 4 wavefronts up to f = 64, 8 beyond (f <= 128)
#define GWTF_FILM_EVAL(MB_)                                                                                                  \
  if (many) hipLaunchKernelGGL((film_eval_kernel<MB_, 64>), grid, dim3(MB_ > 4 ? 512 : 256), 0, st, g, packed_film, film_out, B, G, C, f, eps); \
  else hipLaunchKernelGGL((film_eval_kernel<MB_, 128>), grid, dim3(MB_ > 4 ? 512 : 256), 0, st, g, packed_film, film_out, B, G, C, f, eps)
    switch (FP / 16) {
      case 1: GWTF_FILM_EVAL(1); break;
      case 2: GWTF_FILM_EVAL(2); break;
      case 3: GWTF_FILM_EVAL(3); break;
      case 4: GWTF_FILM_EVAL(4); break;
      case 5: GWTF_FILM_EVAL(5); break;
      case 6: GWTF_FILM_EVAL(6); break;
      case 7: GWTF_FILM_EVAL(7); break;
      default: GWTF_FILM_EVAL(8); break;
    }
#undef GWTF_FILM_EVAL
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Differentiable train-mode FiLM heads (autograd.py _film_train): between the two batched products of a head sits BatchNorm over
// the B latent ROWS (batch statistics, reference flows.py:33-38 / 68-73 nn.BatchNorm1d in train()) and a swish.  One kernel forward
// and one backward instead of ~30 elementwise / reduction launches:
//   x [B][M] (M = heads * f columns, contiguous)   gamma / beta: element (head = (c, x, h), j) at base + c*sc + x*sx + h*sh + j
//   forward : mean[M], var[M] (biased), rstd[M], y = swish(gamma (x - mean) rstd + beta)
//   backward: dx [B][M], dgamma[M], dbeta[M]
// Block = 64 columns x 4 row slices; the column sums go through LDS.  Two-pass variance (torch's var).
namespace {
constexpr int kBsCols = 64, kBsSlices = 4;

__device__ __forceinline__ size_t bs_param_offset(int col, int f, long sc, long sx, long sh) {
  const int j = col % f, head = col / f;
  return (size_t)(head >> 2) * sc + (size_t)((head >> 1) & 1) * sx + (size_t)(head & 1) * sh + j;
}

__device__ __forceinline__ float bs_col_sum(float v, float (*red)[kBsCols], int sl, int tc) {
  red[sl][tc] = v;
  __syncthreads();
  const float s = (red[0][tc] + red[1][tc]) + (red[2][tc] + red[3][tc]);
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(kBsCols * kBsSlices) void bn_swish_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                            const float* __restrict__ beta, long sc, long sx, long sh,
                                                                            int f, int B, int M, float* __restrict__ y,
                                                                            float* __restrict__ mean, float* __restrict__ var,
                                                                            float* __restrict__ rstd) {
  __shared__ float red[kBsSlices][kBsCols];
  const int tc = threadIdx.x % kBsCols, sl = threadIdx.x / kBsCols;
  const int col = blockIdx.x * kBsCols + tc;
  const bool on = col < M;
  float s = 0.f;
  if (on)
    for (int r = sl; r < B; r += kBsSlices) s += x[(size_t)r * M + col];
  const float mu = bs_col_sum(s, red, sl, tc) / (float)B;
  float q = 0.f;
  if (on)
    for (int r = sl; r < B; r += kBsSlices) { const float d = x[(size_t)r * M + col] - mu; q = fmaf(d, d, q); }
  const float v = bs_col_sum(q, red, sl, tc) / (float)B;
  if (!on) return;
  const float rs = 1.0f / sqrtf(v + GWTF_BN_EPS);
  const size_t po = bs_param_offset(col, f, sc, sx, sh);
  const float ga = gamma[po], be = beta[po];
  for (int r = sl; r < B; r += kBsSlices) {
    const float h = fmaf((x[(size_t)r * M + col] - mu) * rs, ga, be);
    y[(size_t)r * M + col] = h / (1.0f + expf(-h));
  }
  if (sl == 0) { mean[col] = mu; var[col] = v; rstd[col] = rs; }
}

__global__ __launch_bounds__(kBsCols * kBsSlices) void bn_swish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                            long sc, long sx, long sh, int f, int B, int M,
                                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                            float* __restrict__ gx, float* __restrict__ ggamma,
                                                                            float* __restrict__ gbeta) {
  __shared__ float red[kBsSlices][kBsCols];
  const int tc = threadIdx.x % kBsCols, sl = threadIdx.x / kBsCols;
  const int col = blockIdx.x * kBsCols + tc;
  const bool on = col < M;
  const size_t po = on ? bs_param_offset(col, f, sc, sx, sh) : 0;
  const float ga = on ? gamma[po] : 0.f, be = on ? beta[po] : 0.f, mu = on ? mean[col] : 0.f, rs = on ? rstd[col] : 0.f;
  auto dh_of = [&](int r, float& xh) {   // d loss / d h of row r (h = the BatchNorm output), and the normalised input
    xh = (x[(size_t)r * M + col] - mu) * rs;
    const float h = fmaf(xh, ga, be);
    const float sg = 1.0f / (1.0f + expf(-h));
    return gy[(size_t)r * M + col] * (sg * (1.0f + h * (1.0f - sg)));
  };
  float sb = 0.f, sg_ = 0.f;
  if (on)
    for (int r = sl; r < B; r += kBsSlices) {
      float xh;
      const float dh = dh_of(r, xh);
      sb += dh;
      sg_ = fmaf(dh, xh, sg_);
    }
  const float dbeta = bs_col_sum(sb, red, sl, tc);
  const float dgamma = bs_col_sum(sg_, red, sl, tc);
  if (!on) return;
  const float k = ga * rs, mb = dbeta / (float)B, mg = dgamma / (float)B;
  for (int r = sl; r < B; r += kBsSlices) {
    float xh;
    const float dh = dh_of(r, xh);
    gx[(size_t)r * M + col] = k * (dh - mb - xh * mg);
  }
  if (sl == 0) { ggamma[col] = dgamma; gbeta[col] = dbeta; }
}
}  // namespace

extern "C" int gwtf_film_bn_swish_forward(const float* x, const float* gamma, const float* beta, long stride_c, long stride_x,
                                          long stride_h, int f, int B, int M, float* y, float* mean, float* var, float* rstd,
                                          void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !var || !rstd || f <= 0 || B <= 0 || M <= 0 || M % (4 * f) != 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(bn_swish_fwd_kernel, dim3((M + kBsCols - 1) / kBsCols), dim3(kBsCols * kBsSlices), 0, (hipStream_t)stream, x,
                     gamma, beta, stride_c, stride_x, stride_h, f, B, M, y, mean, var, rstd);
  return (int)hipGetLastError();
}

extern "C" int gwtf_film_bn_swish_backward(const float* x, const float* gy, const float* gamma, const float* beta, long stride_c,
                                           long stride_x, long stride_h, int f, int B, int M, const float* mean, const float* rstd,
                                           float* gx, float* ggamma, float* gbeta, void* stream) {
  if (!x || !gy || !gamma || !beta || !mean || !rstd || !gx || !ggamma || !gbeta || f <= 0 || B <= 0 || M <= 0 || M % (4 * f) != 0)
    return GWTF_E_BADARG;
  hipLaunchKernelGGL(bn_swish_bwd_kernel, dim3((M + kBsCols - 1) / kBsCols), dim3(kBsCols * kBsSlices), 0, (hipStream_t)stream, x,
                     gy, gamma, beta, stride_c, stride_x, stride_h, f, B, M, mean, rstd, gx, ggamma, gbeta);
  return (int)hipGetLastError();
}
