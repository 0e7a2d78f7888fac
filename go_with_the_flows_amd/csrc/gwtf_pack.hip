// gwtf_pack.hip -- weight packer: raw arena -> (MFMA-fragment-ordered stack weights, FiLM weights).
// One thread per OUTPUT element, grid-stride; pure gather + BatchNorm folding, HBM/L2 bound and tiny
// (a few hundred KB).  Layouts: gwtf_layout.h.  Arithmetic being folded: reference
// lib/networks/flows.py:25-31 (sd0 -> BN -> ReLU -> sd1 -> BN(affine=False)) in eval mode.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "gwtf_layout.h"
#include "../../include/gwtf.h"

namespace {

__device__ __forceinline__ float inv_std(float var) { return 1.0f / sqrtf(var + GWTF_BN_EPS); }

__device__ __forceinline__ bool nonfinite(float x) { return gwtf_nonfinite(x); }   // opaque bit test, see gwtf_layout.h
__device__ __forceinline__ int floor_log2(float x) {   // x > 0, finite
  int e;
  frexpf(x, &e);
  return e - 1;
}

// un-halved abs-form columns of output row jo (gwtf_layout.h): sum_k W1'[jo][k] {wa_k, wb_k, c0_k}, W1' = sd1.weight * sc (the range
// exponents CS cancel between the image and the sd0 record), the folded sd0 as the sd0 section stores it
__device__ inline void abs_columns(const float* rb, const GwtfRaw& R, int f, int jo, int kept, float sc, double* Ca, double* Cb,
                                   double* Cc) {
  const float* bn = rb + R.bn0();
  double a = 0.0, b = 0.0, c = 0.0;
  for (int k = 0; k < f; ++k) {
    const float s = bn[k] * inv_std(bn[3 * f + k]);
    const double w = (double)(rb[R.sd1_w() + (size_t)jo * f + k] * sc);
    a += w * (double)(rb[R.sd0_w(k, 0, kept)] * s);
    if (kept > 1) b += w * (double)(rb[R.sd0_w(k, 1, kept)] * s);
    c += w * (double)(bn[f + k] - bn[2 * f + k] * s);
  }
  *Ca = a; *Cb = b; *Cc = c;
}

// Range scaling exponents + poison of every (coupling, branch) -> packed_film's RS / CS / POISON slots (gwtf_layout.h).
// One workgroup per branch record; runs before the two gather kernels below, which read the exponents.
__global__ __launch_bounds__(256) void pack_scales_kernel(const float* __restrict__ raw, float* __restrict__ pf, int C, int f,
                                                          int G, int FP, int training, int pattern0, int Cper) {
  const GwtfRaw R(f, G);
  const GwtfPackF P(FP, G);
  const int cb = blockIdx.x, c = cb >> 1, t = threadIdx.x;
  const float* rb = raw + (size_t)cb * R.branch_size();
  float* w = pf + (size_t)cb * P.branch_size();
  __shared__ int s_cs[GWTF_MAX_FP], s_bad;
  if (t == 0) s_bad = 0;
  __syncthreads();
  bool bad = false;
  for (size_t i = t; i < R.branch_size(); i += blockDim.x) bad |= nonfinite(rb[i]);
  if (bad) s_bad = 1;
  if (t < FP) {
    int cs = 0;
    if (!training && t < f) {
      const float* bn = rb + R.bn0();
      const float s = bn[t] * inv_std(bn[3 * f + t]);
      const int k = gwtf_pattern_kept((pattern0 + c % Cper) % 6);
      const float wa = rb[R.sd0_w(t, 0, k)] * s, wb = k > 1 ? rb[R.sd0_w(t, 1, k)] * s : 0.f, c0 = bn[f + t] - bn[2 * f + t] * s;
      const float T = fabsf(wa) + fabsf(wb) + fabsf(c0);
      if (T > 0.f && !nonfinite(T)) cs = min(40, max(-40, floor_log2(T)));
    }
    s_cs[t] = cs;
    w[P.cs() + t] = (float)cs;
  }
  __syncthreads();
  if (t < FP) {
    int rs = 0;
    if (!training && t < f) {
      const float sc = inv_std(rb[R.bn1() + f + t]);
      float m = 0.f;
      for (int j = 0; j < f; ++j) m = fmaxf(m, fabsf(ldexpf(rb[R.sd1_w() + (size_t)t * f + j] * sc, s_cs[j])));
      if (gwtf_abs_form(f)) {     // the row also holds the three abs-form columns (gwtf_layout.h): they share its scale
        double Ca, Cb, Cc;
        abs_columns(rb, R, f, t, gwtf_pattern_kept((pattern0 + c % Cper) % 6), sc, &Ca, &Cb, &Cc);
        m = fmaxf(m, fmaxf(fabsf((float)Ca), fmaxf(fabsf((float)Cb), fabsf((float)Cc))));
      }
      if (m > 0.f && !nonfinite(m)) rs = min(50, max(-50, floor_log2(m) - 12));
    }
    w[P.rs() + t] = (float)rs;
  }
  if (t == 0) {
    w[P.poison()] = s_bad ? __builtin_bit_cast(float, 0x7fc00000u) : 0.f;
    w[P.poison() + 1] = 0.f;
  }
}

// packed stack weights (split-f16 fragment images + sd0 parameters), see gwtf_layout.h
__global__ void pack_w_kernel(const float* __restrict__ raw, const float* __restrict__ pf, float* __restrict__ out, int C,
                              int f, int G, int FP, int training, int pattern0, int Cper) {
  const GwtfRaw R(f, G);
  const GwtfPackW P(FP);
  const GwtfPackF PF(FP, G);
  const size_t per = P.coupling_size();
  const size_t total = per * (size_t)C;
  const int MB = P.MB(), KS = P.KS();
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / per);
    size_t o = idx - (size_t)c * per;
    float v = 0.f;
    if (o < 2 * P.a16_size()) {
      // one float slot = two consecutive f16 (j = 2*jp, 2*jp+1) of one lane's 8-element fragment
      const int br = (int)(o / P.a16_size());
      o -= (size_t)br * P.a16_size();
      const float* rb = raw + (size_t)c * R.coupling_size() + (size_t)br * R.branch_size();
      const int jp = (int)(o % 4);
      const int lane = (int)((o / 4) % 64);
      const int part = (int)((o / 256) % 2);
      const int m = (int)((o / 512) % MB);
      const int ks = (int)(o / ((size_t)512 * MB));
      const int jo = 16 * m + (lane & 15);  // output feature (row of sd1.weight)
      __half2 pk = __floats2half2_rn(0.f, 0.f);
      if (ks < KS && jo < f) {
        const float* ex = pf + ((size_t)c * 2 + br) * PF.branch_size();     // range-scaling exponents (pack_scales_kernel)
        const int rs = training ? 0 : (int)ex[PF.rs() + jo];       // train packing: no range scaling (batch statistics normalise)
        const float sc = training ? 1.0f : inv_std(rb[R.bn1() + f + jo]);   // train: sd1_bn is applied via fold1
        const GwtfA16Slot sl = gwtf_a16_slot(f, KS, ks, part, jp);   // f = 33..40: merged image of the short k-step
        const bool absf = gwtf_abs_form(f);                          // ... in abs form: every entry halved, columns in the last pair
        float e[2];
        for (int t = 0; t < 2; ++t) {
          const int ji = 32 * ks + 4 * (2 * sl.jsrc + t) + (lane >> 4);  // input feature
          float w = 0.f;
          if (ji < f && !sl.zero) w = ldexpf(rb[R.sd1_w() + (size_t)jo * f + ji] * sc, (training ? 0 : (int)ex[PF.cs() + ji]) - rs - (absf ? 1 : 0));
          const float hi = __half2float(__float2half_rn(w));
          e[t] = sl.lo ? (w - hi) : hi;
        }
        if (absf && sl.zero && !training) {    // train packing: gwtf_train_fold0 writes the columns per level (batch statistics)
          double Ca, Cb, Cc;
          abs_columns(rb, R, f, jo, gwtf_pattern_kept((pattern0 + c % Cper) % 6), sc, &Ca, &Cb, &Cc);
          gwtf_abs_cols(lane >> 4, (float)ldexp(Ca, -rs - 1), (float)ldexp(Cb, -rs - 1), (float)ldexp(Cc, -rs - 1), &e[0], &e[1]);
        }
        pk = __floats2half2_rn(e[0], e[1]);
      }
      v = __builtin_bit_cast(float, pk);
    } else if ((o -= 2 * P.a16_size()) < 2 * P.sd0_size()) {
      const int br = (int)(o / P.sd0_size());
      o -= (size_t)br * P.sd0_size();
      const float* rb = raw + (size_t)c * R.coupling_size() + (size_t)br * R.branch_size();
      const int j = (int)(o % 8), e = (int)((o / 8) % 3), q = (int)((o / 24) % 4), ks = (int)(o / 96);
      const int ft = 32 * ks + 4 * j + q;
      if (ft < f && !training) {   // train: written per coupling by gwtf_train_fold0
        const float* bn = rb + R.bn0();
        const float s = bn[ft] * inv_std(bn[3 * f + ft]);
        const int k = gwtf_pattern_kept((pattern0 + c % Cper) % 6);
        v = e < 2 ? (e < k ? rb[R.sd0_w(ft, e, k)] * s : 0.f) : bn[f + ft] - bn[2 * f + ft] * s;
        v = ldexpf(v, -(int)pf[((size_t)c * 2 + br) * PF.branch_size() + PF.cs() + ft]);
      }
    }
    out[idx] = v;
  }
}

// packed FiLM weights
__global__ void pack_film_kernel(const float* __restrict__ raw, float* __restrict__ out, int C, int f, int G, int FP,
                                 int training) {
  const GwtfRaw R(f, G);
  const GwtfPackF P(FP, G);
  const size_t perb = P.branch_size();
  const size_t total = perb * 2 * (size_t)C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cb = (int)(idx / perb);  // coupling*2 + branch
    size_t o = idx - (size_t)cb * perb;
    const float* rb = raw + (size_t)cb * R.branch_size();  // coupling_size == 2*branch_size
    const float* rsx = out + (size_t)cb * perb + P.rs();   // row exponents, written by pack_scales_kernel (zeros in train packing)
    float v = 0.f;
    if (o >= P.poison()) continue;                         // POISON, RS, CS: pack_scales_kernel's
    if (o < P.c1()) {
      const int which = (int)(o / P.mlp_size());
      o -= (size_t)which * P.mlp_size();
      const float* bn = rb + R.film_bn(which);
      if (o < (size_t)P.GP() * FP) {  // L0T[i][j] = L0[j][i], rows G..GP-1 zero; eval: four latent columns interleaved
        int i = (int)(o / FP), j = (int)(o % FP);
        if (!training) {               // L0Q[i/4][j][i%4]: the eval kernel's lane (feature j) reads 4 k-slots with one dwordx4
          i = 4 * (int)(o / (4 * (size_t)FP)) + (int)(o % 4);
          j = (int)((o / 4) % FP);
        }
        if (j < f && i < G) v = rb[R.film_l0(which) + (size_t)j * G + i];
      } else if ((o -= (size_t)P.GP() * FP) < (size_t)FP) {  // S
        const int j = (int)o;
        if (j < f) v = training ? bn[j] : bn[j] * inv_std(bn[3 * f + j]);
      } else if ((o -= FP) < (size_t)FP) {  // T
        const int j = (int)o;
        if (j < f) v = training ? bn[f + j] : bn[f + j] - bn[2 * f + j] * (bn[j] * inv_std(bn[3 * f + j]));
      } else if ((o -= FP) < (size_t)FP * FP) {  // L1T[i][j] = L1[j][i]; eval: L1Q[i/4][j][i%4] like L0Q
        int i = (int)(o / FP), j = (int)(o % FP);
        if (!training) {
          i = 4 * (int)(o / (4 * (size_t)FP)) + (int)(o % 4);
          j = (int)((o / 4) % FP);
        }
        if (i < f && j < f) v = rb[R.film_l1(which) + (size_t)j * f + i];
        if (which == 1 && j < f) v = ldexpf(v, -(int)rsx[j]);   // the b head produces b 2^-RS: c = C1' + b'/a = 2^-RS (c1 + b/a)
      } else {  // L1B
        const int j = (int)(o - (size_t)FP * FP);
        if (j < f) v = rb[R.film_l1b(which) + j];
        if (which == 1 && j < f) v = ldexpf(v, -(int)rsx[j]);
      }
    } else if ((o -= P.c1()) < (size_t)FP) {  // C1 = -mean/sqrt(var+eps) of sd1_bn
      const int j = (int)o;
      if (j < f) v = ldexpf(-rb[R.bn1() + j] * inv_std(rb[R.bn1() + f + j]), -(int)rsx[j]);
    } else if ((o -= FP) < 2 * (size_t)FP) {  // W2[w][j]
      const int w = (int)(o / FP), j = (int)(o % FP);
      if (j < f) v = ldexpf(rb[R.sd2_w() + (size_t)w * f + j], (int)rsx[j]);
    } else {  // B2
      const int e = (int)(o - 2 * (size_t)FP);
      if (e < 2) v = rb[R.sd2_b() + e];
    }
    out[idx] = v;
  }
}

// exact-fp32 stack record (GwtfPackX): the fp32 operands of the f x f contraction, same folding and range scaling as pack_w_kernel
__global__ void pack_x_kernel(const float* __restrict__ raw, const float* __restrict__ pf, float* __restrict__ out, int C, int f,
                              int G, int FP, int pattern0, int Cper) {
  const GwtfRaw R(f, G);
  const GwtfPackX P(FP);
  const GwtfPackF PF(FP, G);
  const size_t per = P.coupling_size(), total = per * (size_t)C;
  const int MB = P.MB();
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / per);
    size_t o = idx - (size_t)c * per;
    float v = 0.f;
    if (o < 2 * P.a32_size()) {
      const int br = (int)(o / P.a32_size());
      o -= (size_t)br * P.a32_size();
      const float* rb = raw + (size_t)c * R.coupling_size() + (size_t)br * R.branch_size();
      const float* ex = pf + ((size_t)c * 2 + br) * PF.branch_size();
      const int lane = (int)(o % 64), m = (int)((o / 64) % MB), t = (int)(o / ((size_t)64 * MB));
      const int jo = 16 * m + (lane & 15), ji = 4 * t + (lane >> 4);
      if (jo < f && ji < f)
        v = ldexpf(rb[R.sd1_w() + (size_t)jo * f + ji] * inv_std(rb[R.bn1() + f + jo]), (int)ex[PF.cs() + ji] - (int)ex[PF.rs() + jo]);
    } else if ((o -= 2 * P.a32_size()) < 2 * (size_t)FP * 4) {
      const int br = (int)(o / ((size_t)FP * 4)), ft = (int)((o / 4) % FP), e = (int)(o % 4);
      const float* rb = raw + (size_t)c * R.coupling_size() + (size_t)br * R.branch_size();
      if (ft < f && e < 3) {
        const float* bn = rb + R.bn0();
        const float s = bn[ft] * inv_std(bn[3 * f + ft]);
        const int k = gwtf_pattern_kept((pattern0 + c % Cper) % 6);
        v = e < 2 ? (e < k ? rb[R.sd0_w(ft, e, k)] * s : 0.f) : bn[f + ft] - bn[2 * f + ft] * s;
        v = ldexpf(v, -(int)pf[((size_t)c * 2 + br) * PF.branch_size() + PF.cs() + ft]);
      }
    }
    out[idx] = v;
  }
}

}  // namespace

extern "C" size_t gwtf_packed_x_coupling_floats(int f) { return GwtfPackX(gwtf_padded_width(f)).coupling_size(); }

// packed_film: the eval packing of the SAME raw arena (gwtf_pack_weights_k, training = 0) -- its range-scaling exponents are read
extern "C" int gwtf_pack_weights_exact(const float* raw, const float* packed_film, float* packed_x, int K, int Cper, int f, int G,
                                       int pattern0, void* stream) {
  const int C = K * Cper;
  if (K <= 0 || Cper <= 0 || f <= 0 || G <= 0 || f > GWTF_MAX_FP || !raw || !packed_film || !packed_x || pattern0 < 0 || pattern0 > 5)
    return GWTF_E_BADARG;
  const int FP = gwtf_padded_width(f);
  const size_t total = GwtfPackX(FP).coupling_size() * (size_t)C;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_x_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, raw, packed_film, packed_x, C, f, G, FP, pattern0, Cper);
  return (int)hipGetLastError();
}

// K stacks of Cper couplings each, concatenated (raw [K][Cper][record]): every stack starts again at warp pattern `pattern0`
// (the records store sd0.weight with 1 or 2 kept columns depending on the coupling's pattern)
extern "C" int gwtf_pack_weights_k(const float* raw, float* packed_w, float* packed_film, int K, int Cper, int f, int G,
                                   int pattern0, int training, void* stream) {
  const int C = K * Cper;
  // training == 2: the train pipeline's packing -- the stack weights only (its FiLM heads read the raw arena in place,
  // csrc/gwtf_film_train.hip; no range-scaling exponents in train mode): packed_film may be NULL
  const bool stack_only = training == 2;
  if (K <= 0 || Cper <= 0 || f <= 0 || G <= 0 || f > GWTF_MAX_FP || !raw || !packed_w || (!packed_film && !stack_only) || pattern0 < 0 ||
      pattern0 > 5)
    return GWTF_E_BADARG;
  const int FP = gwtf_padded_width(f);
  hipStream_t st = (hipStream_t)stream;
  const int threads = 256;
  if (!stack_only)
    hipLaunchKernelGGL(pack_scales_kernel, dim3(2 * C), dim3(threads), 0, st, raw, packed_film, C, f, G, FP, training, pattern0, Cper);
  {
    const size_t total = GwtfPackW(FP).coupling_size() * (size_t)C;
    const int blocks = (int)((total + threads - 1) / threads < 2048 ? (total + threads - 1) / threads : 2048);
    hipLaunchKernelGGL(pack_w_kernel, dim3(blocks), dim3(threads), 0, st, raw, packed_film, packed_w, C, f, G, FP, training,
                       pattern0, Cper);
  }
  if (!stack_only) {
    const size_t total = GwtfPackF(FP, G).coupling_size() * (size_t)C;
    const int blocks = (int)((total + threads - 1) / threads < 2048 ? (total + threads - 1) / threads : 2048);
    hipLaunchKernelGGL(pack_film_kernel, dim3(blocks), dim3(threads), 0, st, raw, packed_film, C, f, G, FP, training);
  }
  return (int)hipGetLastError();
}

extern "C" int gwtf_pack_weights(const float* raw, float* packed_w, float* packed_film, int C, int f, int G,
                                 int pattern0, int training, void* stream) {
  return gwtf_pack_weights_k(raw, packed_w, packed_film, 1, C, f, G, pattern0, training, stream);
}
