// Buffer layouts shared by the packer, the FiLM kernel and the fused stack kernel.
// Everything is fp32.  "f" is the natural feature width (reference f_n_features),
// "FP" is f rounded up to a multiple of 16 (one MFMA 16x16x4 row block), G the latent width.
//
// RAW ARENA (input of gwtf_pack_weights): C coupling records in direct order, each made of
// two branch records, branch 0 = logvar, branch 1 = mu.  A branch record is the reference's
// parameters and BatchNorm buffers (lib/networks/flows.py:25-50 / 60-85) flattened row-major in
// this order (k = number of kept coordinates, 1 or 2, known from the coupling's warp pattern; w = warped padded to 2
// rows; every tensor as the module stores it, so the host builds the arena from plain views with one torch.cat):
//   sd0.weight[f][k] then (2-k)*f zeros | sd0_bn.{weight,bias,running_mean,running_var}[f] | sd1.weight[f][f] |
//   sd1_bn.{running_mean,running_var}[f] |
//   film_w0.weight[f][G] | film_w0_bn.{weight,bias,running_mean,running_var}[f] | film_w1.weight[f][f] | film_w1.bias[f] |
//   film_b0.weight[f][G] | film_b0_bn.{...}[f] | film_b1.weight[f][f] | film_b1.bias[f] |
//   sd2.weight[2][f] | sd2.bias[2]
#pragma once
#include <stddef.h>

#ifdef __HIPCC__
#define GWTF_HD __host__ __device__ inline
#else
#define GWTF_HD static inline
#endif

#define GWTF_BN_EPS 1e-5f
#define GWTF_MAX_FP 128        // widest padded feature width of the forward kernels (f_n_features <= 128)
#define GWTF_MAX_FP_TRAIN 96   // ... of the train-mode / backward kernels (their LDS working set exceeds 160 KiB beyond)
#define GWTF_MAX_COMPONENTS 64
#define GWTF_STAT_REPLICAS 64   // copies of every train-mode statistic accumulator (atomic contention spreading)

struct GwtfRaw {  // offsets inside one branch record
  int f, G;
  GWTF_HD GwtfRaw(int f_, int G_) : f(f_), G(G_) {}
  GWTF_HD size_t sd0_w() const { return 0; }
  // sd0.weight[j][e] of a coupling with k kept coordinates (e < k; the record reserves 2*f floats either way)
  GWTF_HD size_t sd0_w(int j, int e, int k) const { return (size_t)j * k + e; }
  GWTF_HD size_t bn0() const { return (size_t)2 * f; }              // gamma, beta, rm, rv
  GWTF_HD size_t sd1_w() const { return (size_t)6 * f; }
  GWTF_HD size_t bn1() const { return (size_t)6 * f + (size_t)f * f; }  // rm, rv
  GWTF_HD size_t film_size() const { return (size_t)f * G + 5 * (size_t)f + (size_t)f * f; }
  GWTF_HD size_t film(int which) const { return (size_t)8 * f + (size_t)f * f + which * film_size(); }
  GWTF_HD size_t film_l0(int which) const { return film(which); }
  GWTF_HD size_t film_bn(int which) const { return film(which) + (size_t)f * G; }
  GWTF_HD size_t film_l1(int which) const { return film_bn(which) + 4 * (size_t)f; }
  GWTF_HD size_t film_l1b(int which) const { return film_l1(which) + (size_t)f * f; }
  GWTF_HD size_t sd2_w() const { return film(2); }
  GWTF_HD size_t sd2_b() const { return sd2_w() + 2 * (size_t)f; }
  GWTF_HD size_t branch_size() const { return sd2_b() + 2; }
  GWTF_HD size_t coupling_size() const { return 2 * branch_size(); }
};

// PACKED STACK WEIGHTS (read by the fused kernel through LDS), per coupling.
// The f x f contraction (sd1) runs on v_mfma_f32_16x16x32_f16 with every fp32 operand split into two
// f16 parts, x = hi + lo (both round-to-nearest, |lo| <= 2^-12 |x|, representation error 2^-23 |x|, f16
// subnormals are honoured by the MFMA -- tools/diag/f16probe.hip), and three products
// W_hi*h_hi + W_hi*h_lo + W_lo*h_hi accumulated in fp32 (the dropped W_lo*h_lo is below 2^-24 relative).
// K positions are dealt round-robin over the four 16-lane quarters so that a k-step with few valid
// features costs every lane equally little: position (q = lane>>4, j) of k-step ks <-> input feature
// 32*ks + 4*j + q.
//   A16[branch][ks][m][part][lane][8] f16 : part 0 = hi, 1 = lo of W[16*m + (lane&15)][32*ks + 4*j + (lane>>4)],
//                                           W = sd1.weight with sd1_bn's 1/sqrt(var+eps) folded into its rows
//   SD0[branch][ks][q][3][8] f32          : {w0[.][0]*s, w0[.][1]*s, beta - mean*s} for features 32*ks + 4*j + q,
//                                           s = gamma/sqrt(var+eps) of sd0_bn
// Exception (f = 33..40: the second k-step has at most two valid k positions per lane): that k-step is contracted by ONE
// MFMA against B' = [h_hi (2) | h_lo (2) | h_hi (2) | 0 0], and its hi image holds A' = [W_hi (2) | W_hi (2) | W_lo (2) | 0 0]
// -- slot pair 0 as above, pairs 1-2 in place of the zeros that pad a short k-step; the lo image is unchanged.
// ABS FORM of these widths (f = 33..40, every packer, every consumer).  relu(x) = (x + |x|)/2 and the pre-activation is linear in
// the kept coordinates, pre_k = wa_k xa + wb_k xb + c0_k, so
//     W relu(pre) = 1/2 W |pre|  +  xa (1/2 W wa) + xb (1/2 W wb) + (1/2 W c0):
// the B operand carries |pre| (the absolute value is a source modifier of v_cvt_pk_f16_f32 / v_fma_mix_f32: no v_max), every A entry
// is stored HALVED, and the three columns Ca = 1/2 W wa, Cb = 1/2 W wb, Cc = 1/2 W c0 with their split-f16 products take the LAST
// slot pair (k positions 6, 7) of the merged image, which the eight k positions of the four quarters q = lane >> 4 fill exactly:
//     q = 0: A = (Ca_hi, Ca_hi)  against  B = (xa_hi, xa_lo)        q = 2: A = (Cb_hi, Cb_lo)  against  B = (xb_lo, xb_hi)
//     q = 1: A = (Ca_lo, Cb_hi)  against  B = (xa_hi, xb_hi)        q = 3: A = (Cc_hi, Cc_lo)  against  B = (1, 1)
// No extra MFMA; 80 v_max per coupling and wavefront become ~40 VALU that split and place xa / xb.
// The record is padded to a whole number of 1-KiB LDS-DMA pieces.
GWTF_HD bool gwtf_abs_form(int f) { return f >= 33 && f <= 40; }
// the two fp32 values (each exactly representable in f16 after rounding to nearest) of quarter q's column slot pair
GWTF_HD void gwtf_abs_cols(int q, float Ca, float Cb, float Cc, float* e0, float* e1) {
  const float ah = (float)(_Float16)Ca, bh = (float)(_Float16)Cb, ch = (float)(_Float16)Cc;
  if (q == 0) { *e0 = ah; *e1 = ah; }
  else if (q == 1) { *e0 = Ca - ah; *e1 = bh; }
  else if (q == 2) { *e0 = bh; *e1 = Cb - bh; }
  else { *e0 = ch; *e1 = Cc - ch; }
}
struct GwtfA16Slot { int jsrc; bool lo, zero; };
// float slot jp (= k positions 2jp, 2jp+1) of image `part` of k-step ks: which pair of k positions, and which f16 part, it holds
GWTF_HD GwtfA16Slot gwtf_a16_slot(int f, int KS, int ks, int part, int jp) {
  const bool merged = KS == 2 && ks == 1 && (f + 3) / 4 - 8 <= 2 && part == 0;
  GwtfA16Slot s;
  s.jsrc = merged ? 0 : jp;
  s.lo = merged ? jp == 2 : part == 1;
  s.zero = merged && jp == 3;          // abs form: the column slot pair (gwtf_abs_cols), written separately
  return s;
}
struct GwtfPackW {
  int FP;
  GWTF_HD GwtfPackW(int FP_) : FP(FP_) {}
  GWTF_HD int MB() const { return FP / 16; }
  GWTF_HD int KS() const { return (FP + 31) / 32; }
  GWTF_HD size_t a16_size() const { return (size_t)KS() * MB() * 2 * 256; }  // floats (2 f16 each), one branch
  GWTF_HD size_t a16(int branch) const { return branch * a16_size(); }
  GWTF_HD size_t sd0_size() const { return (size_t)KS() * 4 * 24; }
  GWTF_HD size_t sd0(int branch) const { return 2 * a16_size() + branch * sd0_size(); }
  GWTF_HD size_t used_size() const { return 2 * a16_size() + 2 * sd0_size(); }
  GWTF_HD size_t coupling_size() const { return (used_size() + 255) / 256 * 256; }
};

// EXACT-FP32 STACK RECORD (gwtf_pack_weights_exact -> csrc/gwtf_stack_exact.hip; the sd1 contraction on v_mfma_f32_16x16x4_f32 with
// UNSPLIT fp32 operands: the reference's own arithmetic, reference flows.py:25-31), per coupling:
//   A32[branch][t][m][lane] f32 : W1'[16 m + (lane & 15)][4 t + (lane >> 4)], t < FP / 4 -- the A operand of MFMA (t, m): row = output
//       feature, k = input feature 4 t + q; W1' = sd1.weight with sd1_bn's 1/sqrt(var + eps) folded in and the SAME power-of-two range
//       scaling as the split images (2^(CS[j] - RS[i]): exact in fp32), so the FiLM record (which carries RS) is shared
//   SD0X[branch][FP][4] f32     : {w0a, w0b, c0, 0} 2^-CS of the folded sd0 in natural feature order
// padded to whole 1-KiB LDS-DMA pieces.
struct GwtfPackX {
  int FP;
  GWTF_HD GwtfPackX(int FP_) : FP(FP_) {}
  GWTF_HD int MB() const { return FP / 16; }
  GWTF_HD int KK() const { return FP / 4; }
  GWTF_HD size_t a32_size() const { return (size_t)FP * FP; }                 // floats, one branch (KK * MB * 64)
  GWTF_HD size_t a32(int branch) const { return branch * a32_size(); }
  GWTF_HD size_t sd0x(int branch) const { return 2 * a32_size() + (size_t)branch * FP * 4; }
  GWTF_HD size_t used_size() const { return 2 * a32_size() + 2 * (size_t)FP * 4; }
  GWTF_HD size_t coupling_size() const { return (used_size() + 255) / 256 * 256; }
};

// BACKWARD RECORD (read by the backward kernels through LDS), per coupling:
//   W1T[branch][mi][ks][part][lane][8] f16 : hi / lo of W1[j][i], the TRANSPOSED sd1 weight as the A operand of
//       dh = W1^T dacc on v_mfma_f32_16x16x32_f16: row i = 16*mi + (lane&15) (input feature of sd1), k-slot (ks, q = lane>>4, e)
//       <-> output feature j = 32*ks + 16*(e>>2) + 4*q + (e&3), i.e. accumulator tile 2*ks + (e>>2), row 4*q + (e&3): the
//       C-layout registers of dacc become the B operand with no data movement
//   SD0N[branch][FP][4] f32 : {w0a, w0b, c0, 0} of the folded sd0 in natural feature order
struct GwtfPackB {
  int FP;
  GWTF_HD GwtfPackB(int FP_) : FP(FP_) {}
  GWTF_HD int MB() const { return FP / 16; }
  GWTF_HD int KS() const { return (FP + 31) / 32; }
  GWTF_HD size_t w1t_size() const { return (size_t)MB() * KS() * 2 * 256; }   // floats (2 f16 each), one branch
  GWTF_HD size_t w1t(int branch) const { return branch * w1t_size(); }
  GWTF_HD size_t sd0n(int branch) const { return 2 * w1t_size() + (size_t)branch * FP * 4; }
  GWTF_HD size_t coupling_size() const { return 2 * w1t_size() + 2 * (size_t)FP * 4; }
};

// value of one float slot (= two f16) of a W1T image; W(j, i) returns the sd1 weight (0 outside the natural width)
template <typename WFn>
GWTF_HD float gwtf_w1t_slot(const GwtfPackB& P, size_t o /* offset inside one branch's image */, WFn W) {
  const int jp = (int)(o % 4), lane = (int)((o / 4) % 64), part = (int)((o / 256) % 2);
  const int ks = (int)((o / 512) % P.KS()), mi = (int)(o / ((size_t)512 * P.KS()));
  const int i = 16 * mi + (lane & 15), q = lane >> 4;
  _Float16 h[2];
  for (int t = 0; t < 2; ++t) {
    const int e = 2 * jp + t;
    const int j = 32 * ks + 16 * (e >> 2) + 4 * q + (e & 3);
    const float w = W(j, i);
    const _Float16 hi = (_Float16)w;
    h[t] = part == 0 ? hi : (_Float16)(w - (float)hi);
  }
  typedef _Float16 gwtf_h2 __attribute__((ext_vector_type(2)));
  const gwtf_h2 pk = {h[0], h[1]};
  return __builtin_bit_cast(float, pk);
}

// PACKED FILM WEIGHTS (read by gwtf_film_forward), per coupling, per branch:
//   for which in {w,b}: L0T[GP][FP] (GP = G rounded up to 16, zero rows beyond G) | S[FP] | T[FP] | L1T[FP][FP] | L1B[FP]
//   C1[FP] (sd1_bn shift: -mean/sqrt(var+eps)) | W2[2][FP] | B2[4] (sd2 bias, 2 used; slot 2 = POISON: 0, or NaN when the
//   coupling's raw record holds a non-finite value -- added to both biases by the FiLM kernel so that diverged weights reach
//   every output, reference training.py:43-46) | RS[FP] | CS[FP]
// RANGE SCALING (eval packing; exact, powers of two only).  The split-f16 contraction needs its operands inside the f16
// range: CS[j] = floor(log2(|w0a_j| + |w0b_j| + |c0_j|)) of the folded sd0 -- the sd0 record is stored times 2^-CS[j], so
// h'_j = relu(pre_j) 2^-CS[j] < 2 max(1, |x|) and f16 overflows only beyond |x| ~ 3e4 (the stack kernel flags that) --
// and RS[i] = floor(log2(max_j |W1'_ij| 2^CS[j])) - 12: the A images hold W1'_ij 2^(CS[j] - RS[i]) (row maximum in [2^12, 2^13):
// hi and lo parts both normal for every entry within 2^-14 of it).  The accumulators then carry 2^-RS[i] (y_i + c_i): RS is
// folded into the FiLM weights that produce c (C1, the b head's second Linear) and, inversely, into W2 (u = W2 a), so the
// FiLM and stack kernels are unchanged.  Train packing stores zeros (batch statistics normalise the activations).
// eval packing stores L0T and L1T with four input columns interleaved, L0Q[GP/4][FP][4] / L1Q[FP/4][FP][4] (same sizes):
// film_eval_kernel's lane owns
// one feature and reads its four k-slots of an MFMA k-group with one 16-byte load, 16 lanes = 256 contiguous bytes.
// eval: S = gamma/sqrt(var+eps), T = beta - mean*S (running statistics)
// train: S = gamma, T = beta (batch statistics are taken inside the FiLM kernel)
struct GwtfPackF {
  int FP, G;
  GWTF_HD GwtfPackF(int FP_, int G_) : FP(FP_), G(G_) {}
  GWTF_HD int GP() const { return (G + 15) / 16 * 16; }
  GWTF_HD size_t mlp_size() const { return (size_t)GP() * FP + 3 * (size_t)FP + (size_t)FP * FP; }
  GWTF_HD size_t l0t(int which) const { return which * mlp_size(); }
  GWTF_HD size_t s(int which) const { return l0t(which) + (size_t)GP() * FP; }
  GWTF_HD size_t t(int which) const { return s(which) + FP; }
  GWTF_HD size_t l1t(int which) const { return t(which) + FP; }
  GWTF_HD size_t l1b(int which) const { return l1t(which) + (size_t)FP * FP; }
  GWTF_HD size_t c1() const { return 2 * mlp_size(); }
  GWTF_HD size_t w2() const { return c1() + FP; }
  GWTF_HD size_t b2() const { return w2() + 2 * (size_t)FP; }
  GWTF_HD size_t poison() const { return b2() + 2; }
  GWTF_HD size_t rs() const { return b2() + 4; }
  GWTF_HD size_t cs() const { return rs() + FP; }
  GWTF_HD size_t branch_size() const { return cs() + FP; }
  GWTF_HD size_t coupling_size() const { return 2 * branch_size(); }
};
#define GWTF_X_LIMIT 3.0e4f   // |coordinate| beyond which the f16 image of sd0's activations may overflow: results are NaN

#ifdef __HIPCC__
// Bit pattern of a float for NaN / Inf tests.  The library is built with -fno-honor-nans: the compiler recognises integer
// idioms such as (bits & 0x7f800000) == 0x7f800000 as floating-point class tests and, allowed to assume that no value is a
// NaN, drops their NaN half (measured: the packer's poison never fired).  The empty asm makes the bits opaque.
__device__ __forceinline__ unsigned gwtf_float_bits(float x) {
  unsigned b = __builtin_bit_cast(unsigned, x);
  asm volatile("" : "+v"(b));
  return b;
}
__device__ __forceinline__ bool gwtf_nonfinite(float x) { return (gwtf_float_bits(x) & 0x7f800000u) == 0x7f800000u; }
#endif

// FILM OUTPUT (written by gwtf_film_forward, read by the fused kernel through LDS), per (shape b, coupling c):
//   for branch in {logvar, mu}:  c[FP] | w20a[FP] | w21a[FP]   with a = eps + exp(cond_w(g)) > 0, b = cond_b(g):
//       c = C1 + b/a   (start value of the sd1 accumulators:  relu(a*(y + C1) + b) = a * relu(y + c))
//       w2ka = W2[k][j] * a
//   bias float4 : {b2_logvar[0], b2_logvar[1], b2_mu[0], b2_mu[1]}
GWTF_HD size_t gwtf_film_out_size(int FP) { return 6 * (size_t)FP + 4; }

// Per-component strides (in floats) of the train-mode pipeline's buffers when K mixture components run through one launch
// (csrc/gwtf_train.hip, "K-batched pipeline"): component k of a launch adds k * stride to the base pointer.  All zero and
// Ctot == Cper for a single stack.  FiLM-side arrays are indexed [shape][Ctot = K*Cper][...] with coupling k*Cper + c.
struct GwtfKS {
  size_t raw, pw, pb, x, pts, mom, ys, bn, gsd0, gbias, gstats, gmom, dw1;
  int Cper, Ctot;
  int tune;      // the call's GWTF_TUNE_* word (include/gwtf.h), carried to every launcher of the pipeline
  int tpw;       // tiles of one shape a workgroup walks in the light backward pass (set by its launcher; 0 / 1 = one)
};

// The gradient combine of the backward level processed BEFORE this one, applied on the fly by this level's passes (train pipeline).
// A level's input gradient is  g_in = g_a + d(moments)/dx = g_a + gM + Q x  (Q_aa = 2 gM_aa, Q_ab = gM_ab) with x that level's INPUT
// = this level's OUTPUT, which the backward kernels recompute anyway: they read the previous level's raw g_a and add the moment path
// themselves instead of a separate pass over the points per level (3 x 12 B per point read + written, one launch on the chain).
// gm: the nine moment gradients of that level [K][16] (written by its sd0 fold's backward; all-reduced when data parallel);
// null: nothing to add (first backward level, eval mode).
struct GwtfCombine {
  const float* gm;
  size_t gm_sk;         // component stride (floats)
};

// warp pattern of coupling c in direct order (reference flows.py:129-148, decoders.py:49-52):
// index (pattern0 + c) % 6 -> 0:[0] 1:[1] 2:[2] 3:[0,1] 4:[0,2] 5:[1,2]
GWTF_HD int gwtf_pattern_kept(int pat) { return pat < 3 ? 2 : 1; }
GWTF_HD void gwtf_pattern_dims(int pat, int* k0, int* k1, int* w0, int* w1) {
  // kept dims (k1 = -1 when only... two kept -> both valid), warped dims (w1 = -1 when one warped)
  switch (pat) {
    case 0: *w0 = 0; *w1 = -1; *k0 = 1; *k1 = 2; break;
    case 1: *w0 = 1; *w1 = -1; *k0 = 0; *k1 = 2; break;
    case 2: *w0 = 2; *w1 = -1; *k0 = 0; *k1 = 1; break;
    case 3: *w0 = 0; *w1 = 1; *k0 = 2; *k1 = -1; break;
    case 4: *w0 = 0; *w1 = 2; *k0 = 1; *k1 = -1; break;
    default: *w0 = 1; *w1 = 2; *k0 = 0; *k1 = -1; break;
  }
}
