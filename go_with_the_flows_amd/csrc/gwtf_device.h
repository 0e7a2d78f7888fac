// gwtf_device.h -- device-side building blocks shared by the forward (gwtf_stack.hip) and backward
// (gwtf_bwd.hip) kernels: tile configuration, the split-f16 sd1 contraction, the quarter transpose-reduce.
#pragma once
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "../../include/gwtf.h"

namespace gwtf_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int MB>
struct Cfg {
  static constexpr int FP = 16 * MB;
  static constexpr int KS = (FP + 31) / 32;               // k-steps of 32 input features
  static constexpr int A16 = KS * MB * 2 * 256;           // floats, one branch (hi + lo f16 fragment images)
  static constexpr int SD0 = KS * 4 * 24;                 // floats, one branch
  static constexpr int PW = (2 * A16 + 2 * SD0 + 255) / 256 * 256;  // packed weights per coupling (whole DMA pieces)
  static constexpr int FS = 6 * FP + 4;                   // FiLM output per (shape, coupling)
  static constexpr int FSP = (FS + 255) / 256 * 256;
  static constexpr int LAYER = PW + FSP;                  // one LDS buffer
};

__device__ __forceinline__ float sel3(float a, float b, float c, int d) { return d == 0 ? a : (d == 1 ? b : c); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// x = hi + lo with both parts rounded to nearest f16 (|lo| <= 2^-12 |x|): v_cvt_pk_f16_f32, two mixed-precision
// FMAs computing h - float(hi) with the f16 operand converted inside the instruction (v_fma_mix_f32; the compiler
// does not form it by itself: it emitted 2 x v_cvt_f32_f16 + v_pk_add_f32), v_cvt_pk_f16_f32: 4 VALU for two values.
// MEASURED AND REJECTED (round 2): writing the lo pair directly with v_fma_mixlo_f16 + v_fma_mixhi_f16 (3 VALU, no second
// conversion).  Same values on paper, and the census drops by 32 VALU per coupling and wave, but (1) the airplane kernel got
// SLOWER (0.495 -> 0.510 ms: the two partial-register writes serialise on one destination) and (2) about one backward pass
// in five came out wrong by 1e-3..1e-2 relative (same inputs, same process: tools/diag/bwd_repeat2.py) although hipcc did
// put an s_nop between the two partial writes -- a forwarding hazard of 16-bit destination writes that the inline asm hides
// from the compiler on the consumer side.  tests/test_gpu_parity.py::test_backward_is_reproducible_run_to_run guards it.
__device__ __forceinline__ void split_pair(f32x2 h, f16x2& hi, f16x2& lo) {
  hi = __builtin_convertvector(h, f16x2);
  f32x2 r;
  const unsigned hib = __builtin_bit_cast(unsigned, hi);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hib), "v"(h[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hib), "v"(h[1]));
  r[0] = r0;
  r[1] = r1;
  lo = __builtin_convertvector(r, f16x2);
}

// split of |h| (abs form, gwtf_layout.h): the absolute value rides on the source modifiers of all three instructions
__device__ __forceinline__ void split_pair_abs(f32x2 h, f16x2& hi, f16x2& lo) {
  unsigned hib;
  asm("v_cvt_pk_f16_f32 %0, |%1|, |%2|" : "=v"(hib) : "v"(h[0]), "v"(h[1]));
  hi = __builtin_bit_cast(f16x2, hib);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, |%2| op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hib), "v"(h[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, |%2| op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hib), "v"(h[1]));
  const f32x2 r = {r0, r1};
  lo = __builtin_convertvector(r, f16x2);
}

// The abs form's B entries for the column slot pair (k positions 6, 7 of the merged k-step) of this lane's quarter q:
// q0 (xa_hi, xa_lo) | q1 (xa_hi, xb_hi) | q2 (xb_lo, xb_hi) | q3 (1, 1)   -- two f16 in one register
__device__ __forceinline__ unsigned abs_form_x_slots(float xa, float xb, int q) {
  const f32x2 xx = {xa, xb};
  f16x2 xh, xl;
  split_pair(xx, xh, xl);
  const unsigned sel = q == 0 ? 0x01000504u : (q == 1 ? 0x07060504u : 0x07060302u);
  const unsigned v = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, xh), __builtin_bit_cast(unsigned, xl), sel);
  return q == 3 ? 0x3C003C00u : v;
}

// Sum over the 16 lanes of a DPP row (= one quarter of the wavefront = the 16 points of a column group); every lane ends
// with the total.  Four v_add_f32 with DPP operand modifiers -- no LDS traffic (a __shfl_xor goes through ds_bpermute).
// MEASURED AND REJECTED (round 2): a transpose-reduce of twelve values at a time (bank-masked v_add_f32_dpp for the mirror
// steps, selects for the quad steps: 29 VALU per twelve values instead of 48, one 64-lane ds_add_f32 instead of twelve
// lane-0 branches).  Correct, 250 fewer VALU and 90 fewer branches per backward wave -- and the airplane train step got 5 %
// SLOWER (19.0 -> 20.0 ms): the backward kernel is latency-bound, and the twelve-deep dependent chain of one block replaces
// 36 independent four-deep chains the scheduler was interleaving with the neighbouring FMAs.
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}

// Row sums that end in an LDS atomic (the per-feature sums over a workgroup's points).  GWTF_ROWSUM_MODE:
//   0  row_sum16 + an atomic from lane 0 of each row.  The compiler sinks the LAST DPP add into the exec-masked block of the atomic and
//      leaves `v_mov_b32 old, 0` + `v_mov_b32_dpp` outside it: 6 VALU per sum (tools/isa_lines.py census, docs/LOG.md round 5).
//   1  the same four steps with the total pinned outside the masked block (an empty asm use): 4 fused v_add_f32_dpp per sum.
//   2  two quad steps only; the four quad leaders of a row (i16 % 4 == 0) add their partials to the same LDS word: 2 VALU per sum,
//      a 4-way same-address conflict per row in the LDS atomic instead of two more dependent DPP steps.
#ifndef GWTF_ROWSUM_MODE
#define GWTF_ROWSUM_MODE 1
#endif
__device__ __forceinline__ float row_sum_part(float v) {
  if (GWTF_ROWSUM_MODE == 0) return row_sum16(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  if (GWTF_ROWSUM_MODE == 1) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  }
  asm("" : "+v"(v));      // the sum is complete in EVERY lane here: nothing of it moves into the owners' masked block
  return v;
}
// lanes of a 16-lane row that hand their row_sum_part to the LDS atomic
__device__ __forceinline__ bool row_sum_owner(int i16) { return GWTF_ROWSUM_MODE == 2 ? (i16 & 3) == 0 : i16 == 0; }

// What a row total does after row_sum_part.  GWTF_ROWSUM_KEEP:
//   0  lane 0 of the row adds it to the workgroup's LDS word (36 exec-masked 4-lane LDS atomics per branch and wave in the merged pass)
//   2  kernels whose workgroup owns ONE tile: lane 0 of the row STORES it into the wave's own slot (plain ds_write: every slot is written
//      once) and the flush adds the four waves' slots -- merged pass 111.0 / 103.9 -> 106.9 / 101.0 us, statistics pass 22.2 -> 21.5.
// Measured and rejected (docs/experiments/r5_rowsum_keeper.patch): PARKING the totals (total idx in lane idx % 16 of register idx / 16,
// one v_cndmask each) and leaving in ceil(NV / 16) full-wave operations -- as LDS atomics 113.3 -> 121.1 us (an LDS float atomic costs
// per ACTIVE LANE: 3 x 64 lanes are the atomic work of 36 x 4, and the 72 selects come on top), as plain stores 107.3 / 100.1 (= mode 2).
#ifndef GWTF_ROWSUM_KEEP
#define GWTF_ROWSUM_KEEP 2
#endif

// Sum the per-quarter partials o[nb] over the four 16-lane quarters so that the lane in quarter q ends
// with the total of point block (q & (NB-1)).
template <int NB>
__device__ __forceinline__ float quarter_reduce(const float (&o)[NB], int q) {
  // A transpose-reduce over the quarters on gfx950's lane-swap instructions: v_permlane16_swap exchanges the odd 16-lane rows of its
  // first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second
  // (tools/diag/swap_probe.hip) -- register-file moves where shuffles make two DEPENDENT ds_bpermute round trips through the LDS
  // crossbar plus selects, on the coupling boundary's critical chain (reduce -> tail -> broadcast -> next fragments).  Every tile
  // shape adds the same pairs in the same order, (Q0 + Q1) + (Q2 + Q3): a shape's result does not depend on the tile it ran with
  // (test_full_size_properties).  Inline asm, because hipcc 7.2 miscompiles the two-result builtins when both results feed arithmetic
  // (reads the second result from the first register, tools/diag/qr_probe.hip); the two v_nop are the wait states a VALU write of an
  // operand needs before the swap reads it (cdna_hip_programming.md T21) -- the hazard recogniser does not look inside asm.
  // PRECONDITION (also of wave_sum / wave_max below): every operand is the result of a PLAIN VALU instruction (v_add / v_fma / v_max /
  // v_mov ...).  An MFMA accumulator or the result of a transcendental (v_exp / v_rcp / v_rsq / v_sqrt: their own forwarding latency)
  // handed in directly needs more wait states than the two v_nop provide and the compiler will not add them -- pass such a value
  // through an arithmetic instruction first.  Every caller today passes v_fmac / v_add / v_max results; the guards are
  // test_backward_is_reproducible_run_to_run and the full-grid tests (tests/test_gpu_fullgrid.py: every launch twice, bit for bit).
  if constexpr (NB == 4) {
    float a0 = o[0], a1 = o[1], a2 = o[2], a3 = o[3];
    asm("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"      // one pair of wait states covers both
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    float s01 = a0 + a1, s23 = a2 + a3;       // rows: b0 (Q0+Q1), b1 (Q0+Q1), b0 (Q2+Q3), b1 (Q2+Q3) | the same for blocks 2, 3
    asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s01), "+v"(s23));
    return s01 + s23;                         // row q: block q
  } else if constexpr (NB == 2) {
    float a0 = o[0], a1 = o[1];
    asm("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
    float s = a0 + a1, t = s;                 // rows: b0 (Q0+Q1), b1 (Q0+Q1), b0 (Q2+Q3), b1 (Q2+Q3)
    asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
    return s + t;                             // row q: block q & 1
  } else {
    float s = o[0], t = s;
    asm("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(s), "+v"(t));   // s: rows 0 0 2 2, t: rows 1 1 3 3
    float u = s + t, v = u;                   // rows: (Q0+Q1) x 2, (Q2+Q3) x 2
    asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));
    return u + v;
  }
}

// Whole-wavefront sum / maximum, every lane ends with the result: DPP inside the 16-lane rows, lane swaps across them -- six
// register-file steps where a shuffle ladder makes six DEPENDENT ds_bpermute round trips.
__device__ __forceinline__ float wave_sum(float v) {
  const float r[1] = {row_sum16(v)};
  return quarter_reduce<1>(r, 0);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
  float t = v;
  asm("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(t));
  float u = fmaxf(v, t), w = u;
  asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
  return fmaxf(u, w);
}

// ---- moment path of the train-mode backward ---------------------------------------------------------------------------------------
// sd0_bn's batch statistics are analytic in the 9 first / second moments of a level's input coordinates (fold0), so the level's
// input gradient carries  gM_a + sum_b Q_ab x_b  per point, Q_aa = 2 gM_aa, Q_ab = gM_ab: the nine moment gradients gM come from the
// five sums {gE0, gE1, gC00, gC01, gC11} of the sd0 fold's backward (r5) and the kept coordinates' moments.
__device__ __forceinline__ int mom2_index(int a, int b) {  // index of S x_a x_b inside the 9-vector, a <= b
  return a == 0 ? 3 + b : (a == 1 ? 5 + b : 8);
}
struct KeptMoments { double e0, e1, c00, c01, c11; };
__device__ inline KeptMoments kept_moments(const float (&mom)[9], int k0, int k1, double n_total) {
  auto M = [&](int i) { float v = 0.f;
#pragma unroll
    for (int u = 0; u < 9; ++u) v = u == i ? mom[u] : v;
    return v; };
  KeptMoments r;
  r.e0 = M(k0) / n_total;
  r.e1 = k1 >= 0 ? M(k1) / n_total : 0.0;
  r.c00 = M(mom2_index(k0, k0)) / n_total - r.e0 * r.e0;
  r.c11 = k1 >= 0 ? M(mom2_index(k1, k1)) / n_total - r.e1 * r.e1 : 0.0;
  r.c01 = k1 >= 0 ? M(mom2_index(k0 < k1 ? k0 : k1, k0 < k1 ? k1 : k0)) / n_total - r.e0 * r.e1 : 0.0;
  return r;
}
// One fold block's share of the nine moment gradients gm[0..8] (the rest of gm[16] zero), LINEAR in its five sums r5: the blocks
// (and, data parallel, the ranks) add their shares up -- float atomics / an all-reduce of 16 floats -- and every consumer reads gM.
__device__ inline void moment_grad_terms(const KeptMoments& km, const double (&r5)[5], int pat, double n_total, float* gm) {
  int k0, k1, w0d, w1d;
  gwtf_pattern_dims(pat, &k0, &k1, &w0d, &w1d);
  const double gC00 = r5[2], gC01 = r5[3], gC11 = r5[4];
  const double gE0 = r5[0] - 2.0 * km.e0 * gC00 - km.e1 * gC01;
  const double gE1 = r5[1] - 2.0 * km.e1 * gC11 - km.e0 * gC01;
  const double inv_n = 1.0 / n_total;
#pragma unroll
  for (int i = 0; i < 16; ++i) gm[i] = 0.f;
  gm[k0] = (float)(gE0 * inv_n);
  gm[mom2_index(k0, k0)] = (float)(gC00 * inv_n);
  if (k1 >= 0) {
    gm[k1] = (float)(gE1 * inv_n);
    gm[mom2_index(k1, k1)] = (float)(gC11 * inv_n);
    gm[mom2_index(k0 < k1 ? k0 : k1, k0 < k1 ? k1 : k0)] = (float)(gC01 * inv_n);
  }
}

// sd0 (+ folded sd0_bn) + ReLU + f16 split in the lane that owns each MFMA k-slot, then the f x f contraction
// acc[m][nb] = cinit[m] + W1'[16m.., :] . h0[:, points of block nb] on v_mfma_f32_16x16x32_f16 (3 products).
// MG: 1 = the caller knows at compile time that the last k-step is the merged abs-form one (f = 33..40: the whole contraction is
// then one basic block), 0 = that it is not, -1 = decided at run time from kk_steps (wave-uniform branches).
template <int MB, int NB, bool KEEP2, int MG = -1>
__device__ __forceinline__ void sd1_contract(const float* __restrict__ L, int br, int kk_steps, int lane, int q,
                                             const float (&xa)[NB], const float (&xb)[NB], const f32x4 (&cinit)[MB],
                                             f32x4 (&acc)[MB][NB]) {
  using K = Cfg<MB>;
  constexpr int KS = K::KS;
  static_assert(MG != 1 || KS == 2, "the merged k-step exists only with two k-steps");
  const int nj_last = MG == 1 ? 2 : kk_steps - 8 * (KS - 1);   // valid k positions per lane in the last k-step (1..8; merged: 1..2, unused)
  const bool merged = MG < 0 ? (KS == 2 && nj_last <= 2) : MG == 1;
    const float* aimg = L + br * K::A16 + lane * 4;
    const float* sd0 = L + 2 * K::A16 + br * K::SD0 + q * 24;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (KS == 2 && ks == 1 && merged) {
        // a last k-step with at most 2 valid k-slots per lane (f = 33..40): its three products in ONE MFMA, the lane's 8
        // k-slots holding B' = [h_hi (2) | h_lo (2) | h_hi (2) | 0 0] against A' = [W_hi | W_hi | W_lo | 0 0] (the hi image of
        // this k-step as gwtf_pack.hip writes it for these widths)
        // ABS FORM (gwtf_layout.h): B holds |pre|, the last slot pair the split kept coordinates against the packed columns
        const f32x4* sp = reinterpret_cast<const f32x4*>(sd0 + ks * 96);
        const f32x2 wa2 = {sp[0][0], sp[0][1]}, wb2 = {KEEP2 ? sp[2][0] : 0.f, KEEP2 ? sp[2][1] : 0.f}, cc2 = {sp[4][0], sp[4][1]};
        f16x8 bm[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const f32x2 xa2 = {xa[nb], xa[nb]}, xb2 = {xb[nb], xb[nb]};
          const f32x2 pre = KEEP2 ? __builtin_elementwise_fma(wa2, xa2, __builtin_elementwise_fma(wb2, xb2, cc2))
                                  : __builtin_elementwise_fma(wa2, xa2, cc2);
          f16x2 hi, lo;
          split_pair_abs(pre, hi, lo);
          const f16x2 xs = __builtin_bit_cast(f16x2, abs_form_x_slots(xa[nb], KEEP2 ? xb[nb] : 0.f, q));
          bm[nb] = f16x8{hi[0], hi[1], lo[0], lo[1], hi[0], hi[1], xs[0], xs[1]};
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const f16x8 am = *reinterpret_cast<const f16x8*>(aimg + ((ks * MB + m) * 2 + 0) * 256);   // A', written by the packer
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bm[nb], acc[m][nb], 0, 0, 0);
        }
        continue;
      }
      // sd0 + sd0_bn + ReLU for this lane's 8 k positions (features 32*ks + 4*j + q), split into f16 hi/lo
      const f32x4* sp = reinterpret_cast<const f32x4*>(sd0 + ks * 96);
      const f32x4 wa[2] = {sp[0], sp[1]};
      const f32x4 wb[2] = {KEEP2 ? sp[2] : f32x4{0.f, 0.f, 0.f, 0.f}, KEEP2 ? sp[3] : f32x4{0.f, 0.f, 0.f, 0.f}};
      const f32x4 cc[2] = {sp[4], sp[5]};
      f16x8 bhi[NB], blo[NB];
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        if (ks + 1 < KS || 2 * jp < nj_last) {     // wave-uniform: skip k positions beyond ceil(f/4)
          // two k positions at a time: packed fp32 FMAs (v_pk_fma_f32: 6 cycles for two FMAs instead of 8)
          const int j0 = 2 * jp;
          const f32x2 wa2 = {wa[j0 >> 2][j0 & 3], wa[j0 >> 2][(j0 & 3) + 1]};
          const f32x2 wb2 = {wb[j0 >> 2][j0 & 3], wb[j0 >> 2][(j0 & 3) + 1]};
          const f32x2 cc2 = {cc[j0 >> 2][j0 & 3], cc[j0 >> 2][(j0 & 3) + 1]};
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const f32x2 xa2 = {xa[nb], xa[nb]}, xb2 = {xb[nb], xb[nb]};
            f32x2 pre = KEEP2 ? __builtin_elementwise_fma(wa2, xa2, __builtin_elementwise_fma(wb2, xb2, cc2))
                              : __builtin_elementwise_fma(wa2, xa2, cc2);
            f16x2 hi, lo;
            if (merged) {                        // abs form (wave-uniform): |pre| here, the linear half in the merged k-step
              split_pair_abs(pre, hi, lo);
            } else {
              pre[0] = fmaxf(pre[0], 0.f);
              pre[1] = fmaxf(pre[1], 0.f);
              split_pair(pre, hi, lo);
            }
            bhi[nb][2 * jp] = hi[0]; bhi[nb][2 * jp + 1] = hi[1];
            blo[nb][2 * jp] = lo[0]; blo[nb][2 * jp + 1] = lo[1];
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            bhi[nb][2 * jp] = bhi[nb][2 * jp + 1] = (_Float16)0.f;
            blo[nb][2 * jp] = blo[nb][2 * jp + 1] = (_Float16)0.f;
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const f16x8 ahi = *reinterpret_cast<const f16x8*>(aimg + ((ks * MB + m) * 2 + 0) * 256);
        const f16x8 alo = *reinterpret_cast<const f16x8*>(aimg + ((ks * MB + m) * 2 + 1) * 256);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi[nb], ks == 0 ? cinit[m] : acc[m][nb], 0, 0, 0);
          acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo[nb], acc[m][nb], 0, 0, 0);
          acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi[nb], acc[m][nb], 0, 0, 0);
        }
      }
    }
}

}  // namespace gwtf_dev
