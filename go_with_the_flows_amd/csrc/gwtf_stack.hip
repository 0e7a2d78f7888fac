// gwtf_stack.hip -- the fused coupling-stack kernel (forward / inverse + log-det), gfx950.
//
// One launch applies all C elementary couplings (reference lib/networks/flows.py:95-117, stacked by
// flows.py:150-160 and decoders.py:61-79) to every point and accumulates sum(logvars)
// (losses.py:14,115).  Coordinates and log-det stay in registers for the whole stack; HBM traffic is
// 12 B/pt in, 24 B/pt out (+36 B/pt per coupling only when the reference's per-layer lists are asked for).
//
// Work decomposition
//   workgroup = 4 waves, one shape b, 4*16*NB consecutive points; wave = 16*NB points.
//   The f x f per-point contraction (sd1) runs on v_mfma_f32_16x16x32_f16 with features on M and points on N,
//   every fp32 operand split into f16 hi + lo and three products accumulated in fp32 (gwtf_layout.h):
//     A[i][k] = W1'[16m+i][feature(k)]   (sd1 weight, sd1_bn scale folded; fragment-ordered images in LDS)
//     B[k][j] = h0[feature(k)][point j]  (sd0 + sd0_bn + ReLU, computed and split in the lane that owns (k,j))
//     D[16m + 4q + r][point j]           (q = lane>>4, r = accumulator register)
//   so ReLU + the f->w contraction (sd2) happen in-lane on the accumulators (FiLM is folded into the
//   accumulator start value and into the sd2 weights by the FiLM kernel), followed by a
//   transpose-reduce over the four lane quarters that leaves quarter q with the totals of point block q.
//   Each lane then owns ONE point for the transcendental tail (softsign, exp, sqrt, affine) and the
//   log-det accumulation; new coordinates are re-broadcast to the quarters with ds_bpermute.
//   Per coupling the packed weights (~2*FP^2 + 6*FP floats) and this shape's FiLM vectors are copied
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a double buffer while the previous coupling
//   computes; one barrier per coupling.
#include "gwtf_device.h"
#include <algorithm>

namespace {

using namespace gwtf_dev;

// The per-point tail (softsign, scale = sqrt(eps + exp(logvar)), inverse affine) on the hardware's reciprocal, square root
// and exp2 units (each within 1-2 ulp) instead of correctly rounded division / sqrtf / libm expf: 2 + 1 + 2 instructions
// against ~11 + ~12 + ~10 per use, 40-75 VALU per coupling and wave.  Operand ranges make the fast forms safe (divisors
// >= ~0.6, logvar in (-1, 1) by the softsign).  Measured against the reference's own fp64 evaluation on the golden decoder
// cases (tools/diag/err_vs_fp64.py): mean coordinate error 3.97e-7 / 3.94e-7 (direct / inverse) against 3.75e-7 / 3.95e-7
// with the correctly rounded forms and 2.81e-7 / 3.36e-7 for the reference's fp32 evaluation itself -- inside the
// noise of fp32 evaluation order, far inside the stated tolerance (2e-5); airplane kernel 0.522 -> 0.498 ms.
__device__ __forceinline__ float tail_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float tail_scale(float eps, float logvar) { return __builtin_amdgcn_sqrtf(eps + __expf(logvar)); }
// INVERSE mode divides by the scale: 1 / sqrt(eps + exp(logvar)) is ONE hardware op (v_rsq_f32) instead of a square root and a
// reciprocal in a row -- one transcendental less on the coupling boundary's dependent chain, and none per coordinate
__device__ __forceinline__ float tail_rscale(float eps, float logvar) { return __builtin_amdgcn_rsqf(eps + __expf(logvar)); }

// Where the pieces of one staged coupling sit in LDS (floats).  Plain: the packed record as it is in global memory
// (gwtf_layout.h GwtfPackW) followed by the shape's FiLM record.  COMPACT (merged widths f = 33..40, MB = 3): the packer leaves
// the lo image of the second k-step unused there (its products ride in the merged hi image A'), so those MB pieces per branch
// are neither copied nor given room:
//     branch br at br * A16:  k-step 0 [m][part] (2 MB images)  |  k-step 1 hi / A' [m] (MB images)
//     SD0 records at 2 * A16  |  FiLM record at FILM (whole 1-KiB DMA pieces before it)
template <int MB, bool COMPACT>
struct LdsLayout {
  using K = Cfg<MB>;
  static constexpr int A16 = COMPACT ? (2 * MB + MB) * 256 : K::A16;
  static constexpr int SD0_BASE = 2 * A16;
  static constexpr int FILM = COMPACT ? (2 * A16 + 2 * K::SD0 + 255) / 256 * 256 : K::PW;
  static constexpr int LAYER = FILM + K::FSP;
  static constexpr int PIECES = FILM / 256;               // 1-KiB DMA pieces of packed weights staged per coupling
  __device__ static constexpr int img(int ks, int m, int part) {   // offset of fragment image (ks, m, part) inside a branch
    return COMPACT ? (ks == 0 ? (m * 2 + part) * 256 : (2 * MB + m) * 256) : ((ks * MB + m) * 2 + part) * 256;
  }
  // destination piece d of the compact layout <- piece of the packed record in global memory
  __device__ static int source_piece(int d) {
    if (!COMPACT) return d;
    constexpr int PB = 3 * MB, GB = 4 * MB;                // pieces per branch: compact, packed record
    if (d >= 2 * PB) return 2 * GB + (d - 2 * PB);          // SD0 records
    const int br = d >= PB ? 1 : 0, r = d - br * PB;
    return br * GB + (r < 2 * MB ? r : 2 * MB + 2 * (r - 2 * MB));
  }
};

// One elementary coupling on the wave's tile.  KEEP2 = two kept coordinates / one warped (patterns 0-2),
// otherwise one kept / two warped (patterns 3-5).  Every VALU instruction here costs the SIMD 4 cycles that
// the matrix pipe cannot use (measured: MFMA and VALU of two waves on one SIMD do not overlap,
// tools/diag/coissue.hip), so the body is specialised to issue as few as possible:
//   - the f x f contraction runs at the f16 matrix rate on split operands (gwtf_layout.h): 3 MFMAs of 16 cycles
//     per 16x16x32 block instead of 8 fp32 MFMAs of 32 cycles, at fp32-grade accuracy;
//   - the accumulators start at c = b'/a (FiLM shift over FiLM scale, a > 0) instead of zero, so
//     relu(a*y + b') * W2  becomes  relu(acc) * (W2*a): one v_max + one v_fma per warped coordinate;
//   - the second sd0 input / second sd2 output only exist in the variant that needs them.
template <int MB, int NB, int MODE, bool KEEP2>
__device__ __forceinline__ void coupling_body(const float* __restrict__ L, int kk_steps, int lane, int q, int k0, int k1,
                                              int w0, int w1, float eps, float s_keep, const float (&x)[NB][3],
                                              float (&xo)[3], float (&mu_d)[3], float (&lv_d)[3]) {
  using K = Cfg<MB>;
  constexpr int FP = K::FP;
  float xa[NB], xb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    xa[nb] = sel3(x[nb][0], x[nb][1], x[nb][2], k0);
    xb[nb] = KEEP2 ? sel3(x[nb][0], x[nb][1], x[nb][2], k1) : 0.f;
  }
  float res[2][2] = {{0.f, 0.f}, {0.f, 0.f}};    // [branch][warped slot] for this lane's own point
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    const float* fe = L + K::PW + br * 3 * FP + 4 * q;   // c | w20a | w21a, this lane's 4 features per block
    f32x4 acc[MB][NB], cinit[MB];   // accumulators start at c: it is the C operand of the first MFMA (no copies)
#pragma unroll
    for (int m = 0; m < MB; ++m) cinit[m] = *reinterpret_cast<const f32x4*>(fe + 16 * m);
    sd1_contract<MB, NB, KEEP2>(L, br, kk_steps, lane, q, xa, xb, cinit, acc);
    // ReLU + sd2 on the accumulators (FiLM already folded in)
    float o0[NB], o1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) o0[nb] = o1[nb] = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(fe + FP + 16 * m);
      f32x4 u1 = {0.f, 0.f, 0.f, 0.f};
      if (!KEEP2) u1 = *reinterpret_cast<const f32x4*>(fe + 2 * FP + 16 * m);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float v = fmaxf(acc[m][nb][r], 0.f);
          o0[nb] = fmaf(u0[r], v, o0[nb]);
          if (!KEEP2) o1[nb] = fmaf(u1[r], v, o1[nb]);
        }
    }
    res[br][0] = quarter_reduce<NB>(o0, q);
    if (!KEEP2) res[br][1] = quarter_reduce<NB>(o1, q);
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(L + K::PW + 6 * FP);

  // transcendental tail on this lane's own point
  const float r_keep = __builtin_amdgcn_rcpf(s_keep);
  float lv_w[2] = {0.f, 0.f}, mu_w[2] = {0.f, 0.f}, sc_w[2] = {s_keep, s_keep};
#pragma unroll
  for (int s = 0; s < (KEEP2 ? 1 : 2); ++s) {
    const float t = res[0][s] + bias[s];
    lv_w[s] = tail_div(t, 1.0f + fabsf(t));          // softsign (flows.py:99)
    mu_w[s] = res[1][s] + bias[2 + s];
    sc_w[s] = MODE == GWTF_MODE_DIRECT ? tail_scale(eps, lv_w[s]) : tail_rscale(eps, lv_w[s]);   // flows.py:113,115 (INVERSE: its reciprocal)
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const bool is0 = d == w0, is1 = !KEEP2 && d == w1;
    lv_d[d] = is0 ? lv_w[0] : (is1 ? lv_w[1] : 0.f);
    mu_d[d] = is0 ? mu_w[0] : (is1 ? mu_w[1] : 0.f);
    const float sc = is0 ? sc_w[0] : (is1 ? sc_w[1] : (MODE == GWTF_MODE_DIRECT ? s_keep : r_keep));
    if (MODE == GWTF_MODE_DIRECT)
      xo[d] = __fadd_rn(__fmul_rn(sc, xo[d]), mu_d[d]);
    else
      xo[d] = __fmul_rn(__fsub_rn(xo[d], mu_d[d]), sc);
  }
}

// Software-pipelined variant of coupling_body for a compile-time number of valid k-slots in the last k-step (NJL):
// without the run-time "skip padded k positions" branches the whole coupling is ONE basic block, and the stages are
// written -- and pinned with sched_group_barrier -- so that vector work sits in the shadow of the matrix pipe: a
// v_mfma_f32_16x16x32_f16 holds the SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md, cycle constants),
// two plain VALU instructions per MFMA are nearly free.
//   A: B fragments (sd0 + ReLU + split) of branch 0                                   VALU only
//   B: MFMAs of branch 0        interleaved with  B fragments of branch 1
//   C: MFMAs of branch 1        interleaved with  ReLU + sd2 dot of branch 0
//   D: ReLU + sd2 dot of branch 1, quarter reduce, transcendental tail                VALU only
// Same arithmetic in the same order per accumulator as coupling_body: bit-identical results.
template <int MB, int NB, int MODE, bool KEEP2, int NJL>
__device__ __forceinline__ void coupling_body_pipe(const float* __restrict__ L, int lane, int q, int k0, int k1, int w0,
                                                   int w1, float eps, float s_keep, const float (&x)[NB][3],
                                                   float (&xo)[3], float (&mu_d)[3], float (&lv_d)[3]) {
  using K = Cfg<MB>;
  constexpr int FP = K::FP, KS = K::KS;
  constexpr int PAIRS_LAST = (NJL + 1) / 2;                       // valid pairs of k-slots in the last k-step
  constexpr int UNITS = ((KS - 1) * 4 + PAIRS_LAST) * NB;         // (pair of k-slots, point block) work items per branch
  constexpr int TRIPLES = KS * MB * NB;                           // (ks, m, nb) MFMA triplets per branch
  float xa[NB], xb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    xa[nb] = sel3(x[nb][0], x[nb][1], x[nb][2], k0);
    xb[nb] = KEEP2 ? sel3(x[nb][0], x[nb][1], x[nb][2], k1) : 0.f;
  }
  // MERGE: a last k-step with at most 2 valid k-slots per lane (f = 33..40) carries its THREE products in ONE MFMA: the 8
  // k-slots of a lane hold B' = [h_hi (2) | h_lo (2) | h_hi (2) | 0 0] against A' = [W_hi | W_hi | W_lo | 0 0] (the hi image
  // of that k-step as gwtf_pack.hip writes it for these widths) -- 4 instead of 6 MFMAs per (m, point block) and 2 instead
  // of 6 filler moves per fragment.  (The legacy K=16 MFMA is no alternative: it costs the same 16 cycles, tools/diag/mfma_k16.hip.)
  constexpr bool MERGE = KS == 2 && NJL <= 2;
  // LDS layout of the staged coupling.  Merged widths (MB = 3): the lo image of the second k-step is never read, so the stack
  // kernel does not stage it (stack_kernel::stage, CompactLds below): 22 KiB per buffer instead of 28 -- THREE workgroups per
  // compute unit instead of two.  Everything else reads the packed record's own layout (Cfg<MB>).
  using LY = LdsLayout<MB, MERGE>;
  // PACK5: a single k-step with 5 valid k-slots per lane (f = 17..20) has 15 products for 16 k-slots of TWO MFMAs:
  //   B1 = [hi01 | hi23 | lo01 | lo23]        against  A1 = [Whi01 | Whi23 | Whi01 | Whi23]
  //   B2 = [(hi4, lo4) | hi01 | hi23 | (hi4, 0)]  against  A2 = [(Whi4, Whi4) | Wlo01 | Wlo23 | (Wlo4, 0)]
  // (held in bhi[..] / blo[..]); 2 instead of 3 MFMAs per (m, point block).
  constexpr bool PACK5 = KS == 1 && NJL == 5;
  constexpr bool PKDOT = true;
  // RV: the last 16-row tile holds at most 4*RV valid output features (f <= 4*(8*(KS-1)+NJL)).  Its rows are dealt to the
  // accumulator TRANSPOSED -- the lane with row slot (q', r) reads the fragment image of row 4r+q' -- so that features
  // 16m+4r+q' land in register r of quarter q' and registers r >= RV are padding in EVERY lane: the epilogue skips them.
  constexpr int RV = (4 * (8 * (KS - 1) + NJL) - 16 * (MB - 1) + 3) / 4 < 4 ? (4 * (8 * (KS - 1) + NJL) - 16 * (MB - 1) + 3) / 4 : 4;
  static_assert(RV >= 1, "the last row tile would be empty");
  const int lane_t = (lane & 48) | ((lane & 3) << 2) | ((lane >> 2) & 3);
  f16x8 bhi[2][KS][NB], blo[2][KS][NB];
  if (!PACK5) {
#pragma unroll
    for (int br = 0; br < 2; ++br)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = MERGE ? 6 : 2 * PAIRS_LAST; e < 8; ++e) bhi[br][KS - 1][nb][e] = blo[br][KS - 1][nb][e] = (_Float16)0.f;
  }

  if (MERGE) {   // abs form: the column slot pair of the merged k-step carries the split kept coordinates (same for both branches)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const f16x2 xs = __builtin_bit_cast(f16x2, abs_form_x_slots(xa[nb], xb[nb], q));
      bhi[0][KS - 1][nb][6] = xs[0]; bhi[0][KS - 1][nb][7] = xs[1];
      bhi[1][KS - 1][nb][6] = xs[0]; bhi[1][KS - 1][nb][7] = xs[1];
    }
  }
  auto unit = [&](int br, int u) {        // u-th (pair, point block) item of branch br
    const int per_ks = 4 * NB;
    const int ks = u / per_ks < KS - 1 ? u / per_ks : KS - 1;
    const int rem = u - ks * per_ks, jp = rem / NB, nb = rem % NB, j0 = 2 * jp;
    const f32x4* sp = reinterpret_cast<const f32x4*>(L + LY::SD0_BASE + br * K::SD0 + q * 24 + ks * 96);
    const f32x4 wa = sp[j0 >> 2], wb = KEEP2 ? sp[2 + (j0 >> 2)] : f32x4{0.f, 0.f, 0.f, 0.f}, cc = sp[4 + (j0 >> 2)];
    const f32x2 wa2 = {wa[j0 & 3], wa[(j0 & 3) + 1]}, wb2 = {wb[j0 & 3], wb[(j0 & 3) + 1]}, cc2 = {cc[j0 & 3], cc[(j0 & 3) + 1]};
    const f32x2 xa2 = {xa[nb], xa[nb]}, xb2 = {xb[nb], xb[nb]};
    if (PACK5 && jp == 2) {                // the odd fifth k-slot: (hi4, lo4) and (hi4, 0)
      const float p4 = fmaxf(KEEP2 ? fmaf(wa2[0], xa[nb], fmaf(wb2[0], xb[nb], cc2[0])) : fmaf(wa2[0], xa[nb], cc2[0]), 0.f);
      const f32x2 p40 = {p4, 0.f};
      const f16x2 y = __builtin_convertvector(p40, f16x2);
      float r4;
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r4) : "v"(__builtin_bit_cast(unsigned, y)), "v"(p4));
      const f32x2 p4r = {p4, r4};
      const f16x2 xx = __builtin_convertvector(p4r, f16x2);
      f16x8& b2 = blo[br][0][nb];
      b2[0] = xx[0]; b2[1] = xx[1]; b2[6] = y[0]; b2[7] = y[1];
      return;
    }
    f32x2 pre = KEEP2 ? __builtin_elementwise_fma(wa2, xa2, __builtin_elementwise_fma(wb2, xb2, cc2))
                      : __builtin_elementwise_fma(wa2, xa2, cc2);
    f16x2 hi, lo;
    if (MERGE) {                           // abs form (gwtf_layout.h): |pre| on the source modifiers, no v_max
      split_pair_abs(pre, hi, lo);
    } else {
      pre[0] = fmaxf(pre[0], 0.f);
      pre[1] = fmaxf(pre[1], 0.f);
      split_pair(pre, hi, lo);
    }
    if (PACK5) {
      f16x8& b1 = bhi[br][0][nb];
      f16x8& b2 = blo[br][0][nb];
      b1[j0] = hi[0]; b1[j0 + 1] = hi[1]; b1[4 + j0] = lo[0]; b1[5 + j0] = lo[1]; b2[2 + j0] = hi[0]; b2[3 + j0] = hi[1];
      return;
    }
    if (MERGE && ks == KS - 1) {
      f16x8& b = bhi[br][ks][nb];
      b[0] = hi[0]; b[1] = hi[1]; b[2] = lo[0]; b[3] = lo[1]; b[4] = hi[0]; b[5] = hi[1];
      return;
    }
    bhi[br][ks][nb][j0] = hi[0]; bhi[br][ks][nb][j0 + 1] = hi[1];
    blo[br][ks][nb][j0] = lo[0]; blo[br][ks][nb][j0 + 1] = lo[1];
  };
  f32x4 acc[2][MB][NB];
  auto triple = [&](int br, int t) {      // t-th (ks, m, nb) triplet of branch br
    const int ks = t / (MB * NB), m = (t / NB) % MB, nb = t % NB;
    const bool tr = RV < 4 && m == MB - 1;                       // transposed row tile
    const float* aimg = L + br * LY::A16 + (tr ? lane_t : lane) * 4;
    const float* cb = L + LY::FILM + br * 3 * FP + 16 * m;
    const f32x4 cinit = tr ? f32x4{cb[q], cb[4 + q], cb[8 + q], cb[12 + q]} : *reinterpret_cast<const f32x4*>(cb + 4 * q);
    if (PACK5) {
      const f32x2 ah = *reinterpret_cast<const f32x2*>(aimg + (m * 2 + 0) * 256);
      const float ah2 = aimg[(m * 2 + 0) * 256 + 2];   // its own load: with ONE b128 load, hipcc 7.2 built a1 in place over element 2 before this use
      const f32x4 al = *reinterpret_cast<const f32x4*>(aimg + (m * 2 + 1) * 256);
      const unsigned w4 = __builtin_bit_cast(unsigned, ah2);                      // (Whi4, 0)
      const unsigned w44 = __builtin_amdgcn_perm(w4, w4, 0x05040100u);            // (Whi4, Whi4)
      const f32x4 a1 = {ah[0], ah[1], ah[0], ah[1]}, a2 = {__builtin_bit_cast(float, w44), al[0], al[1], al[2]};
      acc[br][m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1), bhi[br][0][nb], cinit, 0, 0, 0);
      acc[br][m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2), blo[br][0][nb], acc[br][m][nb], 0, 0, 0);
      return;
    }
    if (MERGE && ks == KS - 1) {
      const f16x8 am = *reinterpret_cast<const f16x8*>(aimg + LY::img(ks, m, 0));   // A', written by the packer
      acc[br][m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bhi[br][ks][nb], acc[br][m][nb], 0, 0, 0);
      return;
    }
    const f16x8 ahi = *reinterpret_cast<const f16x8*>(aimg + LY::img(ks, m, 0));
    const f16x8 alo = *reinterpret_cast<const f16x8*>(aimg + LY::img(ks, m, 1));
    acc[br][m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi[br][ks][nb], ks == 0 ? cinit : acc[br][m][nb], 0, 0, 0);
    acc[br][m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo[br][ks][nb], acc[br][m][nb], 0, 0, 0);
    acc[br][m][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi[br][ks][nb], acc[br][m][nb], 0, 0, 0);
  };
  float o0[2][NB], o1[2][NB];
  f32x2 o01[2][NB];                       // !KEEP2: both warped coordinates' sums as one packed-FMA chain
#pragma unroll
  for (int br = 0; br < 2; ++br)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      o0[br][nb] = o1[br][nb] = 0.f;
      o01[br][nb] = f32x2{0.f, 0.f};
    }
  auto dot_piece = [&](int br, int pc) {  // ReLU + sd2 dot for accumulator row (m, r) and HALF of the point blocks: pc = ((m*4+r)*2+h)
    const int m = pc / 8, r = (pc / 2) % 4, h = pc % 2;
    const bool tr = RV < 4 && m == MB - 1;
    if (tr && r >= RV) return;                                   // padding in every lane
    const float* fe = L + LY::FILM + br * 3 * FP + 16 * m + (tr ? 4 * r + q : 4 * q + r);
    const float u0 = fe[FP], u1 = KEEP2 ? 0.f : fe[2 * FP];
#pragma unroll
    for (int nb = h * (NB / 2 > 0 ? NB / 2 : 1); nb < (NB >= 2 ? (h + 1) * (NB / 2) : (h == 0 ? 1 : 0)); ++nb) {
      const float v = fmaxf(acc[br][m][nb][r], 0.f);
      if (KEEP2) {
        o0[br][nb] = fmaf(u0, v, o0[br][nb]);
      } else if (PKDOT && br == 1) {        // stage D has no MFMAs beside it: only there is v_pk_fma_f32 cheaper than two v_fma (tools/diag/pkfma.hip)
        const f32x2 uu = {u0, u1}, vv = {v, v};
        o01[br][nb] = __builtin_elementwise_fma(uu, vv, o01[br][nb]);
      } else {
        o0[br][nb] = fmaf(u0, v, o0[br][nb]);
        o1[br][nb] = fmaf(u1, v, o1[br][nb]);
      }
    }
  };

  // Two refinements measured at noise level (0.604-0.609 ms against 0.606 for the airplane config) and left off: computing
  // only the first k-step's fragments of branch 0 up front, and starting branch 1's dot tile by tile inside stage C.
  constexpr bool LATEFRAG = false, EARLYDOT = false;
  // stage A: fragments of branch 0 (LATEFRAG: for its FIRST k-step only, the rest rides behind its own MFMAs)
  constexpr int U0 = LATEFRAG ? 4 * NB : UNITS;                    // items computed up front
  constexpr int T0 = MB * NB;                                      // triplets of one k-step
#pragma unroll
  for (int u = 0; u < U0; ++u) unit(0, u);
  // stage B: MFMAs of branch 0; behind them, in this order: the rest of branch 0's fragments, then branch 1's
  constexpr int QB = (UNITS - U0) + UNITS;
  constexpr int PIECES = MB * 8;
#pragma unroll
  for (int t = 0; t < TRIPLES; ++t) {
    triple(0, t);
#pragma unroll
    for (int i = t * QB / TRIPLES; i < (t + 1) * QB / TRIPLES; ++i) {
      // a fragment must precede its first triplet in program order: branch 0's late items are all issued within the
      // first k-step's T0 triplets whenever (UNITS - U0) * TRIPLES <= T0 * QB (asserted)
      if (i < UNITS - U0) unit(0, U0 + i); else unit(1, i - (UNITS - U0));
    }
  }
  static_assert(!LATEFRAG || (UNITS - U0) * TRIPLES <= T0 * QB, "late fragments of branch 0 would miss their k-step");
  static_assert(KS <= 2, "the fragment queue assumes at most two k-steps");
  // stage C: MFMAs of branch 1; behind them the ReLU + sd2 dot of branch 0 (EARLYDOT: then, as soon as a tile of
  // branch 1 has seen its last k-step, that tile's own dot)
  constexpr int TC = EARLYDOT ? (KS - 1) * T0 : TRIPLES;            // triplets that carry branch 0's dot
#pragma unroll
  for (int t = 0; t < TRIPLES; ++t) {
    triple(1, t);
    if (t < TC) {
#pragma unroll
      for (int pc = t * PIECES / TC; pc < (t + 1) * PIECES / TC; ++pc) dot_piece(0, pc);
    } else {
      const int tl = t - TC, m_done = tl / NB;                      // tiles 0 .. m_done-1 of branch 1 are complete
      if (m_done > 0) {
#pragma unroll
        for (int i = (tl % NB) * 8 / NB; i < ((tl % NB) + 1) * 8 / NB; ++i) dot_piece(1, (m_done - 1) * 8 + i);
      }
    }
  }
  // stage D: what is left of branch 1's dot
#pragma unroll
  for (int pc = (EARLYDOT ? (MB - 1) * 8 : 0); pc < PIECES; ++pc) dot_piece(1, pc);
  if (RV < 4) {
    // The skipped registers of the transposed tiles are dead MFMA outputs: keep the whole tuples allocated until here, or
    // the register allocator hands them to other values while the MFMA that still writes them is in flight.
#pragma unroll
    for (int br = 0; br < 2; ++br)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(acc[br][MB - 1][nb]));
  }
  float res[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    if (!KEEP2 && PKDOT && br == 1) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) { o0[br][nb] = o01[br][nb][0]; o1[br][nb] = o01[br][nb][1]; }
    }
    res[br][0] = quarter_reduce<NB>(o0[br], q);
    if (!KEEP2) res[br][1] = quarter_reduce<NB>(o1[br], q);
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(L + LY::FILM + 6 * FP);
  const float r_keep = __builtin_amdgcn_rcpf(s_keep);
  float lv_w[2] = {0.f, 0.f}, mu_w[2] = {0.f, 0.f}, sc_w[2] = {s_keep, s_keep};
#pragma unroll
  for (int s = 0; s < (KEEP2 ? 1 : 2); ++s) {
    const float t = res[0][s] + bias[s];
    lv_w[s] = tail_div(t, 1.0f + fabsf(t));
    mu_w[s] = res[1][s] + bias[2 + s];
    sc_w[s] = MODE == GWTF_MODE_DIRECT ? tail_scale(eps, lv_w[s]) : tail_rscale(eps, lv_w[s]);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const bool is0 = d == w0, is1 = !KEEP2 && d == w1;
    lv_d[d] = is0 ? lv_w[0] : (is1 ? lv_w[1] : 0.f);
    mu_d[d] = is0 ? mu_w[0] : (is1 ? mu_w[1] : 0.f);
    const float sc = is0 ? sc_w[0] : (is1 ? sc_w[1] : (MODE == GWTF_MODE_DIRECT ? s_keep : r_keep));
    if (MODE == GWTF_MODE_DIRECT)
      xo[d] = __fadd_rn(__fmul_rn(sc, xo[d]), mu_d[d]);
    else
      xo[d] = __fmul_rn(__fsub_rn(xo[d], mu_d[d]), sc);
  }
}

// Work list of one launch: component k applies ITS stack to points [begin[k], end[k]) of every shape.
// Passed by value in the kernel arguments (wave-uniform, scalar loads).
// Train-mode statistics pass of ONE coupling: per-feature sum and sum of squares of y1 = sd1(relu(sd0_bn(sd0(x))))
// over all points, both branches (input of sd1_bn's batch statistics, reference flows.py:30,65 in train()).
// Same tile decomposition and contraction as the forward kernel; the accumulators are reduced over the wave's
// points (in-lane over the NB blocks, shuffles over the 16 lanes of a quarter) and added to ystats with one
// atomic per feature and workgroup.   ystats: [GWTF_STAT_REPLICAS][2 branches][FP][2] = {sum y, sum y^2}
template <int MB, int NB, int MG = -1>
__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ p, const float* __restrict__ pw_c,
                                                    float* __restrict__ ystats, int B, int N, int pat, int kk_steps,
                                                    size_t p_sk, size_t pw_sk, size_t ys_sk) {
  using K = Cfg<MB>;
  constexpr int FP = K::FP;
  p += blockIdx.y * p_sk;          // blockIdx.y = mixture component (K-batched train pipeline); strides 0 for a single stack
  pw_c += blockIdx.y * pw_sk;
  ystats += blockIdx.y * ys_sk;
  __shared__ __align__(16) float lds[K::PW];
  constexpr int SW = (GWTF_ROWSUM_KEEP >= 2 && GWTF_ROWSUM_MODE == 1) ? 4 : 1;      // per-wave slots (plain stores) or one set (atomics)
  __shared__ float s_stats[SW][2 * FP * 2];
  for (int t = threadIdx.x; t < SW * 2 * FP * 2; t += blockDim.x) (&s_stats[0][0])[t] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, i16 = lane & 15;
  const int tiles_per_shape = (N + 64 * NB - 1) / (64 * NB);
  const int b = blockIdx.x / tiles_per_shape;
  const int tile = blockIdx.x - b * tiles_per_shape;
  const int n_wave0 = (tile * 4 + wave) * 16 * NB;
#pragma unroll
  for (int i = 0; i < (K::PW / 256 + 3) / 4; ++i) {
    const int piece = wave + 4 * i;
    if (piece < K::PW / 256)
      __builtin_amdgcn_global_load_lds((glb_void*)(pw_c + piece * 256 + lane * 4), (lds_void*)&lds[piece * 256], 16, 0, 0);
  }
  int k0, k1, w0, w1;
  gwtf_pattern_dims(pat, &k0, &k1, &w0, &w1);
  float xa[NB], xb[NB];
  bool valid[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n_wave0 + 16 * nb + i16;
    valid[nb] = n < N;
    xa[nb] = valid[nb] ? p[((size_t)b * 3 + k0) * N + n] : 0.f;
    xb[nb] = (valid[nb] && k1 >= 0) ? p[((size_t)b * 3 + k1) * N + n] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    f32x4 acc[MB][NB], cinit[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) cinit[m] = zero4;
    if (pat < 3) sd1_contract<MB, NB, true, MG>(lds, br, kk_steps, lane, q, xa, xb, cinit, acc);
    else sd1_contract<MB, NB, false, MG>(lds, br, kk_steps, lane, q, xa, xb, cinit, acc);
    float* sw = s_stats[SW == 4 ? wave : 0];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float v = valid[nb] ? acc[m][nb][r] : 0.f;   // points beyond N carry sd0's bias term: exclude
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
        s1 = row_sum_part(s1);
        s2 = row_sum_part(s2);
        if (row_sum_owner(i16)) {
          const int ft = 16 * m + 4 * q + r;
          if (SW == 4) {
            *reinterpret_cast<float2*>(&sw[(br * FP + ft) * 2]) = make_float2(s1, s2);
          } else {
            atomicAdd(&sw[(br * FP + ft) * 2 + 0], s1);   // LDS: 4 waves
            atomicAdd(&sw[(br * FP + ft) * 2 + 1], s2);
          }
        }
      }
  }
  __syncthreads();
  // one global atomic per value and workgroup, spread over GWTF_STAT_REPLICAS copies: thousands of adds to one
  // address serialise (measured 580 us for this kernel with per-wave atomics on a single copy)
  float* rep = ystats + (size_t)(blockIdx.x % GWTF_STAT_REPLICAS) * (2 * FP * 2);
  for (int t = threadIdx.x; t < 2 * FP * 2; t += blockDim.x) {
    float v = s_stats[0][t];
#pragma unroll
    for (int w = 1; w < SW; ++w) v += s_stats[w][t];
    atomicAdd(&rep[t], v);
  }
}

// Optional per-launch extras (all zero for the plain eval forward): a sub-range of couplings, a log-det to
// continue from, and the coordinate moments the train-mode BatchNorm of the NEXT coupling needs.
struct Extras {
  int c_first, c_count;          // couplings processed: c_first, c_first +/- 1, ... (c_count == 0: all C)
  const float* logdet_in;        // [B][3][N] or null (component k: + k * out_stride_k)
  float* moments_out;            // [GWTF_STAT_REPLICAS][16], 9 used {Sx0,Sx1,Sx2,Sx0x0,Sx0x1,Sx0x2,Sx1x1,Sx1x2,Sx2x2}, or null
  size_t moments_stride_k;       // component k accumulates into moments_out + k * moments_stride_k
  int tpw;                       // one-coupling launches: consecutive tiles of ONE shape a workgroup walks (0 / 1 = one tile)
  int* worklist;                 // null, or the flagged-wave list a gwtf_stack_rerun_flagged launch reads (include/gwtf.h)
};

struct Jobs {
  int K;
  int tiles_cum[GWTF_MAX_COMPONENTS + 1];  // prefix sum of B * tiles(k)
  int begin[GWTF_MAX_COMPONENTS], end[GWTF_MAX_COMPONENTS];
};

template <int MB, int NB, int MODE, bool LISTS, int NJL = 0>
__global__ __launch_bounds__(256) void stack_kernel(const float* __restrict__ p, const float* __restrict__ pw,
                                                    const float* __restrict__ film, float* __restrict__ out,
                                                    float* __restrict__ logdet, float* __restrict__ ps,
                                                    float* __restrict__ mus, float* __restrict__ lvs, int B, int N, int C,
                                                    int pattern0, float eps, int kk_steps, const Jobs jobs,
                                                    size_t p_stride_k, size_t out_stride_k, const Extras ex) {
  using K = Cfg<MB>;
  // the pipelined body of the merged widths (MB = 3, NJL = 1 / 2: f = 33..40) reads the compact LDS layout (LdsLayout above):
  // 45 KiB of double buffer per workgroup -> three workgroups per compute unit
  using LY = LdsLayout<MB, (NJL > 0 && K::KS == 2 && NJL <= 2)>;
  // double buffer while two layers fit the 160 KiB of LDS (f <= 96); wider stacks (f = 97..128: 121 / 138 KiB per layer) stage
  // a coupling, compute, and only then stage the next (the weight DMA is exposed: the widths no shipped config uses)
  constexpr int NBUF = 2 * LY::LAYER * 4 <= 160 * 1024 - 512 ? 2 : 1;
  __shared__ __align__(16) float lds[NBUF][LY::LAYER];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, i16 = lane & 15;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one L2), so in launch
  // order every XCD's 4 MiB L2 sees EVERY component's 0.9 MB of weights and every shape's FiLM records: 112 MB of L2 misses
  // per airplane launch against 1.6 MB of input (profiles/r06).  The bijective remap below (cdna_hip_programming.md T1)
  // gives XCD x the contiguous slice [x G/8, (x+1) G/8) of the tile list -- one component (K = 4: two XCDs each) and one
  // eighth of the shapes per XCD.  Speed only: nothing depends on the placement.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  }
  int comp = 0;
  while (comp + 1 < jobs.K && bid >= jobs.tiles_cum[comp + 1]) ++comp;
  const int n_begin = jobs.begin[comp], n_end = jobs.end[comp];
  const int tiles_per_shape = (n_end - n_begin + 64 * NB - 1) / (64 * NB);
  // One-coupling launches of the train pipeline (ex.tpw > 1): a workgroup walks tpw consecutive tiles of ONE shape, so the
  // coupling's weights and the shape's FiLM record are staged once per workgroup instead of once per tile -- such a launch is
  // 45 % prologue (31 us against a 14 us serial-sum floor, docs/LOG.md).  Whole-stack launches: one tile per workgroup.
  const int tpw = ex.tpw > 1 ? ex.tpw : 1;
  const int groups = (tiles_per_shape + tpw - 1) / tpw;
  const int local = bid - jobs.tiles_cum[comp];
  const int b = local / groups;
  const int tile0 = (local - b * groups) * tpw;
  const int own_nb = q & (NB - 1);
  p += comp * p_stride_k;
  out += comp * out_stride_k;
  logdet += comp * out_stride_k;
  if (LISTS) {
    const size_t ls = out_stride_k ? (size_t)C * B * 3 * N : 0;
    ps += comp * ls;
    if (mus) mus += comp * ls;      // the train pipeline's fused consumers keep ps only (the backward's inputs): mus / lvs may be null
    if (lvs) lvs += comp * ls;
  }
  pw += (size_t)comp * C * K::PW;
  const int KC = jobs.K * C;     // FiLM records per shape: [b][component][coupling]

  // global -> LDS staging of one coupling (LDS-DMA, 1 KiB per wave-instruction): whole pieces of packed
  // weights, then this shape's FiLM record (its last piece is partial)
  auto stage = [&](int buf, int c) {
    // wave-uniform base + one 32-bit per-lane byte offset: the global_load_lds address is (SGPR pair + VGPR offset), no 64-bit
    // vector address arithmetic per piece
    const float* src_w = pw + (size_t)c * K::PW;
    const float* src_f = film + ((size_t)b * KC + (size_t)comp * C + c) * K::FS;
    const unsigned voff = lane * 16u;
#pragma unroll
    for (int i = 0; i < (LY::PIECES + 3) / 4; ++i) {
      const int piece = wave + 4 * i;                        // destination piece; its source piece skips what is not staged
      if (piece < LY::PIECES)
        __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(src_w + LY::source_piece(piece) * 256) + voff),
                                         (lds_void*)&lds[buf][piece * 256], 16, 0, 0);
    }
    if (wave < K::FSP / 256 && wave * 256 + lane * 4 < K::FS)
      __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(src_f + wave * 256) + voff),
                                       (lds_void*)&lds[buf][LY::FILM + wave * 256], 16, 0, 0);
  };

  const int n_steps = ex.c_count > 0 ? ex.c_count : C;
  const int c_start = ex.c_count > 0 ? ex.c_first : (MODE == GWTF_MODE_INVERSE ? C - 1 : 0);
  stage(0, c_start);
  const float s_keep = sqrtf(eps + 1.0f);  // scale applied to un-warped coordinates (reference quirk)
  float macc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // coordinate moments of this lane's points (ex.moments_out)
  for (int it = 0; it < tpw; ++it) {       // tpw > 1 only with n_steps == 1: nothing is staged inside the loop
  const int tile = tile0 + it;
  if (tile >= tiles_per_shape) break;
  const int n_wave0 = n_begin + (tile * 4 + wave) * 16 * NB;
  const int n_own = n_wave0 + 16 * own_nb + i16;
  const bool own_valid = n_own < n_end && q < NB;  // q >= NB holds duplicates of quarter q & (NB-1)
  const bool own_inrange = n_own < n_end;
  // this lane's own point (one per lane) and the per-quarter copies used to build the MFMA B operand
  float xo[3], ld[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    xo[d] = own_inrange ? p[((size_t)b * 3 + d) * N + n_own] : 0.f;
    if (ex.logdet_in && own_inrange) ld[d] = ex.logdet_in[comp * out_stride_k + ((size_t)b * 3 + d) * N + n_own];
  }
  float x[NB][3];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int d = 0; d < 3; ++d) x[nb][d] = __shfl(xo[d], 16 * nb + i16);

  // largest |coordinate| this lane's point had at the input of any coupling: beyond GWTF_X_LIMIT the f16 image of sd0's
  // activations can overflow (gwtf_layout.h, range scaling) and the ReLU's v_max would turn the resulting NaN accumulators
  // into zeros -- such a point is flagged instead (NaN result below).  v_max ignores NaN operands: a NaN coordinate is
  // caught by the bit test at the end (NaN and Inf coordinates stay non-finite through every later coupling).
  float xmax = fmaxf(fabsf(xo[0]), fmaxf(fabsf(xo[1]), fabsf(xo[2])));
  float mu_last[3] = {0.f, 0.f, 0.f}, lv_last[3] = {0.f, 0.f, 0.f};

  for (int step = 0; step < n_steps; ++step) {
    const int c = MODE == GWTF_MODE_INVERSE ? c_start - step : c_start + step;
    const int buf = NBUF == 2 ? step & 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NBUF == 2 && step + 1 < n_steps) stage(buf ^ 1, MODE == GWTF_MODE_INVERSE ? c - 1 : c + 1);

    const int pat = (pattern0 + c) % 6;
    int k0, k1, w0, w1;
    gwtf_pattern_dims(pat, &k0, &k1, &w0, &w1);
    float mu_d[3], lv_d[3];
    if constexpr (NJL > 0) {
      if (pat < 3)
        coupling_body_pipe<MB, NB, MODE, true, NJL>(lds[buf], lane, q, k0, k1, w0, w1, eps, s_keep, x, xo, mu_d, lv_d);
      else
        coupling_body_pipe<MB, NB, MODE, false, NJL>(lds[buf], lane, q, k0, k1, w0, w1, eps, s_keep, x, xo, mu_d, lv_d);
    } else {
      if (pat < 3)
        coupling_body<MB, NB, MODE, true>(lds[buf], kk_steps, lane, q, k0, k1, w0, w1, eps, s_keep, x, xo, mu_d, lv_d);
      else
        coupling_body<MB, NB, MODE, false>(lds[buf], kk_steps, lane, q, k0, k1, w0, w1, eps, s_keep, x, xo, mu_d, lv_d);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) ld[d] += lv_d[d];
    if (step + 1 < n_steps) xmax = fmaxf(xmax, fmaxf(fabsf(xo[0]), fmaxf(fabsf(xo[1]), fabsf(xo[2]))));
    if (LISTS && step + 1 == n_steps) {      // the last slot is written after the non-finite / range check below
#pragma unroll
      for (int d = 0; d < 3; ++d) { mu_last[d] = mu_d[d]; lv_last[d] = lv_d[d]; }
    } else if (LISTS && own_valid) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t o = (((size_t)c * B + b) * 3 + d) * N + n_own;
        ps[o] = xo[d];
        if (mus) mus[o] = mu_d[d];
        if (lvs) lvs[o] = lv_d[d];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int d = 0; d < 3; ++d) x[nb][d] = __shfl(xo[d], 16 * nb + i16);
    if (NBUF == 1 && step + 1 < n_steps) {
      __syncthreads();                       // every wave is done with this coupling's weights
      stage(0, MODE == GWTF_MODE_INVERSE ? c - 1 : c + 1);
    }
  }

  {
    // non-finite or out-of-range point -> NaN coordinates AND log-det (integer tests: the library is built -fno-honor-nans)
    unsigned worst = gwtf_float_bits(xmax);                   // magnitudes compare like their bit patterns (sign cleared)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      worst = max(worst, gwtf_float_bits(xo[d]) & 0x7fffffffu);
      const unsigned lb = gwtf_float_bits(ld[d]) & 0x7fffffffu;
      worst = max(worst, lb >= 0x7f800000u ? lb : 0u);          // the log-det only when it is itself Inf / NaN
    }
    const bool bad = worst > __builtin_bit_cast(unsigned, GWTF_X_LIMIT);
    if (bad) {
      const float qnan = __builtin_bit_cast(float, 0x7fc00000u);
#pragma unroll
      for (int d = 0; d < 3; ++d) xo[d] = ld[d] = mu_last[d] = lv_last[d] = qnan;
    }
    // (rare) a wave with a flagged point leaves {shape slot, first point} in the work list of the exact re-run launch that follows
    if (ex.worklist && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) {
      const int idx = atomicAdd(&ex.worklist[0], 1);
      if (idx < GWTF_WORKLIST_CAP) {
        ex.worklist[2 + 2 * idx] = comp * B + b;
        ex.worklist[3 + 2 * idx] = n_own;
      }
    }
  }
  if (LISTS && own_valid) {
    const int c_last = MODE == GWTF_MODE_INVERSE ? c_start - (n_steps - 1) : c_start + (n_steps - 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const size_t o = (((size_t)c_last * B + b) * 3 + d) * N + n_own;
      ps[o] = xo[d];
      if (mus) mus[o] = mu_last[d];
      if (lvs) lvs[o] = lv_last[d];
    }
  }
  if (own_valid) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const size_t o = ((size_t)b * 3 + d) * N + n_own;
      out[o] = xo[d];
      logdet[o] = ld[d];
    }
  }
  if (ex.moments_out) {
    // first and second moments of the output coordinates over all points (input statistics of the next coupling's sd0_bn in train
    // mode): every lane adds its own point's nine products up over the workgroup's tiles; ONE wavefront reduction per workgroup below
    // (per tile it was nine 6-step reductions + a barrier: ~110 of the ~920 VALU of a one-coupling tile)
    const float m0 = own_valid ? xo[0] : 0.f, m1 = own_valid ? xo[1] : 0.f, m2 = own_valid ? xo[2] : 0.f;
    macc[0] += m0; macc[1] += m1; macc[2] += m2;
    macc[3] = fmaf(m0, m0, macc[3]); macc[4] = fmaf(m0, m1, macc[4]); macc[5] = fmaf(m0, m2, macc[5]);
    macc[6] = fmaf(m1, m1, macc[6]); macc[7] = fmaf(m1, m2, macc[7]); macc[8] = fmaf(m2, m2, macc[8]);
  }
  }   // tiles of this workgroup
  if (ex.moments_out) {
#pragma unroll
    for (int i = 0; i < 9; ++i) macc[i] = wave_sum(macc[i]);
    __shared__ float s_mom[4][9];
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) s_mom[wave][i] = macc[i];
    }
    __syncthreads();
    if (threadIdx.x < 9)
      atomicAdd(&ex.moments_out[comp * ex.moments_stride_k + (blockIdx.x % GWTF_STAT_REPLICAS) * 16 + threadIdx.x],
                s_mom[0][threadIdx.x] + s_mom[1][threadIdx.x] + s_mom[2][threadIdx.x] + s_mom[3][threadIdx.x]);
  }
}

template <int MB, int NB>
int launch(const float* p, const float* pw, const float* film, float* out, float* logdet, float* ps, float* mus,
           float* lvs, int B, int N, int C, int pattern0, float eps, int mode, int kk_steps, const int* segs, int K,
           size_t p_stride_k, size_t out_stride_k, const Extras& ex, bool pipe, hipStream_t st) {
  Jobs jobs;
  jobs.K = K;
  jobs.tiles_cum[0] = 0;
  for (int k = 0; k < K; ++k) {
    jobs.begin[k] = segs ? segs[2 * k] : 0;
    jobs.end[k] = segs ? segs[2 * k + 1] : N;
    const int cnt = jobs.end[k] - jobs.begin[k];
    const int tpw = ex.tpw > 1 ? ex.tpw : 1;
    jobs.tiles_cum[k + 1] = jobs.tiles_cum[k] + B * (((cnt + 64 * NB - 1) / (64 * NB) + tpw - 1) / tpw);
  }
  if (jobs.tiles_cum[K] == 0) return 0;
  const dim3 grid((unsigned)jobs.tiles_cum[K]), block(256);
  const bool lists = ps != nullptr;
#define GWTF_LAUNCH(MODE_, LISTS_)                                                                                      \
  hipLaunchKernelGGL((stack_kernel<MB, NB, MODE_, LISTS_>), grid, block, 0, st, p, pw, film, out, logdet, ps, mus, lvs, \
                     B, N, C, pattern0, eps, kk_steps, jobs, p_stride_k, out_stride_k, ex)
  // Software-pipelined body (coupling_body_pipe) for the widths the reference's configs resolve to -- the number of valid
  // k-slots of the last k-step must be a compile-time fact: f = 61..64 (NJL 8), 37..40 (2), 33..36 (1), 17..20 (5) -- and
  // for the tiles whose fragments + both branches' accumulators fit 256 VGPRs.
  const int njl = kk_steps - 8 * (Cfg<MB>::KS - 1);
#define GWTF_PIPED(NJL_)                                                                                                    \
  if (njl == NJL_) {                                                                                                        \
    if (mode == GWTF_MODE_DIRECT) {                                                                                         \
      if (lists) hipLaunchKernelGGL((stack_kernel<MB, NB, GWTF_MODE_DIRECT, true, NJL_>), grid, block, 0, st, p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, kk_steps, jobs, p_stride_k, out_stride_k, ex);   \
      else hipLaunchKernelGGL((stack_kernel<MB, NB, GWTF_MODE_DIRECT, false, NJL_>), grid, block, 0, st, p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, kk_steps, jobs, p_stride_k, out_stride_k, ex);        \
    } else {                                                                                                                \
      if (lists) hipLaunchKernelGGL((stack_kernel<MB, NB, GWTF_MODE_INVERSE, true, NJL_>), grid, block, 0, st, p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, kk_steps, jobs, p_stride_k, out_stride_k, ex);  \
      else hipLaunchKernelGGL((stack_kernel<MB, NB, GWTF_MODE_INVERSE, false, NJL_>), grid, block, 0, st, p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, kk_steps, jobs, p_stride_k, out_stride_k, ex);       \
    }                                                                                                                       \
    return (int)hipGetLastError();                                                                                          \
  }
  if (pipe) {
    if constexpr (MB == 4 && NB <= 2) { GWTF_PIPED(8) }
    if constexpr (MB == 3) { GWTF_PIPED(2) GWTF_PIPED(1) }
    if constexpr (MB == 2) { GWTF_PIPED(5) }
  }
#undef GWTF_PIPED
  if (mode == GWTF_MODE_DIRECT) {
    if (lists) GWTF_LAUNCH(GWTF_MODE_DIRECT, true); else GWTF_LAUNCH(GWTF_MODE_DIRECT, false);
  } else {
    if (lists) GWTF_LAUNCH(GWTF_MODE_INVERSE, true); else GWTF_LAUNCH(GWTF_MODE_INVERSE, false);
  }
#undef GWTF_LAUNCH
  return (int)hipGetLastError();
}

template <int MB>
int launch_nb(int nb, const float* p, const float* pw, const float* film, float* out, float* logdet, float* ps,
              float* mus, float* lvs, int B, int N, int C, int pattern0, float eps, int mode, int kk_steps,
              const int* segs, int K, size_t p_stride_k, size_t out_stride_k, const Extras& ex, bool pipe, hipStream_t st) {
  if constexpr (MB > 4) {      // f > 64: one workgroup per compute unit anyway (LDS); 16 or 32 points per wave keep the accumulators in registers
    if (nb == 1) return launch<MB, 1>(p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, mode, kk_steps, segs, K, p_stride_k, out_stride_k, ex, pipe, st);
    return launch<MB, 2>(p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, mode, kk_steps, segs, K, p_stride_k, out_stride_k, ex, pipe, st);
  } else {
  switch (nb) {
    case 1: return launch<MB, 1>(p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, mode, kk_steps, segs, K, p_stride_k, out_stride_k, ex, pipe, st);
    case 2: return launch<MB, 2>(p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, mode, kk_steps, segs, K, p_stride_k, out_stride_k, ex, pipe, st);
    default: return launch<MB, 4>(p, pw, film, out, logdet, ps, mus, lvs, B, N, C, pattern0, eps, mode, kk_steps, segs, K, p_stride_k, out_stride_k, ex, pipe, st);
  }
  }
}

// ---- tile choice -------------------------------------------------------------------------------------------------------------
// A workgroup carries 64 * NB points of one (shape, component).  The kernel is latency-bound while a compute unit holds one
// workgroup and issue-bound from two on, so what a launch costs is a matter of ROUNDS.  Measured on the MI355X
// (tools/diag/tile_rounds.py, profiles/r13_tile_rounds.txt: one component x 33 couplings, microseconds): with W workgroups a
// compute unit holds ceil(W / 256) of them up to its limit S (LDS / VGPRs: 2 for the widest tiles, 3-4 below), and
//     time(NB, W) = (W div 256 S) * level[S] + level[ceil((W mod 256 S) / 256)]
// reproduces the measurements to +-8 %.  Small tiles are cheap per round but dear per POINT (f = 33: 16 points per wave move a
// quarter of the points of 64 per wave for 45 % of its round), so the chooser prices every tile and takes the cheapest; the rule
// it replaces looked at the point count alone (16 points per wave for everything below 64 K points: 93 us against 78 for
// 24 x 2048 points at f = 33).  What no tile can fix is quantisation: 640 large workgroups (the SVR shard) are 1.25 rounds of
// 512 and cost 1.45 rounds, and a second launch with small tiles for the remainder cannot overlap the first.
struct TileCost { int slots; float level[4]; };     // cost of a round with 1 .. slots workgroups per compute unit
inline TileCost tile_cost(int MB, int nbi /* 0 / 1 / 2 = NB 1 / 2 / 4 */) {
  static const TileCost T[4][3] = {
      /* MB 1 (f <= 16) */ {{4, {31.f, 38.f, 46.f, 55.f}}, {4, {39.f, 49.f, 61.f, 75.f}}, {4, {52.f, 72.f, 96.f, 119.f}}},
      /* MB 2 (f <= 32) */ {{4, {36.f, 44.f, 54.f, 66.f}}, {4, {40.f, 54.f, 69.f, 87.f}}, {4, {54.f, 79.f, 112.f, 148.f}}},
      /* MB 3 (f <= 48) */ {{3, {45.f, 60.f, 79.f, 0.f}}, {3, {55.f, 82.f, 116.f, 0.f}}, {2, {82.f, 140.f, 0.f, 0.f}}},
      /* MB 4 (f <= 64) */ {{2, {60.f, 84.f, 0.f, 0.f}}, {2, {85.f, 133.f, 0.f, 0.f}}, {2, {165.f, 249.f, 0.f, 0.f}}}};
  return T[MB < 1 ? 0 : (MB > 4 ? 3 : MB - 1)][nbi];   // MB > 4: NB <= 2 only, same ratios as MB 4
}
inline float launch_cost(const TileCost& tc, long W) {
  if (W <= 0) return 0.f;
  const long round = 256L * tc.slots, full = W / round, rem = W % round;
  return (float)full * tc.level[tc.slots - 1] + (rem == 0 ? 0.f : tc.level[(rem + 255) / 256 - 1]);
}
// workgroups of a launch at NB over K segments of B shapes
inline long tiles_of(const int* segments, int K, int B, int N, int nb) {
  long W = 0;
  for (int k = 0; k < K; ++k) {
    const int cnt = segments ? segments[2 * k + 1] - segments[2 * k] : N;
    W += (long)B * ((cnt + 64 * nb - 1) / (64 * nb));
  }
  return W;
}
inline int choose_nb(const int* segments, int K, int B, int N, int f, int tune) {
  const int forced = (tune & 0xffff) / 16;
  if (forced == 1 || forced == 2 || forced == 4) return (f > 64 && forced > 2) ? 2 : forced;
  const int MB = gwtf_padded_width(f) / 16;
  int best = 1;
  float best_cost = 0.f;
  for (int nbi = (f > 64 ? 1 : 2); nbi >= 0; --nbi) {          // larger tiles first: they win ties (fewer workgroups)
    const int nb = 1 << nbi;
    const float cst = launch_cost(tile_cost(MB, nbi), tiles_of(segments, K, B, N, nb));
    if (nbi == (f > 64 ? 1 : 2) || cst < 0.97f * best_cost) { best = nb; best_cost = cst; }
  }
  return best;
}

}  // namespace

extern "C" int gwtf_stack_plan(const int* segments, int K, int B, int N, int f, int tune, int* out4) {
  if (!out4 || K <= 0 || K > GWTF_MAX_COMPONENTS || B <= 0 || N <= 0 || f <= 0 || f > GWTF_MAX_FP) return GWTF_E_BADARG;
  const int nb = choose_nb(segments, K, B, N, f, tune);
  out4[0] = 16 * nb;
  out4[1] = (int)tiles_of(segments, K, B, N, nb);
  out4[2] = out4[3] = 0;              // no tail launch: measured, a second (small-tile) launch cannot overlap the first -- DESIGN.md 3.1
  return 0;
}

static int stack_dispatch(const float* p, const float* packed_w, const float* film, float* out, float* logdet, float* ps,
                          float* mus, float* logvars, const int* segments, int K, int B, int N, int C, int f,
                          int pattern0, float eps, int mode, size_t p_stride_k, size_t out_stride_k, const Extras& ex_in,
                          int tune, void* stream) {
  if (B <= 0 || N <= 0 || C <= 0 || f <= 0 || f > GWTF_MAX_FP || K <= 0 || K > GWTF_MAX_COMPONENTS || !p ||
      !packed_w || !film || !out || !logdet)
    return GWTF_E_BADARG;
  if (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE) return GWTF_E_BADARG;
  if (pattern0 < 0 || pattern0 > 5) return GWTF_E_BADARG;
  if (ex_in.c_count < 0 || (ex_in.c_count > 0 && (ex_in.c_first < 0 || ex_in.c_first >= C))) return GWTF_E_BADARG;
  const bool any = ps || mus || logvars, all = ps && mus && logvars, ps_only = ps && !mus && !logvars;
  if (any && !all && !ps_only) return GWTF_E_BADARG;     // the three lists, none, or the coordinates alone (train pipeline)
  long pts = 0;
  for (int k = 0; k < K; ++k) {
    const int b0 = segments ? segments[2 * k] : 0, e0 = segments ? segments[2 * k + 1] : N;
    if (b0 < 0 || e0 < b0 || e0 > N) return GWTF_E_BADARG;
    pts += (long)B * (e0 - b0);
  }
  hipStream_t st = (hipStream_t)stream;
  // points per wave.  Whole-stack launches: the tile whose rounds x round cost is smallest (choose_nb).  One-coupling launches of
  // the train pipeline (ex.c_count == 1) are latency-bound per launch whatever the tile: the largest tile that still gives each
  // of the 1024 SIMDs two waves.
  int nb;
  const int forced = (tune & 0xffff) / 16;
  if (ex_in.c_count == 0 || forced == 1 || forced == 2 || forced == 4) {
    nb = choose_nb(segments, K, B, N, f, tune);
  } else {
    nb = pts >= 2048L * 64 ? 4 : (pts >= 2048L * 32 ? 2 : 1);
    if (f > 64 && nb > 2) nb = 2;
  }
  Extras ex = ex_in;
  ex.tpw = 1;
  if (ex.c_count == 1 && !segments && !(tune & GWTF_TUNE_SINGLE_TILE)) {
    // one-coupling launch on a grid of several rounds: as many tiles of a shape per workgroup as still leave every resident slot
    // a workgroup (the staging of the coupling is then paid once per slot, not once per tile)
    const int MBi = gwtf_padded_width(f) / 16;
    const TileCost tc = tile_cost(MBi, nb == 4 ? 2 : nb - 1);
    const long tps = (N + 64 * nb - 1) / (64 * nb), total = (long)K * B * tps;
    ex.tpw = (int)std::max(1L, std::min(tps, total / (256L * tc.slots)));
  }
  const bool pipe = !(tune & GWTF_TUNE_GENERIC_BODY);
  const int kk_steps = (f + 3) / 4;
#define GWTF_ARGS nb, p, packed_w, film, out, logdet, ps, mus, logvars, B, N, C, pattern0, eps, mode, kk_steps, segments, K, p_stride_k, out_stride_k, ex, pipe, st
  switch (gwtf_padded_width(f) / 16) {
    case 1: return launch_nb<1>(GWTF_ARGS);
    case 2: return launch_nb<2>(GWTF_ARGS);
    case 3: return launch_nb<3>(GWTF_ARGS);
    case 4: return launch_nb<4>(GWTF_ARGS);
    case 5: return launch_nb<5>(GWTF_ARGS);
    case 6: return launch_nb<6>(GWTF_ARGS);
    case 7: return launch_nb<7>(GWTF_ARGS);
    case 8: return launch_nb<8>(GWTF_ARGS);
    default: return GWTF_E_BADARG;
  }
#undef GWTF_ARGS
}

extern "C" int gwtf_stack_forward_flagging(const float* p, const float* packed_w, const float* film, float* out,
                                           float* logdet, float* ps, float* mus, float* logvars, const int* segments,
                                           int K, int B, int N, int C, int f, int pattern0, float eps, int mode,
                                           size_t p_stride_k, size_t out_stride_k, int* worklist, int tune, void* stream) {
  const Extras ex = {0, 0, nullptr, nullptr, 0, 0, worklist};
  return stack_dispatch(p, packed_w, film, out, logdet, ps, mus, logvars, segments, K, B, N, C, f, pattern0, eps, mode,
                        p_stride_k, out_stride_k, ex, tune, stream);
}

extern "C" int gwtf_stack_forward_multi(const float* p, const float* packed_w, const float* film, float* out,
                                        float* logdet, float* ps, float* mus, float* logvars, const int* segments,
                                        int K, int B, int N, int C, int f, int pattern0, float eps, int mode,
                                        size_t p_stride_k, size_t out_stride_k, int tune, void* stream) {
  int* worklist = nullptr;
  const Extras ex = {0, 0, nullptr, nullptr, 0, 0, worklist};
  return stack_dispatch(p, packed_w, film, out, logdet, ps, mus, logvars, segments, K, B, N, C, f, pattern0, eps, mode,
                        p_stride_k, out_stride_k, ex, tune, stream);
}

// One elementary coupling of the stack (train-mode pipeline: BatchNorm statistics are only known coupling by
// coupling): applies coupling `c` to p -> out, continues the log-det from logdet_in, optionally writes list
// slot c and accumulates the coordinate moments of `out` for the next coupling's sd0_bn.
extern "C" int gwtf_train_apply(const float* p, const float* packed_w, const float* film, float* out,
                                const float* logdet_in, float* logdet, float* ps, float* mus, float* logvars,
                                float* moments_out, int c, int B, int N, int C, int f, int pattern0, float eps, int mode,
                                int tune, void* stream) {
  const Extras ex = {c, 1, logdet_in, moments_out, 0, 0, nullptr};
  return stack_dispatch(p, packed_w, film, out, logdet, ps, mus, logvars, nullptr, 1, B, N, C, f, pattern0, eps, mode, 0, 0,
                        ex, tune, stream);
}

// One coupling of K stacks in one launch (K-batched train pipeline, gwtf_train.hip): component k reads p + k * p_stride_k,
// continues logdet + k * out_stride_k, accumulates the next coupling's moments into moments_out + k * moments_stride_k.
int gwtf_internal_apply_k(const float* p, const float* packed_w, const float* film, float* out, const float* logdet_in,
                          float* logdet, float* ps, float* mus, float* logvars, float* moments_out, size_t moments_stride_k,
                          int c, int K, int B, int N, int C, int f, int pattern0, float eps, int mode, size_t p_stride_k,
                          size_t out_stride_k, int tune, void* stream) {
  const Extras ex = {c, 1, logdet_in, moments_out, moments_stride_k};
  return stack_dispatch(p, packed_w, film, out, logdet, ps, mus, logvars, nullptr, K, B, N, C, f, pattern0, eps, mode,
                        p_stride_k, out_stride_k, ex, tune, stream);
}

extern "C" int gwtf_stack_forward(const float* p, const float* packed_w, const float* film, float* out, float* logdet,
                                  float* ps, float* mus, float* logvars, int B, int N, int C, int f, int pattern0,
                                  float eps, int mode, int tune, void* stream) {
  return gwtf_stack_forward_multi(p, packed_w, film, out, logdet, ps, mus, logvars, nullptr, 1, B, N, C, f, pattern0, eps,
                                  mode, 0, 0, tune, stream);
}

namespace {
template <int MB>
int launch_stats(int nb, const float* p, const float* pw_c, float* ystats, int B, int N, int pat, int kk_steps, int K,
                 size_t p_sk, size_t pw_sk, size_t ys_sk, hipStream_t st) {
  const int pts_wg = 64 * nb;
  const dim3 grid((unsigned)(B * ((N + pts_wg - 1) / pts_wg)), (unsigned)K), block(256);
  if constexpr (MB == 3) {          // abs-form widths (f = 33..40): the contraction as one basic block
    if (kk_steps - 8 <= 2) {
      switch (nb) {
        case 1: hipLaunchKernelGGL((stats_kernel<MB, 1, 1>), grid, block, 0, st, p, pw_c, ystats, B, N, pat, kk_steps, p_sk, pw_sk, ys_sk); break;
        case 2: hipLaunchKernelGGL((stats_kernel<MB, 2, 1>), grid, block, 0, st, p, pw_c, ystats, B, N, pat, kk_steps, p_sk, pw_sk, ys_sk); break;
        default: hipLaunchKernelGGL((stats_kernel<MB, 4, 1>), grid, block, 0, st, p, pw_c, ystats, B, N, pat, kk_steps, p_sk, pw_sk, ys_sk); break;
      }
      return (int)hipGetLastError();
    }
  }
  switch (nb) {
    case 1: hipLaunchKernelGGL((stats_kernel<MB, 1>), grid, block, 0, st, p, pw_c, ystats, B, N, pat, kk_steps, p_sk, pw_sk, ys_sk); break;
    case 2: hipLaunchKernelGGL((stats_kernel<MB, 2>), grid, block, 0, st, p, pw_c, ystats, B, N, pat, kk_steps, p_sk, pw_sk, ys_sk); break;
    default: hipLaunchKernelGGL((stats_kernel<MB, 4>), grid, block, 0, st, p, pw_c, ystats, B, N, pat, kk_steps, p_sk, pw_sk, ys_sk); break;
  }
  return (int)hipGetLastError();
}
}  // namespace

// statistics pass of one coupling of K stacks (component k: p + k*p_sk, packed_w_c + k*pw_sk, ystats + k*ys_sk)
int gwtf_internal_stats_k(const float* p, const float* packed_w_c, float* ystats, int K, int B, int N, int f, int pattern,
                          size_t p_sk, size_t pw_sk, size_t ys_sk, int tune, void* stream) {
  if (B <= 0 || N <= 0 || K <= 0 || f <= 0 || f > GWTF_MAX_FP_TRAIN || pattern < 0 || pattern > 5 || !p || !packed_w_c || !ystats)
    return GWTF_E_BADARG;
  const long pts = (long)B * N * K;
  int nb = (tune & 0xffff) / 16;
  if (nb != 1 && nb != 2 && nb != 4) nb = pts >= 2048L * 64 ? 4 : (pts >= 2048L * 32 ? 2 : 1);
  if (f > 64 && nb > 2) nb = 2;
  const int kk_steps = (f + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  switch (gwtf_padded_width(f) / 16) {
    case 1: return launch_stats<1>(nb, p, packed_w_c, ystats, B, N, pattern, kk_steps, K, p_sk, pw_sk, ys_sk, st);
    case 2: return launch_stats<2>(nb, p, packed_w_c, ystats, B, N, pattern, kk_steps, K, p_sk, pw_sk, ys_sk, st);
    case 3: return launch_stats<3>(nb, p, packed_w_c, ystats, B, N, pattern, kk_steps, K, p_sk, pw_sk, ys_sk, st);
    case 4: return launch_stats<4>(nb, p, packed_w_c, ystats, B, N, pattern, kk_steps, K, p_sk, pw_sk, ys_sk, st);
    case 5: return launch_stats<5>(nb, p, packed_w_c, ystats, B, N, pattern, kk_steps, K, p_sk, pw_sk, ys_sk, st);
    case 6: return launch_stats<6>(nb, p, packed_w_c, ystats, B, N, pattern, kk_steps, K, p_sk, pw_sk, ys_sk, st);
    default: return GWTF_E_BADARG;
  }
}

extern "C" int gwtf_train_stats(const float* p, const float* packed_w_c, float* ystats, int B, int N, int f, int pattern,
                                int tune, void* stream) {
  return gwtf_internal_stats_k(p, packed_w_c, ystats, 1, B, N, f, pattern, 0, 0, 0, tune, stream);
}
