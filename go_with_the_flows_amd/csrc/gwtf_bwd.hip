// gwtf_bwd.hip -- backward of ONE elementary coupling (both directions; BatchNorm as a fixed affine: model.eval(),
// or the folded statistics of the train pipeline held constant).
//
// Reference semantics: autograd through CondRealNVPFlow3D.forward (lib/networks/flows.py:95-117), i.e. what
// loss.backward() (training.py:54) computes for this layer.  Formulation in the FOLDED parameters the forward
// kernel consumes (gwtf_layout.h):  per branch X in {logvar, mu}
//     pre = W0f [xa;xb] + c0f      h = relu(pre)      acc = W1p h + c(b)      z = relu(acc)      o_w = u_w(b).z + bias_w
//     lv = softsign(o^lv)   mu = o^mu   s = sqrt(eps + exp(lv))   out_w = (x_w - mu)/s   out_k = x_k/sqrt(eps+1)   ld_w += lv
// One pass per coupling: the forward is RECOMPUTED from the saved coupling input (12 B/pt), then
//     dt -> dz = u.dt -> dacc = dz[acc>0]      (in the accumulators' C layout, in place)
//     dh = W1p^T dacc                           v_mfma_f32_16x16x32_f16, three-product split: the C-layout registers of
//                                               dacc, split into f16 hi/lo, ARE its B operand (k-slot map: GwtfPackB)
//     dpre = dh[pre>0]   dxk = W0f^T dpre   dx_w = g_out_w / s
// Reductions over points (per-shape FiLM record grads dc, du_w; sd0 grads; sd2 bias grads) are done in-lane over
// the point blocks, by shuffles over the 16 lanes of a quarter, in LDS over the 4 waves, then one atomic per value
// and workgroup.  The f x f weight gradient dW1p = sum_p dacc(p) h(p)^T contracts over POINTS, which sit on the lanes
// (MFMA N axis) everywhere else in this kernel.  dacc (already split into f16 hi/lo for the dh product) is transposed
// through LDS -- one [FP][points of the workgroup] image aliased onto the forward-weight buffer, dead by then -- and read
// back as the A operand (rows = dacc features, k-slots = points); h = relu(W0f x + c0f) is simply RECOMPUTED in B-operand
// layout (a lane = one h feature x 8 points: 3 weights, 16 coordinates from a 1-KiB LDS table), so only one matrix is
// transposed.  Each wavefront owns one column of 16x16 output tiles over ALL points of the workgroup (no cross-wave
// reduction), and writes it to a per-workgroup partial that gwtf_dw1_reduce sums in a fixed order: nothing of size
// O(B N f) is written to HBM any more (it was 100-150 MB per coupling).
#include "gwtf_device.h"
#include "gwtf_dw1.h"
#include <algorithm>

#ifndef GWTF_K2_MASK
#define GWTF_K2_MASK 15     // A/B knob: which compile-time-pattern variants the launcher uses (1 / 2: light pass, one warped / one kept; 4 / 8: merged pass)
#endif
#ifndef GWTF_BWD_ABLATE
#define GWTF_BWD_ABLATE 0   // TIMING PROBES of the merged pass (wrong results): 1 no sd0 LDS atomics | 2 no dW1 partial stores | 4 per-wave
                            // scale (no amax barrier) | 8 no dW1 product | 16 no sd0 sums at all | 32 no transposed-dacc stores
#endif
namespace {

using namespace gwtf_dev;

template <int MB>
struct BCfg {
  static constexpr int FP = 16 * MB;
  static constexpr int KS = (FP + 31) / 32;
  static constexpr int W1T = MB * KS * 2 * 256;         // floats, one branch: [mi][ks][part][lane][8 f16] (GwtfPackB)
  static constexpr int PB = 2 * W1T + 2 * FP * 4;       // + SD0N[2][FP][4]
  // LDS holds ONE branch's W1T at a time (+ SD0N of both): branch 1's transposed weights stream in over branch 0's while
  // the dW1 product of branch 0 runs.  12 KB less LDS at f = 37: three workgroups per CU instead of two.
  static constexpr int PB_LDS = W1T + 2 * FP * 4;
};

// VAR selects what one pass over the points does:
//   BW_DIRECT  backward of the coupling itself (BatchNorm as a fixed affine map: the eval-mode backward, and the first half of
//              the two-pass train-mode chain of the per-coupling autograd nodes).
//   BW_STATS   backward of the train-mode statistics pass (gwtf_train_stats): the upstream is g_stats[branch][{d/dSum y,
//              d/dSum y^2}][FP], i.e. dL/dy(p) = gS + 2 gQ y(p) for every point, y = un-biased accumulator (no FiLM record, no
//              tail); everything after dacc is shared.
//   BW_LIGHT / BW_MERGED   the train pipeline's two passes.  g_stats depends on the coupling path only through the per-shape sums of
//              dacc and dacc-weighted activations (the FiLM-record gradients, fold1_bwd_kernel), so LIGHT recomputes the forward and
//              the tail's backward and leaves ONLY those sums (and the sd2 bias sums); MERGED then runs the expensive part -- dh = W1^T
//              dy, the sd0 sums, dx, the dW1 partial -- ONCE on dy = dacc + gS + 2 gQ (acc - c) instead of once per path (both are
//              linear in dy): per level one full pass and one forward-sized pass instead of two full ones, and one set of dW1 partials.
// waves per SIMD the register allocation is held to: the LDS footprint admits three workgroups per CU up to FP = 48
// K2: the warp pattern's number of kept coordinates known at compile time (1: two kept / ONE warped coordinate, patterns 0-2;
// 0: one kept / two warped; -1: read from `pat`).  With one warped coordinate the second output column of sd2 does not exist: its
// u_1 terms, its FiLM-record sum (one of three 16-lane reductions + LDS atomics per feature row in the light pass) and the second
// tail slot fold away -- the train pipeline's launcher picks the variant per level (the pattern is a host-side fact).
enum { BW_DIRECT = 0, BW_STATS = 1, BW_LIGHT = 2, BW_MERGED = 3 };
// FULL: N is a multiple of the workgroup's tile (no point beyond the cloud: the per-element bound selects of the merged pass fold away).
template <int MB, int NB, int VAR, int MG = -1, int K2 = -1, bool FULL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 4 ? 2 : (MB <= 3 ? 3 : 1)))) void bwd_kernel(const float* __restrict__ x_in, const float* __restrict__ g_out,
                                                  const float* __restrict__ g_ld, const float* __restrict__ pw_c,
                                                  const float* __restrict__ pb_c, const float* __restrict__ film,
                                                  float* __restrict__ g_in, float* __restrict__ dw1_ws,
                                                  float* __restrict__ g_film,
                                                  float* __restrict__ g_sd0, float* __restrict__ g_bias,
                                                  const float* __restrict__ g_stats, int B, int N, int C, int c, int pat,
                                                  float eps, int kk_steps, int f, int mode, const GwtfKS ks_,
                                                  const float* __restrict__ g_ps_c, const float* __restrict__ g_lvs_c,
                                                  const GwtfCombine cmb) {
  using K = Cfg<MB>;
  using KB = BCfg<MB>;
  constexpr int FP = K::FP;
  constexpr bool STATS = VAR == BW_STATS, LIGHT = VAR == BW_LIGHT, MERGED = VAR == BW_MERGED;
  {
    // blockIdx.y = mixture component of the K-batched train pipeline (all strides 0, Ctot == C for a single stack)
    const size_t comp = blockIdx.y;
    x_in += comp * ks_.x;
    pw_c += comp * ks_.pw;
    pb_c += comp * ks_.pb;
    g_in += comp * ks_.pts;
    dw1_ws += comp * ks_.dw1;
    g_sd0 += comp * ks_.gsd0;
    if (MERGED) g_stats += comp * ks_.gstats;
    if (!STATS) {
      g_out += comp * ks_.pts;
      g_ld += comp * ks_.pts;
      // gradients entering through this coupling's own list slots (ps[c], logvars[c]; lists are [K][Cper][B][3][N])
      if (g_ps_c) g_ps_c += comp * ks_.Cper * ks_.pts;
      if (g_lvs_c) g_lvs_c += comp * ks_.Cper * ks_.pts;
      g_bias += comp * ks_.gbias;
      c += (int)comp * ks_.Cper;     // FiLM-side arrays: [shape][Ctot][...], coupling k*Cper + c
      C = ks_.Ctot;
    } else {
      g_stats += comp * ks_.gstats;
    }
  }
  __shared__ __align__(16) float lds[K::PW + K::FSP + (LIGHT ? 0 : KB::PB_LDS)];
  // sd0 sums: one set per workgroup (atomics), or one per wave (GWTF_ROWSUM_KEEP >= 2: plain stores, summed by the flush)
  constexpr int SD0W = (GWTF_ROWSUM_KEEP >= 2 && GWTF_ROWSUM_MODE == 1 && !LIGHT) ? 4 : 1;
  __shared__ float s_film[2][3][FP], s_sd0[SD0W][2][3][FP], s_bias[4];
  // MERGED / STATS: the statistics' upstream g_stats [2][2][FP] staged once per workgroup (read from global memory where it is used
  // it was 48 conditional loads -- each its own exec-masked block -- per wave inside the dacc loop)
  __shared__ __align__(16) float s_gst[(STATS || MERGED) ? 4 * FP : 4];
  // dW1 machinery: coordinates of the workgroup's points, per-wave |dacc| maxima, the transposed dacc image
  constexpr int PTS = 64 * NB;                      // points per workgroup
  constexpr int XPITCH = 2 * PTS + 8;               // bytes per feature row (+8: rows land 8 B apart in the banks)
  constexpr int XBYTES = 2 * FP * XPITCH;           // hi image | lo image
  constexpr bool XALIAS = XBYTES <= K::PW * 4;      // fits over the forward weights (dead after the forward recompute)
  __shared__ float2 s_pts[PTS];
  __shared__ float s_amax[2][4];
  __shared__ __align__(16) unsigned char s_xt_own[XALIAS ? 16 : XBYTES];
#ifdef GWTF_BWD_LDS_PAD      // A/B knob: KiB of dead LDS in the merged pass (forces fewer workgroups per compute unit; docs/LOG.md round 5)
  __shared__ float s_pad[MERGED ? GWTF_BWD_LDS_PAD * 256 : 1];
  if (B < 0) { s_pad[threadIdx.x] = eps; g_in[0] = s_pad[N & 255]; }
#endif
  for (int t = threadIdx.x; t < 2 * 3 * FP; t += blockDim.x) (&s_film[0][0][0])[t] = 0.f;
  for (int t = threadIdx.x; t < SD0W * 2 * 3 * FP; t += blockDim.x) (&s_sd0[0][0][0][0])[t] = 0.f;
  if (threadIdx.x < 4) s_bias[threadIdx.x] = 0.f;
  if (STATS || MERGED)
    for (int t = threadIdx.x; t < 4 * FP; t += blockDim.x) s_gst[t] = (t % FP) < f ? g_stats[t] : 0.f;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, i16 = lane & 15;
  const bool combine = (LIGHT || MERGED) && cmb.gm != nullptr;   // on-the-fly gradient combine of the level processed before this one
  const int tiles_per_shape = (N + 64 * NB - 1) / (64 * NB);
  // LIGHT: a workgroup walks `tpw` consecutive tiles of ONE shape -- the weights and the shape's FiLM record are staged once, and
  // the per-shape sums leave the workgroup once (1 / tpw of the global atomics).  Every other variant: one tile (tpw = 1).
  const int tpw = LIGHT && ks_.tpw > 1 ? ks_.tpw : 1;
  const int groups = (tiles_per_shape + tpw - 1) / tpw;
  const int b = blockIdx.x / groups;
  const int tile0 = (blockIdx.x - b * groups) * tpw;
  const int own_nb = q & (NB - 1);

  // stage: forward record | FiLM record of this shape | backward record
  {
    const float* src_f = film + ((size_t)b * C + c) * K::FS + lane * 4;
#pragma unroll
    for (int i = 0; i < (K::PW / 256 + 3) / 4; ++i) {
      const int piece = wave + 4 * i;
      if (piece < K::PW / 256)
        __builtin_amdgcn_global_load_lds((glb_void*)(pw_c + piece * 256 + lane * 4), (lds_void*)&lds[piece * 256], 16, 0, 0);
    }
    if (!STATS && wave < K::FSP / 256 && wave * 256 + lane * 4 < K::FS)
      __builtin_amdgcn_global_load_lds((glb_void*)(src_f + wave * 256), (lds_void*)&lds[K::PW + wave * 256], 16, 0, 0);
    for (int piece = wave; !LIGHT && piece * 256 < KB::W1T; piece += 4)        // W1T of branch 0 (W1T is a multiple of 256 floats)
      __builtin_amdgcn_global_load_lds((glb_void*)(pb_c + piece * 256 + lane * 4),
                                       (lds_void*)&lds[K::PW + K::FSP + piece * 256], 16, 0, 0);
    for (int piece = wave; !LIGHT && piece * 256 < 2 * FP * 4; piece += 4) {   // SD0N of both branches
      if (piece * 256 + lane * 4 < 2 * FP * 4)
        __builtin_amdgcn_global_load_lds((glb_void*)(pb_c + 2 * KB::W1T + piece * 256 + lane * 4),
                                         (lds_void*)&lds[K::PW + K::FSP + KB::W1T + piece * 256], 16, 0, 0);
    }
  }
  int k0, k1, w0, w1;
  gwtf_pattern_dims(pat, &k0, &k1, &w0, &w1);
  const bool keep2 = K2 < 0 ? pat < 3 : K2 == 1;
  const int nw = keep2 ? 1 : 2;
#ifdef GWTF_NO_ONE_W
  constexpr bool ONE_W = false;
#else
  constexpr bool ONE_W = K2 == 1;          // compile-time: one warped coordinate (everything with index 1 of a warped slot is absent)
#endif
  constexpr bool ONE_K = K2 == 0;          // compile-time: one kept coordinate (sd0 has ONE input: its second weight column, x_b and their sums are absent)
  // this workgroup's dW1 partial of branch br, column tile ni (gwtf_dw1.h PARTIAL RECORD: [2][f columns][RP rows]): the lane's four
  // consecutive rows of each row tile as one 16-byte streaming store (the partials are read once, by another kernel)
  auto store_partial = [&](float* ws, int br, int ni, const f32x4 (&dw)[MB]) {
    const int RP = gwtf_dw1::rows_padded(f), col = 16 * ni + i16;
    if ((GWTF_BWD_ABLATE & 2) && MERGED && N > 0) return;
    if (col >= f) return;
    float* out = ws + (size_t)blockIdx.x * gwtf_dw1::rec_floats(f) + ((size_t)br * f + col) * RP + 4 * q;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
      if (16 * mi + 4 * q < RP) __builtin_nontemporal_store(dw[mi], reinterpret_cast<f32x4*>(out + 16 * mi));
  };

  for (int it = 0; it < tpw; ++it) {
  const int tile = tile0 + it;
  if (tile >= tiles_per_shape) break;
  const int n_wave0 = (tile * 4 + wave) * 16 * NB;
  const int n_own = n_wave0 + 16 * own_nb + i16;
  const bool own_inrange = FULL || n_own < N;
  const bool own_valid = own_inrange && q < NB;
  // own point: input coordinates and upstream gradients (zero beyond N: every derived gradient is then zero)
  float xo[3], go[3], gl[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const size_t o = ((size_t)b * 3 + d) * N + n_own;
    xo[d] = own_inrange ? x_in[o] : 0.f;
    go[d] = (!STATS && own_inrange) ? g_out[o] : 0.f;
    gl[d] = (!STATS && own_inrange) ? g_ld[o] : 0.f;
    if (!STATS && own_inrange && g_ps_c) go[d] += g_ps_c[o];
    if (!STATS && own_inrange && g_lvs_c) gl[d] += g_lvs_c[o];
  }
  float xa[NB], xb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    xa[nb] = __shfl(sel3(xo[0], xo[1], xo[2], k0), 16 * nb + i16);
    xb[nb] = keep2 ? __shfl(sel3(xo[0], xo[1], xo[2], k1), 16 * nb + i16) : 0.f;
  }
  if (q < NB) s_pts[wave * 16 * NB + 16 * q + i16] = make_float2(sel3(xo[0], xo[1], xo[2], k0), keep2 ? sel3(xo[0], xo[1], xo[2], k1) : 0.f);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* L = lds;
  const float* LB = lds + K::PW + K::FSP;

  // ---- forward recompute: accumulators of both branches stay live -------------------------------------------
  f32x4 acc[2][MB][NB];
  float res[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    const float* fe = L + K::PW + br * 3 * FP + 4 * q;
    f32x4 cinit[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) cinit[m] = STATS ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(fe + 16 * m);
    // The dispatch stays a RUN-TIME branch on `pat` even where K2 fixes the pattern at compile time: with a constant condition the
    // contraction (its f16 splits are inline asm) is inlined into the surrounding basic block and the merged pass came out wrong
    // (errors of 1e-4 .. O(1) in dL/dx at 128 x 2048, f = 37; the asm-hazard rule of docs/LOG.md: a value produced by inline asm is
    // consumed in the block that produces it).
    if (pat < 3) sd1_contract<MB, NB, true, MG>(L, br, kk_steps, lane, q, xa, xb, cinit, acc[br]);
    else sd1_contract<MB, NB, false, MG>(L, br, kk_steps, lane, q, xa, xb, cinit, acc[br]);
    if (STATS) continue;
    float o0[NB], o1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) o0[nb] = o1[nb] = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(fe + FP + 16 * m);
      const f32x4 u1 = *reinterpret_cast<const f32x4*>(fe + 2 * FP + 16 * m);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float v = fmaxf(acc[br][m][nb][r], 0.f);
          o0[nb] = fmaf(u0[r], v, o0[nb]);
          if (!ONE_W) o1[nb] = fmaf(u1[r], v, o1[nb]);
        }
    }
    res[br][0] = quarter_reduce<NB>(o0, q);
    if (!ONE_W) res[br][1] = quarter_reduce<NB>(o1, q);
  }
  const f32x4 bias = STATS ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(L + K::PW + 6 * FP);
  const float s_keep = sqrtf(eps + 1.0f);

  // ---- tail forward + backward on the own point ------------------------------------------------------------
  float dt[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // [branch][warped slot] = dL/d o
  float xw[2] = {0.f, 0.f}, gow[2] = {0.f, 0.f}, glw[2] = {0.f, 0.f}, gx[2] = {0.f, 0.f};
  const float keep_scale = mode == GWTF_MODE_INVERSE ? 1.0f / s_keep : s_keep;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d == w0) { xw[0] = xo[d]; glw[0] = gl[d]; }
    if (!keep2 && d == w1) { xw[1] = xo[d]; glw[1] = gl[d]; }
  }
  // forward quantities of the tail (hardware rcp / sqrt / exp2 like the forward kernel, gwtf_stack.hip)
  float t_rden[2] = {0.f, 0.f}, t_e[2] = {0.f, 0.f}, t_sc[2] = {1.f, 1.f}, t_rsc[2] = {1.f, 1.f}, t_out[2] = {0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (!STATS && s < nw) {
      const float t = res[0][s] + bias[s];
      const float den = 1.0f + fabsf(t);
      t_rden[s] = __builtin_amdgcn_rcpf(den);
      const float lv = t * t_rden[s];
      const float mu = res[1][s] + bias[2 + s];
      t_e[s] = __expf(lv);
      t_sc[s] = __builtin_amdgcn_sqrtf(eps + t_e[s]);
      t_rsc[s] = __builtin_amdgcn_rcpf(t_sc[s]);
      t_out[s] = mode == GWTF_MODE_INVERSE ? (xw[s] - mu) * t_rsc[s] : fmaf(t_sc[s], xw[s], mu);   // this coupling's output
    }
  }
  if ((LIGHT || MERGED) && combine) {
    // g_out += gM + Q x_out of the level processed before (x_out = this coupling's output = that level's input)
    float xout[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) xout[d] = d == w0 ? t_out[0] : ((!keep2 && d == w1) ? t_out[1] : xo[d] * keep_scale);
    const float* gmv = cmb.gm + (size_t)blockIdx.y * cmb.gm_sk;    // wave-uniform address: scalar loads, no long-lived registers
    const float q00 = 2.f * gmv[3], q01 = gmv[4], q02 = gmv[5], q11 = 2.f * gmv[6], q12 = gmv[7], q22 = 2.f * gmv[8];
    if (own_inrange) {
      go[0] += gmv[0] + q00 * xout[0] + q01 * xout[1] + q02 * xout[2];
      go[1] += gmv[1] + q01 * xout[0] + q11 * xout[1] + q12 * xout[2];
      go[2] += gmv[2] + q02 * xout[0] + q12 * xout[1] + q22 * xout[2];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d == w0) gow[0] = go[d];
    if (!keep2 && d == w1) gow[1] = go[d];
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (!STATS && s < nw) {
      float dsc;
      if (mode == GWTF_MODE_INVERSE) {       // out = (x - mu)/s
        gx[s] = gow[s] * t_rsc[s];
        dsc = -gow[s] * t_out[s] * t_rsc[s];
        dt[1][s] = -gx[s];
      } else {                                // out = s*x + mu
        gx[s] = gow[s] * t_sc[s];
        dsc = gow[s] * xw[s];
        dt[1][s] = gow[s];
      }
      const float dlv = glw[s] + dsc * t_e[s] * (0.5f * t_rsc[s]);
      dt[0][s] = dlv * (t_rden[s] * t_rden[s]);
    }
  }
  float gin[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) gin[d] = STATS ? 0.f : (d == w0 ? gx[0] : ((!keep2 && d == w1) ? gx[1] : go[d] * keep_scale));
  if (!STATS && !MERGED) {  // sd2 bias gradient: sum of dt over the wave's valid points
    float bsum[4] = {own_valid ? dt[0][0] : 0.f, own_valid ? dt[0][1] : 0.f, own_valid ? dt[1][0] : 0.f,
                     own_valid ? dt[1][1] : 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) bsum[i] = wave_sum(bsum[i]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(&s_bias[i], bsum[i]);
    }
  }

  // ---- per branch: dacc, FiLM-record grads, dh = W1p^T dacc, dpre, sd0 grads, dx_kept ----------------------
  float dxa_own = 0.f, dxb_own = 0.f;
  constexpr bool DEFER_DW = MB <= 4 && !LIGHT;      // one column of dW1 tiles per wave: branch 0's can wait in registers (below)
  f32x4 dwk[MB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) dwk[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    float d0[NB], d1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      d0[nb] = __shfl(dt[br][0], 16 * nb + i16);
      d1[nb] = ONE_W ? 0.f : __shfl(dt[br][1], 16 * nb + i16);
    }
    const float* fe = L + K::PW + br * 3 * FP + 4 * q;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 u0 = STATS ? zero4 : *reinterpret_cast<const f32x4*>(fe + FP + 16 * m);
      const f32x4 u1 = STATS ? zero4 : *reinterpret_cast<const f32x4*>(fe + 2 * FP + 16 * m);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ft = 16 * m + 4 * q + r;
        if (STATS) {
          const float gs = s_gst[(br * 2 + 0) * FP + ft], gq2 = 2.0f * s_gst[(br * 2 + 1) * FP + ft];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int n = n_wave0 + 16 * nb + i16;
            const float da = n < N ? fmaf(gq2, acc[br][m][nb][r], gs) : 0.f;
            acc[br][m][nb][r] = da;
          }
          continue;
        }
        if (MERGED) {     // dy = dacc (coupling path) + gS + 2 gQ y (statistics path), y = acc - c; the FiLM sums were LIGHT's job
          const float gs = s_gst[(br * 2 + 0) * FP + ft], gq2 = 2.0f * s_gst[(br * 2 + 1) * FP + ft];
          const float cc = fe[16 * m + r];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int n = n_wave0 + 16 * nb + i16;
            const float a = acc[br][m][nb][r];
            const float da = a > 0.f ? (ONE_W ? u0[r] * d0[nb] : fmaf(u0[r], d0[nb], u1[r] * d1[nb])) : 0.f;
            acc[br][m][nb][r] = (FULL || n < N) ? da + fmaf(gq2, a - cc, gs) : 0.f;
          }
          continue;
        }
        float sdc = 0.f, sdu0 = 0.f, sdu1 = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float a = acc[br][m][nb][r];
          const float z = fmaxf(a, 0.f);
          sdu0 = fmaf(d0[nb], z, sdu0);
          if (!ONE_W) sdu1 = fmaf(d1[nb], z, sdu1);
          const float da = a > 0.f ? (ONE_W ? u0[r] * d0[nb] : fmaf(u0[r], d0[nb], u1[r] * d1[nb])) : 0.f;
          sdc += da;
          acc[br][m][nb][r] = da;
        }
        sdc = row_sum_part(sdc);
        sdu0 = row_sum_part(sdu0);
        if (!ONE_W) sdu1 = row_sum_part(sdu1);
        if (row_sum_owner(i16)) {
          atomicAdd(&s_film[br][0][ft], sdc);
          atomicAdd(&s_film[br][1][ft], sdu0);
          if (!ONE_W) atomicAdd(&s_film[br][2][ft], sdu1);        // (zero on entry: the absent column's sum stays zero)
        }
      }
    }
    if (LIGHT) continue;
    // dh[16mi + 4q + r][point] = sum_j W1p[j][16mi + ..] dacc[j][point] on the f16 MFMA with the three-product split.
    // Gradients have no natural scale (a 1/(B N) loss normalisation puts them near the f16 subnormals), so the WORKGROUP
    // rescales dacc by a power of two that brings its largest magnitude to [2^8, 2^9) and undoes it on dh and dW1: exact.
    // (One scale per workgroup, not per wave: the dW1 product below mixes the waves' points on its K axis.)
    float amax = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(acc[br][m][nb][r]));
    amax = wave_max(amax);
    if (lane == 0) s_amax[br][wave] = amax;
    if (br == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of branch 1's W1T have landed
    if (!((GWTF_BWD_ABLATE & 4) && MERGED)) {
    __syncthreads();   // also: every wave is past its forward recompute (branch 0) / past reading the previous X image (branch 1)
    amax = fmaxf(fmaxf(s_amax[br][0], s_amax[br][1]), fmaxf(s_amax[br][2], s_amax[br][3]));
    }
    const int ebits = (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, amax)) >> 23) & 0xff;
    const bool rescale = ebits >= 16 && ebits <= 240;
    const float up = rescale ? __builtin_bit_cast(float, (262 - ebits) << 23) : 1.0f;     // 2^(8 - (ebits - 127))
    const float down = rescale ? __builtin_bit_cast(float, (ebits - 8) << 23) : 1.0f;     // 2^((ebits - 127) - 8)
    f16x8 dhi[KB::KS][NB], dlo[KB::KS][NB];
#pragma unroll
    for (int ks = 0; ks < KB::KS; ++ks)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int m = 2 * ks + half;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            f16x2 h = {(_Float16)0.f, (_Float16)0.f}, l = {(_Float16)0.f, (_Float16)0.f};
            if (m < MB) {
              const f32x2 v = {acc[br][m < MB ? m : 0][nb][2 * pr] * up, acc[br][m < MB ? m : 0][nb][2 * pr + 1] * up};
              split_pair(v, h, l);
            }
            dhi[ks][nb][4 * half + 2 * pr] = h[0]; dhi[ks][nb][4 * half + 2 * pr + 1] = h[1];
            dlo[ks][nb][4 * half + 2 * pr] = l[0]; dlo[ks][nb][4 * half + 2 * pr + 1] = l[1];
          }
        }
    const float* w1t = LB + lane * 4;
    const f32x4* sd0n = reinterpret_cast<const f32x4*>(LB + KB::W1T + br * FP * 4);
    float pxa[NB], pxb[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) pxa[nb] = pxb[nb] = 0.f;
    constexpr bool SLOT0 = SD0W == 4;                         // per-wave slots, plain stores (gwtf_device.h GWTF_ROWSUM_KEEP)
    const int ws0 = SLOT0 ? wave : 0;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi) {
      f32x4 dh[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dh[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KB::KS; ++ks) {
        const f16x8 ahi = *reinterpret_cast<const f16x8*>(w1t + ((mi * KB::KS + ks) * 2 + 0) * 256);
        const f16x8 alo = *reinterpret_cast<const f16x8*>(w1t + ((mi * KB::KS + ks) * 2 + 1) * 256);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          dh[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, dhi[ks][nb], dh[nb], 0, 0, 0);
          dh[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, dlo[ks][nb], dh[nb], 0, 0, 0);
          dh[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, dhi[ks][nb], dh[nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dh[nb] *= down;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fi = 16 * mi + 4 * q + r;
        const f32x4 sp = sd0n[fi];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float pre = ONE_K ? fmaf(sp[0], xa[nb], sp[2]) : fmaf(sp[0], xa[nb], fmaf(sp[1], xb[nb], sp[2]));
          const float dp = pre > 0.f ? dh[nb][r] : 0.f;
          pxa[nb] = fmaf(sp[0], dp, pxa[nb]);
          if (!ONE_K) pxb[nb] = fmaf(sp[1], dp, pxb[nb]);
          g0 = fmaf(dp, xa[nb], g0);
          if (!ONE_K) g1 = fmaf(dp, xb[nb], g1);
          g2 += dp;
        }
        if ((GWTF_BWD_ABLATE & 16) && MERGED) continue;
        g0 = row_sum_part(g0);
        if (!ONE_K) g1 = row_sum_part(g1);
        g2 = row_sum_part(g2);
        if ((GWTF_BWD_ABLATE & 1) && MERGED) { asm volatile("" :: "v"(g0), "v"(g1), "v"(g2)); continue; }
        if (row_sum_owner(i16)) {
          if (SLOT0) {
            s_sd0[ws0][br][0][fi] = g0;
            if (!ONE_K) s_sd0[ws0][br][1][fi] = g1;
            s_sd0[ws0][br][2][fi] = g2;
          } else {
            atomicAdd(&s_sd0[0][br][0][fi], g0);
            if (!ONE_K) atomicAdd(&s_sd0[0][br][1][fi], g1);          // (zero on entry: the absent column's sum stays zero)
            atomicAdd(&s_sd0[0][br][2][fi], g2);
          }
        }
      }
    }
    dxa_own += quarter_reduce<NB>(pxa, q);
    if (!ONE_K) dxb_own += quarter_reduce<NB>(pxb, q);

    // ---- dW1[br] partial of this workgroup: X = scaled dacc (hi/lo), transposed through LDS; H recomputed --------------
    {
      unsigned char* xt = XALIAS ? reinterpret_cast<unsigned char*>(lds) : s_xt_own;
      const int col = (wave * 16 * NB + i16) * 2;
#pragma unroll
      for (int ks = 0; ks < KB::KS; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int m = 2 * ks + (e >> 2);
            if (m < MB && !((GWTF_BWD_ABLATE & 32) && MERGED)) {
              unsigned char* dst = xt + (16 * m + 4 * q + (e & 3)) * XPITCH + col + 32 * nb;
              *reinterpret_cast<_Float16*>(dst) = dhi[ks][nb][e];
              *reinterpret_cast<_Float16*>(dst + FP * XPITCH) = dlo[ks][nb][e];
            }
          }
      __syncthreads();
      if (br == 0) {   // every wave is past its dh product: branch 1's W1T replaces branch 0's behind the dW1 product
        for (int piece = wave; piece * 256 < KB::W1T; piece += 4)
          __builtin_amdgcn_global_load_lds((glb_void*)(pb_c + KB::W1T + piece * 256 + lane * 4),
                                           (lds_void*)&lds[K::PW + K::FSP + piece * 256], 16, 0, 0);
      }
      for (int ni = wave; ni < MB && !((GWTF_BWD_ABLATE & 8) && MERGED); ni += 4) {   // this wave's column(s) of output tiles: h features 16 ni .. (f > 64: two)
        const f32x4 sp = sd0n[16 * ni + i16];                 // the lane's h feature: {w0a, w0b, c0, -}
        f32x4 dw[MB];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) dw[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2 * NB; ++s) {                     // k-steps of 32 points
          const float4* pp = reinterpret_cast<const float4*>(&s_pts[32 * s + 8 * q]);
          f16x8 hhi, hlo;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const float4 two = pp[e2];                         // points 32 s + 8 q + 2 e2, + 1: (xa, xb) each
            f32x2 h = {fmaxf(ONE_K ? fmaf(sp[0], two.x, sp[2]) : fmaf(sp[0], two.x, fmaf(sp[1], two.y, sp[2])), 0.f),
                       fmaxf(ONE_K ? fmaf(sp[0], two.z, sp[2]) : fmaf(sp[0], two.z, fmaf(sp[1], two.w, sp[2])), 0.f)};
            f16x2 hh, hl;
            split_pair(h, hh, hl);
            hhi[2 * e2] = hh[0]; hhi[2 * e2 + 1] = hh[1];
            hlo[2 * e2] = hl[0]; hlo[2 * e2 + 1] = hl[1];
          }
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            const unsigned char* src = xt + (16 * mi + i16) * XPITCH + 64 * s + 16 * q;
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            const f16x4 a0 = *reinterpret_cast<const f16x4*>(src), a1 = *reinterpret_cast<const f16x4*>(src + 8);
            const f16x4 b0 = *reinterpret_cast<const f16x4*>(src + FP * XPITCH), b1 = *reinterpret_cast<const f16x4*>(src + FP * XPITCH + 8);
            const f16x8 xhi = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const f16x8 xlo = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            dw[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xhi, hhi, dw[mi], 0, 0, 0);
            dw[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xhi, hlo, dw[mi], 0, 0, 0);
            dw[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xlo, hhi, dw[mi], 0, 0, 0);
          }
        }
        // C layout: lane (col = h feature 16 ni + i16, q) holds rows = dacc features 16 mi + 4 q + r: store_partial above
        if (DEFER_DW && br == 0) {
          // branch 0's partial stays in registers until branch 1 is through: stored here, the wave would meet its own 12 stores at the
          // `s_waitcnt vmcnt(0)` that guards branch 1's transposed weights (vmcnt counts stores too) -- an HBM write round trip
          // on the critical path of every workgroup
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) dwk[mi] = dw[mi] * down;
          continue;
        }
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) dw[mi] *= down;
        store_partial(dw1_ws, br, ni, dw);
      }
    }
  }
  if (DEFER_DW && !LIGHT && wave < MB) store_partial(dw1_ws, 0, wave, dwk);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d == k0) gin[d] += dxa_own;
    if (keep2 && d == k1) gin[d] += dxb_own;
  }
  if (!LIGHT && own_valid) {
#pragma unroll
    for (int d = 0; d < 3; ++d) g_in[((size_t)b * 3 + d) * N + n_own] = gin[d];
  }
  }   // tiles of this workgroup
  __syncthreads();
  // flush the workgroup's partial sums: FiLM-record grads are per shape (few workgroups per address); sd0 / bias
  // grads are global, spread over GWTF_STAT_REPLICAS copies
  float* gf = g_film + ((size_t)b * C + c) * (2 * 3 * FP);
  float* gs = g_sd0 + (size_t)(blockIdx.x % GWTF_STAT_REPLICAS) * (2 * 3 * FP);
  for (int t = threadIdx.x; t < 2 * 3 * FP; t += blockDim.x) {
    if (!STATS && !MERGED) atomicAdd(&gf[t], (&s_film[0][0][0])[t]);
    if (!LIGHT) {
      float v = (&s_sd0[0][0][0][0])[t];
#pragma unroll
      for (int w = 1; w < SD0W; ++w) v += (&s_sd0[w][0][0][0])[t];
      atomicAdd(&gs[t], v);
    }
  }
  if (!STATS && !MERGED && threadIdx.x < 4) atomicAdd(&g_bias[(blockIdx.x % GWTF_STAT_REPLICAS) * 4 + threadIdx.x], s_bias[threadIdx.x]);
}

// folded parameters (what autograd differentiates) -> forward + backward packed records, see gwtf_layout.h
//   W1p [C][2][f][f]   W0f [C][2][f][2]   c0f [C][2][f]
__global__ void pack_folded_kernel(const float* __restrict__ W1p, const float* __restrict__ W0f,
                                   const float* __restrict__ c0f, float* __restrict__ pw, float* __restrict__ pb, int C,
                                   int f, int FP) {
  const GwtfPackW P(FP);
  const GwtfPackB PBk(FP);
  const int MB = FP / 16, KS = P.KS();
  const size_t PWs = P.coupling_size();
  const size_t W1T = PBk.w1t_size(), PBs = PBk.coupling_size();
  const size_t total = (PWs + PBs) * (size_t)C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / (PWs + PBs));
    size_t o = idx - (size_t)c * (PWs + PBs);
    float v = 0.f;
    if (o < PWs) {
      if (o < 2 * P.a16_size()) {
        const int br = (int)(o / P.a16_size());
        size_t oo = o - (size_t)br * P.a16_size();
        const int jp = (int)(oo % 4), lane = (int)((oo / 4) % 64), part = (int)((oo / 256) % 2);
        const int m = (int)((oo / 512) % MB), ks = (int)(oo / ((size_t)512 * MB));
        const int jo = 16 * m + (lane & 15);
        _Float16 e[2] = {(_Float16)0.f, (_Float16)0.f};
        if (ks < KS && jo < f) {
          const GwtfA16Slot sl = gwtf_a16_slot(f, KS, ks, part, jp);
          const bool absf = gwtf_abs_form(f);        // gwtf_layout.h: halved entries, columns in the merged image's last slot pair
          const float* Wrow = W1p + (((size_t)c * 2 + br) * f + jo) * f;
          for (int t = 0; t < 2; ++t) {
            const int ji = 32 * ks + 4 * (2 * sl.jsrc + t) + (lane >> 4);
            const float w = (ji < f && !sl.zero) ? (absf ? 0.5f : 1.0f) * Wrow[ji] : 0.f;
            const _Float16 hi = (_Float16)w;
            e[t] = sl.lo ? (_Float16)(w - (float)hi) : hi;
          }
          if (absf && sl.zero) {
            double Ca = 0.0, Cb = 0.0, Cc = 0.0;
            for (int k = 0; k < f; ++k) {
              const double w = Wrow[k];
              Ca += w * (double)W0f[(((size_t)c * 2 + br) * f + k) * 2 + 0];
              Cb += w * (double)W0f[(((size_t)c * 2 + br) * f + k) * 2 + 1];
              Cc += w * (double)c0f[((size_t)c * 2 + br) * f + k];
            }
            float e0, e1;
            gwtf_abs_cols(lane >> 4, (float)(0.5 * Ca), (float)(0.5 * Cb), (float)(0.5 * Cc), &e0, &e1);
            e[0] = (_Float16)e0;
            e[1] = (_Float16)e1;
          }
        }
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 pk = {e[0], e[1]};
        v = __builtin_bit_cast(float, pk);
      } else if (o - 2 * P.a16_size() < 2 * P.sd0_size()) {
        size_t oo = o - 2 * P.a16_size();
        const int br = (int)(oo / P.sd0_size());
        oo -= (size_t)br * P.sd0_size();
        const int j = (int)(oo % 8), e = (int)((oo / 8) % 3), qq = (int)((oo / 24) % 4), ks = (int)(oo / 96);
        const int ft = 32 * ks + 4 * j + qq;
        if (ft < f) v = e < 2 ? W0f[(((size_t)c * 2 + br) * f + ft) * 2 + e] : c0f[((size_t)c * 2 + br) * f + ft];
      }
      pw[(size_t)c * PWs + o] = v;
    } else {
      o -= PWs;
      if (o < 2 * W1T) {
        const int br = (int)(o / W1T);
        size_t oo = o - (size_t)br * W1T;
        const float* Wb = W1p + ((size_t)c * 2 + br) * f * f;
        v = gwtf_w1t_slot(PBk, oo, [&](int j, int i) { return (j < f && i < f) ? Wb[(size_t)j * f + i] : 0.f; });
      } else {
        size_t oo = o - 2 * W1T;
        const int br = (int)(oo / ((size_t)FP * 4)), ft = (int)((oo / 4) % FP), e = (int)(oo % 4);
        if (ft < f && e < 3) v = e < 2 ? W0f[(((size_t)c * 2 + br) * f + ft) * 2 + e] : c0f[((size_t)c * 2 + br) * f + ft];
      }
      pb[(size_t)c * PBs + o] = v;
    }
  }
}

template <int MB, int VAR>
int launch_bwd(int nb, const float* x_in, const float* g_out, const float* g_ld, const float* pw_c, const float* pb_c,
               const float* film, float* g_in, float* dw1_ws, float* g_film, float* g_sd0, float* g_bias,
               const float* g_stats, int B, int N, int C, int c, int pat, float eps, int kk_steps, int f, int mode, int K,
               const GwtfKS& ks, const float* g_ps_c, const float* g_lvs_c, const GwtfCombine& cmb, hipStream_t st) {
  const int pts_wg = 64 * nb;
  const dim3 grid((unsigned)(B * ((N + pts_wg - 1) / pts_wg)), (unsigned)K), block(256);
  if constexpr (MB == 3 && (VAR == BW_LIGHT || VAR == BW_MERGED)) {
    // the train pipeline's two passes at the abs-form widths (f = 33..40): the forward recompute as one basic block (MG = 1)
    if (gwtf_abs_form(f)) {
      const bool small_tile_mg = (ks.tune & GWTF_TUNE_SMALL_LIGHT_TILE) != 0;   // per-call diagnostic (tools/diag/light_tile_check.py)
      if (VAR == BW_LIGHT && nb == 2 && (long)B * N * K >= 256L * 1024 && !small_tile_mg) {
        // 256-point tiles, and as many of a shape's tiles per workgroup as still leave one full round of 512 resident workgroups
        const int tps = (N + 255) / 256;
        GwtfKS k4 = ks;
        k4.tpw = (int)std::max(1L, std::min((long)tps, (long)B * tps * K / 512));
        if (ks.tune & GWTF_TUNE_SINGLE_TILE) k4.tpw = 1;
        const dim3 grid4((unsigned)(B * ((tps + k4.tpw - 1) / k4.tpw)), (unsigned)K);
        if (pat < 3 && (GWTF_K2_MASK & 1)) hipLaunchKernelGGL((bwd_kernel<MB, 4, VAR, 1, 1>), grid4, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, k4, g_ps_c, g_lvs_c, cmb);
        else if (pat >= 3 && (GWTF_K2_MASK & 2)) hipLaunchKernelGGL((bwd_kernel<MB, 4, VAR, 1, 0>), grid4, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, k4, g_ps_c, g_lvs_c, cmb);
        else hipLaunchKernelGGL((bwd_kernel<MB, 4, VAR, 1>), grid4, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, k4, g_ps_c, g_lvs_c, cmb);
      } else if (nb == 1) {
        hipLaunchKernelGGL((bwd_kernel<MB, 1, VAR, 1>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      } else if (pat < 3 && (GWTF_K2_MASK & 4) && N % 128 == 0 && VAR == BW_MERGED) {
        hipLaunchKernelGGL((bwd_kernel<MB, 2, VAR, 1, 1, true>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      } else if (pat >= 3 && (GWTF_K2_MASK & 8) && N % 128 == 0 && VAR == BW_MERGED) {
        hipLaunchKernelGGL((bwd_kernel<MB, 2, VAR, 1, 0, true>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      } else if (pat < 3 && (GWTF_K2_MASK & 4)) {
        hipLaunchKernelGGL((bwd_kernel<MB, 2, VAR, 1, 1>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      } else if (pat >= 3 && (GWTF_K2_MASK & 8)) {
        hipLaunchKernelGGL((bwd_kernel<MB, 2, VAR, 1, 0>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      } else {
        hipLaunchKernelGGL((bwd_kernel<MB, 2, VAR, 1>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      }
      return (int)hipGetLastError();
    }
  }
  if constexpr (VAR == BW_LIGHT && MB <= 3) {   // forward-sized pass: the forward kernel's tile (256 points per workgroup) where it fills the GPU
    const bool small_tile = (ks.tune & GWTF_TUNE_SMALL_LIGHT_TILE) != 0;      // per-call diagnostic (tools/diag/light_tile_check.py)
    if (nb == 2 && (long)B * N * K >= 256L * 1024 && !small_tile) {
      const dim3 grid4((unsigned)(B * ((N + 255) / 256)), (unsigned)K);
      hipLaunchKernelGGL((bwd_kernel<MB, 4, VAR>), grid4, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb);
      return (int)hipGetLastError();
    }
  }
#define GWTF_B(NB_) hipLaunchKernelGGL((bwd_kernel<MB, NB_, VAR>), grid, block, 0, st, x_in, g_out, g_ld, pw_c, pb_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, ks, g_ps_c, g_lvs_c, cmb)
  if (nb == 1) GWTF_B(1); else GWTF_B(2);
#undef GWTF_B
  return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t gwtf_packed_b_coupling_floats(int f) {
  return GwtfPackB(gwtf_padded_width(f)).coupling_size();
}

extern "C" int gwtf_pack_folded(const float* W1p, const float* W0f, const float* c0f, float* packed_w, float* packed_b,
                                int C, int f, void* stream) {
  if (!W1p || !W0f || !c0f || !packed_w || !packed_b || C <= 0 || f <= 0 || f > GWTF_MAX_FP_TRAIN) return GWTF_E_BADARG;
  const int FP = gwtf_padded_width(f);
  hipLaunchKernelGGL(pack_folded_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, W1p, W0f, c0f, packed_w, packed_b, C,
                     f, FP);
  return (int)hipGetLastError();
}

extern "C" int gwtf_coupling_backward_lists(const float* x_in, const float* g_out, const float* g_ld, const float* g_ps_c,
                                            const float* g_lvs_c, const float* packed_w_c, const float* packed_b_c,
                                            const float* film, float* g_in, float* dw1_ws, float* g_film, float* g_sd0,
                                            float* g_bias, int c, int B, int N, int C, int f, int pattern0, float eps, int mode,
                                            void* stream);

static int bwd_points_per_wg(int B, int N) { return (long)B * N >= 2048L * 32 ? 128 : 64; }
static int bwd_grid(int B, int N) {
  const int pts = bwd_points_per_wg(B, N);
  return B * ((N + pts - 1) / pts);
}

static int bwd_dispatch(int var, const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                        const float* packed_b_c, const float* film, float* g_in, float* dw1_ws, float* g_film, float* g_sd0,
                        float* g_bias, const float* g_stats, int c, int B, int N, int C, int f, int pat, float eps, int mode,
                        int K, const GwtfKS& ks, const float* g_ps_c, const float* g_lvs_c, const GwtfCombine& cmb, void* stream) {
  const int kk_steps = (f + 3) / 4;
  const int nb = bwd_points_per_wg(B, N) / 64;
  hipStream_t st = (hipStream_t)stream;
#define GWTF_A x_in, g_out, g_ld, packed_w_c, packed_b_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, g_stats, B, N, C, c, pat, eps, kk_steps, f, mode, K, ks, g_ps_c, g_lvs_c, cmb, st
#define GWTF_V(MB_)                                                                                     \
  switch (var) {                                                                                        \
    case BW_DIRECT: return launch_bwd<MB_, BW_DIRECT>(nb, GWTF_A);                                      \
    case BW_STATS: return launch_bwd<MB_, BW_STATS>(nb, GWTF_A);                                        \
    case BW_LIGHT: return launch_bwd<MB_, BW_LIGHT>(nb, GWTF_A);                                        \
    default: return launch_bwd<MB_, BW_MERGED>(nb, GWTF_A);                                             \
  }
  switch (gwtf_padded_width(f) / 16) {
    case 1: GWTF_V(1)
    case 2: GWTF_V(2)
    case 3: GWTF_V(3)
    case 4: GWTF_V(4)
    case 5: GWTF_V(5)
    case 6: GWTF_V(6)
    default: return GWTF_E_BADARG;       // f > 96: the backward working set (forward + backward records) exceeds the LDS
  }
#undef GWTF_A
#undef GWTF_V
}

extern "C" int gwtf_coupling_backward(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                      const float* packed_b_c, const float* film, float* g_in, float* dw1_ws, float* g_film,
                                      float* g_sd0, float* g_bias, int c, int B, int N, int C, int f, int pattern0, float eps,
                                      int mode, void* stream) {
  return gwtf_coupling_backward_lists(x_in, g_out, g_ld, nullptr, nullptr, packed_w_c, packed_b_c, film, g_in, dw1_ws, g_film,
                                      g_sd0, g_bias, c, B, N, C, f, pattern0, eps, mode, stream);
}

extern "C" int gwtf_coupling_backward_lists(const float* x_in, const float* g_out, const float* g_ld, const float* g_ps_c,
                                            const float* g_lvs_c, const float* packed_w_c, const float* packed_b_c,
                                            const float* film, float* g_in, float* dw1_ws, float* g_film, float* g_sd0,
                                            float* g_bias, int c, int B, int N, int C, int f, int pattern0, float eps, int mode,
                                            void* stream) {
  if (!x_in || !g_out || !g_ld || !packed_w_c || !packed_b_c || !film || !g_in || !dw1_ws || !g_film || !g_sd0 ||
      !g_bias || B <= 0 || N <= 0 || C <= 0 || c < 0 || c >= C || f <= 0 || f > GWTF_MAX_FP_TRAIN || pattern0 < 0 || pattern0 > 5 ||
      (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE))
    return GWTF_E_BADARG;
  GwtfKS ks = {};
  ks.Cper = ks.Ctot = C;
  return bwd_dispatch(BW_DIRECT, x_in, g_out, g_ld, packed_w_c, packed_b_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, nullptr, c,
                      B, N, C, f, (pattern0 + c) % 6, eps, mode, 1, ks, g_ps_c, g_lvs_c, GwtfCombine{}, stream);
}

// K-batched variants (train pipeline, gwtf_train.hip): component k adds k * stride (GwtfKS) to every base pointer
int gwtf_internal_coupling_backward_k(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                      const float* packed_b_c, const float* film, float* g_in, float* dw1_ws, float* g_film,
                                      float* g_sd0, float* g_bias, int c, int K, int B, int N, int f, int pattern0, float eps,
                                      int mode, const GwtfKS& ks, const float* g_ps_c, const float* g_lvs_c, void* stream) {
  return bwd_dispatch(BW_DIRECT, x_in, g_out, g_ld, packed_w_c, packed_b_c, film, g_in, dw1_ws, g_film, g_sd0, g_bias, nullptr, c,
                      B, N, ks.Ctot, f, (pattern0 + c) % 6, eps, mode, K, ks, g_ps_c, g_lvs_c, GwtfCombine{}, stream);
}
// The train pipeline's two passes (BW_LIGHT / BW_MERGED above).  light: only g_film and g_bias are written; merged: g_in, the dW1
// partials and g_sd0, with the statistics path's upstream g_stats [K][2][2][FP] added to dacc.
int gwtf_internal_light_backward_k(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                   const float* film, float* g_film, float* g_bias, int c, int K, int B, int N, int f,
                                   int pattern0, float eps, int mode, const GwtfKS& ks, const float* g_ps_c,
                                   const float* g_lvs_c, const GwtfCombine& cmb, void* stream) {
  return bwd_dispatch(BW_LIGHT, x_in, g_out, g_ld, packed_w_c, packed_w_c /*unused*/, film, g_film /*unused*/, g_film /*unused*/,
                      g_film, g_film /*unused*/, g_bias, nullptr, c, B, N, ks.Ctot, f, (pattern0 + c) % 6, eps, mode, K, ks, g_ps_c,
                      g_lvs_c, cmb, stream);
}
int gwtf_internal_merged_backward_k(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                    const float* packed_b_c, const float* film, const float* g_stats, float* g_in, float* dw1_ws,
                                    float* g_sd0, int c, int K, int B, int N, int f, int pattern0, float eps, int mode,
                                    const GwtfKS& ks, const float* g_ps_c, const float* g_lvs_c, const GwtfCombine& cmb,
                                    void* stream) {
  return bwd_dispatch(BW_MERGED, x_in, g_out, g_ld, packed_w_c, packed_b_c, film, g_in, dw1_ws, g_sd0 /*unused*/, g_sd0,
                      g_sd0 /*unused*/, g_stats, c, B, N, ks.Ctot, f, (pattern0 + c) % 6, eps, mode, K, ks, g_ps_c, g_lvs_c, cmb,
                      stream);
}
int gwtf_internal_stats_backward_k(const float* x_in, const float* g_stats, const float* packed_w_c, const float* packed_b_c,
                                   float* g_in, float* dw1_ws, float* g_sd0, int K, int B, int N, int f, int pattern,
                                   const GwtfKS& ks, void* stream) {
  return bwd_dispatch(BW_STATS, x_in, nullptr, nullptr, packed_w_c, packed_b_c, packed_w_c /*unused*/, g_in, dw1_ws,
                      g_sd0 /*unused*/, g_sd0, g_sd0 /*unused*/, g_stats, 0, B, N, 1, f, pattern, 0.f, GWTF_MODE_INVERSE, K, ks,
                      nullptr, nullptr, GwtfCombine{}, stream);
}

extern "C" int gwtf_stats_backward(const float* x_in, const float* g_stats, const float* packed_w_c,
                                   const float* packed_b_c, float* g_in, float* dw1_ws, float* g_sd0, int B, int N, int f,
                                   int pattern, void* stream) {
  if (!x_in || !g_stats || !packed_w_c || !packed_b_c || !g_in || !dw1_ws || !g_sd0 || B <= 0 || N <= 0 || f <= 0 ||
      f > GWTF_MAX_FP_TRAIN || pattern < 0 || pattern > 5)
    return GWTF_E_BADARG;
  GwtfKS ks = {};
  ks.Cper = ks.Ctot = 1;
  return gwtf_internal_stats_backward_k(x_in, g_stats, packed_w_c, packed_b_c, g_in, dw1_ws, g_sd0, 1, B, N, f, pattern, ks, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// dW1[br][j][i] = sum over the per-workgroup partials the backward kernels leave in the workspace ([partial][2][f][RP], gwtf_dw1.h,
// `passes` consecutive regions of gwtf_dw1_partials(B, N) partials each: coupling path [+ statistics path]); fixed
// summation order -> deterministic.  64 outputs per workgroup x 4 slices of the partial axis, combined through LDS.
namespace {
constexpr int kDw1Stage = gwtf_dw1::kStage;

// stage 1: grid (element tiles of 256, kDw1Stage, K); thousands of workgroups stream the workspace at HBM rate
__global__ __launch_bounds__(256) void dw1_fold_kernel(const float* __restrict__ ws, int n_partials, float* __restrict__ mid,
                                                       int rec, size_t ws_sk) {
  // blockIdx.z = mixture component (K-batched train pipeline)
  gwtf_dw1::fold_block(ws + blockIdx.z * ws_sk, n_partials, mid + blockIdx.z * ws_sk, rec, blockIdx.x, blockIdx.y, threadIdx.x);
}

// stage 2: grid (tiles of 64 outputs, K)
__global__ __launch_bounds__(256) void dw1_reduce_kernel(const float* __restrict__ mid, float* __restrict__ out, int FP, int f,
                                                         size_t branch_stride, size_t ws_sk, size_t out_sk) {
  __shared__ float part[4][64];
  gwtf_dw1::reduce_block(mid + blockIdx.y * ws_sk, out + blockIdx.y * out_sk, f, branch_stride, blockIdx.x, threadIdx.x, part);
}
}  // namespace

extern "C" int gwtf_dw1_partials(int B, int N) { return (B > 0 && N > 0) ? bwd_grid(B, N) : 0; }

extern "C" size_t gwtf_dw1_workspace_floats(int f, int B, int N) {
  return (size_t)gwtf_dw1_partials(B, N) * gwtf_dw1::rec_floats(f);      // [2][f][RP] partials (gwtf_dw1.h)
}

// `workspace`: `passes` regions written by the backward kernels, followed by kDw1Stage records of scratch for the first
// reduction stage (gwtf_dw1_reduce_scratch_floats(f) floats after the last region).
extern "C" size_t gwtf_dw1_reduce_scratch_floats(int f) { return (size_t)kDw1Stage * gwtf_dw1::rec_floats(f); }

// K components: component k's workspace at workspace + k * ws_sk, its gradient blocks at dW1 + k * out_sk
int gwtf_internal_dw1_reduce_k(float* workspace, int passes, float* dW1, size_t branch_stride, int f, int B, int N, int K,
                               size_t ws_sk, size_t out_sk, void* stream) {
  if (!workspace || !dW1 || passes < 1 || f <= 0 || f > GWTF_MAX_FP_TRAIN || B <= 0 || N <= 0 || K <= 0 || branch_stride < (size_t)f * f)
    return GWTF_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int FP = gwtf_padded_width(f), rec = gwtf_dw1::rec_floats(f), n_partials = passes * bwd_grid(B, N);
  float* mid = workspace + (size_t)n_partials * rec;
  hipLaunchKernelGGL(dw1_fold_kernel, dim3((rec + 255) / 256, kDw1Stage, K), dim3(256), 0, st, workspace, n_partials, mid, rec, ws_sk);
  hipLaunchKernelGGL(dw1_reduce_kernel, dim3((gwtf_dw1::rec_floats(f) + 63) / 64, K), dim3(256), 0, st, mid, dW1, FP, f, branch_stride, ws_sk, out_sk);
  return (int)hipGetLastError();
}

extern "C" int gwtf_dw1_reduce(float* workspace, int passes, float* dW1, size_t branch_stride, int f, int B, int N,
                               void* stream) {
  return gwtf_internal_dw1_reduce_k(workspace, passes, dW1, branch_stride, f, B, N, 1, 0, 0, stream);
}
