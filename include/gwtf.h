/* gwtf.h -- C ABI of libgwtf_hip.so: the MI355X (gfx950) implementation of the discrete
 * point-flow decoder hot path of janisgp/go_with_the_flows.
 *
 * The reference has no native code on this path; each entry point below replaces a span of
 * Python/torch code, cited as reference file:line.  Conventions for every function:
 *   - returns 0 on success, otherwise a hipError_t value (or GWTF_E_* below); never throws;
 *   - all pointers are DEVICE pointers to contiguous fp32 (int32 where stated) buffers owned by
 *     the caller; the library allocates nothing and keeps NO mutable global state (no tuning switches, no environment
 *     variables: what a call does is a function of its arguments -- see GWTF_TUNE_* for the per-call tuning word);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and returns immediately:
 *     no internal streams, events or synchronisation (same convention as the reference's own
 *     native ops, lib/metrics/pytorch_structural_losses/src/structural_loss.cpp:35);
 *   - callable from one host thread per process (one process per GPU).
 * Buffer layouts are specified in go_with_the_flows_amd/csrc/gwtf_layout.h.
 */
#ifndef GWTF_H
#define GWTF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GWTF_ABI_VERSION 6
#define GWTF_E_BADARG 10001   /* shape / mode / width outside what the kernels support */
#define GWTF_E_UNSUPPORTED 10002   /* a layer-width list no kernel instantiation was built for */
#define GWTF_MODE_DIRECT 0    /* sampling direction  base -> data (reference models.py:202) */
#define GWTF_MODE_INVERSE 1   /* density direction   data -> base (reference models.py:197) */

/* Per-call tuning word (`tune` arguments; 0 = the library's own choices).  Tests and the tile-calibration tools pass it; results
 * never depend on it beyond fp32 rounding of a different summation order (tests/test_gpu_parity.py pins that). */
#define GWTF_TUNE_DEFAULT 0
#define GWTF_TUNE_POINTS_PER_WAVE(n) ((n) & 0xffff)   /* force 16, 32 or 64 points per wavefront in the forward-sized kernels */
#define GWTF_TUNE_GENERIC_BODY (1 << 30)              /* the run-time-width coupling body instead of the software-pipelined one */
#define GWTF_TUNE_SMALL_LIGHT_TILE (1 << 29)          /* train backward, light pass: 128-point tiles even on large grids */
#define GWTF_TUNE_SINGLE_TILE (1 << 28)               /* no mixed launch (large tiles + a small-tile tail): one tile size per launch */

int gwtf_abi_version(void);
/* Human-readable text for a non-zero return value (static storage). */
const char* gwtf_error_string(int code);
/* Diagnostic: a one-thread kernel on `stream` writes wall_clock64() (100 MHz) to *slot -- a time stamp inside a captured hipGraph. */
int gwtf_diag_stamp(unsigned long long* slot, void* stream);

/* Sizes (in floats) of the buffers the caller must provide. FP = f rounded up to 16. */
int    gwtf_padded_width(int f);
size_t gwtf_raw_coupling_floats(int f, int G);        /* one coupling record of the raw arena   */
size_t gwtf_packed_w_coupling_floats(int f);          /* one coupling of packed stack weights   */
size_t gwtf_packed_film_coupling_floats(int f, int G);/* one coupling of packed FiLM weights    */
size_t gwtf_film_out_floats(int f);                   /* FiLM output per (shape, coupling)      */

/* Weight packer.  Folds eval-mode BatchNorm into the adjacent SharedDot / Linear weights, pads
 * f to FP and lays the weights out in MFMA-fragment / coalesced order.
 * Replaces the per-call parameter reads of nn.BatchNorm1d + SharedDot in
 * lib/networks/flows.py:25-50,60-85 (module construction) as consumed by :95-107.
 *   raw          [C][gwtf_raw_coupling_floats]   parameters + running statistics, direct order
 *   pattern0     warp pattern index of coupling 0 (the raw record stores sd0.weight as the module does, [f][k] with
 *                k = 1 or 2 kept coordinates: the packer needs each coupling's k)
 *   packed_w     [C][gwtf_packed_w_coupling_floats]
 *   packed_film  [C][gwtf_packed_film_coupling_floats]
 *   training     0: fold running statistics (model.eval()); 1: leave the per-shape FiLM BatchNorm
 *                un-folded so gwtf_film_forward takes batch statistics (model.train()). */
int gwtf_pack_weights(const float* raw, float* packed_w, float* packed_film,
                      int C, int f, int G, int pattern0, int training, void* stream);
/* The same for K concatenated stacks of Cper couplings each (the components of a mixture): raw [K][Cper][...], every stack
 * starts again at warp pattern `pattern0`.  training = 2: the train pipeline's packing -- packed_w only (train form), packed_film
 * may be NULL: its FiLM heads read the raw arena in place (gwtf_film_heads_forward). */
int gwtf_pack_weights_k(const float* raw, float* packed_w, float* packed_film,
                        int K, int Cper, int f, int G, int pattern0, int training, void* stream);

/* Per-shape FiLM conditioning for all C couplings: the four Linear->BN->Swish->Linear heads of each
 * coupling applied to the latent g, then a = eps + exp(w(g)), b' = a*c1 + b(g).
 * Replaces T_{mu,logvar}_0_cond_{w,b}(g) and torch.add(eps, torch.exp(.)) in
 * lib/networks/flows.py:100-101,105-106 (modules built at :33-45,68-80).
 *   g        [B][G]
 *   film_out [B][C][gwtf_film_out_floats]
 *   eps      the coupling's `eps` buffer (reference flows.py:21, 1e-6)
 *   training 1: BatchNorm over the B rows uses batch statistics; bn_stats_out (may be NULL)
 *            receives [C][2 branches][2 heads][2][f] = {batch mean, biased batch var}. */
int gwtf_film_forward(const float* g, const float* packed_film, float* film_out, float* bn_stats_out,
                      int B, int G, int C, int f, float eps, int training, void* stream);

/* Differentiable train-mode FiLM heads: the BatchNorm over the B latent rows (batch statistics) + swish that sits between a
 * head's two Linear layers -- T_{mu,logvar}_0_cond_{w,b} = Linear, BatchNorm1d, Swish, Linear (lib/networks/flows.py:33-45,68-80)
 * in train() -- as one kernel per direction; the two Linear layers stay batched library products (autograd.py _film_train).
 *   x      [B][M]   first Linear's outputs of all heads, M = (couplings * 2 branches * 2 heads) * f columns
 *   gamma, beta     BatchNorm weight / bias of head (c, branch x, head h), feature j at base + c*stride_c + x*stride_x + h*stride_h + j
 *                   (views of the raw arena, include/gwtf.h RAW ARENA)
 *   forward : y [B][M] = swish(BN(x)), mean / var (biased) / rstd [M]
 *   backward: gx [B][M], ggamma / gbeta [M] from gy [B][M] */
int gwtf_film_bn_swish_forward(const float* x, const float* gamma, const float* beta, long stride_c, long stride_x, long stride_h,
                               int f, int B, int M, float* y, float* mean, float* var, float* rstd, void* stream);
int gwtf_film_bn_swish_backward(const float* x, const float* gy, const float* gamma, const float* beta, long stride_c, long stride_x,
                                long stride_h, int f, int B, int M, const float* mean, const float* rstd, float* gx, float* ggamma,
                                float* gbeta, void* stream);

/* Fused coupling stack: all C elementary couplings applied to every point, with the log-det
 * accumulation.  Replaces LocalCondRNVPDecoder.forward (lib/networks/decoders.py:61-79) ->
 * CondRealNVPFlow3DTriple.forward (flows.py:150-160) -> CondRealNVPFlow3D.forward (flows.py:95-117)
 * and the `sum(logvars)` of lib/networks/losses.py:14,115.
 *   p        [B][3][N]   input coordinates (data for INVERSE, base samples for DIRECT)
 *   out      [B][3][N]   coordinates after the whole stack (ps[0] for INVERSE, ps[-1] for DIRECT)
 *   logdet   [B][3][N]   sum over the C couplings of logvar (per coordinate; the reference's
 *                        definition of the log-det, NOT including the base logvar0)
 *   ps, mus, logvars     optional (all three NULL, or all three non-NULL) [C][B][3][N]: the
 *                        per-coupling lists the reference returns, slot j = direct-order coupling j
 *   pattern0 warp pattern index of coupling 0 (0 for a decoder / pattern-0 Triple; see gwtf_layout.h)
 *   mode     GWTF_MODE_DIRECT / GWTF_MODE_INVERSE */
int gwtf_stack_forward(const float* p, const float* packed_w, const float* film,
                       float* out, float* logdet, float* ps, float* mus, float* logvars,
                       int B, int N, int C, int f, int pattern0, float eps, int mode, int tune, void* stream);

/* K flow components in ONE launch (the loop over `self.pc_decoder[i]` in Flow_Mixture_Model.decode,
 * lib/networks/flow_mixture.py:163-166).  Component k applies its own C-coupling stack to the points
 * [segments[2k], segments[2k+1]) of every shape (segments == NULL: every component takes all N points).
 *   packed_w [K][C][...]            the K components' packed weights, concatenated
 *   film     [B][K*C][...]          gwtf_film_forward run once on the concatenated FiLM weights with C' = K*C
 *   p        base + k*p_stride_k    (floats; 0: all components read the same clouds)
 *   out, logdet base + k*out_stride_k  (floats; B*3*N for the density path -> [K][B][3][N], the layout
 *                                   gwtf_mixture_nll reads; 0 for the sampling path where the segments partition N)
 *   ps, mus, logvars                optional lists, [K][C][B][3][N] when out_stride_k != 0 else [C][B][3][N]
 *   segments                        HOST array of 2K ints, or NULL;  K <= 64 */
int gwtf_stack_forward_multi(const float* p, const float* packed_w, const float* film,
                             float* out, float* logdet, float* ps, float* mus, float* logvars,
                             const int* segments, int K, int B, int N, int C, int f, int pattern0, float eps,
                             int mode, size_t p_stride_k, size_t out_stride_k, int tune, void* stream);
/* The same stack with the f x f contraction on the EXACT-fp32 matrix instruction (v_mfma_f32_16x16x4_f32, unsplit operands):
 * the arithmetic of the reference's torch.matmul in SharedDot (lib/networks/layers.py:40-45 as called by flows.py:95-117), no
 * operand range limit (csrc/gwtf_stack_exact.hip).  packed_x: gwtf_pack_weights_exact of the same raw arena
 * [K*C][gwtf_packed_x_coupling_floats]; film: the SAME FiLM records gwtf_film_forward wrote for the split kernel.
 *   only_flagged = 0: every tile is computed (on-device A/B of the split-f16 contraction; the fp32-MFMA comparison point).
 *   only_flagged = 1: RE-RUN after gwtf_stack_forward[_multi] with the same arguments -- a workgroup reads its tile's out / logdet,
 *                     returns at once if they are all finite, and otherwise recomputes the tile (all list slots included): points the
 *                     split kernel flagged with NaN because a coordinate left the f16-safe range (|x| > 3e4) come back with the
 *                     reference's finite fp32 values; genuinely non-finite inputs / parameters stay NaN.
 * ps, mus, logvars: all three or none.  tune: GWTF_TUNE_POINTS_PER_WAVE(16 | 32) only. */
size_t gwtf_packed_x_coupling_floats(int f);
int gwtf_pack_weights_exact(const float* raw, const float* packed_film /*eval packing of the same arena: its range exponents*/,
                            float* packed_x, int K, int Cper, int f, int G, int pattern0, void* stream);
int gwtf_stack_forward_exact(const float* p, const float* packed_x, const float* film, float* out, float* logdet, float* ps,
                             float* mus, float* logvars, const int* segments, int K, int B, int N, int C, int f, int pattern0,
                             float eps, int mode, size_t p_stride_k, size_t out_stride_k, int only_flagged, int tune, void* stream);
/* The same pair with a WORK LIST, so that the re-run launch behind a clean pass costs the dispatch of 64 idle workgroups instead of a
 * scan of every tile's flags (7 us -> 2.8 per forward pass of the airplane config):
 *   gwtf_stack_forward_flagging  = gwtf_stack_forward_multi; a wave that flags a point appends {component * B + shape, first point}
 *                                  to worklist[2 + 2 i], i = atomic increment of worklist[0]
 *   gwtf_stack_rerun_flagged     = gwtf_stack_forward_exact(only_flagged = 1) over the listed tiles (all tiles when worklist[0] >
 *                                  GWTF_WORKLIST_CAP); its last workgroup clears worklist[0..1]
 * worklist: GWTF_WORKLIST_INTS ints of device memory, zero before the FIRST use, owned by one stream at a time (the pair keeps it
 * zero between uses). */
#define GWTF_WORKLIST_CAP 2048
#define GWTF_WORKLIST_INTS (2 + 2 * GWTF_WORKLIST_CAP)
int gwtf_stack_forward_flagging(const float* p, const float* packed_w, const float* film, float* out, float* logdet, float* ps,
                                float* mus, float* logvars, const int* segments, int K, int B, int N, int C, int f, int pattern0,
                                float eps, int mode, size_t p_stride_k, size_t out_stride_k, int* worklist, int tune, void* stream);
int gwtf_stack_rerun_flagged(const float* p, const float* packed_x, const float* film, float* out, float* logdet, float* ps, float* mus,
                             float* logvars, const int* segments, int K, int B, int N, int C, int f, int pattern0, float eps, int mode,
                             size_t p_stride_k, size_t out_stride_k, int* worklist, int tune, void* stream);
/* Latent-space loss terms of the training step and their combination with the point NLL (reference lib/networks/losses.py:24-33
 * GaussianFlowNLL, :36-41 GaussianEntropy, :159-170 Flow_Mixture_Loss.forward), one launch each way (csrc/gwtf_latent.hip):
 *   nll [B] per-shape point NLL (gwtf_mixture_nll); z [B][G] = g_prior_samples[0]; mu0, lv0 [G] = the base Gaussian of the prior flow;
 *   flow_lv [n2][B][G] = the prior flow's stacked logvars (g_prior_logvars[1:]); post_lv [B][G] = g_posterior_logvars
 *   workspace: gwtf_latent_loss_workspace_floats(B, G) floats.  out4 = {loss = pw pnll + gw gnll - ew gent, pnll, gnll, gent}.   backward: g_out4 = upstream of the four outputs. */
int gwtf_latent_loss_workspace_floats(int B, int G);       /* device scratch of the forward (block partials; 8-byte aligned) */
int gwtf_latent_loss_forward(const float* nll, const float* z, const float* mu0, const float* lv0, const float* flow_lv,
                             const float* post_lv, float* workspace, float* out4, int B, int G, int n2, float pw, float gw, float ew,
                             void* stream);
int gwtf_latent_loss_backward(const float* g_out4, const float* z, const float* mu0, const float* lv0, float* g_nll, float* g_z,
                              float* g_mu0, float* g_lv0, float* g_flow_lv, float* g_post_lv, int B, int G, int n2, float pw, float gw,
                              float ew, void* stream);
/* Tile plan of a forward launch (what stack_dispatch decides; diagnostic + tests): out[0] = points per wavefront of the main
 * launch, out[1] = its workgroups, out[2] = points per wavefront of the tail launch (0: none), out[3] = its workgroups.
 * The choice minimises resident rounds x the cost of a round of that tile (calibrated, csrc/gwtf_stack.hip tile_cost). */
int gwtf_stack_plan(const int* segments, int K, int B, int N, int f, int tune, int* out4);

/* ---- train-mode (batch-statistic BatchNorm) forward pipeline, reference flows.py:27,30,62,65 under model.train() ----
 * Per coupling, in processing order (inverse: C-1..0):  fold0 -> stats -> fold1 -> apply.  See csrc/gwtf_train.hip.
 * gwtf_pack_weights(training=1) leaves sd1 un-scaled and the sd0 records empty; gwtf_film_forward(training=1)
 * writes RAW FiLM {a, b} as film_out[B][C][2 branches][2][FP] (NOT the eval record) and the FiLM BatchNorm
 * batch statistics.  n_total = number of points the statistics cover (B*N; summed over ranks when sharded). */
/* statistic accumulators are replicated GWTF_STAT_REPLICAS (=64) times to spread atomic contention:
 * moments [64][16] (9 used), ystats [64][2][FP][2]; the fold kernels sum the replicas. */
int gwtf_train_moments(const float* p, float* moments /*pre-zeroed, accumulated*/, int B, int N, void* stream);
int gwtf_train_fold0(const float* raw_c, const float* moments, double n_total, int pattern, float* packed_w_c,
                     float* packed_b_c /*may be NULL: backward record, sd0 section*/, float* bn_batch_c /*[2 branches][4 kinds][2][f]: kind 0 <- {mean, unbiased var} of sd0_bn*/,
                     int f, int G, void* stream);
int gwtf_train_stats(const float* p, const float* packed_w_c, float* ystats /*pre-zeroed, accumulated*/,
                     int B, int N, int f, int pattern, int tune, void* stream);
int gwtf_train_fold1(const float* raw_c, const float* ystats, double n_total, const float* film_raw, float* film_rec,
                     float* bn_batch_c /*kind 1 <- sd1_bn*/, int c, int B, int C, int f, int G, void* stream);
int gwtf_train_apply(const float* p, const float* packed_w, const float* film_rec, float* out, const float* logdet_in,
                     float* logdet, float* ps, float* mus, float* logvars, float* moments_out /*9 or NULL*/,
                     int c, int B, int N, int C, int f, int pattern0, float eps, int mode, int tune, void* stream);

/* The whole single-rank train-mode forward enqueued from C (moments + 4 launches per coupling); workspace sizes
 * are documented at the definition in csrc/gwtf_train.hip.  Result coordinates end in xbuf[(C-1) & 1]. */
int gwtf_train_forward(const float* p, const float* raw, float* packed_w, float* packed_b /*may be NULL*/,
                       const float* film_raw, float* moments,
                       float* ystats, float* bn_batch, float* film_rec, float* xbuf, float* logdet,
                       float* ps, float* mus, float* logvars, int B, int N, int C, int f, int G, int pattern0,
                       float eps, int mode, int tune, void* stream);

/* ---- backward (both directions, BatchNorm as a fixed affine) --------------------------------------------------
 * Autograd of CondRealNVPFlow3D.forward (reference flows.py:95-117 as differentiated by loss.backward(),
 * training.py:54), one coupling per call, in the FOLDED parameters the forward kernel consumes:
 *   W1p [C][2][f][f] = sd1.weight with sd1_bn's scale folded, W0f [C][2][f][2] / c0f [C][2][f] = sd0 with sd0_bn folded,
 *   FiLM record [B][C][6FP+4] = {c, u0, u1} x 2 branches + biases (gwtf_layout.h).
 * gwtf_pack_folded builds the forward record (packed_w) and the backward record (packed_b) from them.
 * gwtf_coupling_backward: x_in = the coupling's input saved by the forward, g_out/g_ld = dL/d(out), dL/d(logdet);
 *   -> g_in [B][3][N]; dw1_ws: per-workgroup partials of dW1p = sum_p dL/dacc(p) relu(sd0)(p)^T, gwtf_dw1_workspace_floats(f,B,N)
 *   floats, summed by gwtf_dw1_reduce; g_film [B][C][2][3][FP] += {dc, du0, du1}; g_sd0 [64][2][3][FP] += {dW0f[:,0], dW0f[:,1], dc0f};
 *   g_bias [64][4] += {db_lv0, db_lv1, db_mu0, db_mu1}   (64 = GWTF_STAT_REPLICAS copies, sum them; all pre-zeroed). */
size_t gwtf_packed_b_coupling_floats(int f);
int gwtf_pack_folded(const float* W1p, const float* W0f, const float* c0f, float* packed_w, float* packed_b,
                     int C, int f, void* stream);
int gwtf_coupling_backward(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                           const float* packed_b_c, const float* film, float* g_in, float* dw1_ws, float* g_film,
                           float* g_sd0, float* g_bias, int c, int B, int N, int C, int f, int pattern0, float eps, int mode,
                           void* stream);

/* The same with gradients that enter through the coupling's own list slots (the reference's forward returns differentiable
 * per-coupling lists, decoders.py:61-79): g_ps_c = dL/d ps[c], g_lvs_c = dL/d logvars[c], each [B][3][N] or NULL. */
int gwtf_coupling_backward_lists(const float* x_in, const float* g_out, const float* g_ld, const float* g_ps_c,
                                 const float* g_lvs_c, const float* packed_w_c, const float* packed_b_c, const float* film,
                                 float* g_in, float* dw1_ws, float* g_film, float* g_sd0, float* g_bias, int c, int B, int N,
                                 int C, int f, int pattern0, float eps, int mode, void* stream);

/* Backward records of the train pipeline: W1T sections from the un-scaled sd1 weights (sd0 sections: gwtf_train_fold0). */
int gwtf_pack_w1t(const float* raw, float* packed_b, int C, int f, int G, void* stream);
/* Backward of ONE coupling of the single-rank train pipeline: light pass (the FiLM-record / bias sums of the coupling path) +
 * fold1 + merged pass (coupling path and statistics path together: everything after dL/dy is linear in it) + fold0 + moments
 * path + dW1 reduction; csrc/gwtf_train.hip, csrc/gwtf_bwd.hip BW_LIGHT / BW_MERGED.  dw1_ws: at least
 * gwtf_dw1_workspace_floats(f, B, N) + gwtf_dw1_reduce_scratch_floats(f) floats.  g_xb is no longer written (one merged pass
 * leaves one gradient buffer, g_xa); the argument stays for ABI stability.  Workspace contract at the definition. */
int gwtf_train_coupling_backward(const float* x_in, const float* g_out, const float* g_ld, const float* raw_c,
                                 const float* packed_w_c, const float* packed_b_c, const float* film_rec,
                                 const float* film_raw, const float* moments_c, const float* ystats_c, float* g_in,
                                 float* g_xa, float* g_xb, float* dw1_ws, float* g_film, float* g_sd0,
                                 float* g_bias, float* g_stats, float* g_mom, float* g_film_raw, float* g_raw_c,
                                 int c, int B, int N, int C, int f, int G, int pattern0, float eps, int mode,
                                 void* stream);

/* Backward of the whole single-rank train-mode stack (the K = 1 case of gwtf_mtrain_backward: host loop in the library, one call
 * per decoder).  Array layouts at GwtfTrainCtx; *final_buf = which half of g_bufs [2][B][3][N] holds dL/dp.  g_xa / g_xb are no
 * longer written (the gradient combine is applied on the fly by the next level's passes); the arguments stay. */
int gwtf_train_backward(const float* p, const float* ps, const float* g_out, const float* g_ld, const float* raw,
                        const float* packed_w, const float* packed_b, const float* film_rec, const float* film_raw,
                        const float* moments, const float* ystats, float* g_bufs, float* g_xa, float* g_xb,
                        float* dw1_ws /*gwtf_mtrain_dw1_floats(f, B, N)*/, float* g_film, float* g_sd0, float* g_bias, float* g_stats,
                        float* g_mom /*[C][16] zero*/, float* g_film_raw, float* g_raw, int* final_buf, int B, int N, int C, int f, int G,
                        int pattern0, float eps, int mode, void* stream);

/* ---- FiLM conditioning heads under autograd (csrc/gwtf_film_train.hip) -----------------------------------------------------
 * Replaces T_*_0_cond_w / T_*_0_cond_b of every coupling (reference lib/networks/flows.py:33-45, 68-80, evaluated at :100-101,
 * 105-106): Linear(G -> f) -> BatchNorm1d over the latent rows -> Swish -> Linear(f -> f); a = eps + exp(scale head), b = shift head.
 * H = 4 KC heads (KC couplings in all: K stacks x C), one workgroup per head, parameters read in place from the raw arena
 * [KC][raw coupling record]; training = 1: batch statistics over the B_all rows (any number <= 65536, walked 128 at a time; all ranks' rows when data parallel),
 * 0: the arena's running statistics.
 *   g [B_all][G]; poison [KC][2] (0 or NaN, added to the scale a: diverged weights reach every output) or NULL
 *   hraw, hn [B_all][H][f] (pre-BatchNorm / post-Swish activations, kept for the backward); stats [3][H][f] = mean, biased var, rstd
 *   film_raw [B][KC][2][2][FP] (rows row0 .. row0 + B of g): {a, b} per branch, the train pipeline's GwtfTrainCtx.film_raw; only the
 *   f valid columns are written (zero the buffer first).
 * Backward: g_film_raw = dL/d film_raw -> g_raw: every FiLM parameter's gradient written (=) at its arena offset (other slots
 * untouched); dhraw [B_all][H][f] scratch; dg_part [gwtf_film_heads_slices(KC, G)][B_all][G]: partial dL/dg, summed by the caller. */
int gwtf_film_heads_slices(int KC, int G);
int gwtf_film_heads_forward(const float* raw, const float* g, const float* poison, float* hraw, float* hn, float* stats,
                            float* film_raw, int KC, int f, int G, int B_all, int row0, int B, float eps, int training,
                            void* stream);
int gwtf_film_heads_backward(const float* raw, const float* g, const float* hraw, const float* hn, const float* stats,
                             const float* film_raw, const float* g_film_raw, float* g_raw, float* dhraw, float* dg_part,
                             int KC, int f, int G, int B_all, int row0, int B, float eps, int training, void* stream);

/* ---- K-batched, phase-split train pipeline (round 2) -------------------------------------------------------------------
 * All K components of a flow mixture (reference flow_mixture.py:163-166: a Python loop over self.pc_decoder) run through
 * every kernel of the train-mode chain together, and the chain is cut exactly where a data-parallel run sums BatchNorm
 * statistics over the ranks (SyncBatchNorm, reference train_ae.py:152): two collectives per depth level and direction.
 * A rank that owns the whole batch calls gwtf_mtrain_forward / gwtf_mtrain_backward; a sharded run calls
 * gwtf_mtrain_phase level by level and all-reduces (sum) the named slab between the phases:
 *     FWD_INIT -> moments[0] (64*16 floats) | per step: FWD_A -> ystats[c] (K*64*2*FP*2) | FWD_B -> moments[step+1] (K*64*16)
 *     per backward step: BWD_A (light pass, fold1) -> g_stats[c] (K*2*2*FP) | BWD_B (merged pass, fold0) -> g_mom[c] (K*48 DOUBLES) | BWD_C
 * with c = step (DIRECT) or C-1-step (INVERSE) going forward, and c = step (INVERSE) or C-1-step (DIRECT) going backward.
 * Every buffer is caller-owned; "zero" = must be zero on entry.  FP = gwtf_padded_width(f), R = 64 statistic replicas. */
#define GWTF_PHASE_FWD_INIT 0
#define GWTF_PHASE_FWD_A 1
#define GWTF_PHASE_FWD_B 2
#define GWTF_PHASE_BWD_A 3
#define GWTF_PHASE_BWD_B 4
#define GWTF_PHASE_BWD_C 5
typedef struct GwtfTrainCtx {
  int K, B, N, C, f, G, pattern0, mode;
  int tune;                  /* GWTF_TUNE_* word for every launch of the pipeline (0 = default) */
  float eps;
  double n_total;            /* points the statistics cover: B*N summed over all ranks */
  const float* p;            /* [B][3][N]            input clouds, shared by the K components */
  const float* raw;          /* [K][C][raw record]   parameters + buffers (gwtf_layout.h GwtfRaw) */
  float* packed_w;           /* [K][C][packed_w]     from gwtf_pack_weights(training=1); fold0 fills the sd0 records */
  float* packed_b;           /* [K][C][packed_b]     from gwtf_pack_w1t; fold0 fills the sd0 sections; NULL: no backward */
  const float* film_raw;     /* [B][K*C][2][2][FP]   raw FiLM {a, b} per shape (batch-statistic FiLM BatchNorm applied) */
  float* film_rec;           /* [B][K*C][gwtf_film_out_floats]  written by fold1, read by apply and the backward */
  float* moments;            /* [C+1][K][R*16]       zero: the R = 64 copies the passes' statistic atomics are spread over */
  float* ystats;             /* [C][K][R*2*FP*2]     zero */
  float* mom_c;              /* [C+1][K][16] or NULL COMPACT moments (9 used) of every level's input: data-parallel runs -- each forward
                              *                       phase then ends with one launch that sums the copies into it, the caller all-reduces
                              *                       it, and every consumer (folds, backward) reads it instead of the copies.
                              *                       Level 0 (the shared input clouds): record [0][0] only */
  float* ys_c;               /* [C][K][2*FP*2] or NULL (both or neither): COMPACT {sum y, sum y^2} per branch and feature, likewise */
  float* bn_batch;           /* [K][C][2][4][2][f]   batch {mean, unbiased var} of sd0_bn (kind 0) and sd1_bn (kind 1) */
  float* xbuf;               /* [2][K][B][3][N]      ping-pong coordinates; result in half gwtf_mtrain_final_forward_half(C) */
  float* logdet;             /* [K][B][3][N] */
  float* ps; float* mus; float* logvars;   /* [K][C][B][3][N] each, or all NULL, or ps alone (required for the backward: ps) */
  /* backward only */
  const float* g_out;        /* [K][B][3][N]  dL/d out */
  const float* g_ld;         /* [K][B][3][N]  dL/d logdet */
  const float* g_ps;         /* [K][C][B][3][N] or NULL: dL/d ps[c], gradients entering through the per-coupling list slots */
  const float* g_lvs;        /* [K][C][B][3][N] or NULL: dL/d logvars[c] */
  float* g_bufs;             /* [2][K][B][3][N]  dL/dp per component ends in half gwtf_mtrain_final_backward_half(C, mode) */
  float* g_xa; float* g_xb;  /* unused (the gradient combine runs inside the next level's passes); kept for layout stability, may be NULL */
  float* dw1_ws;             /* [K][gwtf_mtrain_dw1_floats(f, B, N)] scratch */
  float* g_film;             /* [B][K*C][2][3][FP]   zero */
  float* g_sd0;              /* [C][K][R*2*3*FP]     zero */
  float* g_bias;             /* [C][K][R*4]          zero */
  float* g_stats;            /* [C][K][2*2*FP] */
  float* g_mom;              /* [C][K][16]  zero: the nine moment gradients gM of every level (summed by atomics; all-reduced when data parallel) */
  float* g_film_raw;         /* [B][K*C][2][2][FP]   dL/d film_raw */
  float* g_raw;              /* [K][C][raw record]   zero; receives dW0, dgamma0, dbeta0, dW1, dW2, db2 */
  void* stream;
} GwtfTrainCtx;
/* running = (1 - m) running + m batch for n BatchNorm modules in one launch (torch.nn.BatchNorm1d's train-mode buffer update):
 * table [n][3] = device pointers {running_mean [f], running_var [f], num_batches_tracked (int64, += 1) or 0}; src [n][2][f] =
 * batch {mean, unbiased var} (bn_batch of the train pipeline); momentum [n]. */
int gwtf_bn_running_update(const unsigned long long* table, const float* src, const float* momentum, int n, int f, void* stream);
/* dst[offset_i .. + numel_i) = src_i for n small tensors in one launch: table [n][3] = {src device pointer, offset, numel}
 * (floats).  Builds a stack's raw arena from its parameter / buffer tensors (the host mirror's torch.cat, one launch). */
int gwtf_gather_table(const unsigned long long* table, float* dst, int n, void* stream);
size_t gwtf_mtrain_dw1_floats(int f, int B, int N);
int gwtf_mtrain_phase(const GwtfTrainCtx* ctx, int phase, int step);
int gwtf_mtrain_forward(const GwtfTrainCtx* ctx);
int gwtf_mtrain_backward(const GwtfTrainCtx* ctx);
int gwtf_mtrain_final_forward_half(int C);
int gwtf_mtrain_final_backward_half(int C, int mode);

/* Backward of gwtf_train_stats: g_stats [2][2][FP] = dL/d{sum y, sum y^2} per branch and feature (replicas already
 * summed by the caller) -> g_in (kept coordinates only); dw1_ws: this pass's dW1 partials (as gwtf_coupling_backward);
 * g_sd0 [64][2][3][FP] +=. */
int gwtf_stats_backward(const float* x_in, const float* g_stats, const float* packed_w_c, const float* packed_b_c,
                        float* g_in, float* dw1_ws, float* g_sd0, int B, int N, int f, int pattern, void* stream);

/* The sd1 weight gradient dW1[br][j][i] = sum_p dL/dacc[br][j](p) * h[br][i](p) is accumulated INSIDE the backward kernels
 * (points on the MFMA K axis, csrc/gwtf_bwd.hip); every workgroup leaves a compact [2][f][f] partial in the workspace.
 *   gwtf_dw1_partials(B, N)             partials one backward pass writes
 *   gwtf_dw1_workspace_floats(f, B, N)  floats of one pass's workspace region
 *   gwtf_dw1_reduce_scratch_floats(f)   floats of scratch the reduction needs AFTER the last region of the same buffer
 *   gwtf_dw1_reduce                     fixed-order (deterministic) two-stage sum over `passes` consecutive regions; branch br
 *                                       is written as an [f][f] block at dW1 + br * branch_stride (f*f for a dense [2][f][f];
 *                                       the raw-arena branch size to write a gradient record in place) */
int gwtf_dw1_partials(int B, int N);
size_t gwtf_dw1_workspace_floats(int f, int B, int N);
size_t gwtf_dw1_reduce_scratch_floats(int f);
int gwtf_dw1_reduce(float* workspace, int passes, float* dW1, size_t branch_stride, int f, int B, int N, void* stream);

/* Mixture negative log-likelihood over K flow components.
 * Replaces FlowMixtureNLL.forward (lib/networks/losses.py:88-137; per-component body :112-122 is
 * PointFlowNLL, :11-20).
 *   z, logdet [K][B][3][N]  outputs of gwtf_stack_forward(INVERSE) per component
 *   mu0, lv0  [K][B][3]     base Gaussian of each component (reference models.py:169-193)
 *   logits    [B][K]        un-normalised mixture log-weights
 *   point_lse [B][N]        optional: per-point logsumexp_k(log w_k + log p_k)
 *   nll_shape [B]           per-shape  -sum_n point_lse   (zeroed by the call)
 * The batch mean of nll_shape is the reference's pnll. */
int gwtf_mixture_nll(const float* z, const float* logdet, const float* mu0, const float* lv0,
                     const float* logits, float* point_lse, float* nll_shape,
                     int K, int B, int N, void* stream);

/* Backward of gwtf_mixture_nll: g_nll [B] = dL/d nll_shape; point_lse as returned by the forward.
 * Writes g_z, g_logdet [K][B][3][N]; g_mu0, g_lv0 [K][B][3]; g_logits [B][K] (the last three zeroed by the call). */
int gwtf_mixture_nll_backward(const float* z, const float* logdet, const float* mu0, const float* lv0,
                              const float* logits, const float* point_lse, const float* g_nll, float* g_z,
                              float* g_logdet, float* g_mu0, float* g_lv0, float* g_logits, int K, int B, int N,
                              void* stream);

/* Fused step of the reference's custom Adam / AMSGrad (lib/networks/optimizers.py:15-76; un-scaled decoupled decay
 * p -= wd*p + lr*m_hat/(sqrt(v_hat)+eps), :69-72) over many tensors per launch.  HOST arrays of n_tensors DEVICE
 * pointers; max_exp_avg_sq may be NULL when amsgrad == 0; `step` = 1-based count including this update. */
int gwtf_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   float* const* max_exp_avg_sq, const size_t* numel, int n_tensors, float lr, double beta1, double beta2,
                   float eps, float weight_decay, int step, int amsgrad, void* stream);

/* The same update for ANY number of tensors in one launch, the pointer table in DEVICE memory (a model has ~1200 tensors):
 *   table [n_tensors][5] u64 = {param, exp_avg, exp_avg_sq, max_exp_avg_sq or 0, numel}    grads [n_tensors] u64
 *   chunk_map [n_chunks][2] i32 = {tensor index, chunk index within that tensor}, chunks of gwtf_adam_chunk_elems() elements
 * Only `grads` changes from step to step. */
int gwtf_adam_chunk_elems(void);
int gwtf_adam_step_table(const unsigned long long* table, const unsigned long long* grads, const int* chunk_map, int n_chunks,
                         float lr, double beta1, double beta2, float eps, float weight_decay, int step, int amsgrad,
                         void* stream);

/* Structural losses between point sets -- the reference's only native code (CUDA extension
 * lib/metrics/pytorch_structural_losses, bound in pybind/bind.cpp:9-15).  Point sets are [b][n][3] / [b][m][3].
 *
 * gwtf_nn_distance replaces nn_distance (src/structural_loss.cpp:82-104, nndistance.cu:2-124): squared distance to the
 * nearest point of the other set and the index of the FIRST minimiser, both directions in one launch. */
int gwtf_nn_distance(const float* xyz1, const float* xyz2, float* dist1, int* idx1, float* dist2, int* idx2,
                     int b, int n, int m, void* stream);
/* Replaces nn_distance_grad (structural_loss.cpp:106-123, nndistance.cu:129-169); grad_xyz1/2 are zeroed by the call. */
int gwtf_nn_distance_grad(const float* xyz1, const float* xyz2, const float* grad_dist1, const int* idx1,
                          const float* grad_dist2, const int* idx2, float* grad_xyz1, float* grad_xyz2, int b,
                          int n, int m, void* stream);
/* Replaces ApproxMatch (structural_loss.cpp:22-37, approxmatch.cu:3-182): match [b][m][n] (zeroed by the call),
 * temp [b][2(n+m)] scratch. */
int gwtf_approx_match(const float* xyz1, const float* xyz2, float* match, float* temp, int b, int n, int m,
                      void* stream);
/* ApproxMatch + MatchCost in one schedule that never materialises the (b,m,n) matching: out [b] (zeroed by the call)
 * accumulates sum w * distance level by level.  What match_cost.py:10-23 computes when no gradient is wanted
 * (evaluation_metrics.py:25-30 is its only caller); temp as for gwtf_approx_match. */
int gwtf_emd_cost(const float* xyz1, const float* xyz2, float* temp, float* out, int b, int n, int m, void* stream);
/* Replaces MatchCost (structural_loss.cpp:39-55, approxmatch.cu:184-224): out [b] = sum match * distance. */
int gwtf_match_cost(const float* xyz1, const float* xyz2, const float* match, float* out, int b, int n, int m,
                    void* stream);
/* Replaces MatchCostGrad (structural_loss.cpp:57-75, approxmatch.cu:229-297): d out / d xyz1, d out / d xyz2 (un-scaled
 * by the upstream gradient, as in the reference; match_cost.py:38-45 applies it). */
int gwtf_match_cost_grad(const float* xyz1, const float* xyz2, const float* match, float* grad1, float* grad2,
                         int b, int n, int m, void* stream);

/* PointNet cloud encoder, eval-mode BatchNorm (lib/networks/encoders.py:9-28) fused with the max-pool its caller applies
 * (lib/networks/models.py:127-128).  widths = {3, C0, C1, ..., C_last} (n_widths entries); built instantiations:
 * {3,64,128,256,512} (every shipped config) and {3,64,128,64,128}; others return GWTF_E_UNSUPPORTED / size 0.
 *   raw     per layer l: weight[C_l][C_{l-1}] | bn.weight | bn.bias | bn.running_mean | bn.running_var   (C_l each)
 *   x       [B][3][N];  features [B][C_last][N] or NULL;  pooled [B][C_last] or NULL (zeroed by the call) */
size_t gwtf_encoder_raw_floats(const int* widths, int n_widths);
size_t gwtf_encoder_packed_floats(const int* widths, int n_widths);
int gwtf_encoder_pack(const float* raw, float* packed, const int* widths, int n_widths, void* stream);
int gwtf_encoder_forward(const float* x, const float* packed, float* features, float* pooled, int B, int N,
                         const int* widths, int n_widths, void* stream);

/* PointNet cloud encoder 3-64-128-256-512 under model.train() (batch-statistic BatchNorm1d; SyncBatchNorm = sum the
 * statistic arrays over the ranks between a kernel and its fold), max-pooled, forward AND backward, layer at a time.
 * Replaces, for lib/networks/encoders.py:9-28 + models.py:127-128 inside the training step (training.py:43-54), the chain of
 * library GEMM / batch-norm / elementwise calls and their autograd.  Channel counts C = {3, 64, 128, 256, 512}: layer l maps
 * C[l] -> C[l+1] channels with weight W_l [C[l+1]][C[l]]; kernels exist for layer = 1..3 (layer 0, 3 -> 64, is three FMAs
 * folded into layer 1's prologue).  x is the reference's (B, 3, N) fp32 tensor; the stored activations y_1, y_2 and gradients dA_2, dA_1 are
 * fp32 in tiles of 32 points, [B][ceil(N/32)][C][32] (gwtf_enc_train_act_floats): written and read by these entry points only.
 *   aff     [4][C]  = s, t, mean, rstd of a layer's BatchNorm (a = relu(s y + t));  table0 [64][4] = (s W_0 row, t)
 *   sums    replicated accumulators, zero on entry: the caller sums the 64 replicas (and the ranks) before the fold
 *   bconst  [3][C] + 4: dy = s gm + Q y + R per channel, then {up, down} = the power-of-two scale of the f16-split operand
 * gwtf_enc_train_supported(widths, n) -> 1 when this width list has kernels. */
int gwtf_enc_train_supported(const int* widths, int n_widths);
size_t gwtf_enc_train_units_floats(int layer);
/* floats of one stored activation / gradient array of `channels` channels.  y_1, y_2, dA_2, dA_1 are internal to the pipeline and live in
 * tiles of 32 points, [B][ceil(N / 32)][channels][32] + one spare tile (a wave / a k-step takes 32 points x all channels: one contiguous block), NOT in the
 * reference's (B, C, N); only x (B, 3, N) and the pooled (B, 512) outputs keep the reference's layouts. */
size_t gwtf_enc_train_act_floats(int B, int channels, int N);
/* W [C[l+1]][C[l]] -> MFMA fragment images of W (forward) and W^T (backward), gwtf_enc_train_units_floats(layer) floats each */
int gwtf_enc_train_pack(const float* W, float* units_fwd, float* units_bwd, int layer, void* stream);
/* the same for layers 1, 2, 3 from ONE launch (W_l, forward images uf_l, backward images ub_l) */
int gwtf_enc_train_pack_all(const float* W1, const float* W2, const float* W3, float* uf1, float* ub1, float* uf2, float* ub2,
                            float* uf3, float* ub3, void* stream);
/* mom [64][12] += {sum x (3), sum x x^T (xx xy xz yy yz zz), -} over this rank's points */
int gwtf_enc_train_xmoments(const float* x, float* mom, int B, int N, void* stream);
/* layer 0: batch statistics of y_0 = W_0 x from the summed moments mom12; updates the running statistics (NULL: skip) */
int gwtf_enc_train_fold0(const float* mom12, double n_total, const float* W0, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float* aff, float* table0, void* stream);
/* sums [2][C[layer+1]] = sum y, sum y^2 -> aff of that layer's BatchNorm; running statistics updated (momentum, unbiased var).
 * aff_prev = the aff of the layer below: non-finite statistics there or here poison this whole aff with NaN (NaN propagation) */
int gwtf_enc_train_fold(const float* sums, int layer, double n_total, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float* aff, const float* aff_prev,
                        void* stream);
/* y_out (C[layer+1] channels, tiled) = W_layer . relu(s in + t); in = x and in_tab = table0 for layer 1, else y_{layer-1} and its aff.
 * sums [64][2][C[layer+1]] += {sum y, sum y^2}; ymax[0] = max |y| (bit pattern max, zero on entry).
 * layer 3 stores no y: kmax / kmin [B][512] (zero on entry) receive the 64-bit arg-max (channels with BatchNorm weight gamma3 >= 0)
 * or arg-min (gamma3 < 0) key of y_3 per (shape, channel) -- all the max-pool needs, because BatchNorm + ReLU is monotone in y per
 * channel, rising or falling with the sign of gamma3 (the other array's entry stays zero; y_out may be NULL; kmax / kmin / gamma3
 * NULL for layers 1, 2) */
int gwtf_enc_train_forward(int layer, const float* in, const float* in_tab, const float* units, float* y_out, float* sums,
                           float* ymax, unsigned long long* kmax, unsigned long long* kmin, const float* gamma3, int B, int N,
                           void* stream);
/* pooled (B,512) = max_n relu(s y3 + t) from the keys (max or min by the sign of s), amax = its first arg-max, ystar = y3 there;
 * NaN where a statistic or the extreme is not finite */
int gwtf_enc_train_pool(const unsigned long long* kmax, const unsigned long long* kmin, const float* aff3, float* pooled, int* amax,
                        float* ystar, int B, int N, void* stream);
/* gp = g_pooled where pooled > 0; sums [2][512] = {sum gp, sum gp yhat*}; gmax[0] = max |gp| (zero on entry) */
int gwtf_enc_train_top(const float* g_pooled, const float* pooled, const float* ystar, const float* aff3, float* gp, float* sums,
                       float* gmax, int B, void* stream);
/* layer = 0..3 (BatchNorm of y_layer): sums [2][C[layer+1]] = {sum gm, sum gm yhat} over all points and ranks -> bconst;
 * gmax / ymax (may be NULL): the maxima the operand scale is derived from */
int gwtf_enc_train_bwd_consts(const float* sums, int layer, double n_total, const float* gamma, const float* aff,
                              const float* gmax, const float* ymax, float* bconst, void* stream);
/* Top layer: with dy_3 = s gm_3 + Q y_3 + R and y_3 = W_3 a_2,  dL/da_2(p) = M a_2(p) + v + (rows of the arg-max points),
 * M = W_3^T diag(Q) W_3, v = W_3^T R (two small library GEMMs on the caller's side).
 *   gwtf_enc_train_pack_matrix   fragment images of M * 2^k (rows x kdim row-major, here 256 x 256)
 *   gwtf_enc_train_top_scatter   coef [B][512] (times scale [512] when given: coef = gp, scale = s) -> slot_of [B][N] (row of a point, or -1) and
 *                                extra [B][512][256]: row r of shape b = sum of coef[b][c] W_3[c][:] over the channels whose
 *                                arg-max is that point (only the used rows are written); tables: B (2 * 512 + 2) ints of scratch
 *   gwtf_enc_train_backward_top  dA2 (256 channels, tiled) = (M a_2 + v + extra) masked by a_2 > 0; mconst = v [256] | {2^-k};
 *                                sums [64][3][256] += {sum gm_2, sum gm_2 yhat_2, sum a_2}; gmax2[0] = max |dA2|;
 *                                a2rows [B][512][256] (out): row slot_of[b][n] of shape b = a_2(b, :, n) of every arg-max point n,
 *                                point-major (the kernel has them in registers; other rows are not written) */
int gwtf_enc_train_pack_matrix(const float* W, float* units, int rows, int kdim, void* stream);
int gwtf_enc_train_top_scatter(const float* coef, const float* scale, const int* amax, const float* W3, float* extra, int* slot_of,
                               int* tables, int B, int N, void* stream);
int gwtf_enc_train_backward_top(const float* y2, const float* aff2, const float* units_m, const float* mconst, const float* extra,
                                const int* slot_of, float* dA2, float* sums, float* gmax2, float* a2rows, int B, int N, void* stream);
/* layer = 1, 2: dA_prev (C[layer] channels, tiled) = (W_layer^T dy_layer) masked by a_{layer-1} > 0 (not stored for layer 1), dy from
 * (y_l, up_g = masked dL/da_layer).  y_prev / aff_prev: y_{layer-1} and its aff (layer 1: x, aff_0, and w0 = raw W_0).
 * sums [64][2 (layer 1: 5)][C[layer]] += {sum gm, sum gm yhat (layer 1: , sum gm x_d)} of the layer below;
 * gmax_prev[0] = max |dA_prev| */
int gwtf_enc_train_backward(int layer, const float* y_l, const float* up_g, const float* bconst, const float* units_bwd,
                            const float* y_prev, const float* aff_prev, const float* w0, float* dA_prev, float* sums,
                            float* gmax_prev, int B, int N, void* stream);
/* layer = 1, 2: dW (C[layer+1], C[layer]) = sum over this rank's points of dy_layer a_{layer-1}^T; partials: scratch of
 * gwtf_enc_train_dw_partial_floats floats; tab_prev = aff_{layer-1} (layer 1: table0, y_prev = x).  N % 4 == 0. */
size_t gwtf_enc_train_dw_partial_floats(int layer, int B, int N);
int gwtf_enc_train_dw(int layer, const float* y_l, const float* up_g, const float* bconst, const float* y_prev,
                      const float* tab_prev, float* partials, float* dW, int B, int N, void* stream);
/* layer 3 through the Gram matrix: gram (256,256) = sum_p a_2 a_2^T, S (512,256)[c] = sum_b gp[b][c] a_2(b, amax[b][c]) (read from
 * a2rows / slot_of as gwtf_enc_train_backward_top and gwtf_enc_train_top_scatter left them: call those first); then
 * dW_3 = s (.) S + Q (.) (W_3 gram) + R (x) sum_p a_2  (s, Q, R = bconst of layer 3; sum_p a_2 = row 2 of the sums that
 * gwtf_enc_train_backward_top accumulates). */
int gwtf_enc_train_dw3(const float* gp, const int* amax, const int* slot_of, const float* a2rows, const float* y2, const float* aff2,
                       float* partials, float* gram, float* S, int B, int N, void* stream);
/* The small dense algebra between those kernels (csrc/gwtf_encoder_glue.hip), a launch or two each instead of chains of library calls:
 *   gwtf_stat_compact          out [n] = sum of the `replicas` copies of slab [replicas][n], fixed order
 *   gwtf_enc_train_mform       bconst3 = {s, Q, R} [3][C4] of layer 3 -> units_m = fragment images of M 2^k (M = W_3^T diag(Q) W_3,
 *                              k = 8 - floor(log2 max|M|)), mconst [C3 + 4] = {W_3^T R, 2^-k, 0, 0, 0}: what gwtf_enc_train_backward_top
 *                              reads.  workspace: gwtf_enc_train_mform_workspace_floats(C3) floats.  C3 % 32 == 0, C4 % 64 == 0.
 *   gwtf_enc_train_dw3_finish  dW_3 [C4][C3] = s (.) S + Q (.) (W_3 gram) + R (x) a2sum   (a2sum [C3] = sum_p a_2)
 *   gwtf_enc_train_dw0_finish  dW_0 [C1][3] = s (.) red5[2:5]^T + Q (.) (W_0 Mxx) + R (x) m[:3]; bconst0 = {s, Q, R} [3][C1] of layer 0,
 *                              red5 [5][C1] = the compact sums of gwtf_enc_train_backward(layer 1), mom12 = this rank's coordinate
 *                              moments (gwtf_enc_train_xmoments, compact) */
int gwtf_stat_compact(const float* slab, float* out, int replicas, int n, void* stream);
size_t gwtf_enc_train_mform_workspace_floats(int C3);
int gwtf_enc_train_mform(const float* W3, const float* bconst3, float* workspace, float* units_m, float* mconst, int C3, int C4,
                         void* stream);
int gwtf_enc_train_dw3_finish(const float* bconst3, const float* S, const float* W3, const float* gram, const float* a2sum, float* dW3,
                              int C3, int C4, void* stream);
int gwtf_enc_train_dw0_finish(const float* bconst0, const float* red5, const float* W0, const float* mom12, float* dW0, int C1,
                              void* stream);

/* Global prior flow on the shape latent: the whole GlobalRNVPDecoder (lib/networks/decoders.py:7-38; RealNVPFlowCouple /
 * RealNVPFlow, flows.py:163-243) as ONE launch per direction -- forward (the lists the reference returns) and backward
 * (every parameter, the input, gradients entering through any gs[j] / logvars[j] slot), eval- or train-mode BatchNorm.
 *   raw       gwtf_prior_raw_floats(n_flows, G, F) floats: per elementary flow j (2 per couple), branch mu then logvar:
 *             mlp0.weight[F][Gk] | bn.weight[F] | bn.bias[F] | bn.running_mean[F] | bn.running_var[F] | mlp1.weight[Gw][F] |
 *             mlp1.bias[Gw]     (record j starts at gwtf_prior_raw_offset(n_flows, G, F, j))
 *   g         [B][G], B <= 128;   gs, mus, logvars  [2 n_flows][B][G] direct-ordered lists (decoders.py:24-36)
 *   workspace gwtf_prior_workspace_floats(B, G, F) floats of scratch
 *   bn_stats  [2 n_flows][2 branches][2][F] = batch {mean, biased var} of every BatchNorm (train; may be NULL)
 *   backward: g_gs / g_logvars [2 n_flows][B][G] = dL/d gs[j], dL/d logvars[j] (either may be NULL); g_raw (zero on entry)
 *             receives the parameter gradients in the raw layout; g_g [B][G] = dL/d g. */
size_t gwtf_prior_raw_floats(int n_flows, int G, int F);
size_t gwtf_prior_raw_offset(int n_flows, int G, int F, int j);
size_t gwtf_prior_workspace_floats(int B, int G, int F);
int gwtf_prior_forward(const float* g, const float* raw, float* gs, float* mus, float* logvars, float* workspace,
                       float* bn_stats, int n_flows, int B, int G, int F, float eps, int mode, int training, void* stream);
int gwtf_prior_backward(const float* g, const float* raw, const float* gs, const float* mus, const float* logvars,
                        const float* g_gs, const float* g_logvars, float* workspace, float* g_raw, float* g_g,
                        int n_flows, int B, int G, int F, float eps, int mode, int training, void* stream);

/* Per-shape MLP heads: FeatureEncoder / WeightsEncoder (lib/networks/encoders.py:31-89) = n_layers x [Linear(no bias) ->
 * BatchNorm1d -> Swish], then Linear(bias) heads (mu, logvar), the mixture-weight head ending in log_softmax (:87-91) -- the
 * modules g_posterior, p_prior and mixture_weights_encoder of the model (models.py:51-59, flow_mixture.py:28-32).  ONE LAYER per
 * call, forward and backward; a workgroup owns 16 output columns and all B <= 128 rows, so BatchNorm's column statistics are local.
 *   x [B][Din], W [Dout][Din], bias / gamma / beta [Dout] (each may be NULL), out [B][Dout]
 *   bn_mode  0: no BatchNorm   1: batch statistics; running_mean / running_var (may be NULL) get the momentum update with the
 *            unbiased batch variance applied `bn_updates` times, *num_batches_tracked += bn_updates   2: running statistics
 *   act      0: none   1: swish   2: log_softmax over the columns (Dout <= 16)
 *   ypre [B][Dout] = x W^T (the backward's input), stats [3][Dout] = mean, biased variance, 1/sqrt(var + bn_eps) used
 * backward: g_out [B][Dout] -> g_y [B][Dout] (scratch: dL/d(x W^T)), g_x [B][Din] (NULL: not wanted; accumulate_g_x != 0: added
 *   to what is there -- two heads on one trunk), g_W [Dout][Din], g_bias / g_gamma / g_beta [Dout] (each may be NULL).
 * gwtf_head_layer_supported -> 1 when the kernels cover (B, Din, Dout, act). */
int gwtf_head_layer_supported(int B, int Din, int Dout, int act);
int gwtf_head_layer_forward(const float* x, const float* W, const float* bias, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float bn_eps,
                            int bn_mode, int bn_updates, int act, float* ypre, float* stats, float* out, int B, int Din, int Dout,
                            void* stream);
int gwtf_head_layer_backward(const float* x, const float* W, const float* bias, const float* gamma, const float* beta,
                             const float* ypre, const float* stats, const float* out, const float* g_out, int bn_mode, int act,
                             float* g_y, float* g_x, int accumulate_g_x, float* g_W, float* g_bias, float* g_gamma, float* g_beta,
                             int B, int Din, int Dout, void* stream);
/* The two plain Linear heads of a FeatureEncoder (mu and logvar: y = x W^T + bias, no BatchNorm, no activation; reference
 * encoders.py:55-60) on the same input, one launch forward and two backward where two calls of gwtf_head_layer_* are two and four:
 * ypre_* / out_* [B][Dout_*] as gwtf_head_layer_forward leaves them; g_y_* [B][Dout_*] scratch; g_x [B][Din] (may be NULL) = the sum of
 * both heads' input gradients; g_W_* [Dout_*][Din], g_bias_* [Dout_*] (each may be NULL). */
int gwtf_head_pair_forward(const float* x, const float* Wa, const float* bias_a, const float* Wb, const float* bias_b, float* ypre_a,
                           float* out_a, float* ypre_b, float* out_b, int B, int Din, int Dout_a, int Dout_b, void* stream);
int gwtf_head_pair_backward(const float* x, const float* Wa, const float* bias_a, const float* Wb, const float* bias_b,
                            const float* ypre_a, const float* out_a, const float* ypre_b, const float* out_b, const float* g_out_a,
                            const float* g_out_b, float* g_y_a, float* g_y_b, float* g_x, float* g_Wa, float* g_bias_a, float* g_Wb,
                            float* g_bias_b, int B, int Din, int Dout_a, int Dout_b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GWTF_H */
