// gwtf_train.hip -- the small kernels of the train-mode (batch-statistic BatchNorm) forward pipeline.
//
// In model.train() the four per-point BatchNorm layers of a coupling normalise with statistics over ALL B*N
// points (reference flows.py:27,30,62,65; nn.BatchNorm1d in training mode), so the stack cannot run as one
// launch: coupling c+1's statistics depend on coupling c's output.  Per coupling the host enqueues
//     fold0  ->  stats pass (gwtf_train_stats)  ->  fold1  ->  apply pass (gwtf_train_apply)
//   fold0: sd0_bn statistics are ANALYTIC in the first and second moments of the kept coordinates
//          (y0 = W0 x is linear: mean = W0 E[x], var = W0 Cov(x) W0^T), which the previous apply pass
//          accumulated; folds them into the sd0 record of the packed weights and emits the running-stat update.
//   fold1: sd1_bn mean/var from the stats pass; combines them with the raw FiLM (a, b) of every shape into the
//          record the forward kernel consumes:  relu(a*(s1*y + c1) + b) * W2 = relu(y + c') * (W2*a*s1),
//          c' = -mean1 + b/(a*s1),  s1 = 1/sqrt(var1 + eps)   (sd1 weights stay un-scaled in train mode).
// Everything here is O(f) or O(B*f) work: latency, not throughput.
#include <hip/hip_runtime.h>
#include "gwtf_layout.h"
#include "gwtf_device.h"
#include "gwtf_dw1.h"
#include "../../include/gwtf.h"

using namespace gwtf_dev;   // arrival ticket, moment-gradient helpers

namespace {

// 9 moments {Sx0,Sx1,Sx2,Sx0x0,Sx0x1,Sx0x2,Sx1x1,Sx1x2,Sx2x2} of a (B,3,N) cloud
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ p, float* __restrict__ mom, int B, int N) {
  const int b = blockIdx.y;
  float mv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float x0 = p[((size_t)b * 3 + 0) * N + n], x1 = p[((size_t)b * 3 + 1) * N + n], x2 = p[((size_t)b * 3 + 2) * N + n];
    mv[0] += x0; mv[1] += x1; mv[2] += x2;
    mv[3] = fmaf(x0, x0, mv[3]); mv[4] = fmaf(x0, x1, mv[4]); mv[5] = fmaf(x0, x2, mv[5]);
    mv[6] = fmaf(x1, x1, mv[6]); mv[7] = fmaf(x1, x2, mv[7]); mv[8] = fmaf(x2, x2, mv[8]);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mv[i] += __shfl_down(mv[i], off);
  }
  if ((threadIdx.x & 63) == 0) {
    float* rep = mom + ((blockIdx.x + blockIdx.y) % GWTF_STAT_REPLICAS) * 16;
#pragma unroll
    for (int i = 0; i < 9; ++i) atomicAdd(&rep[i], mv[i]);
  }
}

// One workgroup, thread = (branch, feature).  bn_batch[branch][kind 0][2][f] <- {mean, unbiased var}.
template <int NR /* copies of the moment record to add: 1 = the compact record */>
__global__ void fold0_kernel(const float* __restrict__ raw_c, const float* __restrict__ mom_rep, double n_total, int pat,
                             float* __restrict__ pw_c, float* __restrict__ pb_c, float* __restrict__ bn_batch, int f, int G,
                             int FP, const GwtfKS ks) {
  const int t = threadIdx.x;
  raw_c += blockIdx.x * ks.raw;      // blockIdx.x = mixture component (K-batched pipeline); strides 0 for a single stack
  mom_rep += blockIdx.x * ks.mom;
  pw_c += blockIdx.x * ks.pw;
  if (pb_c) pb_c += blockIdx.x * ks.pb;
  bn_batch += blockIdx.x * ks.bn;
  __shared__ float mom[9];
  if (t < 9) {
    float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) sacc += mom_rep[r * 16 + t];
    mom[t] = sacc;
  }
  __syncthreads();
  __shared__ float s_sd0[2][3][GWTF_MAX_FP];     // folded sd0 of both branches: the abs-form columns below need every feature's
  const bool on = t < 2 * f;
  const int br = on ? t / f : 0, j = on ? t % f : 0;
  const GwtfRaw R(f, G);
  const GwtfPackW P(FP);
  const float* rb = raw_c + (size_t)br * R.branch_size();
  if (on) {
  int k0, k1, w0, w1;
  gwtf_pattern_dims(pat, &k0, &k1, &w0, &w1);
  // moments in double: Cov = E[xx] - E[x]E[x] cancels
  const double e0 = mom[k0] / n_total, e1 = k1 >= 0 ? mom[k1] / n_total : 0.0;
  const double c00 = mom[mom2_index(k0, k0)] / n_total - e0 * e0;
  const double c11 = k1 >= 0 ? mom[mom2_index(k1, k1)] / n_total - e1 * e1 : 0.0;
  const double c01 = k1 >= 0 ? mom[mom2_index(k0 < k1 ? k0 : k1, k0 < k1 ? k1 : k0)] / n_total - e0 * e1 : 0.0;
  const int kk = gwtf_pattern_kept(pat);
  const double wa = rb[R.sd0_w(j, 0, kk)], wb = kk > 1 ? rb[R.sd0_w(j, 1, kk)] : 0.0;
  const double mean = wa * e0 + wb * e1;
  double var = wa * wa * c00 + 2.0 * wa * wb * c01 + wb * wb * c11;
  if (var < 0.0) var = 0.0;
  const float* bn = rb + R.bn0();
  const float s = bn[j] / sqrtf((float)var + GWTF_BN_EPS);
  float* sd0 = pw_c + P.sd0(br) + (size_t)(j / 32) * 96 + (size_t)((j % 32) % 4) * 24 + (j % 32) / 4;
  sd0[0] = (float)wa * s;
  sd0[8] = (float)wb * s;
  sd0[16] = bn[f + j] - (float)mean * s;
  if (pb_c) {   // backward record: sd0 parameters in natural feature order (csrc/gwtf_bwd.hip)
    float* n4 = pb_c + GwtfPackB(FP).sd0n(br) + (size_t)j * 4;
    n4[0] = sd0[0];
    n4[1] = sd0[8];
    n4[2] = sd0[16];
    n4[3] = 0.f;
  }
  float* bb = bn_batch + ((size_t)br * 4 + 0) * 2 * f;
  bb[j] = (float)mean;
  bb[f + j] = (float)(var * (n_total / (n_total > 1.0 ? n_total - 1.0 : 1.0)));
  s_sd0[br][0][j] = sd0[0]; s_sd0[br][1][j] = sd0[8]; s_sd0[br][2][j] = sd0[16];
  }
  if (!gwtf_abs_form(f)) return;
  // ABS FORM (gwtf_layout.h): this level's columns 1/2 W1 {wa, wb, c0} into the last slot pair of the merged k-step's image (the
  // train packer left it empty and stored the sd1 weights halved).  Thread = (k third, branch, output row): 3 x 2 f <= 240 threads
  // share a row's three sums (the kernel is one block on the level's dependency chain: latency is what counts).
  __shared__ double s_col[3][2][3][48];
  __syncthreads();
  {
    const int part = t / (2 * f), tt = t % (2 * f), pbr = tt / f, pjo = tt % f;
    if (part < 3) {
      const float* wr = raw_c + (size_t)pbr * R.branch_size() + R.sd1_w() + (size_t)pjo * f;
      const int k0_ = part * ((f + 2) / 3), k1_ = min(f, k0_ + (f + 2) / 3);
      double a = 0.0, b = 0.0, c = 0.0;
      for (int k = k0_; k < k1_; ++k) {
        const double w = wr[k];
        a += w * (double)s_sd0[pbr][0][k];
        b += w * (double)s_sd0[pbr][1][k];
        c += w * (double)s_sd0[pbr][2][k];
      }
      s_col[part][pbr][0][pjo] = a; s_col[part][pbr][1][pjo] = b; s_col[part][pbr][2][pjo] = c;
    }
  }
  __syncthreads();
  if (!on) return;
  const int jo = j, MB = FP / 16;
  const double Ca = s_col[0][br][0][jo] + s_col[1][br][0][jo] + s_col[2][br][0][jo];
  const double Cb = s_col[0][br][1][jo] + s_col[1][br][1][jo] + s_col[2][br][1][jo];
  const double Cc = s_col[0][br][2][jo] + s_col[1][br][2][jo] + s_col[2][br][2][jo];
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  for (int q = 0; q < 4; ++q) {
    float e0, e1;
    gwtf_abs_cols(q, (float)(0.5 * Ca), (float)(0.5 * Cb), (float)(0.5 * Cc), &e0, &e1);
    const h2 pk = {(_Float16)e0, (_Float16)e1};
    pw_c[P.a16(br) + ((size_t)(1 * MB + jo / 16) * 2 + 0) * 256 + (size_t)(q * 16 + jo % 16) * 4 + 3] = __builtin_bit_cast(float, pk);
  }
}

// grid = B workgroups, thread = (branch, feature).  Workgroup 0 also emits the running-stat update of sd1_bn.
template <int NR /* copies of ystats to add: 1 = the compact record */>
__global__ void fold1_kernel(const float* __restrict__ raw_c, const float* __restrict__ ystats, double n_total,
                             const float* __restrict__ film_raw, float* __restrict__ film_rec,
                             float* __restrict__ bn_batch, int c, int C, int f, int G, int FP, const GwtfKS ks) {
  const int t = threadIdx.x, b = blockIdx.x;
  if (t >= 2 * FP) return;
  raw_c += blockIdx.y * ks.raw;      // blockIdx.y = mixture component
  ystats += blockIdx.y * ks.ys;
  bn_batch += blockIdx.y * ks.bn;
  c += blockIdx.y * ks.Cper;         // FiLM-side arrays: [shape][Ctot][...]
  C = ks.Ctot;
  const int br = t / FP, j = t % FP;
  const GwtfRaw R(f, G);
  const size_t FS = gwtf_film_out_size(FP);
  float* rec = film_rec + ((size_t)b * C + c) * FS + (size_t)br * 3 * FP + j;
  if (j >= f) {
    rec[0] = rec[FP] = rec[2 * FP] = 0.f;
    return;
  }
  const float* rb = raw_c + (size_t)br * R.branch_size();
  float ys = 0.f, yq = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    ys += ystats[(size_t)r * (2 * FP * 2) + (br * FP + j) * 2];
    yq += ystats[(size_t)r * (2 * FP * 2) + (br * FP + j) * 2 + 1];
  }
  const double mean = ys / n_total;
  double var = yq / n_total - mean * mean;
  if (var < 0.0) var = 0.0;
  const float s1 = 1.0f / sqrtf((float)var + GWTF_BN_EPS);
  const float* fr = film_raw + (((size_t)b * C + c) * 2 + br) * 2 * FP;
  const float a = fr[j], bsh = fr[FP + j];
  rec[0] = -(float)mean + bsh / (a * s1);
  rec[FP] = rb[R.sd2_w() + j] * a * s1;
  rec[2 * FP] = rb[R.sd2_w() + f + j] * a * s1;
  if (j < 2) film_rec[((size_t)b * C + c) * FS + 6 * FP + 2 * br + j] = rb[R.sd2_b() + j];
  if (b == 0) {
    float* bb = bn_batch + ((size_t)br * 4 + 1) * 2 * f;
    bb[j] = (float)mean;
    bb[f + j] = (float)(var * (n_total / (n_total > 1.0 ? n_total - 1.0 : 1.0)));
  }
}

}  // namespace

extern "C" int gwtf_train_moments(const float* p, float* moments, int B, int N, void* stream) {
  if (!p || !moments || B <= 0 || N <= 0) return GWTF_E_BADARG;
  const int bx = (N + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(moments_kernel, dim3(bx < 1 ? 1 : bx, B), dim3(256), 0, (hipStream_t)stream, p, moments, B, N);
  return (int)hipGetLastError();
}

// internal K-batched pieces defined in gwtf_stack.hip / gwtf_bwd.hip
int gwtf_internal_stats_k(const float* p, const float* packed_w_c, float* ystats, int K, int B, int N, int f, int pattern,
                          size_t p_sk, size_t pw_sk, size_t ys_sk, int tune, void* stream);
int gwtf_internal_apply_k(const float* p, const float* packed_w, const float* film, float* out, const float* logdet_in,
                          float* logdet, float* ps, float* mus, float* logvars, float* moments_out, size_t moments_stride_k,
                          int c, int K, int B, int N, int C, int f, int pattern0, float eps, int mode, size_t p_stride_k,
                          size_t out_stride_k, int tune, void* stream);
int gwtf_internal_coupling_backward_k(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                      const float* packed_b_c, const float* film, float* g_in, float* dw1_ws, float* g_film,
                                      float* g_sd0, float* g_bias, int c, int K, int B, int N, int f, int pattern0, float eps,
                                      int mode, const GwtfKS& ks, const float* g_ps_c, const float* g_lvs_c, void* stream);
int gwtf_internal_stats_backward_k(const float* x_in, const float* g_stats, const float* packed_w_c, const float* packed_b_c,
                                   float* g_in, float* dw1_ws, float* g_sd0, int K, int B, int N, int f, int pattern,
                                   const GwtfKS& ks, void* stream);
int gwtf_internal_light_backward_k(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                   const float* film, float* g_film, float* g_bias, int c, int K, int B, int N, int f,
                                   int pattern0, float eps, int mode, const GwtfKS& ks, const float* g_ps_c,
                                   const float* g_lvs_c, const GwtfCombine& cmb, void* stream);
int gwtf_internal_merged_backward_k(const float* x_in, const float* g_out, const float* g_ld, const float* packed_w_c,
                                    const float* packed_b_c, const float* film, const float* g_stats, float* g_in, float* dw1_ws,
                                    float* g_sd0, int c, int K, int B, int N, int f, int pattern0, float eps, int mode,
                                    const GwtfKS& ks, const float* g_ps_c, const float* g_lvs_c, const GwtfCombine& cmb,
                                    void* stream);
int gwtf_internal_dw1_reduce_k(float* workspace, int passes, float* dW1, size_t branch_stride, int f, int B, int N, int K,
                               size_t ws_sk, size_t out_sk, void* stream);

namespace {
GwtfKS single_ks(int C) {
  GwtfKS ks = {};
  ks.Cper = ks.Ctot = C;
  return ks;
}
}  // namespace

extern "C" int gwtf_train_fold0(const float* raw_c, const float* moments, double n_total, int pattern, float* packed_w_c,
                                float* packed_b_c, float* bn_batch_c, int f, int G, void* stream) {
  if (!raw_c || !moments || !packed_w_c || !bn_batch_c || f <= 0 || f > GWTF_MAX_FP_TRAIN || G <= 0 || pattern < 0 ||
      pattern > 5 || n_total < 1.0)
    return GWTF_E_BADARG;
  hipLaunchKernelGGL(fold0_kernel<GWTF_STAT_REPLICAS>, dim3(1), dim3(2 * GWTF_MAX_FP), 0, (hipStream_t)stream, raw_c, moments, n_total,
                     pattern, packed_w_c, packed_b_c, bn_batch_c, f, G, gwtf_padded_width(f), single_ks(1));
  return (int)hipGetLastError();
}

extern "C" int gwtf_train_fold1(const float* raw_c, const float* ystats, double n_total, const float* film_raw,
                                float* film_rec, float* bn_batch_c, int c, int B, int C, int f, int G, void* stream) {
  if (!raw_c || !ystats || !film_raw || !film_rec || !bn_batch_c || f <= 0 || f > GWTF_MAX_FP_TRAIN || G <= 0 || B <= 0 ||
      c < 0 || c >= C || n_total < 1.0)
    return GWTF_E_BADARG;
  hipLaunchKernelGGL(fold1_kernel<GWTF_STAT_REPLICAS>, dim3(B), dim3(2 * GWTF_MAX_FP), 0, (hipStream_t)stream, raw_c, ystats, n_total,
                     film_raw, film_rec, bn_batch_c, c, C, f, G, gwtf_padded_width(f), single_ks(C));
  return (int)hipGetLastError();
}

// =====================================================================================================================
// Backward of the folds (train mode) and the fused per-coupling train backward.  Same arithmetic as autograd through the
// torch folds in autograd.py (which remain the multi-rank path); here each fold's backward is ONE small launch.
// =====================================================================================================================
namespace {

// un-scaled sd1 weights -> backward records' W1T images (k-slot map of csrc/gwtf_bwd.hip); SD0N is written by fold0
__global__ void pack_w1t_kernel(const float* __restrict__ raw, float* __restrict__ pb, int C, int f, int G, int FP) {
  const GwtfRaw R(f, G);
  const GwtfPackB P(FP);
  const size_t W1T = P.w1t_size(), PBs = P.coupling_size();
  const size_t total = 2 * W1T * (size_t)C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / (2 * W1T));
    size_t o = idx - (size_t)c * 2 * W1T;
    const int br = (int)(o / W1T);
    o -= (size_t)br * W1T;
    const float* rb = raw + (size_t)c * R.coupling_size() + (size_t)br * R.branch_size() + R.sd1_w();
    pb[(size_t)c * PBs + (size_t)br * W1T + o] =
        gwtf_w1t_slot(P, o, [&](int j, int i) { return (j < f && i < f) ? rb[(size_t)j * f + i] : 0.f; });
  }
}

// Writes g_a / g_bsh rows of this coupling, dW2, db2 and gS.  Every (branch, feature) is independent: grid = (branch,
// tile of 16 features), block = 16 slices (over shapes / statistic replicas) x 16 features, so the work that used to sit
// on one CU (16 wavefronts time-slicing 4 SIMDs: 14 us) is spread over 2 * FP/16 of them.
constexpr int kF1Slices = 16;

__global__ __launch_bounds__(kF1Slices * 16) void fold1_bwd_kernel(
    const float* __restrict__ raw_c, const float* __restrict__ ystats, double n_total, const float* __restrict__ film_raw,
    const float* __restrict__ g_film, const float* __restrict__ g_bias, float* __restrict__ g_film_raw,
    float* __restrict__ g_raw_c, float* __restrict__ g_stats, int c, int B, int C, int f, int G, int FP, const GwtfKS ks,
    int NR /* copies of ystats to add: 1 = the compact record */) {
  raw_c += blockIdx.z * ks.raw;      // blockIdx.z = mixture component
  ystats += blockIdx.z * ks.ys;
  g_bias += blockIdx.z * ks.gbias;
  g_raw_c += blockIdx.z * ks.raw;
  g_stats += blockIdx.z * ks.gstats;
  c += blockIdx.z * ks.Cper;
  C = ks.Ctot;
  __shared__ float st_part[kF1Slices][16][2];
  __shared__ double acc_part[kF1Slices][16][4];
  __shared__ float bias_part[kF1Slices][2];
  const int t = threadIdx.x % 16, sl = threadIdx.x / 16;
  const int br = blockIdx.x, j = blockIdx.y * 16 + t;
  const bool on = j < f;
  const GwtfRaw R(f, G);
  // every global load this thread needs is issued BEFORE the first barrier (one memory round trip instead of three):
  // its statistic replicas, and the FiLM records of its first kPre shapes (covers B <= 64; the rest loops normally)
  constexpr int kPre = 4;
  float pa[kPre], pb[kPre], pgc[kPre], pg0[kPre], pg1[kPre];
  const float* rb = raw_c + (size_t)br * R.branch_size();
  float* grb = g_raw_c + (size_t)br * R.branch_size();
  const float w20 = on ? rb[R.sd2_w() + j] : 0.f, w21 = on ? rb[R.sd2_w() + f + j] : 0.f;
#pragma unroll
  for (int i = 0; i < kPre; ++i) {
    const int b = sl + i * kF1Slices;
    const bool ok = on && b < B;
    const float* fr = film_raw + (((size_t)(ok ? b : 0) * C + c) * 2 + br) * 2 * FP;
    const float* gf = g_film + (((size_t)(ok ? b : 0) * C + c) * 2 + br) * 3 * FP;
    pa[i] = ok ? fr[j] : 1.f; pb[i] = ok ? fr[FP + j] : 0.f;
    pgc[i] = ok ? gf[j] : 0.f; pg0[i] = ok ? gf[FP + j] : 0.f; pg1[i] = ok ? gf[2 * FP + j] : 0.f;
  }
  {
    float ys = 0.f, yq = 0.f;
    constexpr int kPer = GWTF_STAT_REPLICAS / kF1Slices;
    if (on && NR == GWTF_STAT_REPLICAS) {          // (all four replicas of this slice in flight at once)
      float2 v[kPer];
#pragma unroll
      for (int i = 0; i < kPer; ++i)
        v[i] = *reinterpret_cast<const float2*>(&ystats[(size_t)(sl + i * kF1Slices) * (2 * FP * 2) + (br * FP + j) * 2]);
#pragma unroll
      for (int i = 0; i < kPer; ++i) { ys += v[i].x; yq += v[i].y; }
    } else if (on) {
      for (int r = sl; r < NR; r += kF1Slices) {
        ys += ystats[(size_t)r * (2 * FP * 2) + (br * FP + j) * 2];
        yq += ystats[(size_t)r * (2 * FP * 2) + (br * FP + j) * 2 + 1];
      }
    }
    st_part[sl][t][0] = ys;
    st_part[sl][t][1] = yq;
    if (blockIdx.y == 0 && t < 2) {
      float bv[kPer], bsum = 0.f;
#pragma unroll
      for (int i = 0; i < kPer; ++i) bv[i] = g_bias[(sl + i * kF1Slices) * 4 + 2 * br + t];
#pragma unroll
      for (int i = 0; i < kPer; ++i) bsum += bv[i];
      bias_part[sl][t] = bsum;
    }
  }
  __syncthreads();
  float ys = 0.f, yq = 0.f;
#pragma unroll
  for (int i = 0; i < kF1Slices; ++i) { ys += st_part[i][t][0]; yq += st_part[i][t][1]; }
  const double mean = ys / n_total;
  double var = yq / n_total - mean * mean;
  if (var < 0.0) var = 0.0;
  const float s1 = 1.0f / sqrtf((float)var + GWTF_BN_EPS);
  double g_m1 = 0.0, g_s1 = 0.0, gw20 = 0.0, gw21 = 0.0;
  auto one = [&](int b, float a, float bs, float gc, float gu0, float gu1) {
    float* go = g_film_raw + (((size_t)b * C + c) * 2 + br) * 2 * FP;
    const float guw = gu0 * w20 + gu1 * w21;
    go[j] = -gc * bs / (a * a * s1) + guw * s1;   // dL/da
    go[FP + j] = gc / (a * s1);                   // dL/dbsh
    gw20 += (double)gu0 * a * s1;
    gw21 += (double)gu1 * a * s1;
    g_s1 += (double)(-gc * bs / (a * s1 * s1)) + (double)guw * a;
    g_m1 -= gc;
  };
  if (on) {
#pragma unroll
    for (int i = 0; i < kPre; ++i)
      if (sl + i * kF1Slices < B) one(sl + i * kF1Slices, pa[i], pb[i], pgc[i], pg0[i], pg1[i]);
    for (int b = sl + kPre * kF1Slices; b < B; b += kF1Slices) {
      const float* fr = film_raw + (((size_t)b * C + c) * 2 + br) * 2 * FP;
      const float* gf = g_film + (((size_t)b * C + c) * 2 + br) * 3 * FP;
      one(b, fr[j], fr[FP + j], gf[j], gf[FP + j], gf[2 * FP + j]);
    }
  }
  acc_part[sl][t][0] = g_m1; acc_part[sl][t][1] = g_s1; acc_part[sl][t][2] = gw20; acc_part[sl][t][3] = gw21;
  __syncthreads();
  if (sl != 0) return;
  if (!on) {                      // padded features of the last tile
    g_stats[(br * 2 + 0) * FP + j] = 0.f;
    g_stats[(br * 2 + 1) * FP + j] = 0.f;
    return;
  }
  g_m1 = g_s1 = gw20 = gw21 = 0.0;
#pragma unroll
  for (int i = 0; i < kF1Slices; ++i) {
    g_m1 += acc_part[i][t][0]; g_s1 += acc_part[i][t][1]; gw20 += acc_part[i][t][2]; gw21 += acc_part[i][t][3];
  }
  const double g_v1 = -0.5 * g_s1 * (double)s1 * s1 * s1;
  g_stats[(br * 2 + 0) * FP + j] = (float)((g_m1 - 2.0 * mean * g_v1) / n_total);
  g_stats[(br * 2 + 1) * FP + j] = (float)(g_v1 / n_total);
  grb[R.sd2_w() + j] = (float)gw20;
  grb[R.sd2_w() + f + j] = (float)gw21;
  if (blockIdx.y == 0 && t < 2) {
    float bsum = 0.f;
#pragma unroll
    for (int i = 0; i < kF1Slices; ++i) bsum += bias_part[i][t];
    grb[R.sd2_b() + t] = bsum;
  }
}

// g_sd0 replicas -> dW0, dgamma0, dbeta0, and this workgroup's share of the five sums the moment gradients need.
// Every (branch, feature) is independent up to those sums: grid = (branch, tile of 16 features), block = 16 slices (over
// the replicas) x 16 features.  Each block turns its five sums into its SHARE of the nine moment gradients gM (linear in them) and
// adds it to gm [16] with float atomics (gm zero before the level's launch): 2 * FP/16 adds per value, no finishing kernel.
// block (bx = branch, by = tile of 16 features of n_by, bz = mixture component)
__device__ __forceinline__ void fold0_bwd_block(
    const float* __restrict__ raw_c, const float* __restrict__ mom_rep, double n_total, int pat,
    const float* __restrict__ g_sd0, float* __restrict__ g_raw_c, float* __restrict__ gm, int f, int G, int FP,
    const GwtfKS& ks, int bx, int by, int bz, int n_by, int NR /* copies of the moment record */) {
  raw_c += bz * ks.raw;
  mom_rep += bz * ks.mom;
  g_sd0 += bz * ks.gsd0;
  g_raw_c += bz * ks.raw;
  gm += bz * ks.gmom;
  const int t = threadIdx.x % 16, sl = threadIdx.x / 16;
  const int br = bx, j = by * 16 + t;
  __shared__ float mom_part[kF1Slices][9];
  __shared__ double gs_part[kF1Slices][16][3];
  __shared__ double red[16][5];   // per feature: gE0, gE1, gC00, gC01, gC11 contributions
  const bool on = j < f;
  // (a thread's loads all issued before the first is used: ONE memory round trip where a rolled loop over its four replicas was four)
  constexpr int kPer = GWTF_STAT_REPLICAS / kF1Slices;
  if (t < 9) {
    float sacc = 0.f;
    if (NR == GWTF_STAT_REPLICAS) {
      float v[kPer];
#pragma unroll
      for (int i = 0; i < kPer; ++i) v[i] = mom_rep[(sl + i * kF1Slices) * 16 + t];
#pragma unroll
      for (int i = 0; i < kPer; ++i) sacc += v[i];
    } else {
      for (int r = sl; r < NR; r += kF1Slices) sacc += mom_rep[r * 16 + t];
    }
    mom_part[sl][t] = sacc;
  }
  {
    double g0 = 0.0, g1 = 0.0, gc = 0.0;
    if (on) {
      float v[kPer][3];
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const float* gs = g_sd0 + (size_t)(sl + i * kF1Slices) * (2 * 3 * FP) + (size_t)br * 3 * FP;
        v[i][0] = gs[j]; v[i][1] = gs[FP + j]; v[i][2] = gs[2 * FP + j];
      }
#pragma unroll
      for (int i = 0; i < kPer; ++i) { g0 += v[i][0]; g1 += v[i][1]; gc += v[i][2]; }
    }
    gs_part[sl][t][0] = g0; gs_part[sl][t][1] = g1; gs_part[sl][t][2] = gc;
  }
  __syncthreads();
  if (sl != 0) return;
  float mom[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < kF1Slices; ++r) v += mom_part[r][i];
    mom[i] = v;
  }
  int k0, k1, w0d, w1d;
  gwtf_pattern_dims(pat, &k0, &k1, &w0d, &w1d);
  const KeptMoments km = kept_moments(mom, k0, k1, n_total);
  const double e0 = km.e0, e1 = km.e1, c00 = km.c00, c01 = km.c01, c11 = km.c11;
#pragma unroll
  for (int i = 0; i < 5; ++i) red[t][i] = 0.0;
  if (on) {
    const GwtfRaw R(f, G);
    const float* rb = raw_c + (size_t)br * R.branch_size();
    float* grb = g_raw_c + (size_t)br * R.branch_size();
    const int kk = gwtf_pattern_kept(pat);
    const double wa = rb[R.sd0_w(j, 0, kk)], wb = kk > 1 ? rb[R.sd0_w(j, 1, kk)] : 0.0;
    const double gamma = rb[R.bn0() + j];
    const double mean = wa * e0 + wb * e1;
    double var = wa * wa * c00 + 2.0 * wa * wb * c01 + wb * wb * c11;
    if (var < 0.0) var = 0.0;
    const double isd = 1.0 / sqrt((double)((float)var + GWTF_BN_EPS));
    const double s = gamma * isd;
    double g0 = 0.0, g1 = 0.0, gc = 0.0;
#pragma unroll
    for (int r = 0; r < kF1Slices; ++r) { g0 += gs_part[r][t][0]; g1 += gs_part[r][t][1]; gc += gs_part[r][t][2]; }
    const double g_s = g0 * wa + g1 * wb - gc * mean;
    const double g_mean = -gc * s;
    const double g_var = g_s * gamma * (-0.5) * isd * isd * isd;
    const double gwa = g0 * s + g_mean * e0 + g_var * 2.0 * (wa * c00 + wb * c01);
    const double gwb = g1 * s + g_mean * e1 + g_var * 2.0 * (wa * c01 + wb * c11);
    grb[R.sd0_w(j, 0, kk)] = (float)gwa;
    if (kk > 1) grb[R.sd0_w(j, 1, kk)] = (float)gwb;      // k = 1: the second half of the reserved 2f floats stays zero
    grb[R.bn0() + j] = (float)(g_s * isd);       // d gamma
    grb[R.bn0() + f + j] = (float)gc;            // d beta
    red[t][0] = g_mean * wa;
    red[t][1] = g_mean * wb;
    red[t][2] = g_var * wa * wa;
    red[t][3] = g_var * 2.0 * wa * wb;
    red[t][4] = g_var * wb * wb;
  }
  // slice 0 is the 16 lanes of one wavefront quarter: LDS writes above are ordered before the reads below by the
  // wavefront's in-order LDS queue + waitcnt (same wavefront, no barrier needed)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (t == 0) {
    double r5[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += red[u][i];
      r5[i] = acc;
    }
    float share[16];
    moment_grad_terms(km, r5, pat, n_total, share);
#pragma unroll
    for (int i = 0; i < 9; ++i) atomicAdd(&gm[i], share[i]);
  }
}

__global__ __launch_bounds__(kF1Slices * 16) void fold0_bwd_kernel(
    const float* __restrict__ raw_c, const float* __restrict__ mom_rep, double n_total, int pat,
    const float* __restrict__ g_sd0, float* __restrict__ g_raw_c, float* __restrict__ gm, int f, int G, int FP,
    const GwtfKS ks, int NR) {
  fold0_bwd_block(raw_c, mom_rep, n_total, pat, g_sd0, g_raw_c, gm, f, G, FP, ks, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, NR);
}

// g_in = g_a [+ g_b: a second pass's share, null when one merged pass wrote g_a] + d(moments)/dx:  gM_a + sum_b Q_ab x_b,
// Q_aa = 2 gM_aa, Q_ab = gM_ab, with the level's nine moment gradients gM (fold0_bwd_block).  The train pipeline applies this on the
// fly inside the next level's passes (GwtfCombine); as a pass of its own it is left for the LAST level (dL/dp) and the per-coupling
// entry point.   block (bx of n_bx = slice of the points, by = shape, bz = mixture component)
__device__ __forceinline__ void combine_block(const float* __restrict__ x, const float* __restrict__ ga,
                                              const float* __restrict__ gb, const float* __restrict__ gm_k,
                                              float* __restrict__ g_in, int B, int N, const GwtfKS& ks, int bx, int by, int bz,
                                              int n_bx) {
  x += bz * ks.x;
  ga += bz * ks.pts;
  if (gb) gb += bz * ks.pts;
  g_in += bz * ks.pts;
  const float* gm = gm_k + bz * ks.gmom;
  const int b = by;
  const float g0 = gm[0], g1 = gm[1], g2 = gm[2];
  const float q00 = 2.f * gm[3], q01 = gm[4], q02 = gm[5], q11 = 2.f * gm[6], q12 = gm[7], q22 = 2.f * gm[8];
  for (int n = bx * blockDim.x + threadIdx.x; n < N; n += n_bx * blockDim.x) {
    const size_t o0 = ((size_t)b * 3 + 0) * N + n, o1 = o0 + N, o2 = o1 + N;
    const float x0 = x[o0], x1 = x[o1], x2 = x[o2];
    const float a0 = gb ? ga[o0] + gb[o0] : ga[o0], a1 = gb ? ga[o1] + gb[o1] : ga[o1], a2 = gb ? ga[o2] + gb[o2] : ga[o2];
    g_in[o0] = a0 + g0 + q00 * x0 + q01 * x1 + q02 * x2;
    g_in[o1] = a1 + g1 + q01 * x0 + q11 * x1 + q12 * x2;
    g_in[o2] = a2 + g2 + q02 * x0 + q12 * x1 + q22 * x2;
  }
}

__global__ __launch_bounds__(256) void combine_kernel(const float* __restrict__ x, const float* __restrict__ ga,
                                                      const float* __restrict__ gb, const float* __restrict__ gm,
                                                      float* __restrict__ g_in, int B, int N, const GwtfKS ks) {
  combine_block(x, ga, gb, gm, g_in, B, N, ks, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// The tail of a backward level: the sd0 fold's backward (its share of the moment gradients gM feeds the NEXT level's passes, which
// apply the gradient combine on the fly: gwtf_layout.h GwtfCombine), beside it stage 1 of this level's dW1 partial reduction and
// stage 2 of the PREVIOUS level's (nothing on the dependency chain waits for either) -- one launch.  Blocks [0, n_role0) fold, the
// next n_role1 run stage 1, the rest stage 2 (mid_prev == null: none).
__global__ __launch_bounds__(256) void bwd_tail1_kernel(const float* __restrict__ raw_c, const float* __restrict__ mom,
                                                        double n_total, int pat, const float* __restrict__ g_sd0,
                                                        float* __restrict__ g_raw_c, float* __restrict__ gm, int f, int G, int FP,
                                                        const GwtfKS ks, int K, int NR, const float* __restrict__ ws, int n_partials,
                                                        float* __restrict__ mid, int rec, const float* __restrict__ mid_prev,
                                                        float* __restrict__ dW1_prev, size_t branch_stride) {
  __shared__ float red[4][64];
  const int n_by = FP / 16, n_role0 = 2 * n_by * K, gx = (rec + 255) / 256, n_role1 = gx * gwtf_dw1::kStage * K;
  if ((int)blockIdx.x < n_role0) {
    const int b = blockIdx.x;
    fold0_bwd_block(raw_c, mom, n_total, pat, g_sd0, g_raw_c, gm, f, G, FP, ks, b % 2, (b / 2) % n_by, b / (2 * n_by), n_by, NR);
    return;
  }
  if ((int)blockIdx.x < n_role0 + n_role1) {
    const int b = blockIdx.x - n_role0;
    const int bx = b % gx, by = (b / gx) % gwtf_dw1::kStage, bz = b / (gx * gwtf_dw1::kStage);
    gwtf_dw1::fold_block(ws + bz * ks.dw1, n_partials, mid + bz * ks.dw1, rec, bx, by, threadIdx.x);
    return;
  }
  const int b = blockIdx.x - n_role0 - n_role1, g2 = (gwtf_dw1::rec_floats(f) + 63) / 64;
  gwtf_dw1::reduce_block(mid_prev + (b / g2) * ks.dw1, dW1_prev + (b / g2) * ks.raw, f, branch_stride, b % g2, threadIdx.x, red);
}

// After the LAST backward level: the one gradient combine no later pass applies (dL/dp of the input clouds) beside stage 2 of that
// level's dW1 reduction.
__global__ __launch_bounds__(256) void bwd_tail2_kernel(const float* __restrict__ x, const float* __restrict__ ga,
                                                        const float* __restrict__ gm, float* __restrict__ g_in, int B, int N,
                                                        const GwtfKS ks, int K, int n_bx, const float* __restrict__ mid,
                                                        float* __restrict__ dW1, int f, size_t branch_stride) {
  __shared__ float red[4][64];
  const int n_role0 = n_bx * B * K;
  if ((int)blockIdx.x < n_role0) {
    const int b = blockIdx.x;
    combine_block(x, ga, nullptr, gm, g_in, B, N, ks, b % n_bx, (b / n_bx) % B, b / (n_bx * B), n_bx);
    return;
  }
  const int b = blockIdx.x - n_role0, gx = (gwtf_dw1::rec_floats(f) + 63) / 64;
  const int bx = b % gx, bz = b / gx;
  gwtf_dw1::reduce_block(mid + bz * ks.dw1, dW1 + bz * ks.raw, f, branch_stride, bx, threadIdx.x, red);
}

// sum of the R copies of a replicated statistic slab [K][R][n] -> the contiguous record [K][n] a data-parallel run all-reduces and
// its consumers then read (fixed order: the same sum on every run)
__global__ __launch_bounds__(256) void stat_compact_kernel(const float* __restrict__ slab, float* __restrict__ out, int n,
                                                           size_t slab_sk) {
  constexpr int R = GWTF_STAT_REPLICAS;      // a compile-time trip count: all 64 loads of a value in flight at once
  const float* s = slab + (size_t)blockIdx.x * slab_sk;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = s[(size_t)r * n + i];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int r = 0; r < R; r += 4) { a0 += v[r]; a1 += v[r + 1]; a2 += v[r + 2]; a3 += v[r + 3]; }
    out[(size_t)blockIdx.x * n + i] = (a0 + a1) + (a2 + a3);
  }
}

}  // namespace

extern "C" int gwtf_pack_w1t(const float* raw, float* packed_b, int C, int f, int G, void* stream) {
  if (!raw || !packed_b || C <= 0 || f <= 0 || f > GWTF_MAX_FP_TRAIN || G <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(pack_w1t_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, raw, packed_b, C, f, G,
                     gwtf_padded_width(f));
  return (int)hipGetLastError();
}

// =====================================================================================================================
// K-BATCHED, PHASE-SPLIT TRAIN PIPELINE.  The K components of a flow mixture (reference flow_mixture.py:163-166 loops over
// them in Python) go through every kernel of the train-mode chain TOGETHER -- grid dimension = component -- and the chain
// is cut into phases at exactly the points where a data-parallel run must sum statistics over the ranks
// (SyncBatchNorm semantics, reference train_ae.py:152):
//
//   forward, per depth level (coupling c = step or C-1-step):
//     GWTF_PHASE_FWD_INIT           moments of the input clouds                        -> all-reduce mom_c[0][0]    (16)
//     GWTF_PHASE_FWD_A   fold0 (K workgroups) + statistics pass (K x tiles)            -> all-reduce ys_c[c]        (K x 2*FP*2)
//     GWTF_PHASE_FWD_B   fold1 (B x K) + apply pass (K x tiles, moments of the output) -> all-reduce mom_c[step+1]  (K x 16)
//   backward, per depth level in reverse:
//     GWTF_PHASE_BWD_A   light pass (FiLM-record / bias sums of the coupling path) + fold1 backward -> all-reduce g_stats[c] (K x 2*2*FP)
//     GWTF_PHASE_BWD_B   merged backward (coupling + statistics path) + tail (fold0 backward | dW1 stage 1 | the previous level's
//                        dW1 stage 2)                                                  -> all-reduce g_mom[c]       (K x 16)
//     GWTF_PHASE_BWD_C   nothing, except after the LAST level: the final gradient combine (dL/dp) + that level's dW1 stage 2
//   The gradient combine of a level (moment path: g += gM + Q x) is applied on the fly by the NEXT level's light and merged passes
//   (gwtf_layout.h GwtfCombine), which recompute that level's input coordinates anyway.
//
// Statistics are accumulated by atomics spread over 64 copies.  One rank: the folds add the copies up themselves.  Data parallel
// (ctx.mom_c / ys_c given): each forward phase ends with ONE small launch that sums the copies into a compact [K][n] record -- what
// goes on the wire (3 KB per collective at f = 37, K = 4) and what every consumer then reads.  (Letting the last workgroup of the
// pass do that sum -- an arrival ticket -- was built and measured: every workgroup then waits for its own atomics and a returning
// ticket add before it can retire, statistics pass 23.0 -> 32.0 us, apply pass 29.0 -> 31.1; docs/LOG.md round 4.)
//
// i.e. TWO small collectives per depth level and direction for all K x 2 branches (66 per forward of the 33-coupling
// configs; the reference's SyncBatchNorm issues 1056), 4 + 4 launches per level instead of 4K + 10K.  A single rank runs
// all phases back to back from one C call (gwtf_mtrain_forward / gwtf_mtrain_backward).  Buffer layouts: GwtfTrainCtx in
// include/gwtf.h; per-level records are contiguous over K so that one collective covers a level.
// =====================================================================================================================
namespace {
struct Dims {
  int FP;
  size_t RC, PW, PB, MS, YS, YC, BS, XS, GS0, GB, GST, GM, FS;
};
Dims dims_of(const GwtfTrainCtx* t) {
  Dims d;
  d.FP = gwtf_padded_width(t->f);
  d.RC = gwtf_raw_coupling_floats(t->f, t->G);
  d.PW = gwtf_packed_w_coupling_floats(t->f);
  d.PB = gwtf_packed_b_coupling_floats(t->f);
  d.MS = (size_t)GWTF_STAT_REPLICAS * 16;
  d.YC = (size_t)2 * d.FP * 2;
  d.YS = (size_t)GWTF_STAT_REPLICAS * d.YC;
  d.BS = (size_t)2 * 4 * 2 * t->f;
  d.XS = (size_t)t->B * 3 * t->N;
  d.GS0 = (size_t)GWTF_STAT_REPLICAS * 2 * 3 * d.FP;
  d.GB = (size_t)GWTF_STAT_REPLICAS * 4;
  d.GST = (size_t)2 * 2 * d.FP;
  d.GM = 16;
  d.FS = gwtf_film_out_size(d.FP);
  return d;
}
// one component's dW1 workspace: the merged pass's partials + TWO stage-1 scratch records (stage 2 of a level runs beside stage 1 of
// the next)
size_t dw1_region(int f, int B, int N) {
  return gwtf_dw1_workspace_floats(f, B, N) + 2 * gwtf_dw1_reduce_scratch_floats(f);
}
// Where the consumers of a forward statistic read it: the pass's 64 copies (one rank), or the compact record (data parallel)
struct StatView { const float* p; size_t sk; int nr; };
StatView moments_of(const GwtfTrainCtx* t, const Dims& d, int level) {
  if (t->mom_c) return {t->mom_c + (size_t)level * t->K * 16, level == 0 ? (size_t)0 : (size_t)16, 1};
  return {t->moments + (size_t)level * t->K * d.MS, level == 0 ? (size_t)0 : d.MS, GWTF_STAT_REPLICAS};   // level 0: the shared input clouds
}
StatView ystats_of(const GwtfTrainCtx* t, const Dims& d, int c) {
  if (t->ys_c) return {t->ys_c + (size_t)c * t->K * d.YC, d.YC, 1};
  return {t->ystats + (size_t)c * t->K * d.YS, d.YS, GWTF_STAT_REPLICAS};
}
GwtfKS strides_of(const GwtfTrainCtx* t, const Dims& d, bool first_level) {
  GwtfKS ks = {};
  ks.raw = (size_t)t->C * d.RC;
  ks.pw = (size_t)t->C * d.PW;
  ks.pb = (size_t)t->C * d.PB;
  ks.x = first_level ? 0 : d.XS;       // the first level reads the shared input clouds
  ks.pts = d.XS;
  ks.mom = first_level ? 0 : (t->mom_c ? 16 : d.MS);     // ... and their (single) moment record
  ks.ys = t->ys_c ? d.YC : d.YS;
  ks.bn = (size_t)t->C * d.BS;
  ks.gsd0 = d.GS0;
  ks.gbias = d.GB;
  ks.gstats = d.GST;
  ks.gmom = d.GM;
  ks.dw1 = dw1_region(t->f, t->B, t->N);
  ks.Cper = t->C;
  ks.Ctot = t->K * t->C;
  ks.tune = t->tune;
  return ks;
}
bool ctx_ok(const GwtfTrainCtx* t, bool backward) {
  if (!t || t->K <= 0 || t->K > GWTF_MAX_COMPONENTS || t->B <= 0 || t->N <= 0 || t->C <= 0 || t->f <= 0 || t->f > GWTF_MAX_FP_TRAIN ||
      t->G <= 0 || t->pattern0 < 0 || t->pattern0 > 5 || t->n_total < 1.0)
    return false;
  if (t->mode != GWTF_MODE_DIRECT && t->mode != GWTF_MODE_INVERSE) return false;
  if (!t->p || !t->raw || !t->packed_w || !t->film_raw || !t->film_rec || !t->bn_batch || !t->xbuf || !t->logdet) return false;
  if ((t->mom_c == nullptr) != (t->ys_c == nullptr)) return false;          // both compact records, or neither
  const bool copies = t->moments && t->ystats;
  if (!backward && !copies) return false;                                   // the passes' atomics always land in the copies
  if (backward && !copies && !t->mom_c) return false;
  const bool any = t->ps || t->mus || t->logvars, all = t->ps && t->mus && t->logvars, ps_only = t->ps && !t->mus && !t->logvars;
  if (any && !all && !ps_only) return false;
  if (backward && (!t->ps || !t->packed_b || !t->g_out || !t->g_ld || !t->g_bufs || !t->dw1_ws ||
                   !t->g_film || !t->g_sd0 || !t->g_bias || !t->g_stats || !t->g_mom || !t->g_film_raw || !t->g_raw))
    return false;
  return true;
}
void compact(const float* slab, size_t slab_sk, float* out, int K, int n, hipStream_t st) {
  hipLaunchKernelGGL(stat_compact_kernel, dim3(K), dim3(256), 0, st, slab, out, n, slab_sk);
}
}  // namespace

extern "C" size_t gwtf_mtrain_dw1_floats(int f, int B, int N) { return dw1_region(f, B, N); }

extern "C" int gwtf_mtrain_phase(const GwtfTrainCtx* t, int phase, int step) {
  const bool bwd = phase == GWTF_PHASE_BWD_A || phase == GWTF_PHASE_BWD_B || phase == GWTF_PHASE_BWD_C;
  if (!ctx_ok(t, bwd) || step < 0 || step >= t->C) return GWTF_E_BADARG;
  const Dims d = dims_of(t);
  const int K = t->K, C = t->C, B = t->B, N = t->N, f = t->f, G = t->G, FP = d.FP;
  hipStream_t st = (hipStream_t)t->stream;
  if (phase == GWTF_PHASE_FWD_INIT) {
    int rc = gwtf_train_moments(t->p, t->moments, B, N, t->stream);
    if (!rc && t->mom_c) compact(t->moments, 0, t->mom_c, 1, 16, st);
    return rc ? rc : (int)hipGetLastError();
  }

  if (!bwd) {
    const int c = t->mode == GWTF_MODE_DIRECT ? step : C - 1 - step;
    const int pat = (t->pattern0 + c) % 6;
    const GwtfKS ks = strides_of(t, d, step == 0);
    const float* cur = step == 0 ? t->p : t->xbuf + (size_t)((step - 1) & 1) * K * d.XS;
    const StatView mom = moments_of(t, d, step), ys = ystats_of(t, d, c);
    float* ys_copies = t->ystats + (size_t)c * K * d.YS;
    if (phase == GWTF_PHASE_FWD_A) {
      float* pb_c = t->packed_b ? t->packed_b + (size_t)c * d.PB : nullptr;
      if (mom.nr == 1)
        hipLaunchKernelGGL(fold0_kernel<1>, dim3(K), dim3(2 * GWTF_MAX_FP), 0, st, t->raw + (size_t)c * d.RC, mom.p, t->n_total, pat,
                           t->packed_w + (size_t)c * d.PW, pb_c, t->bn_batch + (size_t)c * d.BS, f, G, FP, ks);
      else
        hipLaunchKernelGGL(fold0_kernel<GWTF_STAT_REPLICAS>, dim3(K), dim3(2 * GWTF_MAX_FP), 0, st, t->raw + (size_t)c * d.RC, mom.p,
                           t->n_total, pat, t->packed_w + (size_t)c * d.PW, pb_c, t->bn_batch + (size_t)c * d.BS, f, G, FP, ks);
      int rc = gwtf_internal_stats_k(cur, t->packed_w + (size_t)c * d.PW, ys_copies, K, B, N, f, pat, ks.x, ks.pw, d.YS, t->tune, t->stream);
      if (!rc && t->ys_c) compact(ys_copies, d.YS, t->ys_c + (size_t)c * K * d.YC, K, (int)d.YC, st);
      return rc ? rc : (int)hipGetLastError();
    }
    if (phase == GWTF_PHASE_FWD_B) {
      if (ys.nr == 1)
        hipLaunchKernelGGL(fold1_kernel<1>, dim3(B, K), dim3(2 * GWTF_MAX_FP), 0, st, t->raw + (size_t)c * d.RC, ys.p, t->n_total,
                           t->film_raw, t->film_rec, t->bn_batch + (size_t)c * d.BS, c, C, f, G, FP, ks);
      else
        hipLaunchKernelGGL(fold1_kernel<GWTF_STAT_REPLICAS>, dim3(B, K), dim3(2 * GWTF_MAX_FP), 0, st, t->raw + (size_t)c * d.RC, ys.p,
                           t->n_total, t->film_raw, t->film_rec, t->bn_batch + (size_t)c * d.BS, c, C, f, G, FP, ks);
      float* nxt = t->xbuf + (size_t)(step & 1) * K * d.XS;
      float* mom_next = step + 1 < C ? t->moments + (size_t)(step + 1) * K * d.MS : nullptr;
      int rc = gwtf_internal_apply_k(cur, t->packed_w, t->film_rec, nxt, step > 0 ? t->logdet : nullptr, t->logdet, t->ps, t->mus,
                                     t->logvars, mom_next, d.MS, c, K, B, N, C, f, t->pattern0, t->eps, t->mode, ks.x, d.XS, t->tune,
                                     t->stream);
      if (!rc && mom_next && t->mom_c) compact(mom_next, d.MS, t->mom_c + (size_t)(step + 1) * K * 16, K, 16, st);
      return rc ? rc : (int)hipGetLastError();
    }
    return GWTF_E_BADARG;
  }

  // backward: `step` counts the couplings in the REVERSE of the forward's processing order
  const bool inverse = t->mode == GWTF_MODE_INVERSE;
  const int c = inverse ? step : C - 1 - step;
  const int fstep = inverse ? C - 1 - c : c;            // this coupling's position in the forward order
  const int pat = (t->pattern0 + c) % 6;
  const bool from_p = inverse ? c + 1 >= C : c == 0;    // this coupling read the shared input clouds
  GwtfKS ks = strides_of(t, d, fstep == 0);
  const float* x_in = from_p ? t->p : t->ps + (size_t)(inverse ? c + 1 : c - 1) * d.XS;
  if (!from_p) ks.x = (size_t)C * d.XS;                 // lists are [K][C][B][3][N]
  const int c_prev = inverse ? c - 1 : c + 1;           // the coupling handled by the previous backward step
  const float* cur = step == 0 ? t->g_out : t->g_bufs + (size_t)(c_prev & 1) * K * d.XS;   // that step's RAW gradient (its combine: below)
  float* nxt = t->g_bufs + (size_t)(c & 1) * K * d.XS;
  const StatView mom = moments_of(t, d, fstep), ys = ystats_of(t, d, c);
  float* g_sd0 = t->g_sd0 + (size_t)c * K * d.GS0;
  float* g_bias = t->g_bias + (size_t)c * K * d.GB;
  float* g_stats = t->g_stats + (size_t)c * K * d.GST;
  float* g_mom = t->g_mom + (size_t)c * K * d.GM;
  const float* pw_c = t->packed_w + (size_t)c * d.PW;
  const float* pb_c = t->packed_b + (size_t)c * d.PB;
  float* g_raw_c = t->g_raw + (size_t)c * d.RC;
  const float* g_ps_c = t->g_ps ? t->g_ps + (size_t)c * d.XS : nullptr;
  const float* g_lvs_c = t->g_lvs ? t->g_lvs + (size_t)c * d.XS : nullptr;
  const GwtfRaw R(f, G);
  const int rec = gwtf_dw1::rec_floats(f), n_partials = gwtf_dw1_partials(B, N);
  const size_t scratch = gwtf_dw1_reduce_scratch_floats(f);
  float* mid = t->dw1_ws + (size_t)n_partials * rec + (size_t)(c & 1) * scratch;
  // the previous step's gradient combine, applied on the fly by this step's passes: its input was this coupling's output
  GwtfCombine cmb = {};
  if (step > 0) {
    cmb.gm = t->g_mom + (size_t)c_prev * K * d.GM;
    cmb.gm_sk = d.GM;
  }
  if (phase == GWTF_PHASE_BWD_A) {
    int rc = gwtf_internal_light_backward_k(x_in, cur, t->g_ld, pw_c, t->film_rec, t->g_film, g_bias, c, K, B, N, f, t->pattern0,
                                            t->eps, t->mode, ks, g_ps_c, g_lvs_c, cmb, t->stream);
    if (rc) return rc;
    hipLaunchKernelGGL(fold1_bwd_kernel, dim3(2, FP / 16, K), dim3(kF1Slices * 16), 0, st, t->raw + (size_t)c * d.RC, ys.p,
                       t->n_total, t->film_raw, t->g_film, g_bias, t->g_film_raw, g_raw_c, g_stats, c, B, C, f, G, FP, ks, ys.nr);
    return (int)hipGetLastError();
  }
  if (phase == GWTF_PHASE_BWD_B) {
    int rc = gwtf_internal_merged_backward_k(x_in, cur, t->g_ld, pw_c, pb_c, t->film_rec, g_stats, nxt, t->dw1_ws, g_sd0, c, K, B,
                                             N, f, t->pattern0, t->eps, t->mode, ks, g_ps_c, g_lvs_c, cmb, t->stream);
    if (rc) return rc;
    // sd0 fold's backward | stage 1 of this level's dW1 reduction | stage 2 of the previous level's: one launch
    const float* mid_prev = step > 0 ? t->dw1_ws + (size_t)n_partials * rec + (size_t)(c_prev & 1) * scratch : nullptr;
    float* dW1_prev = step > 0 ? t->g_raw + (size_t)c_prev * d.RC + R.sd1_w() : nullptr;
    const unsigned blocks = 2u * (FP / 16) * K + (unsigned)((rec + 255) / 256) * gwtf_dw1::kStage * K +
                            (step > 0 ? (unsigned)((gwtf_dw1::rec_floats(f) + 63) / 64) * K : 0u);      // stage 2 walks the record's elements
    hipLaunchKernelGGL(bwd_tail1_kernel, dim3(blocks), dim3(256), 0, st, t->raw + (size_t)c * d.RC, mom.p, t->n_total, pat, g_sd0,
                       g_raw_c, g_mom, f, G, FP, ks, K, mom.nr, t->dw1_ws, n_partials, mid, rec, mid_prev, dW1_prev, R.branch_size());
    return (int)hipGetLastError();
  }
  // GWTF_PHASE_BWD_C: only after the last level -- its gradient combine (dL/dp, in place) + stage 2 of its dW1 reduction
  if (step + 1 < C) return 0;
  const int bxn = (N + 255) / 256, n_bx = bxn < 64 ? bxn : 64;
  const unsigned blocks = (unsigned)n_bx * B * K + (unsigned)((gwtf_dw1::rec_floats(f) + 63) / 64) * K;
  hipLaunchKernelGGL(bwd_tail2_kernel, dim3(blocks), dim3(256), 0, st, x_in, nxt, g_mom, nxt, B, N, ks, K, n_bx, mid,
                     g_raw_c + R.sd1_w(), f, R.branch_size());
  return (int)hipGetLastError();
}

extern "C" int gwtf_mtrain_forward(const GwtfTrainCtx* t) {
  if (!ctx_ok(t, false)) return GWTF_E_BADARG;
  int rc = gwtf_mtrain_phase(t, GWTF_PHASE_FWD_INIT, 0);
  for (int step = 0; step < t->C && !rc; ++step) {
    rc = gwtf_mtrain_phase(t, GWTF_PHASE_FWD_A, step);
    if (!rc) rc = gwtf_mtrain_phase(t, GWTF_PHASE_FWD_B, step);
  }
  return rc;
}

extern "C" int gwtf_mtrain_backward(const GwtfTrainCtx* t) {
  if (!ctx_ok(t, true)) return GWTF_E_BADARG;
  int rc = 0;
  for (int step = 0; step < t->C && !rc; ++step) {
    rc = gwtf_mtrain_phase(t, GWTF_PHASE_BWD_A, step);
    if (!rc) rc = gwtf_mtrain_phase(t, GWTF_PHASE_BWD_B, step);
    if (!rc) rc = gwtf_mtrain_phase(t, GWTF_PHASE_BWD_C, step);
  }
  return rc;
}

// ---- running statistics of all BatchNorm modules of a stack in ONE launch ---------------------------------------------------
// table [n][3] u64 = device pointers {running_mean, running_var, num_batches_tracked (int64) or 0} of module i; src [n][2][f] =
// batch {mean, unbiased var} (the pipeline's bn_batch); momentum [n].  running = (1 - m) running + m batch, counter += 1:
// torch.nn.BatchNorm1d's update (reference flows.py:27-42 modules in train mode), instead of five _foreach_ calls over n tensors.
namespace {
__global__ void bn_running_update_kernel(const unsigned long long* __restrict__ table, const float* __restrict__ src,
                                         const float* __restrict__ momentum, int n, int f) {
  const int i = blockIdx.x;
  float* rm = reinterpret_cast<float*>(table[3 * (size_t)i]);
  float* rv = reinterpret_cast<float*>(table[3 * (size_t)i + 1]);
  long long* nbt = reinterpret_cast<long long*>(table[3 * (size_t)i + 2]);
  const float m = momentum[i];
  for (int j = threadIdx.x; j < f; j += blockDim.x) {
    rm[j] = (1.0f - m) * rm[j] + m * src[((size_t)i * 2 + 0) * f + j];
    rv[j] = (1.0f - m) * rv[j] + m * src[((size_t)i * 2 + 1) * f + j];
  }
  if (nbt && threadIdx.x == 0) *nbt += 1;
}
}  // namespace

extern "C" int gwtf_bn_running_update(const unsigned long long* table, const float* src, const float* momentum, int n, int f,
                                      void* stream) {
  if (!table || !src || !momentum || n <= 0 || f <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, table, src, momentum, n, f);
  return (int)hipGetLastError();
}

// ---- the raw arena of a stack gathered from its ~1300 parameter / buffer tensors in ONE launch -----------------------------------
// table [n][3] u64 = {source device pointer, destination offset (floats), number of floats}; one workgroup per tensor (they are
// small: at most f G floats).  torch.cat of the same views is a launch per 128 inputs.
namespace {
__global__ void gather_table_kernel(const unsigned long long* __restrict__ table, float* __restrict__ dst, int n) {
  const unsigned long long* row = table + 3 * (size_t)blockIdx.x;
  const float* __restrict__ src = reinterpret_cast<const float*>(row[0]);
  float* __restrict__ out = dst + row[1];
  const size_t m = row[2];
  // eight loads in flight per thread (a rolled load -> store loop is a round trip per iteration: 37 of them for a FiLM head's first
  // layer, which set the launch's time)
  constexpr int U = 8;
  const size_t step = (size_t)U * blockDim.x;
  size_t i0 = 0;
  for (; i0 + step <= m; i0 += step) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i0 + u * blockDim.x + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) out[i0 + u * blockDim.x + threadIdx.x] = v[u];
  }
  {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * blockDim.x + threadIdx.x;
      v[u] = i < m ? src[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * blockDim.x + threadIdx.x;
      if (i < m) out[i] = v[u];
    }
  }
}
}  // namespace

extern "C" int gwtf_gather_table(const unsigned long long* table, float* dst, int n, void* stream) {
  if (!table || !dst || n <= 0) return GWTF_E_BADARG;
  hipLaunchKernelGGL(gather_table_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table, dst, n);
  return (int)hipGetLastError();
}

// which half of xbuf / g_bufs holds the final coordinates / dL/dp of component k: base + half * K*B*3*N + k * B*3*N
extern "C" int gwtf_mtrain_final_forward_half(int C) { return (C - 1) & 1; }
extern "C" int gwtf_mtrain_final_backward_half(int C, int mode) { return (mode == GWTF_MODE_INVERSE ? C - 1 : 0) & 1; }

// ---- single-stack entry points (the ABI of round 1), now thin wrappers over the K-batched pipeline with K = 1 ----------
extern "C" int gwtf_train_forward(const float* p, const float* raw, float* packed_w, float* packed_b, const float* film_raw,
                                  float* moments, float* ystats, float* bn_batch, float* film_rec, float* xbuf,
                                  float* logdet, float* ps, float* mus, float* logvars, int B, int N, int C, int f, int G,
                                  int pattern0, float eps, int mode, int tune, void* stream) {
  GwtfTrainCtx t = {};
  t.K = 1; t.B = B; t.N = N; t.C = C; t.f = f; t.G = G; t.pattern0 = pattern0; t.mode = mode; t.eps = eps; t.tune = tune;
  t.n_total = (double)B * N;
  t.p = p; t.raw = raw; t.packed_w = packed_w; t.packed_b = packed_b; t.film_raw = film_raw; t.film_rec = film_rec;
  t.moments = moments; t.ystats = ystats; t.bn_batch = bn_batch; t.xbuf = xbuf; t.logdet = logdet;
  t.ps = ps; t.mus = mus; t.logvars = logvars; t.stream = stream;
  return gwtf_mtrain_forward(&t);
}

extern "C" int gwtf_train_backward(const float* p, const float* ps, const float* g_out, const float* g_ld, const float* raw,
                                   const float* packed_w, const float* packed_b, const float* film_rec, const float* film_raw,
                                   const float* moments, const float* ystats, float* g_bufs, float* g_xa, float* g_xb,
                                   float* dw1_ws, float* g_film, float* g_sd0, float* g_bias, float* g_stats, float* g_mom,
                                   float* g_film_raw, float* g_raw, int* final_buf, int B, int N, int C, int f, int G,
                                   int pattern0, float eps, int mode, void* stream) {
  if (!final_buf) return GWTF_E_BADARG;
  GwtfTrainCtx t = {};
  t.K = 1; t.B = B; t.N = N; t.C = C; t.f = f; t.G = G; t.pattern0 = pattern0; t.mode = mode; t.eps = eps;
  t.n_total = (double)B * N;
  t.p = p; t.raw = raw; t.packed_w = const_cast<float*>(packed_w); t.packed_b = const_cast<float*>(packed_b);
  t.film_raw = film_raw; t.film_rec = const_cast<float*>(film_rec);
  t.moments = const_cast<float*>(moments); t.ystats = const_cast<float*>(ystats);
  // forward-only fields the backward does not touch: any non-null pointer satisfies the context check
  t.bn_batch = g_stats; t.xbuf = g_bufs; t.logdet = g_xa;
  t.ps = const_cast<float*>(ps); t.mus = const_cast<float*>(ps); t.logvars = const_cast<float*>(ps);
  t.g_out = g_out; t.g_ld = g_ld; t.g_bufs = g_bufs; t.g_xa = g_xa; t.g_xb = g_xb; t.dw1_ws = dw1_ws; t.g_film = g_film;
  t.g_sd0 = g_sd0; t.g_bias = g_bias; t.g_stats = g_stats; t.g_mom = g_mom; t.g_film_raw = g_film_raw; t.g_raw = g_raw;
  t.stream = stream;
  *final_buf = gwtf_mtrain_final_backward_half(C, mode);
  return gwtf_mtrain_backward(&t);
}

// Backward of ONE coupling of the single-rank train pipeline (kept for the per-coupling autograd nodes and their tests):
// the three backward phases of the K = 1 pipeline on caller-addressed per-coupling buffers.
extern "C" int gwtf_train_coupling_backward(const float* x_in, const float* g_out, const float* g_ld, const float* raw_c,
                                            const float* packed_w_c, const float* packed_b_c, const float* film_rec,
                                            const float* film_raw, const float* moments_c, const float* ystats_c,
                                            float* g_in, float* g_xa, float* g_xb, float* dw1_ws, float* g_film, float* g_sd0,
                                            float* g_bias, float* g_stats, float* g_mom, float* g_film_raw, float* g_raw_c,
                                            int c, int B, int N, int C, int f, int G, int pattern0, float eps, int mode,
                                            void* stream) {
  if (!x_in || !g_out || !g_ld || !raw_c || !packed_w_c || !packed_b_c || !film_rec || !film_raw || !moments_c ||
      !ystats_c || !g_in || !g_xa || !g_xb || !dw1_ws || !g_film || !g_sd0 || !g_bias || !g_stats || !g_mom ||
      !g_film_raw || !g_raw_c)
    return GWTF_E_BADARG;
  const int FP = gwtf_padded_width(f), pat = (pattern0 + c) % 6;
  const double n_total = (double)B * N;
  hipStream_t st = (hipStream_t)stream;
  const GwtfKS ks = single_ks(C);
  if (B <= 0 || N <= 0 || C <= 0 || c < 0 || c >= C || f <= 0 || f > GWTF_MAX_FP_TRAIN || pattern0 < 0 || pattern0 > 5 ||
      (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE))
    return GWTF_E_BADARG;
  int rc = gwtf_internal_light_backward_k(x_in, g_out, g_ld, packed_w_c, film_rec, g_film, g_bias, c, 1, B, N, f, pattern0, eps,
                                          mode, ks, nullptr, nullptr, GwtfCombine{}, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(fold1_bwd_kernel, dim3(2, FP / 16, 1), dim3(kF1Slices * 16), 0, st, raw_c, ystats_c, n_total, film_raw, g_film,
                     g_bias, g_film_raw, g_raw_c, g_stats, c, B, C, f, G, FP, ks, GWTF_STAT_REPLICAS);
  rc = gwtf_internal_merged_backward_k(x_in, g_out, g_ld, packed_w_c, packed_b_c, film_rec, g_stats, g_xa, dw1_ws, g_sd0, c, 1, B,
                                       N, f, pattern0, eps, mode, ks, nullptr, nullptr, GwtfCombine{}, stream);
  if (rc) return rc;
  hipError_t me = hipMemsetAsync(g_mom, 0, 16 * sizeof(float), st);   // the nine moment gradients gM, accumulated by the fold's blocks
  if (me != hipSuccess) return (int)me;
  hipLaunchKernelGGL(fold0_bwd_kernel, dim3(2, FP / 16, 1), dim3(kF1Slices * 16), 0, st, raw_c, moments_c, n_total, pat, g_sd0,
                     g_raw_c, g_mom, f, G, FP, ks, GWTF_STAT_REPLICAS);
  const int bx = (N + 255) / 256;
  hipLaunchKernelGGL(combine_kernel, dim3(bx < 64 ? bx : 64, B, 1), dim3(256), 0, st, x_in, g_xa, static_cast<const float*>(nullptr),
                     g_mom, g_in, B, N, ks);
  const GwtfRaw R(f, G);
  return gwtf_dw1_reduce(dw1_ws, 1, g_raw_c + R.sd1_w(), R.branch_size(), f, B, N, stream);
}
