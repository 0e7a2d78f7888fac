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
#include "../../include/gwtf.h"

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

__device__ __forceinline__ int mom2_index(int a, int b) {  // index of S x_a x_b inside the 9-vector, a <= b
  return a == 0 ? 3 + b : (a == 1 ? 5 + b : 8);
}

// One workgroup, thread = (branch, feature).  bn_batch[branch][kind 0][2][f] <- {mean, unbiased var}.
__global__ void fold0_kernel(const float* __restrict__ raw_c, const float* __restrict__ mom_rep, double n_total, int pat,
                             float* __restrict__ pw_c, float* __restrict__ bn_batch, int f, int G, int FP) {
  const int t = threadIdx.x;
  __shared__ float mom[9];
  if (t < 9) {
    float sacc = 0.f;
    for (int r = 0; r < GWTF_STAT_REPLICAS; ++r) sacc += mom_rep[r * 16 + t];
    mom[t] = sacc;
  }
  __syncthreads();
  if (t >= 2 * f) return;
  const int br = t / f, j = t % f;
  const GwtfRaw R(f, G);
  const GwtfPackW P(FP);
  const float* rb = raw_c + (size_t)br * R.branch_size();
  int k0, k1, w0, w1;
  gwtf_pattern_dims(pat, &k0, &k1, &w0, &w1);
  // moments in double: Cov = E[xx] - E[x]E[x] cancels
  const double e0 = mom[k0] / n_total, e1 = k1 >= 0 ? mom[k1] / n_total : 0.0;
  const double c00 = mom[mom2_index(k0, k0)] / n_total - e0 * e0;
  const double c11 = k1 >= 0 ? mom[mom2_index(k1, k1)] / n_total - e1 * e1 : 0.0;
  const double c01 = k1 >= 0 ? mom[mom2_index(k0 < k1 ? k0 : k1, k0 < k1 ? k1 : k0)] / n_total - e0 * e1 : 0.0;
  const double wa = rb[R.sd0_w() + j], wb = rb[R.sd0_w() + f + j];
  const double mean = wa * e0 + wb * e1;
  double var = wa * wa * c00 + 2.0 * wa * wb * c01 + wb * wb * c11;
  if (var < 0.0) var = 0.0;
  const float* bn = rb + R.bn0();
  const float s = bn[j] / sqrtf((float)var + GWTF_BN_EPS);
  float* sd0 = pw_c + P.sd0(br) + (size_t)(j / 32) * 96 + (size_t)((j % 32) % 4) * 24 + (j % 32) / 4;
  sd0[0] = (float)wa * s;
  sd0[8] = (float)wb * s;
  sd0[16] = bn[f + j] - (float)mean * s;
  float* bb = bn_batch + ((size_t)br * 4 + 0) * 2 * f;
  bb[j] = (float)mean;
  bb[f + j] = (float)(var * (n_total / (n_total > 1.0 ? n_total - 1.0 : 1.0)));
}

// grid = B workgroups, thread = (branch, feature).  Workgroup 0 also emits the running-stat update of sd1_bn.
__global__ void fold1_kernel(const float* __restrict__ raw_c, const float* __restrict__ ystats, double n_total,
                             const float* __restrict__ film_raw, float* __restrict__ film_rec,
                             float* __restrict__ bn_batch, int c, int C, int f, int G, int FP) {
  const int t = threadIdx.x, b = blockIdx.x;
  if (t >= 2 * FP) return;
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
  for (int r = 0; r < GWTF_STAT_REPLICAS; ++r) {
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

extern "C" int gwtf_train_fold0(const float* raw_c, const float* moments, double n_total, int pattern, float* packed_w_c,
                                float* bn_batch_c, int f, int G, void* stream) {
  if (!raw_c || !moments || !packed_w_c || !bn_batch_c || f <= 0 || f > GWTF_MAX_FP || G <= 0 || pattern < 0 ||
      pattern > 5 || n_total < 1.0)
    return GWTF_E_BADARG;
  hipLaunchKernelGGL(fold0_kernel, dim3(1), dim3(2 * GWTF_MAX_FP), 0, (hipStream_t)stream, raw_c, moments, n_total, pattern,
                     packed_w_c, bn_batch_c, f, G, gwtf_padded_width(f));
  return (int)hipGetLastError();
}

extern "C" int gwtf_train_fold1(const float* raw_c, const float* ystats, double n_total, const float* film_raw,
                                float* film_rec, float* bn_batch_c, int c, int B, int C, int f, int G, void* stream) {
  if (!raw_c || !ystats || !film_raw || !film_rec || !bn_batch_c || f <= 0 || f > GWTF_MAX_FP || G <= 0 || B <= 0 ||
      c < 0 || c >= C || n_total < 1.0)
    return GWTF_E_BADARG;
  hipLaunchKernelGGL(fold1_kernel, dim3(B), dim3(2 * GWTF_MAX_FP), 0, (hipStream_t)stream, raw_c, ystats, n_total, film_raw,
                     film_rec, bn_batch_c, c, C, f, G, gwtf_padded_width(f));
  return (int)hipGetLastError();
}

// Whole train-mode forward of one stack on one rank (no cross-rank statistics): enqueues moments + 4 launches per
// coupling from C, so the host cost per coupling is four hipLaunchKernel calls instead of a Python iteration.
// Workspace (caller-owned, pre-zeroed where stated): moments [(C+1)][64][16] zero, ystats [C][64][2*FP*2] zero,
// bn_batch [C][2][4][2][f], film_rec [B][C][FS], xbuf [2][B][3][N].  Result coordinates end in
// xbuf[(C-1) & 1]; packed_w is modified (sd0 records filled in).
extern "C" int gwtf_train_forward(const float* p, const float* raw, float* packed_w, const float* film_raw,
                                  float* moments, float* ystats, float* bn_batch, float* film_rec, float* xbuf,
                                  float* logdet, float* ps, float* mus, float* logvars, int B, int N, int C, int f, int G,
                                  int pattern0, float eps, int mode, void* stream) {
  if (!p || !raw || !packed_w || !film_raw || !moments || !ystats || !bn_batch || !film_rec || !xbuf || !logdet)
    return GWTF_E_BADARG;
  if (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE) return GWTF_E_BADARG;
  const int FP = gwtf_padded_width(f);
  const size_t R = gwtf_raw_coupling_floats(f, G), PW = gwtf_packed_w_coupling_floats(f);
  const size_t MS = (size_t)GWTF_STAT_REPLICAS * 16, YS = (size_t)GWTF_STAT_REPLICAS * 2 * FP * 2;
  const size_t BS = (size_t)2 * 4 * 2 * f, XS = (size_t)B * 3 * N;
  const double n_total = (double)B * N;
  int rc = gwtf_train_moments(p, moments, B, N, stream);
  if (rc) return rc;
  const float* cur = p;
  for (int step = 0; step < C; ++step) {
    const int c = mode == GWTF_MODE_DIRECT ? step : C - 1 - step;
    const int pat = (pattern0 + c) % 6;
    rc = gwtf_train_fold0(raw + c * R, moments + step * MS, n_total, pat, packed_w + c * PW, bn_batch + c * BS, f, G, stream);
    if (rc) return rc;
    rc = gwtf_train_stats(cur, packed_w + c * PW, ystats + c * YS, B, N, f, pat, stream);
    if (rc) return rc;
    rc = gwtf_train_fold1(raw + c * R, ystats + c * YS, n_total, film_raw, film_rec, bn_batch + c * BS, c, B, C, f, G, stream);
    if (rc) return rc;
    float* nxt = xbuf + (size_t)(step & 1) * XS;
    rc = gwtf_train_apply(cur, packed_w, film_rec, nxt, step > 0 ? logdet : nullptr, logdet, ps, mus, logvars,
                          step + 1 < C ? moments + (step + 1) * MS : nullptr, c, B, N, C, f, pattern0, eps, mode, stream);
    if (rc) return rc;
    cur = nxt;
  }
  return 0;
}
