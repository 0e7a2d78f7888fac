// gwtf_stack_exact.hip -- the coupling stack with the f x f contraction (sd1) on the EXACT-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32: unsplit fp32 operands, fp32 accumulate -- the arithmetic of the reference's torch.matmul
// (lib/networks/layers.py:40-45 inside flows.py:95-117), with no operand range limit.  Three uses (docs/LOG.md round 5):
//   1. the on-device A/B of the split-f16 contraction of gwtf_stack.hip (tests/test_gpu_exact.py: every golden, the full grids);
//   2. the RE-RUN of flagged tiles: the split kernel marks a point whose coordinate left the f16-safe range (|x| > GWTF_X_LIMIT)
//      with NaN; launched with only_flagged = 1 on that launch's outputs, a workgroup here looks at its tile's results, leaves at
//      once when they are all finite (one read of 24 B per point, no staging) and otherwise recomputes the tile -- out-of-range
//      points come back with the reference's finite fp32 values instead of NaN (reference flows.py:113-115 has no such limit);
//   3. the honest comparison point of bench.py's roofline: the same stack priced on the unit whose dtype the result has.
// Same tile decomposition, FiLM record, epilogue and transcendental tail as the generic body of gwtf_stack.hip (coupling_body);
// only the contraction and its operand record (gwtf_layout.h GwtfPackX) differ.  One barrier per coupling, weights double-buffered
// through LDS by LDS-DMA while they fit (f <= 80), single-buffered beyond.
#include "gwtf_device.h"
#include <algorithm>

namespace {

using namespace gwtf_dev;

__device__ __forceinline__ float tail_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float tail_scale(float eps, float logvar) { return __builtin_amdgcn_sqrtf(eps + __expf(logvar)); }
__device__ __forceinline__ float tail_rscale(float eps, float logvar) { return __builtin_amdgcn_rsqf(eps + __expf(logvar)); }

template <int MB>
struct XCfg {
  static constexpr int FP = 16 * MB;
  static constexpr int KK = FP / 4;
  static constexpr int A32 = FP * FP;                                     // floats, one branch
  static constexpr int PX = (2 * A32 + 2 * FP * 4 + 255) / 256 * 256;     // GwtfPackX::coupling_size
  static constexpr int FS = 6 * FP + 4;
  static constexpr int FSP = (FS + 255) / 256 * 256;
  static constexpr int LAYER = PX + FSP;
};

constexpr int kLook = 8;      // tiles a workgroup of a re-run launch looks at

struct XJobs {
  int K;
  int tiles_cum[GWTF_MAX_COMPONENTS + 1];
  int begin[GWTF_MAX_COMPONENTS], end[GWTF_MAX_COMPONENTS];
};

// one elementary coupling on the wave's 16 NB points; L: staged GwtfPackX record followed by the shape's FiLM record
template <int MB, int NB, int MODE, bool KEEP2>
__device__ __forceinline__ void coupling_exact(const float* __restrict__ L, int kk4, int lane, int q, int k0, int k1, int w0, int w1,
                                               float eps, float s_keep, const float (&x)[NB][3], float (&xo)[3], float (&mu_d)[3],
                                               float (&lv_d)[3]) {
  using X = XCfg<MB>;
  constexpr int FP = X::FP;
  float xa[NB], xb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    xa[nb] = sel3(x[nb][0], x[nb][1], x[nb][2], k0);
    xb[nb] = KEEP2 ? sel3(x[nb][0], x[nb][1], x[nb][2], k1) : 0.f;
  }
  float res[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    const float* fe = L + X::PX + br * 3 * FP + 4 * q;            // c | w20a | w21a of this lane's 4 features per row block
    const float* a32 = L + br * X::A32 + lane;
    const f32x4* sd0 = reinterpret_cast<const f32x4*>(L + 2 * X::A32 + br * FP * 4);
    f32x4 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[m][nb] = *reinterpret_cast<const f32x4*>(fe + 16 * m);
    for (int t = 0; t < kk4; ++t) {                                 // k-step of 4 input features: this lane's is 4 t + q
      const f32x4 sp = sd0[4 * t + q];
      float a[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) a[m] = a32[(t * MB + m) * 64];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float pre = KEEP2 ? fmaf(sp[0], xa[nb], fmaf(sp[1], xb[nb], sp[2])) : fmaf(sp[0], xa[nb], sp[2]);
        const float h = fmaxf(pre, 0.f);
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], h, acc[m][nb], 0, 0, 0);
      }
    }
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
  const f32x4 bias = *reinterpret_cast<const f32x4*>(L + X::PX + 6 * FP);
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

template <int MB, int NB, int MODE, bool LISTS>
__global__ __launch_bounds__(256) void stack_exact_kernel(const float* __restrict__ p, const float* __restrict__ px,
                                                          const float* __restrict__ film, float* __restrict__ out,
                                                          float* __restrict__ logdet, float* __restrict__ ps, float* __restrict__ mus,
                                                          float* __restrict__ lvs, int B, int N, int C, int pattern0, float eps, int kk4,
                                                          const XJobs jobs, size_t p_stride_k, size_t out_stride_k, int only_flagged,
                                                          int* __restrict__ wl) {
  using X = XCfg<MB>;
  constexpr int NBUF = 2 * X::LAYER * 4 <= 160 * 1024 - 512 ? 2 : 1;
  __shared__ __align__(16) float lds[NBUF][X::LAYER];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, i16 = lane & 15;
  const int own_nb = q & (NB - 1);
  const int KC = jobs.K * C;
  const float s_keep = sqrtf(eps + 1.0f);
  const int c_start = MODE == GWTF_MODE_INVERSE ? C - 1 : 0;
  // A workgroup walks the tiles bid, bid + gridDim.x, ...: one tile each when every tile is computed (only_flagged = 0: grid = number
  // of tiles); in re-run mode the launch is a few fat workgroups (launch_exact: at most 512) that each LOOK at many tiles -- looking is
  // a 4-byte read per point -- and compute only the flagged ones: the re-run launch behind a clean forward pass cost 8 us as one
  // workgroup per tile (4096 workgroups with 45 KB of LDS each just to read 128 values and leave), ~3 us this way.
  // re-run mode, phase 1: ALL of this workgroup's tiles are looked at in one round of loads (a tile at a time would be a dependent
  // load + barrier per tile: 8 tiles x ~1 us behind a clean airplane launch), the flagged ones leave a bit in an LDS word
  // re-run mode WITH A WORK LIST (gwtf_stack_rerun_flagged): the split launch left {shape slot, first point} of every wave that flagged
  // a point in wl[2..], their number in wl[0]; a small fixed grid (64 workgroups) walks that list (an empty list -- every clean pass -- is one scalar
  // load per workgroup), checks the listed tile's flags (a tile can be listed by several of its waves: the first re-run clears them) and
  // recomputes it.  More entries than the list holds: every tile is a candidate.  The last workgroup to finish clears the counters.
  const bool wlmode = only_flagged && wl != nullptr;
  int n_cand = jobs.tiles_cum[jobs.K];
  bool listed = false;
  if (wlmode) {
    const int cnt = wl[0];
    if (cnt <= GWTF_WORKLIST_CAP) { n_cand = cnt; listed = true; }
  }
  __shared__ unsigned s_flagged;
  unsigned flagged_mask = 0xffffffffu;
  if (only_flagged && !wlmode) {
    if (threadIdx.x == 0) s_flagged = 0u;
    __syncthreads();
    // (all kLook loads are issued before the first is tested: tested one by one they are kLook dependent round trips -- 6.8 us for 8)
    float first[kLook];
#pragma unroll
    for (int i = 0; i < kLook; ++i) {
      const int bid = (int)blockIdx.x + i * (int)gridDim.x;
      first[i] = 0.f;
      if (bid < jobs.tiles_cum[jobs.K]) {
        int comp = 0;
        while (comp + 1 < jobs.K && bid >= jobs.tiles_cum[comp + 1]) ++comp;
        const int n_begin = jobs.begin[comp], n_end = jobs.end[comp];
        const int tps = (n_end - n_begin + 64 * NB - 1) / (64 * NB);
        const int local = bid - jobs.tiles_cum[comp];
        const int b = local / tps, tile = local - b * tps;
        const int n = n_begin + tile * 64 * NB + (int)threadIdx.x;          // 64 NB <= 256 points per tile: thread t looks at point t
        // (the split kernel sets ALL of a flagged point's coordinates and log-dets to NaN, and a non-finite value in any of them flags
        // the point: its first coordinate tells -- 4 B per point to read)
        if ((int)threadIdx.x < 64 * NB && n < n_end) first[i] = out[comp * out_stride_k + (size_t)b * 3 * N + n];
      }
    }
    unsigned mine = 0u;
#pragma unroll
    for (int i = 0; i < kLook; ++i) mine |= gwtf_nonfinite(first[i]) ? 1u << i : 0u;
    if (mine) atomicOr(&s_flagged, mine);
    __syncthreads();
    flagged_mask = s_flagged;
    if (flagged_mask == 0u) return;
  }
  int it = 0;
  for (int cand = blockIdx.x; cand < n_cand; cand += gridDim.x, ++it) {
  int bid = cand;
  if (wlmode) {
    if (listed) {
      const int slot = wl[2 + 2 * cand], n_first = wl[3 + 2 * cand];
      const int lc = slot / B, lb = slot - lc * B;
      const int ltps = (jobs.end[lc] - jobs.begin[lc] + 64 * NB - 1) / (64 * NB);
      bid = jobs.tiles_cum[lc] + lb * ltps + (n_first - jobs.begin[lc]) / (64 * NB);
    }
    int fc = 0;
    while (fc + 1 < jobs.K && bid >= jobs.tiles_cum[fc + 1]) ++fc;
    const int f_begin = jobs.begin[fc], f_end = jobs.end[fc];
    const int ftps = (f_end - f_begin + 64 * NB - 1) / (64 * NB);
    const int flocal = bid - jobs.tiles_cum[fc];
    const int fb = flocal / ftps, ftile = flocal - fb * ftps;
    const int fn = f_begin + ftile * 64 * NB + (int)threadIdx.x;
    float first = 0.f;
    if ((int)threadIdx.x < 64 * NB && fn < f_end) first = out[fc * out_stride_k + (size_t)fb * 3 * N + fn];
    if (!__syncthreads_or(gwtf_nonfinite(first) ? 1 : 0)) continue;
  } else if (only_flagged && !((flagged_mask >> (it & 31)) & 1u)) continue;      // (launch_exact sizes the grid so that it < kLook)
  int comp = 0;
  while (comp + 1 < jobs.K && bid >= jobs.tiles_cum[comp + 1]) ++comp;
  const int n_begin = jobs.begin[comp], n_end = jobs.end[comp];
  const int tiles_per_shape = (n_end - n_begin + 64 * NB - 1) / (64 * NB);
  const int local = bid - jobs.tiles_cum[comp];
  const int b = local / tiles_per_shape, tile = local - b * tiles_per_shape;
  const float* pk = p + comp * p_stride_k;
  float* outk = out + comp * out_stride_k;
  float* ldk = logdet + comp * out_stride_k;
  const size_t ls = (LISTS && out_stride_k) ? (size_t)comp * C * B * 3 * N : 0;
  const float* pxk = px + (size_t)comp * C * X::PX;
  const int n_wave0 = n_begin + (tile * 4 + wave) * 16 * NB;
  const int n_own = n_wave0 + 16 * own_nb + i16;
  const bool own_inrange = n_own < n_end, own_valid = own_inrange && q < NB;


  auto stage = [&](int buf, int c) {
    const float* src_w = pxk + (size_t)c * X::PX;
    const float* src_f = film + ((size_t)b * KC + (size_t)comp * C + c) * X::FS;
    const unsigned voff = lane * 16u;
    for (int piece = wave; piece < X::PX / 256; piece += 4)
      __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(src_w + piece * 256) + voff),
                                       (lds_void*)&lds[buf][piece * 256], 16, 0, 0);
    if (wave < X::FSP / 256 && wave * 256 + lane * 4 < X::FS)
      __builtin_amdgcn_global_load_lds((glb_void*)(reinterpret_cast<const char*>(src_f + wave * 256) + voff),
                                       (lds_void*)&lds[buf][X::PX + wave * 256], 16, 0, 0);
  };
  stage(0, c_start);
  float xo[3], ld[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 3; ++d) xo[d] = own_inrange ? pk[((size_t)b * 3 + d) * N + n_own] : 0.f;
  float x[NB][3];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int d = 0; d < 3; ++d) x[nb][d] = __shfl(xo[d], 16 * nb + i16);
  float mu_last[3] = {0.f, 0.f, 0.f}, lv_last[3] = {0.f, 0.f, 0.f};
  bool bad = false;          // a NaN / Inf coordinate at the input of any coupling (v_max would turn NaN activations into zeros)
#pragma unroll
  for (int d = 0; d < 3; ++d) bad |= gwtf_nonfinite(xo[d]);

  for (int step = 0; step < C; ++step) {
    const int c = MODE == GWTF_MODE_INVERSE ? c_start - step : c_start + step;
    const int buf = NBUF == 2 ? step & 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NBUF == 2 && step + 1 < C) stage(buf ^ 1, MODE == GWTF_MODE_INVERSE ? c - 1 : c + 1);
    const int pat = (pattern0 + c) % 6;
    int k0, k1, w0, w1;
    gwtf_pattern_dims(pat, &k0, &k1, &w0, &w1);
    float mu_d[3], lv_d[3];
    if (pat < 3) coupling_exact<MB, NB, MODE, true>(lds[buf], kk4, lane, q, k0, k1, w0, w1, eps, s_keep, x, xo, mu_d, lv_d);
    else coupling_exact<MB, NB, MODE, false>(lds[buf], kk4, lane, q, k0, k1, w0, w1, eps, s_keep, x, xo, mu_d, lv_d);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ld[d] += lv_d[d];
      bad |= gwtf_nonfinite(xo[d]);
    }
    if (LISTS && step + 1 == C) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { mu_last[d] = mu_d[d]; lv_last[d] = lv_d[d]; }
    } else if (LISTS && own_valid) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t o = ls + (((size_t)c * B + b) * 3 + d) * N + n_own;
        ps[o] = xo[d]; mus[o] = mu_d[d]; lvs[o] = lv_d[d];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int d = 0; d < 3; ++d) x[nb][d] = __shfl(xo[d], 16 * nb + i16);
    if (NBUF == 1 && step + 1 < C) {
      __syncthreads();
      stage(0, MODE == GWTF_MODE_INVERSE ? c - 1 : c + 1);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) bad |= gwtf_nonfinite(ld[d]);
  if (bad) {     // non-finite anywhere on the point's way -> NaN coordinates AND log-det (the reference's isnan(loss) guard, training.py:43-46)
    const float qnan = __builtin_bit_cast(float, 0x7fc00000u);
#pragma unroll
    for (int d = 0; d < 3; ++d) xo[d] = ld[d] = mu_last[d] = lv_last[d] = qnan;
  }
  if (own_valid) {
    if (LISTS) {
      const int c_last = MODE == GWTF_MODE_INVERSE ? 0 : C - 1;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const size_t o = ls + (((size_t)c_last * B + b) * 3 + d) * N + n_own;
        ps[o] = xo[d]; mus[o] = mu_last[d]; lvs[o] = lv_last[d];
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const size_t o = ((size_t)b * 3 + d) * N + n_own;
      outk[o] = xo[d];
      ldk[o] = ld[d];
    }
  }
  __syncthreads();      // every wave is done with the staged weights before the next flagged tile's staging overwrites them
  }   // tiles of this workgroup
  if (wlmode && threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&wl[1], 1) == (int)gridDim.x - 1) {       // every workgroup has read the count: clear it for the next split launch
      wl[0] = 0;
      wl[1] = 0;
    }
  }
}

template <int MB, int NB>
int launch_exact(const float* p, const float* px, const float* film, float* out, float* logdet, float* ps, float* mus, float* lvs,
                 int B, int N, int C, int pattern0, float eps, int mode, int kk4, const int* segs, int K, size_t p_stride_k,
                 size_t out_stride_k, int only_flagged, int* wl, hipStream_t st) {
  XJobs jobs;
  jobs.K = K;
  jobs.tiles_cum[0] = 0;
  for (int k = 0; k < K; ++k) {
    jobs.begin[k] = segs ? segs[2 * k] : 0;
    jobs.end[k] = segs ? segs[2 * k + 1] : N;
    jobs.tiles_cum[k + 1] = jobs.tiles_cum[k] + B * ((jobs.end[k] - jobs.begin[k] + 64 * NB - 1) / (64 * NB));
  }
  if (jobs.tiles_cum[K] == 0) return 0;
  // re-run launches: a workgroup LOOKS at kLook tiles (their flags are bits of one LDS word)
  static const int rerun_wgs = [] {
    const char* e = getenv("GWTF_RERUN_WGS");
    const int v = e ? atoi(e) : 0;
    // a clean pass pays for the DISPATCH of these workgroups (45 KB of LDS each) and nothing else: 7.0 / 4.6 / 2.8 / 2.2 us for
    // 256 / 128 / 64 / 32 of them (airplane launch, inside a hipGraph); 64 keeps a pass with up to 64 flagged tiles in one round
    return v > 0 ? v : 64;
  }();
  const unsigned n_wl = (unsigned)(jobs.tiles_cum[K] < rerun_wgs ? jobs.tiles_cum[K] : rerun_wgs);
  const dim3 grid(only_flagged ? (wl ? n_wl : (unsigned)((jobs.tiles_cum[K] + kLook - 1) / kLook)) : (unsigned)jobs.tiles_cum[K]), block(256);
#define GWTF_X(MODE_, LISTS_)                                                                                                    \
  hipLaunchKernelGGL((stack_exact_kernel<MB, NB, MODE_, LISTS_>), grid, block, 0, st, p, px, film, out, logdet, ps, mus, lvs, B, \
                     N, C, pattern0, eps, kk4, jobs, p_stride_k, out_stride_k, only_flagged, wl)
  if (mode == GWTF_MODE_DIRECT) { if (ps) GWTF_X(GWTF_MODE_DIRECT, true); else GWTF_X(GWTF_MODE_DIRECT, false); }
  else { if (ps) GWTF_X(GWTF_MODE_INVERSE, true); else GWTF_X(GWTF_MODE_INVERSE, false); }
#undef GWTF_X
  return (int)hipGetLastError();
}

}  // namespace

// see include/gwtf.h
static int exact_dispatch(const float* p, const float* packed_x, const float* film, float* out, float* logdet, float* ps,
                          float* mus, float* logvars, const int* segments, int K, int B, int N, int C, int f,
                          int pattern0, float eps, int mode, size_t p_stride_k, size_t out_stride_k, int only_flagged, int* wl,
                          int tune, void* stream) {
  if (B <= 0 || N <= 0 || C <= 0 || f <= 0 || f > GWTF_MAX_FP || K <= 0 || K > GWTF_MAX_COMPONENTS || !p || !packed_x || !film || !out ||
      !logdet || (mode != GWTF_MODE_DIRECT && mode != GWTF_MODE_INVERSE) || pattern0 < 0 || pattern0 > 5)
    return GWTF_E_BADARG;
  if ((ps || mus || logvars) && !(ps && mus && logvars)) return GWTF_E_BADARG;
  for (int k = 0; k < K; ++k) {
    const int b0 = segments ? segments[2 * k] : 0, e0 = segments ? segments[2 * k + 1] : N;
    if (b0 < 0 || e0 < b0 || e0 > N) return GWTF_E_BADARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int kk4 = (f + 3) / 4;
  const int forced = (tune & 0xffff) / 16;
  // 32 points per wave (16 beyond f = 64: the accumulators of both point blocks would not fit); GWTF_TUNE_POINTS_PER_WAVE overrides
  const int nb = (forced == 1 || forced == 2) ? forced : (f > 64 ? 1 : 2);
#define GWTF_XA p, packed_x, film, out, logdet, ps, mus, logvars, B, N, C, pattern0, eps, mode, kk4, segments, K, p_stride_k, out_stride_k, only_flagged, wl, st
#define GWTF_XM(MB_) return nb == 1 ? launch_exact<MB_, 1>(GWTF_XA) : launch_exact<MB_, 2>(GWTF_XA);
  switch (gwtf_padded_width(f) / 16) {
    case 1: GWTF_XM(1)
    case 2: GWTF_XM(2)
    case 3: GWTF_XM(3)
    case 4: GWTF_XM(4)
    case 5: GWTF_XM(5)
    case 6: GWTF_XM(6)
    case 7: GWTF_XM(7)
    case 8: GWTF_XM(8)
    default: return GWTF_E_BADARG;
  }
#undef GWTF_XM
#undef GWTF_XA
}

extern "C" int gwtf_stack_forward_exact(const float* p, const float* packed_x, const float* film, float* out, float* logdet, float* ps,
                                        float* mus, float* logvars, const int* segments, int K, int B, int N, int C, int f,
                                        int pattern0, float eps, int mode, size_t p_stride_k, size_t out_stride_k, int only_flagged,
                                        int tune, void* stream) {
  return exact_dispatch(p, packed_x, film, out, logdet, ps, mus, logvars, segments, K, B, N, C, f, pattern0, eps, mode, p_stride_k,
                        out_stride_k, only_flagged, nullptr, tune, stream);
}

extern "C" int gwtf_stack_rerun_flagged(const float* p, const float* packed_x, const float* film, float* out, float* logdet, float* ps,
                                        float* mus, float* logvars, const int* segments, int K, int B, int N, int C, int f,
                                        int pattern0, float eps, int mode, size_t p_stride_k, size_t out_stride_k, int* worklist,
                                        int tune, void* stream) {
  if (!worklist) return GWTF_E_BADARG;
  return exact_dispatch(p, packed_x, film, out, logdet, ps, mus, logvars, segments, K, B, N, C, f, pattern0, eps, mode, p_stride_k,
                        out_stride_k, 1, worklist, tune, stream);
}
