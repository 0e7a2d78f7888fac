// gwtf_gemm.h -- one-workgroup fp32 GEMM + column reductions for the per-SHAPE Linear -> BatchNorm -> Swish heads (gwtf_heads.hip).  Exact fp32 products on
// v_mfma_f32_16x16x4_f32: these ops are latency-bound chains of a few MFLOP, nothing to gain from reduced-precision tricks.
#ifndef GWTF_GEMM_H
#define GWTF_GEMM_H
#include <hip/hip_runtime.h>

namespace gwtf_gemm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;

constexpr int kMaxM = 128;   // rows one gemm_direct call covers (4 accumulators per wave); longer batches: one call per 128 rows

typedef const __attribute__((address_space(1))) float* gptr_c;   // global address space spelled out: a NOINLINE device
typedef __attribute__((address_space(1))) float* gptr;           // function's plain pointers are generic (flat_load, slow)
typedef __attribute__((address_space(3))) float* lptr;           // LDS, for the same reason

// Between phases the wavefronts of THE workgroup exchange data through global memory.  __syncthreads() carries a
// workgroup-scope release / acquire fence, which is all the AMDGPU memory model asks for: the wavefronts of one workgroup
// share their compute unit's L1, and stores are write-through.
__device__ inline void phase_sync() { __syncthreads(); }

__device__ inline float swishf(float h) { return h / (1.0f + expf(-h)); }

// Per-column sums over the B rows with ALL threads: thread t owns column t % ncols and the rows rg, rg + RG, ... with
// rg = t / ncols, RG = kThreads / ncols (a one-thread-per-column loop is B dependent round trips).  Eight loads in flight
// per thread; partials combined through LDS (red: RG * ncols floats).
struct ColMap { int col, rg, RG; bool on; };
__device__ inline ColMap col_map(int ncols) {
  ColMap m;
  m.RG = kThreads / ncols;
  m.col = threadIdx.x % ncols;
  m.rg = threadIdx.x / ncols;
  m.on = m.rg < m.RG;
  return m;
}
template <class FV>
__device__ __forceinline__ float col_sum(const ColMap& m, int ncols, int B, FV val, float* __restrict__ red) {
  float s = 0.f;
  if (m.on) {
    int b = m.rg;
    for (; b + 7 * m.RG < B; b += 8 * m.RG) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = val(b + u * m.RG, m.col);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < B; b += m.RG) s += val(b, m.col);
    red[m.rg * ncols + m.col] = s;
  }
  __syncthreads();
  float tot = 0.f;
  if (m.on)
    for (int r = 0; r < m.RG; ++r) tot += red[r * ncols + m.col];
  __syncthreads();
  return tot;
}

// The same sum over values a thread already HOLDS: v[i] belongs to row rg + i RG (B <= KEEP RG).  A layer's kernels use a column's
// values two or three times (mean, variance, output; the backward's two sums and dL/dy): fetched once, each use is one LDS round
// instead of a dependent round trip to global memory.  Same order of additions as col_sum's loop for these sizes: the same bits.
template <int KEEP>
__device__ __forceinline__ float col_sum_kept(const ColMap& m, int ncols, int B, const float (&v)[KEEP], float* __restrict__ red) {
  if (m.on) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < KEEP; ++i)
      if (m.rg + i * m.RG < B) s += v[i];
    red[m.rg * ncols + m.col] = s;
  }
  __syncthreads();
  float tot = 0.f;
  if (m.on)
    for (int r = 0; r < m.RG; ++r) tot += red[r * ncols + m.col];
  __syncthreads();
  return tot;
}

// C (M x N) [+]= A (M x K) . B^T (N x K), M <= 128, any N, for the whole workgroup (8 wavefronts), operands read STRAIGHT from global memory /
// L2 into the MFMA's registers (v_mfma_f32_16x16x4_f32: exact fp32 products) -- no LDS staging, no barriers: the staged version
// of gwtf_gemm.h (built for the prior flow's chain of tiny dependent products) spent ~6 us per 64-wide K chunk on scalar
// staging loads here (g_posterior forward + backward 550 us against 150 us for the library path; tools/diag/heads_time.py).
// A(i, k) at A + i*sai + k*sak, B(j, k) at Bm + j*sbj + k*sbk.  A K step covers 16 k values: lane (r = lane & 15, q = lane >> 4)
// holds k = k0 + 4q .. 4q+3 of row r -- ONE 16-byte load when k is the contiguous index (sak == 1, everything 16-byte aligned),
// four 4-byte loads otherwise (coalesced over the 16 rows when the row index is the contiguous one) -- and MFMA j of the step
// consumes element j: both operands use the same k order, which is all a contraction needs.
// Wave w owns column tile w & 3 and the row tiles (w >> 2), (w >> 2) + 2, ...: up to 4 accumulators.
// Loads carry NO lane-dependent control flow and no select between them (a select after a load is a wait for THAT load: the
// first version serialised its 12 loads per pass that way, 57 us per launch): rows beyond M / N are read from a clamped
// (valid) address and simply never stored -- a row of A or B only feeds its own output row / column -- and only the K tail,
// which does feed valid outputs, is zeroed, after all loads of the pass have been issued.
template <bool VEC>
__device__ __forceinline__ f32x4 load_k4_full(const float* __restrict__ rowp, long sk, int k) {
  f32x4 v;
  if (VEC) {
    v = *reinterpret_cast<const f32x4*>(rowp + k);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rowp[(long)(k + j) * sk];
  }
  return v;
}
__device__ __forceinline__ f32x4 load_k4_tail(const float* __restrict__ rowp, long sk, int k, int K) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = rowp[(long)min(k + j, K - 1) * sk];
  return v;
}
__device__ __forceinline__ f32x4 zero_tail(f32x4 v, int k, int K) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = k + j < K ? v[j] : 0.f;
  return v;
}
template <bool AVEC, bool BVEC, int CNT>
__device__ __forceinline__ void gemm_wave(int M, int N, int K, const float* __restrict__ A, long sai, long sak,
                                          const float* __restrict__ Bm, long sbj, long sbk, float* __restrict__ C, long ldc,
                                          bool accumulate, int nt, int m0, int r16, int q) {
  constexpr int U = 4;                                         // a pass = 4 steps of 16 k: one memory round trip per 64 k
  f32x4 acc[CNT];
  const float* arow[CNT];
#pragma unroll
  for (int u = 0; u < CNT; ++u) {
    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    arow[u] = A + (long)min(16 * (m0 + 2 * u) + r16, M - 1) * sai;
  }
  const float* brow = Bm + (long)min(16 * nt + r16, N - 1) * sbj;
  int k0 = 0;
  // (Measured and rejected: issuing pass p + 1's loads before pass p's MFMAs -- two passes of operands in registers -- made
  // every product SLOWER, 255 -> 344 us for the g_posterior module: hipcc serialises the doubled register set.)
#pragma unroll 1
  for (; k0 + 16 * U <= K; k0 += 16 * U) {
    f32x4 a[U][CNT], b[U];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      b[s] = load_k4_full<BVEC>(brow, sbk, ks);
#pragma unroll
      for (int u = 0; u < CNT; ++u) a[s][u] = load_k4_full<AVEC>(arow[u], sak, ks);
    }
#pragma unroll
    for (int s = 0; s < U; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < CNT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][u][j], b[s][j], acc[u], 0, 0, 0);
  }
  if (k0 < K) {                                                // the K tail (wave-uniform branch): clamped loads, then zeros
    f32x4 ta[U][CNT], tb[U];
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      tb[s] = load_k4_tail(brow, sbk, ks, K);
#pragma unroll
      for (int u = 0; u < CNT; ++u) ta[s][u] = load_k4_tail(arow[u], sak, ks, K);
    }
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int ks = k0 + 16 * s + 4 * q;
      tb[s] = zero_tail(tb[s], ks, K);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < CNT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s][u][j], tb[s][j], acc[u], 0, 0, 0);
    }
  }
  const int n = 16 * nt + r16;
  if (n < N) {
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * (m0 + 2 * u) + 4 * q + r;
        if (m < M) {
          const long o = (long)m * ldc + n;
          C[o] = accumulate ? C[o] + acc[u][r] : acc[u][r];
        }
      }
    }
  }
}
// NOT inlined: the prior flow calls it 14 times per flow step and an inlined copy per call site (x 12 template variants) made the
// kernel's code footprint the bottleneck (the products took ~20 us each: instruction-cache misses).  One body per (AVEC, BVEC):
// the row-tile count of the wave is a switch inside.
template <bool AVEC, bool BVEC>
static __device__ __attribute__((noinline)) void gemm_direct_t(int M, int N, int K, gptr_c A, long sai, long sak, gptr_c Bm, long sbj,
                                                               long sbk, gptr C, long ldc, bool accumulate) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
  const int MT = (M + 15) / 16, NT = (N + 15) / 16;
  const int m0 = wave >> 2;                                     // wave w: column tiles (w & 3) + 4 g, row tiles (w >> 2), (w >> 2) + 2, ...
  if (m0 >= MT) return;
  const float* Ag = (const float*)A;
  const float* Bg = (const float*)Bm;
  float* Cg = (float*)C;
#pragma unroll 1
  for (int nt = wave & 3; nt < NT; nt += 4) {
    switch ((MT - m0 + 1) / 2) {                               // row tiles of this wave: wave-uniform
      case 1: gemm_wave<AVEC, BVEC, 1>(M, N, K, Ag, sai, sak, Bg, sbj, sbk, Cg, ldc, accumulate, nt, m0, r16, q); break;
      case 2: gemm_wave<AVEC, BVEC, 2>(M, N, K, Ag, sai, sak, Bg, sbj, sbk, Cg, ldc, accumulate, nt, m0, r16, q); break;
      case 3: gemm_wave<AVEC, BVEC, 3>(M, N, K, Ag, sai, sak, Bg, sbj, sbk, Cg, ldc, accumulate, nt, m0, r16, q); break;
      default: gemm_wave<AVEC, BVEC, 4>(M, N, K, Ag, sai, sak, Bg, sbj, sbk, Cg, ldc, accumulate, nt, m0, r16, q); break;
    }
  }
}
__device__ __forceinline__ void gemm_direct(int M, int N, int K, const float* __restrict__ A, long sai, long sak,
                                            const float* __restrict__ Bm, long sbj, long sbk, float* __restrict__ C, long ldc,
                                            bool accumulate) {
  // 16-byte loads where k is the contiguous index and every row starts 16-byte aligned (wave-uniform decision)
  const bool avec = sak == 1 && (sai & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
  const bool bvec = sbk == 1 && (sbj & 3) == 0 && (reinterpret_cast<size_t>(Bm) & 15) == 0;
  if (avec && bvec) gemm_direct_t<true, true>(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate);
  else if (avec) gemm_direct_t<true, false>(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate);
  else gemm_direct_t<false, false>(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate);
}

// ---- split-K product for LONG contractions, NARROW output blocks ----------------------------------------------------------------------
// The exact fp32 MFMA (v_mfma_f32_16x16x4_f32, 32 cycles) makes even these small products THROUGHPUT-bound when a workgroup owns 64
// output columns: 64 x 64 x 512 is 2048 MFMAs = 16 k cycles on the four SIMDs of ONE compute unit (7 us) while 248 others idle
// (measured: 12-24 us per head-layer launch).  So the head kernels give a workgroup only 16 * NJ columns (NJ = 1: 32 workgroups for a
// 512-wide layer), and within it the EIGHT WAVES CUT K: wave w contracts the slice [w S, (w + 1) S) (S = K / 8 rounded up to 16) for
// the four row tiles of a 64-row block -- its loads go out in one or two rounds instead of K / 64 dependent passes -- writes its
// partial block to its own LDS slab, and all threads add the slabs in wave order (deterministic: no float atomics) into C [+]=.
// slab: 8 * 64 * 16 NJ floats of LDS.  Operand description as gemm_direct.  Not inlined.
template <bool AVEC, bool BVEC, int NJ>
static __device__ __attribute__((noinline)) void gemm_splitk_t(int M, int N, int K, gptr_c Ag, long sai, long sak, gptr_c Bg, long sbj,
                                                               long sbk, gptr Cg, long ldc, bool accumulate, lptr slab) {
  constexpr int NC = 16 * NJ, kSlab = 64 * NC;
  const float* A = (const float*)Ag;
  const float* Bm = (const float*)Bg;
  float* C = (float*)Cg;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, q = lane >> 4;
  const int MT = (M + 15) / 16;
  const int S = ((K + 7) / 8 + 15) / 16 * 16;
  const int nw = (K + S - 1) / S;                                // waves with a non-empty slice
  const int k_lo = wave * S, k_hi = min(K, k_lo + S);
  const float* brow[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) brow[j] = Bm + (long)min(16 * j + r16, N - 1) * sbj;
  lptr mine = slab + wave * kSlab;
#pragma unroll 1
  for (int mh = 0; mh < MT; mh += 4) {                           // 64 rows at a time: 4 NJ accumulators per wave
    if (wave < nw) {
      f32x4 acc[4][NJ];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* arow[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) arow[u] = A + (long)min(16 * (mh + u) + r16, M - 1) * sai;
#pragma unroll 1
      for (int k0 = k_lo; k0 < k_hi; k0 += 64) {                 // four 16-k steps per round
        const bool tail = k0 + 64 > K;                           // wave-uniform
        f32x4 a[4][4], b[4][NJ];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (k0 + 16 * s >= k_hi) break;                        // wave-uniform: a slice may end inside a round
          const int ks = k0 + 16 * s + 4 * q;
#pragma unroll
          for (int j = 0; j < NJ; ++j) b[s][j] = tail ? load_k4_tail(brow[j], sbk, ks, K) : load_k4_full<BVEC>(brow[j], sbk, ks);
#pragma unroll
          for (int u = 0; u < 4; ++u) a[s][u] = tail ? load_k4_tail(arow[u], sak, ks, K) : load_k4_full<AVEC>(arow[u], sak, ks);
        }
        if (tail) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              if (k0 + 16 * s < k_hi) b[s][j] = zero_tail(b[s][j], k0 + 16 * s + 4 * q, K);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (k0 + 16 * s < k_hi) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                  acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][u][t], b[s][j][t], acc[u][j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) mine[(16 * u + 4 * q + r) * NC + 16 * j + r16] = acc[u][j][r];
    }
    __syncthreads();
    for (int t = tid; t < kSlab; t += kThreads) {
      const int m = 16 * mh + t / NC, n = t % NC;
      if (m < M && n < N) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += slab[w * kSlab + t];
        const long o = (long)m * ldc + n;
        C[o] = accumulate ? C[o] + v : v;
      }
    }
    __syncthreads();
  }
}
template <int NJ>
__device__ __forceinline__ void gemm_splitk(int M, int N, int K, const float* __restrict__ A, long sai, long sak,
                                            const float* __restrict__ Bm, long sbj, long sbk, float* __restrict__ C, long ldc,
                                            bool accumulate, float* slab) {
  const bool avec = sak == 1 && (sai & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
  const bool bvec = sbk == 1 && (sbj & 3) == 0 && (reinterpret_cast<size_t>(Bm) & 15) == 0;
  if (avec && bvec) gemm_splitk_t<true, true, NJ>(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate, (lptr)slab);
  else if (avec) gemm_splitk_t<true, false, NJ>(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate, (lptr)slab);
  else gemm_splitk_t<false, false, NJ>(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate, (lptr)slab);
}

// ---- C (M <= 16 MU x N) = A^T-in-LDS . B for a SHORT contraction and a WIDE output (the weight gradient of a head layer: M = the
// workgroup's output columns of the layer, K = the batch rows (<= 128), N = Din) -- the waves cut N: wave w owns column tiles w, w + 8,
// ... with all MU row tiles.  A(i, k) = at[k * pitch + i] sits in LDS (the dL/dy block the workgroup has just produced: rows k >= K must be finite,
// they meet zeros); B(j, k) = Bm[j + k * sbk] (j contiguous: coalesced over the 16 lanes of a row group).
template <int MU>
static __device__ __attribute__((noinline)) void gemm_nsplit_lds(int M, int N, int K, lptr at, int pitch, gptr_c Bg, long sbk, gptr Cg,
                                                                 long ldc, bool accumulate) {
  const float* Bm = (const float*)Bg;
  float* C = (float*)Cg;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, q = lane >> 4;
  const int NT = (N + 15) / 16;
#pragma unroll 1
  for (int nt = wave; nt < NT; nt += 8) {
    f32x4 acc[MU];
#pragma unroll
    for (int u = 0; u < MU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bcol = Bm + min(16 * nt + r16, N - 1);
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += 64) {
      f32x4 b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = load_k4_tail(bcol, sbk, k0 + 16 * s + 4 * q, K);     // clamped: always a valid address
      if (k0 + 64 > K) {
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = zero_tail(b[s], k0 + 16 * s + 4 * q, K);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (k0 + 16 * s < K) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int k = k0 + 16 * s + 4 * q + t;
#pragma unroll
            for (int u = 0; u < MU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[k * pitch + 16 * u + r16], b[s][t], acc[u], 0, 0, 0);
          }
        }
      }
    }
    const int n = 16 * nt + r16;
    if (n < N) {
#pragma unroll
      for (int u = 0; u < MU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * u + 4 * q + r;
          if (m < M) {
            const long o = (long)m * ldc + n;
            C[o] = accumulate ? C[o] + acc[u][r] : acc[u][r];     // the same lane owns the element in every call: row blocks of a long batch add up in order
          }
        }
    }
  }
}

}  // namespace gwtf_gemm
#endif  // GWTF_GEMM_H
