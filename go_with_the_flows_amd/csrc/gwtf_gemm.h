// gwtf_gemm.h -- one-workgroup fp32 GEMM + column reductions for the per-SHAPE parts of the model (B <= 128 rows): the global prior
// flow (gwtf_prior.hip) and the Linear -> BatchNorm -> Swish heads (gwtf_heads.hip).  Exact fp32 products on
// v_mfma_f32_16x16x4_f32: these ops are latency-bound chains of a few MFLOP, nothing to gain from reduced-precision tricks.
#ifndef GWTF_GEMM_H
#define GWTF_GEMM_H
#include <hip/hip_runtime.h>

namespace gwtf_gemm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;

// C (M x N) [+]= A (M x K) . B^T (N x K) for the whole workgroup.  Operands are described by element strides (A(i,k) at
// A + i*sai + k*sak, B(j,k) at Bm + j*sbj + k*sbk: row-major, transposed and strided gathers alike) and staged through LDS
// in K chunks of 64 by ALL threads -- every load of a chunk is issued before the first LDS store, so a staging step is ONE
// memory round trip -- then contracted on v_mfma_f32_16x16x4_f32 from LDS (exact fp32 products).  Wavefront w owns the tile
// columns w and w + 8 and every row tile (<= 16 accumulators: M <= 128, N <= 256); C is written row-major with pitch ldc.
// Everything that is not a GEMM (BatchNorm, Swish, the affine map) runs as separate flat element loops over all threads:
// the first versions ran such epilogues inside the unrolled tile loop (hundreds of spilled registers) or read operands
// straight from L2 inside the MFMA loop (a dependent round trip per MFMA): 2.2 - 2.4 ms per 14-flow forward.
// Row pitch 65: the 16 rows of an operand fragment fall into different LDS banks.
constexpr int kKC = 64, kPitch = kKC + 1, kMaxM = 128, kMaxN = 256, kMT = kMaxM / 16;

typedef const __attribute__((address_space(1))) float* gptr_c;   // global address space spelled out: a NOINLINE device
typedef __attribute__((address_space(1))) float* gptr;           // function's plain pointers are generic (flat_load, slow)

// Not inlined on purpose: the kernels call it 4 - 10 times per flow, and inlined copies (64 accumulator + 32 staging
// registers each, scheduled together) spilled several hundred registers.  LDS buffers are function-local statics (one
// allocation shared by all calls).
static __device__ __attribute__((noinline)) void gemm_staged_impl(int M, int N, int K, gptr_c A, long sai, long sak, gptr_c Bm, long sbj,
                                                           long sbk, gptr C, long ldc, bool accumulate) {
  __shared__ float As[kMaxM * kPitch], Bs[kMaxN * kPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, q = lane >> 4;
  const int MT = (M + 15) / 16, NT = (N + 15) / 16;
  f32x4 acc[2 * kMT];
#pragma unroll
  for (int u = 0; u < 2 * kMT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = (sak < 0 ? -sak : sak) <= (sai < 0 ? -sai : sai);
  const bool b_kfast = (sbk < 0 ? -sbk : sbk) <= (sbj < 0 ? -sbj : sbj);
  const int Mp = MT * 16, Np = NT * 16;
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += kKC) {
    __syncthreads();                         // the previous chunk's fragments have been read
    // Branch-free staging.  K-contiguous operand: thread -> (row (tid >> 6) + 8u, k = tid & 63); K-major operand: thread ->
    // (row tid & (W-1), k = tid / W + u * (512 / W)).  One offset per thread, constant increments per element; out-of-range
    // elements load a clamped (valid) address and are replaced by 0 with a select.  (Per-element index arithmetic with
    // run-time selects and predicated loads cost ~10 K VALU + 3 K branches per flow and wavefront in the first versions.)
    constexpr int kDA = kMaxM * kKC / kThreads, kDB = kMaxN * kKC / kThreads, kHB = kDB / 2;
    const int isai = (int)sai, isak = (int)sak, isbj = (int)sbj, isbk = (int)sbk;
    const int tid = threadIdx.x;
    const int ai0 = a_kfast ? tid >> 6 : tid & (kMaxM - 1), ak0 = a_kfast ? tid & 63 : tid / kMaxM;
    const int adi = a_kfast ? kThreads / 64 : 0, adk = a_kfast ? 0 : kThreads / kMaxM;
    const int bj0 = b_kfast ? tid >> 6 : tid & (kMaxN - 1), bk0 = b_kfast ? tid & 63 : tid / kMaxN;
    const int bdj = b_kfast ? kThreads / 64 : 0, bdk = b_kfast ? 0 : kThreads / kMaxN;
    float va[kDA], vb[kHB];
#pragma unroll
    for (int u = 0; u < kDA; ++u) {
      const int i = ai0 + u * adi, k = k0 + ak0 + u * adk;
      const float v = A[min(i, M - 1) * isai + min(k, K - 1) * isak];
      va[u] = (i < M && k < K) ? v : 0.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int u = 0; u < kHB; ++u) {
        const int uu = h * kHB + u;
        const int j = bj0 + uu * bdj, k = k0 + bk0 + uu * bdk;
        const float v = Bm[min(j, N - 1) * isbj + min(k, K - 1) * isbk];
        vb[u] = (j < N && k < K) ? v : 0.f;
      }
      if (h == 0) {
#pragma unroll
        for (int u = 0; u < kDA; ++u) {
          const int i = ai0 + u * adi, k = ak0 + u * adk;
          if (i < Mp) As[i * kPitch + k] = va[u];
        }
      }
#pragma unroll
      for (int u = 0; u < kHB; ++u) {
        const int uu = h * kHB + u;
        const int j = bj0 + uu * bdj, k = bk0 + uu * bdk;
        if (j < Np) Bs[j * kPitch + k] = vb[u];
      }
    }
    __syncthreads();
    const int kc = K - k0 < kKC ? K - k0 : kKC;
#pragma unroll 2
    for (int kk = 0; kk < kc; kk += 4) {
      float a[kMT];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) a[mt] = As[(16 * mt + i16) * kPitch + kk + q];      // rows >= Mp: stale but finite, never stored
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int nt = wave + c * kWaves;
        if (nt < NT) {
          const float b = Bs[(16 * nt + i16) * kPitch + kk + q];
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt)
            if (mt < MT) acc[c * kMT + mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b, acc[c * kMT + mt], 0, 0, 0);
        }
      }
    }
  }
  const int ildc = (int)ldc;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      const int nt = wave + c * kWaves, n = 16 * nt + i16;
      __builtin_amdgcn_sched_barrier(0);       // one tile at a time: 64 stores scheduled together need 64 addresses at once
      if (nt < NT && mt < MT && n < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * mt + 4 * q + r;
          if (m < M) {
            const int o = m * ildc + n;
            C[o] = accumulate ? C[o] + acc[c * kMT + mt][r] : acc[c * kMT + mt][r];
          }
        }
      }
    }
}

__device__ __forceinline__ void gemm_staged(int M, int N, int K, const float* A, long sai, long sak, const float* Bm, long sbj,
                                            long sbk, float* C, long ldc, bool accumulate, float*, float*) {
  gemm_staged_impl(M, N, K, (gptr_c)A, sai, sak, (gptr_c)Bm, sbj, sbk, (gptr)C, ldc, accumulate);
}

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


}  // namespace gwtf_gemm
#endif  // GWTF_GEMM_H
