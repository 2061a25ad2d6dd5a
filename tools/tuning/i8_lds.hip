// LDS-staged i8 MFMA GEMM prototype (harness): Y[m][n] = sum_k X[m][k] * W[n][k]  (int32 -> float)
//
// Variants kept here because their (negative) result shaped the product kernel (csrc/tq_linear_i8.hip):
//   gemm_lds<WT>            the product tiling; optional XCD-aware tile mapping (gm x gn)
//   gemm_lds_pf<WT, D>      register prefetch ring, plain C++ (the compiler sinks the loads: no effect)
//   gemm_lds_asm<WT, D>     the same ring with inline-asm loads + explicit partial vmcnt (really D deep)
//   gemm_lds_bk<WT, KS>     wider K slab per barrier
// Measured (M = 1024 tokens; 768->768 / 768->3072 / 3072->768): 4.7 / 9.4 / 10.5 us for ALL of them, warm or
// with 12 rotating operand sets (L2-cold): neither load latency, nor barrier count, nor cross-XCD operand
// replication bounds these GEMMs -- the LDS does: a 32x32 wave tile reads one 1 KB fragment per MFMA, i.e.
// 48 KB of LDS traffic per 128-byte K slab and workgroup against 128 B/clk/CU.  Only a larger wave tile
// changes that (128x128 blocks: 38.7 vs 48 us at M = 8192) and there are too few tiles for that at M = 1024.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int PITCH = 160;   // 128 B of K + 32 B pad: conflict-free ds_read_b128 under the real 4 x 16 lane grouping (144 is 2-way)

// WT = wave tile edge (32 or 64); block = 2 x 2 waves -> block tile BT = 2 * WT; BK = 128 bytes
// gm x gn = 8: XCD x (= blockIdx % 8, the hardware's round-robin) owns the tile rectangle (x / gn, x % gn) of a
// gm x gn partition of the tile grid, so its L2 only has to hold X / gm and W / gn.  gm == 0: plain mapping.
template <int WT>
__global__ __launch_bounds__(256) void gemm_lds(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                uint32_t M, uint32_t N, uint32_t K, uint32_t gm = 0, uint32_t gn = 0) {
  constexpr int BT = 2 * WT, NI = WT / 16, MI = WT / 16;
  constexpr int LPT = BT * 128 / 16 / 256;        // 16-byte loads per thread per operand per stage (BT=64: 2, BT=128: 4)
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];   // [2 stages][2 operands][BT][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT;
  uint32_t tn = blockIdx.x / tiles_m, tm = blockIdx.x % tiles_m;
  if (gm != 0) {
    const uint32_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const uint32_t pm = tiles_m / gm, pn = (N / BT) / gn;          // tiles per partition edge
    tm = (xcd / gn) * pm + local % pm;
    tn = (xcd % gn) * pn + local / pm;
  }
  const uint32_t n0 = tn * BT, m0 = tm * BT;
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;          // wave's sub-tile inside the block tile
  const int r16 = lane & 15, kg = lane >> 4;

  // global -> register staging: thread t covers row (t >> 3) + 32 * r, bytes (t & 7) * 16 of the 128-byte K slab
  const int grow = tid >> 3, gcol = (tid & 7) * 16;
  const int8_t* wsrc = W + (size_t)(n0 + grow) * K + gcol;
  const int8_t* xsrc = X + (size_t)(m0 + grow) * K + gcol;
  v4i rw[LPT], rx[LPT];
  auto gload = [&](uint32_t k) {
#pragma unroll
    for (int r = 0; r < LPT; ++r) {
      rw[r] = *reinterpret_cast<const v4i*>(wsrc + (size_t)r * 32 * K + k);
      rx[r] = *reinterpret_cast<const v4i*>(xsrc + (size_t)r * 32 * K + k);
    }
  };
  auto lstore = [&](int stage) {
    int8_t* bw = lds + (size_t)stage * 2 * BT * PITCH;
    int8_t* bx = bw + (size_t)BT * PITCH;
#pragma unroll
    for (int r = 0; r < LPT; ++r) {
      *reinterpret_cast<v4i*>(bw + (grow + r * 32) * PITCH + gcol) = rw[r];
      *reinterpret_cast<v4i*>(bx + (grow + r * 32) * PITCH + gcol) = rx[r];
    }
  };

  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  gload(0);
  lstore(0);
  __syncthreads();
  const uint32_t nk = K / 128;
  for (uint32_t kb = 0; kb < nk; ++kb) {
    const bool more = kb + 1 < nk;
    if (more) gload((kb + 1) * 128);
    const int8_t* bw = lds + (size_t)(kb & 1) * 2 * BT * PITCH;
    const int8_t* bx = bw + (size_t)BT * PITCH;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      v4i fw[NI], fx[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + (wn + i * 16 + r16) * PITCH + s * 64 + kg * 16);
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + (wm + j * 16 + r16) * PITCH + s * 64 + kg * 16);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore((kb + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + wn + i * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const uint32_t m = m0 + wm + j * 16 + r16;
      *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
    }
  }
}

// Same tiling, but the global loads run D K-slabs ahead of the MFMAs (register ring, statically unrolled):
// with one slab in flight the K loop costs one global-load latency per slab.
template <int WT, int D>
__global__ __launch_bounds__(256) void gemm_lds_pf(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                   uint32_t M, uint32_t N, uint32_t K) {
  constexpr int BT = 2 * WT, NI = WT / 16, MI = WT / 16;
  constexpr int LPT = BT * 128 / 16 / 256;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];   // [2 stages][2 operands][BT][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT;
  const uint32_t n0 = (blockIdx.x / tiles_m) * BT, m0 = (blockIdx.x % tiles_m) * BT;
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;
  const int r16 = lane & 15, kg = lane >> 4;
  const int grow = tid >> 3, gcol = (tid & 7) * 16;
  const int8_t* wsrc = W + (size_t)(n0 + grow) * K + gcol;
  const int8_t* xsrc = X + (size_t)(m0 + grow) * K + gcol;
  v4i rw[D][LPT], rx[D][LPT];
  const uint32_t nk = K / 128;
  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
#define GLOAD(S, KB)                                                                         \
  _Pragma("unroll") for (int r = 0; r < LPT; ++r) {                                          \
    rw[S][r] = *reinterpret_cast<const v4i*>(wsrc + (size_t)r * 32 * K + (size_t)(KB) * 128); \
    rx[S][r] = *reinterpret_cast<const v4i*>(xsrc + (size_t)r * 32 * K + (size_t)(KB) * 128); \
  }
#define LSTORE(S, STAGE)                                                                     \
  {                                                                                          \
    int8_t* bw_ = lds + (size_t)(STAGE) * 2 * BT * PITCH;                                    \
    int8_t* bx_ = bw_ + (size_t)BT * PITCH;                                                  \
    _Pragma("unroll") for (int r = 0; r < LPT; ++r) {                                        \
      *reinterpret_cast<v4i*>(bw_ + (grow + r * 32) * PITCH + gcol) = rw[S][r];              \
      *reinterpret_cast<v4i*>(bx_ + (grow + r * 32) * PITCH + gcol) = rx[S][r];              \
    }                                                                                        \
  }
  // prologue: slabs 0 .. D-1 in flight
#pragma unroll
  for (int s = 0; s < D; ++s) { const uint32_t k0 = (uint32_t)s < nk ? s : nk - 1; GLOAD(s, k0) }
  LSTORE(0, 0)
  __syncthreads();
  for (uint32_t kb0 = 0; kb0 < nk; kb0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const uint32_t kb = kb0 + s;
      if (kb < nk) {
        // slab kb is in LDS stage kb & 1; register slot s is free now: refill it with slab kb + D
        { const uint32_t nxt = kb + D < nk ? kb + D : nk - 1; GLOAD(s, nxt) }   // unconditional: keeps vmcnt static
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ABOVE the MFMA section (else the scheduler sinks them and reuses their registers)
        const int8_t* bw = lds + (size_t)(kb & 1) * 2 * BT * PITCH;
        const int8_t* bx = bw + (size_t)BT * PITCH;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v4i fw[NI], fx[MI];
#pragma unroll
          for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + (wn + i * 16 + r16) * PITCH + h * 64 + kg * 16);
#pragma unroll
          for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + (wm + j * 16 + r16) * PITCH + h * 64 + kg * 16);
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
        if (kb + 1 < nk) {
          // slab kb + 1 sits in register slot (s + 1) % D
          if (s + 1 < D) { LSTORE(s + 1, (kb + 1) & 1) } else { LSTORE(0, (kb + 1) & 1) }
        }
        __syncthreads();
      }
    }
  }
#undef GLOAD
#undef LSTORE
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + wn + i * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const uint32_t m = m0 + wm + j * 16 + r16;
      *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
    }
  }
}

// Prefetch ring with the global loads issued through inline asm: the compiler can neither sink them below
// the MFMA section nor recycle their destination registers, and the waits are explicit partial vmcnt's.
__device__ __forceinline__ void gload16(v4i& dst, const int8_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin(v4i& v) { asm volatile("" : "+v"(v)); }

template <int WT, int D>
__global__ __launch_bounds__(256) void gemm_lds_asm(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                    uint32_t M, uint32_t N, uint32_t K) {
  constexpr int BT = 2 * WT, NI = WT / 16, MI = WT / 16;
  constexpr int LPT = BT * 128 / 16 / 256;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT;
  const uint32_t n0 = (blockIdx.x / tiles_m) * BT, m0 = (blockIdx.x % tiles_m) * BT;
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;
  const int r16 = lane & 15, kg = lane >> 4;
  const int grow = tid >> 3, gcol = (tid & 7) * 16;
  const int8_t* wsrc = W + (size_t)(n0 + grow) * K + gcol;
  const int8_t* xsrc = X + (size_t)(m0 + grow) * K + gcol;
  v4i rw[D][LPT], rx[D][LPT];
  const uint32_t nk = K / 128;
  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
#define GLOADA(S, KB)                                                                         \
  _Pragma("unroll") for (int r = 0; r < LPT; ++r) {                                           \
    gload16(rw[S][r], wsrc + (size_t)r * 32 * K + (size_t)(KB) * 128);                        \
    gload16(rx[S][r], xsrc + (size_t)r * 32 * K + (size_t)(KB) * 128);                        \
  }
#define LSTOREA(S, STAGE)                                                                     \
  {                                                                                           \
    int8_t* bw_ = lds + (size_t)(STAGE) * 2 * BT * PITCH;                                     \
    int8_t* bx_ = bw_ + (size_t)BT * PITCH;                                                   \
    _Pragma("unroll") for (int r = 0; r < LPT; ++r) { pin(rw[S][r]); pin(rx[S][r]); }         \
    _Pragma("unroll") for (int r = 0; r < LPT; ++r) {                                         \
      *reinterpret_cast<v4i*>(bw_ + (grow + r * 32) * PITCH + gcol) = rw[S][r];               \
      *reinterpret_cast<v4i*>(bx_ + (grow + r * 32) * PITCH + gcol) = rx[S][r];               \
    }                                                                                         \
  }
#pragma unroll
  for (int s = 0; s < D; ++s) { const uint32_t k0 = (uint32_t)s < nk ? s : nk - 1; GLOADA(s, k0) }
  wait_vm<2 * LPT * (D - 1)>();
  LSTOREA(0, 0)
  __syncthreads();
  for (uint32_t kb0 = 0; kb0 < nk; kb0 += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const uint32_t kb = kb0 + s;
      if (kb < nk) {
        { const uint32_t nxt = kb + D < nk ? kb + D : nk - 1; GLOADA(s, nxt) }
        const int8_t* bw = lds + (size_t)(kb & 1) * 2 * BT * PITCH;
        const int8_t* bx = bw + (size_t)BT * PITCH;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          v4i fw[NI], fx[MI];
#pragma unroll
          for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + (wn + i * 16 + r16) * PITCH + h * 64 + kg * 16);
#pragma unroll
          for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + (wm + j * 16 + r16) * PITCH + h * 64 + kg * 16);
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
        }
        if (kb + 1 < nk) {
          // slab kb + 1 (register slot (s + 1) % D) was requested D - 1 slabs ago; D - 1 newer slabs stay in flight
          wait_vm<2 * LPT * (D - 1)>();
          if (s + 1 < D) { LSTOREA(s + 1, (kb + 1) & 1) } else { LSTOREA(0, (kb + 1) & 1) }
        }
        __syncthreads();
      }
    }
  }
  wait_vm<0>();
  // the (redundant, clamped) tail loads are still landing in the ring registers: keep them reserved until here
#pragma unroll
  for (int s = 0; s < D; ++s)
#pragma unroll
    for (int r = 0; r < LPT; ++r) { pin(rw[s][r]); pin(rx[s][r]); }
#undef GLOADA
#undef LSTOREA
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + wn + i * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const uint32_t m = m0 + wm + j * 16 + r16;
      *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
    }
  }
}

// Wider K slab per barrier (BK = 128 * KS bytes): the per-slab cost is a fixed latency chain
// (LDS write -> barrier -> LDS read -> dependent MFMAs -> barrier), so fewer, fatter slabs amortise it.
template <int WT, int KS>
__global__ __launch_bounds__(256) void gemm_lds_bk(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                   uint32_t M, uint32_t N, uint32_t K) {
  constexpr int BT = 2 * WT, NI = WT / 16, MI = WT / 16, BK = 128 * KS, PITCHK = BK + 16;
  constexpr int LPT = BT * BK / 16 / 256;
  constexpr int CPR = BK / 16;                    // 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];   // [2 stages][2 operands][BT][PITCHK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT;
  const uint32_t n0 = (blockIdx.x / tiles_m) * BT, m0 = (blockIdx.x % tiles_m) * BT;
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;
  const int r16 = lane & 15, kg = lane >> 4;
  const int grow = tid / CPR, gcol = (tid % CPR) * 16;
  constexpr int RSTEP = 256 / CPR;                // rows covered per load round
  const int8_t* wsrc = W + (size_t)(n0 + grow) * K + gcol;
  const int8_t* xsrc = X + (size_t)(m0 + grow) * K + gcol;
  v4i rw[LPT], rx[LPT];
  auto gload = [&](uint32_t k) {
#pragma unroll
    for (int r = 0; r < LPT; ++r) {
      rw[r] = *reinterpret_cast<const v4i*>(wsrc + (size_t)r * RSTEP * K + k);
      rx[r] = *reinterpret_cast<const v4i*>(xsrc + (size_t)r * RSTEP * K + k);
    }
  };
  auto lstore = [&](int stage) {
    int8_t* bw = lds + (size_t)stage * 2 * BT * PITCHK;
    int8_t* bx = bw + (size_t)BT * PITCHK;
#pragma unroll
    for (int r = 0; r < LPT; ++r) {
      *reinterpret_cast<v4i*>(bw + (grow + r * RSTEP) * PITCHK + gcol) = rw[r];
      *reinterpret_cast<v4i*>(bx + (grow + r * RSTEP) * PITCHK + gcol) = rx[r];
    }
  };
  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
  gload(0);
  lstore(0);
  __syncthreads();
  const uint32_t nk = K / BK;
  for (uint32_t kb = 0; kb < nk; ++kb) {
    const bool more = kb + 1 < nk;
    if (more) gload((kb + 1) * BK);
    const int8_t* bw = lds + (size_t)(kb & 1) * 2 * BT * PITCHK;
    const int8_t* bx = bw + (size_t)BT * PITCHK;
#pragma unroll
    for (int h = 0; h < 2 * KS; ++h) {
      v4i fw[NI], fx[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + (wn + i * 16 + r16) * PITCHK + h * 64 + kg * 16);
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + (wm + j * 16 + r16) * PITCHK + h * 64 + kg * 16);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore((kb + 1) & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + wn + i * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const uint32_t m = m0 + wm + j * 16 + r16;
      *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
    }
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t shapes[][3] = {{1024, 768, 768}, {1024, 3072, 768}, {1024, 768, 3072}, {1024, 2304, 768}, {8192, 3072, 768}};
  constexpr int NB = 12;     // rotating operand sets: every launch sees L2-cold (MALL-warm) operands, like a model forward
  for (auto& sh : shapes) {
    const uint32_t M = sh[0], N = sh[1], K = sh[2];
    int8_t *X[NB], *W[NB]; float* Y[NB];
    int8_t* hX = (int8_t*)malloc((size_t)M * K); int8_t* hW = (int8_t*)malloc((size_t)N * K);
    srand(M + N + K);
    for (size_t i = 0; i < (size_t)M * K; ++i) hX[i] = (int8_t)(rand() % 255 - 127);
    for (size_t i = 0; i < (size_t)N * K; ++i) hW[i] = (int8_t)(rand() % 255 - 127);
    for (int b = 0; b < NB; ++b) {
      CK(hipMalloc(&X[b], (size_t)M * K)); CK(hipMalloc(&W[b], (size_t)N * K)); CK(hipMalloc(&Y[b], (size_t)M * N * 4));
      CK(hipMemcpy(X[b], hX, (size_t)M * K, hipMemcpyHostToDevice)); CK(hipMemcpy(W[b], hW, (size_t)N * K, hipMemcpyHostToDevice));
    }
    float* hY = (float*)malloc((size_t)M * N * 4);
    auto check = [&](const char* name) {
      CK(hipMemcpy(hY, Y[0], (size_t)M * N * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int t = 0; t < 2000; ++t) {
        uint32_t m = rand() % M, n = rand() % N; long sacc = 0;
        for (uint32_t k = 0; k < K; ++k) sacc += (long)hX[(size_t)m * K + k] * hW[(size_t)n * K + k];
        if ((float)sacc != hY[(size_t)m * N + n]) ++bad;
      }
      if (bad) printf("   %s: %d / 2000 sampled outputs WRONG\n", name, bad);
    };
    auto run = [&](const char* name, auto launch) {
      CK(hipMemset(Y[0], 0, (size_t)M * N * 4));
      launch(0); CK(hipStreamSynchronize(st)); check(name);
      float best_w = 1e9f, best_c = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        for (int w = 0; w < 5; ++w) launch(0); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); for (int r = 0; r < 48; ++r) launch(0); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_w = ms / 48 < best_w ? ms / 48 : best_w;
        CK(hipEventRecord(e0, st)); for (int r = 0; r < 48; ++r) launch(r % NB); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best_c = ms / 48 < best_c ? ms / 48 : best_c;
      }
      printf("M=%u N=%u K=%u %-26s warm %7.2f us   rotating %7.2f us\n", M, N, K, name, best_w * 1e3, best_c * 1e3);
    };
#define RUN_LDS(WT, NAME) run(NAME, [&](int b) { hipLaunchKernelGGL((gemm_lds<WT>), dim3((M / (2 * WT)) * (N / (2 * WT))), dim3(256), 2 * 2 * (2 * WT) * PITCH, st, X[b], W[b], Y[b], M, N, K); })
#define RUN_PF(WT, D, NAME) run(NAME, [&](int b) { hipLaunchKernelGGL((gemm_lds_pf<WT, D>), dim3((M / (2 * WT)) * (N / (2 * WT))), dim3(256), 2 * 2 * (2 * WT) * PITCH, st, X[b], W[b], Y[b], M, N, K); })
    RUN_LDS(32, "lds 64x64");
    RUN_LDS(16, "lds 32x32");
#define RUN_XCD(WT, GM, GN, NAME) if ((M / (2 * WT)) % GM == 0 && (N / (2 * WT)) % GN == 0) run(NAME, [&](int b) { hipLaunchKernelGGL((gemm_lds<WT>), dim3((M / (2 * WT)) * (N / (2 * WT))), dim3(256), 2 * 2 * (2 * WT) * PITCH, st, X[b], W[b], Y[b], M, N, K, GM, GN); })
    RUN_XCD(32, 8, 1, "lds 64x64 xcd 8x1");
    RUN_XCD(32, 4, 2, "lds 64x64 xcd 4x2");
    RUN_XCD(32, 2, 4, "lds 64x64 xcd 2x4");
    RUN_XCD(32, 1, 8, "lds 64x64 xcd 1x8");
    RUN_XCD(16, 4, 2, "lds 32x32 xcd 4x2");
    RUN_XCD(16, 2, 4, "lds 32x32 xcd 2x4");
#define RUN_ASM(WT, D, NAME) run(NAME, [&](int b) { hipLaunchKernelGGL((gemm_lds_asm<WT, D>), dim3((M / (2 * WT)) * (N / (2 * WT))), dim3(256), 2 * 2 * (2 * WT) * PITCH, st, X[b], W[b], Y[b], M, N, K); })
#define RUN_BK(WT, KS, NAME) if (K % (128 * KS) == 0) run(NAME, [&](int b) { hipLaunchKernelGGL((gemm_lds_bk<WT, KS>), dim3((M / (2 * WT)) * (N / (2 * WT))), dim3(256), 2 * 2 * (2 * WT) * (128 * KS + 16), st, X[b], W[b], Y[b], M, N, K); })
    RUN_BK(32, 2, "lds 64x64 BK256");
    RUN_ASM(32, 2, "asm 64x64 ring 2");
    RUN_PF(32, 2, "lds 64x64 prefetch 2");
    RUN_PF(32, 3, "lds 64x64 prefetch 3");
    RUN_PF(32, 4, "lds 64x64 prefetch 4");
    if (M % 128 == 0 && N % 128 == 0) { RUN_LDS(64, "lds 128x128"); RUN_PF(64, 2, "lds 128x128 prefetch 2"); }
    for (int b = 0; b < NB; ++b) { CK(hipFree(X[b])); CK(hipFree(W[b])); CK(hipFree(Y[b])); }
    free(hX); free(hW); free(hY);
  }
  return 0;
}
