// Round-4 integer-GEMM main-loop harness: Y[m][n] = sum_k X[m][k] * W[n][k] (i8 x i8 -> i32) at M = 8192, N = 3072, K = 768
// (and the M = 1024 BERT shapes), generic in wave tile, waves per block, LDS ring depth -- to find what bounds the
// product kernel's main loop (csrc/tq_linear_i8.hip: 23 us where the matrix cores need 9.8 us).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tuning/i8_v4.hip -o tools/tuning/i8_v4 && tools/tuning/i8_v4
//
// Per variant: correctness of a sampled set of outputs against the CPU, then us per launch (HIP events, 30 launches).
// mode bits: 1 = no MFMA (loads + LDS reads only), 2 = no operand DMA, 4 = no LDS fragment reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), \
                                                        (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WM x WN: wave tile (tokens x features); NWM x NWN waves per block; ST: LDS ring depth (128-byte K slabs; REG loader: 2)
// LOADER 0: global_load_lds_dwordx4 (LDS-DMA)   1: global_load_dwordx4 -> registers -> ds_write_b128
// parts (compile time): DO_LOAD operand loads, DO_READ LDS fragment reads, DO_MFMA
template <int WM, int WN, int NWM, int NWN, int ST, int LOADER, bool DO_LOAD, bool DO_READ, bool DO_MFMA, int PAD>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm(const int8_t* __restrict__ X, const int8_t* __restrict__ W,
                                                        int* __restrict__ Y, uint32_t M, uint32_t N, uint32_t K, int store) {
  constexpr int NW = NWM * NWN, BM = WM * NWM, BN = WN * NWN, NI = WN / 16, MI = WM / 16;
  constexpr int ROWS = BN + BM, STB = ROWS * 128;          // per stage: [W rows | X rows] x 128 B
  constexpr int LPW = ROWS / 8 / NW;                        // 1 KB pieces per wave and slab
  static_assert(ROWS % (8 * NW) == 0, "rows must split evenly over the waves");
  static_assert(LOADER == 0 || ST == 2, "register-staged loader: double buffer");
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BM;
  const uint32_t n0 = (blockIdx.x / tiles_m) * BN, m0 = (blockIdx.x % tiles_m) * BM;
  const int wn = (wave / NWM) * WN, wm = (wave % NWM) * WM;
  const int r16 = lane & 15, kg = lane >> 4, swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};
  const uint32_t nk = K / 128;

  const int8_t* src[LPW];
  int dst[LPW];                                              // REG loader: this lane's swizzled byte offset inside a stage
#pragma unroll
  for (int q = 0; q < LPW; ++q) {
    const int row = (wave * LPW + q) * 8 + (lane >> 3);       // row of the stage: [0, BN) = W, [BN, ROWS) = X
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    src[q] = row < BN ? W + (size_t)(n0 + row) * K + chunk * 16 : X + (size_t)(m0 + row - BN) * K + chunk * 16;
    dst[q] = row * 128 + (lane & 7) * 16;
  }
  v4i stage_regs[LOADER == 1 ? LPW : 1];
  auto issue = [&](uint32_t kb) {
    if (LOADER == 0) {
      int8_t* b = lds + (kb % ST) * STB + wave * LPW * 1024;
#pragma unroll
      for (int q = 0; q < LPW; ++q) GLDS16(src[q] + kb * 128, b + q * 1024);
    } else {
#pragma unroll
      for (int q = 0; q < LPW; ++q) stage_regs[q] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src[q] + kb * 128));
    }
  };
  auto commit = [&](uint32_t kb) {                            // REG loader: registers -> LDS stage kb % 2
#pragma unroll
    for (int q = 0; q < LPW; ++q) *reinterpret_cast<v4i*>(lds + (kb % ST) * STB + dst[q]) = stage_regs[q];
  };

  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  if (DO_LOAD) {
    if (LOADER == 0) {
#pragma unroll
      for (int s = 0; s < ST - 1; ++s)
        if ((uint32_t)s < nk) issue(s);
    } else {
      issue(0);
      commit(0);
    }
  }
  for (uint32_t kb = 0; kb < nk; ++kb) {
    if (LOADER == 0) {
      // slab kb must have landed: at most the (ST - 2) younger slabs of this wave may still be in flight
      if (kb + ST - 1 <= nk) wait_vm<(ST - 2) * LPW>(); else wait_vm<0>();
      __syncthreads();
      if (DO_LOAD && kb + ST - 1 < nk) issue(kb + ST - 1);
    } else {
      __syncthreads();                                        // stage kb % 2 written by everybody; stage (kb + 1) % 2 free
      if (DO_LOAD && kb + 1 < nk) issue(kb + 1);
    }
    const int8_t* bw = lds + (kb % ST) * STB + wn * 128;
    const int8_t* bx = lds + (kb % ST) * STB + BN * 128 + wm * 128;
    v4i fw[2][NI], fx[2][MI];
    auto load_frags = [&](int s2) {
      if (!DO_READ) {
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" : "=v"(fw[s2][i]));
#pragma unroll
        for (int j = 0; j < MI; ++j) asm volatile("" : "=v"(fx[s2][j]));
        return;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[s2][i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s2]);
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[s2][j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off[s2]);
    };
    load_frags(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0) load_frags(1);
      __builtin_amdgcn_sched_barrier(0);
      if (!DO_MFMA) {
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(fw[s][i]));
#pragma unroll
        for (int j = 0; j < MI; ++j) asm volatile("" ::"v"(fx[s][j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[s][i], fx[s][j], acc[i][j], 0, 0, 0);
    }
    if (LOADER == 1 && DO_LOAD && kb + 1 < nk) commit(kb + 1);
  }
  if (store) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        *reinterpret_cast<v4i*>(Y + (size_t)(m0 + wm + j * 16 + r16) * N + n0 + wn + i * 16 + kg * 4) = acc[i][j];
  } else {
    int t = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) t ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
    if (t == 0x7ffffff1) Y[0] = t;
  }
}


struct Prob { uint32_t M, N, K; };

static std::vector<int8_t> hX, hW;
static int8_t *dX, *dW;
static int* dY;

// ---- overlap experiment: the product tiling (64x64 wave tiles, 2x2 waves, LDS-DMA, 2 stages) + a synthetic VALU
// epilogue of EPI dependent fma per output element (the product's GELU + quantizer epilogue is ~36 issue slots per
// element), as (a) one block per tile, (b) persistent blocks of TPB tiles, optionally with the SECOND block to arrive on
// a CU delayed by `delay_clk` shader clocks once, so that its epilogues meet the first block's main loops.

template <int EPI, int KS, int ST, int OCC, bool LATE_ISSUE>
__global__ __launch_bounds__(256, OCC) void gemm_epi(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                      uint32_t M, uint32_t N, uint32_t K) {
  // 64x64 wave tiles, 2x2 waves, LDS-DMA ring of ST slabs of KS (64 / 128) bytes of K; OCC blocks per CU wanted
  constexpr int WT = 64, BT = 128, NI = 4, MI = 4, OPB = BT * KS, STB = 2 * OPB, KSTEPS = KS / 64;
  constexpr int RPI = 1024 / KS;                 // rows per 1 KB DMA piece (8 / 16)
  constexpr int LPW = BT / RPI / 4;              // pieces per wave, operand and slab (4 / 2)
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT;
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;
  const int r16 = lane & 15, kg = lane >> 4;
  int off[KSTEPS];
  if (KS == 128) {
    const int swz = (r16 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) off[s] = r16 * 128 + (((4 * s + kg) ^ swz) << 4);
  } else {
    const int R = (r16 >> 2) & 3, g = R ^ ((R & 1) << 1);
    off[0] = r16 * 64 + ((kg ^ g) << 4);
  }
  const uint32_t nk = K / KS;
  const uint32_t tile = blockIdx.x;
  const uint32_t n0 = (tile / tiles_m) * BT, m0 = (tile % tiles_m) * BT;
  const int8_t *wsrc[LPW], *xsrc[LPW];
#pragma unroll
  for (int q = 0; q < LPW; ++q) {
    const int row = (wave * LPW + q) * RPI + lane / (KS / 16);
    int chunk;
    if (KS == 128) chunk = (lane & 7) ^ ((row >> 1) & 7);
    else { const int R = (row >> 2) & 3; chunk = (lane & 3) ^ (R ^ ((R & 1) << 1)); }
    wsrc[q] = W + (size_t)(n0 + row) * K + chunk * 16;
    xsrc[q] = X + (size_t)(m0 + row) * K + chunk * 16;
  }
  auto issue = [&](uint32_t kb) {
    int8_t* bw = lds + (kb % ST) * STB + wave * LPW * 1024;
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
      GLDS16(wsrc[q] + kb * KS, bw + q * 1024);
      GLDS16(xsrc[q] + kb * KS, bw + OPB + q * 1024);
    }
  };
  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if ((uint32_t)s < nk) issue(s);
  for (uint32_t kb = 0; kb < nk; ++kb) {
    if (kb + ST - 1 <= nk) wait_vm<(ST - 2) * 2 * LPW>(); else wait_vm<0>();
    __syncthreads();
    if (!LATE_ISSUE && kb + ST - 1 < nk) issue(kb + ST - 1);
    const int8_t* bw = lds + (kb % ST) * STB + wn * KS;
    const int8_t* bx = lds + (kb % ST) * STB + OPB + wm * KS;
    v4i fw[KSTEPS][NI], fx[KSTEPS][MI];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[s][i] = *reinterpret_cast<const v4i*>(bw + i * 16 * KS + off[s]);
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[s][j] = *reinterpret_cast<const v4i*>(bx + j * 16 * KS + off[s]);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[s][i], fx[s][j], acc[i][j], 0, 0, 0);
      if (LATE_ISSUE && s == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (kb + ST - 1 < nk) issue(kb + ST - 1);       // slab kb - 1 was released by this iteration's barrier
      }
    }
  }
  if (EPI < 0) {                                        // correctness run: raw accumulators
    int* Yi = reinterpret_cast<int*>(Y);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        *reinterpret_cast<v4i*>(Yi + (size_t)(m0 + wm + j * 16 + r16) * N + n0 + wn + i * 16 + kg * 4) = acc[i][j];
    return;
  }
  // synthetic epilogue: EPI dependent fma per element, 16 elements in flight
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float v[MI][4];
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j][r] = (float)acc[i][j][r];
#pragma unroll 2
    for (int e = 0; e < EPI; ++e)
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = __builtin_fmaf(v[j][r], 0.999f, 0.125f);
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += v[j][r];
  }
  if (sum == 123456.789f) Y[tile] = sum;
}

template <int EPI, int KS, int ST, int OCC, bool LATE>
static float epi_time(Prob p, int pad_kb) {
  const uint32_t grid = (p.M / 128) * (p.N / 128);
  auto k = gemm_epi<EPI, KS, ST, OCC, LATE>;
  const size_t lds = (size_t)ST * 2 * 128 * KS + pad_kb * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, dX, dW, reinterpret_cast<float*>(dY), p.M, p.N, p.K);
  CK(hipEventRecord(a));
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, dX, dW, reinterpret_cast<float*>(dY), p.M, p.N, p.K);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0f / 30;
}

template <int KS, int ST, int OCC, bool LATE>
static void run_epi(const char* name, Prob p, int pad_kb = 0) {
  // correctness
  auto k = gemm_epi<-1, KS, ST, OCC, LATE>;
  const size_t lds = (size_t)ST * 2 * 128 * KS + pad_kb * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipMemset(dY, 0xff, (size_t)p.M * p.N * 4));
  hipLaunchKernelGGL(k, dim3((p.M / 128) * (p.N / 128)), dim3(256), lds, 0, dX, dW, reinterpret_cast<float*>(dY), p.M, p.N, p.K);
  CK(hipDeviceSynchronize());
  std::vector<int> hY((size_t)p.M * p.N);
  CK(hipMemcpy(hY.data(), dY, hY.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < 3000; ++t) {
    const uint32_t m = (uint32_t)((t * 2654435761u) % p.M), n = (uint32_t)((t * 40503u + 17) % p.N);
    int ref = 0;
    for (uint32_t kk = 0; kk < p.K; ++kk) ref += (int)hX[(size_t)m * p.K + kk] * (int)hW[(size_t)n * p.K + kk];
    bad += ref != hY[(size_t)m * p.N + n];
  }
  const float t0 = epi_time<0, KS, ST, OCC, LATE>(p, pad_kb), t17 = epi_time<17, KS, ST, OCC, LATE>(p, pad_kb),
              t34 = epi_time<34, KS, ST, OCC, LATE>(p, pad_kb);
  printf("%-46s KS %3d ST %d lds %3zu KB %s : main loop %6.2f us | + 17 fma/elem %6.2f | + 34 fma/elem %6.2f\n", name, KS, ST, lds >> 10,
         bad ? "WRONG" : "ok", t0, t17, t34);
  fflush(stdout);
}

// ---- persistent blocks with the SECOND block to arrive on a CU delayed once by `delay_clk` shader clocks, so that its
// epilogues meet the first block's main loops (product tiling, synthetic epilogue of EPI fma per output)
__device__ __forceinline__ uint32_t cu_slot() {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
  return ((xcc & 15) * 8 + se) * 32 + sh * 16 + cu;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_persist(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                        uint32_t M, uint32_t N, uint32_t K, uint32_t tiles_per_block,
                                                        uint32_t delay_clk, uint32_t* __restrict__ cu_count) {
  constexpr int WT = 64, BT = 128, NI = 4, MI = 4, LPW = 4, OPB = BT * 128, STB = 2 * OPB;
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT, n_tiles = tiles_m * (N / BT);
  const int wn = (wave >> 1) * WT, wm = (wave & 1) * WT;
  const int r16 = lane & 15, kg = lane >> 4, swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};
  const uint32_t nk = K / 128;
  if (delay_clk) {
    __shared__ uint32_t arrival;
    if (tid == 0) arrival = atomicAdd(cu_count + cu_slot(), 1u);
    __syncthreads();
    if (arrival & 1) {
      const uint64_t t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < delay_clk) __builtin_amdgcn_s_sleep(8);
    }
  }
  const uint32_t first = blockIdx.x * tiles_per_block, last = min(first + tiles_per_block, n_tiles);
  for (uint32_t tile = first; tile < last; ++tile) {
    const uint32_t n0 = (tile / tiles_m) * BT, m0 = (tile % tiles_m) * BT;
    const int8_t *wsrc[LPW], *xsrc[LPW];
#pragma unroll
    for (int q = 0; q < LPW; ++q) {
      const int row = wave * (WT / 2) + q * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      wsrc[q] = W + (size_t)(n0 + row) * K + chunk * 16;
      xsrc[q] = X + (size_t)(m0 + row) * K + chunk * 16;
    }
    auto issue = [&](int stage, uint32_t k) {
      int8_t* bw = lds + stage * STB + wave * (WT / 2) * 128;
#pragma unroll
      for (int q = 0; q < LPW; ++q) {
        GLDS16(wsrc[q] + k, bw + q * 1024);
        GLDS16(xsrc[q] + k, bw + OPB + q * 1024);
      }
    };
    v4i acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    issue(0, 0);
    for (uint32_t kb = 0; kb < nk; ++kb) {
      wait_vm<0>();
      __syncthreads();
      if (kb + 1 < nk) issue((kb + 1) & 1, (kb + 1) * 128);
      const int8_t* bw = lds + (kb & 1) * STB + wn * 128;
      const int8_t* bx = lds + (kb & 1) * STB + OPB + wm * 128;
      v4i fw[2][NI], fx[2][MI];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) fw[s2][i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s2]);
#pragma unroll
        for (int j = 0; j < MI; ++j) fx[s2][j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off[s2]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[s][i], fx[s][j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v[MI][4];
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[j][r] = (float)acc[i][j][r];
#pragma unroll 2
      for (int e = 0; e < EPI; ++e)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j][r] = __builtin_fmaf(v[j][r], 0.999f, 0.125f);
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += v[j][r];
    }
    if (sum == 123456.789f) Y[tile] = sum;
  }
}

template <int EPI>
static void run_persist(const char* name, Prob p, uint32_t tpb, uint32_t delay_clk, uint32_t* d_count, float* dYf) {
  const uint32_t n_tiles = (p.M / 128) * (p.N / 128), grid = (n_tiles + tpb - 1) / tpb;
  auto k = gemm_persist<EPI>;
  const size_t lds = 2 * 2 * 128 * 128;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float total = 0.0f;
  const int reps = 20;
  for (int w = 0; w < reps + 3; ++w) {
    CK(hipMemsetAsync(d_count, 0, 4096 * 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, dX, dW, dYf, p.M, p.N, p.K, tpb, delay_clk, d_count);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (w >= 3) total += ms;
  }
  printf("%-50s EPI %3d tiles/block %u delay %6u clk grid %-5u: %6.2f us (single launches, one event pair each)\n", name, EPI, tpb,
         delay_clk, grid, total * 1000.0f / reps);
  fflush(stdout);
}

// ---- W-resident persistent variant: a block (8 waves: 4 x 2 of 32 x 64) owns ONE 128-feature panel of W for its whole run of
// M tiles -- the panel's K slabs stay in LDS (K * 128 B = 96 KB at K = 768) -- and streams X through a 3-slab ring that runs
// ahead across tile boundaries.  Operand traffic per tile: X only (98 KB instead of 196 KB).
template <int EPI, int KMAX>
__global__ __launch_bounds__(512, 1) void gemm_wres(const int8_t* __restrict__ X, const int8_t* __restrict__ W, int* __restrict__ Y,
                                                     uint32_t M, uint32_t N, uint32_t K, uint32_t tiles_per_block, int store) {
  constexpr int WM = 32, WN = 64, NI = WN / 16, MI = WM / 16, ST = 3, XSL = 128 * 128;     // X slab bytes
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];                             // [W: K/128 slabs x 128 rows x 128 B][X ring]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nk = K / 128, tiles_m = M / 128, n_tiles = tiles_m * (N / 128);
  int8_t* wl = lds;
  int8_t* xl = lds + (size_t)nk * XSL;
  const int wn = (wave >> 2) * WN, wm = (wave & 3) * WM;
  const int r16 = lane & 15, kg = lane >> 4, swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};
  const uint32_t first = blockIdx.x * tiles_per_block, last = min(first + tiles_per_block, n_tiles);
  if (first >= last) return;
  // this wave's 2 pieces (16 rows) of every 128-row slab
  const int rowA = wave * 16 + (lane >> 3), rowB = rowA + 8;
  const int chA = (lane & 7) ^ ((rowA >> 1) & 7), chB = (lane & 7) ^ ((rowB >> 1) & 7);
  uint32_t panel = 0xffffffffu;
  const uint32_t total_slabs = (last - first) * nk;
  auto x_issue = [&](uint32_t s) {             // flat slab index of this block's run
    const uint32_t tile = first + s / nk, kb = s % nk;
    const uint32_t m0 = (tile % tiles_m) * 128;
    int8_t* b = xl + (s % ST) * XSL + wave * 2048;
    GLDS16(X + (size_t)(m0 + rowA) * K + kb * 128 + chA * 16, b);
    GLDS16(X + (size_t)(m0 + rowB) * K + kb * 128 + chB * 16, b + 1024);
  };
  x_issue(0);
  if (total_slabs > 1) x_issue(1);
  uint32_t s = 0;
  for (uint32_t tile = first; tile < last; ++tile) {
    const uint32_t n0 = (tile / tiles_m) * 128, m0 = (tile % tiles_m) * 128;
    if (tile / tiles_m != panel) {             // (re)load the W panel: nk slabs x 2 pieces per wave
      panel = tile / tiles_m;
      __syncthreads();                          // nobody reads the old panel any more
      for (uint32_t kb = 0; kb < nk; ++kb) {
        int8_t* b = wl + (size_t)kb * XSL + wave * 2048;
        GLDS16(W + (size_t)(n0 + rowA) * K + kb * 128 + chA * 16, b);
        GLDS16(W + (size_t)(n0 + rowB) * K + kb * 128 + chB * 16, b + 1024);
      }
    }
    v4i acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    for (uint32_t kb = 0; kb < nk; ++kb, ++s) {
      // slab s (and, first time round, the W panel) must have landed; the younger X slab may still be in flight
      if (s + 1 < total_slabs && !(tile == first && kb == 0) && kb != 0) wait_vm<2>(); else wait_vm<0>();
      __syncthreads();
      if (s + 2 < total_slabs) x_issue(s + 2);
      const int8_t* bw = wl + (size_t)kb * XSL + wn * 128;
      const int8_t* bx = xl + (s % ST) * XSL + wm * 128;
      v4i fw[2][NI], fx[2][MI];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) fw[s2][i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s2]);
#pragma unroll
        for (int j = 0; j < MI; ++j) fx[s2][j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off[s2]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[s2][i], fx[s2][j], acc[i][j], 0, 0, 0);
      }
    }
    if (store) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          *reinterpret_cast<v4i*>(Y + (size_t)(m0 + wm + j * 16 + r16) * N + n0 + wn + i * 16 + kg * 4) = acc[i][j];
    } else {
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float v[MI][4];
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j][r] = (float)acc[i][j][r];
#pragma unroll 2
        for (int e = 0; e < EPI; ++e)
#pragma unroll
          for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = __builtin_fmaf(v[j][r], 0.999f, 0.125f);
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) sum += v[j][r];
      }
      if (sum == 123456.789f) Y[tile] = (int)sum;
    }
  }
}

template <int EPI>
static float wres_time(Prob p, uint32_t tpb, bool check, int* bad_out) {
  const uint32_t n_tiles = (p.M / 128) * (p.N / 128), grid = (n_tiles + tpb - 1) / tpb;
  auto k = gemm_wres<EPI, 768>;
  const size_t lds = (size_t)(p.K / 128) * 128 * 128 + 3 * 128 * 128;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (check) {
    CK(hipMemset(dY, 0xff, (size_t)p.M * p.N * 4));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, dX, dW, dY, p.M, p.N, p.K, tpb, 1);
    CK(hipDeviceSynchronize());
    std::vector<int> hY((size_t)p.M * p.N);
    CK(hipMemcpy(hY.data(), dY, hY.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 3000; ++t) {
      const uint32_t m = (uint32_t)((t * 2654435761u) % p.M), n = (uint32_t)((t * 40503u + 17) % p.N);
      int ref = 0;
      for (uint32_t kk = 0; kk < p.K; ++kk) ref += (int)hX[(size_t)m * p.K + kk] * (int)hW[(size_t)n * p.K + kk];
      bad += ref != hY[(size_t)m * p.N + n];
    }
    *bad_out = bad;
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, dX, dW, dY, p.M, p.N, p.K, tpb, 0);
  CK(hipEventRecord(a));
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, dX, dW, dY, p.M, p.N, p.K, tpb, 0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0f / 30;
}

static void run_wres(Prob p) {
  for (uint32_t tpb : {6u, 8u, 4u, 3u}) {
    int bad = 0;
    const float t0 = wres_time<0>(p, tpb, true, &bad), t17 = wres_time<17>(p, tpb, false, &bad), t34 = wres_time<34>(p, tpb, false, &bad);
    printf("W-resident persistent, 8 waves, %u tiles per block (grid %u): %s  main loop %6.2f us | + 17 fma/elem %6.2f | + 34 fma/elem %6.2f\n",
           tpb, ((p.M / 128) * (p.N / 128) + tpb - 1) / tpb, bad ? "WRONG" : "ok", t0, t17, t34);
    fflush(stdout);
  }
}

// ---- do MFMA and VALU work of DIFFERENT waves on one SIMD overlap?  Block = 512 threads (2 waves per SIMD on one CU when
// one block per CU): role bit per wave: 1 = MFMA chain (16 independent accumulators), 2 = VALU fma chains.  `who` selects which
// waves work: 1 = the first four waves run MFMAs, the rest idle; 2 = the last four run VALU; 3 = both at once.
__global__ __launch_bounds__(512) void overlap_probe(float* out, int who, int iters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(who & 1)) return;
    v4i acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = v4i{0, 0, 0, 0};
    v4i a = v4i{lane, 1, 2, 3}, b = v4i{3, 2, 1, lane};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t ^= acc[i][0] ^ acc[i][3];
    if (t == 0x7ffffff1) out[0] = 1.0f;
  } else {
    if (!(who & 2)) return;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 7; ++r)                      // 16 x 7 fma per iteration ~ 270 cycles ~ the 16 MFMAs' 256
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.125f);
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += v[i];
    if (sum == 123456.789f) out[1] = sum;
  }
}

static void run_overlap_probe(float* dYf) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 2000;
  for (int who = 1; who <= 3; ++who) {
    hipLaunchKernelGGL(overlap_probe, dim3(256), dim3(512), 0, 0, dYf, who, iters);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(overlap_probe, dim3(256), dim3(512), 0, 0, dYf, who, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("overlap probe, 256 blocks x 8 waves, %d iterations, %s: %8.1f us\n", iters,
           who == 1 ? "MFMA waves only (16 x 16x16x64 i8 per iteration)" : who == 2 ? "VALU waves only (112 v_fma_f32 per iteration)" : "both kinds together", ms * 1000.0f);
  }
  fflush(stdout);
}

template <int WM, int WN, int NWM, int NWN, int ST, int LOADER, bool DL, bool DR, bool DM, int PAD>
static float time_one(Prob p, bool check, int* bad_out) {
  constexpr int BM = WM * NWM, BN = WN * NWN;
  const size_t lds = (size_t)ST * (BM + BN) * 128 + PAD * 1024;      // PAD KB of unused LDS: caps the blocks per CU
  auto k = gemm<WM, WN, NWM, NWN, ST, LOADER, DL, DR, DM, PAD>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((p.M / BM) * (p.N / BN)), block(64 * NWM * NWN);
  if (check) {
    CK(hipMemset(dY, 0xff, (size_t)p.M * p.N * 4));
    hipLaunchKernelGGL(k, grid, block, lds, 0, dX, dW, dY, p.M, p.N, p.K, 1);
    CK(hipDeviceSynchronize());
    std::vector<int> hY((size_t)p.M * p.N);
    CK(hipMemcpy(hY.data(), dY, hY.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 3000; ++t) {
      const uint32_t m = (uint32_t)((t * 2654435761u) % p.M), n = (uint32_t)((t * 40503u + 17) % p.N);
      int ref = 0;
      for (uint32_t kk = 0; kk < p.K; ++kk) ref += (int)hX[(size_t)m * p.K + kk] * (int)hW[(size_t)n * p.K + kk];
      bad += ref != hY[(size_t)m * p.N + n];
    }
    *bad_out = bad;
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, grid, block, lds, 0, dX, dW, dY, p.M, p.N, p.K, 0);
  CK(hipEventRecord(a));
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(k, grid, block, lds, 0, dX, dW, dY, p.M, p.N, p.K, 0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms * 1000.0f / 30;
}

template <int WM, int WN, int NWM, int NWN, int ST, int LOADER, int PAD = 0, bool PROBE = false>
static void run(const char* name, Prob p) {
  constexpr int BM = WM * NWM, BN = WN * NWN;
  if (p.M % BM || p.N % BN) { printf("%-40s M=%u N=%u K=%u: shape not a tile multiple\n", name, p.M, p.N, p.K); return; }
  int bad = 0;
  const float full = time_one<WM, WN, NWM, NWN, ST, LOADER, true, true, true, PAD>(p, true, &bad);
  const size_t lds = (size_t)ST * (BM + BN) * 128 + PAD * 1024;
  printf("%-40s M=%-5u N=%-5u K=%-5u grid %-5u lds %3zu KB %s full %6.2f us (%.2f POP/s)", name, p.M, p.N, p.K,
         (p.M / BM) * (p.N / BN), lds >> 10, bad ? "WRONG" : "ok", full, 2.0 * p.M * p.N * p.K / (full * 1e-6) / 1e15);
  if (PROBE) {
    const float sk = time_one<WM, WN, NWM, NWN, ST, LOADER, false, false, false, PAD>(p, false, &bad);
    const float ld = time_one<WM, WN, NWM, NWN, ST, LOADER, true, false, false, PAD>(p, false, &bad);
    const float rd = time_one<WM, WN, NWM, NWN, ST, LOADER, false, true, false, PAD>(p, false, &bad);
    const float mf = time_one<WM, WN, NWM, NWN, ST, LOADER, false, false, true, PAD>(p, false, &bad);
    const float ldrd = time_one<WM, WN, NWM, NWN, ST, LOADER, true, true, false, PAD>(p, false, &bad);
    const float rdmf = time_one<WM, WN, NWM, NWN, ST, LOADER, false, true, true, PAD>(p, false, &bad);
    const float ldmf = time_one<WM, WN, NWM, NWN, ST, LOADER, true, false, true, PAD>(p, false, &bad);
    printf(" | barriers %5.2f | +load %5.2f | +read %5.2f | +mfma %5.2f | load+read %5.2f | read+mfma %5.2f | load+mfma %5.2f", sk, ld, rd,
           mf, ldrd, rdmf, ldmf);
  }
  printf("\n");
  fflush(stdout);
}


// ---- warp specialisation: NL extra LOADER waves per block issue all the LDS-DMA of a slab; the NWM x NWN compute waves only
// read fragments and run MFMAs (a compute wave that issues its own 8 pieces per slab sits ~150 cycles in each: in-order issue).
template <int WM, int WN, int NWM, int NWN, int NL, int ST>
__global__ __launch_bounds__(64 * (NWM * NWN + NL)) void gemm_spec(const int8_t* __restrict__ X, const int8_t* __restrict__ W,
                                                                   int* __restrict__ Y, uint32_t M, uint32_t N, uint32_t K, int store) {
  constexpr int NW = NWM * NWN, BM = WM * NWM, BN = WN * NWN, NI = WN / 16, MI = WM / 16;
  constexpr int ROWS = BN + BM, STB = ROWS * 128;
  constexpr int LPL = ROWS / 8 / NL;                        // 1 KB pieces per loader wave and slab
  static_assert(ROWS % (8 * NL) == 0, "rows must split evenly over the loader waves");
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BM;
  const uint32_t n0 = (blockIdx.x / tiles_m) * BN, m0 = (blockIdx.x % tiles_m) * BM;
  const uint32_t nk = K / 128;
  if (wave >= NW) {                                          // ---- loader wave
    const int lw = wave - NW;
    const int8_t* src[LPL];
#pragma unroll
    for (int q = 0; q < LPL; ++q) {
      const int row = (lw * LPL + q) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      src[q] = row < BN ? W + (size_t)(n0 + row) * K + chunk * 16 : X + (size_t)(m0 + row - BN) * K + chunk * 16;
    }
    auto issue = [&](uint32_t kb) {
      int8_t* b = lds + (kb % ST) * STB + lw * LPL * 1024;
#pragma unroll
      for (int q = 0; q < LPL; ++q) GLDS16(src[q] + kb * 128, b + q * 1024);
    };
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
      if ((uint32_t)s < nk) issue(s);
    for (uint32_t kb = 0; kb < nk; ++kb) {
      if (kb + ST - 1 <= nk) wait_vm<(ST - 2) * LPL>(); else wait_vm<0>();
      __syncthreads();                                       // slab kb published; slab kb - 1 no longer read
      if (kb + ST - 1 < nk) issue(kb + ST - 1);
    }
    return;
  }
  const int wn = (wave / NWM) * WN, wm = (wave % NWM) * WM;
  const int r16 = lane & 15, kg = lane >> 4, swz = (r16 >> 1) & 7;
  const int off[2] = {r16 * 128 + ((kg ^ swz) << 4), r16 * 128 + (((4 + kg) ^ swz) << 4)};
  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
  for (uint32_t kb = 0; kb < nk; ++kb) {
    __syncthreads();
    const int8_t* bw = lds + (kb % ST) * STB + wn * 128;
    const int8_t* bx = lds + (kb % ST) * STB + BN * 128 + wm * 128;
    v4i fw[2][NI], fx[2][MI];
    auto load_frags = [&](int s2) {
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[s2][i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off[s2]);
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[s2][j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off[s2]);
    };
    load_frags(0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0) load_frags(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[s][i], fx[s][j], acc[i][j], 0, 0, 0);
    }
  }
  if (store) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        *reinterpret_cast<v4i*>(Y + (size_t)(m0 + wm + j * 16 + r16) * N + n0 + wn + i * 16 + kg * 4) = acc[i][j];
  } else {
    int t = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) t ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
    if (t == 0x7ffffff1) Y[0] = t;
  }
}

template <int WM, int WN, int NWM, int NWN, int NL, int ST>
static void run_spec(const char* name, Prob p, int pad_kb = 0) {
  constexpr int BM = WM * NWM, BN = WN * NWN;
  const size_t lds = (size_t)ST * (BM + BN) * 128 + pad_kb * 1024;
  auto k = gemm_spec<WM, WN, NWM, NWN, NL, ST>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid((p.M / BM) * (p.N / BN)), block(64 * (NWM * NWN + NL));
  CK(hipMemset(dY, 0xff, (size_t)p.M * p.N * 4));
  hipLaunchKernelGGL(k, grid, block, lds, 0, dX, dW, dY, p.M, p.N, p.K, 1);
  CK(hipDeviceSynchronize());
  std::vector<int> hY((size_t)p.M * p.N);
  CK(hipMemcpy(hY.data(), dY, hY.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < 3000; ++t) {
    const uint32_t m = (uint32_t)((t * 2654435761u) % p.M), n = (uint32_t)((t * 40503u + 17) % p.N);
    int ref = 0;
    for (uint32_t kk = 0; kk < p.K; ++kk) ref += (int)hX[(size_t)m * p.K + kk] * (int)hW[(size_t)n * p.K + kk];
    bad += ref != hY[(size_t)m * p.N + n];
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, grid, block, lds, 0, dX, dW, dY, p.M, p.N, p.K, 0);
  CK(hipEventRecord(a));
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(k, grid, block, lds, 0, dX, dW, dY, p.M, p.N, p.K, 0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-62s M=%-5u N=%-5u K=%-5u lds %3zu KB %s  %6.2f us\n", name, p.M, p.N, p.K, lds >> 10, bad ? "WRONG" : "ok", ms * 1000.0f / 30);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const Prob all[] = {{8192, 3072, 768}, {1024, 3072, 768}, {1024, 768, 768}, {1024, 768, 3072}};
  const int nprob = argc > 1 ? atoi(argv[1]) : 4;
  size_t maxX = 0, maxW = 0, maxY = 0;
  for (auto p : all) {
    maxX = std::max(maxX, (size_t)p.M * p.K); maxW = std::max(maxW, (size_t)p.N * p.K); maxY = std::max(maxY, (size_t)p.M * p.N);
  }
  hX.resize(maxX); hW.resize(maxW);
  srand(7);
  for (auto& v : hX) v = (int8_t)(rand() % 255 - 127);
  for (auto& v : hW) v = (int8_t)(rand() % 15 - 7);
  CK(hipMalloc(&dX, maxX)); CK(hipMalloc(&dW, maxW)); CK(hipMalloc(&dY, maxY * 4));
  CK(hipMemcpy(dX, hX.data(), maxX, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), maxW, hipMemcpyHostToDevice));
  if (getenv("I8_RING")) {      // deeper LDS rings for the small-tile kernel the M = 1024 shapes run
    for (int pi = 1; pi < 4; ++pi) {
      const Prob p = all[pi];
      run<32, 32, 2, 2, 2, 0, 0, false>("DMA 32x32 2x2 2 stages (product at M = 1024)", p);
      run<32, 32, 2, 2, 3, 0, 0, false>("DMA 32x32 2x2 3 stages", p);
      run<32, 32, 2, 2, 4, 0, 0, false>("DMA 32x32 2x2 4 stages", p);
      run<32, 32, 2, 2, 6, 0, 0, false>("DMA 32x32 2x2 6 stages", p);
      run<16, 32, 2, 2, 4, 0, 0, false>("DMA 16x32 2x2 (32x64 tiles) 4 stages", p);
      run<32, 16, 2, 2, 4, 0, 0, false>("DMA 32x16 2x2 (64x32 tiles) 4 stages", p);
      run<16, 32, 2, 2, 2, 0, 0, false>("DMA 16x32 2x2 (32x64 tiles) 2 stages", p);
    }
    return 0;
  }
  if (getenv("I8_SPEC")) {
    for (int pi = 0; pi < nprob; ++pi) {
      const Prob p = all[pi];
      run<64, 64, 2, 2, 2, 0, 0, false>("baseline: every wave loads (64x64 2x2, 2 st, 2 blk/CU)", p);
      run_spec<64, 64, 2, 2, 1, 2>("4 compute + 1 loader wave, 2 st", p);
      run_spec<64, 64, 2, 2, 2, 2>("4 compute + 2 loader waves, 2 st", p);
      run_spec<64, 64, 2, 2, 4, 2>("4 compute + 4 loader waves, 2 st", p);
      run_spec<64, 64, 2, 2, 2, 2>("4 compute + 2 loader waves, 2 st, 1 blk/CU", p, 32);
      run_spec<32, 64, 4, 2, 2, 2>("8 compute (32x64) + 2 loader waves, 2 st", p);
      run_spec<32, 64, 4, 2, 4, 2>("8 compute (32x64) + 4 loader waves, 2 st", p);
      run_spec<64, 64, 2, 2, 2, 3>("4 compute + 2 loader waves, 3 st (1 blk/CU)", p);
      run_spec<32, 32, 2, 2, 1, 2>("64x64 tile: 4 compute (32x32) + 1 loader, 2 st", p);
      run_spec<32, 32, 2, 2, 2, 2>("64x64 tile: 4 compute (32x32) + 2 loaders, 2 st", p);
      run<32, 32, 2, 2, 2, 0, 0, false>("64x64 tile baseline: every wave loads (32x32 2x2)", p);
    }
    return 0;
  }
  {
    float* dYf;
    CK(hipMalloc(&dYf, 1 << 20));
    const Prob p = all[0];
    run_overlap_probe(dYf);
    run_wres(p);
    run_epi<128, 2, 2, false>("product tiling (2 blocks/CU)", p);
    run_epi<128, 2, 2, true>("product tiling, DMA issued after k-step 0", p);
    run_epi<64, 2, 2, false>("64-byte slabs, 2 st (32 KB: up to 4 blocks/CU?)", p);
    run_epi<64, 3, 2, false>("64-byte slabs, 3 st (48 KB: 3 blocks/CU)", p);
    run_epi<64, 4, 2, false>("64-byte slabs, 4 st (64 KB: 2 blocks/CU)", p);
    run_epi<64, 3, 3, false>("64-byte slabs, 3 st, launch_bounds 3", p);
    run_epi<64, 2, 4, false>("64-byte slabs, 2 st, launch_bounds 4", p);
    run_epi<64, 3, 3, true>("64-byte slabs, 3 st, lb 3, late issue", p);
    run_epi<128, 2, 2, false>("product tiling, 1 block/CU (pad)", p, 32);
    uint32_t* d_count;
    CK(hipMalloc(&d_count, 4096 * 4));
    run_persist<34>("persistent, 3 tiles per block (2 blocks/CU)", p, 3, 0, d_count, dYf);
    for (uint32_t d : {2000u, 4000u, 8000u, 12000u, 100000u})
      run_persist<34>("persistent 3, 2nd block of every CU delayed once", p, 3, d, d_count, dYf);
    run_persist<34>("persistent, 6 tiles per block (1 block/CU)", p, 6, 0, d_count, dYf);
  }
  if (argc > 2) return 0;
  for (int pi = 0; pi < nprob; ++pi) {
    const Prob p = all[pi];
    // (X / W of a smaller problem are the leading bytes of the big buffers, read with the problem's own K pitch)
    run<64, 64, 2, 2, 2, 0, 0, true>("DMA 64x64 2x2 2st (2 blk/CU) = product", p);
    run<32, 64, 4, 2, 2, 0, 0, true>("DMA 32x64 4x2 8 waves, 128x128 blk (2 blk/CU)", p);
    run<64, 32, 2, 4, 2, 0, 0, true>("DMA 64x32 2x4 8 waves, 128x128 blk (2 blk/CU)", p);
    run<32, 32, 4, 4, 2, 0, 0, false>("DMA 32x32 4x4 16 waves, 128x128 blk (2 blk/CU)", p);
    run<64, 64, 2, 2, 2, 0, 32, true>("DMA 64x64 2x2 2st, 1 blk/CU", p);
    run<64, 64, 2, 2, 2, 1, 0, true>("REG 64x64 2x2 (2 blk/CU)", p);
    run<64, 64, 2, 2, 2, 1, 32, true>("REG 64x64 2x2, 1 blk/CU", p);
    run<128, 64, 2, 2, 2, 0, 0, true>("DMA 128x64 2x2 2st (1 blk/CU)", p);
    run<128, 64, 2, 2, 2, 1, 0, true>("REG 128x64 2x2 (1 blk/CU)", p);
    run<128, 128, 2, 2, 2, 0, 0, true>("DMA 128x128 2x2 2st (1 blk/CU)", p);
    run<128, 128, 2, 2, 2, 1, 0, true>("REG 128x128 2x2 (1 blk/CU)", p);
    run<64, 64, 4, 2, 2, 1, 0, true>("REG 64x64 4x2 8 waves (1 blk/CU)", p);
    run<32, 32, 2, 2, 2, 0, 0, true>("DMA 32x32 2x2 2st", p);
    run<32, 32, 2, 2, 2, 1, 0, true>("REG 32x32 2x2", p);
    run<32, 64, 2, 2, 2, 1, 0, false>("REG 32x64 2x2", p);
    run<64, 32, 2, 2, 2, 1, 0, false>("REG 64x32 2x2", p);
  }
  return 0;
}
