// i8 MFMA GEMM prototype, second generation (harness): Y[m][n] = sum_k X[m][k] * W[n][k]  (int32 -> float)
//
//   gemm_lds<64>      the round-1 product tiling (register-staged loads, padded LDS rows) -- baseline
//   gemm_glds<ST>     128 x 128 block, operands straight from global memory into LDS (global_load_lds_dwordx4), unpadded
//                     128-byte rows with an XOR chunk swizzle applied on the SOURCE address; ST = 1: output tile staged
//                     through LDS so that every store instruction writes whole 128-byte lines
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int PITCH = 160;

template <int WT>
__global__ __launch_bounds__(256) void gemm_lds(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
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

#define GLDS16(gp, lp) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)

// 128 x 128 block tile, 2 x 2 waves of 64 x 64, K slab 128 bytes, 2 stages of [W 128 x 128 B | X 128 x 128 B] = 64 KB
template <int ST>
__global__ __launch_bounds__(256, 2) void gemm_glds(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                    uint32_t M, uint32_t N, uint32_t K) {
  constexpr int NI = 4, MI = 4;
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / 128;
  const uint32_t n0 = (blockIdx.x / tiles_m) * 128, m0 = (blockIdx.x % tiles_m) * 128;
  const int wn = (wave >> 1) * 64, wm = (wave & 1) * 64;
  const int r16 = lane & 15, kg = lane >> 4;
  // loader: wave w moves rows [32 w, 32 w + 32) of both operand tiles, 8 rows (1 KB of LDS) per instruction;
  // lane l -> row (l >> 3), LDS slot (l & 7), which holds source chunk slot ^ ((row >> 1) & 7)
  const int lrow = wave * 32 + (lane >> 3);
  const int8_t* wsrc[4];
  const int8_t* xsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = lrow + q * 8;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    wsrc[q] = W + (size_t)(n0 + row) * K + chunk * 16;
    xsrc[q] = X + (size_t)(m0 + row) * K + chunk * 16;
  }
  auto issue = [&](int stage, uint32_t k) {
    int8_t* bw = lds + stage * 32768 + wave * 32 * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      GLDS16(wsrc[q] + k, bw + q * 1024);
      GLDS16(xsrc[q] + k, bw + 16384 + q * 1024);
    }
  };
  // reader: fragment rows wn + 16 i + r16 (W), wm + 16 j + r16 (X); k chunk c = 4 s + kg sits in slot c ^ ((r16 >> 1) & 7)
  const int sw = (r16 >> 1) & 7;
  const int off0 = r16 * 128 + ((kg ^ sw) << 4), off1 = r16 * 128 + (((4 + kg) ^ sw) << 4);

  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  issue(0, 0);
  const uint32_t nk = K / 128;
  for (uint32_t kb = 0; kb < nk; ++kb) {
    __syncthreads();
    if (kb + 1 < nk) issue((kb + 1) & 1, (kb + 1) * 128);
    const int8_t* bw = lds + (kb & 1) * 32768 + wn * 128;
    const int8_t* bx = lds + (kb & 1) * 32768 + 16384 + wm * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int off = s ? off1 : off0;
      v4i fw[NI], fx[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + i * 2048 + off);
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + j * 2048 + off);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
  }
  if (ST == 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint32_t n = n0 + wn + i * 16 + kg * 4;
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const uint32_t m = m0 + wm + j * 16 + r16;
        *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
      }
    }
  } else {
    // staged: each wave parks 32 rows x 64 fp32 of its tile in its own 8.5 KB of LDS (row pitch 272 B), then stores
    // 4 rows x 256 B (= 8 whole lines) per instruction; two halves (j = 0, 1 | 2, 3)
    __syncthreads();                                  // everybody is done with the operand stages
    constexpr int SP = 272;
    int8_t* mine = lds + wave * (32 * SP);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * h + jj;
          *reinterpret_cast<f32x4*>(mine + (jj * 16 + r16) * SP + (i * 16 + kg * 4) * 4) =
              f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
        }
      // wave-private region: program order + lgkmcnt is all the synchronisation needed
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = t * 4 + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(mine + row * SP + (lane & 15) * 16);
        const uint32_t m = m0 + wm + h * 32 + row;
        f32x4* dst = reinterpret_cast<f32x4*>(Y + (size_t)m * N + n0 + wn + (lane & 15) * 4);
        if (ST == 1) *dst = v;
        else if (ST == 3) __builtin_nontemporal_store(v, dst);
        else if (v.x == 12345.5f) *dst = v;          // ST == 2: no stores (mainloop + staging only)
      }
    }
  }
}

// 4-stage ring of 64-byte K slabs (16 KB per stage: W 128 x 64 B | X 128 x 64 B), three slabs in flight, counted vmcnt
// waits (the compiler's __syncthreads would drain the queue), raw s_barrier.  NF = 1: tile index runs over n fastest
// (the X row tile is reused by consecutive blocks, W -- 2.4 MB -- stays L2 resident).
template <int ST, int NF>
__global__ __launch_bounds__(256, 2) void gemm_glds4(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                     uint32_t M, uint32_t N, uint32_t K) {
  constexpr int NI = 4, MI = 4, S = 4;
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / 128, tiles_n = N / 128;
  const uint32_t n0 = (NF ? blockIdx.x % tiles_n : blockIdx.x / tiles_m) * 128, m0 = (NF ? blockIdx.x / tiles_n : blockIdx.x % tiles_m) * 128;
  const int wn = (wave >> 1) * 64, wm = (wave & 1) * 64;
  const int r16 = lane & 15, kg = lane >> 4;
  const int8_t* wsrc[2];
  const int8_t* xsrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = wave * 32 + q * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
    wsrc[q] = W + (size_t)(n0 + row) * K + chunk * 16;
    xsrc[q] = X + (size_t)(m0 + row) * K + chunk * 16;
  }
  auto issue = [&](int stage, uint32_t k) {
    int8_t* bw = lds + stage * 16384 + wave * 32 * 64;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      GLDS16(wsrc[q] + k, bw + q * 1024);
      GLDS16(xsrc[q] + k, bw + 8192 + q * 1024);
    }
  };
  const int off = r16 * 64 + ((kg ^ ((0 - (r16 >> 2)) & 3)) << 4);

  v4i acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  const uint32_t nk = K / 64;           // >= 3
  issue(0, 0);
  issue(1, 64);
  issue(2, 128);
  for (uint32_t kb = 0; kb < nk; ++kb) {
    // slab kb landed?  younger slabs still in flight: min(2, nk - 1 - kb) x 4 loads
    const uint32_t left = nk - 1 - kb;
    if (left >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (left == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kb + 3 < nk) issue((kb + 3) & 3, (kb + 3) * 64);
    const int8_t* bw = lds + (kb & 3) * 16384 + wn * 64 + off;
    const int8_t* bx = lds + (kb & 3) * 16384 + 8192 + wm * 64 + off;
    v4i fw[NI], fx[MI];
#pragma unroll
    for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(bw + i * 1024);
#pragma unroll
    for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(bx + j * 1024);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
  }
  if (ST == 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint32_t n = n0 + wn + i * 16 + kg * 4;
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const uint32_t m = m0 + wm + j * 16 + r16;
        *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
      }
    }
  } else {
    __syncthreads();
    constexpr int SP = 272;
    int8_t* mine = lds + wave * (32 * SP);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * h + jj;
          *reinterpret_cast<f32x4*>(mine + (jj * 16 + r16) * SP + (i * 16 + kg * 4) * 4) =
              f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
        }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int row = t * 4 + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(mine + row * SP + (lane & 15) * 16);
        const uint32_t m = m0 + wm + h * 32 + row;
        *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n0 + wn + (lane & 15) * 4) = v;
      }
    }
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t shapes[][3] = {{1024, 768, 768}, {1024, 3072, 768}, {1024, 768, 3072}, {8192, 768, 768}, {8192, 3072, 768}, {8192, 768, 3072}, {8192, 8192, 8192}};
  constexpr int NB = 4;
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
      for (int t = 0; t < 3000; ++t) {
        uint32_t m = rand() % M, n = rand() % N; long sacc = 0;
        for (uint32_t k = 0; k < K; ++k) sacc += (long)hX[(size_t)m * K + k] * hW[(size_t)n * K + k];
        if ((float)sacc != hY[(size_t)m * N + n]) ++bad;
      }
      if (bad && !strstr(name, "nostore")) printf("   %s: %d / 3000 sampled outputs WRONG\n", name, bad);
    };
    auto run = [&](const char* name, auto launch) {
      CK(hipMemset(Y[0], 0, (size_t)M * N * 4));
      launch(0); CK(hipStreamSynchronize(st)); check(name);
      float best_c = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        for (int w = 0; w < 3; ++w) launch(0); CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventRecord(e0, st)); for (int r = 0; r < 24; ++r) launch(r % NB); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); best_c = ms / 24 < best_c ? ms / 24 : best_c;
      }
      printf("M=%u N=%u K=%u %-22s %8.2f us  %7.1f TOP/s\n", M, N, K, name, best_c * 1e3, 2.0 * M * N * K / (best_c * 1e-3) / 1e12);
    };
    run("lds 128x128 (r1)", [&](int b) { hipLaunchKernelGGL((gemm_lds<64>), dim3((M / 128) * (N / 128)), dim3(256), 2 * 2 * 128 * PITCH, st, X[b], W[b], Y[b], M, N, K); });
    run("glds 128x128", [&](int b) { hipLaunchKernelGGL((gemm_glds<0>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    run("glds 128x128 staged", [&](int b) { hipLaunchKernelGGL((gemm_glds<1>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    run("glds 128x128 nt-staged", [&](int b) { hipLaunchKernelGGL((gemm_glds<3>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    if (0) run("glds4 ring", [&](int b) { hipLaunchKernelGGL((gemm_glds4<0, 0>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    if (0) run("glds4 ring staged", [&](int b) { hipLaunchKernelGGL((gemm_glds4<1, 0>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    if (0) run("glds4 ring staged nfast", [&](int b) { hipLaunchKernelGGL((gemm_glds4<1, 1>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    run("glds 128x128 nostore", [&](int b) { hipLaunchKernelGGL((gemm_glds<2>), dim3((M / 128) * (N / 128)), dim3(256), 65536, st, X[b], W[b], Y[b], M, N, K); });
    for (int b = 0; b < NB; ++b) { CK(hipFree(X[b])); CK(hipFree(W[b])); CK(hipFree(Y[b])); }
    free(hX); free(hW); free(hY);
  }
  return 0;
}
