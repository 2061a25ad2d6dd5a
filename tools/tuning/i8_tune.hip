// tuning harness for the i8 MFMA GEMM main loop (not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: full; 1: no global stores (one dummy store); 2: no loads in loop (reuse first fragments)
template <int TN, int TM, int MODE, int WPB>
__global__ __launch_bounds__(WPB * 64) void k64(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                uint32_t M, uint32_t N, uint32_t K) {
  constexpr int NI = TN / 16, MI = TM / 16;
  const int lane = threadIdx.x & 63;
  const uint32_t tiles_m = M / TM;
  const uint32_t tile = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (tile >= tiles_m * (N / TN)) return;
  const uint32_t n0 = (tile / tiles_m) * TN, m0 = (tile % tiles_m) * TM;
  const int r16 = lane & 15, kg = lane >> 4;
  const int8_t* wp = W + (size_t)(n0 + r16) * K + kg * 16;
  const int8_t* xp = X + (size_t)(m0 + r16) * K + kg * 16;
  v4i acc[NI][MI];
  for (int i = 0; i < NI; ++i) for (int j = 0; j < MI; ++j) acc[i][j] = v4i{0, 0, 0, 0};
  v4i fw[NI], fx[MI];
#pragma unroll
  for (int i = 0; i < NI; ++i) fw[i] = *reinterpret_cast<const v4i*>(wp + (size_t)i * 16 * K);
#pragma unroll
  for (int j = 0; j < MI; ++j) fx[j] = *reinterpret_cast<const v4i*>(xp + (size_t)j * 16 * K);
  for (uint32_t k = 64; k <= K; k += 64) {
    v4i nw[NI], nx[MI];
    const bool more = k < K;
    if (more && MODE != 2) {
#pragma unroll
      for (int i = 0; i < NI; ++i) nw[i] = *reinterpret_cast<const v4i*>(wp + (size_t)i * 16 * K + k);
#pragma unroll
      for (int j = 0; j < MI; ++j) nx[j] = *reinterpret_cast<const v4i*>(xp + (size_t)j * 16 * K + k);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[i], fx[j], acc[i][j], 0, 0, 0);
    if (more && MODE != 2) {
#pragma unroll
      for (int i = 0; i < NI; ++i) fw[i] = nw[i];
#pragma unroll
      for (int j = 0; j < MI; ++j) fx[j] = nx[j];
    }
  }
  if (MODE == 1) {
    int s = 0;
    for (int i = 0; i < NI; ++i) for (int j = 0; j < MI; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 0x7fffffff) Y[0] = 1.f;
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t n = n0 + i * 16 + kg * 4;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const uint32_t m = m0 + j * 16 + r16;
      *reinterpret_cast<f32x4*>(Y + (size_t)m * N + n) = f32x4{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
    }
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t shapes[][3] = {{1024, 768, 768}, {1024, 3072, 768}, {8192, 3072, 768}};
  for (auto& sh : shapes) {
    const uint32_t M = sh[0], N = sh[1], K = sh[2];
    int8_t *X, *W; float* Y;
    CK(hipMalloc(&X, (size_t)M * K)); CK(hipMalloc(&W, (size_t)N * K)); CK(hipMalloc(&Y, (size_t)M * N * 4));
    CK(hipMemset(X, 1, (size_t)M * K)); CK(hipMemset(W, 2, (size_t)N * K));
    auto run = [&](const char* name, auto launch) {
      for (int w = 0; w < 5; ++w) launch(); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st)); for (int r = 0; r < 50; ++r) launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("M=%u N=%u K=%u %-34s %7.2f us  %6.0f TOPS\n", M, N, K, name, ms / 50 * 1e3, 2.0 * M * N * K / (ms / 50 * 1e-3) / 1e12);
    };
#define RUN(TN, TM, MODE, WPB) run(#TN "x" #TM " mode" #MODE " wpb" #WPB, [&] { hipLaunchKernelGGL((k64<TN, TM, MODE, WPB>), dim3(((M / TM) * (N / TN) + WPB - 1) / WPB), dim3(WPB * 64), 0, st, X, W, Y, M, N, K); })
    RUN(32, 32, 0, 4); RUN(32, 32, 1, 4); RUN(32, 32, 2, 4);
    RUN(64, 64, 0, 4); RUN(64, 64, 1, 4); RUN(64, 64, 2, 4);
    RUN(64, 64, 0, 1); RUN(64, 64, 0, 2); RUN(32, 32, 0, 1); RUN(32, 32, 0, 8);
    RUN(64, 32, 0, 4); RUN(32, 64, 0, 4);
    CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y));
  }
  return 0;
}
