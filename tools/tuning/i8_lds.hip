// LDS-staged i8 MFMA GEMM prototype (harness): Y[m][n] = sum_k X[m][k] * W[n][k]  (int32 -> float)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int PITCH = 144;   // 128 B of K + 16 B pad: conflict-free ds_read_b128 over 16 rows

// WT = wave tile edge (32 or 64); block = 2 x 2 waves -> block tile BT = 2 * WT; BK = 128 bytes
template <int WT>
__global__ __launch_bounds__(256) void gemm_lds(const int8_t* __restrict__ X, const int8_t* __restrict__ W, float* __restrict__ Y,
                                                uint32_t M, uint32_t N, uint32_t K) {
  constexpr int BT = 2 * WT, NI = WT / 16, MI = WT / 16;
  constexpr int LPT = BT * 128 / 16 / 256;        // 16-byte loads per thread per operand per stage (BT=64: 2, BT=128: 4)
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];   // [2 stages][2 operands][BT][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_m = M / BT;
  const uint32_t n0 = (blockIdx.x / tiles_m) * BT, m0 = (blockIdx.x % tiles_m) * BT;
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

int main() {
  hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t shapes[][3] = {{1024, 768, 768}, {1024, 3072, 768}, {1024, 768, 3072}, {8192, 3072, 768}, {128, 128, 128}};
  for (auto& sh : shapes) {
    const uint32_t M = sh[0], N = sh[1], K = sh[2];
    int8_t *X, *W; float* Y;
    int8_t* hX = (int8_t*)malloc((size_t)M * K); int8_t* hW = (int8_t*)malloc((size_t)N * K);
    srand(M + N + K);
    for (size_t i = 0; i < (size_t)M * K; ++i) hX[i] = (int8_t)(rand() % 255 - 127);
    for (size_t i = 0; i < (size_t)N * K; ++i) hW[i] = (int8_t)(rand() % 255 - 127);
    CK(hipMalloc(&X, (size_t)M * K)); CK(hipMalloc(&W, (size_t)N * K)); CK(hipMalloc(&Y, (size_t)M * N * 4));
    CK(hipMemcpy(X, hX, (size_t)M * K, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW, (size_t)N * K, hipMemcpyHostToDevice));
    float* hY = (float*)malloc((size_t)M * N * 4);
    auto check = [&](const char* name) {
      CK(hipMemcpy(hY, Y, (size_t)M * N * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int t = 0; t < 2000; ++t) {
        uint32_t m = rand() % M, n = rand() % N; long s = 0;
        for (uint32_t k = 0; k < K; ++k) s += (long)hX[(size_t)m * K + k] * hW[(size_t)n * K + k];
        if ((float)s != hY[(size_t)m * N + n]) ++bad;
      }
      printf("   %s: %d / 2000 sampled outputs wrong\n", name, bad);
    };
    auto run = [&](const char* name, auto launch) {
      CK(hipMemset(Y, 0, (size_t)M * N * 4));
      launch(); CK(hipStreamSynchronize(st)); check(name);
      for (int w = 0; w < 5; ++w) launch(); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st)); for (int r = 0; r < 50; ++r) launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("M=%u N=%u K=%u %-20s %7.2f us  %6.0f TOPS\n", M, N, K, name, ms / 50 * 1e3, 2.0 * M * N * K / (ms / 50 * 1e-3) / 1e12);
    };
    if (M % 64 == 0 && N % 64 == 0) run("lds wt32 (64x64)", [&] { hipLaunchKernelGGL((gemm_lds<32>), dim3((M / 64) * (N / 64)), dim3(256), 2 * 2 * 64 * PITCH, st, X, W, Y, M, N, K); });
    if (M % 128 == 0 && N % 128 == 0) run("lds wt64 (128x128)", [&] { hipLaunchKernelGGL((gemm_lds<64>), dim3((M / 128) * (N / 128)), dim3(256), 2 * 2 * 128 * PITCH, st, X, W, Y, M, N, K); });
    CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y)); free(hX); free(hW); free(hY);
  }
  return 0;
}
