#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
// A [16][64] i8 row-major, B [16][64] i8 row-major (B holds columns n as rows): D[m][n] = sum_k A[m][k]*B[n][k]
__global__ void probe(const int8_t* A, const int8_t* B, int* D, int hyp) {
  const int l = threadIdx.x;
  int8_t a[16], b[16];
  for (int j = 0; j < 16; ++j) {
    int k = hyp == 0 ? (l >> 4) * 16 + j : ((j < 8) ? (l >> 4) * 8 + j : 32 + (l >> 4) * 8 + (j - 8));
    a[j] = A[(l & 15) * 64 + k];
    b[j] = B[(l & 15) * 64 + k];
  }
  v4i va, vb, acc = {0, 0, 0, 0};
  __builtin_memcpy(&va, a, 16); __builtin_memcpy(&vb, b, 16);
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(va, vb, acc, 0, 0, 0);
  // assumed C layout: col = l & 15, row = (l >> 4) * 4 + r
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
int main() {
  int8_t hA[16 * 64], hB[16 * 64]; int hD[256], ref[256];
  srand(1);
  for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { int s = 0; for (int k = 0; k < 64; ++k) s += (int)hA[m * 64 + k] * (int)hB[n * 64 + k]; ref[m * 16 + n] = s; }
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 2; ++hyp) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { if (hD[m * 16 + n] != ref[m * 16 + n]) ++bad; if (hD[n * 16 + m] != ref[m * 16 + n]) ++badT; }
    printf("hyp %d: mismatches (D[m][n]) %d, transposed %d\n", hyp, bad, badT);
  }
  return 0;
}
