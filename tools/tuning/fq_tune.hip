// Standalone tuning harness for the K1 kernel family (not part of libtq_hip.so).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../transformer-quantization_amd/csrc/tq_device.h"
using namespace tq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, bool NTL, bool NTS, bool MATH, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_stride(const u32x4* __restrict__ x, u32x4* __restrict__ y, uint64_t n_vec, QP p) {
  const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
  uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
  for (; i + (U - 1) * stride < n_vec; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? ld_stream(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32x4 o = v[u];
      if (MATH) {
        float f[8];
        Store<TQ_BF16>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = q_dequant(q_index(f[j], p), p);
        o = Store<TQ_BF16>::pack(f);
      }
      if (NTS) st_stream(y + i + u * stride, o); else y[i + u * stride] = o;
    }
  }
  for (; i < n_vec; i += stride) {
    u32x4 o = x[i];
    if (MATH) { float f[8]; Store<TQ_BF16>::unpack(o, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = q_dequant(q_index(f[j], p), p);
      o = Store<TQ_BF16>::pack(f); }
    y[i] = o;
  }
}

// contiguous chunk per block: block b owns vectors [b*chunk, (b+1)*chunk)
template <int U, bool NTL, bool NTS, bool MATH, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_chunk(const u32x4* __restrict__ x, u32x4* __restrict__ y, uint64_t n_vec, QP p) {
  const uint64_t chunk = (n_vec + gridDim.x - 1) / gridDim.x;
  const uint64_t beg = (uint64_t)blockIdx.x * chunk;
  const uint64_t end = min(beg + chunk, n_vec);
  uint64_t i = beg + threadIdx.x;
  for (; i + (U - 1) * BLOCK < end; i += U * BLOCK) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? ld_stream(x + i + u * BLOCK) : x[i + u * BLOCK];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32x4 o = v[u];
      if (MATH) {
        float f[8];
        Store<TQ_BF16>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = q_dequant(q_index(f[j], p), p);
        o = Store<TQ_BF16>::pack(f);
      }
      if (NTS) st_stream(y + i + u * BLOCK, o); else y[i + u * BLOCK] = o;
    }
  }
  for (; i < end; i += BLOCK) {
    u32x4 o = x[i];
    if (MATH) { float f[8]; Store<TQ_BF16>::unpack(o, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = q_dequant(q_index(f[j], p), p);
      o = Store<TQ_BF16>::pack(f); }
    y[i] = o;
  }
}

struct Variant { const char* name; void (*launch)(const u32x4*, u32x4*, uint64_t, QP, unsigned, hipStream_t); int block; };

template <int U, bool NTL, bool NTS, bool MATH, int BLOCK, bool CHUNK>
void launch(const u32x4* x, u32x4* y, uint64_t n_vec, QP p, unsigned grid, hipStream_t st) {
  if (CHUNK) hipLaunchKernelGGL((k_chunk<U, NTL, NTS, MATH, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, x, y, n_vec, p);
  else hipLaunchKernelGGL((k_stride<U, NTL, NTS, MATH, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, x, y, n_vec, p);
}

int main(int argc, char** argv) {
  const uint64_t n = (uint64_t)1024 * 512 * 768;
  const uint64_t n_vec = n / 8;
  u32x4 *x, *y;
  CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2));
  std::vector<uint16_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) { float f = (float)((int)(rand() % 2001) - 1000) / 300.0f; uint32_t b; memcpy(&b, &f, 4); h[i] = b >> 16; }
  for (uint64_t off = 0; off < n * 2; off += h.size() * 2) CK(hipMemcpy((char*)x + off, h.data(), std::min<uint64_t>(h.size() * 2, n * 2 - off), hipMemcpyHostToDevice));
  QP p = {0.03f, 128.0f, 0.0f, 255.0f};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<Variant> vs = {
#define V(U, NTL, NTS, MATH, BLOCK, CHUNK) {#U "," #NTL "," #NTS "," #MATH "," #BLOCK "," #CHUNK, launch<U, NTL, NTS, MATH, BLOCK, CHUNK>, BLOCK}
    V(1, true, true, true, 256, true), V(2, true, true, true, 256, true), V(4, true, true, true, 256, true),
    V(1, true, true, true, 512, true), V(2, true, true, true, 512, true), V(4, true, true, true, 512, true),
    V(1, true, true, true, 1024, true), V(2, true, true, true, 1024, true), V(4, true, true, true, 1024, true),
    V(2, false, false, true, 512, true), V(2, true, false, true, 512, true), V(2, false, true, true, 512, true),
    V(2, true, true, false, 512, true), V(1, true, true, false, 1024, true),
    V(2, true, true, true, 512, false), V(1, true, true, true, 512, false), V(1, true, true, true, 1024, false),
  };
  const unsigned grids[] = {16384, 24576, 32768, 49152, 65536, 98304, 196608};
  printf("%-28s", "U,NTL,NTS,MATH,BLOCK,CHUNK");
  for (unsigned g : grids) printf(" g=%-7u", g);
  printf("   (GB/s, 4 B/elem)\n");
  for (auto& v : vs) {
    printf("%-28s", v.name);
    for (unsigned g : grids) {
      for (int w = 0; w < 3; ++w) v.launch(x, y, n_vec, p, g, st);
      CK(hipStreamSynchronize(st));
      const int reps = 10;
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) v.launch(x, y, n_vec, p, g, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %-9.0f", n * 4.0 / (ms / reps * 1e-3) / 1e9);
    }
    printf("\n");
  }
  return 0;
}
