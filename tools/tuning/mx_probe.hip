// would 4-bit operands on the block-scaled matrix path (v_mfma_scale_f32_16x16x128_f8f6f4, FP6 E2M3
// holding the half-integer-centred 4-bit indices, unit E8M0 scales) break the i8 MFMA + VALU floor of the integer Linear?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tuning/mx_probe.hip -o tools/tuning/mx_probe
// Three questions, one binary:
//   (0) layout + exactness: random half-integers k - 7.5 (A) and w + 0.5 (B) in FP6 E2M3, K = 128 .. 512, against the exact
//       integer contraction on the host (products are multiples of 0.25, every partial sum < 2^24 quarter-units => fp32
//       accumulation is exact in any order);
//   (1) cross-wave: waves 0-3 issue the scaled MFMA (fp6 / fp8 / fp4 operand formats), waves 4-7 v_fma_f32: alone, alone, together
//       (the i8 opcode measured additive: profiles/r04/mfma_valu_overlap.txt);
//   (2) intra-wave: F scalar fmas after every MFMA.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// operand format codes of the f8f6f4 instruction (cbsz for A, blgp for B): 0 fp8 e4m3, 1 bf8 e5m2, 2 fp6 e2m3, 3 bf6 e3m2, 4 fp4 e2m1
#define FMT_FP8 0
#define FMT_FP6 2
#define FMT_FP4 4

// ---- (0) exactness ---------------------------------------------------------------------------------------------------------
// FP6 E2M3 (bias 1): s eem mmm.  |v| = m/8 (e = 0), (1 + m/8) 2^(e-1) otherwise.  Half-integers h/2, h odd, |h| <= 15:
static uint8_t fp6_of_half_units(int h) {          // value = h / 2
  uint8_t s = h < 0 ? 0x20 : 0;
  int a = h < 0 ? -h : h;                          // |value| = a / 2
  // a/2 in units of 1/8 = 4a
  int u = 4 * a;                                   // eighths
  uint8_t e, m;
  if (u < 8) { e = 0; m = (uint8_t)u; }
  else if (u < 16) { e = 1; m = (uint8_t)(u - 8); }
  else if (u < 32) { e = 2; if (u & 1) { printf("not representable\n"); exit(1); } m = (uint8_t)((u - 16) / 2); }
  else { e = 3; if (u & 3) { printf("not representable\n"); exit(1); } m = (uint8_t)((u - 32) / 4); }
  return s | (e << 3) | m;
}

// One wave computes a 16 x 16 tile over K = 128 * nk.  A [16][K], B [16][K] (both K-contiguous) as packed FP6: per (row, 32-k
// block) 24 bytes.  Assumed register layout: lane l holds row l & 15, k block l >> 4 (32 consecutive k), element j at bits 6j.
__global__ void exact_k(const uint8_t* A6, const uint8_t* B6, float* C, int nk) {
  const int lane = threadIdx.x & 63, r = lane & 15, kb = lane >> 4;
  v4f acc = v4f{0, 0, 0, 0};
  for (int t = 0; t < nk; ++t) {
    const uint32_t* pa = (const uint32_t*)(A6 + ((size_t)r * nk * 4 + t * 4 + kb) * 24);
    const uint32_t* pb = (const uint32_t*)(B6 + ((size_t)r * nk * 4 + t * 4 + kb) * 24);
    v8i a = v8i{(int)pa[0], (int)pa[1], (int)pa[2], (int)pa[3], (int)pa[4], (int)pa[5], 0, 0};
    v8i b = v8i{(int)pb[0], (int)pb[1], (int)pb[2], (int)pb[3], (int)pb[4], (int)pb[5], 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FMT_FP6, FMT_FP6, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  // C/D: col = lane & 15, row = (lane >> 4) * 4 + i
#pragma unroll
  for (int i = 0; i < 4; ++i) C[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc[i];
}

static void exactness() {
  for (int nk = 1; nk <= 4; ++nk) {
    const int K = 128 * nk;
    std::vector<int> ha(16 * K), hb(16 * K);
    std::vector<uint8_t> A6(16 * nk * 4 * 24, 0), B6(16 * nk * 4 * 24, 0);
    uint32_t s = 12345u + nk;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 10); };
    auto pack = [&](std::vector<uint8_t>& dst, const std::vector<int>& h) {
      for (int r = 0; r < 16; ++r)
        for (int k = 0; k < K; ++k) {
          const uint8_t c = fp6_of_half_units(h[r * K + k]);
          uint8_t* blk = dst.data() + ((size_t)r * nk * 4 + k / 32) * 24;
          const int bit = (k % 32) * 6;
          for (int q = 0; q < 6; ++q)
            if (c & (1 << q)) blk[(bit + q) >> 3] |= (uint8_t)(1 << ((bit + q) & 7));
        }
    };
    for (auto& v : ha) v = 2 * (int)(rnd() % 16) - 15;          // (k - 7.5) * 2, k in 0..15
    for (auto& v : hb) v = 2 * ((int)(rnd() % 16) - 8) + 1;     // (w + 0.5) * 2, w in -8..7
    if (nk == 4) { for (int k = 0; k < K; ++k) { ha[k] = 15; hb[k] = 15; } }   // worst case row 0 x row 0: K * 56.25
    pack(A6, ha); pack(B6, hb);
    uint8_t *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A6.size())); CK(hipMalloc(&dB, B6.size())); CK(hipMalloc(&dC, 256 * 4));
    CK(hipMemcpy(dA, A6.data(), A6.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B6.data(), B6.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(exact_k, dim3(1), dim3(64), 0, 0, dA, dB, dC, nk);
    std::vector<float> C(256);
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        long long q = 0;                                        // quarter units
        for (int k = 0; k < K; ++k) q += (long long)ha[i * K + k] * hb[j * K + k];
        if ((double)C[i * 16 + j] != (double)q / 4.0) {
          if (bad < 4) printf("  mismatch [%d][%d]: got %.4f want %.4f\n", i, j, C[i * 16 + j], (double)q / 4.0);
          ++bad;
        }
      }
    printf("exactness   fp6 x fp6, K = %3d: %d of 256 outputs differ from the exact integer contraction%s\n", K, bad,
           bad ? "  (assumed layout wrong or accumulation inexact)" : "");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  fflush(stdout);
}

// ---- (1) cross-wave ----------------------------------------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ v4f mx(v8i a, v8i b, v4f c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int FMT>   // FMT < 0: the i8 opcode (the r04 reference point, same harness)
__global__ __launch_bounds__(512) void cross_k(float* out, int who, int iters, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(who & 1)) return;
    if (FMT >= 0) {
      v4f acc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
      v8i a = v8i{lane, 1, 2, 3, 4, 5, 6, 7}, b = v8i{3, 2, 1, lane, 9, 8, 7, 6};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = mx<FMT < 0 ? 0 : FMT>(a, b, acc[i]);
      }
      float t = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][3];
      if (t == 123456.789f) out[0] = 1.0f;
    } else {
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
    }
  } else {
    if (!(who & 2)) return;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i) * 1e-3f;
    const float c1 = 0.999f, c2 = 0.125f;
    for (int it = 0; it < iters; ++it)
      for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
      }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += v[i];
    if (sum == 123456.789f) out[1] = sum;
  }
}

// ---- (2) intra-wave ----------------------------------------------------------------------------------------------------------
template <int FMT, int F>
__global__ __launch_bounds__(512) void intra_k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  v4f acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
  v8i a = v8i{lane, 1, 2, 3, 4, 5, 6, 7}, b = v8i{3, 2, 1, lane, 9, 8, 7, 6};
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i) * 1e-3f;
  const float c1 = 0.999f, c2 = 0.125f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] = mx<FMT>(a, b, acc[i]);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int j = (i * F + f) & 15;
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
      }
    }
  }
  float t = 0, sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { t += acc[i][0] + acc[i][3]; sum += v[i]; }
  if (t == 123456.789f || sum == 123456.789f) out[0] = sum;
}

static float timed(void (*launch)()) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipEventRecord(a));
  launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0f;
}

static float* g_out;
static int g_iters = 2000, g_who, g_reps, g_threads;

template <int FMT>
static void cross(const char* name, int reps, double ops_per_mfma) {
  g_reps = reps;
  float t[4];
  for (g_who = 1; g_who <= 3; ++g_who)
    t[g_who] = timed([] { hipLaunchKernelGGL((cross_k<FMT>), dim3(256), dim3(512), 0, 0, g_out, g_who, g_iters, g_reps); });
  const double total_ops = ops_per_mfma * 16.0 * g_iters * 4 * 256;
  printf("cross-wave  %-22s 16 MFMA/iter vs %3d v_fma/iter:  MFMA alone %7.1f us (%6.0f Tops/s, %5.1f cyc/MFMA)  VALU alone %7.1f  together %7.1f us   (sum %7.1f, max %7.1f)\n",
         name, 16 * reps, t[1], total_ops / t[1] * 1e-6, t[1] * 2400.0 / (g_iters * 16.0), t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
  fflush(stdout);
}

template <int FMT, int F>
static void intra(const char* name) {
  float t[2];
  for (int w = 0; w < 2; ++w) {
    g_threads = w ? 512 : 256;
    t[w] = timed([] { hipLaunchKernelGGL((intra_k<FMT, F>), dim3(256), dim3(g_threads), 0, 0, g_out, g_iters); });
  }
  printf("intra-wave  %-10s F=%d v_fma per MFMA:  1 wave/SIMD %7.1f us (%5.1f cyc/MFMA)   2 waves/SIMD %7.1f us (%5.1f cyc/MFMA)\n", name, F,
         t[0], t[0] * 2400.0f / (g_iters * 16.0f), t[1], t[1] * 2400.0f / (g_iters * 32.0f));
  fflush(stdout);
}

int main() {
  CK(hipMalloc(&g_out, 1 << 20));
  exactness();
  cross<-1>("i8 16x16x64", 4, 2.0 * 16 * 16 * 64);
  cross<FMT_FP8>("f8f6f4 fp8 16x16x128", 4, 2.0 * 16 * 16 * 128);
  cross<FMT_FP6>("f8f6f4 fp6 16x16x128", 4, 2.0 * 16 * 16 * 128);
  cross<FMT_FP4>("f8f6f4 fp4 16x16x128", 4, 2.0 * 16 * 16 * 128);
  cross<FMT_FP6>("f8f6f4 fp6 16x16x128", 2, 2.0 * 16 * 16 * 128);
  intra<FMT_FP6, 0>("fp6"); intra<FMT_FP6, 2>("fp6"); intra<FMT_FP6, 4>("fp6"); intra<FMT_FP6, 8>("fp6");
  intra<FMT_FP8, 0>("fp8"); intra<FMT_FP8, 4>("fp8"); intra<FMT_FP8, 8>("fp8");
  return 0;
}
