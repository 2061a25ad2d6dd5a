// Do i8 MFMAs and VALU work share a SIMD's time?  (lab note for docs/history/DESIGN_rounds_1-4.md section 8, integer GEMM)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tuning/mfma_valu_overlap.hip -o tools/tuning/mfma_valu_overlap
// Round 4's first probe (i8_v4.hip: overlap_probe) found MFMA waves + VALU waves exactly additive -- but its "VALU" loop
// had been SLP-packed by hipcc into v_pk_fma_f32, which the microarchitecture guide lists as an anti-lever beside MFMAs.
// This probe pins the VALU instruction with inline assembly and asks three questions:
//   (1) cross-wave: waves 0-3 of a 512-thread block issue v_mfma_i32_16x16x64_i8, waves 4-7 issue ONE kind of VALU
//       instruction (scalar fma / packed fma / rndne / med3 / exp / cvt_pk_u8): alone, alone, together;
//   (2) intra-wave: one wave per SIMD, F scalar fmas placed after every MFMA, F = 0..8: where does the time start to rise?
//   (3) the same with two waves per SIMD both running the interleaved stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum { K_FMA = 0, K_PKFMA, K_RNDNE, K_MED3, K_EXP, K_CVTU8, K_MUL, K_KINDS };
static const char* kKind[] = {"v_fma_f32", "v_pk_fma_f32", "v_rndne_f32", "v_med3_f32", "v_exp_f32", "v_cvt_pk_u8_f32", "v_mul_f32"};

template <int KIND>
__device__ __forceinline__ void valu16(float (&v)[16], v2f (&p)[8], float c1, float c2) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
    if (KIND == K_PKFMA && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(v2f{c1, c1}), "v"(v2f{c2, c2}));
    if (KIND == K_RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]));
    if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
    if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    if (KIND == K_CVTU8) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(v[i]) : "v"(c1));
    if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
  }
}

// (1) cross-wave.  `reps` VALU groups of 16 (8 packed) instructions per iteration against 16 MFMAs per iteration.
template <int KIND>
__global__ __launch_bounds__(512) void cross_k(float* out, int who, int iters, int reps) {
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
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = v2f{v[2 * i], v[2 * i + 1]};
    const float c1 = 0.999f, c2 = 0.125f;
    for (int it = 0; it < iters; ++it)
      for (int r = 0; r < reps; ++r) valu16<KIND>(v, p, c1, c2);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += p[i][0] + p[i][1];
    if (sum == 123456.789f) out[1] = sum;
  }
}

// (2)/(3) intra-wave: F scalar VALU instructions of KIND after every MFMA, 16 independent accumulators.
template <int KIND, int F>
__global__ __launch_bounds__(512) void intra_k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  v4i acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = v4i{0, 0, 0, 0};
  v4i a = v4i{lane, 1, 2, 3}, b = v4i{3, 2, 1, lane};
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(lane + i) * 1e-3f;
  const float c1 = 0.999f, c2 = 0.125f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int j = (i * F + f) & 15;
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
        if (KIND == K_RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[j]));
        if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
        if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
      }
    }
  }
  int t = 0;
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { t ^= acc[i][0] ^ acc[i][3]; sum += v[i]; }
  if (t == 0x7ffffff1 || sum == 123456.789f) out[0] = sum;
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

template <int KIND>
static void cross(int reps) {
  g_reps = reps;
  float t[4];
  for (g_who = 1; g_who <= 3; ++g_who)
    t[g_who] = timed([] { hipLaunchKernelGGL((cross_k<KIND>), dim3(256), dim3(512), 0, 0, g_out, g_who, g_iters, g_reps); });
  const int per_it = (KIND == K_PKFMA ? 8 : 16) * reps;
  printf("cross-wave  %-16s %3d/iter vs 16 MFMA/iter:  MFMA alone %7.1f  VALU alone %7.1f  together %7.1f us   (sum %7.1f, max %7.1f)\n",
         kKind[KIND], per_it, t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
  fflush(stdout);
}

template <int KIND, int F>
static void intra() {
  float t[2];
  for (int w = 0; w < 2; ++w) {
    g_threads = w ? 512 : 256;
    t[w] = timed([] { hipLaunchKernelGGL((intra_k<KIND, F>), dim3(256), dim3(g_threads), 0, 0, g_out, g_iters); });
  }
  // cycles per MFMA slot at 2.4 GHz, per SIMD: 1 wave/SIMD issues iters*16 MFMAs; 2 waves/SIMD issue twice that
  printf("intra-wave  %-12s F=%d fillers per MFMA:  1 wave/SIMD %7.1f us (%5.1f cyc/MFMA)   2 waves/SIMD %7.1f us (%5.1f cyc/MFMA)\n", kKind[KIND], F,
         t[0], t[0] * 2400.0f / (g_iters * 16.0f), t[1], t[1] * 2400.0f / (g_iters * 32.0f));
  fflush(stdout);
}

int main() {
  CK(hipMalloc(&g_out, 1 << 20));
  cross<K_FMA>(4); cross<K_FMA>(7);
  cross<K_MUL>(4);
  cross<K_PKFMA>(7);
  cross<K_RNDNE>(4);
  cross<K_MED3>(4);
  cross<K_EXP>(2);
  cross<K_CVTU8>(4);
  intra<K_FMA, 0>(); intra<K_FMA, 1>(); intra<K_FMA, 2>(); intra<K_FMA, 3>(); intra<K_FMA, 4>(); intra<K_FMA, 6>(); intra<K_FMA, 8>();
  intra<K_RNDNE, 2>(); intra<K_RNDNE, 4>();
  intra<K_MED3, 2>(); intra<K_MED3, 4>();
  intra<K_EXP, 1>(); intra<K_EXP, 2>();
  return 0;
}
