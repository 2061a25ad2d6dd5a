// VALU issue-rate probe for gfx950: wave-cycles per instruction for scalar fp32 FMA, packed fp32 FMA / MUL / ADD,
// v_med3_f32, v_rndne_f32 and the quarter-rate transcendentals, with 8 independent chains per lane, at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tuning/valu_probe tools/tuning/valu_probe.hip && tools/tuning/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CH 8
#define IT 4096
template <int OP>
__global__ void k(float* out, float a, float b) {
  f2 v[CH];
  for (int i = 0; i < CH; ++i) v[i] = f2{a + i + threadIdx.x, b + i};
  const f2 m = {a, a}, c = {b, b};
  for (int it = 0; it < IT; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (OP == 0) v[i].x = __builtin_fmaf(v[i].x, a, b);
      if (OP == 1) v[i] = __builtin_elementwise_fma(v[i], m, c);
      if (OP == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(m));
      if (OP == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(c));
      if (OP == 4) v[i].x = __builtin_amdgcn_fmed3f(v[i].x, a, b);
      if (OP == 5) asm volatile("v_rndne_f32 %0, %1" : "=v"(v[i].x) : "v"(v[i].x));
      if (OP == 6) v[i].x = __builtin_amdgcn_exp2f(v[i].x);
      if (OP == 7) v[i].x = __builtin_amdgcn_rcpf(v[i].x);
      if (OP == 8) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i].x) : "v"(v[i].x), "v"(a));
    }
  }
  float s = 0;
  for (int i = 0; i < CH; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd;   // 256 CUs x (256 threads = 4 waves = 1 per SIMD) x waves_per_simd
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  const double inst_per_simd = (double)IT * CH * waves_per_simd;          // wave-instructions one SIMD issues
  printf("%-14s waves/SIMD %d: %8.1f us  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name,
         waves_per_simd, ms * 1e3, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w); run<8>("v_mul_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_mul_f32", w); run<3>("v_pk_add_f32", w);
    run<4>("v_med3_f32", w); run<5>("v_rndne_f32", w); run<6>("v_exp_f32", w); run<7>("v_rcp_f32", w);
  }
  return 0;
}
