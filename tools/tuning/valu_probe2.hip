// Issue-rate probe, part 2 (round 4): the conversion / integer / select instructions of the integer Linear's epilogue and the
// cost of an LDS table GATHER (ds_read_b32 / b64 / b128 at per-lane pseudo-random entries of a 6 KB table vs lane-linear
// addresses), 8 independent chains per lane, at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tuning/valu_probe2 tools/tuning/valu_probe2.hip && tools/tuning/valu_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CH 8
#define IT 2048
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void k(float* out, float a, float b, int ia) {
  __shared__ __attribute__((aligned(16))) uint32_t tab[4096];          // 16 KB
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = i * 2654435761u;
  __syncthreads();
  float v[CH];
  uint32_t u[CH];
  for (int i = 0; i < CH; ++i) { v[i] = a + i + threadIdx.x; u[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ ia; }
  for (int it = 0; it < IT; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (OP == 0) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
      if (OP == 1) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
      if (OP == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(v[i]));
      if (OP == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(ia));
      if (OP == 4) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(u[i]) : "v"(u[i]), "v"(ia));
      if (OP == 5) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(ia));
      if (OP == 6) asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(v[i]), "v"(a) : "vcc");
      if (OP == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(a), "v"(b));
      if (OP == 8) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(a), "v"(b));
      // LDS: the address chain u -> read -> u keeps the reads dependent per chain, 8 chains in flight
      if (OP == 10) { u[i] = tab[u[i] & 4095]; }                                          // b32 gather, 16 KB
      if (OP == 11) { u[i] = tab[(u[i] % 1536u)]; }                                        // b32 gather, 6 KB
      if (OP == 12) { const u2 e = *reinterpret_cast<const u2*>(&tab[(u[i] % 768u) * 2]); u[i] = e.x ^ e.y; }   // b64 gather, 768 entries
      if (OP == 13) { const u4 e = *reinterpret_cast<const u4*>(&tab[(u[i] % 384u) * 4]); u[i] = e.x ^ e.w; }   // b128 gather, 384 entries
      if (OP == 14) { const u2 e = *reinterpret_cast<const u2*>(&tab[((threadIdx.x & 63) * 2 + (u[i] & 1) * 128) & 4095]); u[i] = e.x + i; }   // b64 lane-linear
      if (OP == 15) { u[i] = tab[((threadIdx.x & 63) + (u[i] & 1) * 64) & 4095] + i; }     // b32 lane-linear
    }
  }
  float s = 0;
  for (int i = 0; i < CH; ++i) s += v[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int waves_per_simd) {
  float* out;
  const int blocks = 256 * waves_per_simd;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f, 12345);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) k<OP><<<blocks, 256>>>(out, 1.0001f, 0.5f, 12345);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double inst_per_simd = (double)IT * CH * waves_per_simd;
  printf("%-34s waves/SIMD %d: %8.1f us  -> %6.2f cycles (2.4 GHz) per wave-instruction per SIMD, %6.2f per CU\n", name, waves_per_simd,
         ms * 1e3, ms * 1e6 / inst_per_simd * 2.4, ms * 1e6 / inst_per_simd * 2.4 / 4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<7>("v_fma_f32", w); run<0>("v_cvt_f32_i32", w); run<1>("v_cvt_u32_f32", w); run<2>("v_cvt_pk_u8_f32", w); run<3>("v_add_u32", w);
    run<4>("v_lshl_add_u32", w); run<5>("v_and_b32", w); run<6>("v_cmp_ge_f32 + v_cndmask_b32 (2)", w); run<8>("v_med3_f32", w);
    run<10>("ds_read_b32 gather 16 KB (+and)", w); run<11>("ds_read_b32 gather 6 KB (+mod)", w); run<12>("ds_read_b64 gather 768 x 8 B (+mod)", w);
    run<13>("ds_read_b128 gather 384 x 16 B (+mod)", w); run<14>("ds_read_b64 lane-linear", w); run<15>("ds_read_b32 lane-linear", w);
  }
  return 0;
}
