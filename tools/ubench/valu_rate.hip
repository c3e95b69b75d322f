// Micro-benchmark: issue rate of the integer VALU ops the IMDCT kernel is made of, on gfx950.
// Each kernel runs ITER x 16 independent instances of one op per lane, 4 waves/SIMD resident.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 20000
#define OPS 16

template <int OP>
__global__ __launch_bounds__(256) void k(int32_t *out, int32_t seed) {
  int32_t a[OPS];
  int32_t b = seed + threadIdx.x;
#pragma unroll
  for (int i = 0; i < OPS; i++) a[i] = seed * (i + 1) + threadIdx.x;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < OPS; i++) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if (OP == 7) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 8) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 9) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
      if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  int32_t s = 0;
#pragma unroll
  for (int i = 0; i < OPS; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int32_t *d, int wps = 4) {
  const int blocks = 256 * wps;  // wps blocks of 4 waves per CU -> wps waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winst = (double)blocks * 4 * ITER * OPS;           // wave-instructions
  double per_simd = winst / 1024.0;                         // per SIMD
  double ns_per = ms * 1e6 / per_simd;
  printf("%-16s wps=%d %8.3f ms  %6.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wps, ms, ns_per, ns_per * 2.4);
}

int main() {
  int32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w = 1; w <= 8; w *= 2) { run<0>("v_add_u32", d, w); run<10>("v_fma_f32", d, w); run<1>("v_mul_hi_i32", d, w); }
  run<0>("v_add_u32", d); run<1>("v_mul_hi_i32", d); run<2>("v_lshlrev_b32", d); run<3>("v_add_i32 clamp", d);
  run<4>("v_mul_lo_u32", d); run<5>("v_mul_i32_i24", d); run<6>("v_cndmask_b32", d); run<7>("v_max3_i32", d);
  run<8>("v_mad_i32_i24", d); run<9>("v_add_lshl_u32", d); run<10>("v_fma_f32", d); run<11>("v_pk_add_u16", d);
  return 0;
}
