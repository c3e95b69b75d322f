// Dependent-issue latency of one wave: cycles per instruction for chains of 1, 2, 4 independent VALU adds,
// mul_hi, and LDS round trips, with 1 wave per workgroup and 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int OP>
__global__ void chain(unsigned *out, long long *cyc, int iters) {
  __shared__ unsigned lds[256];
  unsigned v[CHAINS];
  for (int c = 0; c < CHAINS; c++) v[c] = threadIdx.x + c;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(u + 1));
        if (OP == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(v[c]) : "v"(0x7fff1234 + u));
        if (OP == 2) { v[c] = lds[v[c] & 63]; asm volatile("" : "+v"(v[c])); }
      }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  unsigned s = 0;
  for (int c = 0; c < CHAINS; c++) s += v[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS, int OP>
void run(const char *name, int blocks) {
  unsigned *out;
  long long *cyc;
  hipMalloc(&out, blocks * 64 * 4);
  hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((chain<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((chain<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += c;
  avg /= blocks;
  printf("%-28s chains=%d blocks=%5d : %.2f cycles per instruction (per wave), %.2f per chain step\n", name, CHAINS, blocks,
         avg / (iters * 16.0 * CHAINS), avg / (iters * 16.0));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int blocks : {256, 1024, 2048, 4096, 8192}) { /* 256 CUs: 1, 4, 8, 16, 32 waves per CU */
    run<1, 0>("v_add_u32 dependent", blocks);
    run<4, 0>("v_add_u32 4 chains", blocks);
    run<1, 1>("v_mul_hi_i32 dependent", blocks);
    run<1, 2>("ds_read dependent", blocks);
    run<4, 2>("ds_read 4 chains", blocks);
  }
  return 0;
}
