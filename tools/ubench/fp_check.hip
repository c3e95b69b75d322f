// IEEE check of the float / double operations esbr_core.h relies on: device results against the host's (x86-64 SSE2).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp_check.hip -o fp_check && ./fp_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__host__ __device__ inline void ops(float a, float b, double c, double d, float *of, double *od) {
  of[0] = a / b;
  of[1] = 1.0f / a;
  of[2] = (float)sqrt((double)a);
  of[3] = (float)sqrt(c);
  of[4] = (float)sqrt(a * c / (b + 1));
  of[5] = (float)(a * (b / (a + 1e-17)));
  of[6] = a * b + a;
  of[7] = (float)sqrt(a * c / fabs(b + 1e-17));
  od[0] = c / d;
  od[1] = sqrt(c);
  od[2] = a / (1 + a + 1e-17);
  od[3] = c * d;
}

__global__ void k(const float *a, const float *b, const double *c, const double *d, float *of, double *od, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ops(a[i], b[i], c[i], d[i], of + 8 * i, od + 4 * i);
}

int main() {
  const int n = 1 << 20;
  float *a = (float *)malloc(n * 4), *b = (float *)malloc(n * 4), *of = (float *)malloc(n * 32), *hf = (float *)malloc(n * 32);
  double *c = (double *)malloc(n * 8), *d = (double *)malloc(n * 8), *od = (double *)malloc(n * 32), *hd = (double *)malloc(n * 32);
  srand(1);
  for (int i = 0; i < n; i++) {
    a[i] = (float)ldexp((double)rand() / RAND_MAX + 0.5, rand() % 60 - 20);
    b[i] = (float)ldexp((double)rand() / RAND_MAX + 0.5, rand() % 60 - 20);
    c[i] = ldexp((double)rand() / RAND_MAX + 0.5, rand() % 80 - 30);
    d[i] = ldexp((double)rand() / RAND_MAX + 0.5, rand() % 80 - 30);
    ops(a[i], b[i], c[i], d[i], hf + 8 * i, hd + 4 * i);
  }
  float *da, *db, *dof; double *dc, *dd, *dod;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 8); hipMalloc(&dd, n * 8); hipMalloc(&dof, n * 32); hipMalloc(&dod, n * 32);
  hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dc, c, n * 8, hipMemcpyHostToDevice); hipMemcpy(dd, d, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dd, dof, dod, n);
  hipMemcpy(of, dof, n * 32, hipMemcpyDeviceToHost); hipMemcpy(od, dod, n * 32, hipMemcpyDeviceToHost);
  for (int j = 0; j < 8; j++) {
    int bad = 0;
    for (int i = 0; i < n; i++) bad += memcmp(&of[8 * i + j], &hf[8 * i + j], 4) != 0;
    printf("float op %d: %d / %d differ\n", j, bad, n);
  }
  for (int j = 0; j < 4; j++) {
    int bad = 0;
    for (int i = 0; i < n; i++) bad += memcmp(&od[4 * i + j], &hd[4 * i + j], 8) != 0;
    printf("double op %d: %d / %d differ\n", j, bad, n);
  }
  return 0;
}
