// Device pow / log10 (double) against the host C library, after the cast to float the PVC decoder and the pre-flattening apply:
// how often does a float word differ, and by how many double ulps do the doubles differ?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pow_check.hip -o pow_check && ./pow_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void k(const float *r, double *p, double *l, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    p[i] = pow(10.0, r[i] / 10.0);
    l[i] = log10((double)fabsf(r[i]) + 0.1);
  }
}

int main() {
  const int n = 1 << 22;
  float *r = (float *)malloc(n * 4), *dr;
  double *p = (double *)malloc(n * 8), *l = (double *)malloc(n * 8), *dp, *dl;
  srand(5);
  for (int i = 0; i < n; i++) r[i] = (float)((rand() / (double)RAND_MAX) * 160.0 - 20.0);
  hipMalloc(&dr, n * 4); hipMalloc(&dp, n * 8); hipMalloc(&dl, n * 8);
  hipMemcpy(dr, r, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dr, dp, dl, n);
  hipMemcpy(p, dp, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(l, dl, n * 8, hipMemcpyDeviceToHost);
  long pf = 0, pd = 0, lf = 0, ld = 0, pmax = 0, lmax = 0;
  for (int i = 0; i < n; i++) {
    const double hp = pow(10.0, r[i] / 10.0), hl = log10((double)fabsf(r[i]) + 0.1);
    int64_t a, b;
    memcpy(&a, &hp, 8); memcpy(&b, &p[i], 8);
    long d = labs(a - b);
    pd += d != 0; if (d > pmax) pmax = d;
    pf += (float)hp != (float)p[i];
    memcpy(&a, &hl, 8); memcpy(&b, &l[i], 8);
    d = labs(a - b);
    ld += d != 0; if (d > lmax) lmax = d;
    lf += (float)hl != (float)l[i];
  }
  printf("n %d\npow(10, r/10): doubles differing %ld (max %ld ulp), float words differing %ld\nlog10: doubles differing %ld (max %ld ulp), float words differing %ld\n",
         n, pd, pmax, pf, ld, lmax, lf);
  return 0;
}
