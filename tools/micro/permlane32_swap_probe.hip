#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o, const float* in) {
  float v = in[threadIdx.x];
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  o[threadIdx.x] = fmaxf(a, b);
  o[64 + threadIdx.x] = a;
  o[128 + threadIdx.x] = b;
}
int main() {
  float h[64], r[192], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
  hipMalloc(&d, 256); hipMalloc(&o, 768);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, d);
  hipMemcpy(r, o, 768, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) { float e = fmaxf(h[i], h[i ^ 32]); if (r[i] != e) ++bad; }
  printf("permlane32_swap half-wave max: %d mismatches; a[0]=%g a[32]=%g b[0]=%g b[32]=%g (in[0]=%g in[32]=%g)\n", bad, r[64], r[96], r[128], r[160], h[0], h[32]);
  return bad != 0;
}
