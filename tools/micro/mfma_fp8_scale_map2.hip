// For v_mfma_scale_f32_16x16x128_f8f6f4: the scale byte of lane (row 3, group g) scales WHICH operand bytes of row 3?
// Operands and scales come from memory (host-prepared); A is one only in 16 bytes [32*g2 + 16*h, +16) of row 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* A, const uint8_t* B, const int* SA, const int* SB, float* D) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  v8i a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = 0; b[w] = 0;
    for (int t = 0; t < 4; ++t) {
      a[w] |= (int)A[r * 128 + 32 * g + 4 * w + t] << (8 * t);
      b[w] |= (int)B[(32 * g + 4 * w + t) * 16 + r] << (8 * t);
    }
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, SA[l], 0, SB[l]);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
int main() {
  uint8_t hA[2048], hB[2048]; int hSA[64], hSB[64]; float hD[256];
  uint8_t *dA, *dB; int *dSA, *dSB; float* dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dD, 1024);
  memset(hB, 0x38, 2048);
  for (int i = 0; i < 64; ++i) hSB[i] = 0x7F7F7F7F;
  hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 256, hipMemcpyHostToDevice);
  for (int g = 0; g < 4; ++g) {
    printf("A-scale of lane (row 3, group %d) scales A bytes:", g);
    for (int g2 = 0; g2 < 4; ++g2) for (int h = 0; h < 2; ++h) {
      memset(hA, 0, 2048);
      memset(hA + 3 * 128 + 32 * g2 + 16 * h, 0x38, 16);
      for (int i = 0; i < 64; ++i) hSA[i] = 0x7F7F7F7F;
      hSA[16 * g + 3] = 0x7F7F7F80;
      hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dSA, hSA, 256, hipMemcpyHostToDevice);
      k<<<1, 64>>>(dA, dB, dSA, dSB, dD);
      hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
      const float v = hD[3 * 16 + 0];
      if (v == 32.f) printf(" [%d,%d)", 32 * g2 + 16 * h, 32 * g2 + 16 * h + 16); else if (v != 16.f) printf(" ?%g", v);
    }
    printf("\n");
  }
  // same for the B operand: B one only in k rows [32*g2+16*h, +16) of column 3; A all ones
  memset(hA, 0x38, 2048);
  for (int i = 0; i < 64; ++i) hSA[i] = 0x7F7F7F7F;
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dSA, hSA, 256, hipMemcpyHostToDevice);
  for (int g = 0; g < 4; ++g) {
    printf("B-scale of lane (col 3, group %d) scales B k-rows:", g);
    for (int g2 = 0; g2 < 4; ++g2) for (int h = 0; h < 2; ++h) {
      memset(hB, 0, 2048);
      for (int kk = 32 * g2 + 16 * h; kk < 32 * g2 + 16 * h + 16; ++kk) hB[kk * 16 + 3] = 0x38;
      for (int i = 0; i < 64; ++i) hSB[i] = 0x7F7F7F7F;
      hSB[16 * g + 3] = 0x7F7F7F80;
      hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 256, hipMemcpyHostToDevice);
      k<<<1, 64>>>(dA, dB, dSA, dSB, dD);
      hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
      const float v = hD[0 * 16 + 3];
      if (v == 32.f) printf(" [%d,%d)", 32 * g2 + 16 * h, 32 * g2 + 16 * h + 16); else if (v != 16.f) printf(" ?%g", v);
    }
    printf("\n");
  }
  return 0;
}
