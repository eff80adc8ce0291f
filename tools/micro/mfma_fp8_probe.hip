// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit (E8M0 = 127) block scales:
// checks the assumed operand map  lane l: A[row l&15][k = 32*(l>>4) + j], B[k = 32*(l>>4) + j][col l&15], j = 0..31
// and the C/D map col = l&15, row = 4*(l>>4) + reg, on exact small-integer data.   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ static uint8_t enc_e4m3(int v) {  // exact for |v| <= 8 integers (and more); OCP e4m3fn
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0;
  int a = v < 0 ? -v : v;
  int e = 0;
  while ((a >> (e + 1)) > 0) ++e;          // a in [2^e, 2^(e+1))
  int m = ((a << 3) >> e) & 7;             // 3 mantissa bits (exact when a < 16 and representable)
  return s | (uint8_t)((e + 7) << 3) | (uint8_t)m;
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D) {  // A [16][128], B [128][16] (k-major rows), D [16][16]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  v8i a, b;
  uint8_t ab[32], bb[32];
  for (int j = 0; j < 32; ++j) {
    ab[j] = A[r * 128 + 32 * g + j];
    bb[j] = B[(32 * g + j) * 16 + r];
  }
  for (int w = 0; w < 8; ++w) {
    a[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    b[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A fp8*/, 0 /*B fp8*/, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}

// Block scales: lane (row r, k-block g) supplies the E8M0 scale of ITS 32 elements of A in byte `opsel` of the scale VGPR.
__global__ void probe_scale(const uint8_t* A, const uint8_t* B, const uint8_t* sA, float* D, float* D3) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  v8i a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = 0; b[w] = 0;
    for (int t = 0; t < 4; ++t) {
      a[w] |= (int)A[r * 128 + 32 * g + 4 * w + t] << (8 * t);
      b[w] |= (int)B[(32 * g + 4 * w + t) * 16 + r] << (8 * t);
    }
  }
  const int sc = sA[r * 4 + g];  // scale byte of (row r, k-block g)
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sc | 0x55555500, 0, 0x7F7F7F7F);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
  f32x4 c3 = {0.f, 0.f, 0.f, 0.f};  // same scale in byte 3, selected with opsel 3
  c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 3, (sc << 24) | 0x00555555, 0, 0x7F7F7F7F);
  for (int i = 0; i < 4; ++i) D3[(4 * g + i) * 16 + r] = c3[i];
}

int main() {
  uint8_t hA[16 * 128], hB[128 * 16];
  int iA[16 * 128], iB[128 * 16];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; hA[i] = enc_e4m3(iA[i]); }
  for (int i = 0; i < 128 * 16; ++i) { iB[i] = rand() % 7 - 3; hB[i] = enc_e4m3(iB[i]); }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  float hD[256];
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      int ref = 0;
      for (int k = 0; k < 128; ++k) ref += iA[m * 128 + k] * iB[k * 16 + n];
      if (hD[m * 16 + n] != (float)ref) { if (bad < 5) printf("mismatch D[%d][%d] = %g, want %d\n", m, n, hD[m * 16 + n], ref); ++bad; }
    }
  printf("mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit scales), assumed operand map: %s (%d mismatches)\n", bad ? "WRONG" : "EXACT", bad);
  uint8_t hS[64];
  for (int i = 0; i < 64; ++i) hS[i] = (uint8_t)(127 + (rand() % 7 - 3));  // 2^-3 .. 2^3 per (row, k-block)
  uint8_t* dS; float* dD3;
  hipMalloc(&dS, 64); hipMalloc(&dD3, 256 * 4);
  hipMemcpy(dS, hS, 64, hipMemcpyHostToDevice);
  probe_scale<<<1, 64>>>(dA, dB, dS, dD, dD3);
  float hD3[256];
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  hipMemcpy(hD3, dD3, sizeof hD3, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double ref = 0;
      for (int k = 0; k < 128; ++k) ref += ldexp((double)(iA[m * 128 + k] * iB[k * 16 + n]), hS[m * 4 + k / 32] - 127);
      if (hD[m * 16 + n] != (float)ref || hD3[m * 16 + n] != (float)ref) {
        if (bad2 < 5) printf("scale mismatch D[%d][%d] = %g / %g, want %g\n", m, n, hD[m * 16 + n], hD3[m * 16 + n], ref);
        ++bad2;
      }
    }
  printf("per-lane E8M0 scale of the A operand (row = lane&15, k-block = lane>>4), opsel 0 and 3: %s (%d mismatches)\n",
         bad2 ? "WRONG" : "EXACT", bad2);
  return (bad != 0) | (bad2 != 0);
}
