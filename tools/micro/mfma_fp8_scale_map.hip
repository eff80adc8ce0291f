// Which lane's E8M0 scale byte applies to which (row/col, k-block) of v_mfma_scale_f32_16x16x128_f8f6f4?
// A = B = ones (e4m3 0x38), A k-block g of row r weighted by marker so a doubled scale shows where it lands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(int L, int which, int opsel, float* D) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  v8i a, b;
  for (int w = 0; w < 8; ++w) { a[w] = 0x38383838; b[w] = 0x38383838; }  // 1.0
  int sa = 0x7F7F7F7F, sb = 0x7F7F7F7F;
  if (l == L) { if (which == 0) sa = 0x7F7F7F7F + (1 << (8 * opsel)); else sb = 0x7F7F7F7F + (1 << (8 * opsel)); }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa, 1, sb);
  else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sa, 2, sb);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sa, 3, sb);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}

int main() {
  float* dD; hipMalloc(&dD, 1024);
  float h[256];
  for (int which = 0; which < 2; ++which)
    for (int opsel = 0; opsel < 4; opsel += 3)
      for (int L = 0; L < 64; L += (L < 20 ? 1 : 11)) {
        k<<<1, 64>>>(L, which, opsel, dD);
        hipMemcpy(h, dD, 1024, hipMemcpyDeviceToHost);
        int nrow = 0, ncol = 0, row = -1, col = -1; float val = 128;
        for (int m = 0; m < 16; ++m) { int c = 0; for (int n = 0; n < 16; ++n) if (h[m * 16 + n] != 128.f) { ++c; val = h[m * 16 + n]; } if (c == 16) { ++nrow; row = m; } }
        for (int n = 0; n < 16; ++n) { int c = 0; for (int m = 0; m < 16; ++m) if (h[m * 16 + n] != 128.f) ++c; if (c == 16) { ++ncol; col = n; } }
        int nd = 0; for (int i = 0; i < 256; ++i) if (h[i] != 128.f) ++nd;
        printf("%s scale, opsel %d, lane %2d doubled: %3d entries differ; full rows %d (row %d), full cols %d (col %d), value %g\n",
               which ? "B" : "A", opsel, L, nd, nrow, row, ncol, col, val);
      }
  return 0;
}
