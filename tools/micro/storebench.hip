// Store-pattern microbenchmark: every wave writes 1 KiB per instruction in one of three shapes.
//   mode 0: 16 rows x 64 B  (lane = row fr, 16-B chunk fg)      -- the GEMM epilogue's f32 / paired-f16 pattern
//   mode 1:  8 rows x 128 B
//   mode 2:  4 rows x 256 B
//   mode 3: 16 rows x 32 B with 8-byte stores (the old f16 pattern), 512 B per instruction
// Rows are `pitch` bytes apart (an output matrix row).  8 waves per CU resident (2 blocks x 4 waves, like the GEMM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(char* out, long pitch, int rows_per_wave_iter, int iters, long rows_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 4 + wave;          // global wave id -> 64-row band, 1 column panel of 1 KiB/row
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int it = 0; it < iters; ++it) {
    // band of 64 rows; column offset advances with it
    const long row0 = (gw * 64) % rows_total;
    const long col = (long)it * 256;  // bytes
#pragma unroll
    for (int r = 0; r < 64; r += (MODE == 0 ? 16 : MODE == 1 ? 8 : MODE == 2 ? 4 : 16)) {
      if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f32x4*)(out + (row0 + r + (lane & 15)) * pitch + col + c * 64 + (lane >> 4) * 16) = v;
      } else if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c) *(f32x4*)(out + (row0 + r + (lane >> 3)) * pitch + col + c * 128 + (lane & 7) * 16) = v;
      } else if (MODE == 2) {
        *(f32x4*)(out + (row0 + r + (lane >> 4)) * pitch + col + (lane & 15) * 16) = v;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) *(f32x2*)(out + (row0 + r + (lane & 15)) * pitch + col + c * 32 + (lane >> 4) * 8) = f32x2{v[0], v[1]};
      }
    }
  }
}
int main() {
  const long rows = 217728, pitch = 2560 * 2;  // the ds1 GEGLU-sized f16 output: 1.1 GB
  char* buf; hipMalloc(&buf, rows * pitch);
  const int iters = pitch / 256;  // each wave writes its 64-row band completely: 256 B per row per iteration
  const int blocks = (int)(rows / 64 / 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, pitch, 0, iters, rows);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, pitch, 0, iters, rows);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, pitch, 0, iters, rows);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, pitch, 0, iters, rows);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("mode %d: %.1f us  %.2f TB/s\n", mode, ms * 1e3, rows * pitch / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
