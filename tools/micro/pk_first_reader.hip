// Standalone reproducer attempt for the packed-fp32 first-reader rule (seva_common.h).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/pk_first_reader.hip -o /tmp/pk_first_reader && /tmp/pk_first_reader [launches]
// Round 3: `v_pk_fma_f32` / `v_pk_add_f32` as the FIRST reader of registers a memory-pipeline return had just written computed,
// in 4-25 % of the launches and only while ANOTHER kernel shared the CU, with the registers' previous content in lanes 48-63.
// This program isolates the instruction sequence: the victim pre-loads a sentinel into v[20:21], loads a (w0, w1) pair over it,
// waits vmcnt(0) and forms r = w * x + b with ONE v_pk_fma_f32 -- directly (variant 0) or behind two in-place v_mov_b32 (variant
// 1, what first_read() emits).  A second stream runs a streaming copy the whole time.  Every output is checked against fmaf on
// the host; a stale read shows up as a result formed from the sentinel.  The outcome of ONE run is kept next to this file
// (pk_first_reader.log): whether the isolated sequence reproduces the fault or not, the artefact lets someone else try.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int PLAIN_FIRST>
__global__ __launch_bounds__(256) void victim(const f32x2* __restrict__ w, const float* __restrict__ x, f32x2* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float xv = x[i];
  f32x2 xx = {xv, xv}, bb = {0.25f, -0.5f}, r;
  const f32x2* p = w + i;
  const float sentinel = 12345.0f;
  asm volatile("" : "+v"(xx), "+v"(bb));  // operands are in registers (and waited for) before the sequence starts
  if (PLAIN_FIRST)
    asm volatile("v_mov_b32 v20, %3\n\tv_mov_b32 v21, %3\n\ts_nop 4\n\tglobal_load_dwordx2 v[20:21], %1, off\n\ts_waitcnt vmcnt(0)\n\t"
                 "v_mov_b32 v20, v20\n\tv_mov_b32 v21, v21\n\tv_pk_fma_f32 %0, v[20:21], %2, %4\n\ts_nop 1"
                 : "=&v"(r) : "v"(p), "v"(xx), "v"(sentinel), "v"(bb) : "v20", "v21", "memory");
  else
    asm volatile("v_mov_b32 v20, %3\n\tv_mov_b32 v21, %3\n\ts_nop 4\n\tglobal_load_dwordx2 v[20:21], %1, off\n\ts_waitcnt vmcnt(0)\n\t"
                 "v_pk_fma_f32 %0, v[20:21], %2, %4\n\ts_nop 1"
                 : "=&v"(r) : "v"(p), "v"(xx), "v"(sentinel), "v"(bb) : "v20", "v21", "memory");
  out[i] = r;
}
__global__ void stream_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400, n = 1 << 20;
  std::vector<f32x2> hw(n);
  std::vector<float> hx(n);
  srand(7);
  for (int i = 0; i < n; ++i) { hw[i] = f32x2{(float)(rand() % 2001 - 1000) / 64.f, (float)(rand() % 2001 - 1000) / 64.f}; hx[i] = (float)(rand() % 513 - 256) / 16.f; }
  f32x2 *dw, *dout; float* dx; float4 *ca, *cb;
  const size_t cn = (size_t)64 << 20;  // 1 GiB copy buffers: past the Infinity Cache
  CK(hipMalloc(&dw, n * sizeof(f32x2))); CK(hipMalloc(&dout, n * sizeof(f32x2))); CK(hipMalloc(&dx, n * sizeof(float)));
  CK(hipMalloc(&ca, cn * sizeof(float4))); CK(hipMalloc(&cb, cn * sizeof(float4)));
  CK(hipMemcpy(dw, hw.data(), n * sizeof(f32x2), hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), n * sizeof(float), hipMemcpyHostToDevice));
  CK(hipMemset(ca, 1, cn * sizeof(float4)));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  std::vector<f32x2> ho(n);
  for (int load = 0; load < 2; ++load)
    for (int variant = 0; variant < 2; ++variant) {
      long bad_launches = 0, bad_elems = 0, bad_hi_lanes = 0, stale = 0;
      for (int l = 0; l < launches; ++l) {
        if (load) hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, s2, ca, cb, cn / 8);
        CK(hipMemsetAsync(dout, 0xff, n * sizeof(f32x2), s1));
        if (variant) hipLaunchKernelGGL(victim<1>, dim3(n / 256), dim3(256), 0, s1, dw, dx, dout, n);
        else hipLaunchKernelGGL(victim<0>, dim3(n / 256), dim3(256), 0, s1, dw, dx, dout, n);
        CK(hipMemcpyAsync(ho.data(), dout, n * sizeof(f32x2), hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
        long b = 0;
        for (int i = 0; i < n; ++i) {
          const float e0 = fmaf(hw[i][0], hx[i], 0.25f), e1 = fmaf(hw[i][1], hx[i], -0.5f);
          if (ho[i][0] != e0 || ho[i][1] != e1) {
            ++b;
            bad_hi_lanes += (i & 63) >= 48;
            stale += ho[i][0] == fmaf(12345.0f, hx[i], 0.25f) || ho[i][1] == fmaf(12345.0f, hx[i], -0.5f);
          }
        }
        bad_elems += b; bad_launches += b != 0;
      }
      CK(hipDeviceSynchronize());
      printf("second stream %s, first reader %s: %ld of %d launches wrong, %ld elements (%ld in lanes 48-63, %ld formed from the sentinel)\n",
             load ? "streaming" : "idle     ", variant ? "v_mov_b32 (plain)" : "v_pk_fma_f32     ", bad_launches, launches, bad_elems, bad_hi_lanes, stale);
    }
  return 0;
}
