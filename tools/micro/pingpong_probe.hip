// What two SIMD partners (waves w and w + 4 of one 512-thread workgroup, one workgroup per CU) cost each other on gfx950 when they
// alternate roles under a workgroup barrier -- the ground truth behind attn4_kernel's schedule (csrc/attention.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pingpong_probe.hip -o /tmp/pingpong_probe && /tmp/pingpong_probe
// A segment of role M = 32 x v_mfma_f32_32x32x16_f16 on four accumulators (fragments in registers); of role V = 64 v_exp_f32 + 32
// v_cvt_pkrtz + 32 v_dot2 (attention's softmax stream for 64 keys x 64 queries per wave); role I = nothing (straight to the barrier).
// Modes: 0  both groups M in every segment (the pipe shared);       1  M | V ping-pong (group 1 one segment behind);
//        2  group 0: M, group 1: idle at the barrier;               3  group 0: V, group 1: idle;
//        4  M | V ping-pong with s_setprio 1 in V;                  5  M | V ping-pong with s_setprio 1 in M;
//        8  both groups run M and V INTERLEAVED in one stream (1 MFMA, 2 exp, cvt, dot2, sub) in every segment;  9  group 0 only.
// PAD: s_nop cycles of the M wave behind each of its MFMAs.
// Prints cycles per segment (s_memtime around the loop / segments) for wave 0 and wave 4 of workgroup 0, and the kernel's wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int MODE, int PAD = 0>
__global__ __launch_bounds__(512, 1) void probe(const half8_t* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps, int segs) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), group = wave >> 2;
  half8_t a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(i * 64 + lane) & 1023]; b[i] = in[(512 + i * 64 + lane) & 1023]; }
  for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((half8_t*)smem)[i] = in[i & 1023];
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float sc[64];
  for (int i = 0; i < 64; ++i) sc[i] = -0.01f * (float)(lane + i);
  float ls = 0.f;
  const fp16x2_t ones2 = {(__fp16)1.0f, (__fp16)1.0f};
  __syncthreads();
  auto seg_m = [&]() {
    if (MODE == 5) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 7], acc[j], 0, 0, 0);
        // PAD x 8 idle cycles of THIS wave behind every MFMA: its next MFMA does not sit at the head of the SIMD's vector issue
        // while the matrix pipe is busy, so the partner's VALU instructions can issue
        if (PAD >= 1) asm volatile("s_nop 7");
        if (PAD >= 2) asm volatile("s_nop 7");
        if (PAD >= 3) asm volatile("s_nop 7");
        if (PAD >= 4) asm volatile("s_nop 3");
      }
    if (MODE == 5) __builtin_amdgcn_s_setprio(0);
  };
  auto seg_v = [&]() {
    if (MODE == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float e0 = __builtin_amdgcn_exp2f(sc[2 * i]), e1 = __builtin_amdgcn_exp2f(sc[2 * i + 1]);
      const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(e0, e1);
      ls = __builtin_amdgcn_fdot2(pk, ones2, ls, false);
      sc[2 * i] = e0 - 1.0f;  // keeps the stream dependent on itself across segments, off the critical path
    }
    if (MODE == 4) __builtin_amdgcn_s_setprio(0);
  };
  float pe0 = __builtin_amdgcn_exp2f(sc[0]), pe1 = __builtin_amdgcn_exp2f(sc[1]);  // mode 8 / 9: the exponentials one slot ahead
  auto seg_mv = [&]() {  // the work of one M and one V segment interleaved in ONE wave's stream: per slot 1 MFMA, the two exponentials of the
                         // NEXT pair and pack / row sum / update of the current one (no instruction waits for the one in front of it)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 7], acc[j], 0, 0, 0);
        const int k = 4 * i + j, kn = (k + 1) & 31;
        const float n0 = __builtin_amdgcn_exp2f(sc[2 * kn]), n1 = __builtin_amdgcn_exp2f(sc[2 * kn + 1]);
        const fp16x2_t pk = __builtin_amdgcn_cvt_pkrtz(pe0, pe1);
        ls = __builtin_amdgcn_fdot2(pk, ones2, ls, false);
        sc[2 * k] = pe0 - 1.0f;
        pe0 = n0;
        pe1 = n1;
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  f32x4 acc4[16];
  if (MODE == 10 || MODE == 11)
    for (int j = 0; j < 16; ++j) acc4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto seg_m16 = [&]() {  // the same FLOP as seg_m with v_mfma_f32_16x16x32_f16: 64 MFMAs on sixteen 16 x 16 accumulators (the same 64 registers)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + (j >> 2)) & 7], b[(i + j) & 7], acc4[j], 0, 0, 0);
  };
  const auto barrier = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < segs; ++s) {
    const bool even = ((s + group) & 1) == 0;
    if (MODE == 0) seg_m();
    else if (MODE == 2) { if (group == 0) seg_m(); }
    else if (MODE == 3) { if (group == 0) seg_v(); }
    else if (MODE == 10) { if (group == 0) seg_m16(); }
    else if (MODE == 11) { if (even) seg_m16(); else seg_v(); }
    else if (MODE == 8) seg_mv();
    else if (MODE == 9) { if (group == 0) seg_mv(); }
    else { if (even) seg_m(); else seg_v(); }
    barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && lane == 0) stamps[wave] = t1 - t0;
  float r = ls;
  for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
  for (int i = 0; i < 64; ++i) r += sc[i];
  if (MODE == 10 || MODE == 11) for (int j = 0; j < 16; ++j) r += acc4[j][0] + acc4[j][1] + acc4[j][2] + acc4[j][3];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, int PAD = 0>
void run(const half8_t* in, float* out, unsigned long long* st, const char* what) {
  const int segs = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, PAD>), dim3(256), dim3(512), 0, 0, in, out, st, segs);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<MODE, PAD>), dim3(256), dim3(512), 0, 0, in, out, st, segs);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8];
  CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
  printf("mode %d pad %d  %-62s | cycles per segment: wave0 %6.0f wave4 %6.0f | %7.1f ns per segment\n", MODE, PAD, what, (double)h[0] / segs, (double)h[4] / segs, ms * 1e6 / segs);
}

int main() {
  half8_t* in; float* out; unsigned long long* st;
  CK(hipMalloc(&in, 1024 * sizeof(half8_t))); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&st, 64));
  _Float16 h[8192];
  srand(3);
  for (int i = 0; i < 8192; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 4096.f);
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>(in, out, st, "both groups M in every segment");
  run<2>(in, out, st, "group 0 M, group 1 idle at the barrier");
  run<3>(in, out, st, "group 0 V, group 1 idle at the barrier");
  run<1>(in, out, st, "M | V ping-pong");
  run<4>(in, out, st, "M | V ping-pong, s_setprio 1 in V");
  run<5>(in, out, st, "M | V ping-pong, s_setprio 1 in M");
  run<10>(in, out, st, "group 0 M as 64 x v_mfma_f32_16x16x32_f16, group 1 idle");
  run<11>(in, out, st, "M (16x16x32) | V ping-pong");
  run<9>(in, out, st, "group 0: M and V interleaved in one stream (1 MFMA : 5 VALU); group 1 idle");
  run<8>(in, out, st, "both groups: M and V interleaved in one stream, every segment");
  run<1, 1>(in, out, st, "M | V ping-pong, 8 idle cycles behind every MFMA");
  run<1, 2>(in, out, st, "M | V ping-pong, 16 idle cycles behind every MFMA");
  run<1, 3>(in, out, st, "M | V ping-pong, 24 idle cycles behind every MFMA");
  run<1, 4>(in, out, st, "M | V ping-pong, 28 idle cycles behind every MFMA");
  run<2, 3>(in, out, st, "group 0 M with 24 idle cycles per MFMA, group 1 idle");
  run<0, 3>(in, out, st, "both groups M with 24 idle cycles per MFMA");
  return 0;
}
