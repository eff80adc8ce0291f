// Shared device/host helpers for libseva_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/seva_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEVA_WAVE 64

// host side ----------------------------------------------------------------------------------
void seva_set_error(const char* fmt, ...);
int seva_check_launch(const char* what);

#define SEVA_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      seva_set_error(__VA_ARGS__);       \
      return SEVA_ERR_ARG;               \
    }                                    \
  } while (0)

// Benchmark / debugging knobs (capi.hip).  Read ONCE from the SEVA_* environment when the library is loaded (no
// getenv on the launch path); tests and tools change them at run time through seva_set_knob().  -1 = unset.
struct SevaKnobs {
  int gemm_chunks, gemm_dbg, gemm_stagger, gemm_bm, gemm_bn, gemm_astat;
  int attn_dbg, attn_no_tr, attn_two, attn_split;
  int gn_min_iter;
  int conv_win;  // 0: per-tap gather everywhere; 1: window kernel, two 4-wave workgroups per CU; 2: 8-wave 256-row tile; unset: default
};
extern SevaKnobs g_seva_knobs;

// profiling (capi.hip) -- brackets a launch with events when enabled
struct SevaProfScope {
  int cls;
  double work;   // algorithmic FLOP (classes 0-2) or bytes (classes 3-4)
  double bytes;  // algorithmic HBM bytes: every operand read once, every result written once
  hipStream_t stream;
  hipEvent_t e0, e1;
  bool on;
  SevaProfScope(int cls, double work, hipStream_t stream, double bytes = -1.0);  // bytes < 0: same as work
  ~SevaProfScope();
};

// device side --------------------------------------------------------------------------------
// THE PACKED-FP32 FIRST-READER RULE (checked on the emitted ISA of every kernel of the library by
// tests/test_isa_concurrency_cpu.py, tools/isa_lint.py; reproducer tools/micro/pk_first_reader.hip).  Round 3 found two kernels that
// were not repeatable while ANOTHER kernel shared the CU: in both, a `v_pk_*_f32` instruction was the first reader of a VGPR that a
// memory-pipeline return (VMEM load, `ds_bpermute_b32`) had just written, and now and then it computed with the register's previous
// content in lanes 48-63.  With a plain 32-bit VALU instruction as the first reader both were clean over thousands of launches.
// hipcc (ROCm 7.2) forms packed fp32 math freely (vector types, the SLP vectoriser), so every value that comes out of a load /
// LDS read / lane permute and may feed packed math goes through first_read(): one `v_mov_b32` in place (no extra register), which
// the compiler cannot fold away and which is then the register's first reader.
#ifdef SEVA_NO_FIRST_READ  // A/B builds only (make variant EXTRA=-DSEVA_NO_FIRST_READ): what the rule costs; never shipped (the ISA test fails)
__device__ __forceinline__ float first_read(float v) { return v; }
__device__ __forceinline__ f32x2 first_read(f32x2 v) { return v; }
__device__ __forceinline__ f32x4 first_read(f32x4 v) { return v; }
#else
__device__ __forceinline__ float first_read(float v) {
  asm("v_mov_b32 %0, %0" : "+v"(v));
  return v;
}
__device__ __forceinline__ f32x2 first_read(f32x2 v) {
  asm("v_mov_b32 %0, %0" : "+v"(v[0]));
  asm("v_mov_b32 %0, %0" : "+v"(v[1]));
  return v;
}
__device__ __forceinline__ f32x4 first_read(f32x4 v) {
  asm("v_mov_b32 %0, %0" : "+v"(v[0]));
  asm("v_mov_b32 %0, %0" : "+v"(v[1]));
  asm("v_mov_b32 %0, %0" : "+v"(v[2]));
  asm("v_mov_b32 %0, %0" : "+v"(v[3]));
  return v;
}
#endif
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// four floats -> four OCP e4m3 bytes (round to nearest even, saturating at +-448), byte i = value i
__device__ __forceinline__ int pack_fp8x4(float a, float b, float c, float d) {
  const auto cl = [](float v) { return __builtin_fminf(__builtin_fmaxf(v, -448.f), 448.f); };
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), r, true);
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// GELU with the exact (erf) definition, reference F.gelu at seva/modules/transformer.py:15.
// erf by Abramowitz & Stegun 7.1.26: |abs error| <= 1.5e-7, branch-free, 2 transcendentals + 9 FMA-class
// ops.  libm erff costs ~55 instructions with two divergent branches per call, which made the GEGLU
// epilogue 3x longer than the K=320 main loop; the approximation error is 3 orders of magnitude below
// the fp16 resolution of the value it produces.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);  // exp(-z^2)
  const float erf_abs = fmaf(-poly, e, 1.0f);                                  // erf(|x|/sqrt2)
  const float erf_signed = copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erf_signed);
}

// GEGLU epilogue value * gelu(gate) on four features at once, written on float2 so that hipcc emits
// packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth per issue slot).  erf by
// Abramowitz & Stegun 7.1.28, erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16 (|error| <= 3e-7; measured
// against float64 GELU over [-12, 12]: max abs error 8.8e-7): one quarter-rate op (rcp) per value
// instead of two (rcp + exp2) and 14 packed full-rate ops per pair.  q^16 overflows to +inf for
// |gate| > ~26, where rcp gives 0 and erf 1, the correct limit.
__device__ __forceinline__ f32x2 geglu2(f32x2 v, f32x2 g) {
  const auto c2 = [](float c) { return f32x2{c, c}; };
  const f32x2 z = __builtin_elementwise_abs(g) * 0.70710678118654752440f;
  f32x2 q = __builtin_elementwise_fma(z, c2(0.0000430638f), c2(0.0002765672f));
  q = __builtin_elementwise_fma(q, z, c2(0.0001520143f));
  q = __builtin_elementwise_fma(q, z, c2(0.0092705272f));
  q = __builtin_elementwise_fma(q, z, c2(0.0422820123f));
  q = __builtin_elementwise_fma(q, z, c2(0.0705230784f));
  q = __builtin_elementwise_fma(q, z, c2(1.0f));
  q = q * q;
  q = q * q;
  q = q * q;
  q = q * q;
  const f32x2 r = {__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1])};
  const f32x2 e = __builtin_elementwise_fma(r, c2(-1.0f), c2(1.0f));  // erf(|g| / sqrt2)
  const f32x2 es = {__builtin_copysignf(e[0], g[0]), __builtin_copysignf(e[1], g[1])};
  const f32x2 h = g * 0.5f;
  return v * __builtin_elementwise_fma(h, es, h);
}
__device__ __forceinline__ f32x4 geglu4(f32x4 v, f32x4 g) {
  const f32x2 lo = geglu2(f32x2{v[0], v[1]}, f32x2{g[0], g[1]});
  const f32x2 hi = geglu2(f32x2{v[2], v[3]}, f32x2{g[2], g[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
