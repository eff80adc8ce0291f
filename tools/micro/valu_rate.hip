// Issue cost of the VALU / transcendental / MFMA instructions of the attention tile loop, per wave-instruction and SIMD:
// each wave runs a long unrolled stream of ONE instruction kind on independent registers; with W waves per SIMD the time per
// (instruction x wave) tells whether the unit is shared (time ~ W) and what one issue costs in cycles.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = seed + threadIdx.x * 1e-3f + i;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  half8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(seed + i); hb[i] = (_Float16)(seed - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
        if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
        if (KIND == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
        if (KIND == 4) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
        if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + 1) & 15]));
        if (KIND == 7) asm volatile("v_exp_f32 %0, %0\n\tv_mul_f32 %1, %1, %1" : "+v"(r[i]), "+v"(r[(i + 8) & 15]));  // trans + plain pairs
      }
      if (KIND == 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
      }
      if (KIND == 9 || KIND == 10) {  // interleaved stream: per group 1 MFMA + 3 exp (+ 2 half-rate VALU for KIND 10)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
          asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2" : "+v"(r[3 * g]), "+v"(r[3 * g + 1]), "+v"(r[3 * g + 2]));
          if (KIND == 10) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n\tv_max3_f32 %1, %1, %0, %0" : "+v"(r[12 + (g & 1)]), "+v"(r[14 + (g & 1)]));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (KIND == 11 || KIND == 12) {  // the attention slot: LDS fragment read one slot ahead, MFMA, one softmax pair unit
        extern __shared__ char lds[];
        typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
        const fp16x2 ones = {(__fp16)1.0f, (__fp16)1.0f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const half8 cur = ha;
          ha = *(const half8*)(lds + ((threadIdx.x * 16 + g * 4096 + u * 1024) & 32767));  // next slot's fragment
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, hb, acc, 0, 0, 0);
          const float e0 = __builtin_amdgcn_exp2f(r[2 * g]), e1 = __builtin_amdgcn_exp2f(r[2 * g + 1]);
          const fp16x2 pk = __builtin_amdgcn_cvt_pkrtz(e0, e1);
          r[8 + g] = __builtin_amdgcn_fdot2(pk, ones, r[8 + g], false);
          if (KIND == 12) { r[2 * g] = acc[g] * 1e-30f; r[2 * g + 1] = acc[g + 4] * 1e-30f; }  // exp inputs depend on MFMA results
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (KIND == 8) {  // dependent chain of one exp (latency)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[0]));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i] + acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int per_iter) {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 20000;
  for (int waves = 1; waves <= 4; ++waves) {  // waves per SIMD (one block of waves*4 waves per CU)
    const int threads = waves * 256;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<256, threads, 32768>>>(d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<256, threads, 32768>>>(d, iters, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ns_per = ms * 1e6 / ((double)iters * per_iter * waves);
    printf("%-28s waves/SIMD %d: %7.3f ns per (instruction x wave) on a SIMD  = %5.1f cycles @2.4 GHz\n", name, waves, ns_per, ns_per * 2.4);
  }
  hipFree(d);
}

int main() {
  run<1>("v_mul_f32", 64);
  run<0>("v_exp_f32", 64);
  run<2>("v_max3_f32", 64);
  run<3>("v_cvt_pkrtz_f16_f32", 64);
  run<4>("v_dot2c_f32_f16", 64);
  run<5>("v_mov_b32", 64);
  run<7>("v_exp_f32 + v_mul_f32 pair", 64);
  run<8>("v_exp_f32 dependent chain", 64);
  run<6>("v_mfma_f32_32x32x16_f16", 16);
  run<9>("group {1 MFMA + 3 exp}", 16);
  run<10>("group {1 MFMA + 3 exp + 2 half-rate}", 16);
  run<11>("slot {LDS frag, MFMA, pair unit}", 16);
  run<12>("slot ..., exp inputs from MFMA acc", 16);
  return 0;
}
