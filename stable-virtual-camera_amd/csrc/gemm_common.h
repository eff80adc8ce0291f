// Shared by gemm.hip (128x128 two-stage kernel) and gemm_ring.hip (8-wave multi-stage ring kernel).
#pragma once
#include "seva_common.h"

static __device__ uint4 g_zero_page[8];  // 128 B of zeros: source of padding taps (zero-initialised)

struct SevaGemmArgs {
  const half_t* a;
  const half_t* w;
  const float* bias;
  const float* row_add;
  const float* residual;
  float* out_f32;
  half_t* out_f16;
  uint8_t* out_f8;        // FP8 kernels, GEGLU epilogue: e4m3 hidden activations (the next fp8 GEMM's A operand)
  const uint8_t* w_exp;   // FP8 kernels: per-output-channel E8M0 scale byte (127 + e): weight row n is q_n * 2^e
  float* sk_ws;           // split-K workspace (flags + raw partial tiles), or null: see gemm.hip
  float* ch_stats;        // optional [ceil(M / 64)][2][N]: per 64-row block and output channel, sum and sum of squares of out_f32
  const half_t* a2;       // MODE 3: second A operand [M][lda2] whose K2 columns follow the conv's 9 * cin (K = 9 cin + K2)
  int64_t lda2;
  int32_t nk1;            // MODE 3: K-tiles of the conv part (9 * cin / 64)
  int64_t M, N, K;        // FP8 kernels: K, lda, cin count 2-byte units (= pairs of e4m3 elements)
  int64_t lda, ldr, ldo32, ldo16, ldo8;
  int64_t rows_per_group, ldra;
  int32_t n, ih, iw, cin, oh, ow, stride, upsample;
  int32_t pad_lo;    // conv: zero rows/cols before pixel 0 (1, or 0 for bottom/right-only padding)
  int32_t tiles_m, tiles_n;
  int32_t n_chunks;  // each block walks tiles_n / n_chunks consecutive N-tiles of one M-tile
  float col_scale;   // features < col_scale_n are multiplied by col_scale (plain epilogue)
  int32_t col_scale_n;
  int32_t stagger;   // start-delay quantum (s_sleep units of 64 clocks), 0 = none
  int32_t dbg;       // ablation bits (SEVA_GEMM_DBG, timing only, results wrong): 1 no loads after the
                     // first stage, 2 no MFMA, 4 every block loads tile (0,0)
};
typedef SevaGemmArgs GemmArgs;

static __device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                   0);
}

// LDS-DMA issued from inline asm: hipcc (ROCm 7.2) treats the builtin as an LDS store that may alias
// every later ds_read and protects it with `s_waitcnt vmcnt(0)`, which drains a multi-stage prefetch
// ring at every K-tile.  Hidden in asm, the DMA is ordered only by the kernel's own counted
// s_waitcnt vmcnt(N) + barrier.  M0 (LDS byte address of the wave's 1 KiB destination) is written in
// the same statement that consumes it and restored afterwards (compiler-reserved register).
static __device__ __forceinline__ unsigned lds_addr_u32(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
static __device__ __forceinline__ void glds16_raw(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_wave_base)
      : "memory");
}

// Workgroup-to-workgroup hand-off of raw accumulator tiles (split-K / stream-K): agent-scope (sc1) relaxed atomic dword stores and
// loads, which go past the XCD-private L2.  Protocol: sc1 payload stores -> s_waitcnt vmcnt(0) -> barrier -> sc1 flag store;
// sc1 flag load (spin) -> barrier -> sc1 payload loads.  Two things that were tried and are NOT done:
//  * an agent-scope release / acquire FENCE pair around plain accesses: a fence writes back / invalidates the whole L2 of the
//    XCD, and with one per workgroup the operand panels of every other workgroup on that XCD kept being evicted (stream-K ran
//    25 % slower than the unsplit kernel; even the producer-side release alone cost the split-K path its whole gain);
//  * hand-written `global_load_dwordx4 ... sc1` asm for the payload: hipcc moved the destination registers before the
//    hand-placed wait and a few 16-byte pieces (sometimes garbage) reached the accumulators -- only when successive launches
//    sent DIFFERENT data through the workspace (tools/sk_repro.py).  These are compiler-tracked.
static __device__ __forceinline__ void st_coherent_x4(float* p, f32x4 v) {  // four agent-scope (sc1) dword stores
#pragma unroll
  for (int r = 0; r < 4; ++r) __hip_atomic_store(p + r, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ f32x4 ld_coherent_x4(const float* p) {  // four agent-scope (sc1) dword loads, compiler-tracked
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = __hip_atomic_load(p + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
static __device__ __forceinline__ void publish_coherent() {  // after the payload stores, before the barrier that precedes the flag
#ifdef SEVA_HANDOFF_RELEASE_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // L2 write-back + wait
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the agent-scope stores of this thread are acknowledged
#endif
}

// Bijective XCD-aware remap (blocks b, b+8, ... share an XCD): each XCD gets one contiguous run of
// logical tiles, so the A row-panel it streams is fetched into that XCD's L2 once.
static __device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  const int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + (bid >> 3);
}

// sum over the 16 lanes of a DPP row, result in every lane (row_ror 8, 4, 2, 1: a fixed association)
static __device__ __forceinline__ float row16_sum(float v) {
#define SEVA_ROR_ADD(N)                                                                                              \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false))
  SEVA_ROR_ADD(8);
  SEVA_ROR_ADD(4);
  SEVA_ROR_ADD(2);
  SEVA_ROR_ADD(1);
#undef SEVA_ROR_ADD
  return v;
}

// conv_win.hip: 3x3 / stride 1 / pad 1 conv with the tile's input window staged in LDS.  0 = launched, 1 = not applicable, < 0 = error
// e4m3 x e4m3 -> fp32 on the block-scaled MFMA (gemm.hip: FP8); wscale = four packed E8M0 weight-scale bytes, OPSEL picks one
typedef int v8i_t __attribute__((ext_vector_type(8)));
template <int OPSEL>
static __device__ __forceinline__ f32x4 mfma_f8_sel(half8_t w_lo, half8_t w_hi, half8_t a_lo, half8_t a_hi, f32x4 c, int wscale) {
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  const v4i_t wl = __builtin_bit_cast(v4i_t, w_lo), wh = __builtin_bit_cast(v4i_t, w_hi);
  const v4i_t al = __builtin_bit_cast(v4i_t, a_lo), ah = __builtin_bit_cast(v4i_t, a_hi);
  const v8i_t w = {wl[0], wl[1], wl[2], wl[3], wh[0], wh[1], wh[2], wh[3]};
  const v8i_t a = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, c, 0 /*e4m3*/, 0 /*e4m3*/, OPSEL, wscale, 0, 0x7F7F7F7F);
}
// scale byte (j & 3) of the packed per-lane scale word of block j
static __device__ __forceinline__ f32x4 mfma_f8(int j, half8_t w_lo, half8_t w_hi, half8_t a_lo, half8_t a_hi, f32x4 c, int wscale) {
  switch (j & 3) {
    case 0: return mfma_f8_sel<0>(w_lo, w_hi, a_lo, a_hi, c, wscale);
    case 1: return mfma_f8_sel<1>(w_lo, w_hi, a_lo, a_hi, c, wscale);
    case 2: return mfma_f8_sel<2>(w_lo, w_hi, a_lo, a_hi, c, wscale);
    default: return mfma_f8_sel<3>(w_lo, w_hi, a_lo, a_hi, c, wscale);
  }
}


int seva_conv_win_launch(const GemmArgs& a, hipStream_t s, bool fp8 = false);
