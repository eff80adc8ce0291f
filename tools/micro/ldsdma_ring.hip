// Operand-stream microbenchmark for the GEMM core (round 3): how fast does the LDS-DMA stream of a tiled GEMM run as a
// function of (a) how many bytes a workgroup keeps in flight and (b) the shape of one DMA piece?
//
// The kernel is the GEMM main loop with its indexing kept (A [M][K] row panel per M-tile, W [N][K] tile, XCD remap, one
// tile per workgroup) and its arithmetic optional:
//   VAR 0  full K-tile stages (BK = 64: 8 rows x 128 B per DMA piece), 2 LDS buffers: issue stage kt+1, work on kt,
//          vmcnt(0), barrier                                   -- the production structure
//   VAR 1  half stages (BK = 32: 16 rows x 64 B per piece), ring of 4 half-buffers, THREE half-stages ahead, counted vmcnt
//   VAR 2  as 1, TWO half-stages ahead
//   VAR 3  full stages, ring of 3 buffers, two stages ahead, counted vmcnt (needs 1.5x the LDS)
// WORK bit 0: fragment reads (ds_read_b128, conflict-free swizzle) + MFMAs of the real tile; bit 1: MFMAs only (on registers)
// Output: us per launch, LDS-DMA bytes / clk / CU (at the nominal 2.4 GHz), TFLOP/s where WORK != 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include <vector>
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ void glds16_raw(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_wave_base)
               : "memory");
}
template <int N>
static __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
static __device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  const int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + (bid >> 3);
}
static __device__ __forceinline__ int f4(int q) { return ((q & 1) << 1) ^ (((q >> 1) & 1) * 3); }  // 64-B-row swizzle key

template <int BM, int BN, int NW, int VAR, int WORK>
__global__ __launch_bounds__(64 * NW, 8 / NW) void k(const half_t* A, const half_t* W, long M, long N, long K, int tiles_m,
                                                      int tiles_n, float* sink) {
  constexpr bool HALF = VAR == 1 || VAR == 2;
  constexpr int NBUF = VAR == 0 ? 2 : VAR == 3 ? 3 : 4;
  constexpr int ROWB = HALF ? 64 : 128;              // bytes per LDS row
  constexpr int RPP = 1024 / ROWB;                   // rows per DMA piece
  constexpr int BUF_BYTES = (BM + BN) * ROWB;
  constexpr int PIECES = (BM + BN) / RPP;            // per (half-)stage
  static_assert(PIECES % NW == 0, "pieces must split evenly over the waves");
  constexpr int P = PIECES / NW;                     // per wave
  constexpr int WMW = NW == 8 ? 4 : 2, WNW = 2, WM = BM / WMW, WN = BN / WNW, MI = WM / 16, NJ = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int work = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = work / tiles_n, tn = work - tm * tiles_n;
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // per-wave pieces: piece id = wave + NW * i; rows [id * RPP, +RPP) of the stacked [A rows | B rows] image
  const half_t* src[P];
  unsigned dst[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int id = wave + NW * i;
    const int row = id * RPP + (HALF ? lane >> 2 : lane >> 3);  // stacked row
    const int pc = HALF ? lane & 3 : lane & 7;
    const bool isA = row < BM;
    const int r = isA ? row : row - BM;
    const int q = HALF ? pc ^ f4((r >> 2) & 3) : pc ^ ((r >> 1) & 7);
    long g = isA ? m0 + r : n0 + r;
    const long lim = isA ? M : N;
    if (g >= lim) g = lim - 1;
    src[i] = (isA ? A : W) + g * K + q * 8;
    dst[i] = id * 1024;
  }
  auto issue = [&](int step, int buf) {  // step = K-tile (full) or half-step
#pragma unroll
    for (int i = 0; i < P; ++i) glds16_raw(src[i] + (long)step * (ROWB / 2), lds0 + buf * BUF_BYTES + dst[i]);
  };
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8_t ra = {1, 2, 3, 4, 5, 6, 7, 8}, rb = {1, 1, 1, 1, 1, 1, 1, 1};
  // one 32-deep k-step of the wave's tile out of buffer `buf` (s: which 64-B half of a 128-B row, full stages only)
  auto kstep = [&](int buf, int s) {
    if (WORK == 0) return;
    const char* base = smem + buf * BUF_BYTES;
    half8_t af[MI], bf[NJ];
    if (WORK & 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * WM + 16 * i + fr;
        const int off = HALF ? r * 64 + ((fg ^ f4((r >> 2) & 3)) << 4) : r * 128 + (((4 * s + fg) ^ ((r >> 1) & 7)) << 4);
        af[i] = *(const half8_t*)(base + off);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = wn * WN + 16 * j + fr;
        const int off = HALF ? (BM + r) * 64 + ((fg ^ f4((r >> 2) & 3)) << 4)
                             : (BM + r) * 128 + (((4 * s + fg) ^ ((r >> 1) & 7)) << 4);
        bf[j] = *(const half8_t*)(base + off);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = ra;
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = rb;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
  };
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const int nk = (int)(K / 64);
  if (VAR == 0) {
    issue(0, 0);
    wait_vm<0>();
    barrier();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      kstep(kt & 1, 0);
      kstep(kt & 1, 1);
      wait_vm<0>();
      barrier();
    }
  } else if (VAR == 3) {
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vm<P>(); else wait_vm<0>();  // stage kt landed; kt+1 may fly
      barrier();
      int nb = buf + 2; if (nb >= 3) nb -= 3;
      if (kt + 2 < nk) issue(kt + 2, nb);  // buffer of kt-1: every wave passed the barrier after reading it
      kstep(buf, 0);
      kstep(buf, 1);
      if (++buf == 3) buf = 0;
    }
  } else {
    constexpr int D = VAR == 1 ? 3 : 2;  // half-stages ahead
    const int nh = 2 * nk;
    for (int h = 0; h < D && h < nh; ++h) issue(h, h & 3);
    for (int h = 0; h < nh; ++h) {
      // half-stage h landed; up to D-1 younger ones may fly (tail: fewer were issued -> stricter waits are still right)
      const int younger = nh - 1 - h < D - 1 ? nh - 1 - h : D - 1;
      if (younger >= 2) wait_vm<2 * P>();
      else if (younger == 1) wait_vm<P>();
      else wait_vm<0>();
      barrier();
      if (h + D < nh) issue(h + D, (h + D) & 3);  // D = 3: buffer of h-1, free since this barrier; D = 2: buffer of h-2
      kstep(h & 3, 0);
    }
  }
  if (sink) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) t += acc[i][j];
    if (t[0] + t[1] + t[2] + t[3] == 12345.678f) sink[threadIdx.x] = t[0];
  }
}

template <int BM, int BN, int NW, int VAR, int WORK>
static float run(const half_t* A, const half_t* W, long M, long N, long K, float* sink, int reps) {
  constexpr bool HALF = VAR == 1 || VAR == 2;
  constexpr int NBUF = VAR == 0 ? 2 : VAR == 3 ? 3 : 4;
  constexpr int lds = NBUF * (BM + BN) * (HALF ? 64 : 128);
  auto fn = k<BM, BN, NW, VAR, WORK>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(tm * tn), dim3(64 * NW), lds, 0, A, W, M, N, K, tm, tn, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

static void need(long M, long N, long K) {
  if (M * K > 54432L * 5760 || N * K > 10240L * 5120 || K % 128) { printf("shape exceeds the allocations\n"); exit(2); }
}

template <int BM, int BN, int NW>
static void sweep(const char* name, const half_t* A, const half_t* W, long M, long N, long K, float* sink) {
  need(M, N, K);
  const long tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  const double dma_bytes = (double)tm * tn * (BM + BN) * 2.0 * K;
  const double flops = 2.0 * M * N * K;
  printf("%s  M=%ld N=%ld K=%ld tile %dx%d waves %d (%ld tiles, %.2f GB through LDS-DMA)\n", name, M, N, K, BM, BN, NW, tm * tn,
         dma_bytes / 1e9);
  auto line = [&](const char* what, float us, bool fl) {
    printf("   %-46s %8.1f us  %5.1f B/clk/CU", what, us, dma_bytes / (us * 1e-6) / 256.0 / 2.4e9);
    if (fl) printf("  %6.0f TFLOP/s", flops / (us * 1e-6) / 1e12);
    printf("\n");
  };
  constexpr bool fit3 = 3 * (BM + BN) * 128 <= 160 * 1024;
  line("DMA only  full stage, 2 buffers (production)", run<BM, BN, NW, 0, 0>(A, W, M, N, K, sink, 5), false);
  line("DMA only  half stages, 2 ahead", run<BM, BN, NW, 2, 0>(A, W, M, N, K, sink, 5), false);
  line("DMA only  half stages, 3 ahead", run<BM, BN, NW, 1, 0>(A, W, M, N, K, sink, 5), false);
  if constexpr (fit3) line("DMA only  full stages, 3 buffers, 2 ahead", run<BM, BN, NW, 3, 0>(A, W, M, N, K, sink, 5), false);
  line("MFMA regs only + DMA  full stage, 2 buffers", run<BM, BN, NW, 0, 2>(A, W, M, N, K, sink, 5), true);
  line("MFMA regs only + DMA  half stages, 3 ahead", run<BM, BN, NW, 1, 2>(A, W, M, N, K, sink, 5), true);
  line("reads+MFMA + DMA  full stage, 2 buffers", run<BM, BN, NW, 0, 1>(A, W, M, N, K, sink, 5), true);
  line("reads+MFMA + DMA  half stages, 2 ahead", run<BM, BN, NW, 2, 1>(A, W, M, N, K, sink, 5), true);
  line("reads+MFMA + DMA  half stages, 3 ahead", run<BM, BN, NW, 1, 1>(A, W, M, N, K, sink, 5), true);
  if constexpr (fit3) line("reads+MFMA + DMA  full stages, 3 buffers", run<BM, BN, NW, 3, 1>(A, W, M, N, K, sink, 5), true);
}

// VAR 5: the A operand never touches LDS.  Waves are stacked 4 x 1 (32 rows x the full 160-column tile each: the A-in-registers
// layout of the production ASTAT kernel), the W tile is staged by LDS-DMA as before (2 buffers of 160 x 128 B), and every wave
// loads the A fragments of the NEXT K-tile straight from global memory into a second register set (fragment-shaped: 16 rows x
// 64 B per wave-instruction) while it computes on the current one.  LDS-DMA bytes per FLOP: 0.0078 instead of 0.0125 (160 x 160)
// or 0.0141 (128 x 160 with A staged); bytes into the CU per FLOP 0.0137.
template <int WORK>
__global__ __launch_bounds__(256, 2) void k_adirect(const half_t* A, const half_t* W, long M, long N, long K, int tiles_m, int tiles_n,
                                                    float* sink) {
  constexpr int BM = 128, BN = 160, NW = 4, MI = 2, NJ = 10, P = BN / 8 / NW;  // 5 DMA pieces per wave and stage
  constexpr int BUF_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int work = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = work / tiles_n, tn = work - tm * tiles_n;
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const half_t* src[P];
  unsigned dst[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int id = wave + NW * i, r = id * 8 + (lane >> 3), q = (lane & 7) ^ ((r >> 1) & 7);
    long g = n0 + r;
    if (g >= N) g = N - 1;
    src[i] = W + g * K + q * 8;
    dst[i] = id * 1024;
  }
  const int fr = lane & 15, fg = lane >> 4;
  const half_t* ap[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    long g = m0 + 32 * wave + 16 * i + fr;
    if (g >= M) g = M - 1;
    ap[i] = A + g * K + fg * 8;
  }
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < P; ++i) glds16_raw(src[i] + (long)kt * 64, lds0 + buf * BUF_BYTES + dst[i]);
  };
  auto loadA = [&](half8_t (&a)[MI][2], int kt) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) a[i][s] = *(const half8_t*)(ap[i] + (long)kt * 64 + s * 32);
  };
  auto compute = [&](const half8_t (&a)[MI][2], int buf) {
    if (WORK == 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" ::"v"(a[i][s]));
      return;
    }
    const char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8_t bf[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = 16 * j + fr;
        bf[j] = *(const half8_t*)(base + r * 128 + (((4 * s + fg) ^ ((r >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], a[i][s], acc[i][j], 0, 0, 0);
    }
  };
  auto sync = [&]() {
    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const int nk = (int)(K / 64);  // even in every shape used here
  half8_t a0[MI][2], a1[MI][2];
  issue(0, 0);
  loadA(a0, 0);
  sync();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) { issue(kt + 1, 1); loadA(a1, kt + 1); }
    compute(a0, 0);
    sync();
    if (kt + 1 < nk) {
      if (kt + 2 < nk) { issue(kt + 2, 0); loadA(a0, kt + 2); }
      compute(a1, 1);
      sync();
    }
  }
  if (sink) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) t += acc[i][j];
    if (t[0] + t[1] + t[2] + t[3] == 12345.678f) sink[threadIdx.x] = t[0];
  }
}

template <int WORK>
static float run_adirect(const half_t* A, const half_t* W, long M, long N, long K, float* sink, int reps) {
  constexpr int lds = 2 * 160 * 128;
  auto fn = k_adirect<WORK>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tm = (int)((M + 127) / 128), tn = (int)((N + 159) / 160);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(tm * tn), dim3(256), lds, 0, A, W, M, N, K, tm, tn, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

static void sweep_adirect(const char* name, const half_t* A, const half_t* W, long M, long N, long K, float* sink) {
  need(M, N, K);
  const long tm = (M + 127) / 128, tn = (N + 159) / 160;
  const double dma = (double)tm * tn * 160 * 2.0 * K, direct = (double)tm * tn * 128 * 2.0 * K, flops = 2.0 * M * N * K;
  printf("%s  M=%ld N=%ld K=%ld tile 128x160, A straight to registers (%ld tiles, %.2f GB LDS-DMA + %.2f GB direct)\n", name, M, N, K,
         tm * tn, dma / 1e9, direct / 1e9);
  const float t0 = run_adirect<0>(A, W, M, N, K, sink, 5), t1 = run_adirect<1>(A, W, M, N, K, sink, 5);
  printf("   %-46s %8.1f us  %5.1f B/clk/CU into the CU (DMA part %.1f)\n", "loads only   W by LDS-DMA + A to VGPRs", t0,
         (dma + direct) / (t0 * 1e-6) / 256.0 / 2.4e9, dma / (t0 * 1e-6) / 256.0 / 2.4e9);
  printf("   %-46s %8.1f us  %5.1f B/clk/CU into the CU (DMA part %.1f)  %6.0f TFLOP/s\n", "reads+MFMA + W by LDS-DMA + A to VGPRs", t1,
         (dma + direct) / (t1 * 1e-6) / 256.0 / 2.4e9, dma / (t1 * 1e-6) / 256.0 / 2.4e9, flops / (t1 * 1e-6) / 1e12);
}

// VAR 6 ("ping-pong"): ONE 8-wave workgroup per CU, ring of THREE full stages (two in flight), the waves in two groups of four
// (waves 0-3 / 4-7: one wave of each group per SIMD) that run the same program ONE PHASE APART: while group 0 multiplies K-tile k
// (M phase: MFMAs only, fragments already in registers) group 1 reads its fragments of K-tile k out of LDS and issues its share
// of stage k+2 (R phase), and vice versa; one workgroup barrier per phase.  Stage k+2 goes into the buffer of stage k-1, which
// both groups have finished reading one barrier earlier; a wave waits for its pieces of stage k+1 (counted vmcnt: stage k+2 stays
// in flight) at the end of phase 2k+1, one barrier before group 0 reads them.  SPLIT of a wave's DMA pieces are issued inside
// its M phase (between the MFMAs) instead of its R phase.
template <int BM, int BN, int WORK, int SPLIT>
__global__ __launch_bounds__(512, 1) void k_pingpong(const half_t* A, const half_t* W, long M, long N, long K, int tiles_m, int tiles_n,
                                                     float* sink) {
  constexpr int TP = (BM + BN) / 8;                  // DMA pieces per stage
  constexpr int PMAX = (TP + 7) / 8;
  constexpr int BUF_BYTES = (BM + BN) * 128;
  constexpr int WM = BM / 4, WN = BN / 2, MI = WM / 16, NJ = WN / 16;
  static_assert(TP % 8 == 0 || TP % 8 == 4, "pieces: equal inside a group of four waves");
  static_assert(3 * BUF_BYTES <= 160 * 1024, "three stages must fit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int row0 = grp * (BM / 2) + (wq >> 1) * WM, col0 = (wq & 1) * WN;  // the wave's output tile
  const int work = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = work / tiles_n, tn = work - tm * tiles_n;
  const long m0 = (long)tm * BM, n0 = (long)tn * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const half_t* src[PMAX];
  unsigned dst[PMAX];
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    int id = wave + 8 * i;
    if (id >= TP) id = TP - 1;  // (never issued: the group's piece count excludes it)
    const int row = id * 8 + (lane >> 3), pc = lane & 7;
    const bool isA = row < BM;
    const int r = isA ? row : row - BM;
    const int q = pc ^ ((r >> 1) & 7);
    long g = isA ? m0 + r : n0 + r;
    const long lim = isA ? M : N;
    if (g >= lim) g = lim - 1;
    src[i] = (isA ? A : W) + g * K + q * 8;
    dst[i] = id * 1024;
  }
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8_t af[2][MI], bf[2][NJ];
  const int nk = (int)(K / 64);
  auto barrier = [&]() {  // phase boundary: nothing (MFMAs included) may be scheduled across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  auto body = [&](auto gtag) {
    constexpr int G = decltype(gtag)::value;
    constexpr int PG = (TP % 8 == 0) ? TP / 8 : (G == 0 ? TP / 8 + 1 : TP / 8);  // this group's pieces per stage
    constexpr int SP = SPLIT < PG ? SPLIT : PG;                                      // issued in the M phase
    auto issue = [&](int kt, int i0, int i1) {
      int buf = kt % 3;
#pragma unroll
      for (int i = 0; i < PMAX; ++i)
        if (i >= i0 && i < i1) glds16_raw(src[i] + (long)kt * 64, lds0 + buf * BUF_BYTES + dst[i]);
    };
    auto R = [&](int k) {
      if (k + 2 < nk) issue(k + 2, SP, PG);
      if (WORK) {
        const char* base = smem + (k % 3) * BUF_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int r = row0 + 16 * i + fr;
            af[s][i] = *(const half8_t*)(base + r * 128 + (((4 * s + fg) ^ ((r >> 1) & 7)) << 4));
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int r = col0 + 16 * j + fr;
            bf[s][j] = *(const half8_t*)(base + (BM + r) * 128 + (((4 * s + fg) ^ ((r >> 1) & 7)) << 4));
          }
        }
      }
    };
    auto Mph = [&](int k) {
      if (WORK) {
        constexpr int TOT = 2 * MI * NJ, EVERY = SP > 0 ? TOT / (SP + 1) : TOT + 1;
        int cnt = 0, ip = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
              ++cnt;
              if (SP > 0 && cnt % EVERY == 0 && ip < SP) {
                __builtin_amdgcn_sched_barrier(0);
                if (k + 2 < nk) issue(k + 2, ip, ip + 1);
                __builtin_amdgcn_sched_barrier(0);
                ++ip;
              }
            }
      } else if (SP > 0) {
        if (k + 2 < nk) issue(k + 2, 0, SP);
      }
    };
    // prologue: stages 0 and 1 (every wave: all of its pieces)
    issue(0, 0, PG);
    if (nk > 1) issue(1, 0, PG);
    if (nk > 1) wait_vm<PG>(); else wait_vm<0>();
    barrier();
    if (G == 1) barrier();  // group 1 starts one phase late
    for (int k = 0; k < nk; ++k) {
      R(k);
      if (G == 1) {  // end of phase 2k+1: stage k+1 (read by group 0 in the next phase) has landed; the R part of stage k+2 may fly
        if (k + 2 < nk) wait_vm<PG - SP>(); else wait_vm<0>();
      }
      barrier();
      Mph(k);
      if (G == 0) {  // end of phase 2k+1: the same wait for this group's pieces of stage k+1; all of stage k+2 may fly
        if (k + 2 < nk) wait_vm<PG>(); else wait_vm<0>();
        barrier();
      } else if (k + 1 < nk) {
        barrier();
      }
    }
  };
  if (grp == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
  if (sink) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) t += acc[i][j];
    if (t[0] + t[1] + t[2] + t[3] == 12345.678f) sink[threadIdx.x] = t[0];
  }
}

template <int BM, int BN, int WORK, int SPLIT>
static float run_pingpong(const half_t* A, const half_t* W, long M, long N, long K, float* sink, int reps) {
  constexpr int lds = 3 * (BM + BN) * 128;
  auto fn = k_pingpong<BM, BN, WORK, SPLIT>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(tm * tn), dim3(512), lds, 0, A, W, M, N, K, tm, tn, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

template <int BM, int BN>
static void sweep_pingpong(const char* name, const half_t* A, const half_t* W, long M, long N, long K, float* sink) {
  need(M, N, K);
  const long tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  const double dma = (double)tm * tn * (BM + BN) * 2.0 * K, flops = 2.0 * M * N * K;
  printf("%s  M=%ld N=%ld K=%ld tile %dx%d, 8 waves ping-pong, 3 stages (%ld tiles = %.2f rounds of 256, %.2f GB through LDS-DMA)\n", name,
         M, N, K, BM, BN, tm * tn, tm * tn / 256.0, dma / 1e9);
  auto line = [&](const char* what, float us, bool fl) {
    printf("   %-46s %8.1f us  %5.1f B/clk/CU", what, us, dma / (us * 1e-6) / 256.0 / 2.4e9);
    if (fl) printf("  %6.0f TFLOP/s", flops / (us * 1e-6) / 1e12);
    printf("\n");
  };
  line("DMA only, all pieces in the R phase", run_pingpong<BM, BN, 0, 0>(A, W, M, N, K, sink, 5), false);
  line("reads+MFMA + DMA, all pieces in the R phase", run_pingpong<BM, BN, 1, 0>(A, W, M, N, K, sink, 5), true);
  line("reads+MFMA + DMA, 2 pieces in the M phase", run_pingpong<BM, BN, 1, 2>(A, W, M, N, K, sink, 5), true);
  line("reads+MFMA + DMA, 3 pieces in the M phase", run_pingpong<BM, BN, 1, 3>(A, W, M, N, K, sink, 5), true);
  line("reads+MFMA + DMA, all pieces in the M phase", run_pingpong<BM, BN, 1, 8>(A, W, M, N, K, sink, 5), true);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const long Kmax = 5120;
  half_t *A, *W;
  float* sink;
  const long Aelems = 54432L * 5760;  // largest A used: 54432 x 5760 (> 217728 x 1280)
  hipMalloc(&A, Aelems * 2);
  hipMalloc(&W, 10240 * Kmax * 2);
  hipMalloc(&sink, 4096);
  // random-ish fill (clock under load depends on the data): small integers scaled
  {
    std::vector<half_t> h(1 << 24);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (half_t)(((int)(s >> 20) % 2001 - 1000) / 1000.0f); }
    for (long off = 0; off < Aelems; off += (long)h.size())
      hipMemcpy(A + off, h.data(), std::min<long>(h.size(), Aelems - off) * 2, hipMemcpyHostToDevice);
    for (long off = 0; off < 10240 * Kmax; off += (long)h.size())
      hipMemcpy(W + off, h.data(), std::min<long>(h.size(), 10240 * Kmax - off) * 2, hipMemcpyHostToDevice);
  }
  if (getenv("RING_FULL")) {
  sweep<160, 160, 4>("ds2 ff2 +res    ", A, W, 54432, 640, 2560, sink);
  sweep_adirect("ds2 ff2 +res    ", A, W, 54432, 640, 2560, sink);
  sweep_adirect("ds2 attn_out    ", A, W, 54432, 640, 640, sink);
  sweep<160, 160, 4>("ds4 ff2         ", A, W, 13608, 1280, 5120, sink);
  sweep_adirect("ds4 ff2         ", A, W, 13608, 1280, 5120, sink);
  sweep_adirect("ds1 ff2         ", A, W, 217728, 320, 1280, sink);
  sweep<160, 160, 4>("ds4 geglu-like  ", A, W, 13608, 10240, 1280, sink);
  sweep_adirect("ds4 geglu-like  ", A, W, 13608, 10240, 1280, sink);
  sweep<128, 128, 4>("ds2 conv-like128", A, W, 54432, 640, 5760, sink);
  sweep_adirect("ds2 conv-like   ", A, W, 54432, 640, 5760, sink);
  sweep<256, 128, 8>("ds2 ff2 8 waves ", A, W, 54432, 640, 2560, sink);
  sweep<256, 128, 8>("ds4 ff2 8 waves ", A, W, 13608, 1280, 5120, sink);
  sweep<192, 192, 8>("ds4 ff2 8w 192  ", A, W, 13608, 1280, 5120, sink);
  }
  sweep<160, 160, 4>("ds4 ff2         ", A, W, 13608, 1280, 5120, sink);
  sweep_pingpong<192, 192>("ds4 ff2         ", A, W, 13608, 1280, 5120, sink);
  sweep_pingpong<192, 160>("ds4 ff2         ", A, W, 13608, 1280, 5120, sink);
  sweep_pingpong<256, 160>("ds4 ff2         ", A, W, 13608, 1280, 5120, sink);
  sweep<160, 160, 4>("ds2 ff2 +res    ", A, W, 54432, 640, 2560, sink);
  sweep_pingpong<192, 160>("ds2 ff2 +res    ", A, W, 54432, 640, 2560, sink);
  sweep_pingpong<256, 160>("ds2 ff2 +res    ", A, W, 54432, 640, 2560, sink);
  sweep_pingpong<192, 160>("ds2 attn_out    ", A, W, 54432, 640, 640, sink);
  sweep_pingpong<192, 160>("ds1 ff2         ", A, W, 217728, 320, 1280, sink);
  sweep_pingpong<256, 160>("ds1 ff2         ", A, W, 217728, 320, 1280, sink);
  sweep<160, 160, 4>("ds2 geglu-like  ", A, W, 54432, 5120, 640, sink);
  sweep_pingpong<192, 160>("ds2 geglu-like  ", A, W, 54432, 5120, 640, sink);
  sweep_pingpong<256, 160>("ds2 geglu-like  ", A, W, 54432, 5120, 640, sink);
  sweep_pingpong<192, 160>("ds2 conv-like   ", A, W, 54432, 640, 5760, sink);
  return 0;
}
