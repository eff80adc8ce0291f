// Stream-K variant of the plain-epilogue GEMM / implicit-GEMM 3x3 convolution of gemm.hip (128 x BN x 64 tiles, 4 waves 2x2,
// fp32 output with bias / row_add / fp32 residual / epilogue GroupNorm statistics), for launches whose tile count sits
// between rounds of the chip's 512 workgroup slots (the 18 x 18 level: 856 tiles = 1.67 rounds; 36 x 36: 1704 = 3.33).
//
// The (tile, K-tile) iteration space is cut into equal contiguous ranges, one per workgroup; a tile that straddles two
// ranges is computed by a CHAIN: the first workgroup runs K-tiles [0, k) from the usual start (zeros / the residual tile) and
// exports its raw accumulators; the next one LOADS THEM AS ITS ACCUMULATORS and continues with [k, nk).  Every output element
// therefore sees exactly the MFMA sequence of the unsplit kernel: results are bitwise those of gemm_kernel (tested), whatever
// the cut points -- no split-K re-association, no dependence on the batch size.
//
// Ordering / deadlock freedom: a workgroup runs the segment it EXPORTS first and the segment it IMPORTS last, and the producer
// of a workgroup's import is the workgroup 8 block ids below it (same XCD slice, dispatched earlier): by the time anyone waits,
// its producer has long published.  The wait is bounded all the same (error slot 16383 of the workspace counts give-ups).
// Workspace layout (shared with the split-K path of gemm.hip): 16384 int flags, then one 128 x BN fp32 slot per workgroup.
#include "gemm_common.h"

#include <atomic>

namespace {

constexpr int BK = 64;

__device__ __forceinline__ float row16_sum_sk(float v) {
#define SEVA_ROR_ADD(N)                                                                                              \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false))
  SEVA_ROR_ADD(8);
  SEVA_ROR_ADD(4);
  SEVA_ROR_ADD(2);
  SEVA_ROR_ADD(1);
#undef SEVA_ROR_ADD
  return v;
}

template <int BN, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_sk_kernel(GemmArgs p) {
  constexpr int BM = 128, WM = 64, WN = BN / 2, MI = WM / 16, NJ = WN / 16;
  constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const lds_a = smem;
  char* const lds_b = smem + 2 * A_BYTES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int sr = lane >> 3, sp = lane & 7;
  const int fr = lane & 15, fg = lane >> 4;
  const int nk = (int)(p.K / BK);

  // ---- this workgroup's range of (tile, K-tile) units inside its XCD slice of the tiles ----
  const int xg = blockIdx.x & 7, j = blockIdx.x >> 3, nbg = gridDim.x >> 3;
  const int T = p.tiles_m * p.tiles_n;
  const int tg0 = (int)((int64_t)xg * T / 8), tg1 = (int)((int64_t)(xg + 1) * T / 8);
  const int64_t units = (int64_t)(tg1 - tg0) * nk;
  const int64_t u0 = (int64_t)j * units / nbg, u1 = (int64_t)(j + 1) * units / nbg;
  if (u1 <= u0) return;  // (the host guarantees >= nk units per workgroup; kept for safety)
  const int t0 = (int)(u0 / nk), k0 = (int)(u0 - (int64_t)t0 * nk);
  const int t1 = (int)((u1 - 1) / nk), k1 = (int)(u1 - (int64_t)t1 * nk);  // tile t1 is run up to K-tile k1 (1..nk)

  int* const flags = (int*)p.sk_ws;
  float* const slots = p.sk_ws + 16384;

  // fragment-read byte offsets (gemm.hip: natural row order, chunk swizzle key (row >> 1) & 7)
  int a_off[2], b_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ra = wm * WM + fr;
    a_off[s] = ra * 128 + (((4 * s + fg) ^ ((ra >> 1) & 7)) << 4);
    const int rb = wn * WN + fr;
    b_off[s] = rb * 128 + (((4 * s + fg) ^ ((rb % WN >> 1) & 7)) << 4);
  }

  // One segment: K-tiles [kb, ke) of tile `tile` (relative to the slice).  imp: start from the partial the previous workgroup
  // exported; exp: export the accumulators instead of running the epilogue.
  auto run = [&](int tile, int kb, int ke, bool imp, bool exp) {
    const int gid = tg0 + tile;
    const int tm = gid / p.tiles_n, tn = gid - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    // ---- staging state ----
    const half_t* a_ptr[A_PASSES];
    unsigned a_mask[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int row = wave * (BM / 4) + 8 * i + sr;
      const int q = sp ^ ((row >> 1) & 7);
      a_mask[i] = 0;
      int64_t m = m0 + row;
      if (m >= p.M) m = p.M - 1;
      if (MODE == 0) {
        a_ptr[i] = p.a + m * p.lda + q * 8;
      } else {
        const int ohw = p.oh * p.ow;
        const int img = (int)(m / ohw);
        const int rem = (int)(m - (int64_t)img * ohw);
        const int oy = rem / p.ow, ox = rem - oy * p.ow;
        const int by = oy * p.stride - p.pad_lo, bx = ox * p.stride - p.pad_lo;
        a_ptr[i] = p.a + (int64_t)img * p.ih * p.iw * p.cin + ((int64_t)by * p.iw + bx) * p.cin + q * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = by + t / 3, ix = bx + t % 3;
          if ((iy >= 0) & (iy < p.ih) & (ix >= 0) & (ix < p.iw)) a_mask[i] |= 1u << t;
        }
      }
    }
    const half_t* b_ptr[B_PASSES];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int row = wave * (BN / 4) + 8 * i + sr;
      const int q = sp ^ ((row % WN >> 1) & 7);
      int64_t n = n0 + row;
      if (n >= p.N) n = p.N - 1;
      b_ptr[i] = p.w + n * p.K + q * 8;
    }
    int st_ky = 0, st_kx = 0, st_ci0 = 0;
    if (MODE == 1) {
      const int tap = kb * BK / p.cin;
      st_ci0 = kb * BK - tap * p.cin;
      st_ky = tap / 3;
      st_kx = tap - 3 * st_ky;
    }
    auto stage = [&](int buf, int kt) {
      char* const la = lds_a + buf * A_BYTES + wave * (BM / 4) * 128;
      char* const lb = lds_b + buf * B_BYTES + wave * (BN / 4) * 128;
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) glds16(a_ptr[i] + (int64_t)kt * BK, la + i * 1024);
      } else {
        const int ky = st_ky, kx = st_kx, ci0 = st_ci0;
        st_ci0 += BK;
        if (st_ci0 == p.cin) {
          st_ci0 = 0;
          if (++st_kx == 3) {
            st_kx = 0;
            ++st_ky;
          }
        }
        const int64_t tap_off = ((int64_t)ky * p.iw + kx) * p.cin + ci0;
        const unsigned bit = 1u << (3 * ky + kx);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          const void* g = (a_mask[i] & bit) ? (const void*)(a_ptr[i] + tap_off) : (const void*)g_zero_page;
          glds16(g, la + i * 1024);
        }
      }
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i) glds16(b_ptr[i] + (int64_t)kt * BK, lb + i * 1024);
    };

    stage(0, kb);
    f32x4 acc[MI][NJ];
    if (imp) {
      // the previous workgroup of this slice (8 block ids below) exported K-tiles [.., kb) of this tile
      const int prod = blockIdx.x - 8;
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flags + prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < (1 << 21)) {
          __builtin_amdgcn_s_sleep(32);
          ++spins;
        }
        if (spins == (1 << 21)) atomicAdd(flags + 16383, 1);
        __hip_atomic_store(flags + prod, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
      }
      __syncthreads();
      const float* const part = slots + (int64_t)prod * (BM * BN) + threadIdx.x * 4;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = ld_coherent_x4(part + (i * NJ + jj) * 1024);

    } else if (p.residual) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int64_t m = m0 + wm * WM + 16 * i + fr;
        if (m >= p.M) m = p.M - 1;
        const float* rp = p.residual + m * p.ldr;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          int64_t f = n0 + wn * WN + 16 * jj + 4 * fg;
          if (f > p.N - 4) f = p.N - 4;
          acc[i][jj] = *(const f32x4*)(rp + f);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // stage kb (and the accumulator loads) landed
    for (int kt = kb; kt < ke; ++kt) {
      const int cur = (kt - kb) & 1;
      if (kt + 1 < ke) stage(cur ^ 1, kt + 1);
      const char* const ta = lds_a + cur * A_BYTES;
      const char* const tb = lds_b + cur * B_BYTES;
      half8_t af[2][MI], bf[2][NJ];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[s][i] = *(const half8_t*)(ta + a_off[s] + i * (16 * 128));
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) bf[s][jj] = *(const half8_t*)(tb + b_off[s] + jj * (16 * 128));
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s][jj], af[s][i], acc[i][jj], 0, 0, 0);
      __syncthreads();
    }
    if (exp) {
      float* const part = slots + (int64_t)blockIdx.x * (BM * BN) + threadIdx.x * 4;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) st_coherent_x4(part + (i * NJ + jj) * 1024, acc[i][jj]);
      publish_coherent();  // this thread's stores are visible device-wide
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // ---- epilogue (gemm.hip, EPI == 0, natural row order): bias, row_add, stores, GroupNorm statistics ----
    f32x4 bj[NJ];
    int fj[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      int64_t f = n0 + wn * WN + 16 * jj + 4 * fg;
      if (f > p.N - 4) f = p.N - 4;
      fj[jj] = (int)f;
      bj[jj] = p.bias ? *(const f32x4*)(p.bias + f) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int64_t m = m0 + wm * WM + 16 * i + fr;
      const int64_t mc = m < p.M ? m : p.M - 1;
      f32x4 v[NJ];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) v[jj] = acc[i][jj] + bj[jj];
      if (p.row_add) {
        const float* rp = p.row_add + (mc / p.rows_per_group) * p.ldra;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) v[jj] += *(const f32x4*)(rp + fj[jj]);
      }
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = v[jj];
      const bool row_ok = m < p.M;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int64_t f = n0 + wn * WN + 16 * jj + 4 * fg;
        if (!row_ok || f >= p.N) continue;
        if (p.out_f32) *(f32x4*)(p.out_f32 + m * p.ldo32 + f) = v[jj];
        if (p.out_f16) {
          half4_t h = {(half_t)v[jj][0], (half_t)v[jj][1], (half_t)v[jj][2], (half_t)v[jj][3]};
          *(half4_t*)(p.out_f16 + m * p.ldo16 + f) = h;
        }
      }
    }
    if (p.ch_stats != nullptr) {
      const int64_t mw = m0 + wm * WM;
      float* const sp2 = p.ch_stats + (mw >> 6) * 2 * p.N;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, qsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x4 vm = mw + 16 * i + fr < p.M ? acc[i][jj] : f32x4{0.f, 0.f, 0.f, 0.f};
          ssum += vm;
          qsum += vm * acc[i][jj];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ssum[r] = row16_sum_sk(ssum[r]);
          qsum[r] = row16_sum_sk(qsum[r]);
        }
        const int64_t f = n0 + wn * WN + 16 * jj + 4 * fg;
        if (fr == 0 && mw < p.M && f < p.N) {
          *(f32x4*)(sp2 + f) = ssum;
          *(f32x4*)(sp2 + p.N + f) = qsum;
        }
      }
    }
  };

  // ---- segments: the exported one first, the imported one last ----
  if (t0 == t1) {
    run(t0, k0, k1, k0 > 0, k1 < nk);
    return;
  }
  if (k1 < nk) run(t1, 0, k1, false, true);
  const int full_lo = k0 > 0 ? t0 + 1 : t0, full_hi = k1 < nk ? t1 - 1 : t1;
  for (int t = full_lo; t <= full_hi; ++t) run(t, 0, nk, false, false);
  if (k0 > 0) run(t0, k0, nk, true, false);
}

template <int BN, int MODE>
int launch_sk(const GemmArgs& a, int nblocks, hipStream_t s) {
  constexpr int lds = 2 * (128 + BN) * 128;
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t dev_bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_relaxed) & dev_bit)) {
    (void)hipFuncSetAttribute((const void*)gemm_sk_kernel<BN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_devs.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((gemm_sk_kernel<BN, MODE>), dim3((unsigned)nblocks), dim3(256), lds, s, a);
  return seva_check_launch("gemm_sk_kernel");
}

}  // namespace

// a: fully populated arguments (tiles_m / tiles_n for 128 x bn tiles, sk_ws set); mode 0 plain, 1 conv3x3
int seva_gemm_streamk_launch(const GemmArgs& a, int mode, int bn, int nblocks, hipStream_t s) {
  if (mode == 0) return bn == 160 ? launch_sk<160, 0>(a, nblocks, s) : launch_sk<128, 0>(a, nblocks, s);
  return bn == 160 ? launch_sk<160, 1>(a, nblocks, s) : launch_sk<128, 1>(a, nblocks, s);
}
