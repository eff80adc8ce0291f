// 256x256x64 phased GEMM / implicit-GEMM conv for gfx950: two wave groups in anti-phase.
//
// 8 waves (2 M x 4 N), each owning a 128x64 output block = four 64x32 C-quadrants.  A K-tile is
// consumed in 4 phases, one quadrant (16 MFMAs of 16x16x32) per phase:
//     phase 0: Q(a0,b0)   reads A sub-block a0 (8 x ds_read_b128) + B sub-block b0 (4)
//     phase 1: Q(a0,b1)   reads b1 (4)            [a0 stays in registers]
//     phase 2: Q(a1,b1)   reads a1 (8)            [b1 stays]
//     phase 3: Q(a1,b0)   reads b0 (4)            [a1 stays]
// Each phase is  [issue one 16 KB LDS-DMA group of the NEXT K-tile | fragment ds_reads |
// s_waitcnt vmcnt(4)] BAR1 [16 MFMAs] BAR2.  Waves 4-7 run one barrier behind waves 0-3, so on
// every SIMD one wave is in its load segment while the other is in its MFMA segment.
//
// LDS (128 KB): 2 K-tile buffers x 4 groups of 16 KB, grouped by FIRST USE, not by tile half:
//     a0 = tile rows {0-63, 128-191}   a1 = rows {64-127, 192-255}
//     b0 = tile cols {64w+0..31}       b1 = cols {64w+32..63}          (w = 0..3)
// and issued in the order a0', b0', b1', a1' during phases 0..3 of the previous K-tile, so every
// group has >= 3 phases to land before its first reader and `vmcnt(4)` (two younger groups of 2
// LDS-DMA instructions per wave) is the right count in every phase.
//
// Hazards (one barrier = one "event"; group 1's k-th barrier is group 0's (k+1)-th):
//   RAW  a reader takes group X at the top of phase P, i.e. after ITS BAR2(P-1); every wave waited
//        for its own share of X before ITS BAR1(P-1), which is the same or an earlier event for
//        both stagger directions.
//   WAR  a group is overwritten >= 2 phases (4 events) after its last ds_read, whose data the
//        reader consumed (lgkmcnt) before its next barrier; the stagger skews by one event.
// Operand layout, swizzle, MFMA orientation and epilogue contract are those of gemm.hip.
#include "gemm_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int PB_BM = 256, PB_BN = 256, PB_BK = 64;
constexpr int GROUP_BYTES = 16384;                 // 128 rows x 128 B
constexpr int BUF_BYTES = 4 * GROUP_BYTES;         // a0 | a1 | b0 | b1
constexpr int OFF_A0 = 0, OFF_A1 = GROUP_BYTES, OFF_B0 = 2 * GROUP_BYTES, OFF_B1 = 3 * GROUP_BYTES;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void bar_raw() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#define bar()                       \
  do {                              \
    if (!(dbg & 16)) bar_raw();     \
  } while (0)

template <int MODE, int EPI, bool DBGK>
__global__ __launch_bounds__(512, 2) void gemm_phase_kernel(GemmArgs p) {
  const int dbg = DBGK ? p.dbg : 0;  // ablation bits only exist in the DBGK instantiation
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // waves 0-3: rows 0-127 (group 0); 4-7: rows 128-255
  const int fr = lane & 15, fg = lane >> 4;

  const int work = xcd_remap(blockIdx.x, p.tiles_m * p.n_chunks);
  const int tm = work / p.n_chunks, chunk = work - tm * p.n_chunks;
  const int tn_begin = (int)((int64_t)chunk * p.tiles_n / p.n_chunks);
  const int tn_end = (int)((int64_t)(chunk + 1) * p.tiles_n / p.n_chunks);
  const int64_t m0 = (int64_t)tm * PB_BM;
  const int nk = (int)(p.K / PB_BK);
  const int total = (tn_end - tn_begin) * nk;

  // ---- LDS-DMA staging: every group = 16 wave-instructions of 8 rows; wave w issues #2w, #2w+1 ----
  // State is kept as 32-bit BYTE offsets against the uniform base pointers (the launcher checks that
  // every operand is < 4 GiB) to stay inside the 256-VGPR budget next to 128 accumulators.
  const int sr = lane >> 3, sp = lane & 7;
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));
  const char* const a_base = (const char*)p.a;
  const char* const w_base = (const char*)p.w;
  uint32_t a_off32[2][2];   // [ai][i]  MODE 0: row start + swizzled chunk; MODE 1: image start
  int a_yx[2][2];           // MODE 1: (oy*stride-1) << 16 | (ox*stride-1) & 0xffff
  auto q_of = [&](int i) { return sp ^ ((4 * i + (sr >> 1)) & 7); };  // (gr>>1)&7, gr = 16w+8i+sr
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gr = 16 * wave + 8 * i + sr;  // row inside the 128-row group
      const int trow = (gr >> 6) * 128 + 64 * x + (gr & 63);
      int64_t m = m0 + trow;
      if (m >= p.M) m = p.M - 1;
      if (MODE == 0) {
        a_off32[x][i] = (uint32_t)((m * p.lda + q_of(i) * 8) * 2);
        a_yx[x][i] = 0;
      } else {
        const int ohw = p.oh * p.ow;
        const int img = (int)(m / ohw);
        const int rem = (int)(m - (int64_t)img * ohw);
        const int oy = rem / p.ow, ox = rem - oy * p.ow;
        a_yx[x][i] = ((oy * p.stride - 1) << 16) | ((ox * p.stride - 1) & 0xffff);
        a_off32[x][i] = (uint32_t)((int64_t)img * p.ih * p.iw * p.cin * 2);
      }
    }
  uint32_t b_off32[2][2];   // [bj][i] weight row start + swizzled chunk of the tile being ISSUED
  auto set_b_tile = [&](int tn) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int gr = 16 * wave + 8 * i + sr;
        int64_t n = (int64_t)tn * PB_BN + (gr >> 5) * 64 + 32 * x + (gr & 31);
        if (n >= p.N) n = p.N - 1;
        b_off32[x][i] = (uint32_t)((n * p.K + q_of(i) * 8) * 2);
      }
  };

  // issue cursor over the flattened (N-tile, K-tile) sequence: the K-tile whose groups are being staged
  int i_tn = tn_begin, i_kt = 0;
  auto issue_a = [&](int buf, int x) {
    const unsigned dst = smem_base + buf * BUF_BYTES + (x ? OFF_A1 : OFF_A0) + wave * 2048;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16_raw(a_base + (a_off32[x][i] + (uint32_t)(i_kt * (PB_BK * 2))), dst + i * 1024);
    } else {
      const int k0 = i_kt * PB_BK;
      const int tap = k0 / p.cin;
      const int ci0 = k0 - tap * p.cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int eh = p.upsample ? 2 * p.ih : p.ih, ew = p.upsample ? 2 * p.iw : p.iw;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int iy = (a_yx[x][i] >> 16) + ky, ix = (int)(short)(a_yx[x][i] & 0xffff) + kx;
        const bool ok = (iy >= 0) & (iy < eh) & (ix >= 0) & (ix < ew);
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        const uint32_t off = a_off32[x][i] + (uint32_t)(((sy * p.iw + sx) * p.cin + ci0 + q_of(i) * 8) * 2);
        glds16_raw(ok ? (const void*)(a_base + off) : (const void*)g_zero_page, dst + i * 1024);
      }
    }
  };
  auto issue_b = [&](int buf, int x) {
    const unsigned dst = smem_base + buf * BUF_BYTES + (x ? OFF_B1 : OFF_B0) + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16_raw(w_base + (b_off32[x][i] + (uint32_t)(i_kt * (PB_BK * 2))), dst + i * 1024);
  };
  auto advance_issue = [&]() {
    if (++i_kt == nk) {
      i_kt = 0;
      ++i_tn;
      if (i_tn < tn_end) set_b_tile(i_tn);
    }
  };

  // fragment-read offsets inside a group (swizzle term only depends on fr)
  int a_off[2], b_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sw = ((4 * s + fg) ^ ((fr >> 1) & 7)) << 4;
    a_off[s] = (wm * 64 + fr) * 128 + sw;
    b_off[s] = (wn * 32 + fr) * 128 + sw;
  }

  f32x4 acc[2][4][2][2];  // [ai][i][bj][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  half8_t af[4][2], bf[2][2];  // [i][s], [j][s]
  auto read_a = [&](const char* base) {
    if (dbg & 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" : "=v"(af[i][s]));
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) af[i][s] = *(const half8_t*)(base + a_off[s] + i * 2048);
  };
  auto read_b = [&](const char* base) {
    if (dbg & 8) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" : "=v"(bf[j][s]));
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) bf[j][s] = *(const half8_t*)(base + b_off[s] + j * 2048);
  };
  auto mfma_quadrant = [&](auto ai_c, auto bj_c) {
    constexpr int AI = decltype(ai_c)::value, BJ = decltype(bj_c)::value;
    if (dbg & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" ::"v"(af[i][s]));
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" ::"v"(bf[j][s]));
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[AI][i][BJ][j] =
              __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][s], af[i][s], acc[AI][i][BJ][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue: all four groups of K-tile 0 ----
  set_b_tile(tn_begin);
  issue_a(0, 0);
  issue_b(0, 0);
  issue_b(0, 1);
  issue_a(0, 1);
  advance_issue();
  wait_vm<0>();
  bar();
  if (wm == 1) bar();  // stagger: waves 4-7 run one barrier behind

  int c_tn = tn_begin, c_kt = 0;
  auto ktile = [&](auto buf_c, int g) {
    constexpr int BUF = decltype(buf_c)::value;
    const char* const cur = smem + BUF * BUF_BYTES;
    const bool has_next = g + 1 < total && !(dbg & 1);
    // ---- phase 0: Q(a0,b0) ----
    if (has_next) issue_a(BUF ^ 1, 0);
    read_a(cur + OFF_A0);
    read_b(cur + OFF_B0);
    if (has_next) wait_vm<4>(); else wait_vm<0>();
    bar();
    mfma_quadrant(I0{}, I0{});
    bar();
    // ---- phase 1: Q(a0,b1) ----
    if (has_next) issue_b(BUF ^ 1, 0);
    read_b(cur + OFF_B1);
    if (has_next) wait_vm<4>(); else wait_vm<0>();
    bar();
    mfma_quadrant(I0{}, I1{});
    bar();
    // ---- phase 2: Q(a1,b1) ----
    if (has_next) issue_b(BUF ^ 1, 1);
    read_a(cur + OFF_A1);
    if (has_next) wait_vm<4>(); else wait_vm<0>();
    bar();
    mfma_quadrant(I1{}, I1{});
    bar();
    // ---- phase 3: Q(a1,b0) ----
    if (has_next) {
      issue_a(BUF ^ 1, 1);
      advance_issue();
    }
    read_b(cur + OFF_B0);
    if (has_next) wait_vm<4>(); else wait_vm<0>();
    bar();
    mfma_quadrant(I1{}, I0{});
    bar();

    if (++c_kt < nk) return;
    // ---- tile finished: epilogue (lane holds features f..f+3 of token m) ----
    c_kt = 0;
    const int64_t n0 = (int64_t)c_tn * PB_BN;
    ++c_tn;
    // launder the lane id: everything the epilogue derives from it is recomputed here, once per
    // tile, instead of being hoisted out of the K loop and held in (spilled) registers across it
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int fr = lane_e & 15, fg = lane_e >> 4;
    if (EPI == 0) {
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        f32x4 bjv[2];
        int fj[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int64_t f = n0 + wn * 64 + 32 * bj + 16 * j + 4 * fg;
          if (f > p.N - 4) f = p.N - 4;
          fj[j] = (int)f;
          bjv[j] = p.bias ? *(const f32x4*)(p.bias + f) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ai = 0; ai < 2; ++ai)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + wm * 128 + 64 * ai + 16 * i + fr;
            const int64_t mc = m < p.M ? m : p.M - 1;
            f32x4 v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) v[j] = acc[ai][i][bj][j] + bjv[j];
            if (p.row_add) {
              const float* rp = p.row_add + (mc / p.rows_per_group) * p.ldra;
#pragma unroll
              for (int j = 0; j < 2; ++j) v[j] += *(const f32x4*)(rp + fj[j]);
            }
            if (p.residual) {
              const float* rp = p.residual + mc * p.ldr;
#pragma unroll
              for (int j = 0; j < 2; ++j) v[j] += *(const f32x4*)(rp + fj[j]);
            }
            const bool row_ok = m < p.M;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int64_t f = n0 + wn * 64 + 32 * bj + 16 * j + 4 * fg;
              if (!row_ok || f >= p.N) continue;
              if (p.out_f32) *(f32x4*)(p.out_f32 + m * p.ldo32 + f) = v[j];
              if (p.out_f16) {
                half4_t h = {(half_t)v[j][0], (half_t)v[j][1], (half_t)v[j][2], (half_t)v[j][3]};
                *(half4_t*)(p.out_f16 + m * p.ldo16 + f) = h;
              }
            }
          }
      }
    } else {
      // wave's 64 weight rows = [16 v | 16 v | 16 g | 16 g]: bj 0 = values, bj 1 = gates
      f32x4 bv[2], bg[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int64_t fv = n0 + wn * 64 + 16 * j + 4 * fg;
        if (fv > p.N - 36) fv = p.N - 36;
        bv[j] = p.bias ? *(const f32x4*)(p.bias + fv) : f32x4{0.f, 0.f, 0.f, 0.f};
        bg[j] = p.bias ? *(const f32x4*)(p.bias + fv + 32) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ai = 0; ai < 2; ++ai)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t m = m0 + wm * 128 + 64 * ai + 16 * i + fr;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int64_t fv = n0 + wn * 64 + 16 * j + 4 * fg;
            const f32x4 v = acc[ai][i][0][j] + bv[j], gt = acc[ai][i][1][j] + bg[j];
            const int64_t fo = (n0 + wn * 64) / 2 + 16 * j + 4 * fg;
            const f32x4 o = geglu4(v, gt);
            if (m >= p.M || fv >= p.N) continue;
            if (p.out_f32) *(f32x4*)(p.out_f32 + m * p.ldo32 + fo) = o;
            if (p.out_f16) {
              half4_t h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
              *(half4_t*)(p.out_f16 + m * p.ldo16 + fo) = h;
            }
          }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  int g = 0;
  for (; g + 1 < total; g += 2) {
    ktile(I0{}, g);
    ktile(I1{}, g + 1);
  }
  if (g < total) ktile(I0{}, g);
}

template <int MODE, int EPI>
int launch_phase(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 2 * BUF_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_phase_kernel<MODE, EPI, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_phase_kernel<MODE, EPI, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int64_t a_bytes = MODE == 0 ? a.M * a.lda * 2 : (int64_t)a.n * a.ih * a.iw * a.cin * 2;
  if (a_bytes >= (1LL << 32) || a.N * a.K * 2 >= (1LL << 32)) {
    seva_set_error("gemm_phase: operand larger than 4 GiB (32-bit staging offsets)");
    return SEVA_ERR_UNSUPPORTED;
  }
  GemmArgs args = a;
  args.dbg = 0;
  if (g_seva_knobs.gemm_dbg > 0) args.dbg = g_seva_knobs.gemm_dbg;
  args.tiles_m = (int)((a.M + PB_BM - 1) / PB_BM);
  args.tiles_n = (int)((a.N + PB_BN - 1) / PB_BN);
  int chunks = (512 + args.tiles_m - 1) / args.tiles_m;  // one workgroup per CU: aim at >= 2 rounds
  if (chunks < 1) chunks = 1;
  if (chunks > args.tiles_n) chunks = args.tiles_n;
  args.n_chunks = chunks;
  const int64_t nb = (int64_t)args.tiles_m * chunks;
  if (nb <= 0 || nb > 0x7fffffff) {
    seva_set_error("gemm_phase: bad grid %lld", (long long)nb);
    return SEVA_ERR_ARG;
  }
  if (args.dbg)
    hipLaunchKernelGGL((gemm_phase_kernel<MODE, EPI, true>), dim3((unsigned)nb), dim3(512), lds, s, args);
  else
    hipLaunchKernelGGL((gemm_phase_kernel<MODE, EPI, false>), dim3((unsigned)nb), dim3(512), lds, s, args);
  return seva_check_launch("gemm_phase_kernel");
}

}  // namespace

int seva_gemm_phase_launch(const GemmArgs& a, int mode, int epilogue, hipStream_t s) {
  if (a.col_scale_n > 0 || (mode == 1 && a.pad_lo != 1)) {
    seva_set_error("experimental gemm kernels do not implement col_scale / bottom-right-only padding");
    return SEVA_ERR_UNSUPPORTED;
  }
  if (epilogue == 1) return launch_phase<0, 1>(a, s);
  return mode == 0 ? launch_phase<0, 0>(a, s) : launch_phase<1, 0>(a, s);
}
