// Multi-stage ring GEMM / implicit-GEMM conv for gfx950: the large-tile sibling of gemm.hip.
//
// Why a second kernel: the two-stage 128x128 kernel keeps only one 32 KB stage in flight per
// workgroup and needs 15 KB of L2->LDS traffic per MFLOP; per-CU LDS-DMA delivers ~25 (HBM) to ~70
// (L2) GB/s, so that structure tops out near 0.7 PFLOP/s.  Here
//   * 8 waves (512 threads) compute a 256x128 (BK 64, 3 stages, 144 KB LDS) or 256x256 (BK 32,
//     4 stages, 128 KB LDS) tile: 11.4 / 7.6 KB per MFLOP;
//   * the LDS-DMA stream is CONTINUOUS: stage g+STAGES-1 is issued while stage g is consumed, across
//     K-tiles and across the N-tiles a workgroup walks for its M-tile (the flattened (tile, k) sequence
//     never drains), so 96 KB stay in flight per CU;
//   * one raw s_barrier per K-tile; LDS-DMA completion is tracked with COUNTED s_waitcnt vmcnt(N)
//     (never 0 in steady state).  RAW: every wave waits for its own share of stage g, then the
//     barrier makes all shares visible.  WAR: the buffer refilled after the barrier of iteration g is
//     the one read in iteration g-1, which every wave finished before arriving at that barrier.
// Operand layout, swizzles (source-side, involutive), MFMA orientation (weights as the A operand) and
// the epilogue contract are identical to gemm.hip.
#include "gemm_common.h"

namespace {

template <int BK>
__device__ __forceinline__ int swz(int row, int chunk) {
  if (BK == 64) return chunk ^ ((row >> 1) & 7);     // 128-byte rows
  return chunk ^ ((4 - ((row >> 2) & 3)) & 3);       // 64-byte rows
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int BK, int STAGES, int WAVES_M, int WAVES_N, int MODE, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_ring_kernel(GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int ROWB = BK * 2;          // bytes per LDS row
  constexpr int LPR = ROWB / 16;        // lanes (16-byte chunks) per row
  constexpr int RPI = 64 / LPR;         // rows per 1 KiB wave-instruction
  constexpr int A_PASSES = BM / RPI / NW, B_PASSES = BN / RPI / NW;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int G = A_PASSES + B_PASSES;  // LDS-DMA instructions per wave per stage
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MI = WM / 16, NJ = WN / 16, KSTEPS = BK / 32;
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "staging split");
  static_assert(EPI == 0 || NJ == 4, "GEGLU epilogue needs a 64-wide wave tile");
  static_assert(STAGES >= 3 && STAGES <= 4, "ring depth");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;

  const int work = xcd_remap(blockIdx.x, p.tiles_m * p.n_chunks);
  const int tm = work / p.n_chunks, chunk = work - tm * p.n_chunks;
  const int tn_begin = (int)((int64_t)chunk * p.tiles_n / p.n_chunks);
  const int tn_end = (int)((int64_t)(chunk + 1) * p.tiles_n / p.n_chunks);
  const int64_t m0 = (int64_t)tm * BM;
  const int nk = (int)(p.K / BK);
  const int total = (tn_end - tn_begin) * nk;

  // ---- staging state ----
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));
  const int sr = lane / LPR, sp = lane - sr * LPR;
  const half_t* a_ptr[A_PASSES];
  int a_by[A_PASSES], a_bx[A_PASSES], a_q[A_PASSES];
  int64_t a_img[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int row = (wave * A_PASSES + i) * RPI + sr;
    const int q = swz<BK>(row, sp);
    a_q[i] = q;
    int64_t m = m0 + row;
    if (m >= p.M) m = p.M - 1;
    if (MODE == 0) {
      a_ptr[i] = p.a + m * p.lda + q * 8;
      a_by[i] = a_bx[i] = 0;
      a_img[i] = 0;
    } else {
      const int ohw = p.oh * p.ow;
      const int img = (int)(m / ohw);
      const int rem = (int)(m - (int64_t)img * ohw);
      const int oy = rem / p.ow, ox = rem - oy * p.ow;
      a_by[i] = oy * p.stride - 1;
      a_bx[i] = ox * p.stride - 1;
      a_img[i] = (int64_t)img * p.ih * p.iw * p.cin;
      a_ptr[i] = nullptr;
    }
  }
  int b_row[B_PASSES], b_q[B_PASSES];
  const half_t* b_ptr[B_PASSES];  // weight row pointers of the tile currently being ISSUED
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    b_row[i] = (wave * B_PASSES + i) * RPI + sr;
    b_q[i] = swz<BK>(b_row[i], sp);
  }
  auto set_b_tile = [&](int tn) {
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      int64_t n = (int64_t)tn * BN + b_row[i];
      if (n >= p.N) n = p.N - 1;
      b_ptr[i] = p.w + n * p.K + b_q[i] * 8;
    }
  };

  auto stage = [&](int buf, int tn, int kt) {
    const unsigned la = smem_base + buf * STAGE_BYTES + wave * A_PASSES * 1024;
    const unsigned lb = smem_base + buf * STAGE_BYTES + A_BYTES + wave * B_PASSES * 1024;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) glds16_raw(a_ptr[i] + (int64_t)kt * BK, la + i * 1024);
    } else {
      const int k0 = kt * BK;
      const int tap = k0 / p.cin;
      const int ci0 = k0 - tap * p.cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int eh = p.upsample ? 2 * p.ih : p.ih, ew = p.upsample ? 2 * p.iw : p.iw;
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        const int iy = a_by[i] + ky, ix = a_bx[i] + kx;
        const bool ok = (iy >= 0) & (iy < eh) & (ix >= 0) & (ix < ew);
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        const half_t* src = p.a + a_img[i] + ((int64_t)sy * p.iw + sx) * p.cin + ci0 + a_q[i] * 8;
        const void* g = ok ? (const void*)src : (const void*)g_zero_page;
        glds16_raw(g, la + i * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) glds16_raw(b_ptr[i] + (int64_t)kt * BK, lb + i * 1024);
  };

  // issue cursor over the flattened (tile, k) sequence
  int i_tn = tn_begin, i_kt = 0, i_buf = 0;
  set_b_tile(tn_begin);
  auto issue = [&]() {
    stage(i_buf, i_tn, i_kt);
    if (++i_kt == nk) {
      i_kt = 0;
      ++i_tn;
      set_b_tile(i_tn);  // pointer math once per tile, not per stage
    }
    if (++i_buf == STAGES) i_buf = 0;
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < total) issue();

  const int fr = lane & 15, fg = lane >> 4;
  int a_off[KSTEPS], b_off[KSTEPS];  // hoisted fragment-read offsets (swizzle term is the same for rows 16 apart)
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int ra = wm * WM + fr, rb = wn * WN + fr;
    const int q = (BK == 64 ? 4 * s : 0) + fg;
    a_off[s] = ra * ROWB + (swz<BK>(ra, q) << 4);
    b_off[s] = rb * ROWB + (swz<BK>(rb, q) << 4);
  }
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int c_tn = tn_begin, c_kt = 0, c_buf = 0;
  for (int g = 0; g < total; ++g) {
    // stage g must have landed; up to STAGES-2 younger stages stay in flight
    const int younger = total - 1 - g;
    if (STAGES == 4 && younger >= 2) wait_vmcnt<2 * G>();
    else if (younger >= 1) wait_vmcnt<G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (g + STAGES - 1 < total) issue();  // refills the buffer consumed in iteration g-1

    const char* const ta = smem + c_buf * STAGE_BYTES;
    const char* const tb = ta + A_BYTES;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      half8_t af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const half8_t*)(ta + a_off[s] + i * (16 * ROWB));
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = *(const half8_t*)(tb + b_off[s] + j * (16 * ROWB));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (++c_buf == STAGES) c_buf = 0;
    if (++c_kt < nk) continue;

    // ---- tile finished: epilogue (lane holds features f..f+3 of token m), then reset ----
    c_kt = 0;
    const int64_t n0 = (int64_t)c_tn * BN;
    ++c_tn;
    // launder the lane id so epilogue address math is recomputed per tile, not held across the K loop
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int fr = lane_e & 15, fg = lane_e >> 4;
    if (EPI == 0) {
      f32x4 bj[NJ];
      int fj[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int64_t f = n0 + wn * WN + 16 * j + 4 * fg;
        if (f > p.N - 4) f = p.N - 4;  // clamp loads; stores are guarded below
        fj[j] = (int)f;
        bj[j] = p.bias ? *(const f32x4*)(p.bias + f) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wm * WM + 16 * i + fr;
        const int64_t mc = m < p.M ? m : p.M - 1;
        // loads AND arithmetic are unconditional (clamped addresses) so that no load result is
        // left pending across the loop back-edge (hipcc would protect the register reuse with a
        // vmcnt(0) that also drains the LDS-DMA ring); only the stores are guarded
        f32x4 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j] = acc[i][j] + bj[j];
        if (p.row_add) {
          const float* rp = p.row_add + (mc / p.rows_per_group) * p.ldra;
#pragma unroll
          for (int j = 0; j < NJ; ++j) v[j] += *(const f32x4*)(rp + fj[j]);
        }
        if (p.residual) {
          const float* rp = p.residual + mc * p.ldr;
#pragma unroll
          for (int j = 0; j < NJ; ++j) v[j] += *(const f32x4*)(rp + fj[j]);
        }
        const bool row_ok = m < p.M;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int64_t f = n0 + wn * WN + 16 * j + 4 * fg;
          if (!row_ok || f >= p.N) continue;
          if (p.out_f32) *(f32x4*)(p.out_f32 + m * p.ldo32 + f) = v[j];
          if (p.out_f16) {
            half4_t h = {(half_t)v[j][0], (half_t)v[j][1], (half_t)v[j][2], (half_t)v[j][3]};
            *(half4_t*)(p.out_f16 + m * p.ldo16 + f) = h;
          }
        }
      }
    } else {
      f32x4 bv[2], bg[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int64_t fv = n0 + wn * WN + 16 * j + 4 * fg;
        if (fv > p.N - 36) fv = p.N - 36;
        bv[j] = p.bias ? *(const f32x4*)(p.bias + fv) : f32x4{0.f, 0.f, 0.f, 0.f};
        bg[j] = p.bias ? *(const f32x4*)(p.bias + fv + 32) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wm * WM + 16 * i + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t fv = n0 + wn * WN + 16 * j + 4 * fg;
          if (fv >= p.N) continue;
          const f32x4 v = acc[i][j] + bv[j], gt = acc[i][j + 2] + bg[j];
          const int64_t fo = (n0 + wn * WN) / 2 + 16 * j + 4 * fg;
          const f32x4 o = geglu4(v, gt);
          if (p.out_f32) *(f32x4*)(p.out_f32 + m * p.ldo32 + fo) = o;
          if (p.out_f16) {
            half4_t h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            *(half4_t*)(p.out_f16 + m * p.ldo16 + fo) = h;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

template <int BM, int BN, int BK, int STAGES, int WAVES_M, int WAVES_N, int MODE, int EPI>
int launch_ring(const GemmArgs& a, hipStream_t s, int blocks_per_cu = 1) {
  constexpr int lds = STAGES * (BM + BN) * BK * 2;
  static bool attr_set = false;
  auto kern = gemm_ring_kernel<BM, BN, BK, STAGES, WAVES_M, WAVES_N, MODE, EPI>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  GemmArgs args = a;
  args.tiles_m = (int)((a.M + BM - 1) / BM);
  args.tiles_n = (int)((a.N + BN - 1) / BN);
  // one workgroup per CU is resident (LDS): aim at >= 2 rounds of 256 CUs, else keep M-tiles whole
  const int kTargetBlocks = 512 * blocks_per_cu;
  int chunks = (kTargetBlocks + args.tiles_m - 1) / args.tiles_m;
  if (chunks < 1) chunks = 1;
  if (chunks > args.tiles_n) chunks = args.tiles_n;
  args.n_chunks = chunks;
  const int64_t nb = (int64_t)args.tiles_m * chunks;
  if (nb <= 0 || nb > 0x7fffffff) {
    seva_set_error("gemm_ring: bad grid %lld", (long long)nb);
    return SEVA_ERR_ARG;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(WAVES_M * WAVES_N * 64), lds, s, args);
  return seva_check_launch("gemm_ring_kernel");
}

}  // namespace

int seva_gemm_ring_launch(const GemmArgs& a, int mode, int epilogue, int cfg, hipStream_t s) {
  if (a.col_scale_n > 0 || (mode == 1 && a.pad_lo != 1)) {
    seva_set_error("experimental gemm kernels do not implement col_scale / bottom-right-only padding");
    return SEVA_ERR_UNSUPPORTED;
  }
  if (cfg == 3) {  // 128x128x32, 4 stages (64 KB LDS -> 2 workgroups per CU), waves 2x2: 64x64 per wave
    if (epilogue == 1) return launch_ring<128, 128, 32, 4, 2, 2, 0, 1>(a, s, 2);
    return mode == 0 ? launch_ring<128, 128, 32, 4, 2, 2, 0, 0>(a, s, 2)
                     : launch_ring<128, 128, 32, 4, 2, 2, 1, 0>(a, s, 2);
  }
  if (cfg == 2) {  // 256x256x32, 4 stages, waves 2(M) x 4(N): 128x64 per wave
    if (epilogue == 1) return launch_ring<256, 256, 32, 4, 2, 4, 0, 1>(a, s);
    return mode == 0 ? launch_ring<256, 256, 32, 4, 2, 4, 0, 0>(a, s)
                     : launch_ring<256, 256, 32, 4, 2, 4, 1, 0>(a, s);
  }
  // 256x128x64, 3 stages, waves 4(M) x 2(N): 64x64 per wave
  if (epilogue == 1) return launch_ring<256, 128, 64, 3, 4, 2, 0, 1>(a, s);
  return mode == 0 ? launch_ring<256, 128, 64, 3, 4, 2, 0, 0>(a, s)
                   : launch_ring<256, 128, 64, 3, 4, 2, 1, 0>(a, s);
}
