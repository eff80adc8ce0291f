// 3x3 / stride 1 / pad 1 convolution as implicit GEMM with the INPUT WINDOW of the output tile staged in LDS
// (reference: seva/modules/layers.py:101,113 -- the two 3x3 convs of every ResBlock; seva/model.py:57).
//
// gemm.hip's conv mode gathers one A tile per (tap, 64-channel slab): every input pixel of a tile crosses L2 -> LDS nine
// times (PMC, round 3: 4.1x the algorithmic bytes), and the LDS-DMA stream is what bounds that main loop.  Here the
// reduction runs slab-outer / tap-inner:
//
//   for each 64-channel slab s:   window = input pixels of the tile + halo, 128 B per pixel, staged ONCE
//     for each tap (ky, kx):      A fragments = window rows shifted by ky * (W + 2) + kx;  weight tile (tap, s) streamed
//
// The window is a contiguous range of a PADDED pixel index space: every image row gets ONE frame cell in front (the right
// frame of row r is the left frame of row r + 1), every image one frame row on top (the bottom frame of image i is the top
// frame of image i + 1), images stacked: with Wp = W + 1 and S = (H + 1) Wp, output pixel m = (img, y, x) sits at
// P(m) = img * S + (y + 1) Wp + (x + 1) and its tap (ky, kx) at P(m) + (ky - 1) Wp + (kx - 1), for every pixel of every image --
// border taps land on frame cells, which the fill routes to the zero page.  A tile of BM consecutive output pixels needs
// BM + (row crossings) + (W + 2 per image crossing) + 2 (W + 2) + 1 window pixels whatever its alignment, and the tap shift is
// one workgroup-uniform scalar.  Tiles are consecutive pixels of the whole batch where that fits the window capacity (36x36
// and smaller at 160 rows), consecutive pixels of ONE image otherwise (72x72).  Per slab and 160 x 160 tile the LDS-DMA traffic drops from
// 9 * (160 + 160) * 128 B = 360 KiB to 316 * 128 B + 9 * 160 * 128 B = 220 KiB; the weight tile is now the main stream.
//
// Everything else is gemm.hip's core: K-tile 64, 128-byte LDS rows, lane-linear LDS-DMA with the chunk swizzle
// (chunk c of row r at c ^ ((r >> 1) & 7)) applied on the per-lane SOURCE address and on the fragment reads, weight
// fragment as the MFMA A operand (features on the lane), fp32 residual loaded straight into the accumulators, GroupNorm
// statistics from the epilogue.  The window's swizzle key is the WINDOW pixel index, so a tap shift that is not a multiple
// of 16 leaves some 2-way bank conflicts on the A fragment reads (measured: see DESIGN.md section 4).
//
// Wide images (the VAE decoder's 144 .. 576 px rows: a linear window of BM pixels + two image rows does not fit LDS): 2-D tiles of
// 16 output columns x BM / 16 output rows (TW = 16).  The window is the tile's (rows + 2) x 18 source pixels stored row after row
// (row pitch 18), MFMA block i of the tile is its row i, the tap shift is ky * 18 + kx -- the same uniform scalar -- and the halo costs
// 27 % more pixels than the tile has whatever the image width.  GroupNorm statistics blocks are then 4 tile rows x 16 pixels (any
// partition of an image's pixels into 64-pixel blocks serves the consumer, which sums the blocks of an image).
//
// NW = 4: two workgroups per CU (the window single-buffered: 320 px + two 160-row weight stages = 80 KiB each).
// NW = 8: one workgroup per CU on a 256-row tile, window double-buffered and refilled piece by piece under the taps.
#include "gemm_common.h"

#include <atomic>
#include <type_traits>

namespace {

constexpr int BK = 64;  // fp16 elements per K-tile -> 128-byte LDS rows

struct ConvWinGeom {
  uint32_t mul_hw, mul_iw, mul_sp, mul_wp;  // floor(2^32 / d) + 1: x / d == mulhi(x, mul) over the launch's range of x (host-checked)
  int32_t Wp, Sp, hw;                       // hw = OUTPUT pixels per image; mul_iw divides by the OUTPUT width ow
  int32_t ow;
  int32_t tiles_m, tiles_n;
  int32_t tpi;  // 0: M-tiles are consecutive BM-pixel ranges of the whole batch; > 0: tiles per image (a tile never leaves its image)
  int32_t lin_ok;  // host only: the multiply-high divisions of the linear tiles are exact for this launch
};

// UP: the conv input is the nearest-2x upsampled image (reference layers.py:35-46: F.interpolate(scale_factor=2) then conv): the window
// is staged from the SOURCE image and tap (ky, kx) of output pixel (y, x) reads source pixel ((y + ky - 1) >> 1, (x + kx - 1) >> 1)
// (A third weight stage with a counted vmcnt -- tap g + 2 issued under tap g, this tap's own DMA left in flight across the barrier -- was
// built and measured on the 8-wave family: 3 - 10 % SLOWER on every shape, profiles/r04_kconvwin_variants.log; removed.)
// FP8 (BASELINE config 5; gemm.hip's e4m3 scheme): a 128-byte window / weight row is 128 e4m3 channels instead of 64 f16 ones (cin, K and the
// pointers count 2-byte units, so every address here is unchanged); the two 16-byte fragment reads of a (tap, slab) form ONE 32-byte operand of
// v_mfma_scale_f32_16x16x128_f8f6f4, the per-output-channel power-of-two weight scale rides as the E8M0 block scale of the weight operand.
template <int BM, int BN, int NW, int WCAP, bool DBW, bool STATS, bool UP = false, int TW = 0, bool FP8 = false>
__global__ __launch_bounds__(64 * NW, 2) void conv_win_kernel(GemmArgs p, ConvWinGeom g) {
  constexpr bool T2D = TW > 0;
  static_assert(TW == 0 || TW == 16, "2-D tiles are 16 output columns wide: an MFMA block is a tile row");
  constexpr int TH = BM / 16;                                  // output rows of a 2-D tile
  constexpr int SW2 = UP ? 8 : 16, SH2 = UP ? TH / 2 : TH;     // its source extent; window = (SH2 + 2) x (SW2 + 2) pixels
  constexpr int PITCH2 = SW2 + 2, WL2 = (SH2 + 2) * PITCH2;
  static_assert(!T2D || (WL2 <= WCAP && TH % 2 == 0), "2-D window capacity");
  constexpr int WMW = NW / 2, WNW = 2;
  constexpr int WM = BM / WMW, WN = BN / WNW;  // per-wave tile
  constexpr int MI = WM / 16, NJ = WN / 16;
  static_assert(WM % 16 == 0 && WN % 16 == 0, "per-wave tile must be whole MFMA blocks");
  static_assert(!STATS || WM == 64, "statistics: a wave owns a 64-row block");
  constexpr int NPC = WCAP / 8, PPW = (NPC + NW - 1) / NW;  // window pieces (8 pixels x 128 B), per wave
  constexpr int NBP = BN / 8, BPW = (NBP + NW - 1) / NW;    // weight pieces per stage, per wave
  constexpr int WIN_BYTES = WCAP * 128, B_BYTES = BN * 128;
  constexpr int NWB = DBW ? 2 : 1;
  static_assert(WCAP % 8 == 0, "window capacity in whole pieces");
  static_assert(!DBW || PPW <= 8, "double-buffered window: one piece per wave and tap");

  extern __shared__ __attribute__((aligned(16))) char smem[];  // [window x NWB][weight stage 0][weight stage 1]
  const char* const lds_win = smem;
  const char* const lds_b = smem + NWB * WIN_BYTES;
  const unsigned lds_base_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int sr = lane >> 3, sp = lane & 7;
  const int fr = lane & 15, fg = lane >> 4;

  const int work = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = work / g.tiles_n, tn = work - tm * g.tiles_n;  // sibling N-tiles of an M-tile are neighbours on one XCD
  // rows [m0, m_end) of the [M][N] output belong to this tile (2-D tiles: BM rows of ONE image that are not consecutive; see row_m)
  uint32_t m0, m_end;
  uint32_t t_img = 0, t_y0 = 0, t_x0 = 0, t_k = 0;  // 2-D: image, output origin of the tile, tile number inside the image
  if constexpr (T2D) {
    t_img = (uint32_t)tm / (uint32_t)g.tpi;
    t_k = (uint32_t)tm - t_img * (uint32_t)g.tpi;
    const uint32_t tpr = (uint32_t)g.ow / 16u, ty = t_k / tpr;
    t_y0 = ty * TH;
    t_x0 = (t_k - ty * tpr) * 16u;
    m0 = t_img * (uint32_t)g.hw;
    m_end = m0 + (uint32_t)g.hw;
  } else if (g.tpi > 0) {
    const uint32_t img = (uint32_t)tm / (uint32_t)g.tpi, k = (uint32_t)tm - img * (uint32_t)g.tpi;
    m0 = img * (uint32_t)g.hw + k * BM;
    m_end = m0 + BM < (img + 1) * (uint32_t)g.hw ? m0 + BM : (img + 1) * (uint32_t)g.hw;
  } else {
    m0 = (uint32_t)tm * BM;
    m_end = m0 + BM < (uint32_t)p.M ? m0 + BM : (uint32_t)p.M;
  }
  const int64_t n0 = (int64_t)tn * BN;
  // output row of MFMA block i (of this wave) and lane fr; linear tiles: may lie past the tile (m >= m_end: clamp for loads, no store)
  auto row_m = [&](int i) -> uint32_t {
    if constexpr (T2D) return m0 + (t_y0 + (uint32_t)(wm * (WM / 16) + i)) * (uint32_t)g.ow + t_x0 + (uint32_t)fr;
    else return m0 + (uint32_t)(wm * WM + 16 * i + fr);
  };
  const int pitch = T2D ? PITCH2 : g.Wp;  // window row pitch

  // padded index of the source pixel under the CENTRE tap of output pixel m (UP: of the pixel it is upsampled from); xo = its column
  auto pad_index = [&](uint32_t m, uint32_t& yo, uint32_t& xo) -> uint32_t {
    const uint32_t img = __umulhi(m, g.mul_hw);
    const uint32_t rem = m - img * (uint32_t)g.hw;
    yo = __umulhi(rem, g.mul_iw);
    xo = rem - yo * (uint32_t)g.ow;
    const uint32_t y = UP ? yo >> 1 : yo, x = UP ? xo >> 1 : xo;
    return img * (uint32_t)g.Sp + (y + 1) * (uint32_t)g.Wp + x + 1;
  };
  uint32_t q0 = 0;
  int WL = WL2;
  if constexpr (!T2D) {
    uint32_t y_a, x_a, y_b, x_b;
    const uint32_t P0 = pad_index(m0, y_a, x_a), P1 = pad_index(m_end - 1, y_b, x_b);
    // window = [first source pixel any tap reads, last one]: plain conv: consecutive pixels, one halo of Wp + 1 on either side;
    // UP: output rows 2r and 2r + 1 read the same source row, so the window holds whole source rows (first row's start .. last row's end)
    q0 = (UP ? P0 - (x_a >> 1) : P0) - (uint32_t)(g.Wp + 1);   // padded index of window pixel 0
    WL = (int)((UP ? P1 - (x_b >> 1) + (uint32_t)p.iw - 1 : P1) - q0) + g.Wp + 2;  // window pixels this tile reads (<= WCAP)
  }
  const int npc = __builtin_amdgcn_readfirstlane((WL + 7) >> 3);

  // ---- window fill: lane (pixel 8 pc + sr, physical chunk sp) of piece pc fetches logical chunk sp ^ key(pixel) ----
  int woff[PPW];  // byte offset of the lane's 16 bytes in slab 0, or -1: frame pixel / past the batch -> zero page
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = i * NW + wave;
    const int j = 8 * pc + sr;
    const int chunk = sp ^ ((j >> 1) & 7);
    bool ok;
    uint32_t pix;
    if constexpr (T2D) {  // window pixel j = (wy, wx) of the (SH2 + 2) x PITCH2 block around the tile's source pixels
      const int wy = j / PITCH2, wx = j - wy * PITCH2;
      const int sy = (int)(UP ? t_y0 >> 1 : t_y0) - 1 + wy, sx = (int)(UP ? t_x0 >> 1 : t_x0) - 1 + wx;
      ok = (j < WL2) & (sy >= 0) & (sy < p.ih) & (sx >= 0) & (sx < p.iw);
      pix = (uint32_t)sy * (uint32_t)p.iw + (uint32_t)sx;  // inside the tile's image: its base is added as a 64-bit scalar (a_img)
    } else {
      const uint32_t q = q0 + (uint32_t)j;
      const uint32_t img = __umulhi(q, g.mul_sp);
      const uint32_t rem = q - img * (uint32_t)g.Sp;
      const uint32_t py = __umulhi(rem, g.mul_wp);
      const uint32_t px = rem - py * (uint32_t)g.Wp;
      ok = (img < (uint32_t)p.n) & (py >= 1u) & (px >= 1u) & (j < WL);  // row 0 / column 0 of an image block are frame cells
      pix = (img * (uint32_t)p.ih + (py - 1)) * (uint32_t)p.iw + (px - 1);
    }
    woff[i] = ok ? (int)((pix * (uint32_t)p.cin + (uint32_t)chunk * 8u) * 2u) : -1;
  }
  // 2-D tiles: offsets are relative to the tile's image, so nothing here depends on the batch size (whether this kernel runs must be a
  // function of per-sample dimensions only: per-frame results are then identical whatever the number of frames per pass)
  const char* const a_img = (const char*)p.a + (T2D ? (int64_t)t_img * p.ih * p.iw * p.cin * 2 : (int64_t)0);
  auto fill_piece = [&](int i, int s, int wb) {  // i: compile-time after unrolling
    const int pc = i * NW + wave;
    if (pc < npc) {  // wave-uniform
      const char* const src = woff[i] >= 0 ? a_img + (int64_t)s * 128 + woff[i] : (const char*)g_zero_page;
      glds16_raw(src, lds_base_u32 + wb * WIN_BYTES + pc * 1024);
    }
  };
  auto fill_window = [&](int s, int wb) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) fill_piece(i, s, wb);
  };

  // ---- weight stage: BN rows x 128 B of K-tile (tap t, slab s) = columns t * cin + 64 s ----
  const half_t* b_ptr[BPW];
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int row = 8 * (i * NW + wave) + sr;
    const int q = sp ^ ((row >> 1) & 7);
    int64_t n = n0 + row;
    if (n >= p.N) n = p.N - 1;
    b_ptr[i] = p.w + n * p.K + q * 8;
  }
  auto stage_w = [&](int buf, int kcol) {
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const int pc = i * NW + wave;
      if (NBP % NW == 0 || pc < NBP) glds16_raw(b_ptr[i] + kcol, lds_base_u32 + NWB * WIN_BYTES + buf * B_BYTES + pc * 1024);
    }
  };

  // ---- fragment addresses ----
  int b_off[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int rb = wn * WN + fr;
    b_off[s2] = rb * 128 + (((4 * s2 + fg) ^ ((rb >> 1) & 7)) << 4);
  }
  // window pixel of row (16 i + fr) of the wave's tile: plain conv at tap (0, 0), the tap shift ky * Wp + kx is a uniform scalar;
  // UP: at ky = 0 / 1 / 2 with kx = 1, plus the pixel's column parity (the kx shift is (xpar - 1, 0, xpar))
  int a_base[MI], a_rm[UP ? MI : 1], a_rp[UP ? MI : 1], a_xp[UP ? MI : 1];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int ctr, yp, xp;  // window pixel under the centre tap; (UP) row / column parity of the output pixel
    if constexpr (T2D) {
      const int row = wm * (WM / 16) + i;  // tile row = MFMA block
      ctr = UP ? ((row >> 1) + 1) * PITCH2 + (fr >> 1) + 1 : (row + 1) * PITCH2 + fr + 1;
      yp = row & 1;  // the tile's origin is even in both directions
      xp = fr & 1;
    } else {
      uint32_t m = row_m(i), yo, xo;
      if (m >= m_end) m = m_end - 1;
      ctr = (int)(pad_index(m, yo, xo) - q0);
      yp = (int)(yo & 1);
      xp = (int)(xo & 1);
    }
    if constexpr (UP) {
      a_base[i] = ctr;
      a_rm[i] = ctr + (yp - 1) * pitch;
      a_rp[i] = ctr + yp * pitch;
      a_xp[i] = xp;
    } else {
      a_base[i] = ctr - pitch - 1;
    }
  }

  const int nslab = p.cin / BK;
  fill_window(0, 0);
  stage_w(0, 0);
  // FP8: E8M0 scale bytes of this lane's weight rows (MFMA row fr of block j), four blocks per word (plain byte loads, retired below)
  constexpr int NSC = (NJ + 3) / 4;
  int wsc[FP8 ? NSC : 1];
  if constexpr (FP8) {
#pragma unroll
    for (int w = 0; w < NSC; ++w) wsc[w] = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int64_t n = n0 + wn * WN + 16 * j + fr;
      if (n >= p.N) n = p.N - 1;
      wsc[j >> 2] |= (int)p.w_exp[n] << (8 * (j & 3));
    }
  }

  f32x4 acc[MI][NJ];
  if (p.residual) {  // the residual tile goes straight into the accumulators (clamped addresses; stores are guarded)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int64_t m = row_m(i);
      if (m >= m_end) m = m_end - 1;
      const float* rp = p.residual + m * p.ldr;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int64_t f = n0 + wn * WN + 16 * j + 4 * fg;
        if (f > p.N - 4) f = p.N - 4;
        acc[i][j] = *(const f32x4*)(rp + f);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // retire the residual loads for hipcc's wait bookkeeping HERE: a load still "possibly pending" at the loop header gets a literal
  // vmcnt in front of its first use in every iteration, and that literal would also drain the LDS-DMA the compiler cannot see
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[i][j]));
  if constexpr (FP8) {
#pragma unroll
    for (int w = 0; w < NSC; ++w) asm volatile("" : "+v"(wsc[w]));
  }

  int cur = 0;  // weight stage of the K-tile being computed
  for (int s = 0; s < nslab; ++s) {
    const char* const win = lds_win + (DBW ? (s & 1) * WIN_BYTES : 0);
    const bool more = s + 1 < nslab;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t < 8) stage_w(cur ^ 1, (t + 1) * p.cin + BK * s);
      else if (more) stage_w(cur ^ 1, BK * (s + 1));
      if constexpr (DBW) {  // the next slab's window, one piece per wave and tap, into the other buffer
        if (more && t < PPW) fill_piece(t, s + 1, (s + 1) & 1);
      }
      const char* const tb = lds_b + cur * B_BYTES;
      int toff = UP ? 0 : (t / 3) * pitch + (t % 3);
      asm volatile("" : "+s"(toff));  // opaque: the nine taps' fragment addresses are formed here, not hoisted out of the slab loop (45 registers)
      // fragment reads + MFMAs of the tap.  FIRST: every fragment read is ISSUED before the first MFMA (hipcc otherwise re-uses one
      // register quad for the second k-step's window fragments and waits for each read right in front of the five MFMAs that need it:
      // ~100 exposed cycles four times per tap); the MFMAs then start behind counted lgkmcnt waits as the fragments arrive in order.
      // Measured (profiles/r04_kconvwin_frags_first.log): +1.5 ... 9 % on the 4-wave family (two independent workgroups per CU), -8 ... 10 %
      // on the 8-wave family, whose waves would all burst their 18 reads right behind the shared barrier: it keeps hipcc's interleaving.
      const auto compute_tap = [&](auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        half8_t af[2][MI], bf[2][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          int j;
          if constexpr (UP) {
            const int ky = t / 3, kx = t % 3;  // compile-time after unrolling
            j = (ky == 0 ? a_rm[i] : ky == 1 ? a_base[i] : a_rp[i]) + (kx == 0 ? a_xp[i] - 1 : kx == 1 ? 0 : a_xp[i]) + toff;
          } else {
            j = a_base[i] + toff;
          }
          const int addr = j * 128 + ((fg ^ ((j >> 1) & 7)) << 4);
          af[0][i] = *(const half8_t*)(win + addr);
          af[1][i] = *(const half8_t*)(win + (addr ^ 64));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < NJ; ++j) bf[s2][j] = *(const half8_t*)(tb + b_off[s2] + j * 2048);
        if constexpr (FIRST) __builtin_amdgcn_sched_barrier(0);
        if constexpr (FP8) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_f8(j, bf[0][j], bf[1][j], af[0][i], af[1][i], acc[i][j], wsc[j >> 2]);
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
        }
      };
      if constexpr (NW == 4 && !FP8) {
        compute_tap(std::true_type{});
      } else {  // (giving the two wave groups of the 8-wave workgroup different orders under a wave-uniform branch spills 139+ registers)
        compute_tap(std::false_type{});
      }
      if constexpr (FP8) {  // pin the tap's MFMAs in front of its barrier (the same pin on the f16 8-wave family: conv class +0.3 ms, not done): hipcc otherwise sinks all nine taps' scaled MFMAs behind the last
                            // barrier of the slab and parks their fragments in scratch (2 KB; seen in the ISA)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[i][j]));
      }
      if constexpr (!DBW) {
        if (t == 8 && more) {  // every wave has read the last tap's fragments: the window is free for the next slab
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          fill_window(s + 1, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      cur ^= 1;
    }
  }

  // ---- epilogue (gemm.hip's plain one): lane holds features f .. f+3 (rows of D) of pixel m (column of D) ----
  f32x4 bj[NJ];
  int fj[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int64_t f = n0 + wn * WN + 16 * j + 4 * fg;
    if (f > p.N - 4) f = p.N - 4;  // clamp loads; stores are guarded below
    fj[j] = (int)f;
    bj[j] = p.bias ? first_read(*(const f32x4*)(p.bias + f)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int64_t m = row_m(i);
    const int64_t mc = m < m_end ? m : m_end - 1;
    f32x4 v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = acc[i][j] + bj[j];
    if (p.row_add) {
      const float* rp = p.row_add + (mc / p.rows_per_group) * p.ldra;
#pragma unroll
      for (int j = 0; j < NJ; ++j) v[j] += first_read(*(const f32x4*)(rp + fj[j]));
    }
    if constexpr (STATS) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = v[j];  // the statistics pass reads the final values
    }
    const bool row_ok = m < m_end;
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
  if constexpr (STATS) {
    if (p.ch_stats != nullptr) {  // per 64-row block and channel: sum and sum of squares (gemm.hip, same association)
      // block number: linear tiles: 64 consecutive rows of the tensor; 2-D tiles: the wave's 4 tile rows x 16 pixels, numbered inside the image
      const int64_t mw = (int64_t)m0 + wm * WM;
      const int64_t blk = T2D ? (int64_t)t_img * (g.hw >> 6) + (int64_t)t_k * (BM / 64) + wm : mw >> 6;
      float* const sp_ = p.ch_stats + blk * 2 * p.N;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, qsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const f32x4 vm = (T2D || mw + 16 * i + fr < m_end) ? acc[i][j] : f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the tile contribute nothing
          ssum += vm;
          qsum += vm * acc[i][j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ssum[r] = row16_sum(ssum[r]);
          qsum[r] = row16_sum(qsum[r]);
        }
        const int64_t f = n0 + wn * WN + 16 * j + 4 * fg;
        if (fr == 0 && (T2D || mw < m_end) && f < p.N) {
          *(f32x4*)(sp_ + f) = ssum;
          *(f32x4*)(sp_ + p.N + f) = qsum;
        }
      }
    }
  }
}

uint32_t magic_u32(uint32_t d) { return (uint32_t)(0x100000000ull / d) + 1u; }

template <int BM, int BN, int NW, int WCAP, bool DBW, bool STATS, bool UP = false, int TW = 0, bool FP8 = false>
int launch_win(const GemmArgs& a, const ConvWinGeom& g0, hipStream_t s) {
  constexpr int lds = (DBW ? 2 : 1) * WCAP * 128 + 2 * BN * 128;
  static_assert(lds <= 160 * 1024, "LDS per workgroup");
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t dev_bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_relaxed) & dev_bit)) {
    (void)hipFuncSetAttribute((const void*)conv_win_kernel<BM, BN, NW, WCAP, DBW, STATS, UP, TW, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_devs.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  ConvWinGeom g = g0;
  g.tiles_n = (int)((a.N + BN - 1) / BN);
  if constexpr (TW > 0) {
    // 2-D tiles: whole tiles only, and (statistics) whole 64-pixel blocks per image
    constexpr int TH = BM / 16;
    if (a.ow % 16 != 0 || a.oh % TH != 0 || g.hw % 64 != 0) return 1;
    g.tpi = (a.oh / TH) * (a.ow / 16);
    g.tiles_m = a.n * g.tpi;
  } else {
  if (!g.lin_ok) return 1;
  // the widest window any tile needs must fit the instantiation's capacity: consecutive pixels of the whole batch if that fits
  // (a tile may then straddle images), else consecutive pixels of one image, else the launch is not for this kernel
  const auto window_len = [&](int64_t ma, int64_t mb) {  // the kernel's WL for output rows [ma, mb]
    const auto idx = [&](int64_t m, int64_t& x) {
      const int64_t img = m / g.hw, rem = m % g.hw, y = rem / g.ow;
      x = rem % g.ow;
      return img * g.Sp + ((UP ? y >> 1 : y) + 1) * g.Wp + (UP ? x >> 1 : x) + 1;
    };
    int64_t xa, xb;
    const int64_t pa = idx(ma, xa), pb = idx(mb, xb);
    const int64_t q0 = (UP ? pa - (xa >> 1) : pa) - (g.Wp + 1);
    return (UP ? pb - (xb >> 1) + a.iw - 1 : pb) - q0 + g.Wp + 2;
  };
  const auto widest = [&](int tpi) {
    const int64_t tiles = tpi > 0 ? (int64_t)tpi : (a.M + BM - 1) / BM;  // per-image tiling: every image has the same windows
    int64_t wl_max = 0;
    for (int64_t t = 0; t < tiles; ++t) {
      const int64_t lim = tpi > 0 ? g.hw : a.M;
      const int64_t ma = t * BM, mb = ma + BM < lim ? ma + BM : lim;
      const int64_t wl = window_len(ma, mb - 1);
      if (wl > wl_max) wl_max = wl;
    }
    return wl_max;
  };
  const int tpi = (g.hw + BM - 1) / BM;
  // (the scan over the tiles of the whole batch is bounded: a window of BM pixels + two rows cannot fit once a row exceeds the capacity)
  if (g.Wp + 1 <= WCAP && (int64_t)a.M / BM <= 65536 && widest(0) <= WCAP) {
    g.tpi = 0;
    g.tiles_m = (int)((a.M + BM - 1) / BM);
  } else if (g.Wp + 1 <= WCAP && widest(tpi) <= WCAP && !(a.ch_stats != nullptr && g.hw % 64 != 0)) {  // (statistics: 64-row blocks of the WHOLE tensor)
    g.tpi = tpi;
    g.tiles_m = a.n * tpi;
  } else {
    return 1;  // not applicable: the caller tries 2-D tiles, then falls back to the per-tap gather
  }
  }
  const int64_t nb = (int64_t)g.tiles_m * g.tiles_n;
  if (nb <= 0 || nb > 0x7fffffff) {
    seva_set_error("conv_win: bad grid %lld", (long long)nb);
    return SEVA_ERR_ARG;
  }
  hipLaunchKernelGGL((conv_win_kernel<BM, BN, NW, WCAP, DBW, STATS, UP, TW, FP8>), dim3((unsigned)nb), dim3(64 * NW), lds, s, a, g);
  return seva_check_launch("conv_win_kernel");
}

}  // namespace

// 0 = launched, 1 = not applicable (the caller uses the per-tap gather of gemm.hip), < 0 = error
int seva_conv_win_launch(const GemmArgs& a, hipStream_t s, bool fp8) {
  const int knob = g_seva_knobs.conv_win;
  if (knob == 0) return 1;
  if (a.stride != 1 || a.pad_lo != 1 || a.a2 != nullptr || a.sk_ws != nullptr) return 1;
  const int up = a.upsample ? 2 : 1;
  if (a.oh != up * a.ih || a.ow != up * a.iw || a.iw < 2 || a.ih < 2) return 1;
  const bool narrow = a.N <= 32 && a.N % 4 == 0;  // the UNet's head (4 channels), the VAE's conv_out
  if (a.cin % 64 != 0 || (a.N % 160 != 0 && a.N % 128 != 0 && !narrow) || a.K != 9LL * a.cin) return 1;
  ConvWinGeom g{};
  g.Wp = a.iw + 1;             // padded SOURCE space (UP: the image before the nearest-2x upsample)
  g.Sp = (a.ih + 1) * g.Wp;
  g.hw = a.oh * a.ow;
  g.ow = a.ow;
  // 31-bit byte offsets into the image; exactness of the multiply-high divisions of the LINEAR tiles (2-D tiles divide by constants
  // only): mulhi(x, floor(2^32 / d) + 1) == x / d for every x with x * e < 2^32, e = (floor(2^32 / d) + 1) * d - 2^32 in (0, d]
  if ((uint64_t)a.ih * a.iw * a.cin * 2 >= (1ull << 31)) return 1;  // one image
  const auto div_exact = [](uint64_t x_max, uint32_t d) {
    const uint64_t e = (uint64_t)magic_u32(d) * d - (1ull << 32);
    return x_max < (1ull << 32) && x_max * e < (1ull << 32);
  };
  const uint64_t q_max = (uint64_t)a.n * g.Sp + 1024;
  g.lin_ok = (uint64_t)a.n * a.ih * a.iw * a.cin * 2 < (1ull << 31) && div_exact((uint64_t)a.M, (uint32_t)g.hw) &&
             div_exact((uint64_t)g.hw, (uint32_t)g.ow) && div_exact(q_max, (uint32_t)g.Sp) && div_exact((uint64_t)g.Sp, (uint32_t)g.Wp);
  g.mul_hw = magic_u32((uint32_t)g.hw);
  g.mul_iw = magic_u32((uint32_t)g.ow);
  g.mul_sp = magic_u32((uint32_t)g.Sp);
  g.mul_wp = magic_u32((uint32_t)g.Wp);
  const bool stats = a.ch_stats != nullptr;
  if (fp8) {
    // e4m3 operands (the C >= 640 levels in fp8 mode): 128-column tiles only (with 160 columns the 8-register operand tuples of the scaled
    // MFMA no longer fit beside 100 accumulators: 2 KB of scratch; gemm.hip's e4m3 kernels found the same); cin counts 2-byte units
    if (a.N % 128 != 0 || a.w_exp == nullptr) return 1;
    const bool eight = knob == 2;  // two 4-wave workgroups per CU are faster on every e4m3 shape of a step (profiles/r04_kconvwin_fp8.log)
    if (a.upsample) return 1;  // (the engine keeps the three upsample convs in f16)
    int rc = eight ? launch_win<256, 128, 8, 416, true, true, false, 0, true>(a, g, s) : launch_win<128, 128, 4, 288, false, true, false, 0, true>(a, g, s);
    if (rc == 1) rc = eight ? launch_win<128, 128, 4, 288, false, true, false, 0, true>(a, g, s) : launch_win<256, 128, 8, 416, true, true, false, 0, true>(a, g, s);
    return rc;
  }
  if (narrow) {
    // a conv with a handful of output channels is bound by reading its input: the per-tap gather reads it nine times (head conv of a step:
    // 346 us), the window once.  32-column tile (one MFMA block per wave column; the upper wave column idles when N <= 16)
    if (stats || a.upsample) return 1;
    int rc = launch_win<160, 32, 4, 320, false, false>(a, g, s);
    if (rc == 1) rc = launch_win<128, 32, 4, 184, false, false, false, 16>(a, g, s);
    return rc;
  }
  if (a.N % 160 != 0) {
    // 128-column family (the VAE's 128 / 256 / 512 channels): linear tiles where the window fits (72 px rows), else 2-D tiles of 16 output
    // columns (144 .. 576 px rows).  Two 4-wave workgroups per CU on 128-row tiles: 2 - 11 % faster than the 8-wave 256-row tile on every
    // decoder shape (profiles/r04_kconvwin_vae.log), which stays behind knob conv_win = 2.
    const bool eight = knob == 2;
    int rc;
    if (a.upsample) {
      rc = eight ? launch_win<256, 128, 8, 416, true, true, true>(a, g, s) : launch_win<128, 128, 4, 288, false, true, true>(a, g, s);
      if (rc == 1) rc = eight ? launch_win<256, 128, 8, 104, true, true, true, 16>(a, g, s) : launch_win<128, 128, 4, 64, false, true, true, 16>(a, g, s);
    } else {
      rc = eight ? launch_win<256, 128, 8, 416, true, true>(a, g, s) : launch_win<128, 128, 4, 288, false, true>(a, g, s);
      if (rc == 1) rc = eight ? launch_win<256, 128, 8, 328, true, true, false, 16>(a, g, s) : launch_win<128, 128, 4, 184, false, true, false, 16>(a, g, s);
    }
    return rc;
  }
  if (a.upsample) {
    // fused nearest-2x upsample (the three Upsample convs of a step): the window over the SOURCE image is small (a quarter of the pixels),
    // the 8-wave 256-row tile always fits; the 4-wave family serves launches too small to fill the CUs with 256-row tiles
    const double t8 = (double)((a.M + 255) / 256) * (double)((a.N + 159) / 160);
    const bool eight = knob == 2 || (knob != 1 && t8 >= 256.0);
    int rc = eight ? launch_win<256, 160, 8, 416, true, true, true>(a, g, s) : launch_win<128, 160, 4, 288, false, true, true>(a, g, s);
    if (rc == 1) rc = eight ? launch_win<128, 160, 4, 288, false, true, true>(a, g, s) : launch_win<256, 160, 8, 416, true, true, true>(a, g, s);
    return rc;
  }
  // Two instantiation families, bitwise equal to each other (same reduction order): two 4-wave workgroups per CU on 160-row tiles
  // (128 with statistics) or one 8-wave workgroup on a 256-row tile with the window double-buffered.  The 8-wave tile moves a third
  // fewer LDS-DMA bytes per FLOP and is ~5 % faster where its tile count fills whole rounds of the 256 CUs; the choice is made from
  // how the launch quantises (measured: 72x72 and 18x18 at batch 42 prefer 8 waves, 36x36 prefers 4: tools/kconvwin.py).  Which of
  // the two RUNS may depend on the batch; whether the window kernel runs at all depends on per-sample dimensions only.
  const auto launch4 = [&]() { return stats ? launch_win<128, 160, 4, 288, false, true>(a, g, s) : launch_win<160, 160, 4, 320, false, false>(a, g, s); };
  const auto launch8 = [&]() { return launch_win<256, 160, 8, 416, true, true>(a, g, s); };
  bool eight;
  if (knob == 1 || knob == 2) {
    eight = knob == 2;
  } else {
    const double tn = (double)((a.N + 159) / 160);
    const double t8 = (double)((a.M + 255) / 256) * tn, t4 = (double)((a.M + (stats ? 127 : 159)) / (stats ? 128 : 160)) * tn;
    const double r8 = t8 / 256.0, r4 = t4 / 512.0;
    const double f4 = r4 - (double)(int64_t)r4;
    const double tail4 = f4 > 0.0 ? (f4 <= 0.5 ? 0.55 : 0.55 + 0.9 * (f4 - 0.5)) : 0.0;  // a partly filled round of 4-wave workgroups runs one per CU
    const double cost8 = 0.95 * (double)(int64_t)(r8 + 0.999999) / r8, cost4 = ((double)(int64_t)r4 + tail4) / r4;
    eight = t8 >= 256.0 && cost8 < cost4;
  }
  int rc = eight ? launch8() : launch4();
  if (rc == 1) rc = eight ? launch4() : launch8();  // the other family may still fit (window capacity is per family)
  return rc;
}
