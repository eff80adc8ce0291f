// fp16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   out[m][n] = sum_k A[m][k] * W[n][k] (+bias) (+row_add) (+residual)      (seva_hip.h)
//
// Structure: BMxBN output tile per 256-thread workgroup (4 waves as 2x2), K-tile 64 (128-byte LDS
// rows), both operands staged by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction)
// into a double-buffered LDS ring, one barrier per K-tile.  The LDS image is lane-linear; the
// bank-conflict swizzle (16-byte chunk c of row r sits at chunk c ^ ((r>>1)&7)) is applied on the
// per-lane SOURCE address and again on the ds_read_b128 fragment reads.  MFMA is
// v_mfma_f32_16x16x32_f16 with the WEIGHT fragment as the A operand, so each lane ends up with 4
// consecutive output features of one row and the epilogue (bias, broadcast add, residual, GEGLU)
// uses 8/16-byte vector accesses.
//
// The conv variant gathers the A tile straight from the NHWC image (per-lane source address per
// (ky,kx) tap, zero page for padding, optional stride 2 or nearest-2x upsampled source), so
// im2col / F.interpolate outputs never exist in HBM.
#include "gemm_common.h"

#include <stdlib.h>

#include <atomic>

namespace {

constexpr int BK = 64;  // fp16 elements per K-tile -> 128-byte LDS rows

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DBGK: ablation instantiation (run-time p.dbg bits, SEVA_GEMM_DBG); the production one folds them away
// PAIRED: weight-row -> MFMA-row assignment that gives a lane 8 consecutive features (f16-only outputs, GEGLU)
// ASTAT (K <= 320, f16-only outputs): the A row-panel of the workgroup lives in REGISTERS.  Waves are
// stacked 4x1 (32 rows x the full tile width each), every wave loads its 32 x K fragment set once
// (<= 80 VGPRs) and only the weight tile is staged through LDS: 44 % fewer LDS-DMA bytes per FLOP for
// the ds1 QKV / GEGLU projections, whose time was 15-23 % A re-staging (SEVA_GEMM_DBG=128).
//
// FP8 (BASELINE config 5): both operands are OCP e4m3 bytes and the product runs on the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 (2x the f16 MFMA rate, half the operand bytes per FLOP).  A K-tile is still 128-byte
// LDS rows -- now 128 e4m3 elements -- so staging, swizzle and fragment reads are byte-for-byte those of the f16 kernel
// (pointers and K / lda / cin count 2-byte units); the two 16-byte fragment reads of a K-tile form ONE 32-byte MFMA
// operand (k order inside the tile is permuted identically for both operands, which a dot product does not see).
// The per-output-channel weight scale is a power of two, 2^e[n], handed to the MFMA as the E8M0 block scale of the
// weight operand (every k-block of row n carries 127 + e[n]; measured: tools/micro/mfma_fp8_scale_map2.hip), the
// activation operand has unit scale: the accumulator holds the de-quantised product, so every epilogue
// (residual-in-accumulator, GEGLU, column scale) is the f16 kernel's.
// (mfma_f8: gemm_common.h)

// NW: waves per workgroup.  4 (2 x 2 wave tiles, two workgroups per CU) everywhere in production; the experimental library also
// instantiates 8 (2 x 4 wave tiles, ONE workgroup per CU): two N-sibling 128 x 160 tiles fused so that their A rows are staged once
// (128 x 320: 0.0109 operand bytes per FLOP; same per-wave tile, registers and waves per SIMD as 128 x 160) -- measured slower.
template <int BM, int BN, int MODE, int EPI, bool DBGK, bool PAIRED, bool ASTAT, bool FP8 = false, bool SPLITK = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 8 / NW) void gemm_kernel(GemmArgs p) {
  const int dbg = DBGK ? p.dbg : 0;
  static_assert(NW == 4 || (NW == 8 && !DBGK && !PAIRED && !ASTAT && !FP8 && !SPLITK), "8 waves: plain fp32-output kernels only");
  constexpr int WNW = NW / 2;  // waves along N
  constexpr int WM = ASTAT ? BM / 4 : BM / 2, WN = ASTAT ? BN : BN / WNW;  // per-wave tile
  constexpr int MI = WM / 16, NJ = WN / 16;
  constexpr int A_PASSES = ASTAT ? 0 : BM / (8 * NW), B_PASSES = BN / (8 * NW);  // 8-row wave-instructions per wave
  static_assert(ASTAT || BM % (8 * NW) == 0, "A rows must split evenly over the waves");
  static_assert(BN % (8 * NW) == 0, "B rows must split evenly over the waves");
  constexpr int A_BYTES = ASTAT ? 0 : BM * 128, B_BYTES = BN * 128;
  constexpr int KS_A = 10;  // ASTAT: k-steps of 32 held in registers (K <= 320)
  static_assert(!ASTAT || (PAIRED && MODE == 0 && !DBGK), "ASTAT rides on the ASYNC f16-only schedule");
  static_assert(EPI == 0 || (NJ % 4 == 0 && PAIRED), "GEGLU epilogue needs 64-wide groups, paired rows");
  // Weight-row -> MFMA-row assignment.  D row r of a 16-row block lands in lane group fg = r >> 2, so
  // with the natural order a lane owns 4 consecutive output features per block and the blocks of a
  // lane are 16 features apart: 8-byte f16 stores, 32-byte fragments per token row.  Instead, blocks
  // are taken in PAIRS (2p, 2p+1) over 32 weight rows and MFMA row r of block 2p+e reads tile row
  //     32p + 8*(r>>2) + 4e + (r&3),
  // so lane group fg owns features 32p + 8fg .. 8fg+7 across the pair: one 16-byte f16 store (two
  // adjacent 16-byte f32 stores) per pair, i.e. 64 contiguous f16 bytes per token row and store.  Only
  // the LDS read address changes; an odd last block (NJ = 5) keeps the natural order.  Measured: f16-out
  // GEMMs 4-11 % faster, but f32 outputs 2-9 % SLOWER (each f32 store then writes 16-byte pieces with
  // 16-byte holes), so PAIRED is only instantiated for f16-only outputs and the GEGLU epilogue.
  // The B tile's chunk swizzle key follows the rows read together: pairs region bits {1,3,4} of the
  // wave-relative row, natural region bits {1,2,3}.
  constexpr int NJP = PAIRED ? (NJ & ~1) : 0;  // blocks handled in pairs
  auto b_key = [](int row) {
    const int rw = row % WN;  // wave-relative
    return rw < NJP * 16 ? (((rw >> 1) & 1) | (((rw >> 3) & 3) << 1)) : ((rw >> 1) & 7);
  };
  // first output feature (relative to n0 + wn*WN) of block j for lane group fg
  auto feat_of = [](int j, int fg) { return j < NJP ? 32 * (j >> 1) + 8 * fg + 4 * (j & 1) : 16 * j + 4 * fg; };

  // FP8: weight row (relative to n0 + wn*WN) that MFMA row `r` of block j reads (the fragment-read row of b_frag_off)
  auto wrow_of = [](int j, int r) { return j < NJP ? 32 * (j >> 1) + 4 * (j & 1) + 8 * (r >> 2) + (r & 3) : 16 * j + r; };
  constexpr int NSC = (NJ + 3) / 4;  // packed scale words per lane
  // instantiations that can emit GroupNorm statistics (GemmArgs::ch_stats): a wave owns a 64-row block of the f32 output
  constexpr bool STATS_OK = EPI == 0 && !PAIRED && !ASTAT && !DBGK && WM == 64 && NJ >= 4;
  // order of the fragment reads (see ktile): measured per kernel family, same box (profiles/r04_ab_gemm_frags_first.log): the f16-only /
  // GEGLU kernels gain 2 - 5 % (36x36 GEGLU 419 -> 399 us), the fp32-output 160 x 160 kernels LOSE 5 % with the half measure their registers allow
  constexpr int FRAGS_FIRST = (DBGK || FP8 || ASTAT || !PAIRED) ? 0 : BN == 160 ? 1 : 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [A buf0][A buf1][B buf0][B buf1]
  char* const lds_a = smem;
  char* const lds_b = smem + 2 * A_BYTES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = ASTAT ? wave : wave / WNW, wn = ASTAT ? 0 : wave % WNW;

  // Experiment knob (SEVA_GEMM_STAGGER): pseudo-random start delay to de-phase the workgroups'
  // main loops and epilogues.  Measured: no gain at any quantum, so it lives in the ablation build only.
  if (DBGK && p.stagger > 0) {
    const unsigned h = (blockIdx.x * 2654435761u) >> 28;  // 0..15
    for (unsigned i = 0; i < h * (unsigned)p.stagger; ++i) __builtin_amdgcn_s_sleep(16);  // 1024 clocks each
  }

  // block -> (M-tile, chunk of N-tiles).  A block walks its N-tiles itself: the A row-panel is
  // fetched from HBM by the first tile and re-read from L2 by the others, instead of 8..80 sibling
  // blocks all stalling on the same HBM miss for every K-tile.
  // Split-K over TWO workgroups per tile (GemmArgs::sk_ws; the ds8 convs: 27 x 8 tiles of 128 rows cannot fill 512 workgroup
  // slots, and 64-row tiles move 55 % more LDS-DMA bytes per FLOP with half the MFMA work per barrier).  The first half of
  // the grid takes the UPPER half of K from zero accumulators and exports them raw to the workspace (producers: lowest
  // block ids, dispatched first, so a consumer never waits for a workgroup that has not been dispatched); the second half
  // takes the lower half of K (seeded with the residual as usual), adds the partner's partial and runs the epilogue.
  // Fixed association -> deterministic, and the same for every batch size (the host picks the split from per-sample
  // dimensions only).
  // (its own instantiation, SPLITK: the 128x160 kernels have no registers to spare for it)
  constexpr bool SPLIT_OK = SPLITK;
  static_assert(!SPLITK || (EPI == 0 && !PAIRED && !ASTAT && !DBGK && MODE != 2 && MODE != 3 && BM == 128), "split-K: plain f32 epilogue only");
  const int nk_all = (int)(p.K / BK);
  int sk_half = -1, kb = 0, ke = nk_all;  // this block's K-tile range [kb, ke)
  int bid = blockIdx.x;
  if constexpr (SPLIT_OK) {
    if (p.sk_ws != nullptr) {
      const int nb = p.tiles_m * p.n_chunks;
      sk_half = bid < nb ? 1 : 0;
      if (!sk_half) bid -= nb;
      kb = sk_half ? nk_all / 2 : 0;
      ke = sk_half ? nk_all : nk_all / 2;
    }
  }
  const int work = xcd_remap(bid, p.tiles_m * p.n_chunks);
  const int tm = work / p.n_chunks, chunk = work - tm * p.n_chunks;
  const int tn_begin = (int)((int64_t)chunk * p.tiles_n / p.n_chunks);
  const int tn_end = (int)((int64_t)(chunk + 1) * p.tiles_n / p.n_chunks);
  const int64_t m0 = (int64_t)tm * BM;

  // ---- staging state: lane (r = lane>>3, phys chunk = lane&7) of each 8-row wave-instruction ----
  const int sr = lane >> 3, sp = lane & 7;
  constexpr int AP = A_PASSES > 0 ? A_PASSES : 1;  // (ASTAT stages no A tile)
  // MODE 0 plain GEMM; MODE 1 conv3x3 (fast gather: per-lane base pointer + 9-bit tap-validity mask, tap offsets are
  // workgroup-uniform scalars); MODE 2 conv3x3 with the fused nearest-2x upsample (general per-tap address math);
  // MODE 3 = MODE 1 followed by K-tiles of a second plain operand (seva_gemm_desc.a2: the folded 1x1 skip conv)
  const half_t* a_ptr[AP];         // MODE 0: running source pointer; MODE 1: pixel (oy*stride - pad, ox*stride - pad)
  unsigned a_mask[AP];             // MODE 1: bit (3*ky + kx) set = tap inside the image
  int a_by[AP], a_bx[AP];          // MODE 2: top-left input coords (conv-input space)
  int64_t a_img[AP];               // MODE 2: element offset of image n
  int a_q[AP];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int row = wave * (BM / NW) + 8 * i + sr;  // row inside the tile
    const int q = sp ^ ((row >> 1) & 7);           // logical 16-B chunk this lane fetches
    a_q[i] = q;
    a_mask[i] = 0;
    int64_t m = m0 + row;
    if (m >= p.M) m = p.M - 1;
    if (dbg & 4) m = row;
    if (MODE == 0) {
      a_ptr[i] = p.a + m * p.lda + q * 8;
      a_by[i] = a_bx[i] = 0;
      a_img[i] = 0;
    } else {
      const int ohw = p.oh * p.ow;
      const int img = (int)(m / ohw);
      const int rem = (int)(m - (int64_t)img * ohw);
      const int oy = rem / p.ow, ox = rem - oy * p.ow;
      a_by[i] = oy * p.stride - p.pad_lo;
      a_bx[i] = ox * p.stride - p.pad_lo;
      a_img[i] = (int64_t)img * p.ih * p.iw * p.cin;
      a_ptr[i] = nullptr;
      if (MODE == 1 || MODE == 3) {
        // may point outside the image for border pixels: only dereferenced for taps whose mask bit is set
        a_ptr[i] = p.a + a_img[i] + ((int64_t)a_by[i] * p.iw + a_bx[i]) * p.cin + q * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = a_by[i] + t / 3, ix = a_bx[i] + t % 3;
          if ((iy >= 0) & (iy < p.ih) & (ix >= 0) & (ix < p.iw)) a_mask[i] |= 1u << t;
        }
      }
    }
  }
  const half_t* b_ptr[B_PASSES];
  auto set_b_tile = [&](int tn) {
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int row = wave * (BN / NW) + 8 * i + sr;
      const int q = sp ^ b_key(row);
      int64_t n = (int64_t)tn * BN + row;
      if (n >= p.N) n = p.N - 1;
      if (dbg & 4) n = row;
      b_ptr[i] = p.w + n * p.K + q * 8;
    }
  };

  int st_ky = 0, st_kx = 0, st_ci0 = 0;  // MODE 1: tap / channel position of the next K-tile to stage
  auto stage = [&](int buf, int kt) {
    char* const la = lds_a + buf * A_BYTES + wave * (BM / NW) * 128;
    char* const lb = lds_b + buf * B_BYTES + wave * (BN / NW) * 128;
    if (MODE == 0) {
      if (!(dbg & 128)) {  // ablation bit 128: no A-operand staging (bound for an A-stationary kernel)
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) glds16(a_ptr[i] + (int64_t)kt * BK, la + i * 1024);
      }
    } else {
      // K-tiles are staged in order 0, 1, 2, ...: the (tap row, tap column, channel offset) of the tile being
      // staged is a running scalar state (two integer divisions per K-tile otherwise, emulated on the VALU).
      // block-uniform: a K-tile never straddles taps (cin % 64 == 0)
      if (kt == kb) {  // first K-tile of this block's range (0 unless split-K)
        const int tap = kb * BK / p.cin;
        st_ci0 = kb * BK - tap * p.cin;
        st_ky = tap / 3;
        st_kx = tap - 3 * st_ky;
      }
      const int ky = st_ky, kx = st_kx, ci0 = st_ci0;
      st_ci0 += BK;
      if (st_ci0 == p.cin) {
        st_ci0 = 0;
        if (++st_kx == 3) {
          st_kx = 0;
          ++st_ky;
        }
      }
      if (MODE == 3 && kt >= p.nk1) {
        // MODE 3: the K-tiles behind the nine taps come from a SECOND, plain row-major operand a2 [M][lda2] (the ResBlock's 1x1
        // skip conv folded into its second 3x3 conv: one accumulation, no fp32 round trip of the skip result).  Addresses are
        // formed per K-tile (a handful of VALU for the few extra tiles) so that no pointer array stays live next to the gather's.
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          int64_t m = m0 + wave * (BM / NW) + 8 * i + sr;
          if (m >= p.M) m = p.M - 1;
          glds16(p.a2 + m * p.lda2 + (int64_t)(kt - p.nk1) * BK + a_q[i] * 8, la + i * 1024);
        }
      } else if (MODE == 1 || MODE == 3) {
        const int64_t tap_off = ((int64_t)ky * p.iw + kx) * p.cin + ci0;  // elements, workgroup-uniform
        const unsigned bit = 1u << (3 * ky + kx);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          const void* g = (a_mask[i] & bit) ? (const void*)(a_ptr[i] + tap_off) : (const void*)g_zero_page;
          glds16(g, la + i * 1024);
        }
      } else {
        const int eh = 2 * p.ih, ew = 2 * p.iw;  // MODE 2: taps index the nearest-2x upsampled image
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          const int iy = a_by[i] + ky, ix = a_bx[i] + kx;
          const bool ok = (iy >= 0) & (iy < eh) & (ix >= 0) & (ix < ew);
          const int sy = iy >> 1, sx = ix >> 1;
          const half_t* src = p.a + a_img[i] + ((int64_t)sy * p.iw + sx) * p.cin + ci0 + a_q[i] * 8;
          const void* g = ok ? (const void*)src : (const void*)g_zero_page;
          glds16(g, la + i * 1024);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) glds16(b_ptr[i] + (int64_t)kt * BK, lb + i * 1024);
  };

  // ASYNC schedule (f16-only outputs: fused QKV, GEGLU, 28 ms of a step).  The epilogue's stores used to
  // be followed at once by the next tile's vmcnt(0): the store->ack latency (and, all workgroups
  // storing together, the HBM write burst) was fully exposed: 33-37 % of the ds1 shapes' time
  // (tools/kablate.py bits 64 / 256).  VMEM completes in issue order, so a wait for a load issued
  // AFTER the stores also waits for the stores; therefore BOTH first stages of the next tile are
  // issued BEFORE the epilogue and waited for with counted vmcnt(N) that leaves the S store
  // instructions in flight; the first wait that covers them is the one at the end of K-tile 1.
  // All LDS-DMA comes from inline asm here (hipcc would guard every ds_read behind a builtin DMA with
  // vmcnt(0)), and the bias row travels with stage 0 into a per-wave 1 KiB LDS slot so that no
  // compiler-visible global load (whose literal vmcnt would drain everything) remains in the loop.
  constexpr bool ASYNC = PAIRED && !DBGK && MODE == 0;
  constexpr int G = A_PASSES + B_PASSES;  // LDS-DMA instructions per wave per stage
  constexpr int S_ST = EPI == 0 ? MI * (NJP / 2 + (NJ - NJP)) : MI * (NJ / 4);  // f16 store instructions per interior tile
  const unsigned lds_base_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));
  // bias slots: [parity][wave] x 1 KiB; only ASTAT alternates the parity per tile (it reads the bias lazily
  // in the epilogue, while the next tile's bias is already in flight)
  const unsigned bias_slot_u32 = lds_base_u32 + 2 * (A_BYTES + B_BYTES) + wave * 1024;
  const char* const bias_slot = smem + 2 * (A_BYTES + B_BYTES) + wave * 1024;
  auto bias_par = [&](int tn) { return ASTAT ? ((tn - tn_begin) & 1) * 4096 : 0; };
  // FP8: per-wave 1 KiB slot behind the bias slots for the tile's weight-scale bytes (lane L brings 16 of them)
  constexpr int BIAS_SLOTS = ASTAT ? 8192 : 4096;
  const unsigned wexp_slot_u32 = lds_base_u32 + 2 * (A_BYTES + B_BYTES) + BIAS_SLOTS + wave * 1024;
  const char* const wexp_slot = smem + 2 * (A_BYTES + B_BYTES) + BIAS_SLOTS + wave * 1024;
  auto stage_wexp = [&](int tn) {
    int64_t f = (int64_t)tn * BN + 16 * lane;
    if (f > p.N - 16) f = p.N - 16;  // N % 16 == 0 (host-checked): in-range lanes are never shifted
    glds16_raw(p.w_exp + f, wexp_slot_u32);
  };
  auto stage_async = [&](int buf, int kt) {
    const unsigned la = lds_base_u32 + buf * A_BYTES + wave * (BM / NW) * 128;
    const unsigned lb = lds_base_u32 + 2 * A_BYTES + buf * B_BYTES + wave * (BN / NW) * 128;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) glds16_raw(a_ptr[i] + (int64_t)kt * BK, la + i * 1024);
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) glds16_raw(b_ptr[i] + (int64_t)kt * BK, lb + i * 1024);
  };
  auto stage_bias = [&](int tn) {  // lane L fetches bias[n0 + 4L .. +3] (clamped) into its wave's slot
    int64_t f = (int64_t)tn * BN + 4 * lane;
    if (f > p.N - 4) f = p.N - 4;
    glds16_raw(p.bias + f, bias_slot_u32 + bias_par(tn));
  };
  auto wait_counted = [&](bool stage1_flying, bool stores_flying) {
    if (stage1_flying) {
      if (stores_flying) wait_vm<G + S_ST>();
      else wait_vm<G>();
    } else {
      if (stores_flying) wait_vm<S_ST>();
      else wait_vm<0>();
    }
  };
  auto barrier_raw = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const int fr = lane & 15, fg = lane >> 4;  // fragment row / k-group
  const int nk = (int)(p.K / BK);
  // fragment-read byte offsets, hoisted: rows 16 apart share the swizzle term, so tile i / j of a
  // wave is `base + i*2048` (an immediate), not a recomputed XOR per ds_read
  int a_off[2], b_off[2], b_off_last[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ra = wm * WM + fr;
    a_off[s] = ra * 128 + (((4 * s + fg) ^ ((ra >> 1) & 7)) << 4);
    // paired blocks: block 2p+e adds (32p + 4e) rows, which leaves the key bits untouched
    const int rb = wn * WN + (PAIRED ? 8 * (fr >> 2) + (fr & 3) : fr);
    b_off[s] = rb * 128 + (((4 * s + fg) ^ b_key(rb)) << 4);
    const int rl = wn * WN + 16 * (NJ - 1) + fr;  // odd last block, natural order
    b_off_last[s] = rl * 128 + (((4 * s + fg) ^ b_key(rl)) << 4);
  }
  auto b_frag_off = [&](int s, int j) {
    if (!PAIRED) return b_off[s] + j * (16 * 128);
    return j < NJP ? b_off[s] + (32 * (j >> 1) + 4 * (j & 1)) * 128 : b_off_last[s];
  };

  // ASTAT: this wave's 32 x K A fragments, loaded once (MFMA B-operand layout: lane (fr, fg) holds
  // row fr, k = 32*ks + 8*fg .. +7), then retired for hipcc's wait-count bookkeeping before the loop
  half8_t areg[ASTAT ? MI : 1][ASTAT ? KS_A : 1];
  if constexpr (ASTAT) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int64_t m = m0 + wm * WM + 16 * i + fr;
      if (m >= p.M) m = p.M - 1;
      const half_t* ap = p.a + m * p.lda + 8 * fg;
#pragma unroll
      for (int ks = 0; ks < KS_A; ++ks) {
        const int kk = 32 * ks < (int)p.K ? 32 * ks : 0;  // K < 320: unused k-steps re-read column 0
        areg[i][ks] = *(const half8_t*)(ap + kk);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int ks = 0; ks < KS_A; ++ks) asm volatile("" : "+v"(areg[i][ks]));
  }

  set_b_tile(tn_begin);
  if (ASYNC) {
    if (p.bias) stage_bias(tn_begin);
    if (FP8) stage_wexp(tn_begin);
    stage_async(0, 0);
    if (nk > 1) stage_async(1, 1);
  } else {
    stage(0, kb);
  }
  bool stores_flying = false;  // ASYNC: the previous tile's S_ST stores may still be in flight
  for (int tn = tn_begin; tn < tn_end; ++tn) {
    const int64_t n0 = (int64_t)tn * BN;
    f32x4 acc[MI][NJ];
    // The residual tile is loaded straight INTO the accumulators (the MFMAs add the product on top):
    // its 16 loads per lane are in flight together with this tile's stage-0 DMA and cost no extra
    // registers, instead of four load->wait->store round trips in the epilogue.  Addresses are
    // clamped (M and N tails); the stores are guarded.
    if (EPI == 0 && p.residual && !(dbg & 32) && sk_half != 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int64_t m = m0 + wm * WM + 16 * i + fr;
        if (m >= p.M) m = p.M - 1;
        const float* rp = p.residual + m * p.ldr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          int64_t f = n0 + wn * WN + feat_of(j, fg);
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

    if (ASYNC) {
      wait_counted(nk > 1, stores_flying);  // stage 0 (+ bias) landed; stage 1 and old stores fly on
      barrier_raw();
    } else {
      __syncthreads();  // stage 0 of this tile (and the residual) has landed (vmcnt(0) + barrier)
    }
    // FP8: E8M0 scale bytes of this lane's weight rows, 4 blocks per word.  ASYNC schedule: out of the wave's LDS slot
    // (they rode in with stage 0, like the bias: no compiler-visible global load may sit in that loop); otherwise
    // plain global byte loads.
    int wsc[FP8 ? NSC : 1];
    if constexpr (FP8) {
#pragma unroll
      for (int w = 0; w < NSC; ++w) wsc[w] = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn * WN + wrow_of(j, lane & 15);
        int b;
        if (ASYNC) {
          b = *(const unsigned char*)(wexp_slot + row);
        } else {
          int64_t n = n0 + row;
          if (n >= p.N) n = p.N - 1;
          b = p.w_exp[n];
        }
        wsc[j >> 2] |= b << (8 * (j & 3));
      }
      if (ASYNC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto ktile = [&](int kt) {
      const int cur = (kt - kb) & 1;
      if (ASYNC) {
        if (kt >= 1 && kt + 1 < nk) stage_async(cur ^ 1, kt + 1);  // K-tile 1 was issued a tile ago
      } else {
        if (kt + 1 < ke && !(dbg & 1)) stage(cur ^ 1, kt + 1);
      }
      const char* const ta = lds_a + cur * A_BYTES;
      const char* const tb = lds_b + cur * B_BYTES;
      // all 16 fragment reads of the K-tile are issued first: the second k-step's fragments land
      // while the first k-step's MFMAs execute (the compiler waits with a counted lgkmcnt)
      if constexpr (ASTAT && FP8) {
        half8_t bfr[2][NJ];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < NJ; ++j) bfr[s2][j] = *(const half8_t*)(tb + b_frag_off(s2, j));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = mfma_f8(j, bfr[0][j], bfr[1][j], areg[i][2 * kt], areg[i][2 * kt + 1], acc[i][j], wsc[j >> 2]);
      } else if constexpr (ASTAT) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          half8_t bfr[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) bfr[j] = *(const half8_t*)(tb + b_frag_off(s2, j));
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], areg[i][2 * kt + s2], acc[i][j], 0, 0, 0);
        }
      } else {
      half8_t af[2][MI], bf[2][NJ];
      if (!(dbg & 8) || kt == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < MI; ++i) af[s][i] = *(const half8_t*)(ta + a_off[s] + i * (16 * 128));
#pragma unroll
          for (int j = 0; j < NJ; ++j) bf[s][j] = *(const half8_t*)(tb + b_frag_off(s, j));
          // FRAGS_FIRST (conv_win.hip measured what hipcc's own order costs: it re-uses a register quad for late fragments and waits for each
          // read right in front of the MFMAs that need it): 2 = every fragment read of the K-tile is ISSUED before its first MFMA; 1 = the
          // first k-step's reads are (the tiles that have no registers for both: 20 - 70 dwords of spill otherwise); 0 = hipcc's order
          if (FRAGS_FIRST == 2 ? s == 1 : FRAGS_FIRST == 1 ? s == 0 : false) __builtin_amdgcn_sched_barrier(0);  // (folds after unrolling)
        }
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(af[s][i]));
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("" : "=v"(bf[s][j]));
        }
      }
      if constexpr (FP8) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = mfma_f8(j, bf[0][j], bf[1][j], af[0][i], af[1][i], acc[i][j], wsc[j >> 2]);
        // the K-tile's scaled MFMAs stay in front of its barrier: hipcc otherwise sinks half of them behind it, their fragments with them
        // (conv_win.hip met the extreme form).  Same-box A/B of the fp8 step, two interleaved rounds: GEMM class 36.95 -> 36.14 ms
        // (profiles/r04_ab_fp8_mfma_pin.log); 128 x 160 e4m3 tiles then compile with 13 - 19 dwords of scratch and gain nothing more
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[i][j]));
      } else
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!(dbg & 2)) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[s][j], af[s][i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[s][i]));
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bf[s][j]));
        }
      }
      // the K-tile's MFMAs stay in front of its barrier: left alone, hipcc leaves 15 of the 50 (160 x 160) behind it, their fragments live across
      // the barrier.  Same-box A/B, three interleaved rounds of the step: GEMM class 43.57 -> 43.17 ms (profiles/r04_ab_gemm_mfma_pin.log)
      if constexpr (!PAIRED && !FP8 && !DBGK) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[i][j]));
      }
      }  // !ASTAT
      if (ASYNC) {
        // K-tile kt+1 must have landed.  It was issued before the old stores when kt == 0 (they stay
        // in flight), after them otherwise (vmcnt(0) then covers them, two K-tiles after their issue).
        if (kt + 1 < nk) {
          if (kt == 0) wait_counted(false, stores_flying);
          else wait_vm<0>();
        }
        barrier_raw();
      } else {
        if (!(dbg & 16)) __syncthreads();
      }
    };
    if constexpr (ASTAT) {
      // fully unrolled so that areg[][2*kt+s] is a static register index (a run-time index would
      // send the array to scratch)
#pragma unroll
      for (int kt = 0; kt < KS_A / 2; ++kt)
        if (kt < nk) ktile(kt);
    } else {
      for (int kt = kb; kt < ke; ++kt) ktile(kt);
    }
    // bias of THIS tile out of the LDS slot before the next tile's stage 0 overwrites it
    constexpr int NB = NJ;  // GEGLU: per 64-row group [v e=0, v e=1, g e=0, g e=1]
    f32x4 bias_r[NB];
    if (ASYNC && !ASTAT) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int idx = EPI == 0 ? wn * WN + feat_of(j, fg)
                                 : wn * WN + 64 * (j >> 2) + 8 * fg + 4 * (j & 1) + 32 * ((j >> 1) & 1);
        bias_r[j] = p.bias ? first_read(*(const f32x4*)(bias_slot + idx * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const char* const bias_cur = bias_slot + bias_par(tn);
    auto bias_idx = [&](int j) {  // tile-relative float index of block j's 4 bias values for this lane
      return EPI == 0 ? wn * WN + feat_of(j, fg) : wn * WN + 64 * (j >> 2) + 8 * fg + 4 * (j & 1) + 32 * ((j >> 1) & 1);
    };
    auto bias_at = [&](int j) -> f32x4 {
      if (!p.bias) return f32x4{0.f, 0.f, 0.f, 0.f};
      if (ASTAT) return first_read(*(const f32x4*)(bias_cur + bias_idx(j) * 4));
      if (ASYNC) return bias_r[j];
      // staged-A debug build: straight from global memory (GEGLU rows: value at +0, gate at +32 of each 64)
      int64_t f = n0 + bias_idx(j);
      if (f > p.N - 4) f = p.N - 4;
      return first_read(*(const f32x4*)(p.bias + f));
    };
    // both LDS buffers are free: start the next tile's first stage(s) before the epilogue
    if (tn + 1 < tn_end) {
      set_b_tile(tn + 1);
      if (ASYNC) {
        if (p.bias) stage_bias(tn + 1);
        if (FP8) stage_wexp(tn + 1);
        stage_async(0, 0);
        if (nk > 1) stage_async(1, 1);
      } else {
        stage(0, kb);
      }
    }
    if constexpr (SPLIT_OK) {
      if (sk_half >= 0) {
        // workspace: [16384 flags (int)] [per tile: MI x NJ x 256 lanes x f32x4, lane-linear]
        int* const flag = (int*)p.sk_ws + ((int64_t)tm * p.tiles_n + tn);
        float* const part = p.sk_ws + 16384 + ((int64_t)tm * p.tiles_n + tn) * (BM * BN) + threadIdx.x * 4;
        if (sk_half == 1) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) st_coherent_x4(part + (i * NJ + j) * 1024, acc[i][j]);  // (no L2-wide fence: gemm_common.h)
          publish_coherent();                                 // this thread's stores are visible device-wide ...
          __syncthreads();                                    // ... for every thread of the workgroup ...
          if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ... then the flag
          continue;  // no epilogue: the partner finishes the tile
        }
        if (threadIdx.x == 0) {
          // bounded wait (a producer is always dispatched before its consumer; the bound only guards against a hang if that
          // assumption were ever violated: the tile is then wrong and error slot 16383 counts it)
          int spins = 0;
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < (1 << 21)) {
            __builtin_amdgcn_s_sleep(32);
            ++spins;
          }
          if (spins == (1 << 21)) atomicAdd((int*)p.sk_ws + 16383, 1);
          __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch / graph replay
        }
        __syncthreads();
        {
          f32x4 pt[NJ];
#pragma unroll
          for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) pt[j] = first_read(ld_coherent_x4(part + (i * NJ + j) * 1024));
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] += pt[j];
          }
        }
      }
    }
    // ASYNC bookkeeping: S_ST is exact only for an interior tile (every guarded store executes) with
    // 16-byte stores; anything else falls back to vmcnt(0)-strength waits (a smaller count is always safe)
    stores_flying = ASYNC && m0 + BM <= p.M && n0 + BN <= p.N && (p.ldo16 & 7) == 0 && !p.out_f32 &&
                    !(FP8 && p.out_f8 && p.out_f16) && tn + 1 < tn_end;  // (f8 + f16 together: twice the stores)

    // ---- epilogue: lane holds features f..f+3 (rows of D) of token m (column of D) ----
    if constexpr (EPI == 0 && PAIRED) {
      // f16-only output, a (pair of) block(s) at a time: at most two bias / addend vectors live
      const bool pitch16_ok = (p.ldo16 & 7) == 0;  // 16-byte f16 stores need an 8-element row pitch
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int64_t m = m0 + wm * WM + 16 * i + fr;
        const int64_t mc = m < p.M ? m : p.M - 1;
        const bool row_ok = m < p.M;
        const int64_t ms = (dbg & 256) ? (m & 127) : m;  // ablation bit 256: stores land in an L2-resident region
        const float* const rp = p.row_add ? p.row_add + (mc / p.rows_per_group) * p.ldra : nullptr;
#pragma unroll
        for (int j = 0; j < NJ; j += (j < NJP ? 2 : 1)) {
          const bool pair = j < NJP;
          const int64_t f = n0 + wn * WN + feat_of(j, fg);
          f32x4 v0 = acc[i][j] + bias_at(j);
          f32x4 v1 = pair ? acc[i][pair ? j + 1 : j] + bias_at(pair ? j + 1 : j) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.col_scale_n > 0) {
            if (f < p.col_scale_n) v0 *= p.col_scale;
            if (f + 4 < p.col_scale_n) v1 *= p.col_scale;
          }
          if (rp && !(dbg & 32)) {
            int64_t f0 = f, f1 = f + 4;
            if (f0 > p.N - 4) f0 = p.N - 4;
            if (f1 > p.N - 4) f1 = p.N - 4;
            v0 += first_read(*(const f32x4*)(rp + f0));
            if (pair) v1 += first_read(*(const f32x4*)(rp + f1));
          }
          if (!row_ok || f >= p.N) continue;
          if (dbg & 64) {
            asm volatile("" ::"v"(v0), "v"(v1));
            continue;
          }
          if (pair && f + 8 <= p.N && pitch16_ok) {
            half8_t h = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3],
                         (half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
            *(half8_t*)(p.out_f16 + ms * p.ldo16 + f) = h;
          } else {
            half4_t h0 = {(half_t)v0[0], (half_t)v0[1], (half_t)v0[2], (half_t)v0[3]};
            *(half4_t*)(p.out_f16 + ms * p.ldo16 + f) = h0;
            if (pair && f + 4 < p.N) {
              half4_t h1 = {(half_t)v1[0], (half_t)v1[1], (half_t)v1[2], (half_t)v1[3]};
              *(half4_t*)(p.out_f16 + ms * p.ldo16 + f + 4) = h1;
            }
          }
        }
      }
    } else if constexpr (EPI == 0) {
      // per 16-row block: issue every addend load (bias is hoisted; row_add and residual = 8
      // independent 16-byte loads in flight), then add + store
      f32x4 bj[NJ];
      int fj[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int64_t f = n0 + wn * WN + feat_of(j, fg);
        if (f > p.N - 4) f = p.N - 4;  // clamp loads; stores are guarded below
        fj[j] = (int)f;
        bj[j] = p.bias ? first_read(*(const f32x4*)(p.bias + f)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // GroupNorm statistics of the tensor being written (ch_stats): the wave's 64 rows x WN channels are summed here, while
      // the values are in registers, so that the consuming GroupNorm needs no statistics pass over the fp32 tensor.
      // Per channel (not per group: any grouping / channel concatenation can be formed later) and per 64-ROW BLOCK of the
      // output: blocks are aligned to multiples of 64 rows of the whole tensor, so for images of hw % 64 == 0 pixels a block
      // never straddles two samples and a sample's partial sums are bitwise independent of what it is batched with.
      // It runs as a second pass AFTER the stores, one 16-channel block at a time (8 live sums), over the accumulator
      // registers, which the store pass leaves holding the final values (accumulating all 2 x NJ x 4 sums alongside the
      // store loop, or re-forming the values from bias / row_add in the second pass, spilled the 128x160 kernels).
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
        if (p.col_scale_n > 0) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            if (n0 + wn * WN + feat_of(j, fg) < p.col_scale_n) v[j] *= p.col_scale;
        }
        if (p.row_add && !(dbg & 32)) {
          const float* rp = p.row_add + (mc / p.rows_per_group) * p.ldra;
#pragma unroll
          for (int j = 0; j < NJ; ++j) v[j] += first_read(*(const f32x4*)(rp + fj[j]));
        }
        const bool row_ok = m < p.M;
        if constexpr (STATS_OK) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = v[j];  // (a register rename) the statistics pass reads the final values
        }
        const int64_t ms = (dbg & 256) ? (m & 127) : m;  // ablation bit 256: stores land in an L2-resident region
        const bool pitch16_ok = (p.ldo16 & 7) == 0;  // 16-byte f16 stores need an 8-element row pitch
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int64_t f = n0 + wn * WN + feat_of(j, fg);
          if (!row_ok || f >= p.N) continue;
          if (dbg & 64) {
            asm volatile("" ::"v"(v[j]));
            continue;
          }
          if (p.out_f32) *(f32x4*)(p.out_f32 + ms * p.ldo32 + f) = v[j];
          if (p.out_f16) {
            if (j + 1 < NJP && (j & 1) == 0 && f + 8 <= p.N && pitch16_ok) {
              // both halves of the pair in range: one 16-byte store of 8 consecutive features
              half8_t h = {(half_t)v[j][0],     (half_t)v[j][1],     (half_t)v[j][2],     (half_t)v[j][3],
                           (half_t)v[j + 1][0], (half_t)v[j + 1][1], (half_t)v[j + 1][2], (half_t)v[j + 1][3]};
              *(half8_t*)(p.out_f16 + ms * p.ldo16 + f) = h;
            } else if (j < NJP && (j & 1) == 1 && f + 4 <= p.N && pitch16_ok) {
              // second half of a pair: already written by the 16-byte store above
            } else {
              half4_t h = {(half_t)v[j][0], (half_t)v[j][1], (half_t)v[j][2], (half_t)v[j][3]};
              *(half4_t*)(p.out_f16 + ms * p.ldo16 + f) = h;
            }
          }
        }
      }
      if constexpr (STATS_OK) {
        if (p.ch_stats != nullptr) {
          const int64_t mw = m0 + wm * WM;  // first row of the wave's 64-row block
          float* const sp = p.ch_stats + (mw >> 6) * 2 * p.N;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, qsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              const f32x4 vm = mw + 16 * i + fr < p.M ? acc[i][j] : f32x4{0.f, 0.f, 0.f, 0.f};  // rows past M contribute nothing
              ssum += vm;
              qsum += vm * acc[i][j];
            }
            // over the 16 token rows held by the 16 lanes of a DPP row (lanes 16 fg .. 16 fg + 15): rotate-and-add
            // all-reduce in a fixed association; lane fr == 0 of each row stores its 4 channels
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ssum[r] = row16_sum(ssum[r]);
              qsum[r] = row16_sum(qsum[r]);
            }
            const int64_t f = n0 + wn * WN + feat_of(j, fg);
            if (fr == 0 && mw < p.M && f < p.N) {
              *(f32x4*)(sp + f) = ssum;
              *(f32x4*)(sp + p.N + f) = qsum;
            }
          }
        }
      }
    } else {
      // every 64 weight rows of the wave = [32 v | 32 g] -> 32 output features; with the paired row
      // assignment lane group fg owns value AND gate of features 8fg .. 8fg+7 (blocks 4q,4q+1 = v; 4q+2,4q+3 = g)
#pragma unroll
      for (int gq = 0; gq < NJ / 4; ++gq) {
        const int base = wn * WN + 64 * gq;  // tile-relative first weight row of the group
        f32x4 bv[2], bg[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          bv[e] = bias_at(4 * gq + e);
          bg[e] = bias_at(4 * gq + 2 + e);
        }
        const bool cols_ok = n0 + base < p.N;  // N % 64 == 0: a group's 64 rows are all in or all out
        const int64_t fo = (n0 + base) / 2 + 8 * fg;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int64_t m = m0 + wm * WM + 16 * i + fr;
          const f32x4 o0 = geglu4(acc[i][4 * gq + 0] + bv[0], acc[i][4 * gq + 2] + bg[0]);
          const f32x4 o1 = geglu4(acc[i][4 * gq + 1] + bv[1], acc[i][4 * gq + 3] + bg[1]);
          if (m >= p.M || !cols_ok) continue;
          if (p.out_f32) {
            *(f32x4*)(p.out_f32 + m * p.ldo32 + fo) = o0;
            *(f32x4*)(p.out_f32 + m * p.ldo32 + fo + 4) = o1;
          }
          if (p.out_f16) {
            half8_t h = {(half_t)o0[0], (half_t)o0[1], (half_t)o0[2], (half_t)o0[3],
                         (half_t)o1[0], (half_t)o1[1], (half_t)o1[2], (half_t)o1[3]};
            *(half8_t*)(p.out_f16 + m * p.ldo16 + fo) = h;
          }
          if constexpr (FP8) {
            if (p.out_f8) {  // e4m3 hidden activations, saturating (8 consecutive features = one 8-byte store)
              const int lo = pack_fp8x4(o0[0], o0[1], o0[2], o0[3]);
              const int hi = pack_fp8x4(o1[0], o1[1], o1[2], o1[3]);
              typedef int v2i_t __attribute__((ext_vector_type(2)));
              *(v2i_t*)(p.out_f8 + m * p.ldo8 + fo) = v2i_t{lo, hi};
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int MODE, int EPI, bool PAIRED, bool ASTAT = false, bool FP8 = false, bool SPLITK = false, int NW = 4>
int launch_p(const GemmArgs& a, hipStream_t s) {
  // + bias slots (ASYNC) + weight-scale slots (FP8 ASYNC)
  constexpr int lds = 2 * ((ASTAT ? 0 : BM) + BN) * 128 + (ASTAT ? 8192 : PAIRED ? 4096 : 0) + (FP8 && PAIRED ? 4096 : 0);
  constexpr bool DBG_BUILD = !ASTAT && !FP8 && !SPLITK && NW == 4;  // the ablation instantiation only exists for the staged-A f16 kernels
  // the dynamic-LDS attribute is per device: one bit per device ordinal and instantiation (a second GPU in the
  // same process would otherwise launch 72-80 KB kernels without it)
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t dev_bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_relaxed) & dev_bit)) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, MODE, EPI, false, PAIRED, ASTAT, FP8, SPLITK, NW>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (DBG_BUILD)
      (void)hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, MODE, EPI, true, PAIRED, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_devs.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  GemmArgs args = a;
  args.tiles_m = (int)((a.M + BM - 1) / BM);
  args.tiles_n = (int)((a.N + BN - 1) / BN);
  // Schedule: an M-tile's N-tiles are split over `chunks` sibling workgroups that are adjacent in the
  // XCD-remapped order, i.e. co-resident on one XCD: they stream the same A row-panel through that
  // XCD's L2 at the same time.  One workgroup walking ALL N-tiles (the former default for big M)
  // re-reads an 80..320 KB panel per N-tile while 512 such panels (40..160 MB) compete for 32 MB of
  // L2.  Measured over every shape of a step (tools/ksweep_chunks.py, profiles/r01_ksweep_chunks.log):
  // narrow outputs (<= 5 N-tiles) want one tile per workgroup, wide ones ~4 tiles per workgroup, and
  // the split must be even (2,1,1,1 tiles is 30 % slower than 1,1,1,1,1).
  constexpr int kTargetBlocks = 1024;
  const int base = (kTargetBlocks + args.tiles_m - 1) / args.tiles_m;  // >= two rounds of the chip
  int chunks;
  if (args.tiles_n <= 5) {
    chunks = args.tiles_n;
  } else {
    int per = 0;
    for (int t : {4, 5, 3, 2})
      if (args.tiles_n % t == 0) { per = t; break; }
    chunks = per ? args.tiles_n / per : (args.tiles_n + 3) / 4;
  }
  if (ASTAT) chunks = 1;  // the A panel sits in registers: re-loading it per sibling is pure cost (re-swept: c1 best)
  if (chunks < base) {
    chunks = base;
    for (int c = base; c <= 2 * base && c <= args.tiles_n; ++c)  // nearest even split above `base`
      if (args.tiles_n % c == 0) { chunks = c; break; }
  }
  // knob gemm_chunks (SEVA_GEMM_CHUNKS=n, benchmarking) overrides the heuristic
  if (g_seva_knobs.gemm_chunks > 0) chunks = g_seva_knobs.gemm_chunks;
  if (chunks < 1) chunks = 1;
  if (chunks > args.tiles_n) chunks = args.tiles_n;
  args.n_chunks = chunks;
  args.dbg = g_seva_knobs.gemm_dbg > 0 ? g_seva_knobs.gemm_dbg : 0;
  args.stagger = g_seva_knobs.gemm_stagger > 0 ? g_seva_knobs.gemm_stagger : 0;
  int64_t nb = (int64_t)args.tiles_m * chunks;
  if (SPLITK) nb *= 2;  // split-K: producers (upper half of K) in the first half of the grid, consumers in the second
  if (nb <= 0 || nb > 0x7fffffff) {
    seva_set_error("gemm: bad grid %lld", (long long)nb);
    return SEVA_ERR_ARG;
  }
  if constexpr (DBG_BUILD) {
    if (args.dbg || args.stagger) {
      hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE, EPI, true, PAIRED, false>), dim3((unsigned)nb), dim3(256), lds, s, args);
      return seva_check_launch("gemm_kernel");
    }
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, MODE, EPI, false, PAIRED, ASTAT, FP8, SPLITK, NW>), dim3((unsigned)nb), dim3(64 * NW), lds, s, args);
  return seva_check_launch("gemm_kernel");
}

template <int BM, int BN, int MODE, int EPI, bool FP8 = false>
int launch(const GemmArgs& a, hipStream_t s) {
  // knob gemm_astat = 0 (SEVA_GEMM_ASTAT=0) disables the A-in-registers variant (benchmarking)
  const bool astat_on = g_seva_knobs.gemm_astat != 0;
  const bool dbg_run = g_seva_knobs.gemm_dbg >= 0 || g_seva_knobs.gemm_stagger >= 0;
  const bool half_out = a.out_f16 || (FP8 && a.out_f8);  // 2-byte (or e4m3) outputs only: ASYNC schedule
  // (the A-in-registers variant is f16-only: with both k-steps' fragments live for one 128-deep MFMA it spills)
  if constexpr (EPI == 1) {
    if constexpr (BM == 128 && !FP8) {
      if (astat_on && !dbg_run && a.K <= 320 && half_out && !a.out_f32) return launch_p<BM, BN, MODE, EPI, true, true, FP8>(a, s);
    }
    return launch_p<BM, BN, MODE, EPI, true, false, FP8>(a, s);
  } else {
    if constexpr (MODE == 0 && BN >= 128) {
      if (a.out_f16 && !a.out_f32 && !a.residual) {
        if constexpr (BM == 128 && !FP8) {
          if (astat_on && !dbg_run && a.K <= 320) return launch_p<BM, BN, MODE, EPI, true, true, FP8>(a, s);
        }
        return launch_p<BM, BN, MODE, EPI, true, false, FP8>(a, s);
      }
    }
    return launch_p<BM, BN, MODE, EPI, false, false, FP8>(a, s);
  }
}

}  // namespace

namespace {
template <bool FP8>
int gemm_entry(const seva_gemm_desc* d, seva_stream_t stream) {
  SEVA_REQUIRE(d != nullptr, "gemm: null desc");
  SEVA_REQUIRE(d->a && d->w, "gemm: null operand");
  SEVA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: empty problem M=%lld N=%lld K=%lld",
               (long long)d->M, (long long)d->N, (long long)d->K);
  constexpr int KU = FP8 ? 2 : 1;  // e4m3 elements per 2-byte unit of the kernel's K / lda / cin arithmetic
  SEVA_REQUIRE(d->K % (BK * KU) == 0, "gemm: K=%lld not a multiple of %d", (long long)d->K, BK * KU);
  SEVA_REQUIRE(d->N % 4 == 0, "gemm: N=%lld not a multiple of 4", (long long)d->N);
  SEVA_REQUIRE(d->out_f32 || d->out_f16 || (FP8 && d->out_f8), "gemm: no output");
  if (FP8) {
    SEVA_REQUIRE(d->w_exp != nullptr, "gemm fp8: w_exp (per-channel E8M0 scale bytes) is required");
    SEVA_REQUIRE(d->N % 16 == 0 && d->N > 32, "gemm fp8: N=%lld must be a multiple of 16 and > 32", (long long)d->N);
    SEVA_REQUIRE(!d->upsample, "gemm fp8: the fused-upsample conv stays on the f16 kernel");
    SEVA_REQUIRE(!d->out_f8 || (d->epilogue == 1 && d->ldo8 % 8 == 0 && (uintptr_t)d->out_f8 % 8 == 0),
                 "gemm fp8: out_f8 is the GEGLU epilogue's output (row pitch and pointer multiples of 8)");
  } else {
    SEVA_REQUIRE(!d->out_f8 && !d->w_exp, "gemm f16: out_f8 / w_exp belong to seva_gemm_fp8");
  }
  SEVA_REQUIRE(d->mode == 0 || d->mode == 1, "gemm: bad mode %d", d->mode);
  SEVA_REQUIRE(d->epilogue == 0 || d->epilogue == 1, "gemm: bad epilogue %d", d->epilogue);
  SEVA_REQUIRE(!d->row_add || d->rows_per_group > 0, "gemm: row_add needs rows_per_group");
  SEVA_REQUIRE(((uintptr_t)d->a | (uintptr_t)d->w | (uintptr_t)d->bias | (uintptr_t)d->row_add |
                (uintptr_t)d->residual | (uintptr_t)d->out_f32 | (uintptr_t)d->out_f16) % 16 == 0,
               "gemm: pointers must be 16-byte aligned");
  SEVA_REQUIRE((!d->residual || d->ldr % 4 == 0) && (!d->out_f32 || d->ldo32 % 4 == 0) &&
                   (!d->out_f16 || d->ldo16 % 4 == 0),
               "gemm: row pitches must be multiples of 4");
  GemmArgs a{};
  a.a = (const half_t*)d->a;
  a.w = (const half_t*)d->w;
  a.bias = d->bias;
  a.row_add = d->row_add;
  a.residual = d->residual;
  a.out_f32 = d->out_f32;
  a.out_f16 = (half_t*)d->out_f16;
  a.out_f8 = (uint8_t*)d->out_f8;
  a.w_exp = (const uint8_t*)d->w_exp;
  a.ch_stats = d->ch_stats;
  a.sk_ws = nullptr;
  SEVA_REQUIRE(!d->ch_stats || (d->out_f32 && d->epilogue == 0 && d->N >= 128 && (uintptr_t)d->ch_stats % 16 == 0 &&
                                d->col_scale_n == 0),
               "gemm: ch_stats needs the plain epilogue with an fp32 output, N >= 128, no col_scale, a 16-byte aligned buffer");
  a.M = d->M; a.N = d->N; a.K = d->K / KU;
  a.lda = d->lda / KU; a.ldr = d->ldr; a.ldo32 = d->ldo32; a.ldo16 = d->ldo16; a.ldo8 = d->ldo8;
  a.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1;
  a.col_scale = d->col_scale;
  a.col_scale_n = d->col_scale_n;
  SEVA_REQUIRE(d->col_scale_n >= 0 && d->col_scale_n % 4 == 0 && d->col_scale_n <= d->N,
               "gemm: col_scale_n=%d invalid", d->col_scale_n);
  SEVA_REQUIRE(d->col_scale_n == 0 || (!d->residual && !d->row_add && d->epilogue == 0 && d->N > 32),
               "gemm: col_scale needs the plain epilogue without residual / row_add");
  a.ldra = d->ld_row_add > 0 ? d->ld_row_add : d->N;
  SEVA_REQUIRE(a.ldra % 4 == 0, "gemm: ld_row_add must be a multiple of 4");
  if (d->mode == 1) {
    SEVA_REQUIRE(d->cin > 0 && d->cin % (64 * KU) == 0, "conv: cin=%d not a multiple of %d", d->cin, 64 * KU);
    SEVA_REQUIRE(d->K == 9LL * d->cin + (d->a2 ? d->K2 : 0), "conv: K=%lld != 9*cin (+ K2)", (long long)d->K);
    if (d->a2) {
      SEVA_REQUIRE(!FP8 && !d->upsample && d->K2 > 0 && d->K2 % 64 == 0 && d->lda2 >= d->K2 && d->lda2 % 8 == 0 &&
                       (uintptr_t)d->a2 % 16 == 0 && d->N > 32 && d->out_f32,
                   "conv: the folded second operand a2 needs the f16 stride-any 3x3 conv without upsample, K2 %% 64 == 0, lda2 >= K2 "
                   "(multiple of 8), N > 32, an fp32 output");
      a.a2 = (const half_t*)d->a2;
      a.lda2 = d->lda2;
      a.nk1 = (int)(9LL * d->cin / BK);
    }
    SEVA_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d", d->stride);
    SEVA_REQUIRE(!(d->upsample && d->stride != 1), "conv: upsample needs stride 1");
    const int eh = d->upsample ? 2 * d->ih : d->ih, ew = d->upsample ? 2 * d->iw : d->iw;
    const int pad_sum = d->pad_br_only ? 1 : 2;
    SEVA_REQUIRE(d->oh == (eh + pad_sum - 3) / d->stride + 1 && d->ow == (ew + pad_sum - 3) / d->stride + 1,
                 "conv: output %dx%d inconsistent with input %dx%d stride %d up %d pad_br_only %d", d->oh, d->ow,
                 d->ih, d->iw, d->stride, d->upsample, d->pad_br_only);
    SEVA_REQUIRE(d->M == (int64_t)d->n * d->oh * d->ow, "conv: M != n*oh*ow");
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.cin = d->cin / KU; a.oh = d->oh; a.ow = d->ow;
    a.stride = d->stride; a.upsample = d->upsample;
    a.pad_lo = d->pad_br_only ? 0 : 1;
  } else {
    SEVA_REQUIRE(d->lda >= d->K && d->lda % (8 * KU) == 0, "gemm: lda=%lld invalid", (long long)d->lda);
  }
  hipStream_t s = (hipStream_t)stream;
  SEVA_REQUIRE(d->alg_K >= 0 && d->alg_K <= d->K, "gemm: alg_K=%lld outside [0, K]", (long long)d->alg_K);
  const double flops = 2.0 * (double)d->M * (double)d->N * (double)(d->alg_K > 0 ? d->alg_K : d->K);  // reference-equivalent FLOP (seva_hip.h)
  // algorithmic HBM bytes: A (conv: the NHWC image) and W read once, residual read once, each output written once
  const double a_elems = (d->mode == 1 ? (double)d->n * d->ih * d->iw * d->cin : (double)d->M * (double)d->K);
  const double a2_bytes = (d->mode == 1 && d->a2) ? 2.0 * (double)d->M * (double)d->K2 : 0.0;
  const double n_out = d->epilogue == 1 ? (double)d->N / 2 : (double)d->N;
  const double esz = FP8 ? 1.0 : 2.0;
  const double alg_bytes = esz * a_elems + a2_bytes + esz * (double)d->N * (double)d->K + (d->bias ? 4.0 * (double)d->N : 0.0) +
                           (double)d->M * n_out * ((d->residual ? 4.0 : 0.0) + (d->out_f32 ? 4.0 : 0.0) +
                                                   (d->out_f16 ? 2.0 : 0.0) + (d->out_f8 ? 1.0 : 0.0));
  SevaProfScope prof(d->mode == 1 ? 1 : 0, flops, s, alg_bytes);
  if (d->epilogue == 1) {
    SEVA_REQUIRE(d->N % 64 == 0, "geglu: N=%lld not a multiple of 64", (long long)d->N);
    SEVA_REQUIRE(d->mode == 0, "geglu: plain mode only");
    SEVA_REQUIRE(!d->out_f16 || d->ldo16 % 8 == 0, "geglu: f16 row pitch must be a multiple of 8");
  }
  const bool narrow = d->N <= 32;
  if constexpr (FP8) {
    // e4m3 operands: the K >= 640 GEMMs / cin >= 640 convs of the ds2..ds8 levels.  Same tile-shape heuristics.
    bool half_m8 = ((d->M + 127) / 128) * ((d->N + 159) / 160) < 320 && d->M > 64;
    if (g_seva_knobs.gemm_bm > 0) half_m8 = g_seva_knobs.gemm_bm == 64;
    if (d->ch_stats) half_m8 = false;  // statistics are emitted per wave-owned 64-row block: 128-row tiles only
    bool wide8 = d->N % 160 == 0;
    if (g_seva_knobs.gemm_bn > 0) wide8 = g_seva_knobs.gemm_bn == 160;
    // (160-row GEGLU tiles, the f16 default, were measured here too: 255 registers with a small spill, no gain)
    if (d->epilogue == 1) return half_m8 ? launch<64, 128, 0, 1, true>(a, s) : launch<128, 128, 0, 1, true>(a, s);
    // 128-row tiles are 128 wide only: 128x160 with both k-steps' fragments live exceeds 256 VGPRs (spills)
    if (d->mode == 0) {
      if (half_m8) return wide8 ? launch<64, 160, 0, 0, true>(a, s) : launch<64, 128, 0, 0, true>(a, s);
      return launch<128, 128, 0, 0, true>(a, s);
    }
    // 3x3 / stride 1 / pad 1 convs: the window-staged kernel (conv_win.hip, e4m3 instantiations of its 128-column family)
    if (d->mode == 1 && d->a2 == nullptr && g_seva_knobs.gemm_dbg < 0 && g_seva_knobs.gemm_bm <= 0 && g_seva_knobs.gemm_bn <= 0 && g_seva_knobs.gemm_chunks <= 0) {
      const int rc = seva_conv_win_launch(a, s, true);
      if (rc <= 0) return rc;
    }
    if (half_m8) return wide8 ? launch<64, 160, 1, 0, true>(a, s) : launch<64, 128, 1, 0, true>(a, s);
    return launch<128, 128, 1, 0, true>(a, s);
  } else {
  // Small problems (the ds8 level: 27 x 8 tiles of 128 rows on 512 workgroup slots) get 64-row tiles: twice the
  // workgroups, both slots of a CU busy.  SEVA_GEMM_BM=64|128 forces the height (benchmark knob).
  bool half_m = ((d->M + 127) / 128) * ((d->N + 159) / 160) < 320 && d->M > 64;
  if (g_seva_knobs.gemm_bm > 0) half_m = g_seva_knobs.gemm_bm == 64;
  if (d->ch_stats) half_m = false;  // statistics are emitted per wave-owned 64-row block: 128-row tiles only
  const bool two_src = d->mode == 1 && d->a2 != nullptr;  // MODE 3: instantiated for 128- and 160-row tiles, never split-K
  if (two_src) half_m = false;
  // Split-K = 2 for convolutions over SMALL IMAGES (<= 128 output pixels per sample: the ds8 level, 9 x 9) with a long
  // reduction: 128-row tiles, two workgroups per tile, instead of 64-row tiles.  The choice looks at per-sample dimensions
  // only, so a sample's result does not depend on the batch size.
  if (d->splitk_ws && d->mode == 1 && !two_src && !d->upsample && !narrow && (int64_t)d->oh * d->ow <= 128 && d->K / BK >= 16 &&
      (d->K / BK) % 2 == 0 && g_seva_knobs.gemm_bm <= 0 && g_seva_knobs.gemm_dbg < 0 && g_seva_knobs.gemm_stagger < 0) {
    const int bn = (g_seva_knobs.gemm_bn > 0 ? g_seva_knobs.gemm_bn == 160 : d->N % 160 == 0) ? 160 : 128;
    const int64_t tiles = ((d->M + 127) / 128) * ((d->N + bn - 1) / bn);
    SEVA_REQUIRE((uintptr_t)d->splitk_ws % 16 == 0, "gemm: splitk_ws must be 16-byte aligned");
    // A workspace that cannot hold this launch's tiles is an ERROR (it was a silent fall-back to the unsplit 64-row kernel in
    // round 3: the reduction order of a sample would then have depended on the batch it was launched in and on the size of the
    // caller's workspace -- the opposite of what the per-sample split rule promises).  Size it with the formula of seva_hip.h.
    SEVA_REQUIRE(tiles < 16383 && d->splitk_ws_bytes >= (int64_t)(16384 + tiles * 128 * bn) * 4,
                 "gemm: splitk_ws too small for this launch: %lld tiles of 128 x %d need %lld bytes (and fewer than 16383 tiles), got %lld",
                 (long long)tiles, bn, (long long)((16384 + tiles * 128 * bn) * 4), (long long)d->splitk_ws_bytes);
    a.sk_ws = d->splitk_ws;
    half_m = false;
  }
  // 3x3 / stride 1 / pad 1 convs whose tile window fits LDS: the input window (+ halo) is staged once per 64-channel slab and the nine
  // taps read it through shifted fragment addresses (conv_win.hip); everything else keeps the per-tap gather below
  if (d->mode == 1 && !two_src && g_seva_knobs.gemm_dbg < 0 && g_seva_knobs.gemm_stagger < 0 && g_seva_knobs.gemm_bm <= 0 &&
      g_seva_knobs.gemm_bn <= 0 && g_seva_knobs.gemm_chunks <= 0) {
    const int rc = seva_conv_win_launch(a, s);
    if (rc <= 0) return rc;
  }
  if (d->epilogue == 1) {
    // GEGLU tiles are 128 wide (the epilogue pairs 64-row value / gate groups), so the cheaper operand stream comes from the
    // other side: 160 x 128 tiles -- 10 % fewer LDS-DMA bytes per FLOP, 40 instead of 32 MFMAs per wave and barrier, 215
    // registers, still two workgroups per CU.  Bitwise the same outputs; ds2 / ds4 -6 %, the 9x9 level -15 % against its 64-row
    // tiles (tools/kgeglu_bm.py).  K <= 320 keeps the A-in-registers kernel (a 128-row design).
    const bool dbg_run = g_seva_knobs.gemm_dbg >= 0 || g_seva_knobs.gemm_stagger >= 0;
    const bool tall = g_seva_knobs.gemm_bm == 160 || (g_seva_knobs.gemm_bm <= 0 && !dbg_run && d->K > 320 && d->M >= 1024);
    if (tall) return launch_p<160, 128, 0, 1, true>(a, s);
    return half_m ? launch<64, 128, 0, 1>(a, s) : launch<128, 128, 0, 1>(a, s);
  }
  // 128x160 tiles: every channel count of the network (320 .. 10240) is a multiple of 160, so no MFMA
  // column is idle (N = 320: 2 tiles instead of 3 with the last half empty), and a tile needs 10 %
  // fewer LDS-DMA bytes and fragment reads per FLOP than 128x128.  (128x64 tiles, tried earlier, were
  // 5-25 % slower: profiles/r01_kbench_bn64.log.)  SEVA_GEMM_BN=128|160 forces the width (benchmark knob).
  bool wide = d->N % 160 == 0;
  if (g_seva_knobs.gemm_bn > 0) wide = g_seva_knobs.gemm_bn == 160;
  // 160 x 160 tiles for the fp32-output kernels (not the f16-only ASYNC ones: their bias slots would not fit): 0.0125 operand bytes
  // per FLOP instead of 0.0141, 50 instead of 40 MFMAs per wave and barrier; two workgroups take EXACTLY the CU's 160 KiB of LDS
  // and all 256 registers (no spill in GEMM mode, 7 dwords in conv mode).  Bitwise the same outputs; -3 ... -10 % on every shape
  // measured, also where 160-row tiles quantise worse (tools/ktile160.py).  Launches that emit GroupNorm statistics keep 128
  // rows (a wave must own a 64-row block), as do the small ones (64-row tiles / split-K) and the fused-upsample conv.
  {
    const bool dbg_run = g_seva_knobs.gemm_dbg >= 0 || g_seva_knobs.gemm_stagger >= 0;
    const bool f16_only = d->mode == 0 && d->out_f16 && !d->out_f32 && !d->residual;
    const bool big = g_seva_knobs.gemm_bm == 160 ||
                     (g_seva_knobs.gemm_bm <= 0 && g_seva_knobs.gemm_bn <= 0 && g_seva_knobs.gemm_chunks <= 0 && !dbg_run && !half_m && d->M >= 2048);
    if (big && wide && !narrow && !d->ch_stats && !d->upsample && !a.sk_ws && !f16_only) {
      if (two_src) return launch_p<160, 160, 3, 0, false>(a, s);
      return d->mode == 0 ? launch_p<160, 160, 0, 0, false>(a, s) : launch_p<160, 160, 1, 0, false>(a, s);
    }
  }
  if (two_src) return wide ? launch<128, 160, 3, 0>(a, s) : launch<128, 128, 3, 0>(a, s);
  if (d->mode == 0) {
    if (narrow) return launch<128, 32, 0, 0>(a, s);
    if (half_m) return wide ? launch<64, 160, 0, 0>(a, s) : launch<64, 128, 0, 0>(a, s);
    return wide ? launch<128, 160, 0, 0>(a, s) : launch<128, 128, 0, 0>(a, s);
  }
  if (d->upsample) {  // the three Upsample convs of a step: general gather (MODE 2)
    if (narrow) return launch<128, 32, 2, 0>(a, s);
    if (half_m) return wide ? launch<64, 160, 2, 0>(a, s) : launch<64, 128, 2, 0>(a, s);
    return wide ? launch<128, 160, 2, 0>(a, s) : launch<128, 128, 2, 0>(a, s);
  }
  if (narrow) return launch<128, 32, 1, 0>(a, s);
  if (a.sk_ws) return wide ? launch_p<128, 160, 1, 0, false, false, false, true>(a, s) : launch_p<128, 128, 1, 0, false, false, false, true>(a, s);
  if (half_m) return wide ? launch<64, 160, 1, 0>(a, s) : launch<64, 128, 1, 0>(a, s);
  return wide ? launch<128, 160, 1, 0>(a, s) : launch<128, 128, 1, 0>(a, s);
  }  // !FP8
}
}  // namespace

extern "C" int seva_gemm_f16(const seva_gemm_desc* d, seva_stream_t stream) { return gemm_entry<false>(d, stream); }

// e4m3 x e4m3 -> fp32 on the block-scaled MFMA (BASELINE config 5): A and W are OCP e4m3 bytes, K counts e4m3 elements
// (K % 128 == 0; conv: cin % 128 == 0), w_exp[n] = 127 + e[n] is the weight row's power-of-two scale.
extern "C" int seva_gemm_fp8(const seva_gemm_desc* d, seva_stream_t stream) { return gemm_entry<true>(d, stream); }
