// Flash-style scaled-dot-product attention for head dim 64 on gfx950 (MI355X), fp16 in/out.
//
// One wave owns 32 query rows.  Scores are computed TRANSPOSED, S^T = K * Q^T with
// v_mfma_f32_32x32x16_f16 (A = K rows from LDS, B = Q fragments kept in registers), so the query
// sits on the lane and a lane's 16 accumulator registers per 32-key block are 16 keys of ITS
// query: row max / row sum are register-local plus one exchange with lane^32.  The f32 scores,
// exponentiated and packed to fp16 in place, are directly the B operand of the second product
// O^T = V^T * P^T (accumulator-as-operand, k order permuted consistently on both operands); the
// V^T fragments come from the row-major V tile in LDS through ds_read_b64_tr_b16.  Online softmax
// in the log2 domain, fp32 throughout.
//
// K tile: 128-byte rows, 16-byte chunk c of row r stored at c ^ ((r>>1)&7)  (conflict-free b128 row reads)
// V tile: 128-byte rows, chunk c of row r stored at c ^ (((r>>1)&1)<<2)     (conflict-free tr reads)
//
// Regimes: per-frame / joint attention use 4 waves x 64-key tiles; the temporal
// regime (L = num_frames <= 32, batch = pixels) uses one wave per (pixel, head) with a 32-key
// tile and reads its tokens through the token stride, so no (b t) s c <-> (b s) t c transpose exists.
#include "seva_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

struct AttnArgs {
  const half_t* q;
  const half_t* k;
  const half_t* v;
  half_t* out;
  int64_t q_sb0, q_sb1, q_sl;
  int64_t k_sb0, k_sb1, k_sl;
  int64_t o_sb0, o_sb1, o_sl;
  int32_t nb1, heads, lq, lk, qblocks;
  float scale_log2;  // softmax scale * log2(e)  (unused when q is pre-scaled)
  int32_t dbg;       // ablation bits (SEVA_ATTN_DBG; timing only): 1 no K/V reloads, 2 no softmax, 4 no P*V, 8 no Q*K
  // K/V split (attn2_kernel<.., true> + attn_combine_kernel): every (batch, head, query block) is computed by `nsplit` workgroups,
  // each over a contiguous range of whole K/V tiles, which leave their UN-normalised fp32 O rows and (m, l) in the workspace
  float* part_o;     // [nsplit][batch * heads * lq][64]
  float* part_ml;    // [nsplit][batch * heads * lq][2]
  int32_t nsplit;
};

typedef short short8_t __attribute__((ext_vector_type(8)));

// LDS-DMA from inline asm (see gemm_common.h: hipcc would otherwise drain it with vmcnt(0) before
// every ds_read); ordered only by the counted waits + barriers below.
__device__ __forceinline__ void glds16_raw(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_wave_base)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ half8_t tr_read_pair(const char* a0, const char* a1) {
  short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a0);
  short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a1);
  short8_t s = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8_t, s);
}

__device__ __forceinline__ int v_chunk_swz(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }
__device__ __forceinline__ int k_chunk_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// DBGK: the ablation instantiation reads p.dbg at run time; the production one has no such branches
// (they split the tile body into dozens of basic blocks and stop the scheduler interleaving
// LDS reads, MFMAs and the softmax VALU work across them).
// PRE: q already carries scale*log2(e) (folded into the producing GEMM's fp32 epilogue).  The score
// accumulators then START at -m_run (splat per key block: 16 moves instead of 32 v_fma), so
// S^T = K Q^T comes out of the MFMA already in exp2's argument form.
// max over the two half-waves (lanes l and l ^ 32) in every lane.  v_permlane32_swap exchanges the upper half of its first
// operand with the lower half of its second: (a, b) = (v, v) -> a = {v.lo, v.lo}, b = {v.hi, v.hi}.  Inline asm: the
// builtin of this toolchain returns element 0 of its result pair for both elements (checked on a probe kernel); the s_nop
// covers the VALU-write -> permlane-read wait states the compiler would insert for the builtin.
__device__ __forceinline__ float half_wave_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

template <int NW, int KT, bool USE_TR, bool DBGK, bool PRE>
__global__ __launch_bounds__(NW * 64, 3) void attn_kernel(AttnArgs p) {
  constexpr int KB = KT / 32;               // 32-key blocks per tile
  const int dbg = DBGK ? p.dbg : 0;

  // 3-deep ring of K/V tiles filled by LDS-DMA two tiles ahead: [buf][K tile | V tile]
  __shared__ __attribute__((aligned(16))) char smem[3 * 2 * KT * 128];
  constexpr int BUF_BYTES = 2 * KT * 128;
  constexpr int IP = KT / 8 / NW;  // 8-row LDS-DMA instructions per wave per operand per tile
  constexpr int G = 2 * IP;        // ... per wave per tile (K and V)
  static_assert(KT % (8 * NW) == 0, "tile rows must split over the waves");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = lane & 31, hh = lane >> 5;

  // logical id = ((batch * heads) + head) * qblocks + qblock.  Hardware deals blocks b, b+8, ... to
  // one XCD; the bijective remap gives every XCD one contiguous run of logical ids, so the q-blocks
  // of a (batch, head) pair -- which all stream the same K/V -- share one L2.
  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7;
    bid = ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
  }
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  const int head = bid % p.heads;
  const int batch = bid / p.heads;
  const int b0 = batch / p.nb1, b1 = batch - b0 * p.nb1;

  const half_t* const qbase = p.q + b0 * p.q_sb0 + b1 * p.q_sb1 + head * 64;
  const half_t* const kbase = p.k + b0 * p.k_sb0 + b1 * p.k_sb1 + head * 64;
  const half_t* const vbase = p.v + b0 * p.k_sb0 + b1 * p.k_sb1 + head * 64;
  half_t* const obase = p.out + b0 * p.o_sb0 + b1 * p.o_sb1 + head * 64;

  const int qrow = qb * (32 * NW) + wave * 32 + qi;
  const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;

  half8_t qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *(const half8_t*)(qbase + (int64_t)qrow_c * p.q_sl + 16 * s + 8 * hh);

  f32x16 acc_o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[d][r] = 0.f;
  // m_run is the exponent reference in the log2 domain (scaled scores); it may lag the true running
  // max by up to RESCALE_THR (deferred rescale), so probabilities are bounded by 2^RESCALE_THR.
  constexpr float RESCALE_THR = 8.0f;
  float m_run = PRE ? 0.f : -1e30f, l_run = 0.f;
  // PRE: -m_run held as a 16-register block = the C operand of the first score MFMA of every key block (rebuilt only when
  // m_run moves, i.e. in the rare rescale branch, instead of 15 v_mov per tile)
  f32x16 neg_m[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) neg_m[0][r] = neg_m[1][r] = 0.f;
  const float c = p.scale_log2;

  const int nt = (p.lk + KT - 1) / KT;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned smem_base =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  // lane (r = lane>>3, physical chunk = lane&7) of each 8-row wave-instruction; the swizzle is applied
  // on the SOURCE chunk (the LDS image of an LDS-DMA is lane-linear)
  const int sr = lane >> 3, sp = lane & 7;
  // Tiles are issued strictly in order 0, 1, 2, ...: the per-lane source pointers of the NEXT tile to issue are
  // running state, advanced by the workgroup-uniform tile stride (the 64-bit key * stride products cost four
  // quarter-rate v_mul_lo_u32 / two v_mad_u64_u32 per tile in a VALU-bound kernel).  Only a partial last tile
  // recomputes its addresses, to clamp the rows past lk onto the last key.
  const half_t* kp[IP];
  const half_t* vp[IP];
#pragma unroll
  for (int i = 0; i < IP; ++i) {
    const int row = 8 * (wave_u * IP + i) + sr;
    const int key = row < p.lk ? row : p.lk - 1;
    kp[i] = kbase + (int64_t)key * p.k_sl + k_chunk_swz(row, sp) * 8;
    vp[i] = vbase + (int64_t)key * p.k_sl + v_chunk_swz(row, sp) * 8;
  }
  const int64_t tile_stride = (int64_t)KT * p.k_sl;
  const bool ragged = (p.lk % KT) != 0;
  auto issue_tile = [&](int kt, int buf) {
    const bool clamp = ragged && kt == nt - 1 && kt > 0;  // (tile 0 was clamped when the pointers were built)
#pragma unroll
    for (int i = 0; i < IP; ++i) {
      const unsigned dst = smem_base + buf * BUF_BYTES + 8 * (wave_u * IP + i) * 128;
      if (clamp) {
        const int row = 8 * (wave_u * IP + i) + sr;
        int key = kt * KT + row;
        if (key >= p.lk) key = p.lk - 1;
        const int64_t roff = (int64_t)key * p.k_sl;
        glds16_raw(kbase + roff + k_chunk_swz(row, sp) * 8, dst);
        glds16_raw(vbase + roff + v_chunk_swz(row, sp) * 8, dst + KT * 128);
      } else {
        glds16_raw(kp[i], dst);
        glds16_raw(vp[i], dst + KT * 128);
      }
      kp[i] += tile_stride;
      vp[i] += tile_stride;
    }
  };

  const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;

  // One K/V tile.  MASKED (tail tile) is compile-time so that the tail mask costs nothing on full tiles.  The LDS buffer
  // index is a RUN-TIME (wave-uniform) value: with the three buffers as three unrolled copies of this body hipcc gave the
  // loop-carried O accumulators different registers in different copies and moved all 32 of them (16 v_mov_b64 behind two
  // `s_nop 11` MFMA-result hazards) in every tile, and rebuilt the 16-register splat of -m_run; one body has one assignment.
  auto tile = [&](int buf, auto masked_c, int kt) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const char* const lds_k = smem + buf * BUF_BYTES;
    const char* const lds_v = lds_k + KT * 128;
    // tile kt+2 -> buffer (kt+2)%3, last read in iteration kt-1 (every wave passed that barrier)
    const bool more2 = kt + 2 < nt && !(dbg & 1);
    if (more2) issue_tile(kt + 2, buf == 0 ? 2 : buf - 1);  // (buf + 2) % 3

    // ---- S^T = K Q^T ----
    f32x16 sc[KB];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int krow = 32 * kb + qi;
      if (!(dbg & 8)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const half8_t kf =
              *(const half8_t*)(lds_k + krow * 128 + (k_chunk_swz(krow, 2 * s + hh) << 4));
          if (s == 0) {
            f32x16 c0;
            if (PRE) {
              c0 = neg_m[kb & 1];
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) c0[r] = 0.f;
            }
            sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], c0, 0, 0, 0);
          } else {
            sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sc[kb], 0, 0, 0);
          }
        }
      } else {
        asm volatile("" : "=v"(sc[kb]));
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (MASKED) {  // keys >= lk exist only in the last tile
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KT + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= p.lk) sc[kb][r] = -1e30f;
        }
    }
    // ---- online softmax, log2 domain, deferred rescale ----
    float mx = -1e30f;
    if (!(dbg & 2)) {
      // one max chain per key block (independent: half the dependent-issue depth), then the other half-wave's maximum of the
      // same query by v_permlane32_swap (a VALU op; ds_bpermute costs an LDS round trip and an lgkmcnt(0) per tile)
      float mk[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        mk[kb] = sc[kb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mk[kb] = fmaxf(mk[kb], sc[kb][r]);
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) mx = fmaxf(mx, mk[kb]);
      mx = half_wave_max(mx);
    }
    if (PRE) {
      // scores are already relative to m_run: rescale when a row maximum exceeds the threshold, and
      // unconditionally on the first tile (m_run starts at 0; nothing accumulated yet, so alpha = 1)
      const bool first = kt == 0;
      if (!(dbg & 2) && (first || __any(mx > RESCALE_THR))) {  // wave-uniform
        const float delta = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) neg_m[0][r] = neg_m[1][r] = -m_run;
        asm volatile("" : "+v"(neg_m[0]), "+v"(neg_m[1]));  // two distinct blocks (not one value the compiler may merge)
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[kb][r] -= delta;
      }
    } else {
    const float mxs = mx * c;  // c > 0
    if (!(dbg & 2) && __any(mxs > m_run + RESCALE_THR)) {  // wave-uniform
      const float m_new = fmaxf(m_run, mxs);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[d][r] *= alpha;
    }
    }
    // P = exp2(S*c - m) packed to fp16 pairs (round-toward-zero); the row sum is taken from the
    // ROUNDED values (v_dot2_f32_f16 with ones) so numerator (P*V) and normaliser see identical
    // probabilities and the truncation cancels in O = sum(p v) / sum(p).
    typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
    const fp16x2_t ones2 = {(__fp16)1.0f, (__fp16)1.0f};
    float lsum4[4] = {0.f, 0.f, 0.f, 0.f};  // four independent dot2 chains
    half8_t pf[KB][2];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        fp16x2_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float e0, e1;
          if (!(dbg & 2)) {
            if (PRE) {
              e0 = __builtin_amdgcn_exp2f(sc[kb][8 * s2 + 2 * j]);
              e1 = __builtin_amdgcn_exp2f(sc[kb][8 * s2 + 2 * j + 1]);
            } else {
              e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][8 * s2 + 2 * j], c, -m_run));
              e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][8 * s2 + 2 * j + 1], c, -m_run));
            }
          } else {
            e0 = sc[kb][8 * s2 + 2 * j];
            e1 = sc[kb][8 * s2 + 2 * j + 1];
          }
          pk[j] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
          if (!(dbg & 2)) lsum4[j] = __builtin_amdgcn_fdot2(pk[j], ones2, lsum4[j], false);
        }
        pf[kb][s2] = __builtin_bit_cast(half8_t, pk);
      }
    l_run += (lsum4[0] + lsum4[1]) + (lsum4[2] + lsum4[3]);

    // ---- O^T += V^T P^T ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int r0 = 32 * kb + 16 * s2 + 4 * hh;
          half8_t vf;
          if (USE_TR) {
            const int colbyte = (32 * db + 16 * (g16 & 1) + 4 * p4) * 2;
            const int ch = colbyte >> 4, within = colbyte & 15;
            const int ra = r0 + q4, rb = r0 + 8 + q4;
            vf = tr_read_pair(lds_v + ra * 128 + (v_chunk_swz(ra, ch) << 4) + within,
                              lds_v + rb * 128 + (v_chunk_swz(rb, ch) << 4) + within);
          } else {
            const int d = 32 * db + qi;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int row = r0 + (j & 3) + 8 * (j >> 2);
              vf[j] = *(const half_t*)(lds_v + row * 128 + (v_chunk_swz(row, d >> 3) << 4) + (d & 7) * 2);
            }
          }
          if (!(dbg & 4)) acc_o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][s2], acc_o[db], 0, 0, 0);
          else asm volatile("" ::"v"(vf), "v"(pf[kb][s2]));
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- stage tile kt+1 into the other buffer (its last readers passed the previous barrier) ----
    // tile kt+1 (issued one iteration ago) must have landed; tile kt+2's G instructions may fly on
    if (more2) wait_vm<G>();
    else wait_vm<0>();
    __syncthreads();
  };

  issue_tile(0, 0);
  if (nt > 1) {
    issue_tile(1, 1);
    wait_vm<G>();
  } else {
    wait_vm<0>();
  }
  // Retire the Q loads HERE as far as hipcc's wait-count bookkeeping is concerned.  Otherwise they are
  // "possibly pending" at the loop header (merged over the back-edge), the compiler guards the first
  // use of qf[0..3] in EVERY tile with s_waitcnt vmcnt(3)..vmcnt(0), and that vmcnt(0) drains the
  // LDS-DMA ring (which it cannot see: the DMAs are issued from inline asm) once per tile.
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));
  __syncthreads();
  const int nfull = p.lk / KT;  // tiles that need no mask (the ragged tail tile, if any, is the last one)
  int buf = 0;                  // buffer index == tile % 3
  for (int kt = 0; kt < nfull; ++kt) {
    tile(buf, std::false_type{}, kt);
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (nfull < nt) tile(buf, std::true_type{}, nfull);

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  // ---- epilogue: O through LDS so that every global store is a full 128-byte row ----
  // (per-lane 8-byte stores at a row stride touch 32 lines per instruction and amplify HBM writes;
  // profiles/r01_traffic.json).  Each wave owns a 32-row x 128-byte image in the (now idle) K/V
  // buffers; 16-byte chunk c of row r sits at c ^ (r & 7).
  __syncthreads();  // every wave is done reading K/V tiles
  char* const ow = smem + wave * (32 * 128);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      half4_t h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc_o[db][4 * t + r] * inv);
      const int d0 = 32 * db + 8 * t + 4 * hh;          // first of 4 consecutive head dims
      const int chunk = d0 >> 3, piece = (d0 >> 2) & 1;  // 16-byte chunk, 8-byte half
      *(half4_t*)(ow + qi * 128 + ((chunk ^ (qi & 7)) << 4) + (piece << 3)) = h;
    }
  // same wave reads back its own image: no workgroup barrier needed, only the LDS round trip
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  const int q0 = qb * (32 * NW) + wave * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3), pchunk = lane & 7;
    const uint4 v = *(const uint4*)(ow + row * 128 + (pchunk << 4));
    const int lchunk = pchunk ^ (row & 7);
    if (q0 + row < p.lq) *(uint4*)(obase + (int64_t)(q0 + row) * p.o_sl + lchunk * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Two-query-block variant for long sequences (engine path: pre-scaled q, tr-reads): a wave owns 64 query rows as two 32-row
// blocks A and B (2 waves per SIMD) and every K fragment and V^T fragment it reads from LDS feeds BOTH blocks' MFMAs:
//   scores(A, B) | softmax(A) | softmax(B) | PV(A, B)
// Per query that is half the LDS fragment reads, half the K/V LDS-DMA bytes and instructions (a workgroup's tile serves 256
// queries) and half the per-tile bookkeeping of attn_kernel -- the components whose costs ADD UP there (ablations of
// the pipelined kernel).  Same LDS ring, operand layouts, online-softmax arithmetic and rounding as
// attn_kernel<4, 64, true, false, true>; one loop body with a run-time buffer index, running source pointers, half-wave maximum
// by v_permlane32_swap (as there).
template <int KT, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attn2_kernel(AttnArgs p) {
  constexpr int NW = 4, QB = 2, KB = KT / 32;
  // SPLIT: this workgroup walks only the K/V tiles [ksp * tps, (ksp + 1) * tps) of its (batch, head) and writes a partial result.
  // Workgroups of one (batch, head, split) are adjacent in the launch order (they stream the same K/V range through one L2).
  __shared__ __attribute__((aligned(16))) char smem[3 * 2 * KT * 128];
  constexpr int BUF_BYTES = 2 * KT * 128;
  constexpr int IP = KT / 8 / NW;
  constexpr int G = 2 * IP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = lane & 31, hh = lane >> 5;

  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7;
    bid = ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
  }
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  int ksp = 0;
  if (SPLIT) {
    ksp = bid % p.nsplit;
    bid /= p.nsplit;
  }
  const int head = bid % p.heads;
  const int batch = bid / p.heads;
  const int b0 = batch / p.nb1, b1 = batch - b0 * p.nb1;
  // this workgroup's key range [key0, key0 + lk): the whole sequence, or the split's run of whole tiles
  int key0 = 0, lk = p.lk;
  if (SPLIT) {
    const int nt_all = (p.lk + KT - 1) / KT, tps = (nt_all + p.nsplit - 1) / p.nsplit;
    key0 = ksp * tps * KT;
    const int key1 = (ksp + 1) * tps * KT < p.lk ? (ksp + 1) * tps * KT : p.lk;
    lk = key1 - key0;  // > 0: the host only splits when every split gets at least one tile
  }

  const half_t* const qbase = p.q + b0 * p.q_sb0 + b1 * p.q_sb1 + head * 64;
  const half_t* const kbase = p.k + b0 * p.k_sb0 + b1 * p.k_sb1 + head * 64 + (int64_t)key0 * p.k_sl;
  const half_t* const vbase = p.v + b0 * p.k_sb0 + b1 * p.k_sb1 + head * 64 + (int64_t)key0 * p.k_sl;
  half_t* const obase = p.out + b0 * p.o_sb0 + b1 * p.o_sb1 + head * 64;

  half8_t qf[QB][4];
#pragma unroll
  for (int c = 0; c < QB; ++c) {
    const int qrow = qb * (32 * QB * NW) + wave * (32 * QB) + 32 * c + qi;
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[c][s] = *(const half8_t*)(qbase + (int64_t)qrow_c * p.q_sl + 16 * s + 8 * hh);
  }

  // a wave whose 64 query rows all lie past lq (ragged last query block: 5184 = 20.25 x 256) keeps feeding the K/V ring and
  // meeting the barriers but computes nothing: its MFMA / VALU / LDS-read slots go to the other workgroup on the CU
  const bool active = __builtin_amdgcn_readfirstlane(qb * (32 * QB * NW) + wave * (32 * QB)) < p.lq;
  f32x16 acc_o[QB][2];
  float m_run[QB], l_run[QB];
  // -m_run as a 16-register block per query block: the C operand of the first score MFMA of every key block (as in attn_kernel),
  // rebuilt only in the rare rescale branch: 64 v_mov per tile fewer in a loop whose VALU work, not its MFMA work, sets the pace
  f32x16 neg_m[QB];
#pragma unroll
  for (int c = 0; c < QB; ++c) {
    m_run[c] = 0.f;
    l_run[c] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) neg_m[c][r] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[c][d][r] = 0.f;
  }
  constexpr float P_SUM_BOUND = 16384.0f;  // a lane's partial row sum at or above 2^14: look at the maximum, rescale (see the tile body)

  const int nt = (lk + KT - 1) / KT;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned smem_base =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int sr = lane >> 3, sp = lane & 7;
  const half_t* kp[IP];
  const half_t* vp[IP];
#pragma unroll
  for (int i = 0; i < IP; ++i) {
    const int row = 8 * (wave_u * IP + i) + sr;
    const int key = row < lk ? row : lk - 1;
    kp[i] = kbase + (int64_t)key * p.k_sl + k_chunk_swz(row, sp) * 8;
    vp[i] = vbase + (int64_t)key * p.k_sl + v_chunk_swz(row, sp) * 8;
  }
  const int64_t tile_stride = (int64_t)KT * p.k_sl;
  const bool ragged = (lk % KT) != 0;
  auto issue_tile = [&](int kt, int buf) {  // tiles are issued strictly in order 0, 1, 2, ...
    const bool clamp = ragged && kt == nt - 1 && kt > 0;
#pragma unroll
    for (int i = 0; i < IP; ++i) {
      const unsigned dst = smem_base + buf * BUF_BYTES + 8 * (wave_u * IP + i) * 128;
      if (clamp) {
        const int row = 8 * (wave_u * IP + i) + sr;
        int key = kt * KT + row;
        if (key >= lk) key = lk - 1;
        const int64_t roff = (int64_t)key * p.k_sl;
        glds16_raw(kbase + roff + k_chunk_swz(row, sp) * 8, dst);
        glds16_raw(vbase + roff + v_chunk_swz(row, sp) * 8, dst + KT * 128);
      } else {
        glds16_raw(kp[i], dst);
        glds16_raw(vp[i], dst + KT * 128);
      }
      kp[i] += tile_stride;
      vp[i] += tile_stride;
    }
  };
  const int g16 = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
  typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
  const fp16x2_t ones2 = {(__fp16)1.0f, (__fp16)1.0f};

  auto tile = [&](int buf, auto masked_c, int kt) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const char* const lds_k = smem + buf * BUF_BYTES;
    const char* const lds_v = lds_k + KT * 128;
    const bool more2 = kt + 2 < nt;
    if (more2) issue_tile(kt + 2, buf == 0 ? 2 : buf - 1);  // (buf + 2) % 3

    if (active) {
    // ---- S^T = K Q^T for both query blocks off ONE K fragment; accumulators start at -m_run (q carries scale*log2e) ----
    f32x16 sc[QB][KB];
    __builtin_amdgcn_s_setprio(1);
    if constexpr (!SPLIT) {
      // every K fragment of the tile is ISSUED before the first score MFMA (hipcc otherwise reads one pair ahead and waits for it in front
      // of every two MFMAs).  Measured, same box (profiles/r04_kattn_frags_first.log): per-frame attention at 72x72 1834 -> 1788 us; the K/V-split
      // instantiation spills 7 registers with it and keeps hipcc's order; the same for the V^T fragments of the P V phase loses 1 - 3 %
      half8_t kfa[KB][4];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int krow = 32 * kb + qi;
#pragma unroll
        for (int s = 0; s < 4; ++s) kfa[kb][s] = *(const half8_t*)(lds_k + krow * 128 + (k_chunk_swz(krow, 2 * s + hh) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c = 0; c < QB; ++c)
            sc[c][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfa[kb][s], qf[c][s], s == 0 ? neg_m[c] : sc[c][kb], 0, 0, 0);
    } else {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int krow = 32 * kb + qi;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8_t kf = *(const half8_t*)(lds_k + krow * 128 + (k_chunk_swz(krow, 2 * s + hh) << 4));
#pragma unroll
        for (int c = 0; c < QB; ++c) {
          if (s == 0) {
            sc[c][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[c][s], neg_m[c], 0, 0, 0);
          } else {
            sc[c][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[c][s], sc[c][kb], 0, 0, 0);
          }
        }
      }
    }
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- online softmax per query block: mask, running maximum, rare rescale; exp2 / pack / row sums ----
    half8_t pf[QB][KB][2];
#pragma unroll
    for (int c = 0; c < QB; ++c) {
      if (MASKED) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * KT + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= lk) sc[c][kb][r] = -1e30f;
          }
      }
      // exp2 / pack / row sums FIRST, straight off the scores relative to m_run; the running maximum is looked at only when a
      // partial row sum says a probability may have left the safe range (any P >= 2^14 makes its lane's sum >= 2^14; f16 holds
      // 2^16 - 32 and cvt_pkrtz never rounds up), and on the first tile (m_run = 0 there: the true maximum protects the small
      // probabilities from underflow).  The 21 v_max3 + the half-wave exchange per query block leave the steady-state loop.
      float ls[4];
      const auto exp_pack = [&]() {
        ls[0] = ls[1] = ls[2] = ls[3] = 0.f;  // four partial sums: short dependent dot2 chains
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            fp16x2_t pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float e0 = __builtin_amdgcn_exp2f(sc[c][kb][8 * s2 + 2 * j]);
              const float e1 = __builtin_amdgcn_exp2f(sc[c][kb][8 * s2 + 2 * j + 1]);
              pk[j] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
              ls[j] = __builtin_amdgcn_fdot2(pk[j], ones2, ls[j], false);
            }
            pf[c][kb][s2] = __builtin_bit_cast(half8_t, pk);
          }
        return (ls[0] + ls[1]) + (ls[2] + ls[3]);
      };
      float tot = exp_pack();
      const bool first = kt == 0;
      if (__builtin_expect(first || __any(!(tot < P_SUM_BOUND)), 0)) {  // wave-uniform, rare (NaN / inf sums land here too)
        float mk[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          mk[kb] = sc[c][kb][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mk[kb] = fmaxf(mk[kb], sc[c][kb][r]);
        }
        float mx = mk[0];
#pragma unroll
        for (int kb = 1; kb < KB; ++kb) mx = fmaxf(mx, mk[kb]);
        mx = half_wave_max(mx);
        const float delta = first ? mx : fmaxf(mx, 0.f);
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        m_run[c] += delta;
        l_run[c] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) neg_m[c][r] = -m_run[c];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc_o[c][d][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[c][kb][r] -= delta;
        tot = exp_pack();
      }
      l_run[c] += tot;
    }
    // ---- O^T += V^T P^T for both query blocks off ONE V^T fragment ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int r0 = 32 * kb + 16 * s2 + 4 * hh;
          const int colbyte = (32 * db + 16 * (g16 & 1) + 4 * p4) * 2;
          const int ch = colbyte >> 4, within = colbyte & 15;
          const int ra = r0 + q4, rb = r0 + 8 + q4;
          const half8_t vf = tr_read_pair(lds_v + ra * 128 + (v_chunk_swz(ra, ch) << 4) + within,
                                          lds_v + rb * 128 + (v_chunk_swz(rb, ch) << 4) + within);
#pragma unroll
          for (int c = 0; c < QB; ++c)
            acc_o[c][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[c][kb][s2], acc_o[c][db], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    }  // active
    if (more2) wait_vm<G>();
    else wait_vm<0>();
    __syncthreads();
  };

  issue_tile(0, 0);
  if (nt > 1) {
    issue_tile(1, 1);
    wait_vm<G>();
  } else {
    wait_vm<0>();
  }
#pragma unroll
  for (int c = 0; c < QB; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[c][s]));  // retire the Q loads (see attn_kernel)
  __syncthreads();
  const int nfull = lk / KT;
  int buf = 0;
  for (int kt = 0; kt < nfull; ++kt) {
    tile(buf, std::false_type{}, kt);
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (nfull < nt) tile(buf, std::true_type{}, nfull);

  if constexpr (SPLIT) {
    // partial result: un-normalised O (relative to m_run) as fp32, 16 bytes per lane and (db, t), and (m_run, l) per query row
    const int64_t rows_all = (int64_t)gridDim.x / (p.qblocks * p.nsplit) * p.lq;  // batch * heads * lq
    const int64_t row_bh = ((int64_t)batch * p.heads + head) * p.lq;
#pragma unroll
    for (int c = 0; c < QB; ++c) {
      const int qrow = qb * (32 * QB * NW) + wave * (32 * QB) + 32 * c + qi;
      const float l_tot = l_run[c] + __shfl_xor(l_run[c], 32, 64);
      if (qrow < p.lq) {
        float* const po = p.part_o + ((int64_t)ksp * rows_all + row_bh + qrow) * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            *(f32x4*)(po + 32 * db + 8 * t + 4 * hh) =
                f32x4{acc_o[c][db][4 * t], acc_o[c][db][4 * t + 1], acc_o[c][db][4 * t + 2], acc_o[c][db][4 * t + 3]};
        if (hh == 0) {
          float* const pm = p.part_ml + ((int64_t)ksp * rows_all + row_bh + qrow) * 2;
          pm[0] = m_run[c];
          pm[1] = l_tot;
        }
      }
    }
    return;
  }
  __syncthreads();  // every wave is done reading K/V tiles
#pragma unroll
  for (int c = 0; c < QB; ++c) {
    const float l_tot = l_run[c] + __shfl_xor(l_run[c], 32, 64);
    const float inv = 1.0f / l_tot;
    char* const ow = smem + (wave * QB + c) * (32 * 128);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        half4_t h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc_o[c][db][4 * t + r] * inv);
        const int d0 = 32 * db + 8 * t + 4 * hh;
        const int chunk = d0 >> 3, piece = (d0 >> 2) & 1;
        *(half4_t*)(ow + qi * 128 + ((chunk ^ (qi & 7)) << 4) + (piece << 3)) = h;
      }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < QB; ++c) {
    const char* const ow = smem + (wave * QB + c) * (32 * 128);
    const int q0 = qb * (32 * QB * NW) + wave * (32 * QB) + 32 * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (lane >> 3), pchunk = lane & 7;
      const uint4 v = *(const uint4*)(ow + row * 128 + (pchunk << 4));
      const int lchunk = pchunk ^ (row & 7);
      if (q0 + row < p.lq) *(uint4*)(obase + (int64_t)(q0 + row) * p.o_sl + lchunk * 8) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// attn2_kernel's two-block scheme on v_mfma_f32_16x16x32_f16: a wave owns 64 query rows as FOUR 16-row blocks and every K fragment
// and V^T fragment it reads from LDS feeds all four blocks' MFMAs.  Why the other MFMA shape: the chip holds its clock down under
// matrix load, and lower for v_mfma_f32_32x32x16 than for 16x16x32 -- the same FLOP took 653 ns against 553 ns in bare loops on random
// data (1.63 against 1.97 GHz; tools/micro/pingpong_probe.hip, profiles/r04_pingpong_probe.log; MI355X guide, DVFS item 7), and
// in this kernel MFMA time and VALU time ADD (same probe: neither SIMD partners nor one wave's own stream overlap them).
// Same counts as attn2_kernel per 64 x 64 tile and wave: 8 K fragments, 8 V^T fragment pairs, 64 exp2, 32 packs, 32 dot2; 64 MFMAs of
// 16 cycles instead of 32 of 32; the -m_run C-operand block is 4 registers per query block instead of 16.
//   S^T block (16 keys x 16 queries) = K(16 x 32 d) Q^T: lane (i16 = lane & 15, g = lane >> 4) holds keys 4g .. 4g+3 of query i16
//   P^T as B operand of O^T(16 d x 16 queries) += V^T(16 d x 32 keys) P^T: the 32 keys of a step are two 16-key blocks in the order
//   kappa = 8g + r <-> key 4g + r of block 2j, kappa = 8g + 4 + r <-> key 4g + r of block 2j + 1: a lane's B fragment is its own
//   registers of the two blocks, and the V^T fragment is two ds_read_b64_tr_b16 of rows 32j + 4g .. +3 and 32j + 16 + 4g .. +3.
// K tile: attn_kernel's swizzle (conflict-free for these b128 reads too); V tile: chunk c of row r at c ^ (((r >> 1) & 3) << 1).
// Not bitwise equal to the 32x32x16 kernels (the MFMA shapes sum their k products in different orders): tested against fp64.
__device__ __forceinline__ int v16_chunk_swz(int row, int chunk) { return chunk ^ (((row >> 1) & 3) << 1); }

template <int KT, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attn16_kernel(AttnArgs p) {
  constexpr int NW = 4, NQ = 4, NKB = KT / 16, NJ = KT / 32;
  __shared__ __attribute__((aligned(16))) char smem[3 * 2 * KT * 128];
  constexpr int BUF_BYTES = 2 * KT * 128;
  constexpr int IP = KT / 8 / NW;
  constexpr int G = 2 * IP;
  static_assert(3 * BUF_BYTES >= NW * 64 * 128, "the output staging re-uses the ring");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;

  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7;
    bid = ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
  }
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  int ksp = 0;
  if (SPLIT) {
    ksp = bid % p.nsplit;
    bid /= p.nsplit;
  }
  const int head = bid % p.heads;
  const int batch = bid / p.heads;
  const int b0 = batch / p.nb1, b1 = batch - b0 * p.nb1;
  int key0 = 0, lk = p.lk;
  if (SPLIT) {
    const int nt_all = (p.lk + KT - 1) / KT, tps = (nt_all + p.nsplit - 1) / p.nsplit;
    key0 = ksp * tps * KT;
    const int key1 = (ksp + 1) * tps * KT < p.lk ? (ksp + 1) * tps * KT : p.lk;
    lk = key1 - key0;  // > 0: the host only splits when every split gets at least one tile
  }

  const half_t* const qbase = p.q + b0 * p.q_sb0 + b1 * p.q_sb1 + head * 64;
  const half_t* const kbase = p.k + b0 * p.k_sb0 + b1 * p.k_sb1 + head * 64 + (int64_t)key0 * p.k_sl;
  const half_t* const vbase = p.v + b0 * p.k_sb0 + b1 * p.k_sb1 + head * 64 + (int64_t)key0 * p.k_sl;
  half_t* const obase = p.out + b0 * p.o_sb0 + b1 * p.o_sb1 + head * 64;

  const int wq0 = qb * (64 * NW) + wave * 64;  // first query row of this wave
  half8_t qf[NQ][2];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const int qrow = wq0 + 16 * c + i16;
    const int qrow_c = qrow < p.lq ? qrow : p.lq - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[c][ks] = *(const half8_t*)(qbase + (int64_t)qrow_c * p.q_sl + 32 * ks + 8 * g);
  }
  // a wave whose 64 query rows all lie past lq keeps feeding the K/V ring and meeting the barriers but computes nothing
  const bool active = __builtin_amdgcn_readfirstlane(wq0) < p.lq;
  f32x4 acc_o[NQ][4];  // [query block][16-dim block]: dims 4g .. 4g+3 of query i16
  float m_run[NQ], l_run[NQ];
  f32x4 neg_m[NQ];     // -m_run: the C operand of the first score MFMA of every key block
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    m_run[c] = 0.f;
    l_run[c] = 0.f;
    neg_m[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) acc_o[c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr float P_SUM_BOUND = 16384.0f;  // a lane's partial row sum at or above 2^14: look at the maximum, rescale (as attn2_kernel)

  const int nt = (lk + KT - 1) / KT;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned smem_base =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int sr = lane >> 3, sp = lane & 7;
  const half_t* kp[IP];
  const half_t* vp[IP];
#pragma unroll
  for (int i = 0; i < IP; ++i) {
    const int row = 8 * (wave_u * IP + i) + sr;
    const int key = row < lk ? row : lk - 1;
    kp[i] = kbase + (int64_t)key * p.k_sl + k_chunk_swz(row, sp) * 8;
    vp[i] = vbase + (int64_t)key * p.k_sl + v16_chunk_swz(row, sp) * 8;
  }
  const int64_t tile_stride = (int64_t)KT * p.k_sl;
  const bool ragged = (lk % KT) != 0;
  auto issue_tile = [&](int kt, int buf) {  // tiles are issued strictly in order 0, 1, 2, ...
    const bool clamp = ragged && kt == nt - 1 && kt > 0;
#pragma unroll
    for (int i = 0; i < IP; ++i) {
      const unsigned dst = smem_base + buf * BUF_BYTES + 8 * (wave_u * IP + i) * 128;
      if (clamp) {
        const int row = 8 * (wave_u * IP + i) + sr;
        int key = kt * KT + row;
        if (key >= lk) key = lk - 1;
        const int64_t roff = (int64_t)key * p.k_sl;
        glds16_raw(kbase + roff + k_chunk_swz(row, sp) * 8, dst);
        glds16_raw(vbase + roff + v16_chunk_swz(row, sp) * 8, dst + KT * 128);
      } else {
        glds16_raw(kp[i], dst);
        glds16_raw(vp[i], dst + KT * 128);
      }
      kp[i] += tile_stride;
      vp[i] += tile_stride;
    }
  };
  const int q4 = i16 >> 2, p4 = i16 & 3;
  typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
  const fp16x2_t ones2 = {(__fp16)1.0f, (__fp16)1.0f};

  auto tile = [&](int buf, auto masked_c, int kt) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const char* const lds_k = smem + buf * BUF_BYTES;
    const char* const lds_v = lds_k + KT * 128;
    const bool more2 = kt + 2 < nt;
    if (more2) issue_tile(kt + 2, buf == 0 ? 2 : buf - 1);  // (buf + 2) % 3

    if (active) {
      // ---- S^T = K Q^T for the four query blocks off ONE K fragment; accumulators start at -m_run (q carries scale*log2e) ----
      f32x4 sc[NQ][NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int krow = 16 * kb + i16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const half8_t kf = *(const half8_t*)(lds_k + krow * 128 + (k_chunk_swz(krow, 4 * ks + g) << 4));
#pragma unroll
          for (int c = 0; c < NQ; ++c)
            sc[c][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[c][ks], ks == 0 ? neg_m[c] : sc[c][kb], 0, 0, 0);
        }
      }
      // ---- online softmax per query block (attn2_kernel's: exp2 / pack / row sums first, the maximum only on the rare path) ----
      half8_t pf[NQ][NJ];
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        if (MASKED) {
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * KT + 16 * kb + 4 * g + r >= lk) sc[c][kb][r] = -1e30f;
        }
        float ls[4];
        const auto exp_pack = [&]() {
          ls[0] = ls[1] = ls[2] = ls[3] = 0.f;  // four partial sums: short dependent dot2 chains
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            fp16x2_t pk[4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const float e0 = __builtin_amdgcn_exp2f(sc[c][2 * j + h][2 * e]);
                const float e1 = __builtin_amdgcn_exp2f(sc[c][2 * j + h][2 * e + 1]);
                pk[2 * h + e] = __builtin_amdgcn_cvt_pkrtz(e0, e1);
                ls[2 * h + e] = __builtin_amdgcn_fdot2(pk[2 * h + e], ones2, ls[2 * h + e], false);
              }
            pf[c][j] = __builtin_bit_cast(half8_t, pk);
          }
          return (ls[0] + ls[1]) + (ls[2] + ls[3]);
        };
        float tot = exp_pack();
        const bool first = kt == 0;
        if (__builtin_expect(first || __any(!(tot < P_SUM_BOUND)), 0)) {  // wave-uniform, rare (NaN / inf sums land here too)
          float mx = sc[c][0][0];
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[c][kb][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));  // the four lanes of a query: lane groups g = 0 .. 3
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float delta = first ? mx : fmaxf(mx, 0.f);
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
          m_run[c] += delta;
          l_run[c] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) neg_m[c][r] = -m_run[c];
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[c][d][r] *= alpha;
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[c][kb][r] -= delta;
          tot = exp_pack();
        }
        l_run[c] += tot;
      }
      // ---- O^T += V^T P^T for the four query blocks off ONE V^T fragment ----
      // all eight V^T fragments are ISSUED before the first MFMA (the scores' registers are free by now): read one pair ahead, as hipcc
      // orders them, every group of eight MFMAs waits for its LDS round trip.  Same-box A/B (profiles/r04_kattn16.log): per-frame
      // attention at 72x72 1462 -> 1409 us, joint at 36x36 3429 -> 3401; the same for the K fragments of the score phase changes nothing.
      half8_t vfa[4][NJ];
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int ra = 32 * j + 4 * g + q4, rb = ra + 16;
          const int colbyte = (16 * db + 4 * p4) * 2;
          const int ch = colbyte >> 4, within = colbyte & 15;
          vfa[db][j] = tr_read_pair(lds_v + ra * 128 + (v16_chunk_swz(ra, ch) << 4) + within,
                                    lds_v + rb * 128 + (v16_chunk_swz(rb, ch) << 4) + within);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int c = 0; c < NQ; ++c) acc_o[c][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfa[db][j], pf[c][j], acc_o[c][db], 0, 0, 0);
    }  // active
    if (more2) wait_vm<G>();
    else wait_vm<0>();
    __syncthreads();
  };

  issue_tile(0, 0);
  if (nt > 1) {
    issue_tile(1, 1);
    wait_vm<G>();
  } else {
    wait_vm<0>();
  }
#pragma unroll
  for (int c = 0; c < NQ; ++c)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(qf[c][ks]));  // retire the Q loads (see attn_kernel)
  __syncthreads();
  const int nfull = lk / KT;
  int buf = 0;
  for (int kt = 0; kt < nfull; ++kt) {
    tile(buf, std::false_type{}, kt);
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (nfull < nt) tile(buf, std::true_type{}, nfull);

  float l_tot[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    float l = l_run[c];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    l_tot[c] = l;
  }
  if constexpr (SPLIT) {
    // partial result: un-normalised O (relative to m_run) as fp32, 16 bytes per lane and 16-dim block, and (m_run, l) per query row
    const int64_t rows_all = (int64_t)gridDim.x / (p.qblocks * p.nsplit) * p.lq;  // batch * heads * lq
    const int64_t row_bh = ((int64_t)batch * p.heads + head) * p.lq;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const int qrow = wq0 + 16 * c + i16;
      if (qrow < p.lq) {
        float* const po = p.part_o + ((int64_t)ksp * rows_all + row_bh + qrow) * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) *(f32x4*)(po + 16 * db + 4 * g) = acc_o[c][db];
        if (g == 0) {
          float* const pm = p.part_ml + ((int64_t)ksp * rows_all + row_bh + qrow) * 2;
          pm[0] = m_run[c];
          pm[1] = l_tot[c];
        }
      }
    }
    return;
  }
  __syncthreads();  // every wave is done reading K/V tiles
  char* const ow = smem + wave * (64 * 128);
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    const float inv = 1.0f / l_tot[c];
    const int row = 16 * c + i16;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      half4_t h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc_o[c][db][r] * inv);
      const int d0 = 16 * db + 4 * g;
      const int chunk = d0 >> 3, piece = (d0 >> 2) & 1;
      *(half4_t*)(ow + row * 128 + ((chunk ^ (row & 7)) << 4) + (piece << 3)) = h;
    }
  }
  // same wave reads back its own image: no workgroup barrier needed, only the LDS round trip
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 8 * i + (lane >> 3), pchunk = lane & 7;
    const uint4 v = *(const uint4*)(ow + row * 128 + (pchunk << 4));
    const int lchunk = pchunk ^ (row & 7);
    if (wq0 + row < p.lq) *(uint4*)(obase + (int64_t)(wq0 + row) * p.o_sl + lchunk * 8) = v;
  }
}

// Combine of the K/V-split partials: out = sum_i 2^(m_i - m*) O_i / sum_i 2^(m_i - m*) l_i, one thread per (row, 8 columns).
// Fixed order over the splits, exp2 of exact differences: deterministic, and (the split being a function of the sequence
// lengths only) independent of what else is in the batch.
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs p, int64_t rows_all) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows_all * 8) return;
  const int64_t row = i >> 3;
  const int ch = (int)(i & 7);
  float m[4], l[4], mx = -3.0e38f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < p.nsplit) {
      const float* pm = p.part_ml + ((int64_t)s * rows_all + row) * 2;
      m[s] = pm[0];
      l[s] = pm[1];
      mx = fmaxf(mx, m[s]);
    }
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lt = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < p.nsplit) {
      const float w = __builtin_amdgcn_exp2f(m[s] - mx);
      lt += w * l[s];
      const float* po = p.part_o + ((int64_t)s * rows_all + row) * 64 + 8 * ch;
      const f32x4 a = first_read(*(const f32x4*)po), b = first_read(*(const f32x4*)(po + 4));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] += w * a[r];
        o[4 + r] += w * b[r];
      }
    }
  const float inv = 1.0f / lt;
  half8_t h;
#pragma unroll
  for (int r = 0; r < 8; ++r) h[r] = (half_t)(o[r] * inv);
  // row = (batch * heads + head) * lq + q
  const int64_t q = row % p.lq, bh = row / p.lq;
  const int head = (int)(bh % p.heads);
  const int64_t batch = bh / p.heads;
  const int64_t b0 = batch / p.nb1, b1 = batch - b0 * p.nb1;
  *(half8_t*)(p.out + b0 * p.o_sb0 + b1 * p.o_sb1 + head * 64 + q * p.o_sl + 8 * ch) = h;
}


template <int NW, int KT>
int launch(const AttnArgs& a, int64_t batch, hipStream_t s, bool use_tr, bool pre) {
  AttnArgs args = a;
  args.qblocks = (a.lq + 32 * NW - 1) / (32 * NW);
  const int64_t nb = batch * a.heads * args.qblocks;
  if (nb <= 0 || nb > 0x7fffffff) {
    seva_set_error("attention: bad grid %lld", (long long)nb);
    return SEVA_ERR_ARG;
  }
  const dim3 grid((unsigned)nb), block(NW * 64);
  if (args.dbg && !pre)
    hipLaunchKernelGGL((attn_kernel<NW, KT, true, true, false>), grid, block, 0, s, args);
  else if (!use_tr)
    hipLaunchKernelGGL((attn_kernel<NW, KT, false, false, false>), grid, block, 0, s, args);
  else if (pre)
    hipLaunchKernelGGL((attn_kernel<NW, KT, true, false, true>), grid, block, 0, s, args);
  else
    hipLaunchKernelGGL((attn_kernel<NW, KT, true, false, false>), grid, block, 0, s, args);
  return seva_check_launch("attn_kernel");
}

}  // namespace

extern "C" int seva_attention_f16(const seva_attn_desc* d, seva_stream_t stream) {
  SEVA_REQUIRE(d != nullptr, "attention: null desc");
  SEVA_REQUIRE(d->q && d->k && d->v && d->out, "attention: null pointer");
  SEVA_REQUIRE(d->lq > 0 && d->lk > 0 && d->heads > 0 && d->nb0 > 0 && d->nb1 > 0,
               "attention: empty problem lq=%d lk=%d heads=%d nb=%dx%d", d->lq, d->lk, d->heads,
               d->nb0, d->nb1);
  SEVA_REQUIRE(((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->out) % 16 == 0,
               "attention: pointers must be 16-byte aligned");
  SEVA_REQUIRE((d->q_sb0 | d->q_sb1 | d->q_sl | d->k_sb0 | d->k_sb1 | d->k_sl | d->o_sb0 |
                d->o_sb1 | d->o_sl) % 8 == 0,
               "attention: strides must be multiples of 8 elements");
  AttnArgs a{};
  a.q = (const half_t*)d->q; a.k = (const half_t*)d->k; a.v = (const half_t*)d->v;
  a.out = (half_t*)d->out;
  a.q_sb0 = d->q_sb0; a.q_sb1 = d->q_sb1; a.q_sl = d->q_sl;
  a.k_sb0 = d->k_sb0; a.k_sb1 = d->k_sb1; a.k_sl = d->k_sl;
  a.o_sb0 = d->o_sb0; a.o_sb1 = d->o_sb1; a.o_sl = d->o_sl;
  a.nb1 = d->nb1; a.heads = d->heads; a.lq = d->lq; a.lk = d->lk;
  a.scale_log2 = d->scale * 1.44269504088896340736f;
  a.dbg = g_seva_knobs.attn_dbg > 0 ? g_seva_knobs.attn_dbg : 0;
  const int64_t batch = (int64_t)d->nb0 * d->nb1;
  const bool use_tr = g_seva_knobs.attn_no_tr != 1;  // knob attn_no_tr = 1: scalar LDS reads instead of tr_b16 (debug)
  hipStream_t s = (hipStream_t)stream;
  const double flops = 4.0 * (double)batch * d->heads * (double)d->lq * (double)d->lk * 64.0;
  const double alg_bytes = (double)batch * d->heads * 64.0 * 2.0 * (2.0 * (double)d->lq + 2.0 * (double)d->lk);  // q, out, k, v
  SevaProfScope prof(2, flops, s, alg_bytes);
  const bool pre = d->q_prescaled != 0;
  SEVA_REQUIRE(!pre || use_tr, "attention: q_prescaled is not available on the SEVA_ATTN_NO_TR debug path");
  if (d->lq <= 32) return launch<1, 32>(a, batch, s, use_tr, pre);
  // Long sequences (lq >= 2048): 64 queries per wave, K / V fragments shared by all its query blocks.  Default: attn16_kernel
  // (v_mfma_f32_16x16x32; -3 ... -7 % against attn2_kernel on the long shapes of a step, same-box, profiles/r04_kattn16.log).
  // Knob attn_two: 0 forces attn_kernel, 1 / 2 select attn2_kernel (v_mfma_f32_32x32x16, bitwise equal to attn_kernel) from
  // lq >= 512, 4 selects attn16_kernel from lq >= 512.
  const int two = g_seva_knobs.attn_two;
  if (pre && use_tr && !a.dbg && ((two < 0 && d->lq >= 2048) || ((two == 1 || two == 2 || two == 4) && d->lq >= 512))) {
    const bool k16 = two == 4 || two < 0;
    const int qrows = 256;
    AttnArgs args = a;
    args.qblocks = (a.lq + qrows - 1) / qrows;
    const int64_t nb = batch * a.heads * args.qblocks;
    if (nb <= 0 || nb > 0x7fffffff) {
      seva_set_error("attention: bad grid %lld", (long long)nb);
      return SEVA_ERR_ARG;
    }
    // K/V split (seva_attn_desc.split_ws): two workgroups per query block for long key sequences.  The factor is a function of
    // lk alone (never of the batch); knob attn_split: 0 = never, 2..4 = that factor wherever a workspace is given.
    int nsplit = d->lk >= 6144 ? 2 : 1;
    if (g_seva_knobs.attn_split >= 0) nsplit = g_seva_knobs.attn_split > 4 ? 4 : g_seva_knobs.attn_split;
    const int nt_all = (d->lk + 63) / 64;
    while (nsplit >= 2 && (nt_all + nsplit - 1) / nsplit * (nsplit - 1) >= nt_all) --nsplit;  // every split gets >= 1 tile
    if (nsplit >= 2 && d->split_ws != nullptr) {
      const int64_t rows_all = batch * a.heads * (int64_t)a.lq;
      SEVA_REQUIRE(d->split_ws_bytes >= nsplit * rows_all * 66 * 4 && (uintptr_t)d->split_ws % 16 == 0,
                   "attention: split_ws too small (%lld bytes, need %lld) or misaligned", (long long)d->split_ws_bytes,
                   (long long)(nsplit * rows_all * 66 * 4));
      SEVA_REQUIRE(nb * nsplit <= 0x7fffffff && rows_all * 8 / 256 + 1 <= 0x7fffffff, "attention: split grid too large");
      args.nsplit = nsplit;
      args.part_o = d->split_ws;
      args.part_ml = d->split_ws + (int64_t)nsplit * rows_all * 64;
      if (k16) hipLaunchKernelGGL((attn16_kernel<64, true>), dim3((unsigned)(nb * nsplit)), dim3(256), 0, s, args);
      else hipLaunchKernelGGL((attn2_kernel<64, true>), dim3((unsigned)(nb * nsplit)), dim3(256), 0, s, args);
      int rc = seva_check_launch("attn16_kernel / attn2_kernel <split>");
      if (rc) return rc;
      hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((rows_all * 8 + 255) / 256)), dim3(256), 0, s, args, rows_all);
      return seva_check_launch("attn_combine_kernel");
    }
    if (k16) hipLaunchKernelGGL((attn16_kernel<64>), dim3((unsigned)nb), dim3(256), 0, s, args);
    else hipLaunchKernelGGL((attn2_kernel<64>), dim3((unsigned)nb), dim3(256), 0, s, args);
    return seva_check_launch("attn16_kernel / attn2_kernel");
  }
  return launch<4, 64>(a, batch, s, use_tr, pre);
}
