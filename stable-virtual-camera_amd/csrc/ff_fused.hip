// Fused GEGLU feed-forward for the narrow (C <= 320) transformer level of the UNet:
//
//   out[m][:] = W2 . ( v * gelu_erf(g) ) + b2 (+ residual[m][:]),   [v ; g] = W1 . a[m][:] + b1
//
// i.e. FeedForward of reference seva/modules/transformer.py:18-34 (GEGLU 8-15, Linear 4C -> C) in ONE kernel.  At the
// ds1 level (C = 320, M = 217,728 token rows) the two-kernel form writes the 4C-wide hidden tensor (557 MB f16) and reads it
// back 15 times per denoising step, and both kernels re-stream their weights from L2 per 128-row tile at the L2->LDS
// rate limit; here the hidden activations never leave the registers:
//
//   * 4 waves x 32 token rows (128-row workgroup tile), ONE workgroup per CU, up to 512 registers per wave:
//       A fragments of the wave's 32 rows        (C/32 x 2 x 4 = 80 VGPRs at C = 320; loaded once),
//       stage-1 accumulators of a 64-feature hidden chunk (value + gate: 2 x 8 x 4 = 64),
//       stage-2 accumulators of the whole output row block (2 x C/16 x 4 = 160; start from the residual tile).
//   * per hidden chunk (64 features = 128 interleaved W1 rows): stage 1 = C/64 K-tiles of W1 through a 2-deep LDS ring
//     (LDS-DMA, one barrier per K-tile); GEGLU in registers; with the paired row assignment of gemm.hip a lane then
//     owns 8 consecutive hidden features of its token, which is exactly the B-operand fragment of v_mfma_f32_16x16x32_f16
//     (k = 8*(lane>>4) + j): stage 2 multiplies by the chunk's W2 slice [C rows][64 cols] (second LDS ring, staged one
//     chunk ahead, spread over the chunk's K-tiles) with no shuffle and no LDS round trip of the activations.
//   * the hidden value is rounded to f16 once (as the stored tensor was), accumulation is fp32 throughout.
//
// Weight layouts are those of seva_gemm_f16: W1 [8C][C] rows interleaved in groups of 64 = [32 value | 32 gate]
// (b1 likewise), W2 [C][4C].
#include "gemm_common.h"

#include <atomic>

namespace {

struct FfArgs {
  const half_t* a;
  const half_t* w1;
  const float* b1;
  const half_t* w2;
  const float* b2;
  const float* residual;
  float* out_f32;
  half_t* out_f16;
  int64_t M, lda, ldr, ldo32, ldo16;
  int32_t tiles_m;
  // optional LayerNorm prologue (8-wave kernel): a = LayerNorm(ln_x) computed in registers, `a` unused
  const float* ln_x;
  const float* ln_gamma;
  const float* ln_beta;
  int64_t ldx;
  float ln_eps;
};

template <int N>
__device__ __forceinline__ void ff_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// ---------------------------------------------------------------------------------------------------------------
// 8-wave variant: the same 128-row tile, but TWO waves per 32-row group (2 waves per SIMD -> one wave's LDS / barrier waits
// and GEGLU VALU run under the other's MFMAs; the 4-wave kernel above measured 29 % MFMA-busy with 40 % of its wave time in
// waits and 25 % in VALU, profiles/r02_pmc_ff_fused_4wave.txt).  Wave (mr, h): rows 32*mr .. +31; in stage 1 it takes the
// chunk's GEGLU group q = h (64 of the 128 interleaved W1 rows -> 32 hidden features), the partner's 32 features arrive
// through a 1 KiB-per-fragment LDS exchange; in stage 2 it owns output blocks h*NJ2/2 .. (half of the output row).
// Registers per wave: A 80 + stage-2 accumulators 80 + stage-1 accumulators 32 + fragments: <= 256.
template <int C>
__global__ __launch_bounds__(512, 2) void ff_fused8_kernel(FfArgs p) {
  constexpr int KS = C / 32, NK1 = C / 64, NJ2 = C / 16, NJH = NJ2 / 2, NCH = C / 16;
  constexpr int W1_BYTES = 128 * 128, W2_BYTES = C * 128;
  constexpr int W2_PASSES = C / 64;  // 8-row passes per wave and slice (C/8 rows per wave)
  constexpr int W2_PER_KT = (W2_PASSES + NK1 - 1) / NK1;
  static_assert(NJ2 % 2 == 0, "output blocks split over the two waves of a row group");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [W1 ring: 3 tiles][W2 buf0][W2 buf1][b1: 8C floats][exchange: 4 row groups x 2 halves x 2 token blocks x 1 KiB]
  // W1 K-tiles form ONE sequence over all chunks (tile g = hc * NK1 + kt, buffer g % 3) staged TWO tiles ahead: the L2 ->
  // LDS latency of a tile gets two K-tiles of MFMA time, and the wait at the end of tile g is a counted vmcnt that leaves
  // the loads issued during tile g in flight (with a 2-deep ring and vmcnt(0) every K-tile paid that latency in full:
  // 12.2k cycles per chunk for 3.8k cycles of MFMA).
  constexpr int W1_RING = 3;
  char* const lds_w1 = smem;
  char* const lds_w2 = smem + W1_RING * W1_BYTES;
  float* const lds_b1 = (float*)(smem + W1_RING * W1_BYTES + 2 * W2_BYTES);
  char* const lds_x = smem + W1_RING * W1_BYTES + 2 * W2_BYTES + 8 * C * 4;
  const unsigned lds_base_u32 = __builtin_amdgcn_readfirstlane(lds_addr_u32(smem));

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mr = wave & 3, h = wave >> 2;
  const int fr = lane & 15, fg = lane >> 4;
  const int sr = lane >> 3, sp = lane & 7;

  const int tm = xcd_remap(blockIdx.x, p.tiles_m);
  const int64_t m0 = (int64_t)tm * 128 + mr * 32;

  half8_t areg[2][KS];
  if (p.ln_x) {
    // LayerNorm prologue (reference transformer.py:102-104: x = ff(norm3(x)) + x): the wave's 32 rows are read as fp32, a
    // row lives in the 4 lanes (fr, fg = 0..3) that hold its k = 32ks + 8fg + j; exact two-pass statistics with two
    // xor-shuffles; the normalised row goes straight into the A fragments (one f16 rounding, as the LayerNorm kernel's
    // output had).  Replaces a 95 us read-4B/write-2B pass + its re-read per feed-forward at the ds1 level.
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t m = m0 + 16 * i + fr;
      if (m >= p.M) m = p.M - 1;
      const float* xr = p.ln_x + m * p.ldx + 8 * fg;
      f32x4 v[KS][2];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        v[ks][0] = first_read(*(const f32x4*)(xr + 32 * ks));
        v[ks][1] = first_read(*(const f32x4*)(xr + 32 * ks + 4));
      }
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 2; ++e) sum += v[ks][e][0] + v[ks][e][1] + v[ks][e][2] + v[ks][e][3];
      // (each cross-lane result is first read by a plain 32-bit op: see the note at the GroupNorm modulation, norm.hip)
      sum += __shfl_xor(sum, 16, 64);
      asm volatile("" : "+v"(sum));
      sum += __shfl_xor(sum, 32, 64);
      asm volatile("" : "+v"(sum));
      const float mean = sum / (float)C;
      float ss = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float dlt = v[ks][e][r] - mean;
            ss += dlt * dlt;
          }
      ss += __shfl_xor(ss, 16, 64);
      asm volatile("" : "+v"(ss));
      ss += __shfl_xor(ss, 32, 64);
      asm volatile("" : "+v"(ss));
      const float rstd = rsqrtf(ss / (float)C + p.ln_eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        half_t y[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x4 g4 = first_read(*(const f32x4*)(p.ln_gamma + 32 * ks + 8 * fg + 4 * e));
          const f32x4 b4 = first_read(*(const f32x4*)(p.ln_beta + 32 * ks + 8 * fg + 4 * e));
#pragma unroll
          for (int r = 0; r < 4; ++r) y[4 * e + r] = (half_t)((v[ks][e][r] - mean) * rstd * g4[r] + b4[r]);
        }
        areg[i][ks] = half8_t{y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7]};
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t m = m0 + 16 * i + fr;
      if (m >= p.M) m = p.M - 1;
      const half_t* ap = p.a + m * p.lda + 8 * fg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) areg[i][ks] = *(const half8_t*)(ap + 32 * ks);
    }
  }
  f32x4 acc2[2][NJH];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t m = m0 + 16 * i + fr;
    if (m >= p.M) m = p.M - 1;
#pragma unroll
    for (int jj = 0; jj < NJH; ++jj)
      acc2[i][jj] = p.residual ? *(const f32x4*)(p.residual + m * p.ldr + 16 * (h * NJH + jj) + 4 * fg) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = threadIdx.x; i < 2 * C; i += 512) *(f32x4*)(lds_b1 + 4 * i) = *(const f32x4*)(p.b1 + 4 * i);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(areg[i][ks]));
#pragma unroll
    for (int jj = 0; jj < NJH; ++jj) asm volatile("" : "+v"(acc2[i][jj]));
  }

  auto w1_key = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  const half_t* w1_src[2];  // W1 tile: wave w stages rows 16w .. 16w+15 (2 passes)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 16 + 8 * i + sr;
    w1_src[i] = p.w1 + (int64_t)row * C + (sp ^ w1_key(row)) * 8;
  }
  const half_t* w2_src[2];  // W2 slice: wave w stages rows (C/8)w .. ; passes two apart share the swizzle key
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * (C / 8) + 8 * i + sr;
    w2_src[i] = p.w2 + (int64_t)row * (4 * C) + (sp ^ ((row >> 1) & 7)) * 8;
  }
  auto stage_w1 = [&](int buf, int hc, int kt) {
    const unsigned dst = lds_base_u32 + buf * W1_BYTES + wave * 16 * 128;
    const int64_t off = (int64_t)hc * 128 * C + kt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16_raw(w1_src[i] + off, dst + i * 1024);
  };
  auto stage_w2_part = [&](int buf, int hc, int part) {
    const unsigned dst = lds_base_u32 + W1_RING * W1_BYTES + buf * W2_BYTES + wave * (C / 8) * 128;
#pragma unroll
    for (int u = 0; u < W2_PER_KT; ++u) {
      const int i = part * W2_PER_KT + u;
      if (i < W2_PASSES) glds16_raw(w2_src[i & 1] + (int64_t)(i >> 1) * 16 * (4 * C) + hc * 64, dst + i * 1024);
    }
  };
  auto barrier_raw = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  int w1_off[2], w2_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int rb = 64 * h + 8 * (fr >> 2) + (fr & 3);  // this half's GEGLU group: tile rows 64h ..
    w1_off[s] = rb * 128 + (((4 * s + fg) ^ w1_key(rb)) << 4);
    const int r2 = 16 * (h * NJH) + fr;
    w2_off[s] = r2 * 128 + (((4 * s + fg) ^ ((r2 >> 1) & 7)) << 4);
  }
  char* const x_mine = lds_x + ((mr * 2 + h) * 2) * 1024 + lane * 16;
  const char* const x_q0 = lds_x + (mr * 2 * 2) * 1024 + lane * 16;  // half 0's two fragments, then half 1's

  constexpr int NTILES = NCH * NK1;
  auto stage_w1_seq = [&](int g) { stage_w1(g % W1_RING, g / NK1, g % NK1); };
  auto wait_all_but = [&](int n) {  // wave-uniform count of the loads this wave issued during the current K-tile
    if (n >= 3) ff_wait_vm<3>();
    else if (n == 2) ff_wait_vm<2>();
    else if (n == 1) ff_wait_vm<1>();
    else ff_wait_vm<0>();
  };
  __syncthreads();
  stage_w1_seq(0);
#pragma unroll
  for (int part = 0; part < NK1; ++part) stage_w2_part(0, 0, part);
  if (NTILES > 1) stage_w1_seq(1);
  if (NTILES > 1) ff_wait_vm<2>();  // tile 0 and the first W2 slice have landed; tile 1 (2 loads) flies on
  else ff_wait_vm<0>();
  barrier_raw();

  int g = 0;  // W1 tile sequence number
  for (int hc = 0; hc < NCH; ++hc) {
    // stage-1 accumulators start from the bias (blocks 0,1 = value e = 0,1; blocks 2,3 = gate): no bias registers or adds later
    f32x4 acc1[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* bp = lds_b1 + hc * 128 + 64 * h + 8 * fg + 4 * e;
      acc1[0][e] = acc1[1][e] = *(const f32x4*)bp;
      acc1[0][2 + e] = acc1[1][2 + e] = *(const f32x4*)(bp + 32);
    }
#pragma unroll
    for (int kt = 0; kt < NK1; ++kt) {
      int issued = 0;
      if (g + 2 < NTILES) {
        stage_w1_seq(g + 2);
        issued += 2;
      }
      if (hc + 1 < NCH && kt * W2_PER_KT < W2_PASSES) {
        stage_w2_part((hc + 1) & 1, hc + 1, kt);
        issued += (kt + 1) * W2_PER_KT <= W2_PASSES ? W2_PER_KT : W2_PASSES - kt * W2_PER_KT;
      }
      const char* const tb = lds_w1 + (g % W1_RING) * W1_BYTES;
      half8_t bfr[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[s][j] = *(const half8_t*)(tb + w1_off[s] + (32 * (j >> 1) + 4 * (j & 1)) * 128);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[s][j], areg[i][2 * kt + s], acc1[i][j], 0, 0, 0);
      wait_all_but(issued);  // tile g + 1 (issued a K-tile ago) has landed; this K-tile's loads stay in flight
      barrier_raw();
      ++g;
    }
    // GEGLU of this wave's 32 hidden features (group q = h of the chunk), handed to the partner through LDS
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 o0 = geglu4(acc1[i][0], acc1[i][2]);
      const f32x4 o1 = geglu4(acc1[i][1], acc1[i][3]);
      *(half8_t*)(x_mine + i * 1024) = half8_t{(half_t)o0[0], (half_t)o0[1], (half_t)o0[2], (half_t)o0[3],
                                               (half_t)o1[0], (half_t)o1[1], (half_t)o1[2], (half_t)o1[3]};
    }
    barrier_raw();  // both halves of every row group have published their 32 features
    half8_t hq[2][2];  // [k-step q = half that produced the features][token block i], both read back from LDS
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) hq[q][i] = *(const half8_t*)(x_q0 + (q * 2 + i) * 1024);
    // stage 2: this wave's half of the output blocks, k-step q = 0 / 1 <- hidden features of half 0 / 1
    const char* const t2 = lds_w2 + (hc & 1) * W2_BYTES;
#pragma unroll
    for (int jj = 0; jj < NJH; ++jj) {
      const half8_t w0 = *(const half8_t*)(t2 + w2_off[0] + jj * (16 * 128));
      const half8_t w1f = *(const half8_t*)(t2 + w2_off[1] + jj * (16 * 128));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc2[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, hq[0][i], acc2[i][jj], 0, 0, 0);
        acc2[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f, hq[1][i], acc2[i][jj], 0, 0, 0);
      }
    }
    barrier_raw();  // W2 buffer (hc & 1) and the exchange slots are rewritten during the next chunk
  }

#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + 16 * i + fr;
    const bool ok = m < p.M;
#pragma unroll
    for (int jj = 0; jj < NJH; ++jj) {
      const int f = 16 * (h * NJH + jj) + 4 * fg;
      const f32x4 v = acc2[i][jj] + first_read(*(const f32x4*)(p.b2 + f));
      if (!ok) continue;
      if (p.out_f32) *(f32x4*)(p.out_f32 + m * p.ldo32 + f) = v;
      if (p.out_f16) {
        const half4_t hh = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *(half4_t*)(p.out_f16 + m * p.ldo16 + f) = hh;
      }
    }
  }
}


template <int C>
int ff_launch8(const FfArgs& a, hipStream_t s) {
  constexpr int lds = 3 * 128 * 128 + 2 * C * 128 + 8 * C * 4 + 16 * 1024;
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_relaxed) & bit)) {
    (void)hipFuncSetAttribute((const void*)ff_fused8_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_devs.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((ff_fused8_kernel<C>), dim3((unsigned)a.tiles_m), dim3(512), lds, s, a);
  return seva_check_launch("ff_fused8_kernel");
}

}  // namespace

extern "C" int seva_ff_fused_f16(const seva_ff_desc* d, seva_stream_t stream) {
  SEVA_REQUIRE(d != nullptr, "ff_fused: null desc");
  SEVA_REQUIRE((d->a || d->ln_x) && d->w1 && d->b1 && d->w2 && d->b2, "ff_fused: null operand");
  SEVA_REQUIRE(!d->ln_x || (d->ln_gamma && d->ln_beta && d->ldx >= d->C && d->ldx % 4 == 0 &&
                            ((uintptr_t)d->ln_x | (uintptr_t)d->ln_gamma | (uintptr_t)d->ln_beta) % 16 == 0),
               "ff_fused: LayerNorm prologue needs gamma, beta, a row pitch >= C (multiple of 4), 16-byte aligned pointers");
  SEVA_REQUIRE(d->out_f32 || d->out_f16, "ff_fused: no output");
  SEVA_REQUIRE(d->M > 0, "ff_fused: empty problem");
  SEVA_REQUIRE(d->C == 64 || d->C == 128 || d->C == 256 || d->C == 320, "ff_fused: C=%d unsupported (64, 128, 256, 320)", d->C);
  SEVA_REQUIRE(d->ln_x || (d->lda >= d->C && d->lda % 8 == 0), "ff_fused: lda=%lld invalid", (long long)d->lda);
  SEVA_REQUIRE((!d->residual || d->ldr % 4 == 0) && (!d->out_f32 || d->ldo32 % 4 == 0) && (!d->out_f16 || d->ldo16 % 4 == 0),
               "ff_fused: row pitches must be multiples of 4");
  SEVA_REQUIRE(((uintptr_t)d->a | (uintptr_t)d->w1 | (uintptr_t)d->b1 | (uintptr_t)d->w2 | (uintptr_t)d->b2 |
                (uintptr_t)d->residual | (uintptr_t)d->out_f32 | (uintptr_t)d->out_f16) % 16 == 0,
               "ff_fused: pointers must be 16-byte aligned");
  FfArgs a{};
  a.a = (const half_t*)d->a; a.w1 = (const half_t*)d->w1; a.b1 = d->b1; a.w2 = (const half_t*)d->w2; a.b2 = d->b2;
  a.residual = d->residual; a.out_f32 = d->out_f32; a.out_f16 = (half_t*)d->out_f16;
  a.M = d->M; a.lda = d->lda; a.ldr = d->ldr; a.ldo32 = d->ldo32; a.ldo16 = d->ldo16;
  a.ln_x = d->ln_x; a.ln_gamma = d->ln_gamma; a.ln_beta = d->ln_beta; a.ldx = d->ldx; a.ln_eps = d->ln_eps;
  const int64_t tiles = (d->M + 127) / 128;
  SEVA_REQUIRE(tiles <= 0x7fffffff, "ff_fused: too many rows");
  a.tiles_m = (int)tiles;
  hipStream_t s = (hipStream_t)stream;
  const double C = (double)d->C;
  const double flops = 2.0 * (double)d->M * C * (8.0 * C) + 2.0 * (double)d->M * (4.0 * C) * C;
  const double bytes = (double)d->M * C * ((d->ln_x ? 4.0 : 2.0) + (d->residual ? 4.0 : 0.0) + (d->out_f32 ? 4.0 : 0.0) + (d->out_f16 ? 2.0 : 0.0)) +
                       2.0 * (8.0 * C * C + 4.0 * C * C);
  SevaProfScope prof(0, flops, s, bytes);
  switch (d->C) {
    case 64: return ff_launch8<64>(a, s);
    case 128: return ff_launch8<128>(a, s);
    case 256: return ff_launch8<256>(a, s);
    default: return ff_launch8<320>(a, s);
  }
}
