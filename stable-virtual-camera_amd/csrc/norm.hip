// GroupNorm (+SiLU, +Pluecker scale/shift modulation) and LayerNorm for channels-last fp32
// activations, f16 outputs (the next GEMM's A operand).  HBM-bound: every thread moves 16-byte
// vectors, a tensor is read twice (statistics, apply) and written once as f16.
//
// GroupNorm statistics are deterministic: per-slab partial sums (fixed thread->element map, fixed
// reduction order) are written to a workspace and combined in double precision by the apply kernel.
// The input may be the channel concatenation of two tensors (UNet skip concat), read in place.
#include "seva_common.h"

#include <stdlib.h>

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_THREADS = 1024;
constexpr int GN_MAX_SLABS = 1024;  // slab slots per sample in the workspace (the last one holds the finalised statistics)
constexpr int GN_BASE_SLABS = 64;   // slab cap of images up to 8192 pixels (every UNet level)
constexpr int GN_MAX_GROUPS = 32;
constexpr int GN_MAX_DENSE = 8;

struct GnArgs {
  const float* x1;
  const float* x2;
  const float* gamma;
  const float* beta;
  const float* dense;
  const float* dense_w;
  const float* dense_b;
  half_t* out;      // f16 output (may be null when out8 is set)
  uint8_t* out8;    // optional: e4m3 output (A operand of an fp8 GEMM / conv), pixel pitch ld8 (>= C; pad bytes untouched)
  int64_t ld8;
  half_t* raw_out;  // optional: plain f16 copy of the (concatenated) input, same layout as out
  // split-precision outputs (SPLIT instantiations): pixel pitch 2C, channels [C, 2C) hold lo = f16(v - f32(f16(v))) of the value
  // whose high part f16(v) sits in [0, C).  The consumer GEMM / conv duplicates its weights over the two halves.
  int32_t split_out, split_raw;
  float* ws;
  const float* stats1;  // optional: per-channel 64-row-block partial statistics of x1 / x2 from the producing GEMM epilogue
  const float* stats2;
  int32_t n, hw, c1, c2, groups, dense_c, silu;
  int32_t nslab_stats;  // slabs used by the statistics pass
  int32_t qpb;          // apply pass, channel-split mode: quads per block (gridDim.z > 1)
  int64_t final_off;    // float offset of the finalised (mean, rstd) table inside the workspace
  float eps;
};

// thread -> (pixel lane, first quad); quads advance by `qstep`
struct GnMap {
  int cq, tpp, pl_count, pl, q0, qstep;
  bool active;
  __device__ GnMap(int C) {
    const int nthreads = blockDim.x;
    cq = C >> 2;
    tpp = cq < nthreads ? cq : nthreads;  // threads per pixel
    pl_count = nthreads / tpp;
    const int t = threadIdx.x;
    pl = t / tpp;
    q0 = t - pl * tpp;
    qstep = tpp;
    active = pl < pl_count;
  }
};

__device__ __forceinline__ f32x4 load_quad(const GnArgs& p, int n, int pix, int c) {
  // channel c (multiple of 4) of pixel pix of image n, from whichever source holds it
  if (c < p.c1) return *(const f32x4*)(p.x1 + ((int64_t)n * p.hw + pix) * p.c1 + c);
  return *(const f32x4*)(p.x2 + ((int64_t)n * p.hw + pix) * p.c2 + (c - p.c1));
}

__global__ __launch_bounds__(GN_MAX_THREADS) void gn_stats_kernel(GnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [pl_count][C] sums, then sumsq
  const int C = p.c1 + p.c2;
  const GnMap mp(C);
  const int n = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
  const int p_begin = (int)((int64_t)slab * p.hw / nslab);
  const int p_end = (int)((int64_t)(slab + 1) * p.hw / nslab);
  float* const lsum = lds;
  float* const lsq = lds + mp.pl_count * C;
  if (mp.active) {
    for (int q = mp.q0; q < mp.cq; q += mp.qstep) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
      for (int pix = p_begin + mp.pl; pix < p_end; pix += mp.pl_count) {
        const f32x4 v = first_read(load_quad(p, n, pix, q * 4));  // (summed with packed adds: the first-reader rule, seva_common.h)
        s += v;
        ss += v * v;
      }
      *(f32x4*)(lsum + mp.pl * C + q * 4) = s;
      *(f32x4*)(lsq + mp.pl * C + q * 4) = ss;
    }
  }
  __syncthreads();
  const int g = threadIdx.x;
  if (g < p.groups) {
    const int cpg = C / p.groups;
    float s = 0.f, ss = 0.f;
    for (int pl = 0; pl < mp.pl_count; ++pl)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        s += first_read(lsum[pl * C + c]);
        ss += first_read(lsq[pl * C + c]);
      }
    float* o = p.ws + (((int64_t)n * nslab + slab) * p.groups + g) * 2;
    o[0] = s;
    o[1] = ss;
  }
}

// Slab partials -> (mean, rstd) per (sample, group), once, in fp64, in a fixed order (deterministic).  Stored behind the
// slab area of the workspace (the statistics pass uses at most GN_MAX_SLABS - 1 of the slab slots per sample).
__global__ void gn_finalize_kernel(GnArgs p) {
  // one wave per (sample, group): lane l takes slabs l, l + 64, ... in order, then a fixed xor-shuffle tree in fp64
  const int n = blockIdx.x, g = blockIdx.y, l = threadIdx.x;
  const int C = p.c1 + p.c2, cpg = C / p.groups;
  double s = 0.0, ss = 0.0;
  for (int sl = l; sl < p.nslab_stats; sl += 64) {
    const float* o = p.ws + (((int64_t)n * p.nslab_stats + sl) * p.groups + g) * 2;
    s += (double)o[0];
    ss += (double)o[1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    ss += __shfl_xor(ss, off, 64);
  }
  if (l != 0) return;
  const double cnt = (double)cpg * (double)p.hw;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  float* f = p.ws + p.final_off + ((int64_t)n * p.groups + g) * 2;
  f[0] = (float)mean;
  f[1] = (float)(1.0 / sqrt(var + (double)p.eps));
}

// Same table from the per-channel partial statistics a producing GEMM / conv epilogue wrote (seva_gemm_desc.ch_stats:
// [n * hw / 64][2][c] per source): no pass over the fp32 tensors.  One 256-thread block per (sample, group); thread t takes
// the (row block, channel) items t, t + 256, ... in order (channel fastest), fp64; fixed xor tree per wave, then the four
// wave sums in order: deterministic, and a function of the sample's own blocks only.
__global__ __launch_bounds__(256) void gn_finalize_ch_kernel(GnArgs p) {
  __shared__ double red[4][2];
  const int n = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
  const int C = p.c1 + p.c2, cpg = C / p.groups;
  const int nb = p.hw >> 6;
  const int64_t rb0 = (int64_t)n * nb;
  const int total = cpg * nb;
  double s = 0.0, ss = 0.0;
  for (int idx = t; idx < total; idx += 256) {
    const int b = idx / cpg, c = g * cpg + (idx - b * cpg);
    const bool first = c < p.c1;
    const int cs = first ? p.c1 : p.c2, cl = first ? c : c - p.c1;
    const float* o = (first ? p.stats1 : p.stats2) + (rb0 + b) * 2 * cs + cl;
    s += (double)o[0];
    ss += (double)o[cs];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    ss += __shfl_xor(ss, off, 64);
  }
  if ((t & 63) == 0) {
    red[t >> 6][0] = s;
    red[t >> 6][1] = ss;
  }
  __syncthreads();
  if (t != 0) return;
  s = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
  ss = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
  const double cnt = (double)cpg * (double)p.hw;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  float* f = p.ws + p.final_off + ((int64_t)n * p.groups + g) * 2;
  f[0] = (float)mean;
  f[1] = (float)(1.0 / sqrt(var + (double)p.eps));
}

// DC6: the modulation has exactly 6 components (the Pluecker maps -- every modulated GroupNorm of the network): weight pairs and
// loops are sized for 6 at compile time (48 instead of 64 registers of weights, 6 instead of 8 packed FMAs per channel)
template <bool DENSE, bool DC6 = false, bool SPLIT = false>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(GnArgs p) {
  constexpr int ND = DC6 ? 6 : GN_MAX_DENSE;
  __shared__ float g_mean[GN_MAX_GROUPS], g_rstd[GN_MAX_GROUPS];
  const int C = p.c1 + p.c2;
  const int n = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
  const int cpg = C / p.groups;
  // thread -> (pixel lane, quad): narrow tensors pack several pixels per block, wide ones are split
  // over gridDim.z (256 quads per block); every thread owns exactly one quad of channels
  const int cq = C >> 2;
  // the host sizes the block to the channel count: blockDim.x = pl_count * cq threads when a row of quads fits
  // (several pixels per block), otherwise the quads are split evenly over gridDim.z blocks of p.qpb threads
  int pl, pl_count, q;
  bool active;
  if (gridDim.z == 1) {
    pl_count = blockDim.x / cq;
    pl = threadIdx.x / cq;
    q = threadIdx.x - pl * cq;
    active = pl < pl_count;
  } else {
    pl_count = 1;
    pl = 0;
    q = blockIdx.z * p.qpb + threadIdx.x;
    active = (int)threadIdx.x < p.qpb && q < cq;
  }
  if ((int)threadIdx.x < p.groups) {
    const float* f = p.ws + p.final_off + ((int64_t)n * p.groups + threadIdx.x) * 2;
    g_mean[threadIdx.x] = f[0];
    g_rstd[threadIdx.x] = f[1];
  }
  __syncthreads();
  if (!active) return;
  const int p_begin = (int)((int64_t)slab * p.hw / nslab);
  const int p_end = (int)((int64_t)(slab + 1) * p.hw / nslab);
  const int dc = DENSE ? (DC6 ? 6 : p.dense_c) : 0;
  {
    const int c0 = q * 4;
    // modulation weights as (scale, shift) PAIRS: one v_pk_fma_f32 per (channel, Pluecker component) instead of two
    // v_fma_f32; the "1 +" of (1 + scale) is folded into the pair's bias
    float a[4], b[4];
    f32x2 wmod[4][ND], bmod[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + r, g = c / cpg;
      a[r] = first_read(g_rstd[g]) * first_read(p.gamma[c]);
      b[r] = first_read(p.beta[c]) - first_read(g_mean[g]) * a[r];
      bmod[r] = f32x2{1.0f + (dc ? first_read(p.dense_b[c]) : 0.f), dc ? first_read(p.dense_b[C + c]) : 0.f};
#pragma unroll
      for (int j = 0; j < ND; ++j) wmod[r][j] = f32x2{0.f, 0.f};
    }
    if (DENSE && DC6) {
      // Pluecker modulation (6 components): the thread's 4 channels x 6 weights are 24 consecutive floats of the
      // scale half and 24 of the shift half: 2 x 6 coalesced 16-byte loads instead of 48 scalar ones
      f32x4 ws4[6], wh4[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        ws4[t] = first_read(*(const f32x4*)(p.dense_w + (int64_t)c0 * 6 + 4 * t));
        wh4[t] = first_read(*(const f32x4*)(p.dense_w + (int64_t)(C + c0) * 6 + 4 * t));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int e = 6 * r + j;
          wmod[r][j] = f32x2{ws4[e >> 2][e & 3], wh4[e >> 2][e & 3]};
        }
    } else if (DENSE) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < ND; ++j) {  // branch-free: clamped index, then select
          const int c = c0 + r;
          const int jj = j < dc ? j : (dc > 0 ? dc - 1 : 0);
          const float w0 = p.dense_w[(int64_t)c * dc + jj];
          const float w1 = p.dense_w[(int64_t)(C + c) * dc + jj];
          wmod[r][j] = f32x2{j < dc ? w0 : 0.f, j < dc ? w1 : 0.f};
        }
    }
    constexpr int U = DENSE ? 3 : 2;  // pixels in flight per thread (modulated: 3 waves per SIMD need 4 to keep ~12 MB in flight)
    for (int pix0 = p_begin + pl; pix0 < p_end; pix0 += U * pl_count) {
      f32x4 v[U];
      float dn[U][ND];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int pix = pix0 + u * pl_count;
        if (pix >= p_end) pix = p_end - 1;  // clamped duplicate load, store skipped below
        v[u] = load_quad(p, n, pix, c0);
        if (dc) {
          const float* dp = p.dense + ((int64_t)n * p.hw + pix) * dc;
#pragma unroll
          for (int j = 0; j < ND; ++j) dn[u][j] = (j < dc) ? dp[j] : 0.f;
        }
      }
      // The Pluecker component is broadcast into a REAL register pair (two v_mov) rather than read through op_sel from the
      // registers the load returned into: with `v_pk_fma_f32 ... op_sel` reading freshly loaded registers this kernel now and then
      // computed one (pixel, channel-of-the-quad) wrong in the last 16 lanes of a wave whenever OTHER work ran on the card at the
      // same time (a second stream or a second process; never alone).  Measured variant by variant (DESIGN.md section 4,
      // profiles/r03_groupnorm_concurrency_variants.log): plain FMAs, this form and volatile loads are clean, every op_sel form
      // shows it in 10-25 % of the launches.
      f32x2 dn2[U][ND];
      if (dc) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < ND; ++j) {
            dn2[u][j] = f32x2{dn[u][j], dn[u][j]};
            asm volatile("" : "+v"(dn2[u][j]));
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = pix0 + u * pl_count;
        half4_t h;
        float y4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = v[u][r] * a[r] + b[r];
          if (p.silu) y = silu_f(y);
          if (dc) {
            f32x2 m = bmod[r];  // (1 + scale, shift)
#pragma unroll
            for (int j = 0; j < ND; ++j) m = __builtin_elementwise_fma(wmod[r][j], dn2[u][j], m);
            y = y * m[0] + m[1];
          }
          h[r] = (half_t)y;
          y4[r] = y;
        }
        if (pix < p_end) {
          const int64_t ldo = (SPLIT && p.split_out) ? 2 * C : C, ldr = (SPLIT && p.split_raw) ? 2 * C : C;
          if (p.out) {
            *(half4_t*)(p.out + ((int64_t)n * p.hw + pix) * ldo + c0) = h;
            if (SPLIT && p.split_out) {
              const half4_t l = {(half_t)(y4[0] - (float)h[0]), (half_t)(y4[1] - (float)h[1]), (half_t)(y4[2] - (float)h[2]),
                                 (half_t)(y4[3] - (float)h[3])};
              *(half4_t*)(p.out + ((int64_t)n * p.hw + pix) * ldo + C + c0) = l;
            }
          }
          if (p.out8) *(int*)(p.out8 + ((int64_t)n * p.hw + pix) * p.ld8 + c0) = pack_fp8x4(y4[0], y4[1], y4[2], y4[3]);
          if (p.raw_out) {  // the un-normalised input as f16: A operand of the ResBlock's 1x1 skip conv
            const half4_t hr = {(half_t)v[u][0], (half_t)v[u][1], (half_t)v[u][2], (half_t)v[u][3]};
            *(half4_t*)(p.raw_out + ((int64_t)n * p.hw + pix) * ldr + c0) = hr;
            if (SPLIT && p.split_raw) {
              const half4_t lr = {(half_t)(v[u][0] - (float)hr[0]), (half_t)(v[u][1] - (float)hr[1]), (half_t)(v[u][2] - (float)hr[2]),
                                  (half_t)(v[u][3] - (float)hr[3])};
              *(half4_t*)(p.raw_out + ((int64_t)n * p.hw + pix) * ldr + C + c0) = lr;
            }
          }
        }
      }
    }
  }
}

constexpr int LN_MAXV = 20;  // float4 per lane: one 16-lane group owns a row -> C <= 1280

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One 16-lane group per row (4 rows per wave, 16 per workgroup): each load instruction moves 256
// contiguous bytes per row and a lane keeps C/64 16-byte loads in flight.
// LN_ROWS rows are walked by one 16-lane group: gamma / beta stay in registers, the next row is prefetched.  4 for big
// tensors (ds1 93 -> 79 us); 1 when that would leave fewer workgroups than ~4 per CU.
// OUT: 0 = f16, 1 = e4m3 bytes, 2 = fp32 (a LayerNorm whose result IS the residual stream: CLIP's ln_pre)
template <int NV, int LN_ROWS, int OUT = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        half_t* __restrict__ out, int64_t rows,
                                                        int c, float eps, int64_t ld_out) {  // OUT 1 / 2: `out` is an e4m3 / fp32 buffer
  const int sub = threadIdx.x & 15;
  const int cq = c >> 2;
  // group g of the block owns rows base + g, base + g + 16, ... (consecutive groups touch consecutive rows per pass)
  const int64_t base = (int64_t)blockIdx.x * (16 * LN_ROWS) + (threadIdx.x >> 4);
  constexpr bool HOIST = NV <= 10;  // C = 1280 rows would need 160 registers for gamma / beta alone: load at use
  f32x4 gm[HOIST ? NV : 1], bt[HOIST ? NV : 1];
  if (HOIST) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = sub + 16 * k < cq ? sub + 16 * k : cq - 1;
      gm[HOIST ? k : 0] = first_read(*(const f32x4*)(gamma + i * 4));
      bt[HOIST ? k : 0] = first_read(*(const f32x4*)(beta + i * 4));
    }
  }
  auto load_row = [&](int64_t row, f32x4 (&v)[NV]) {
    if (row >= rows) row = rows - 1;  // clamped duplicate: keeps the whole wave in the shuffles, store skipped
    const float* xr = x + row * c;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = sub + 16 * k < cq ? sub + 16 * k : cq - 1;
      v[k] = *(const f32x4*)(xr + i * 4);
    }
  };
  f32x4 v[NV], nxt[NV];
  load_row(base, v);
#pragma unroll
  for (int it = 0; it < LN_ROWS; ++it) {
    const int64_t row = base + 16 * it;
    if (it + 1 < LN_ROWS) load_row(row + 16, nxt);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (sub + 16 * k < cq) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    const float mean = group16_sum(s) / (float)c;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (sub + 16 * k < cq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = v[k][r] - mean;
          ss += d * d;
        }
      }
    const float rstd = rsqrtf(group16_sum(ss) / (float)c + eps);
    if (row < rows) {
      half_t* orow = out + row * ld_out;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = sub + 16 * k;
        if (i < cq) {
          const f32x4 g4 = HOIST ? gm[HOIST ? k : 0] : first_read(*(const f32x4*)(gamma + i * 4));
          const f32x4 b4 = HOIST ? bt[HOIST ? k : 0] : first_read(*(const f32x4*)(beta + i * 4));
          if constexpr (OUT != 0) {
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (v[k][r] - mean) * rstd * g4[r] + b4[r];
            if constexpr (OUT == 1) *(int*)((uint8_t*)out + row * ld_out + i * 4) = pack_fp8x4(y[0], y[1], y[2], y[3]);
            else *(f32x4*)((float*)out + row * ld_out + i * 4) = f32x4{y[0], y[1], y[2], y[3]};
          } else {
            half4_t h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)((v[k][r] - mean) * rstd * g4[r] + b4[r]);
            *(half4_t*)(orow + i * 4) = h;
          }
        }
      }
    }
    if (it + 1 < LN_ROWS) {
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] = nxt[k];
    }
  }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                           half_t* __restrict__ out, int64_t ldo,
                                                           int cols, int cols_pad, float scale_log2) {
  __shared__ float red[4];
  const float* xr = x + (int64_t)blockIdx.x * ldx;
  half_t* orow = out + (int64_t)blockIdx.x * ldo;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -1e30f;
  for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, first_read(xr[c]) * scale_log2);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += __builtin_amdgcn_exp2f(first_read(xr[c]) * scale_log2 - mx);
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = threadIdx.x; c < cols_pad; c += 256)
    orow[c] = (half_t)(c < cols ? __builtin_amdgcn_exp2f(first_read(xr[c]) * scale_log2 - mx) * inv : 0.f);
}

int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

extern "C" int seva_groupnorm_f16(const seva_groupnorm_desc* d, seva_stream_t stream) {
  SEVA_REQUIRE(d != nullptr, "groupnorm: null desc");
  SEVA_REQUIRE(d->x1 && d->gamma && d->beta && (d->out_f16 || d->out_f8) && d->workspace, "groupnorm: null pointer");
  SEVA_REQUIRE(d->n > 0 && d->hw > 0 && d->c1 > 0 && d->c2 >= 0, "groupnorm: bad shape");
  SEVA_REQUIRE(d->c2 == 0 || d->x2 != nullptr, "groupnorm: c2 > 0 needs x2");
  const int C = d->c1 + d->c2;
  SEVA_REQUIRE(d->c1 % 4 == 0 && d->c2 % 4 == 0, "groupnorm: c1=%d c2=%d must be multiples of 4", d->c1, d->c2);
  SEVA_REQUIRE(d->groups > 0 && d->groups <= GN_MAX_GROUPS && C % d->groups == 0,
               "groupnorm: %d channels / %d groups unsupported", C, d->groups);
  SEVA_REQUIRE(!d->dense || (d->dense_c > 0 && d->dense_c <= GN_MAX_DENSE && d->dense_w && d->dense_b),
               "groupnorm: bad dense modulation args (dense_c=%d)", d->dense_c);
  SEVA_REQUIRE(C <= 4 * GN_MAX_THREADS, "groupnorm: C=%d too large", C);
  GnArgs a{};
  a.x1 = d->x1; a.x2 = d->x2; a.gamma = d->gamma; a.beta = d->beta;
  a.dense = d->dense; a.dense_w = d->dense_w; a.dense_b = d->dense_b;
  a.out = (half_t*)d->out_f16; a.out8 = (uint8_t*)d->out_f8; a.raw_out = (half_t*)d->raw_f16; a.ws = d->workspace;
  a.ld8 = d->ld_out_f8 > 0 ? d->ld_out_f8 : C;
  SEVA_REQUIRE(a.ld8 >= C && a.ld8 % 4 == 0, "groupnorm: ld_out_f8=%lld invalid", (long long)a.ld8);
  a.n = d->n; a.hw = d->hw; a.c1 = d->c1; a.c2 = d->c2; a.groups = d->groups;
  a.dense_c = d->dense_c; a.silu = d->silu; a.eps = d->eps;
  a.stats1 = d->stats1; a.stats2 = d->stats2;
  a.split_out = d->split_out_f16 != 0; a.split_raw = d->split_raw_f16 != 0;
  SEVA_REQUIRE((!a.split_out || d->out_f16) && (!a.split_raw || d->raw_f16), "groupnorm: split_* needs the output it splits");
  SEVA_REQUIRE(!d->stats2 || d->stats1, "groupnorm: stats2 without stats1");
  if (d->stats1) {
    SEVA_REQUIRE(d->hw % 64 == 0, "groupnorm: producer statistics need hw %% 64 == 0 (hw=%d)", d->hw);
    SEVA_REQUIRE(d->c2 == 0 || d->stats2, "groupnorm: statistics must be given for BOTH sources (or none)");
  }
  const int cq = C / 4;
  // wide channel counts get a block of ~cq threads (one quad each) instead of an idle-heavy 256
  const int nthreads = cq <= GN_THREADS ? GN_THREADS : (cq >= GN_MAX_THREADS ? GN_MAX_THREADS : 64 * ((cq + 63) / 64));
  const int tpp = cq < nthreads ? cq : nthreads;
  const int plc = nthreads / tpp;
  // the statistics partition depends on the per-sample shape only (never on n): a sample's result
  // is bit-identical whatever else is in the batch
  // Slabs of the statistics pass: a function of the IMAGE only (never of the batch: a sample's statistics are bitwise the
  // same whatever it is batched with).  Up to 63 for the UNet's images (<= 8192 pixels: with 42 samples that is 2.6k
  // workgroups); large single images (VAE at 576x576: 331,776 pixels, batch 1) get up to 1023, otherwise 63 workgroups
  // would read a 170 MB tensor alone (measured: 37 % of a VAE decode, 290 GB/s).
  const int slab_cap = clampi(d->hw / 128, GN_BASE_SLABS - 1, GN_MAX_SLABS - 1);
  a.nslab_stats = clampi(d->hw / (plc * 16), 1, slab_cap);
  a.final_off = (int64_t)d->n * (GN_MAX_SLABS - 1) * d->groups * 2;
  // apply pass: block = whole pixels' worth of quads (cq <= 256: floor(256/cq) pixels per block, no idle tail
  // beyond the last partial wave) or an even split of the quads over gridDim.z blocks
  const int plc_apply = cq <= GN_THREADS ? GN_THREADS / cq : 1;
  const int zchunks = cq <= GN_THREADS ? 1 : (cq + GN_THREADS - 1) / GN_THREADS;
  a.qpb = (cq + zchunks - 1) / zchunks;
  int apply_threads = zchunks == 1 ? plc_apply * cq : a.qpb;
  if (apply_threads < 64) apply_threads = 64;  // the statistics prologue needs >= `groups` (<= 64) threads
  // pixels per thread: the modulated variant sets up 12 + 4 vector loads of per-channel constants per thread; sweep
  // at the headline shapes (tools/kbench.py norm, SEVA_GN_MIN_ITER): 24 is best at every level (ds1 190, ds2 117,
  // ds4 68 us; 96 left ds2 / ds4 at 158 / 97 us with 1-3 workgroups per CU)
  int min_iter = d->dense ? 24 : 4;
  if (g_seva_knobs.gn_min_iter > 0) min_iter = g_seva_knobs.gn_min_iter;
  const int nslab_apply = clampi(8192 / (d->n * zchunks), 1, clampi(d->hw / (min_iter * plc_apply), 1, 1024));
  hipStream_t s = (hipStream_t)stream;
  const double bytes = (double)d->n * d->hw * C * ((d->stats1 ? 0.0 : 4.0) + 4.0 + 2.0);
  SevaProfScope prof(3, bytes, s);
  const size_t lds = (size_t)plc * C * 2 * sizeof(float);
  int rc;
  if (a.stats1) {
    // statistics came with the data (producer epilogues): only the combine runs
    hipLaunchKernelGGL(gn_finalize_ch_kernel, dim3(d->n, d->groups), dim3(256), 0, s, a);
    rc = seva_check_launch("gn_finalize_ch_kernel");
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nslab_stats, d->n), dim3(nthreads), lds, s, a);
    rc = seva_check_launch("gn_stats_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(d->n, d->groups), dim3(64), 0, s, a);
    rc = seva_check_launch("gn_finalize_kernel");
    if (rc) return rc;
  }
  const bool split = a.split_out || a.split_raw;  // split-precision outputs: their own instantiations (the hot ones stay as they are)
  if (split && d->dense && d->dense_c == 6)
    hipLaunchKernelGGL((gn_apply_kernel<true, true, true>), dim3(nslab_apply, d->n, zchunks), dim3(apply_threads), 0, s, a);
  else if (split && !d->dense)
    hipLaunchKernelGGL((gn_apply_kernel<false, false, true>), dim3(nslab_apply, d->n, zchunks), dim3(apply_threads), 0, s, a);
  else if (split) {
    seva_set_error("groupnorm: split-precision outputs with a %d-component modulation are not instantiated", d->dense_c);
    return SEVA_ERR_ARG;
  } else if (d->dense && d->dense_c == 6)
    hipLaunchKernelGGL((gn_apply_kernel<true, true>), dim3(nslab_apply, d->n, zchunks), dim3(apply_threads), 0, s, a);
  else if (d->dense)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nslab_apply, d->n, zchunks), dim3(apply_threads), 0, s, a);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nslab_apply, d->n, zchunks), dim3(apply_threads), 0, s, a);
  return seva_check_launch("gn_apply_kernel");
}

namespace {
template <int OUT>
int layernorm_entry(const float* x, const float* gamma, const float* beta, void* out, int64_t rows, int32_t c, float eps,
                    seva_stream_t stream, int64_t ld_out = 0) {
  SEVA_REQUIRE(x && gamma && beta && out, "layernorm: null pointer");
  if (ld_out <= 0) ld_out = c;
  SEVA_REQUIRE(ld_out >= c && ld_out % 4 == 0, "layernorm: output row pitch %lld invalid", (long long)ld_out);
  SEVA_REQUIRE(rows > 0 && c > 0 && c % 4 == 0 && c <= 16 * 4 * LN_MAXV,
               "layernorm: rows=%lld c=%d unsupported", (long long)rows, c);
  hipStream_t s = (hipStream_t)stream;
  SevaProfScope prof(3, (double)rows * c * (OUT == 1 ? 5.0 : OUT == 2 ? 8.0 : 6.0), s);
  const int lr = rows >= 64 * 1024 ? 4 : 1;
  const int64_t blocks = (rows + 16 * lr - 1) / (16 * lr);
  SEVA_REQUIRE(blocks <= 0x7fffffff, "layernorm: too many rows");
  const int nv = (c / 4 + 15) / 16;
#define SEVA_LN_LAUNCH(NV)                                                                                  \
  do {                                                                                                      \
    if (lr == 4)                                                                                            \
      hipLaunchKernelGGL((layernorm_kernel<NV, 4, OUT>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma,  \
                         beta, (half_t*)out, rows, c, eps, ld_out);                                         \
    else                                                                                                    \
      hipLaunchKernelGGL((layernorm_kernel<NV, 1, OUT>), dim3((unsigned)blocks), dim3(256), 0, s, x, gamma,  \
                         beta, (half_t*)out, rows, c, eps, ld_out);                                         \
  } while (0)
  if (nv <= 2) SEVA_LN_LAUNCH(2);
  else if (nv <= 5) SEVA_LN_LAUNCH(5);
  else if (nv <= 10) SEVA_LN_LAUNCH(10);
  else SEVA_LN_LAUNCH(20);
#undef SEVA_LN_LAUNCH
  return seva_check_launch("layernorm_kernel");
}
}  // namespace

extern "C" int seva_layernorm_f16(const float* x, const float* gamma, const float* beta,
                                  void* out_f16, int64_t rows, int32_t c, float eps,
                                  seva_stream_t stream) {
  return layernorm_entry<0>(x, gamma, beta, out_f16, rows, c, eps, stream);
}

// same normalisation, output as OCP e4m3 bytes (saturating): A operand of seva_gemm_fp8
extern "C" int seva_layernorm_fp8(const float* x, const float* gamma, const float* beta,
                                  void* out_f8, int64_t rows, int32_t c, float eps, int64_t ld_out,
                                  seva_stream_t stream) {
  return layernorm_entry<1>(x, gamma, beta, out_f8, rows, c, eps, stream, ld_out);
}

// fp32 output (input and output must not overlap)
extern "C" int seva_layernorm_f32(const float* x, const float* gamma, const float* beta,
                                  float* out_f32, int64_t rows, int32_t c, float eps,
                                  seva_stream_t stream) {
  return layernorm_entry<2>(x, gamma, beta, out_f32, rows, c, eps, stream);
}

extern "C" int seva_softmax_rows_f16(const float* x, int64_t ldx, void* out_f16, int64_t ldo,
                                     int64_t rows, int32_t cols, int32_t cols_pad, float scale,
                                     seva_stream_t stream) {
  SEVA_REQUIRE(x && out_f16 && rows > 0 && cols > 0 && cols_pad >= cols && ldx >= cols && ldo >= cols_pad,
               "softmax_rows: bad args");
  SEVA_REQUIRE(rows <= 0x7fffffff, "softmax_rows: too many rows");
  hipStream_t s = (hipStream_t)stream;
  SevaProfScope prof(3, (double)rows * cols * 6.0, s);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, ldx,
                     (half_t*)out_f16, ldo, cols, cols_pad, scale * 1.44269504088896340736f);
  return seva_check_launch("softmax_rows_kernel");
}
