// Kernels of the CLIP ViT-H-14 image conditioner (reference seva/modules/conditioner.py:7-39 ->
// open_clip VisionTransformer + kornia resize).  Runs once per window on the input views (not per step), ~0.01 % of
// the path's arithmetic: everything matrix-shaped reuses the MFMA GEMM (gemm.hip) and the LayerNorm of norm.hip; the
// two kernels here are what those cannot express:
//   * clip_preprocess: kornia.geometry.resize(x, 224, bicubic, align_corners=True, antialias=True) -> (x+1)/2 ->
//     normalize(mean, std), written straight as the f16 patch matrix of the 14x14 / stride-14 patch-embedding GEMM;
//   * attn_small: softmax(q k^T / sqrt(d)) v for SHORT sequences (L = 257) and a head dim that is not 64 (d = 80).
//     fp32 VALU math out of an f16 K/V image in LDS (no MFMA: 21 MFLOP per (frame, head)).
#include "seva_common.h"

#include <atomic>

namespace {

// ---------------------------------------------------------------------------------------------------------------
struct PreArgs {
  const float* x;   // [n][3][H][W], values in [-1, 1]
  half_t* patches;  // [n * gy * gx][ldp]: column c*P*P + ky*P + kx
  int32_t n, H, W, out, P, ldp;
  int32_t ksy, ksx;         // Gaussian anti-alias kernel sizes (odd), 0 = no blur along that axis
  float gy_w[16], gx_w[16];  // normalised 1-D Gaussian taps
  float mean[3], inv_std[3];
};

__device__ __forceinline__ int reflect_idx(int i, int n) {  // torch 'reflect' padding (edge not repeated)
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}
__device__ __forceinline__ void cubic_w(float t, float w[4]) {  // torch upsample_bicubic2d, A = -0.75
  const float A = -0.75f;
  const float t1 = t + 1.f, t2 = 1.f - t, t3 = 2.f - t;
  w[0] = ((A * t1 - 5.f * A) * t1 + 8.f * A) * t1 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * t2 - (A + 3.f)) * t2 * t2 + 1.f;
  w[3] = ((A * t3 - 5.f * A) * t3 + 8.f * A) * t3 - 4.f * A;
}

__global__ __launch_bounds__(256) void clip_preprocess_kernel(PreArgs p) {
  const int64_t total = (int64_t)p.n * 3 * p.out * p.out;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % p.out), oy = (int)((idx / p.out) % p.out);
  const int c = (int)((idx / ((int64_t)p.out * p.out)) % 3), n = (int)(idx / ((int64_t)3 * p.out * p.out));
  const float* img = p.x + ((int64_t)n * 3 + c) * p.H * p.W;
  const float sy = p.out > 1 ? (float)(p.H - 1) / (float)(p.out - 1) : 0.f;
  const float sx = p.out > 1 ? (float)(p.W - 1) / (float)(p.out - 1) : 0.f;
  const float ry = sy * oy, rx = sx * ox;
  const int iy = (int)floorf(ry), ix = (int)floorf(rx);
  float wy[4], wx[4];
  cubic_w(ry - iy, wy);
  cubic_w(rx - ix, wx);
  const int hy = p.ksy / 2, hx = p.ksx / 2;
  float acc = 0.f;
  for (int a = 0; a < 4; ++a) {
    const int ty = min(max(iy - 1 + a, 0), p.H - 1);
    for (int b = 0; b < 4; ++b) {
      const int tx = min(max(ix - 1 + b, 0), p.W - 1);
      float v;
      if (p.ksy == 0) {
        v = img[(int64_t)ty * p.W + tx];
      } else {  // separable Gaussian blur (kornia gaussian_blur2d, border 'reflect') evaluated at the tap
        v = 0.f;
        for (int u = 0; u < p.ksy; ++u) {
          const float* row = img + (int64_t)reflect_idx(ty + u - hy, p.H) * p.W;
          float r = 0.f;
          for (int w = 0; w < p.ksx; ++w) r += p.gx_w[w] * row[reflect_idx(tx + w - hx, p.W)];
          v += p.gy_w[u] * r;
        }
      }
      acc += wy[a] * wx[b] * v;
    }
  }
  const float y = ((acc + 1.f) * 0.5f - p.mean[c]) * p.inv_std[c];
  const int g = p.out / p.P;
  const int py = oy / p.P, ky = oy - py * p.P, px = ox / p.P, kx = ox - px * p.P;
  p.patches[((int64_t)n * g * g + (int64_t)py * g + px) * p.ldp + (c * p.P + ky) * p.P + kx] = (half_t)y;
}

// ---------------------------------------------------------------------------------------------------------------
struct SmallAttnArgs {
  const half_t* q;
  const half_t* k;
  const half_t* v;
  half_t* out;
  int64_t q_sb, q_sl, k_sb, k_sl, o_sb, o_sl;  // batch / token strides in elements; head h at column h * D
  int32_t heads, L, D, qchunks;
  float scale_log2;
};

constexpr int SA_WAVES = 4;

__global__ __launch_bounds__(SA_WAVES * 64) void attn_small_kernel(SmallAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int DP = p.D + 2;               // padded row (halves): odd dword pitch -> conflict-free row-per-lane reads
  const int LP = (p.L + 63) & ~63;
  half_t* const Ks = (half_t*)smem;
  half_t* const Vs = Ks + (size_t)p.L * DP;
  float* const ps = (float*)(Vs + (size_t)p.L * DP + ((p.L * DP) & 1));
  float* const qs = ps + SA_WAVES * LP;
  const int bh = blockIdx.x, b = bh / p.heads, h = bh - b * p.heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const half_t* kb = p.k + (int64_t)b * p.k_sb + h * p.D;
  const half_t* vb = p.v + (int64_t)b * p.k_sb + h * p.D;
  const int dq = p.D >> 1;  // half2 per row
  for (int i = threadIdx.x; i < p.L * dq; i += blockDim.x) {
    const int r = i / dq, d2 = i - r * dq;
    *(half2_t*)(Ks + r * DP + 2 * d2) = *(const half2_t*)(kb + (int64_t)r * p.k_sl + 2 * d2);
    *(half2_t*)(Vs + r * DP + 2 * d2) = *(const half2_t*)(vb + (int64_t)r * p.k_sl + 2 * d2);
  }
  __syncthreads();
  const int rows_per_chunk = (p.L + p.qchunks - 1) / p.qchunks;
  const int r_begin = blockIdx.y * rows_per_chunk, r_end = min(p.L, r_begin + rows_per_chunk);
  float* const pw = ps + wave * LP;
  float* const qw = qs + wave * p.D;
  const int nkk = LP >> 6;
  for (int r = r_begin + wave; r < r_end; r += SA_WAVES) {
    const half_t* qr = p.q + (int64_t)b * p.q_sb + (int64_t)r * p.q_sl + h * p.D;
    for (int d = lane; d < p.D; d += 64) qw[d] = (float)qr[d] * p.scale_log2;
    // scores of keys lane, lane + 64, ...
    float mx = -1e30f;
    for (int kk = 0; kk < nkk; ++kk) {
      const int key = lane + 64 * kk;
      float s = -1e30f;
      if (key < p.L) {
        s = 0.f;
        const half_t* kr = Ks + key * DP;
        for (int d2 = 0; d2 < dq; ++d2) {
          const half2_t kv = *(const half2_t*)(kr + 2 * d2);
          s = fmaf(qw[2 * d2], (float)kv[0], s);
          s = fmaf(qw[2 * d2 + 1], (float)kv[1], s);
        }
      }
      pw[key] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int kk = 0; kk < nkk; ++kk) {
      const int key = lane + 64 * kk;
      const float e = key < p.L ? __builtin_amdgcn_exp2f(pw[key] - mx) : 0.f;
      pw[key] = e;
      sum += e;
    }
    const float inv = 1.0f / wave_sum(sum);
    half_t* orow = p.out + (int64_t)b * p.o_sb + (int64_t)r * p.o_sl + h * p.D;
    for (int d = lane; d < p.D; d += 64) {
      float acc = 0.f;
      for (int j = 0; j < p.L; ++j) acc = fmaf(pw[j], (float)Vs[j * DP + d], acc);
      orow[d] = (half_t)(acc * inv);
    }
  }
}

}  // namespace

extern "C" int seva_clip_preprocess_f16(const float* x, void* patches_f16, int32_t n, int32_t H, int32_t W,
                                        int32_t out_size, int32_t patch, int32_t ld_patches, const float* mean,
                                        const float* std, int32_t antialias, seva_stream_t stream) {
  SEVA_REQUIRE(x && patches_f16 && mean && std, "clip_preprocess: null pointer");
  SEVA_REQUIRE(n > 0 && H > 1 && W > 1 && out_size > 1 && patch > 0 && out_size % patch == 0,
               "clip_preprocess: bad shape n=%d %dx%d -> %d, patch %d", n, H, W, out_size, patch);
  SEVA_REQUIRE(ld_patches >= 3 * patch * patch, "clip_preprocess: ld_patches=%d too small", ld_patches);
  PreArgs a{};
  a.x = x; a.patches = (half_t*)patches_f16;
  a.n = n; a.H = H; a.W = W; a.out = out_size; a.P = patch; a.ldp = ld_patches;
  // kornia.geometry.resize(antialias=True): blur only when down-scaling; sigma = max((factor - 1) / 2, 0.001),
  // kernel size = int(max(4 sigma, 3)) made odd; taps exp(-x^2 / (2 sigma^2)) normalised
  const float fy = (float)H / out_size, fx = (float)W / out_size;
  a.ksy = a.ksx = 0;
  if (antialias && (fy > 1.f || fx > 1.f)) {
    const float sg[2] = {fmaxf((fy - 1.f) * 0.5f, 0.001f), fmaxf((fx - 1.f) * 0.5f, 0.001f)};
    int ks[2];
    float* wt[2] = {a.gy_w, a.gx_w};
    for (int d = 0; d < 2; ++d) {
      ks[d] = (int)fmaxf(4.f * sg[d], 3.f);
      if (ks[d] % 2 == 0) ks[d] += 1;
      SEVA_REQUIRE(ks[d] <= 15, "clip_preprocess: down-scaling factor too large (kernel %d)", ks[d]);
      float sum = 0.f;
      for (int i = 0; i < ks[d]; ++i) {
        const float t = (float)(i - ks[d] / 2);
        wt[d][i] = expf(-t * t / (2.f * sg[d] * sg[d]));
        sum += wt[d][i];
      }
      for (int i = 0; i < ks[d]; ++i) wt[d][i] /= sum;
    }
    a.ksy = ks[0];
    a.ksx = ks[1];
  }
  for (int c = 0; c < 3; ++c) {
    SEVA_REQUIRE(std[c] > 0.f, "clip_preprocess: std must be positive");
    a.mean[c] = mean[c];
    a.inv_std[c] = 1.f / std[c];
  }
  const int64_t total = (int64_t)n * 3 * out_size * out_size;
  hipStream_t s = (hipStream_t)stream;
  SevaProfScope prof(4, (double)total * 6.0, s);
  hipLaunchKernelGGL(clip_preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  return seva_check_launch("clip_preprocess_kernel");
}

extern "C" int seva_attention_small_f16(const void* q, const void* k, const void* v, void* out, int64_t q_sb,
                                        int64_t q_sl, int64_t k_sb, int64_t k_sl, int64_t o_sb, int64_t o_sl,
                                        int32_t batch, int32_t heads, int32_t L, int32_t head_dim, float scale,
                                        seva_stream_t stream) {
  SEVA_REQUIRE(q && k && v && out, "attention_small: null pointer");
  SEVA_REQUIRE(batch > 0 && heads > 0 && L > 0 && head_dim > 0 && head_dim % 2 == 0 && head_dim <= 128,
               "attention_small: bad shape batch=%d heads=%d L=%d d=%d", batch, heads, L, head_dim);
  SEVA_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 4 == 0 &&
                   (q_sb | q_sl | k_sb | k_sl | o_sb | o_sl) % 2 == 0,
               "attention_small: pointers / strides must be 4-byte aligned");
  const int DP = head_dim + 2, LP = (L + 63) & ~63;
  const size_t lds = (size_t)2 * L * DP * 2 + 4 + (size_t)SA_WAVES * (LP + head_dim) * 4;
  SEVA_REQUIRE(lds <= 160 * 1024, "attention_small: L=%d d=%d needs %zu bytes of LDS (max 160 KiB)", L, head_dim, lds);
  static std::atomic<uint64_t> attr_devs{0};  // the dynamic-LDS attribute is per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t dev_bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_relaxed) & dev_bit)) {
    (void)hipFuncSetAttribute((const void*)attn_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs.fetch_or(dev_bit, std::memory_order_relaxed);
  }
  SmallAttnArgs a{};
  a.q = (const half_t*)q; a.k = (const half_t*)k; a.v = (const half_t*)v; a.out = (half_t*)out;
  a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.o_sb = o_sb; a.o_sl = o_sl;
  a.heads = heads; a.L = L; a.D = head_dim;
  a.scale_log2 = scale * 1.44269504088896340736f;
  // enough workgroups to touch every CU: split the query rows when batch * heads is small
  int qchunks = 1;
  while ((int64_t)batch * heads * qchunks < 256 && qchunks < 8 && L / (qchunks * 2) >= 4 * SA_WAVES) qchunks *= 2;
  a.qchunks = qchunks;
  hipStream_t s = (hipStream_t)stream;
  SevaProfScope prof(2, 4.0 * batch * heads * (double)L * L * head_dim, s,
                     (double)batch * heads * head_dim * 2.0 * 4.0 * L);
  hipLaunchKernelGGL(attn_small_kernel, dim3((unsigned)(batch * heads), (unsigned)qchunks), dim3(SA_WAVES * 64), lds, s, a);
  return seva_check_launch("attn_small_kernel");
}
