// Layout changes and small elementwise kernels (HBM- or launch-bound; everything vectorised where
// the layout allows).  See include/seva_hip.h for the reference lines each one replaces.
#include "seva_common.h"

namespace {

constexpr int EW_THREADS = 256;

inline unsigned grid_for(int64_t work, int per_block = EW_THREADS, int64_t cap = 8192) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// [n][c][hw] (two sources) -> [n][hw][cpad] f16; one thread per pixel writes whole 16-byte chunks
// split != 0: channels [C, 2C) (C = c1 + c2) receive the LOW part of the same value, lo = f16(v - f32(f16(v))): with the conv
// weights duplicated over those channels the fp32-accumulating MFMA sees the operand to ~22 bits at no extra K-tile
// (the stem has 11 real channels in a 64-channel K-tile).
__global__ void nchw_to_nhwc_f16_kernel(const float* __restrict__ x1, int c1,
                                        const float* __restrict__ x2, int c2,
                                        const float* __restrict__ scale, half_t* __restrict__ out,
                                        int n, int hw, int cpad, int split) {
  const int64_t total = (int64_t)n * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / hw), pix = (int)(i - (int64_t)img * hw);
    const float sc = scale ? scale[img] : 1.0f;
    half_t* o = out + i * cpad;
    for (int c0 = 0; c0 < cpad; c0 += 8) {
      half8_t h;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int C = c1 + c2;
        const bool lo = split && c0 + r >= C && c0 + r < 2 * C;
        const int c = lo ? c0 + r - C : c0 + r;
        float v = 0.f;
        if (c < c1) v = x1[((int64_t)img * c1 + c) * hw + pix] * sc;
        else if (c < C) v = x2[((int64_t)img * c2 + (c - c1)) * hw + pix];
        h[r] = lo ? (half_t)(v - (float)(half_t)v) : (half_t)v;
      }
      *(half8_t*)(o + c0) = h;
    }
  }
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ x, int64_t ld,
                                        float* __restrict__ out, int n, int c, int hw) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % hw);
    const int64_t t = i / hw;
    const int ch = (int)(t % c), img = (int)(t / c);
    out[i] = x[((int64_t)img * hw + pix) * ld + ch];
  }
}

__global__ void cast_concat_f16_kernel(const float* __restrict__ x1, int c1,
                                       const float* __restrict__ x2, int c2,
                                       half_t* __restrict__ out, int64_t rows) {
  const int cq = (c1 + c2) >> 2;
  const int64_t total = rows * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cq;
    const int c = (int)(i - row * cq) * 4;
    const f32x4 v = (c < c1) ? *(const f32x4*)(x1 + row * c1 + c)
                             : *(const f32x4*)(x2 + row * c2 + (c - c1));
    half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *(half4_t*)(out + row * (c1 + c2) + c) = h;
  }
}

// F.interpolate(mode="bilinear", align_corners=True): src = dst * (in-1)/(out-1)
__global__ void bilinear_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ out,
                                        int n, int c, int sh, int sw, int oh, int ow) {
  const float ry = oh > 1 ? (float)(sh - 1) / (float)(oh - 1) : 0.f;
  const float rx = ow > 1 ? (float)(sw - 1) / (float)(ow - 1) : 0.f;
  const int64_t total = (int64_t)n * oh * ow * c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    int64_t t = i / c;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh), img = (int)(t / oh);
    const float fy = ry * oy, fx = rx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float* s = src + ((int64_t)img * c + ch) * sh * sw;
    const float v = (1.f - ly) * ((1.f - lx) * s[y0 * sw + x0] + lx * s[y0 * sw + x1]) +
                    ly * ((1.f - lx) * s[y1 * sw + x0] + lx * s[y1 * sw + x1]);
    out[i] = v;
  }
}

__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t,
                                          const float* __restrict__ freqs,
                                          half_t* __restrict__ out, int n, int dim) {
  const int half = dim >> 1;
  const int total = n * dim;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / dim, j = i - row * dim;
    float v = 0.f;
    if (j < 2 * half) {
      const float arg = (float)t[row] * freqs[j < half ? j : j - half];
      v = j < half ? cosf(arg) : sinf(arg);
    }
    out[i] = (half_t)v;
  }
}

__global__ void silu_f16_kernel(const float* __restrict__ x, half_t* __restrict__ out,
                                int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (half_t)silu_f(x[i]);
}

__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                               float* __restrict__ out, int64_t count4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4;
       i += (int64_t)gridDim.x * blockDim.x)
    ((f32x4*)out)[i] = first_read(((const f32x4*)a)[i]) + first_read(((const f32x4*)b)[i]);
}

__global__ void replace_blend_kernel(const float* __restrict__ x, const float* __restrict__ rep,
                                     float* __restrict__ out, int n, int c, int hw) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % hw);
    const int64_t t = i / hw;
    const int ch = (int)(t % c), img = (int)(t / c);
    const float* r = rep + (int64_t)img * (c + 1) * hw;
    const float m = r[(int64_t)c * hw + pix];
    out[i] = x[i] * (1.f - m) + r[(int64_t)ch * hw + pix] * m;
  }
}

__global__ void denoiser_combine_kernel(const float* __restrict__ net, const float* __restrict__ x,
                                        const float* __restrict__ c_out,
                                        const float* __restrict__ c_skip, float* __restrict__ out,
                                        int n, int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / chw);
    out[i] = net[i] * c_out[img] + x[i] * c_skip[img];
  }
}

__global__ void add_noise_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ ns, float* __restrict__ out, int n,
                                 int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / chw);
    out[i] = x[i] + eps[i] * ns[img];
  }
}

__global__ void cfg_euler_kernel(const float* __restrict__ x, const float* __restrict__ den2,
                                 const float* __restrict__ scale,
                                 const float* __restrict__ sigma_hat, const float* __restrict__ dt,
                                 float* __restrict__ out, int n, int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / chw);
    const float u = den2[i], c = den2[total + i];
    const float den = u + scale[img] * (c - u);
    const float xv = x[i];
    const float d = (xv - den) / sigma_hat[img];
    out[i] = xv + dt[img] * d;
  }
}

__global__ void cfg_combine_kernel(const float* __restrict__ den2, const float* __restrict__ scale,
                                   float* __restrict__ out, int n, int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float u = den2[i], c = den2[total + i];
    out[i] = u + scale[i / chw] * (c - u);
  }
}

__global__ void euler_step_kernel(const float* __restrict__ x, const float* __restrict__ den,
                                  const float* __restrict__ sigma_hat, const float* __restrict__ dt,
                                  float* __restrict__ out, int n, int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / chw);
    const float xv = x[i];
    out[i] = xv + dt[img] * ((xv - den[i]) / sigma_hat[img]);
  }
}

__global__ void to_d_kernel(const float* __restrict__ x, const float* __restrict__ den,
                            const float* __restrict__ sigma, float* __restrict__ out, int n,
                            int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (x[i] - den[i]) / sigma[i / chw];
}

__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                  float* __restrict__ out, int n, int64_t chw) {
  const int64_t total = (int64_t)n * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[i] * s[i / chw];
}

// ---- conditioning geometry -------------------------------------------------------------------
// Pluecker ray map of one latent pixel per thread, the arithmetic of seva/geometry.py:82-117,160-165 in its
// order: grid (x+.5, y+.5, 1) -> camera (K^-1) -> source-camera frame (pose^-1, homogeneous) ; ray = point -
// centre (kept as a subtraction, like the reference, so the rounding matches) ; normalise ; moment = centre x ray.
__global__ void plucker_kernel(const float* __restrict__ kinv, const float* __restrict__ pose_inv,
                               float* __restrict__ out, int h, int w) {
  const int v = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= h * w) return;
  const float* ki = kinv + v * 9;
  const float* pi = pose_inv + v * 12;
  const int y = pix / w, x = pix - y * w;
  const float gx = (float)x + 0.5f, gy = (float)y + 0.5f;
  float cam[3], wpt[3], ctr[3], ray[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) cam[i] = gx * ki[3 * i] + gy * ki[3 * i + 1] + ki[3 * i + 2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    wpt[i] = cam[0] * pi[4 * i] + cam[1] * pi[4 * i + 1] + cam[2] * pi[4 * i + 2] + pi[4 * i + 3];
    ctr[i] = pi[4 * i + 3];
    ray[i] = wpt[i] - ctr[i];
  }
  const float nrm = sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
  const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps
#pragma unroll
  for (int i = 0; i < 3; ++i) ray[i] *= inv;
  const float m0 = ctr[1] * ray[2] - ctr[2] * ray[1];
  const float m1 = ctr[2] * ray[0] - ctr[0] * ray[2];
  const float m2 = ctr[0] * ray[1] - ctr[1] * ray[0];
  float* o = out + (int64_t)v * 6 * h * w + pix;
  const int64_t cs = (int64_t)h * w;
  o[0] = ray[0]; o[cs] = ray[1]; o[2 * cs] = ray[2];
  o[3 * cs] = m0; o[4 * cs] = m1; o[5 * cs] = m2;
}

// concat channels of do_sample (seva/eval.py:1255-1270): c = [mask_v | plucker_v], uc = [0 | plucker_v]
__global__ void cond_concat_kernel(const float* __restrict__ plucker, const unsigned char* __restrict__ mask,
                                   float* __restrict__ c_concat, float* __restrict__ uc_concat, int hw) {
  const int v = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 7*hw
  if (i >= 7 * hw) return;
  const int ch = i / hw;
  const float val = ch == 0 ? 0.f : plucker[(int64_t)v * 6 * hw + (i - hw)];
  c_concat[(int64_t)v * 7 * hw + i] = ch == 0 ? (mask[v] ? 1.f : 0.f) : val;
  uc_concat[(int64_t)v * 7 * hw + i] = val;
}

}  // namespace

#define EW_LAUNCH(kern, work, ...)                                                         \
  do {                                                                                     \
    hipStream_t s_ = (hipStream_t)stream;                                                  \
    hipLaunchKernelGGL(kern, dim3(grid_for(work)), dim3(EW_THREADS), 0, s_, __VA_ARGS__); \
    return seva_check_launch(#kern);                                                       \
  } while (0)

extern "C" int seva_nchw_to_nhwc_f16(const float* x1, int32_t c1, const float* x2, int32_t c2,
                                     const float* scale, void* out_f16, int32_t n, int32_t hw,
                                     int32_t cpad, seva_stream_t stream) {
  SEVA_REQUIRE(x1 && out_f16 && n > 0 && hw > 0 && c1 > 0, "nchw_to_nhwc: bad args");
  SEVA_REQUIRE(c2 == 0 || x2, "nchw_to_nhwc: c2 > 0 needs x2");
  SEVA_REQUIRE(cpad % 8 == 0 && cpad >= c1 + c2, "nchw_to_nhwc: cpad=%d (c=%d)", cpad, c1 + c2);
  SevaProfScope prof(4, (double)n * hw * ((c1 + c2) * 4.0 + cpad * 2.0), (hipStream_t)stream);
  EW_LAUNCH(nchw_to_nhwc_f16_kernel, (int64_t)n * hw, x1, c1, x2, c2, scale, (half_t*)out_f16, n,
            hw, cpad, 0);
}

extern "C" int seva_nchw_to_nhwc_f16_split(const float* x1, int32_t c1, const float* x2, int32_t c2,
                                           const float* scale, void* out_f16, int32_t n, int32_t hw,
                                           int32_t cpad, seva_stream_t stream) {
  SEVA_REQUIRE(x1 && out_f16 && n > 0 && hw > 0 && c1 > 0, "nchw_to_nhwc_split: bad args");
  SEVA_REQUIRE(c2 == 0 || x2, "nchw_to_nhwc_split: c2 > 0 needs x2");
  SEVA_REQUIRE(cpad % 8 == 0 && cpad >= 2 * (c1 + c2), "nchw_to_nhwc_split: cpad=%d must hold 2 x %d channels", cpad, c1 + c2);
  SevaProfScope prof(4, (double)n * hw * ((c1 + c2) * 4.0 + cpad * 2.0), (hipStream_t)stream);
  EW_LAUNCH(nchw_to_nhwc_f16_kernel, (int64_t)n * hw, x1, c1, x2, c2, scale, (half_t*)out_f16, n,
            hw, cpad, 1);
}

extern "C" int seva_nhwc_to_nchw_f32(const float* x, int64_t ld, float* out, int32_t n, int32_t c,
                                     int32_t hw, seva_stream_t stream) {
  SEVA_REQUIRE(x && out && n > 0 && c > 0 && hw > 0 && ld >= c, "nhwc_to_nchw: bad args");
  SevaProfScope prof(4, (double)n * hw * c * 8.0, (hipStream_t)stream);
  EW_LAUNCH(nhwc_to_nchw_f32_kernel, (int64_t)n * c * hw, x, ld, out, n, c, hw);
}

extern "C" int seva_cast_concat_f16(const float* x1, int32_t c1, const float* x2, int32_t c2,
                                    void* out_f16, int64_t rows, seva_stream_t stream) {
  SEVA_REQUIRE(x1 && out_f16 && rows > 0 && c1 > 0 && c1 % 4 == 0 && c2 % 4 == 0 && c2 >= 0,
               "cast_concat: bad args c1=%d c2=%d", c1, c2);
  SEVA_REQUIRE(c2 == 0 || x2, "cast_concat: c2 > 0 needs x2");
  SevaProfScope prof(4, (double)rows * (c1 + c2) * 6.0, (hipStream_t)stream);
  EW_LAUNCH(cast_concat_f16_kernel, rows * ((c1 + c2) / 4), x1, c1, x2, c2, (half_t*)out_f16, rows);
}

extern "C" int seva_bilinear_to_nhwc_f32(const float* src, float* out, int32_t n, int32_t c,
                                         int32_t sh, int32_t sw, int32_t oh, int32_t ow,
                                         seva_stream_t stream) {
  SEVA_REQUIRE(src && out && n > 0 && c > 0 && sh > 0 && sw > 0 && oh > 0 && ow > 0,
               "bilinear: bad args");
  SevaProfScope prof(4, (double)n * oh * ow * c * 8.0, (hipStream_t)stream);
  EW_LAUNCH(bilinear_to_nhwc_kernel, (int64_t)n * oh * ow * c, src, out, n, c, sh, sw, oh, ow);
}

extern "C" int seva_timestep_embedding_f16(const int64_t* t, const float* freqs, void* out_f16,
                                           int32_t n, int32_t dim, seva_stream_t stream) {
  SEVA_REQUIRE(t && freqs && out_f16 && n > 0 && dim > 1, "timestep_embedding: bad args");
  SevaProfScope prof(4, (double)n * dim * 2.0, (hipStream_t)stream);
  EW_LAUNCH(timestep_embedding_kernel, (int64_t)n * dim, t, freqs, (half_t*)out_f16, n, dim);
}

extern "C" int seva_silu_f16(const float* x, void* out_f16, int64_t count, seva_stream_t stream) {
  SEVA_REQUIRE(x && out_f16 && count > 0, "silu: bad args");
  SevaProfScope prof(4, (double)count * 6.0, (hipStream_t)stream);
  EW_LAUNCH(silu_f16_kernel, count, x, (half_t*)out_f16, count);
}

extern "C" int seva_add_f32(const float* a, const float* b, float* out, int64_t count,
                            seva_stream_t stream) {
  SEVA_REQUIRE(a && b && out && count > 0 && count % 4 == 0, "add: bad args");
  SEVA_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0, "add: unaligned");
  SevaProfScope prof(4, (double)count * 12.0, (hipStream_t)stream);
  EW_LAUNCH(add_f32_kernel, count / 4, a, b, out, count / 4);
}

extern "C" int seva_replace_blend_f32(const float* x, const float* replace, float* out, int32_t n,
                                      int32_t c, int32_t hw, seva_stream_t stream) {
  SEVA_REQUIRE(x && replace && out && n > 0 && c > 0 && hw > 0, "replace_blend: bad args");
  SevaProfScope prof(4, (double)n * c * hw * 16.0, (hipStream_t)stream);
  EW_LAUNCH(replace_blend_kernel, (int64_t)n * c * hw, x, replace, out, n, c, hw);
}

extern "C" int seva_denoiser_combine_f32(const float* net, const float* x, const float* c_out,
                                         const float* c_skip, float* out, int32_t n, int64_t chw,
                                         seva_stream_t stream) {
  SEVA_REQUIRE(net && x && c_out && c_skip && out && n > 0 && chw > 0, "denoiser_combine: bad args");
  SevaProfScope prof(4, (double)n * chw * 12.0, (hipStream_t)stream);
  EW_LAUNCH(denoiser_combine_kernel, (int64_t)n * chw, net, x, c_out, c_skip, out, n, chw);
}

extern "C" int seva_add_noise_f32(const float* x, const float* eps, const float* noise_scale,
                                  float* out, int32_t n, int64_t chw, seva_stream_t stream) {
  SEVA_REQUIRE(x && eps && noise_scale && out && n > 0 && chw > 0, "add_noise: bad args");
  SevaProfScope prof(4, (double)n * chw * 12.0, (hipStream_t)stream);
  EW_LAUNCH(add_noise_kernel, (int64_t)n * chw, x, eps, noise_scale, out, n, chw);
}

extern "C" int seva_cfg_euler_f32(const float* x, const float* den2, const float* scale,
                                  const float* sigma_hat, const float* dt, float* out, int32_t n,
                                  int64_t chw, seva_stream_t stream) {
  SEVA_REQUIRE(x && den2 && scale && sigma_hat && dt && out && n > 0 && chw > 0,
               "cfg_euler: bad args");
  SevaProfScope prof(4, (double)n * chw * 16.0, (hipStream_t)stream);
  EW_LAUNCH(cfg_euler_kernel, (int64_t)n * chw, x, den2, scale, sigma_hat, dt, out, n, chw);
}

extern "C" int seva_cfg_combine_f32(const float* den2, const float* scale, float* out, int32_t n,
                                    int64_t chw, seva_stream_t stream) {
  SEVA_REQUIRE(den2 && scale && out && n > 0 && chw > 0, "cfg_combine: bad args");
  SevaProfScope prof(4, (double)n * chw * 12.0, (hipStream_t)stream);
  EW_LAUNCH(cfg_combine_kernel, (int64_t)n * chw, den2, scale, out, n, chw);
}

extern "C" int seva_euler_step_f32(const float* x, const float* den, const float* sigma_hat,
                                   const float* dt, float* out, int32_t n, int64_t chw,
                                   seva_stream_t stream) {
  SEVA_REQUIRE(x && den && sigma_hat && dt && out && n > 0 && chw > 0, "euler_step: bad args");
  SevaProfScope prof(4, (double)n * chw * 12.0, (hipStream_t)stream);
  EW_LAUNCH(euler_step_kernel, (int64_t)n * chw, x, den, sigma_hat, dt, out, n, chw);
}

extern "C" int seva_scale_rows_f32(const float* x, const float* s, float* out, int32_t n,
                                   int64_t chw, seva_stream_t stream) {
  SEVA_REQUIRE(x && s && out && n > 0 && chw > 0, "scale_rows: bad args");
  SevaProfScope prof(4, (double)n * chw * 8.0, (hipStream_t)stream);
  EW_LAUNCH(scale_rows_kernel, (int64_t)n * chw, x, s, out, n, chw);
}

extern "C" int seva_to_d_f32(const float* x, const float* den, const float* sigma, float* out,
                             int32_t n, int64_t chw, seva_stream_t stream) {
  SEVA_REQUIRE(x && den && sigma && out && n > 0 && chw > 0, "to_d: bad args");
  SevaProfScope prof(4, (double)n * chw * 12.0, (hipStream_t)stream);
  EW_LAUNCH(to_d_kernel, (int64_t)n * chw, x, den, sigma, out, n, chw);
}

extern "C" int seva_plucker_f32(const float* kinv, const float* pose_inv, float* out, int32_t views,
                                int32_t h, int32_t w, seva_stream_t stream) {
  SEVA_REQUIRE(kinv && pose_inv && out && views > 0 && h > 0 && w > 0, "plucker: bad args");
  SEVA_REQUIRE(views <= 65535, "plucker: too many views");
  hipStream_t s_ = (hipStream_t)stream;
  SevaProfScope prof(4, (double)views * h * w * 24.0, s_);
  hipLaunchKernelGGL(plucker_kernel, dim3((h * w + 255) / 256, views), dim3(256), 0, s_, kinv, pose_inv, out, h, w);
  return seva_check_launch("plucker_kernel");
}

extern "C" int seva_cond_concat_f32(const float* plucker, const uint8_t* mask, float* c_concat, float* uc_concat,
                                    int32_t views, int32_t h, int32_t w, seva_stream_t stream) {
  SEVA_REQUIRE(plucker && mask && c_concat && uc_concat && views > 0 && h > 0 && w > 0, "cond_concat: bad args");
  SEVA_REQUIRE(views <= 65535, "cond_concat: too many views");
  hipStream_t s_ = (hipStream_t)stream;
  SevaProfScope prof(4, (double)views * h * w * 80.0, s_);
  hipLaunchKernelGGL(cond_concat_kernel, dim3((7 * h * w + 255) / 256, views), dim3(256), 0, s_, plucker, mask,
                     c_concat, uc_concat, h * w);
  return seva_check_launch("cond_concat_kernel");
}
