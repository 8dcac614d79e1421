// HBM-bound kernels of the hot path: layout packing, max-pool, GroupNorm(+ReLU), ROIAlign, shot mean, correlation,
// sigmoid focal loss.  All NHWC, 16-byte vector accesses per lane, grid-stride loops capped at 2048 blocks
// (cdna_hip_programming.md Guideline 11/13).
#include "osd_common.h"

namespace {

constexpr int kMaxBlocks = 2048;
inline int grid_for(long long work, int threads) {
  long long b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

// ------------------------------------------------------------------------------------------------ weight / image packs
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                        int cout, int cin, int R, int S, int w_rows, int cin_pad, int src_orsi) {
  const long long total = (long long)w_rows * R * S * cin_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_pad);
    long long t = i / cin_pad;
    const int s = (int)(t % S); t /= S;
    const int r = (int)(t % R);
    const int co = (int)(t / R);
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = src_orsi ? w[(((size_t)co * R + r) * S + s) * cin + ci] : w[(((size_t)co * cin + ci) * R + r) * S + s];
      if (scale) v *= scale[co];
    }
    dst[i] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                        int cout, int w_rows) {
  const int total = w_rows * 7 * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % 32, r = (i / 32) % 7, co = i / (32 * 7);
    const int s = j >> 2, c = j & 3;
    float v = 0.f;
    if (co < cout && s < 7 && c < 3) {
      v = w[(((size_t)co * 3 + c) * 7 + r) * 7 + s];
      if (scale) v *= scale[co];
    }
    dst[i] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void pack_image_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int h, int w, int hp, int wp,
                                  int pad_t, int pad_l) {
  const long long total = (long long)n * hp * wp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % wp);
    const int y = (int)((i / wp) % hp);
    const int b = (int)(i / ((long long)wp * hp));
    const int sy = y - pad_t, sx = x - pad_l;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = src[(((size_t)b * 3 + c) * h + sy) * w + sx];
    }
    T* o = dst + i * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = from_f32<T>(v[c]);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int n, int h, int w, int c,
                                    int stride, int c0) {
  const long long total = (long long)n * c * h * w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    long long t = i / w;
    const int y = (int)(t % h); t /= h;
    const int ch = (int)(t % c);
    const int b = (int)(t / c);
    dst[i] = to_f32(src[(((size_t)b * h + y) * w + x) * stride + c0 + ch]);
  }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int c, int h, int w) {
  const long long total = (long long)n * c * h * w;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long long t = i / c;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    dst[i] = from_f32<T>(src[(((size_t)b * c + ch) * h + y) * w + x]);
  }
}

// ------------------------------------------------------------------------------------------------ 16-byte chunk helpers
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Chunk<__bf16> {
  static constexpr int N = 8;
  float v[8];
  __device__ __forceinline__ void load(const __bf16* p) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ __forceinline__ void store(__bf16* p) const {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// ------------------------------------------------------------------------------------------------ max pool 3x3 s2 p1
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int ho, int wo) {
  constexpr int E = Chunk<T>::N;
  const int cch = c / E;
  const long long total = (long long)n * ho * wo * cch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cch);
    long long t = i / cch;
    const int ox = (int)(t % wo); t /= wo;
    const int oy = (int)(t % ho);
    const int b = (int)(t / ho);
    Chunk<T> m;
#pragma unroll
    for (int e = 0; e < E; ++e) m.v[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)w) continue;
        Chunk<T> v;
        v.load(x + (((size_t)b * h + iy) * w + ix) * c + cc * E);
#pragma unroll
        for (int e = 0; e < E; ++e) m.v[e] = fmaxf(m.v[e], v.v[e]);
      }
    }
    m.store(y + i * E);
  }
}

// ------------------------------------------------------------------------------------------------ GroupNorm
constexpr int kGnSplits = 64;  // partial-sum slabs per image: deterministic two-stage reduction, no atomics

// x [n][hw][c]; ws [n][kGnSplits][groups][2].  Block = 256 threads = (256 / cch) pixel lanes x cch chunk columns.
template <typename T>
__global__ void groupnorm_stats_kernel(const T* __restrict__ x, float* __restrict__ ws, int hw, int c, int groups) {
  constexpr int E = Chunk<T>::N;
  const int cch = c / E;                 // chunks per pixel
  const int lanes = blockDim.x / cch;    // pixel lanes
  const int b = blockIdx.y, split = blockIdx.x;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int per = (hw + kGnSplits - 1) / kGnSplits;
  const int p0 = split * per, p1 = min(hw, p0 + per);
  float s = 0.f, ss = 0.f;
  if (pl < lanes) {
    for (int p = p0 + pl; p < p1; p += lanes) {
      Chunk<T> v;
      v.load(x + ((size_t)b * hw + p) * c + cc * E);
#pragma unroll
      for (int e = 0; e < E; ++e) { s += v.v[e]; ss += v.v[e] * v.v[e]; }
    }
  }
  __shared__ float red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  // group g owns chunk columns [g*cpg_chunks, (g+1)*cpg_chunks)
  const int cpg_chunks = (c / groups) / E > 0 ? (c / groups) / E : 1;
  if (threadIdx.x < groups) {
    const int g = threadIdx.x;
    float ts = 0.f, tss = 0.f;
    for (int l = 0; l < lanes; ++l)
      for (int k = 0; k < cpg_chunks; ++k) {
        const int idx = l * cch + g * cpg_chunks + k;
        ts += red[0][idx];
        tss += red[1][idx];
      }
    float* o = ws + (((size_t)b * kGnSplits + split) * groups + g) * 2;
    o[0] = ts;
    o[1] = tss;
  }
}

__global__ void groupnorm_finalize_kernel(const float* __restrict__ ws, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ a, float* __restrict__ bb,
                                          int n, int hw, int c, int groups, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  const int b = i / c, ch = i % c;
  const int g = ch / (c / groups);
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < kGnSplits; ++k) {
    const float* o = ws + (((size_t)b * kGnSplits + k) * groups + g) * 2;
    s += o[0];
    ss += o[1];
  }
  const double cnt = (double)hw * (c / groups);
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float av = gamma[ch] * rstd;
  a[i] = av;
  bb[i] = beta[ch] - (float)mean * av;
}

template <typename T>
__global__ void groupnorm_relu_apply_kernel(const T* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                            T* __restrict__ y, int n, int hw, int c) {
  constexpr int E = Chunk<T>::N;
  const int cch = c / E;
  const long long total = (long long)n * hw * cch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cch);
    const int img = (int)(i / ((long long)cch * hw));
    Chunk<T> v;
    v.load(x + i * E);
    const float* ap = a + (size_t)img * c + cc * E;
    const float* bp = b + (size_t)img * c + cc * E;
#pragma unroll
    for (int e = 0; e < E; ++e) v.v[e] = fmaxf(fmaf(v.v[e], ap[e], bp[e]), 0.f);
    v.store(y + i * E);
  }
}

// ------------------------------------------------------------------------------------------------ ROIAlign forward (K5)
// Semantics of csrc/cuda/ROIAlign_cuda.cu:11-122 (bilinear_interpolate + RoIAlignForward) on an NHWC input — restructured for wave64
// (round 5): ONE WAVEFRONT per output cell (roi, ph, pw).  Everything that depends on the cell only — the sample
// grid, the four tap pixels and weights of every sample, the border rules — is wave-uniform: it is computed once per wavefront on
// the scalar unit (the reference's one-thread-per-output mapping recomputes it for every channel), and the 64 lanes stream the
// channel run of each tap: lane l owns channels 4 l .. 4 l + 3 of every 256-channel slab (8 / 16 bytes per lane and tap: one
// contiguous 512 / 1,024-byte run per wave-instruction).  The per-channel arithmetic is the reference's expression in its order
// (out += w1 v1 + w2 v2 + w3 v3 + w4 v4 per sample, / count at the end).
// the samples of ONE output cell for the lane's four channels of one 256-channel slab: roialign_fwd_kernel's loop body, for the all-levels
// query pooling below.  (roialign_fwd_kernel keeps its own inline copy: moved into this function hipcc contracted the sample-position
// arithmetic differently, and a position one ulp off flips a tap at an integer boundary — four box-head outputs of the `shots5` fixture
// moved by 0.1.  The two forms are pinned separately: the per-level kernel by its reference vectors and the box-head goldens, the
// all-levels kernel by the forward goldens' pooled vectors, and against each other to 1e-5.)
template <typename T>
__device__ __forceinline__ void roialign_cell_slab(const T* __restrict__ img, int h, int w, int c, int ch, bool vec, float rsw, float rsh,
                                                   float bin_w, float bin_h, int gh, int gw, int px, int py, float (&acc)[4]) {
  for (int iy = 0; iy < gh; ++iy) {
    const float yy = rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh;
    for (int ix = 0; ix < gw; ++ix) {
      const float xx = rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw;
      float yv = yy, xv = xx;
      if (yv < -1.0f || yv > (float)h || xv < -1.0f || xv > (float)w) continue;      // wave-uniform
      if (yv <= 0.f) yv = 0.f;
      if (xv <= 0.f) xv = 0.f;
      int yl = (int)yv, xl = (int)xv, yh, xh;
      if (yl >= h - 1) { yh = yl = h - 1; yv = (float)yl; } else { yh = yl + 1; }
      if (xl >= w - 1) { xh = xl = w - 1; xv = (float)xl; } else { xh = xl + 1; }
      const float ly = yv - yl, lx = xv - xl, hy = 1.f - ly, hx = 1.f - lx;
      const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
      const T* t1 = img + ((size_t)yl * w + xl) * c, *t2 = img + ((size_t)yl * w + xh) * c;
      const T* t3 = img + ((size_t)yh * w + xl) * c, *t4 = img + ((size_t)yh * w + xh) * c;
      if (vec) {
        if (ch < c) {
          float v1[4], v2[4], v3[4], v4[4];
          if constexpr (sizeof(T) == 2) {
            const bf16x4 a1 = *reinterpret_cast<const bf16x4*>(t1 + ch), a2 = *reinterpret_cast<const bf16x4*>(t2 + ch);
            const bf16x4 a3 = *reinterpret_cast<const bf16x4*>(t3 + ch), a4 = *reinterpret_cast<const bf16x4*>(t4 + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v1[e] = (float)a1[e]; v2[e] = (float)a2[e]; v3[e] = (float)a3[e]; v4[e] = (float)a4[e]; }
          } else {
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(t1 + ch), a2 = *reinterpret_cast<const f32x4*>(t2 + ch);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(t3 + ch), a4 = *reinterpret_cast<const f32x4*>(t4 + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v1[e] = a1[e]; v2[e] = a2[e]; v3[e] = a3[e]; v4[e] = a4[e]; }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
        }
      } else {                                          // channel counts that are not multiples of 4: one value at a time
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ch + e < c) acc[e] += w1 * to_f32(t1[ch + e]) + w2 * to_f32(t2[ch + e]) + w3 * to_f32(t3[ch + e]) + w4 * to_f32(t4[ch + e]);
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(64) roialign_fwd_kernel(const T* __restrict__ x, const float* __restrict__ rois, float* __restrict__ y, int h,
                                                          int w, int c, int num_rois, float scale, int ph, int pw, int sampling) {
  const int cell = blockIdx.x;                            // (roi, py, px)
  const int px = cell % pw, py = (cell / pw) % ph, r = cell / (pw * ph);
  const int lane = threadIdx.x;
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
  const float roi_w = fmaxf(rew - rsw, 1.f), roi_h = fmaxf(reh - rsh, 1.f);
  const float bin_h = roi_h / (float)ph, bin_w = roi_w / (float)pw;
  const int gh = sampling > 0 ? sampling : (int)ceilf(roi_h / ph);
  const int gw = sampling > 0 ? sampling : (int)ceilf(roi_w / pw);
  const float count = (float)(gh * gw);
  const T* img = x + (size_t)b * h * w * c;
  float* out_row = y + (size_t)cell * c;
  const bool vec = (c & 3) == 0;
  for (int c0 = 0; c0 < c; c0 += 256) {                   // 256-channel slabs (the FPN maps have exactly one)
    const int ch = c0 + lane * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw;
        float yv = yy, xv = xx;
        if (yv < -1.0f || yv > (float)h || xv < -1.0f || xv > (float)w) continue;      // wave-uniform
        if (yv <= 0.f) yv = 0.f;
        if (xv <= 0.f) xv = 0.f;
        int yl = (int)yv, xl = (int)xv, yh, xh;
        if (yl >= h - 1) { yh = yl = h - 1; yv = (float)yl; } else { yh = yl + 1; }
        if (xl >= w - 1) { xh = xl = w - 1; xv = (float)xl; } else { xh = xl + 1; }
        const float ly = yv - yl, lx = xv - xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const T* t1 = img + ((size_t)yl * w + xl) * c, *t2 = img + ((size_t)yl * w + xh) * c;
        const T* t3 = img + ((size_t)yh * w + xl) * c, *t4 = img + ((size_t)yh * w + xh) * c;
        if (vec) {
          if (ch < c) {
            float v1[4], v2[4], v3[4], v4[4];
            if constexpr (sizeof(T) == 2) {
              const bf16x4 a1 = *reinterpret_cast<const bf16x4*>(t1 + ch), a2 = *reinterpret_cast<const bf16x4*>(t2 + ch);
              const bf16x4 a3 = *reinterpret_cast<const bf16x4*>(t3 + ch), a4 = *reinterpret_cast<const bf16x4*>(t4 + ch);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v1[e] = (float)a1[e]; v2[e] = (float)a2[e]; v3[e] = (float)a3[e]; v4[e] = (float)a4[e]; }
            } else {
              const f32x4 a1 = *reinterpret_cast<const f32x4*>(t1 + ch), a2 = *reinterpret_cast<const f32x4*>(t2 + ch);
              const f32x4 a3 = *reinterpret_cast<const f32x4*>(t3 + ch), a4 = *reinterpret_cast<const f32x4*>(t4 + ch);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v1[e] = a1[e]; v2[e] = a2[e]; v3[e] = a3[e]; v4[e] = a4[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
          }
        } else {                                          // channel counts that are not multiples of 4: one value at a time
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ch + e < c) acc[e] += w1 * to_f32(t1[ch + e]) + w2 * to_f32(t2[ch + e]) + w3 * to_f32(t3[ch + e]) + w4 * to_f32(t4[ch + e]);
        }
      }
    }
    if (vec) {
      if (ch < c) *reinterpret_cast<f32x4*>(out_row + ch) = f32x4{acc[0] / count, acc[1] / count, acc[2] / count, acc[3] / count};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ch + e < c) out_row[ch + e] = acc[e] / count;
    }
  }
}

// ---- query pooling of ALL FPN levels in one launch (round 6): SuppAlignLayer's 1 x 1 ROIAlign of every query's whole-image box
// (generalized_rcnn.py:20-52) + batch_pooling's mean over the shots of a target image (:100-104).  One wavefront per (target image,
// level); per shot the cell value is roialign_fwd_kernel's (acc / count), the mean shot_mean_kernel's (sum in shot order, / shots):
// the arithmetic of the 2 x levels launches it replaces (equal to 1e-5, tests/test_gpu_kernels.py).
constexpr int kQPoolLevels = 8;
struct QPoolLevels {
  const void* x[kQPoolLevels];
  float* y[kQPoolLevels];
  int h[kQPoolLevels], w[kQPoolLevels];
  float scale[kQPoolLevels];
};

template <typename T>
__global__ void __launch_bounds__(64) query_pool_levels_kernel(QPoolLevels L, const float* __restrict__ rois, int c, int shots, int sampling) {
  const int lvl = blockIdx.y, bimg = blockIdx.x, lane = threadIdx.x;
  const int h = L.h[lvl], w = L.w[lvl];
  const float scale = L.scale[lvl];
  const T* x = reinterpret_cast<const T*>(L.x[lvl]);
  float* out_row = L.y[lvl] + (size_t)bimg * c;
  const bool vec = (c & 3) == 0;
  for (int c0 = 0; c0 < c; c0 += 256) {
    const int ch = c0 + lane * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < shots; ++k) {
      const float* roi = rois + (size_t)(bimg * shots + k) * 5;
      const int b = (int)roi[0];
      const float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
      const float roi_w = fmaxf(rew - rsw, 1.f), roi_h = fmaxf(reh - rsh, 1.f);
      const float bin_h = roi_h / 1.f, bin_w = roi_w / 1.f;
      const int gh = sampling > 0 ? sampling : (int)ceilf(roi_h / 1);
      const int gw = sampling > 0 ? sampling : (int)ceilf(roi_w / 1);
      const float count = (float)(gh * gw);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      roialign_cell_slab<T>(x + (size_t)b * h * w * c, h, w, c, ch, vec, rsw, rsh, bin_w, bin_h, gh, gw, 0, 0, acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += acc[e] / count;
    }
    if (vec) {
      if (ch < c) *reinterpret_cast<f32x4*>(out_row + ch) = f32x4{s[0] / (float)shots, s[1] / (float)shots, s[2] / (float)shots, s[3] / (float)shots};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ch + e < c) out_row[ch + e] = s[e] / (float)shots;
    }
  }
}

__global__ void shot_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int b, int shots, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * c) return;
  const int img = i / c, ch = i % c;
  float s = 0.f;
  for (int k = 0; k < shots; ++k) s += x[((size_t)img * shots + k) * c + ch];
  y[i] = s / (float)shots;
}

// ------------------------------------------------------------------------------------------------ correlation (K4)
// y[n,p,c] = x[n,p,c] * q[n,c].  grid = (slabs, n): the query vector of image n is staged ONCE per workgroup in LDS
// and reused across that workgroup's target pixels; the target streams through in 16-byte per-lane NHWC accesses
// (a wavefront touches 1 KiB of contiguous pixels x channels per instruction).
template <typename T>
__global__ void __launch_bounds__(256) correlate_kernel(const T* __restrict__ x, const float* __restrict__ q, T* __restrict__ y,
                                                        int hw, int c) {
  constexpr int E = Chunk<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);
  const int img = blockIdx.y;
  for (int i = threadIdx.x; i < c; i += blockDim.x) qs[i] = q[(size_t)img * c + i];
  __syncthreads();
  const int cch = c / E;
  const long long total = (long long)hw * cch;   // chunks in this image
  const T* xi = x + (size_t)img * hw * c;
  T* yi = y + (size_t)img * hw * c;
  // cch is a power of two for every FPN tensor (256 channels); the chunk column of a thread is loop-invariant when
  // the stride (gridDim.x*256) is a multiple of cch, so the LDS reads hoist out of the loop.
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const bool invariant = (stride % cch) == 0;
  float qv[E];
  if (invariant) {
    const int cc = (int)(i % cch);
#pragma unroll
    for (int e = 0; e < E; ++e) qv[e] = qs[cc * E + e];
  }
  for (; i < total; i += stride) {
    if (!invariant) {
      const int cc = (int)(i % cch);
#pragma unroll
      for (int e = 0; e < E; ++e) qv[e] = qs[cc * E + e];
    }
    Chunk<T> v;
    v.load(xi + i * E);
#pragma unroll
    for (int e = 0; e < E; ++e) v.v[e] *= qv[e];
    v.store(yi + i * E);
  }
}

// All FPN levels in ONE launch (forward y_l = x_l * q_l and backward d_feat_l = g_l * q_l of generalized_rcnn.py:307-311):
// a workgroup finds its (level, image, slab) from a table in the kernel arguments (static indices only: scalar selects,
// no scratch copy of the table), stages that image's query vector of that level in LDS and streams its slab with two
// independent 16-byte loads in flight per thread.
constexpr int kCorrLevels = 6;
struct CorrLevels {
  const void* x[kCorrLevels];
  const float* q[kCorrLevels];
  void* y[kCorrLevels];
  int hw[kCorrLevels];
  int bx[kCorrLevels];        // workgroups per image at this level
  int begin[kCorrLevels];     // first workgroup of this level
  int n_levels;
};

template <typename T>
__global__ void __launch_bounds__(256) correlate_levels_kernel(CorrLevels L, int c) {
  constexpr int E = Chunk<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);
  const int b = blockIdx.x;
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < kCorrLevels; ++i)
    if (i < L.n_levels && b >= L.begin[i]) lvl = i;
  const void* xv = L.x[0]; const float* q = L.q[0]; void* yv = L.y[0];
  int hw = L.hw[0], bx = L.bx[0], beg = L.begin[0];
#pragma unroll
  for (int i = 1; i < kCorrLevels; ++i)
    if (lvl == i) { xv = L.x[i]; q = L.q[i]; yv = L.y[i]; hw = L.hw[i]; bx = L.bx[i]; beg = L.begin[i]; }
  const int local = b - beg;
  const int img = local / bx, slab = local - img * bx;
  for (int i = threadIdx.x; i < c; i += blockDim.x) qs[i] = q[(size_t)img * c + i];
  __syncthreads();
  const int cch = c / E;
  const long long total = (long long)hw * cch;
  const T* xi = reinterpret_cast<const T*>(xv) + (size_t)img * hw * c;
  T* yi = reinterpret_cast<T*>(yv) + (size_t)img * hw * c;
  const long long stride = (long long)bx * blockDim.x;
  long long i = slab * (long long)blockDim.x + threadIdx.x;
  // the host makes bx * 256 a multiple of the chunks per pixel: a thread's channel chunk is loop invariant
  const int cc = (int)(i % cch);
  float qv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) qv[e] = qs[cc * E + e];
  for (; i + stride < total; i += 2 * stride) {
    Chunk<T> v0, v1;
    v0.load(xi + i * E);
    v1.load(xi + (i + stride) * E);
#pragma unroll
    for (int e = 0; e < E; ++e) { v0.v[e] *= qv[e]; v1.v[e] *= qv[e]; }
    v0.store(yi + i * E);
    v1.store(yi + (i + stride) * E);
  }
  if (i < total) {
    Chunk<T> v;
    v.load(xi + i * E);
#pragma unroll
    for (int e = 0; e < E; ++e) v.v[e] *= qv[e];
    v.store(yi + i * E);
  }
}

// ------------------------------------------------------------------------------------------------ sigmoid focal loss
// csrc/cuda/SigmoidFocalLoss_cuda.cu:21-58 / :62-101
__global__ void sigmoid_focal_fwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                         float* __restrict__ losses, int total, int classes, float gamma, float alpha) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / classes, d = i % classes, t = targets[n];
    const float c1 = (t == (d + 1)) ? 1.f : 0.f;
    const float c2 = (t >= 0 && t != (d + 1)) ? 1.f : 0.f;
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    const float term1 = powf(1.f - p, gamma) * logf(fmaxf(p, 1.17549435e-38f));
    const float ge = x >= 0.f ? 1.f : 0.f;
    const float term2 = powf(p, gamma) * (-1.f * x * ge - logf(1.f + expf(x - 2.f * x * ge)));
    losses[i] = -c1 * term1 * alpha - c2 * term2 * (1.f - alpha);
  }
}

__global__ void sigmoid_focal_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                         const float* __restrict__ d_losses, float* __restrict__ d_logits, int total,
                                         int classes, float gamma, float alpha) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = i / classes, d = i % classes, t = targets[n];
    const float c1 = (t == (d + 1)) ? 1.f : 0.f;
    const float c2 = (t >= 0 && t != (d + 1)) ? 1.f : 0.f;
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    const float term1 = powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(fmaxf(p, 1.17549435e-38f))));
    const float ge = x >= 0.f ? 1.f : 0.f;
    const float term2 =
        powf(p, gamma) * ((-1.f * x * ge - logf(1.f + expf(x - 2.f * x * ge))) * (1.f - p) * gamma - p);
    d_logits[i] = (-c1 * term1 * alpha - c2 * term2 * (1.f - alpha)) * d_losses[i];
  }
}

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)
#define OSD_DISPATCH_DTYPE(dtype, CALL_F32, CALL_BF16)                            \
  do {                                                                           \
    if ((dtype) == OSD_F32) { CALL_F32; }                                        \
    else if ((dtype) == OSD_BF16) { CALL_BF16; }                                 \
    else return osd_fail(OSD_ERR_INVALID_ARG, "bad dtype %d", (int)(dtype));     \
  } while (0)

extern "C" int osd_pack_conv_weight(const float* w, const float* scale, void* dst, int cout, int cin, int r, int s,
                                    int w_rows, int cin_pad, int dtype, void* stream) {
  return osd_pack_conv_weight_ex(w, scale, dst, cout, cin, r, s, w_rows, cin_pad, 0, dtype, stream);
}

extern "C" int osd_pack_conv_weight_ex(const float* w, const float* scale, void* dst, int cout, int cin, int r, int s,
                                       int w_rows, int cin_pad, int src_orsi, int dtype, void* stream) {
  if (!w || !dst || w_rows < cout || cin_pad < cin) return osd_fail(OSD_ERR_INVALID_ARG, "pack_conv_weight: bad args");
  const long long total = (long long)w_rows * r * s * cin_pad;
  const int g = grid_for(total, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), w, scale, (float*)dst, cout, cin, r, s, w_rows, cin_pad, src_orsi),
      hipLaunchKernelGGL(pack_conv_weight_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), w, scale, (__bf16*)dst, cout, cin, r, s, w_rows, cin_pad, src_orsi));
  return osd_check_launch("pack_conv_weight");
}

extern "C" int osd_pack_stem_weight(const float* w, const float* scale, void* dst, int cout, int w_rows, int dtype,
                                    void* stream) {
  if (!w || !dst || w_rows < cout) return osd_fail(OSD_ERR_INVALID_ARG, "pack_stem_weight: bad args");
  const int g = grid_for((long long)w_rows * 7 * 32, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(pack_stem_weight_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), w, scale, (float*)dst, cout, w_rows),
      hipLaunchKernelGGL(pack_stem_weight_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), w, scale, (__bf16*)dst, cout, w_rows));
  return osd_check_launch("pack_stem_weight");
}

extern "C" int osd_pack_image(const float* src, void* dst, int n, int h, int w, int hp, int wp, int pad_t, int pad_l,
                              int dtype, void* stream) {
  if (!src || !dst || hp < h + pad_t || wp < w + pad_l) return osd_fail(OSD_ERR_INVALID_ARG, "pack_image: bad args");
  const int g = grid_for((long long)n * hp * wp, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(pack_image_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), src, (float*)dst, n, h, w, hp, wp, pad_t, pad_l),
      hipLaunchKernelGGL(pack_image_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), src, (__bf16*)dst, n, h, w, hp, wp, pad_t, pad_l));
  return osd_check_launch("pack_image");
}

extern "C" int osd_nhwc_to_nchw_f32(const void* src, float* dst, int n, int h, int w, int c, int stride, int c0,
                                    int dtype, void* stream) {
  if (!src || !dst || c0 + c > stride) return osd_fail(OSD_ERR_INVALID_ARG, "nhwc_to_nchw: bad args");
  const int g = grid_for((long long)n * h * w * c, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)src, dst, n, h, w, c, stride, c0),
      hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)src, dst, n, h, w, c, stride, c0));
  return osd_check_launch("nhwc_to_nchw");
}

extern "C" int osd_nchw_f32_to_nhwc(const float* src, void* dst, int n, int c, int h, int w, int dtype, void* stream) {
  if (!src || !dst) return osd_fail(OSD_ERR_INVALID_ARG, "nchw_to_nhwc: bad args");
  const int g = grid_for((long long)n * h * w * c, 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), src, (float*)dst, n, c, h, w),
      hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), src, (__bf16*)dst, n, c, h, w));
  return osd_check_launch("nchw_to_nhwc");
}

extern "C" int osd_maxpool3x3s2_fwd(const void* x, void* y, int n, int h, int w, int c, int ho, int wo, int dtype,
                                    void* stream) {
  if (!x || !y || c % 8 != 0) return osd_fail(OSD_ERR_INVALID_ARG, "maxpool: bad args (c must be a multiple of 8)");
  const int e = dtype == OSD_BF16 ? 8 : 4;
  const int g = grid_for((long long)n * ho * wo * (c / e), 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)x, (float*)y, n, h, w, c, ho, wo),
      hipLaunchKernelGGL(maxpool3x3s2_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)x, (__bf16*)y, n, h, w, c, ho, wo));
  return osd_check_launch("maxpool");
}

extern "C" int osd_groupnorm_stats(const void* x, float* ws, int n, int hw, int c, int groups, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!x || !ws || c % e != 0 || c / e > 256 || 256 % (c / e) != 0 || groups > 256 || c % groups != 0 ||
      (c / groups) % e != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_stats: unsupported shape c=%d groups=%d", c, groups);
  dim3 grid(kGnSplits, n);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(groupnorm_stats_kernel<float>, grid, dim3(256), 0, OSD_STREAM(stream), (const float*)x, ws, hw, c, groups),
      hipLaunchKernelGGL(groupnorm_stats_kernel<__bf16>, grid, dim3(256), 0, OSD_STREAM(stream), (const __bf16*)x, ws, hw, c, groups));
  return osd_check_launch("groupnorm_stats");
}

extern "C" int osd_groupnorm_finalize(const float* ws, const float* gamma, const float* beta, float* a, float* b, int n,
                                      int hw, int c, int groups, float eps, void* stream) {
  if (!ws || !gamma || !beta || !a || !b) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_finalize: null argument");
  hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(cdiv(n * c, 256)), dim3(256), 0, OSD_STREAM(stream), ws, gamma, beta, a,
                     b, n, hw, c, groups, eps);
  return osd_check_launch("groupnorm_finalize");
}

extern "C" int osd_groupnorm_relu_apply(const void* x, const float* a, const float* b, void* y, int n, int hw, int c,
                                        int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (!x || !a || !b || !y || c % e != 0) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_relu_apply: bad args");
  const int g = grid_for((long long)n * hw * (c / e), 256);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(groupnorm_relu_apply_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)x, a, b, (float*)y, n, hw, c),
      hipLaunchKernelGGL(groupnorm_relu_apply_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)x, a, b, (__bf16*)y, n, hw, c));
  return osd_check_launch("groupnorm_relu_apply");
}

extern "C" int osd_roialign_fwd(const void* x, const float* rois, float* y, int b, int h, int w, int c, int num_rois,
                                float spatial_scale, int ph, int pw, int sampling_ratio, int dtype, void* stream) {
  (void)b;
  if (num_rois == 0) return OSD_OK;
  if (!x || !rois || !y) return osd_fail(OSD_ERR_INVALID_ARG, "roialign: null argument");
  const long long cells = (long long)num_rois * ph * pw;      // one wavefront per output cell
  if (cells <= 0 || cells > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "roialign: bad roi / cell count");
  const int g = (int)cells;
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(roialign_fwd_kernel<float>, dim3(g), dim3(64), 0, OSD_STREAM(stream), (const float*)x, rois, y, h, w, c, num_rois, spatial_scale, ph, pw, sampling_ratio),
      hipLaunchKernelGGL(roialign_fwd_kernel<__bf16>, dim3(g), dim3(64), 0, OSD_STREAM(stream), (const __bf16*)x, rois, y, h, w, c, num_rois, spatial_scale, ph, pw, sampling_ratio));
  return osd_check_launch("roialign_fwd");
}

extern "C" int osd_shot_mean(const float* x, float* y, int b, int shots, int c, void* stream) {
  if (!x || !y || shots < 1) return osd_fail(OSD_ERR_INVALID_ARG, "shot_mean: bad args");
  hipLaunchKernelGGL(shot_mean_kernel, dim3(cdiv(b * c, 256)), dim3(256), 0, OSD_STREAM(stream), x, y, b, shots, c);
  return osd_check_launch("shot_mean");
}

extern "C" int osd_query_pool_levels(int n_levels, const void* const* xs, const int32_t* hs, const int32_t* ws, const float* scales,
                                     const float* rois, int batch, int shots, int c, int sampling_ratio, float* const* ys, int dtype,
                                     void* stream) {
  if (n_levels < 1 || n_levels > kQPoolLevels || !xs || !hs || !ws || !scales || !rois || !ys || shots < 1 || c < 1)
    return osd_fail(OSD_ERR_INVALID_ARG, "query_pool_levels: bad arguments");
  if (batch == 0) return OSD_OK;
  QPoolLevels L;
  for (int l = 0; l < kQPoolLevels; ++l) {
    const int j = l < n_levels ? l : 0;
    if (!xs[j] || !ys[j] || hs[j] < 1 || ws[j] < 1) return osd_fail(OSD_ERR_INVALID_ARG, "query_pool_levels: bad level %d", j);
    L.x[l] = xs[j]; L.y[l] = ys[j]; L.h[l] = hs[j]; L.w[l] = ws[j]; L.scale[l] = scales[j];
  }
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(query_pool_levels_kernel<float>, dim3(batch, n_levels), dim3(64), 0, OSD_STREAM(stream), L, rois, c, shots, sampling_ratio),
      hipLaunchKernelGGL(query_pool_levels_kernel<__bf16>, dim3(batch, n_levels), dim3(64), 0, OSD_STREAM(stream), L, rois, c, shots, sampling_ratio));
  return osd_check_launch("query_pool_levels");
}

extern "C" int osd_correlate_fwd(const void* x, const float* q, void* y, int n, int hw, int c, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (c % e != 0 || c > 8192) return osd_fail(OSD_ERR_INVALID_ARG, "correlate: bad channel count %d", c);
  if (n == 0 || hw == 0) return OSD_OK;
  if (!x || !q || !y) return osd_fail(OSD_ERR_INVALID_ARG, "correlate: null argument");
  const long long chunks = (long long)hw * (c / e);
  // >= 8 chunks (128 B) per thread where the image is large enough; blocks x images >> 256 CUs at the P3 size
  int bx = (int)((chunks + 256LL * 8 - 1) / (256LL * 8));
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  dim3 grid(bx, n);
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(correlate_kernel<float>, grid, dim3(256), c * sizeof(float), OSD_STREAM(stream), (const float*)x, q, (float*)y, hw, c),
      hipLaunchKernelGGL(correlate_kernel<__bf16>, grid, dim3(256), c * sizeof(float), OSD_STREAM(stream), (const __bf16*)x, q, (__bf16*)y, hw, c));
  return osd_check_launch("correlate");
}

extern "C" int osd_correlate_levels(int n_levels, const void* const* xs, const float* const* qs, void* const* ys,
                                    const int32_t* hws, int n, int c, int dtype, void* stream) {
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (n_levels < 1 || n_levels > kCorrLevels || !xs || !qs || !ys || !hws)
    return osd_fail(OSD_ERR_INVALID_ARG, "correlate_levels: 1..%d levels", kCorrLevels);
  if (c % e != 0 || c > 8192 || 256 % (c / e) != 0) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_levels: bad channel count %d", c);
  if (dtype != OSD_F32 && dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_levels: bad dtype");
  if (n == 0) return OSD_OK;
  CorrLevels L;
  L.n_levels = 0;
  int blocks = 0;
  for (int i = 0; i < n_levels; ++i) {
    if (hws[i] <= 0) continue;
    if (!xs[i] || !qs[i] || !ys[i]) return osd_fail(OSD_ERR_INVALID_ARG, "correlate_levels: null tensor at level %d", i);
    const long long chunks = (long long)hws[i] * (c / e);
    int bx = (int)((chunks + 256LL * 8 - 1) / (256LL * 8));     // >= 8 chunks (128 B) per thread where the level is large enough
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    const int k = L.n_levels++;
    L.x[k] = xs[i]; L.q[k] = qs[i]; L.y[k] = ys[i]; L.hw[k] = hws[i]; L.bx[k] = bx; L.begin[k] = blocks;
    blocks += bx * n;
  }
  if (L.n_levels == 0) return OSD_OK;
  for (int k = L.n_levels; k < kCorrLevels; ++k) { L.x[k] = L.x[0]; L.q[k] = L.q[0]; L.y[k] = L.y[0]; L.hw[k] = 0; L.bx[k] = 1; L.begin[k] = 0x7fffffff; }
  OSD_DISPATCH_DTYPE(dtype,
      hipLaunchKernelGGL(correlate_levels_kernel<float>, dim3(blocks), dim3(256), c * sizeof(float), OSD_STREAM(stream), L, c),
      hipLaunchKernelGGL(correlate_levels_kernel<__bf16>, dim3(blocks), dim3(256), c * sizeof(float), OSD_STREAM(stream), L, c));
  return osd_check_launch("correlate_levels");
}

extern "C" int osd_sigmoid_focal_fwd(const float* logits, const int32_t* targets, float* losses, int m, int classes,
                                     float gamma, float alpha, void* stream) {
  if (!logits || !targets || !losses) return osd_fail(OSD_ERR_INVALID_ARG, "focal_fwd: null argument");
  if (m == 0) return OSD_OK;
  hipLaunchKernelGGL(sigmoid_focal_fwd_kernel, dim3(grid_for((long long)m * classes, 256)), dim3(256), 0, OSD_STREAM(stream),
                     logits, targets, losses, m * classes, classes, gamma, alpha);
  return osd_check_launch("focal_fwd");
}

extern "C" int osd_sigmoid_focal_bwd(const float* logits, const int32_t* targets, const float* d_losses, float* d_logits,
                                     int m, int classes, float gamma, float alpha, void* stream) {
  if (!logits || !targets || !d_losses || !d_logits) return osd_fail(OSD_ERR_INVALID_ARG, "focal_bwd: null argument");
  if (m == 0) return OSD_OK;
  hipLaunchKernelGGL(sigmoid_focal_bwd_kernel, dim3(grid_for((long long)m * classes, 256)), dim3(256), 0, OSD_STREAM(stream),
                     logits, targets, d_losses, d_logits, m * classes, classes, gamma, alpha);
  return osd_check_launch("focal_bwd");
}
