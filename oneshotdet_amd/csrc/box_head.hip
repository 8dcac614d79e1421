// Second-stage few-shot ROI box head, the kernels that are not convolutions:
//   * level-routed 7x7 ROIAlign over the five FPN levels            (modeling/poolers.py:11-42,93-124)
//   * GroupNorm(32, C) + LeakyReLU(0.2) over [R][7][7][C] ROI maps  (roi_heads/box_head/box_head.py:43-66)
//   * arg-max over shots, softmax, BoxCoder.decode, clip            (box_head.py:239-252, box_head/inference.py:46-118,
//                                                                    modeling/box_coder.py:50-95)
// All HBM-bound.  The convolutions / fully connected layers of the head run on the implicit-GEMM kernels.
#include "osd_common.h"

namespace {

inline int grid_for(long long work, int threads) {
  long long g = (work + threads - 1) / threads;
  return (int)(g < 1 ? 1 : (g > 1048576 ? 1048576 : g));
}

template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int E = 4;
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ __forceinline__ void store(float* p) const {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct Chunk<__bf16> {
  static constexpr int E = 8;
  float v[8];
  __device__ __forceinline__ void load(const __bf16* p) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  __device__ __forceinline__ void store(__bf16* p) const {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// two adjacent channels (even index) as one 4- / 8-byte access
__device__ __forceinline__ void load2(const float* p, float& a, float& b) {
  const f32x2 t = *reinterpret_cast<const f32x2*>(p);
  a = t[0]; b = t[1];
}
__device__ __forceinline__ void load2(const __bf16* p, float& a, float& b) {
  const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
  a = __uint_as_float(u << 16);
  b = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ void store2(float* p, float a, float b) {
  f32x2 t = {a, b};
  *reinterpret_cast<f32x2*>(p) = t;
}
__device__ __forceinline__ void store2(__bf16* p, float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 t = {(__bf16)a, (__bf16)b};
  *reinterpret_cast<bf16x2*>(p) = t;
}

struct PoolLevels {
  const void* x[OSD_MAX_ROI_LEVELS];
  int h[OSD_MAX_ROI_LEVELS];
  int w[OSD_MAX_ROI_LEVELS];
  float scale[OSD_MAX_ROI_LEVELS];
  int n_levels;
};

// LevelMapper (poolers.py:33-42) with BoxList.area's "+1" (bounding_box.py:226-236)
__device__ __forceinline__ int map_level(float x1, float y1, float x2, float y2, int k_min, int k_max) {
  const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
  const float s = sqrtf(area);
  float lv = floorf(4.f + log2f(s / 224.f + 1e-6f));
  lv = fminf(fmaxf(lv, (float)k_min), (float)k_max);
  return (int)lv - k_min;
}

// One workgroup per ROI.  The 49 output cells x C channels are walked as (cell, 16-byte channel chunk) items, chunk
// fastest, so the four bilinear taps of a sample are contiguous channel runs of the NHWC level map.  Arithmetic and
// accumulation order of csrc/cuda/ROIAlign_cuda.cu:11-122 (w1*v1 + w2*v2 + w3*v3 + w4*v4 per sample, samples in
// (iy, ix) order, one division by the sample count).  ROIs past counts[image] give zero rows.
// (Tried: staging the ROI's pixel patch, one 64-channel slab per workgroup, in LDS before the taps - bit-identical, but
// 5 % slower end to end, 996 vs 1045 images/s: the vector L1 already serves the repeated taps, and the staged version
// pays a barrier, an LDS round trip and 4x the workgroups.)
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_levels_kernel(PoolLevels lv, const float* __restrict__ boxes,
                                                              const int32_t* __restrict__ counts, T* __restrict__ y, int c,
                                                              int max_rois, int pool, int sampling, int y_stride,
                                                              int32_t* __restrict__ level_out) {
  constexpr int E = Chunk<T>::E;
  const int roi = blockIdx.x;
  const int img = roi / max_rois, ri = roi % max_rois;
  const int chunks = c / E;
  const int items = pool * pool * chunks;
  T* yr = y + (size_t)roi * pool * pool * y_stride;
  const bool live = counts == nullptr || ri < counts[img];
  if (!live) {
    Chunk<T> z;
#pragma unroll
    for (int e = 0; e < E; ++e) z.v[e] = 0.f;
    for (int it = threadIdx.x; it < items; it += blockDim.x) z.store(yr + (size_t)(it / chunks) * y_stride + (it % chunks) * E);
    if (level_out && threadIdx.x == 0) level_out[roi] = -1;
    return;
  }
  const float* bx = boxes + (size_t)roi * 4;
  const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  const int k_min = 3, k_max = 3 + lv.n_levels - 1;     // -log2(scales[0]) .. -log2(scales[-1]) (poolers.py:73-75)
  const int l = map_level(x1, y1, x2, y2, k_min, k_max);
  if (level_out && threadIdx.x == 0) level_out[roi] = l;
  const int h = lv.h[l], w = lv.w[l];
  const float scale = lv.scale[l];
  const T* x = reinterpret_cast<const T*>(lv.x[l]) + (size_t)img * h * w * c;
  const float rsw = x1 * scale, rsh = y1 * scale, rew = x2 * scale, reh = y2 * scale;
  const float roi_w = fmaxf(rew - rsw, 1.f), roi_h = fmaxf(reh - rsh, 1.f);
  const float bin_h = roi_h / (float)pool, bin_w = roi_w / (float)pool;
  const int gh = sampling > 0 ? sampling : (int)ceilf(roi_h / pool);
  const int gw = sampling > 0 ? sampling : (int)ceilf(roi_w / pool);
  const float count = (float)(gh * gw);
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int cell = it / chunks, ch = (it % chunks) * E;
    const int py = cell / pool, px = cell % pool;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (gh <= 2 && gw <= 2) {
      // separable bilinear weights: the cell's samples are merged per axis first (equal pixel indices summed), so the
      // sampling-ratio-2 cell reads 3 x 3 (or fewer) taps instead of 16 — this kernel is bound by its tap loads
      int iyv[4], ixv[4];
      float wyv[4], wxv[4];
      int ny = 0, nx = 0;
      auto axis = [](float v0, int size, int* idx, float* wt, int& n) {
        float v = v0;
        if (v < -1.0f || v > (float)size) return;
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
        const float fr = v - lo;
        const int ids[2] = {lo, hi};
        const float ws[2] = {1.f - fr, fr};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bool found = false;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < n && idx[q] == ids[t]) { wt[q] += ws[t]; found = true; }
          if (!found) { idx[n] = ids[t]; wt[n] = ws[t]; ++n; }
        }
      };
      for (int iy = 0; iy < gh; ++iy) axis(rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh, h, iyv, wyv, ny);
      for (int ix = 0; ix < gw; ++ix) axis(rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw, w, ixv, wxv, nx);
      for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
          Chunk<T> v;
          v.load(x + ((size_t)iyv[a] * w + ixv[b]) * c + ch);
          const float wgt = wyv[a] * wxv[b];
#pragma unroll
          for (int e = 0; e < E; ++e) acc[e] += wgt * v.v[e];
        }
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < E; ++e) o.v[e] = acc[e] / count;
      o.store(yr + (size_t)cell * y_stride + ch);
      continue;
    }
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = rsh + py * bin_h + (iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = rsw + px * bin_w + (ix + .5f) * bin_w / (float)gw;
        float yv = yy, xv = xx;
        if (yv < -1.0f || yv > (float)h || xv < -1.0f || xv > (float)w) continue;
        if (yv <= 0.f) yv = 0.f;
        if (xv <= 0.f) xv = 0.f;
        int yl = (int)yv, xl = (int)xv, yh, xh;
        if (yl >= h - 1) { yh = yl = h - 1; yv = (float)yl; } else { yh = yl + 1; }
        if (xl >= w - 1) { xh = xl = w - 1; xv = (float)xl; } else { xh = xl + 1; }
        const float ly = yv - yl, lx = xv - xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        Chunk<T> v1, v2, v3, v4;
        v1.load(x + ((size_t)yl * w + xl) * c + ch);
        v2.load(x + ((size_t)yl * w + xh) * c + ch);
        v3.load(x + ((size_t)yh * w + xl) * c + ch);
        v4.load(x + ((size_t)yh * w + xh) * c + ch);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += w1 * v1.v[e] + w2 * v2.v[e] + w3 * v3.v[e] + w4 * v4.v[e];
      }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < E; ++e) o.v[e] = acc[e] / count;
    o.store(yr + (size_t)cell * y_stride + ch);
  }
}

// GroupNorm(groups, C) + LeakyReLU(slope) of one ROI map [hw][C] per workgroup (hw = 49): the whole sample sits in
// registers, so mean and the centred second moment are two exact passes over registers and HBM is read once.
// Thread t owns channels (2t, 2t+1) of every pixel; a group's channels are CPG/2 adjacent lanes.  `addend` (optional):
// [n_add][hw][C], row of this sample = sample / rois_per_add * add_stride + add_offset — the query half of the
// concatenated 1x1 conv (box_head.py:146-148), which is the same for every ROI of an image, added before the statistics.
template <typename T, int HW>
__global__ __launch_bounds__(256) void gn_act_rois_kernel(const T* __restrict__ x, const T* __restrict__ addend,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ y, int c, int groups, float eps, float slope,
                                                          int rois_per_add, int add_stride, int add_offset) {
  const int sample = blockIdx.x;
  const int t = threadIdx.x;                 // blockDim.x == c / 2
  const int cpg = c / groups;                // channels per group (even)
  const int lanes = cpg / 2;                 // lanes per group: power of two <= 64
  const T* xs = x + (size_t)sample * HW * c + 2 * t;
  float v0[HW], v1[HW];
#pragma unroll
  for (int p = 0; p < HW; ++p) load2(xs + (size_t)p * c, v0[p], v1[p]);
  if (addend) {
    const T* as = addend + ((size_t)(sample / rois_per_add) * add_stride + add_offset) * HW * c + 2 * t;
#pragma unroll
    for (int p = 0; p < HW; ++p) {
      float a0, a1;
      load2(as + (size_t)p * c, a0, a1);
      v0[p] += a0;
      v1[p] += a1;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < HW; ++p) s += v0[p] + v1[p];
  for (int m = 1; m < lanes; m <<= 1) s += __shfl_xor(s, m);
  const float inv_n = 1.f / (float)(HW * cpg);
  const float mean = s * inv_n;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < HW; ++p) {
    const float d0 = v0[p] - mean, d1 = v1[p] - mean;
    q += d0 * d0 + d1 * d1;
  }
  for (int m = 1; m < lanes; m <<= 1) q += __shfl_xor(q, m);
  const float rstd = rsqrtf(q * inv_n + eps);
  const float a0 = gamma[2 * t] * rstd, a1 = gamma[2 * t + 1] * rstd;
  const float b0 = beta[2 * t] - mean * a0, b1 = beta[2 * t + 1] - mean * a1;
  T* ys = y + (size_t)sample * HW * c + 2 * t;
#pragma unroll
  for (int p = 0; p < HW; ++p) {
    float o0 = fmaf(v0[p], a0, b0), o1 = fmaf(v1[p], a1, b1);
    o0 = o0 >= 0.f ? o0 : o0 * slope;
    o1 = o1 >= 0.f ? o1 : o1 * slope;
    store2(ys + (size_t)p * c, o0, o1);
  }
}

// Per ROI: arg-max over the shots of each class logit and the box deltas that go with it (box_head.py:239-252),
// softmax over the two logits (inference.py:66), BoxCoder.decode of the class-1 deltas (box_coder.py:50-95), clip to
// the image (bounding_box.py:214-219).  pred: [shots][n*max_rois][pstride], columns 0..1 = logits, 2..9 = deltas.
// scores: class-1 probability, or -1 (= dropped by osd_rank_sort_gather) for ROIs past counts[image] and for
// probabilities that do not exceed `score_thresh` (inference.py:136).
template <typename T>
__global__ void box_decode_kernel(const T* __restrict__ pred, const float* __restrict__ rois, const int32_t* __restrict__ counts,
                                  float* __restrict__ scores, float* __restrict__ boxes, float* __restrict__ logits_out,
                                  float* __restrict__ reg_out, int n, int max_rois, int shots, int pstride, float wx,
                                  float wy, float ww, float wh, float clip, float img_h, float img_w, float score_thresh,
                                  const float* __restrict__ img_hw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * max_rois) return;
  const int img = i / max_rois, ri = i % max_rois;
  if (img_hw) {      // per-image true sizes of a padded batch
    img_h = img_hw[2 * img];
    img_w = img_hw[2 * img + 1];
  }
  const size_t shot_stride = (size_t)n * max_rois * pstride;
  const T* p0 = pred + (size_t)i * pstride;
  float l[2];
  int arg[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    l[j] = to_f32(p0[j]);
    arg[j] = 0;
    for (int s = 1; s < shots; ++s) {
      const float v = to_f32(p0[s * shot_stride + j]);
      if (v > l[j]) { l[j] = v; arg[j] = s; }
    }
  }
  float d[8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) d[4 * j + k] = to_f32(p0[arg[j] * shot_stride + 2 + 4 * j + k]);
  if (logits_out) {
    logits_out[2 * (size_t)i] = l[0];
    logits_out[2 * (size_t)i + 1] = l[1];
  }
  if (reg_out)
#pragma unroll
    for (int k = 0; k < 8; ++k) reg_out[8 * (size_t)i + k] = d[k];
  const float m = fmaxf(l[0], l[1]);
  const float e0 = expf(l[0] - m), e1 = expf(l[1] - m);
  const float prob = e1 / (e0 + e1);
  const float* b = rois + (size_t)i * 4;
  const float bw = b[2] - b[0] + 1.f, bh = b[3] - b[1] + 1.f;
  const float cx = b[0] + 0.5f * bw, cy = b[1] + 0.5f * bh;
  const float dx = d[4] / wx, dy = d[5] / wy;
  const float dw = fminf(d[6] / ww, clip), dh = fminf(d[7] / wh, clip);
  const float pcx = dx * bw + cx, pcy = dy * bh + cy;
  const float pw = expf(dw) * bw, ph = expf(dh) * bh;
  float ox1 = pcx - 0.5f * pw, oy1 = pcy - 0.5f * ph;
  float ox2 = pcx + 0.5f * pw - 1.f, oy2 = pcy + 0.5f * ph - 1.f;
  ox1 = fminf(fmaxf(ox1, 0.f), img_w - 1.f);
  oy1 = fminf(fmaxf(oy1, 0.f), img_h - 1.f);
  ox2 = fminf(fmaxf(ox2, 0.f), img_w - 1.f);
  oy2 = fminf(fmaxf(oy2, 0.f), img_h - 1.f);
  const bool live = (counts == nullptr || ri < counts[img]) && prob > score_thresh;
  scores[i] = live ? prob : -1.f;
  float* o = boxes + (size_t)i * 4;
  o[0] = ox1; o[1] = oy1; o[2] = ox2; o[3] = oy2;
}

// add_gt_proposals (modeling/rpn/fcos/inference.py:139-160): per image the kept proposals followed by the ground-truth
// boxes with score 1 (cat_boxlist((proposal, gt_box))); rows past the new count are zero.
__global__ void append_gt_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                 const int32_t* __restrict__ counts, const float* __restrict__ gt, const int32_t* __restrict__ gt_count,
                                 float* __restrict__ out_boxes, float* __restrict__ out_scores, int32_t* __restrict__ out_counts,
                                 int n, int cap, int max_gt, int out_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * out_cap) return;
  const int img = i / out_cap, r = i % out_cap;
  const int c = min(counts[img], cap), g = min(gt_count[img], max_gt);
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, sc = 0.f;
  if (r < c) {
    const float* s = boxes + ((size_t)img * cap + r) * 4;
    b0 = s[0]; b1 = s[1]; b2 = s[2]; b3 = s[3];
    sc = scores[(size_t)img * cap + r];
  } else if (r < c + g) {
    const float* s = gt + ((size_t)img * max_gt + (r - c)) * 4;
    b0 = s[0]; b1 = s[1]; b2 = s[2]; b3 = s[3];
    sc = 1.f;
  }
  float* o = out_boxes + (size_t)i * 4;
  o[0] = b0; o[1] = b1; o[2] = b2; o[3] = b3;
  out_scores[i] = sc;
  if (r == 0) out_counts[img] = c + g;
}

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int osd_append_gt_boxes(const float* boxes, const float* scores, const int32_t* counts, const float* gt_boxes,
                                   const int32_t* gt_count, float* out_boxes, float* out_scores, int32_t* out_counts, int n,
                                   int cap, int max_gt, void* stream) {
  if (!boxes || !scores || !counts || !gt_boxes || !gt_count || !out_boxes || !out_scores || !out_counts || cap < 0 || max_gt < 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "append_gt_boxes: bad args");
  const int out_cap = cap + max_gt;
  if (n == 0 || out_cap == 0) return OSD_OK;
  hipLaunchKernelGGL(append_gt_kernel, dim3(grid_for((long long)n * out_cap, 256)), dim3(256), 0, OSD_STREAM(stream), boxes,
                     scores, counts, gt_boxes, gt_count, out_boxes, out_scores, out_counts, n, cap, max_gt, out_cap);
  return osd_check_launch("append_gt_boxes");
}

extern "C" int osd_roi_pool_levels(int n_levels, const void* const* xs, const int32_t* hs, const int32_t* ws,
                                   const float* scales, const float* boxes, const int32_t* counts, void* y, int n, int c,
                                   int max_rois, int pool, int sampling_ratio, int y_stride, int32_t* level_out,
                                   int dtype, void* stream) {
  if (n_levels < 1 || n_levels > OSD_MAX_ROI_LEVELS || !xs || !hs || !ws || !scales || !boxes || !y)
    return osd_fail(OSD_ERR_INVALID_ARG, "roi_pool_levels: bad args");
  const int e = dtype == OSD_BF16 ? 8 : 4;
  if (c % e != 0 || y_stride < c || y_stride % e != 0 || pool < 1)
    return osd_fail(OSD_ERR_INVALID_ARG, "roi_pool_levels: channels %d / stride %d must be multiples of %d", c, y_stride, e);
  if (n * max_rois == 0) return OSD_OK;
  PoolLevels lv;
  lv.n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    if (!xs[l]) return osd_fail(OSD_ERR_INVALID_ARG, "roi_pool_levels: null level %d", l);
    lv.x[l] = xs[l]; lv.h[l] = hs[l]; lv.w[l] = ws[l]; lv.scale[l] = scales[l];
  }
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(roi_pool_levels_kernel<float>, dim3(n * max_rois), dim3(256), 0, OSD_STREAM(stream), lv, boxes, counts,
                       (float*)y, c, max_rois, pool, sampling_ratio, y_stride, level_out);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL(roi_pool_levels_kernel<__bf16>, dim3(n * max_rois), dim3(256), 0, OSD_STREAM(stream), lv, boxes, counts,
                       (__bf16*)y, c, max_rois, pool, sampling_ratio, y_stride, level_out);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "bad dtype %d", dtype);
  return osd_check_launch("roi_pool_levels");
}

extern "C" int osd_groupnorm_act_rois(const void* x, const void* addend, const float* gamma, const float* beta, void* y,
                                      int n_samples, int hw, int c, int groups, float eps, float slope, int rois_per_add,
                                      int add_stride, int add_offset, int dtype, void* stream) {
  if (!x || !gamma || !beta || !y) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_act_rois: null argument");
  if (hw != 49) return osd_fail(OSD_ERR_UNSUPPORTED, "groupnorm_act_rois: hw %d (only 7x7 ROI maps)", hw);
  const int cpg = groups > 0 ? c / groups : 0;
  const int lanes = cpg / 2;
  if (groups < 1 || c % groups != 0 || cpg % 2 != 0 || lanes < 1 || lanes > 64 || (lanes & (lanes - 1)) != 0 || c > 512 ||
      (c / 2) % 64 != 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "groupnorm_act_rois: c %d groups %d", c, groups);
  if (addend && rois_per_add < 1) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_act_rois: rois_per_add");
  if (n_samples == 0) return OSD_OK;
  if (dtype == OSD_F32)
    hipLaunchKernelGGL((gn_act_rois_kernel<float, 49>), dim3(n_samples), dim3(c / 2), 0, OSD_STREAM(stream), (const float*)x,
                       (const float*)addend, gamma, beta, (float*)y, c, groups, eps, slope, rois_per_add, add_stride,
                       add_offset);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL((gn_act_rois_kernel<__bf16, 49>), dim3(n_samples), dim3(c / 2), 0, OSD_STREAM(stream),
                       (const __bf16*)x, (const __bf16*)addend, gamma, beta, (__bf16*)y, c, groups, eps, slope, rois_per_add,
                       add_stride, add_offset);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "bad dtype %d", dtype);
  return osd_check_launch("groupnorm_act_rois");
}

extern "C" int osd_box_decode(const void* pred, const float* rois, const int32_t* counts, float* scores, float* boxes,
                              float* logits_out, float* reg_out, int n, int max_rois, int shots, int pred_stride,
                              const float* reg_weights, float img_h, float img_w, const float* img_hw, float score_thresh,
                              int dtype, void* stream) {
  if (!pred || !rois || !scores || !boxes || !reg_weights || shots < 1 || pred_stride < 10)
    return osd_fail(OSD_ERR_INVALID_ARG, "box_decode: bad args");
  if (n * max_rois == 0) return OSD_OK;
  const float clip = 4.135166556742356f;   // log(1000/16), box_coder.py:19
  const int g = grid_for((long long)n * max_rois, 256);
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(box_decode_kernel<float>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const float*)pred, rois, counts,
                       scores, boxes, logits_out, reg_out, n, max_rois, shots, pred_stride, reg_weights[0], reg_weights[1],
                       reg_weights[2], reg_weights[3], clip, img_h, img_w, score_thresh, img_hw);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL(box_decode_kernel<__bf16>, dim3(g), dim3(256), 0, OSD_STREAM(stream), (const __bf16*)pred, rois, counts,
                       scores, boxes, logits_out, reg_out, n, max_rois, shots, pred_stride, reg_weights[0], reg_weights[1],
                       reg_weights[2], reg_weights[3], clip, img_h, img_w, score_thresh, img_hw);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "bad dtype %d", dtype);
  return osd_check_launch("box_decode");
}
