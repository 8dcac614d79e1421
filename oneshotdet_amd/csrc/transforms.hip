// transforms.hip — the reference's input transforms fused for gfx950: Resize (PIL bilinear, 8-bit
// fixed point) + RandomHorizontalFlip + ToTensor + Normalize(BGR255 - mean) + the zero padding of to_image_list, written
// straight into the batch tensor the backbones read.  Reference: data/transforms/transforms.py:27-92,
// data/transforms/build.py:39-46, structures/image_list.py:52-70; `F.resize` = PIL.Image.resize(BILINEAR) =
// Pillow src/libImaging/Resample.c (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
// Vertical_8bpc): restated in oracle/transforms_ref.py, which is bit-exact against Pillow and against fixtures recorded
// through the reference's transforms.  These kernels are bit-exact too: integer resampling, and the float steps are the
// reference's float32 operations one by one with contraction off (__f*_rn intrinsics).
#include "osd_common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1), one thread per output index.
// IEEE double operations in the source's order (no fused multiply-add), so the tables equal the host library's.
__global__ void resize_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ kk, int* __restrict__ bounds) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = __ddiv_rn((double)in_size, (double)out_size);
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = filterscale;                       // bilinear: support 1.0 * filterscale
  const double ss = __ddiv_rn(1.0, filterscale);
  const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
  int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
    if (a < 0.0) a = -a;
    const double w = a < 1.0 ? __dsub_rn(1.0, a) : 0.0;
    ww = __dadd_rn(ww, w);
  }
  int* k = kk + (size_t)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double a = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
      if (a < 0.0) a = -a;
      w = a < 1.0 ? __dsub_rn(1.0, a) : 0.0;
      if (ww != 0.0) w = __ddiv_rn(w, ww);
    }
    const double sc = __dmul_rn(w, (double)(1 << kPrecisionBits));
    k[x] = w < 0.0 ? (int)__dadd_rn(-0.5, sc) : (int)__dadd_rn(0.5, sc);
  }
  bounds[xx * 2 + 0] = xmin;
  bounds[xx * 2 + 1] = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ImagingResampleHorizontal_8bpc on an RGB image: tmp[y][xo][c] (uint8)
__global__ void resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp, int in_h, int in_w,
                                int out_w, int ksize, const int* __restrict__ kk, const int* __restrict__ bounds) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)in_h * out_w) return;
  const int y = (int)(i / out_w), xo = (int)(i - (long long)y * out_w);
  const int xmin = bounds[xo * 2], xmax = bounds[xo * 2 + 1];
  const int* k = kk + (size_t)xo * ksize;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  const unsigned char* row = src + ((size_t)y * in_w + xmin) * 3;
  for (int x = 0; x < xmax; ++x) {
    const int w = k[x];
    s0 += row[x * 3 + 0] * w; s1 += row[x * 3 + 1] * w; s2 += row[x * 3 + 2] * w;
  }
  unsigned char* o = tmp + (size_t)i * 3;
  o[0] = (unsigned char)clip8(s0); o[1] = (unsigned char)clip8(s1); o[2] = (unsigned char)clip8(s2);
}

struct NormParams { float mean[3], stdv[3]; int to_bgr255; };

// ImagingResampleVertical_8bpc + hflip + ToTensor + Normalize + zero padding, one thread per pixel of the destination slot
// (dst_h x dst_w, image at (pad_t, pad_l)).  LAYOUT 0: fp32 NCHW [n][3][dst_h][dst_w] (the reference's batch tensor);
// LAYOUT 1: `T` NHWC4 [n][dst_h][dst_w][4] (the stem conv's padded input, see osd_pack_image).
template <typename T, int LAYOUT>
__global__ void resize_v_norm_kernel(const unsigned char* __restrict__ tmp, void* __restrict__ dstv, int tmp_h, int out_h,
                                     int out_w, int flip, int vertical, int ksize, const int* __restrict__ kk,
                                     const int* __restrict__ bounds, NormParams np, int batch_index, int dst_h, int dst_w,
                                     int pad_t, int pad_l) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)dst_h * dst_w) return;
  const int dy = (int)(i / dst_w), dx = (int)(i - (long long)dy * dst_w);
  const int y = dy - pad_t, x = dx - pad_l;
  float v[3] = {0.f, 0.f, 0.f};
  const bool inside = (unsigned)y < (unsigned)out_h && (unsigned)x < (unsigned)out_w;
  if (inside) {
    const int xs = flip ? out_w - 1 - x : x;
    int px[3];
    if (vertical) {
      const int ymin = bounds[y * 2], ymax = bounds[y * 2 + 1];
      const int* k = kk + (size_t)y * ksize;
      int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
      for (int t = 0; t < ymax; ++t) {
        const unsigned char* p = tmp + ((size_t)(ymin + t) * out_w + xs) * 3;
        const int w = k[t];
        s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
      }
      px[0] = clip8(s0); px[1] = clip8(s1); px[2] = clip8(s2);
    } else {
      const unsigned char* p = tmp + ((size_t)y * out_w + xs) * 3;
      px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
    }
#pragma unroll
    for (int co = 0; co < 3; ++co) {
      const int ci = np.to_bgr255 ? 2 - co : co;                          // image[[2, 1, 0]]
      float f = __fdiv_rn((float)px[ci], 255.0f);                         // ToTensor: .float().div(255)
      if (np.to_bgr255) f = __fmul_rn(f, 255.0f);                         // * 255
      v[co] = __fdiv_rn(__fsub_rn(f, np.mean[co]), np.stdv[co]);          // t.sub_(m).div_(s)
    }
  }
  if (LAYOUT == 0) {
    float* dst = reinterpret_cast<float*>(dstv) + (size_t)batch_index * 3 * dst_h * dst_w;
#pragma unroll
    for (int co = 0; co < 3; ++co) dst[((size_t)co * dst_h + dy) * dst_w + dx] = v[co];
  } else {
    T* dst = reinterpret_cast<T*>(dstv) + ((size_t)batch_index * dst_h * dst_w + (size_t)i) * 4;
    dst[0] = from_f32<T>(v[0]); dst[1] = from_f32<T>(v[1]); dst[2] = from_f32<T>(v[2]); dst[3] = from_f32<T>(0.f);
  }
}

inline int resize_ksize(int in_size, int out_size) {
  const double scale = (double)in_size / (double)out_size;
  const double support = scale < 1.0 ? 1.0 : scale;
  return (int)ceil(support) * 2 + 1;
}
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int64_t osd_image_transform_workspace_bytes(int in_h, int in_w, int out_h, int out_w) {
  if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return 0;
  const size_t kw = resize_ksize(in_w, out_w), kh = resize_ksize(in_h, out_h);
  return (int64_t)(al256((size_t)out_w * (kw + 2) * 4) + al256((size_t)out_h * (kh + 2) * 4) + al256((size_t)in_h * out_w * 3));
}

extern "C" int osd_image_transform(const uint8_t* src_rgb_hwc, int in_h, int in_w, int out_h, int out_w, int flip,
                                   int to_bgr255, const float* mean3, const float* std3, void* dst, int layout, int dtype,
                                   int batch_index, int dst_h, int dst_w, int pad_t, int pad_l, void* workspace,
                                   void* stream) {
  if (!src_rgb_hwc || !dst || !workspace || !mean3 || !std3) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: null argument");
  if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || batch_index < 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: bad size");
  if (pad_t < 0 || pad_l < 0 || pad_t + out_h > dst_h || pad_l + out_w > dst_w)
    return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: the %d x %d image does not fit the %d x %d slot at (%d, %d)", out_h, out_w,
                    dst_h, dst_w, pad_t, pad_l);
  if (layout != 0 && layout != 1) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: layout 0 (fp32 NCHW) or 1 (NHWC4)");
  if (layout == 1 && dtype != OSD_F32 && dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: bad dtype");
  hipStream_t st = OSD_STREAM(stream);
  const int kw = resize_ksize(in_w, out_w), kh = resize_ksize(in_h, out_h);
  char* p = static_cast<char*>(workspace);
  int* kkw = reinterpret_cast<int*>(p); int* bw = kkw + (size_t)out_w * kw; p += al256((size_t)out_w * (kw + 2) * 4);
  int* kkh = reinterpret_cast<int*>(p); int* bh = kkh + (size_t)out_h * kh; p += al256((size_t)out_h * (kh + 2) * 4);
  unsigned char* tmp = reinterpret_cast<unsigned char*>(p);
  const bool horiz = out_w != in_w, vert = out_h != in_h;      // ImagingResample skips a pass whose size does not change
  const unsigned char* vsrc = src_rgb_hwc;
  if (horiz) {
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cdiv(out_w, 256)), dim3(256), 0, st, in_w, out_w, kw, kkw, bw);
    const long long n = (long long)in_h * out_w;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src_rgb_hwc, tmp, in_h, in_w, out_w, kw,
                       (const int*)kkw, (const int*)bw);
    vsrc = tmp;
  }
  if (vert) hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cdiv(out_h, 256)), dim3(256), 0, st, in_h, out_h, kh, kkh, bh);
  int rc = osd_check_launch("image_transform: resample");
  if (rc) return rc;
  NormParams np;
  for (int c = 0; c < 3; ++c) { np.mean[c] = mean3[c]; np.stdv[c] = std3[c]; }
  np.to_bgr255 = to_bgr255;
  const long long n = (long long)dst_h * dst_w;
  const dim3 grid((unsigned)((n + 255) / 256));
#define OSD_TV(TT, LL)                                                                                                     \
  hipLaunchKernelGGL((resize_v_norm_kernel<TT, LL>), grid, dim3(256), 0, st, vsrc, dst, in_h, out_h, out_w, flip, vert ? 1 : 0, \
                     kh, (const int*)kkh, (const int*)bh, np, batch_index, dst_h, dst_w, pad_t, pad_l)
  if (layout == 0) OSD_TV(float, 0);
  else if (dtype == OSD_F32) OSD_TV(float, 1);
  else OSD_TV(__bf16, 1);
#undef OSD_TV
  return osd_check_launch("image_transform");
}
