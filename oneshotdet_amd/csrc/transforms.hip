// transforms.hip — the reference's input transforms fused for gfx950: Resize (PIL bilinear, 8-bit
// fixed point) + RandomHorizontalFlip + ToTensor + Normalize(BGR255 - mean) + the zero padding of to_image_list, written
// straight into the batch tensor the backbones read.  Reference: data/transforms/transforms.py:27-92,
// data/transforms/build.py:39-46, structures/image_list.py:52-70; `F.resize` = PIL.Image.resize(BILINEAR) =
// Pillow src/libImaging/Resample.c (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
// Vertical_8bpc): restated in oracle/transforms_ref.py, which is bit-exact against Pillow and against fixtures recorded
// through the reference's transforms.  These kernels are bit-exact too: integer resampling, and the float steps are the
// reference's float32 operations one by one with contraction off (__f*_rn intrinsics).
#include "osd_common.h"
#include <cstddef>

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1), one thread per output index.
// IEEE double operations in the source's order (no fused multiply-add), so the tables equal the host library's.
__device__ __forceinline__ void resize_coeffs_one(int xx, int in_size, int out_size, int ksize, int* __restrict__ kk,
                                                  int* __restrict__ bounds) {
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

// Up to kMaxImages images per launch chain: the per-image geometry and workspace pointers travel in the kernarg segment,
// blockIdx.y picks the image (a batch of 8 is 3 launches, not 8 x 4: at 3-10 us per launch the per-image chain was
// launch-bound at 0.05 of the HBM roofline).
constexpr int kMaxImages = 16;
struct ImageEntry {
  const unsigned char* src;      // RGB uint8 [in_h][in_w][3]
  unsigned char* tmp;            // horizontal pass output [in_h][out_w][3]
  int* kkw; int* bw; int* kkh; int* bh;
  int in_h, in_w, out_h, out_w, kw, kh, flip, horiz, vert, batch_index;
};
struct ImageTable { ImageEntry im[kMaxImages]; };

// the table entry of this workgroup's image, read with scalar loads from the kernarg segment (indexing the by-value struct
// with a runtime index would copy the table to scratch)
__device__ __forceinline__ ImageEntry image_entry(int idx) {
  typedef const __attribute__((address_space(4))) char* kcp;
  typedef unsigned long long u64;
  kcp b = (kcp)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)idx * sizeof(ImageEntry);
#define OSD_IMF(type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(b + offsetof(ImageEntry, field)))
  ImageEntry e;
  e.src = (const unsigned char*)OSD_IMF(u64, src); e.tmp = (unsigned char*)OSD_IMF(u64, tmp);
  e.kkw = (int*)OSD_IMF(u64, kkw); e.bw = (int*)OSD_IMF(u64, bw); e.kkh = (int*)OSD_IMF(u64, kkh); e.bh = (int*)OSD_IMF(u64, bh);
  e.in_h = OSD_IMF(int, in_h); e.in_w = OSD_IMF(int, in_w); e.out_h = OSD_IMF(int, out_h); e.out_w = OSD_IMF(int, out_w);
  e.kw = OSD_IMF(int, kw); e.kh = OSD_IMF(int, kh); e.flip = OSD_IMF(int, flip); e.horiz = OSD_IMF(int, horiz);
  e.vert = OSD_IMF(int, vert); e.batch_index = OSD_IMF(int, batch_index);
#undef OSD_IMF
  return e;
}

__global__ void resize_coeffs_kernel(ImageTable) {        // blockIdx.z: 0 = horizontal tables, 1 = vertical
  const ImageEntry e = image_entry(blockIdx.y);
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.z == 0) {
    if (e.horiz && xx < e.out_w) resize_coeffs_one(xx, e.in_w, e.out_w, e.kw, e.kkw, e.bw);
  } else {
    if (e.vert && xx < e.out_h) resize_coeffs_one(xx, e.in_h, e.out_h, e.kh, e.kkh, e.bh);
  }
}

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ImagingResampleHorizontal_8bpc on an RGB image: tmp[y][xo][c] (uint8)
__global__ void resize_h_kernel(ImageTable) {
  const ImageEntry e = image_entry(blockIdx.y);
  const unsigned char* __restrict__ src = e.src;
  unsigned char* __restrict__ tmp = e.tmp;
  const int in_h = e.in_h, in_w = e.in_w, out_w = e.out_w, ksize = e.kw;
  const int* __restrict__ kk = e.kkw;
  const int* __restrict__ bounds = e.bw;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (!e.horiz || i >= (long long)in_h * out_w) return;
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
__global__ void resize_v_norm_kernel(ImageTable, void* __restrict__ dstv, NormParams np, int dst_h, int dst_w, int pad_t,
                                     int pad_l) {
  const ImageEntry e = image_entry(blockIdx.y);
  const unsigned char* __restrict__ tmp = e.horiz ? e.tmp : e.src;
  const int out_h = e.out_h, out_w = e.out_w, flip = e.flip, vertical = e.vert, ksize = e.kh, batch_index = e.batch_index;
  const int* __restrict__ kk = e.kkh;
  const int* __restrict__ bounds = e.bh;
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

extern "C" int osd_image_transform_batch(int n_images, const uint8_t* const* srcs_rgb_hwc, const int32_t* in_hs,
                                         const int32_t* in_ws, const int32_t* out_hs, const int32_t* out_ws, const int32_t* flips,
                                         int to_bgr255, const float* mean3, const float* std3, void* dst, int layout, int dtype,
                                         int first_batch_index, int dst_h, int dst_w, int pad_t, int pad_l, void* workspace,
                                         void* stream) {
  if (n_images < 0 || !srcs_rgb_hwc || !in_hs || !in_ws || !out_hs || !out_ws || !dst || !workspace || !mean3 || !std3)
    return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: null argument");
  if (first_batch_index < 0 || pad_t < 0 || pad_l < 0) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: bad size");
  if (layout != 0 && layout != 1) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: layout 0 (fp32 NCHW) or 1 (NHWC4)");
  if (layout == 1 && dtype != OSD_F32 && dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: bad dtype");
  hipStream_t st = OSD_STREAM(stream);
  NormParams np;
  for (int c = 0; c < 3; ++c) { np.mean[c] = mean3[c]; np.stdv[c] = std3[c]; }
  np.to_bgr255 = to_bgr255;
  char* p = static_cast<char*>(workspace);
  for (int base = 0; base < n_images; base += kMaxImages) {
    const int cnt = n_images - base < kMaxImages ? n_images - base : kMaxImages;
    ImageTable tab;
    int max_len = 1;                 // longest coefficient table
    long long max_h = 0;             // most horizontal-pass pixels
    bool any_h = false, any_tab = false;
    for (int j = 0; j < kMaxImages; ++j) {
      const int i = base + (j < cnt ? j : 0);
      ImageEntry& e = tab.im[j];
      const int in_h = in_hs[i], in_w = in_ws[i], out_h = out_hs[i], out_w = out_ws[i];
      if (j < cnt) {
        if (!srcs_rgb_hwc[i] || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0)
          return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: bad size of image %d", i);
        if (pad_t + out_h > dst_h || pad_l + out_w > dst_w)
          return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: the %d x %d image does not fit the %d x %d slot at (%d, %d)", out_h,
                          out_w, dst_h, dst_w, pad_t, pad_l);
      }
      e.src = srcs_rgb_hwc[i];
      e.in_h = in_h; e.in_w = in_w; e.out_h = out_h; e.out_w = out_w;
      e.kw = resize_ksize(in_w, out_w); e.kh = resize_ksize(in_h, out_h);
      e.flip = flips ? flips[i] : 0;
      e.horiz = out_w != in_w; e.vert = out_h != in_h;      // ImagingResample skips a pass whose size does not change
      e.batch_index = first_batch_index + i;
      if (j < cnt) {               // workspace slices in the layout osd_image_transform_workspace_bytes sums up
        e.kkw = reinterpret_cast<int*>(p); e.bw = e.kkw + (size_t)out_w * e.kw; p += al256((size_t)out_w * (e.kw + 2) * 4);
        e.kkh = reinterpret_cast<int*>(p); e.bh = e.kkh + (size_t)out_h * e.kh; p += al256((size_t)out_h * (e.kh + 2) * 4);
        e.tmp = reinterpret_cast<unsigned char*>(p); p += al256((size_t)in_h * out_w * 3);
        if (e.horiz) { any_h = any_tab = true; if (out_w > max_len) max_len = out_w; if ((long long)in_h * out_w > max_h) max_h = (long long)in_h * out_w; }
        if (e.vert) { any_tab = true; if (out_h > max_len) max_len = out_h; }
      } else {
        e = tab.im[0];
      }
    }
    if (any_tab) hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cdiv(max_len, 256), cnt, 2), dim3(256), 0, st, tab);
    if (any_h) hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((max_h + 255) / 256), cnt), dim3(256), 0, st, tab);
    int rc = osd_check_launch("image_transform: resample");
    if (rc) return rc;
    const long long n = (long long)dst_h * dst_w;
    const dim3 grid((unsigned)((n + 255) / 256), cnt);
#define OSD_TV(TT, LL) hipLaunchKernelGGL((resize_v_norm_kernel<TT, LL>), grid, dim3(256), 0, st, tab, dst, np, dst_h, dst_w, pad_t, pad_l)
    if (layout == 0) OSD_TV(float, 0);
    else if (dtype == OSD_F32) OSD_TV(float, 1);
    else OSD_TV(__bf16, 1);
#undef OSD_TV
    rc = osd_check_launch("image_transform");
    if (rc) return rc;
  }
  return OSD_OK;
}

extern "C" int osd_image_transform(const uint8_t* src_rgb_hwc, int in_h, int in_w, int out_h, int out_w, int flip,
                                   int to_bgr255, const float* mean3, const float* std3, void* dst, int layout, int dtype,
                                   int batch_index, int dst_h, int dst_w, int pad_t, int pad_l, void* workspace,
                                   void* stream) {
  if (batch_index < 0) return osd_fail(OSD_ERR_INVALID_ARG, "image_transform: bad size");
  const uint8_t* srcs[1] = {src_rgb_hwc};
  const int32_t ih[1] = {in_h}, iw[1] = {in_w}, oh[1] = {out_h}, ow[1] = {out_w}, fl[1] = {flip};
  return osd_image_transform_batch(1, srcs, ih, iw, oh, ow, fl, to_bgr255, mean3, std3, dst, layout, dtype, batch_index, dst_h,
                                   dst_w, pad_t, pad_l, workspace, stream);
}
