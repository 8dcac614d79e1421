// Implicit-GEMM convolution for gfx950 (CDNA4) on MFMA — the K1/K2/K6 kernel family of SURVEY.md §2.1.
//
// GEMM view (per launch):  Y[m][co] = sum_k X[m @ tap(k)][ci(k)] * Wp[co][k],   m = (n, ho, wo),  k = (r, s, ci)
//   * activations are NHWC, so for one filter tap the Cin run of a pixel is contiguous: a K-slice of one stage is a
//     64- or 128-byte contiguous run per output pixel (im2col on the fly, never materialised);
//   * weights are pre-packed [Cout][R][S][Cin] (K contiguous) with FrozenBN folded in (osd_pack_conv_weight);
//   * MFMA operands are swapped: the WEIGHT tile is the MFMA "A" operand (rows = output channels) and the PIXEL
//     tile the "B" operand (columns = pixels).  A lane's 4 accumulator registers are then 4 consecutive output
//     channels of ONE pixel, i.e. one 16-byte (fp32) / 8-byte (bf16) NHWC store, and the epilogue (bias, residual /
//     nearest-2x top-down add, ReLU / exp) is applied in registers;
//   * both tiles live in LDS as [row][K bytes] rows of KB = 64 or 128 bytes with a 16-byte-chunk XOR swizzle that
//     makes every ds_read_b128 fragment read and every ds_write_b128 staging write bank-conflict free
//     (checked exhaustively against the gfx950 lane-group/bank rules);
//   * fp32 uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains) with the k index permuted inside each 16-wide k block
//     so one 16-byte LDS read feeds 4 MFMAs; bf16 uses v_mfma_f32_16x16x32_bf16 (8 k per lane = one 16-byte read);
//   * global -> register -> LDS staging, issued one K-stage ahead of the MFMAs (two LDS buffers, one barrier per
//     stage); zero padding and the M tail are handled by predicating the 16-byte loads;
//   * 1-D grid remapped so consecutive tiles (sharing the same pixel rows / halo) land on the same XCD's L2.
#include "osd_common.h"
#include "conv_params.h"
#include <string.h>

namespace {


template <int KB> __device__ __forceinline__ int swz(int row, int chunk) {
  if constexpr (KB == 64) {
    return row * 64 + ((chunk ^ ((-(row >> 2)) & 3)) << 4);
  } else {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  }
}

__device__ __forceinline__ uint4 relu_chunk(uint4 v, float) {
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = fmaxf(f[i], 0.f);
  return v;
}
__device__ __forceinline__ uint4 relu_chunk(uint4 v, __bf16) {
  unsigned* u = reinterpret_cast<unsigned*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] &= ~(((u[i] >> 15) & 0x00010001u) * 0xFFFFu);
  return v;
}

template <typename T, int BM, int BN, int KB, int WM, int WN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvKParams p) {
  constexpr int CH = KB / 16;                 // 16-byte chunks per tile row
  constexpr int EPC = 16 / (int)sizeof(T);    // elements per chunk
  constexpr int BKE = KB / (int)sizeof(T);    // K elements per stage
  constexpr int RPP = 256 / CH;               // tile rows staged per pass of the 256 threads
  constexpr int PA = (BM + RPP - 1) / RPP;
  constexpr int PB = (BN + RPP - 1) / RPP;
  constexpr int TM = BM / WM / 16;            // 16-pixel MFMA tiles per wave
  constexpr int TN = BN / WN / 16;            // 16-channel MFMA tiles per wave
  constexpr int STAGE = (BM + BN) * KB;
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD, give each XCD a contiguous run of tiles.
  int t;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = t % p.tilesN, tile_m = t / p.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  // ---- per-thread staging coordinates (fixed across K stages) ----
  const int srow = tid / CH;       // row within a pass
  const int schunk = tid % CH;     // 16-byte chunk within the row
  const T* a_ptr[PA];
  int a_hi0[PA], a_wi0[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = i * RPP + srow;
    const int m = m0 + row;
    if (row < BM && m < p.M) {
      const int n_img = m / p.HoWo;
      const int rem = m - n_img * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_ptr[i] = xg + (size_t)n_img * p.sN + schunk * EPC;
      a_hi0[i] = ho * p.sh - p.ph;
      a_wi0[i] = wo * p.sw - p.pw;
    } else {
      a_ptr[i] = xg;
      a_hi0[i] = -0x40000000;
      a_wi0[i] = 0;
    }
  }
  const T* b_ptr[PB];
  bool b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = i * RPP + srow;
    b_ok[i] = (row < BN) && (n0 + row < p.w_rows);
    b_ptr[i] = wg + (size_t)(b_ok[i] ? (n0 + row) : 0) * p.Ktot + schunk * EPC;
  }

  uint4 a_reg[PA], b_reg[PB];
  int kr = 0, ks = 0, kc = 0;  // filter row, filter col, channel offset of the NEXT stage to load

  auto load_stage = [&](int kt) {
    const int hoff = kr, woff = ks, c0 = kc;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int hi = a_hi0[i] + hoff, wi = a_wi0[i] + woff;
      const bool ok = ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *reinterpret_cast<const uint4*>(a_ptr[i] + (hi * p.sH + wi * p.sW + c0));
      a_reg[i] = v;
    }
    const int koff = kt * BKE;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b_ok[i]) v = *reinterpret_cast<const uint4*>(b_ptr[i] + koff);
      b_reg[i] = v;
    }
    kc += BKE;
    if (kc >= p.Cin) {
      kc = 0;
      if (++ks >= p.S) { ks = 0; ++kr; }
    }
  };

  auto store_stage = [&](int buf) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + BM * KB;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = i * RPP + srow;
      if (row < BM) {
        uint4 v = a_reg[i];
        if (p.relu_in) v = relu_chunk(v, T());
        *reinterpret_cast<uint4*>(xs + swz<KB>(row, schunk)) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = i * RPP + srow;
      if (row < BN) *reinterpret_cast<uint4*>(ws + swz<KB>(row, schunk)) = b_reg[i];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;   // row of the 16-row fragment this lane reads
  const int fkq = lane >> 4;    // 16-byte k chunk (0..3) within a 64-byte k block

  auto compute_stage = [&](int buf) {
    const char* xs = smem + buf * STAGE;
    const char* ws = xs + BM * KB;
#pragma unroll
    for (int kb = 0; kb < KB / 64; ++kb) {
      uint4 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wf[i] = *reinterpret_cast<const uint4*>(ws + swz<KB>((wn * TN + i) * 16 + frow, kb * 4 + fkq));
#pragma unroll
      for (int j = 0; j < TM; ++j)
        xf[j] = *reinterpret_cast<const uint4*>(xs + swz<KB>((wm * TM + j) * 16 + frow, kb * 4 + fkq));
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf[i]),
                                                                *reinterpret_cast<const bf16x8*>(&xf[j]), acc[i][j],
                                                                0, 0, 0);
          } else {
            const float* a = reinterpret_cast<const float*>(&wf[i]);
            const float* b = reinterpret_cast<const float*>(&xf[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };

  // ---- main loop: stage kt+1 is fetched from HBM/L2 while stage kt is multiplied ----
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int kt = 0; kt < p.KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < p.KT) load_stage(kt + 1);
    compute_stage(cur);
    if (kt + 1 < p.KT) store_stage(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias (+ residual) (+ activation), 4 consecutive channels of one pixel per lane and tile ----
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (wm * TM + j) * 16 + (lane & 15);
    if (m >= p.M) continue;
    size_t res_off = 0;
    if (p.res_mode == OSD_RES_SAME) {
      res_off = (size_t)m * p.res_stride;
    } else if (p.res_mode != OSD_RES_NONE) {
      const int n_img = m / p.HoWo;
      const int rem = m - n_img * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - (rem / p.Wo) * p.Wo;
      const bool down = p.res_mode == OSD_RES_DOWN2X;
      res_off = ((size_t)(n_img * p.res_h + (down ? (ho << 1) : (ho >> 1))) * p.res_w + (down ? (wo << 1) : (wo >> 1))) * p.res_stride;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int c = n0 + (wn * TN + i) * 16 + (lane >> 4) * 4;
      if (c >= p.Cout) continue;
      const float4 bv = *reinterpret_cast<const float4*>(p.bias + c);
      float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
      if (p.res_mode != OSD_RES_NONE) {
        if constexpr (sizeof(T) == 2) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rg + res_off + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        } else {
          const float4 rv = *reinterpret_cast<const float4*>(rg + res_off + c);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
      }
      if (p.act == OSD_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (p.act == OSD_ACT_EXP_SCALE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = expf(v[e] * (p.act_scale_dev ? *p.act_scale_dev : p.act_scale));
      }
      T* dst = yg + (size_t)m * p.out_stride + c;
      if constexpr (sizeof(T) == 2) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4*>(dst) = o;
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <typename T, int BM, int BN, int KB, int WM, int WN>
int launch_conv(const ConvKParams& pin, hipStream_t stream) {
  ConvKParams p = pin;
  p.tilesM = cdiv(p.M, BM);
  p.tilesN = cdiv(p.Cout, BN);
  constexpr int BKE = KB / (int)sizeof(T);
  if (p.Cin % BKE != 0) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: cin %d not a multiple of %d", p.Cin, BKE);
  p.KT = p.Ktot / BKE;
  constexpr int lds = 2 * (BM + BN) * KB;
  auto kern = conv_igemm_kernel<T, BM, BN, KB, WM, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const long long nblocks = (long long)p.tilesM * p.tilesN;
  if (nblocks <= 0 || nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad grid");
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), lds, stream, p);
  return osd_check_launch("conv_igemm");
}

// tile ids: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 256x16
template <typename T, int KB>
int dispatch_tile(int tile, const ConvKParams& p, hipStream_t s) {
  switch (tile) {
    case 0: return launch_conv<T, 128, 128, KB, 2, 2>(p, s);
    case 1: return launch_conv<T, 128, 64, KB, 4, 1>(p, s);
    case 2: return launch_conv<T, 64, 64, KB, 2, 2>(p, s);
    case 3: return launch_conv<T, 256, 16, KB, 4, 1>(p, s);
  }
  return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad tile id %d", tile);
}

int choose_tile(int M, int cout) {
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("OSD_CONV_TILE");
    forced = e ? atoi(e) : -1;
  }
  if (forced >= 0) return forced;
  if (cout <= 16) return 3;
  const long long t128 = (long long)cdiv(M, 128) * cdiv(cout, 128);
  if (cout % 128 == 0 && t128 >= 512) return 0;
  const long long t12864 = (long long)cdiv(M, 128) * cdiv(cout, 64);
  if (t12864 >= 512) return 1;
  return 2;
}

}  // namespace

extern "C" int osd_conv_algo_count(void) { return 64; }

extern "C" int osd_conv2d_fwd(const osd_conv_desc* d, const void* x, const void* w, const float* bias,
                              const void* res, const void* mask, const float* act_scale_dev, const osd_conv_src2* src2,
                              void* y, void* stream) {
  if (!d || !x || !w || !bias || !y) return osd_fail(OSD_ERR_INVALID_ARG, "conv: null argument");
  if (d->cout % 4 != 0 || d->out_stride % 4 != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "conv: cout/out_stride must be multiples of 4 (got %d/%d)", d->cout,
                    d->out_stride);
  if (d->res_mode != OSD_RES_NONE && (!res || d->res_stride % 4 != 0))
    return osd_fail(OSD_ERR_INVALID_ARG, "conv: residual requested without a valid residual tensor");
  if (d->res_mode < OSD_RES_NONE || d->res_mode > OSD_RES_DOWN2X) return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad res_mode %d", d->res_mode);
  if (d->res_mode == OSD_RES_UP2X && (d->res_h * 2 < d->ho || d->res_w * 2 < d->wo))
    return osd_fail(OSD_ERR_INVALID_ARG, "conv: the nearest-2x addend (%d x %d) is smaller than half the output (%d x %d)", d->res_h, d->res_w, d->ho, d->wo);
  if (d->res_mode == OSD_RES_DOWN2X && (d->res_h < 2 * d->ho - 1 || d->res_w < 2 * d->wo - 1))
    return osd_fail(OSD_ERR_INVALID_ARG, "conv: the every-other-pixel addend (%d x %d) does not cover the output (%d x %d)", d->res_h, d->res_w, d->ho, d->wo);
  if (d->w_rows < d->cout) return osd_fail(OSD_ERR_INVALID_ARG, "conv: w_rows < cout");
  const int epc = d->dtype == OSD_BF16 ? 8 : 4;
  // every staged 16-byte chunk must be 16-byte aligned: dense NHWC (pixel stride multiple of a chunk) or the stem's
  // packed NHWC4 form where only even pixels are ever addressed (s == 1, no padding, stride_w * pixel stride aligned)
  const bool w_ok = (d->in_stride_w % epc == 0) ||
                    (d->s == 1 && d->pad_w == 0 && (d->stride_w * d->in_stride_w) % epc == 0);
  if (!w_ok || d->in_stride_h % epc || d->in_stride_n % epc || d->cin % epc)
    return osd_fail(OSD_ERR_INVALID_ARG, "conv: input strides must keep 16-byte alignment");
  ConvKParams p;
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.mask = mask; p.y = y;
  p.H = d->h; p.W = d->w; p.Cin = d->cin; p.sN = d->in_stride_n; p.sH = d->in_stride_h; p.sW = d->in_stride_w;
  p.Ho = d->ho; p.Wo = d->wo; p.Cout = d->cout; p.HoWo = d->ho * d->wo;
  p.R = d->r; p.S = d->s; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
  p.w_rows = d->w_rows; p.Ktot = d->r * d->s * d->cin; p.out_stride = d->out_stride;
  p.res_mode = d->res_mode; p.res_h = d->res_h; p.res_w = d->res_w; p.res_stride = d->res_stride;
  p.act = d->act; p.act_scale = d->act_scale; p.act_scale_dev = act_scale_dev; p.relu_in = d->relu_in;
  const long long M = (long long)d->n * d->ho * d->wo;
  if (M <= 0 || M > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad M");
  p.M = (int)M; p.tilesM = p.tilesN = p.KT = 0; p.n_seg = 0; p.gn_n = p.gn_groups = 0;
  p.x2 = nullptr; p.w2 = nullptr; p.cin1 = p.x2_sN = p.x2_sH = p.x2_sW = p.st2 = 0;
  if (src2 != nullptr) {
    // second pixel source (see osd_conv_src2): K = cin + cin2, the packed weights hold both parts side by side
    if (!src2->x || d->r != 1 || d->s != 1 || d->pad_h || d->pad_w || d->relu_in)
      return osd_fail(OSD_ERR_INVALID_ARG, "conv: a second source needs a plain 1x1 conv and a tensor");
    if (src2->cin2 <= 0 || src2->cin2 % 64 || d->cin % 64 || src2->stride < 1 ||
        (d->ho - 1) * src2->stride >= src2->h || (d->wo - 1) * src2->stride >= src2->w)
      return osd_fail(OSD_ERR_INVALID_ARG, "conv: second source: channels in 64s, every output pixel inside its %d x %d map", src2->h,
                      src2->w);
    p.x2 = src2->x; p.cin1 = d->cin; p.st2 = src2->stride;
    p.x2_sW = src2->cin2; p.x2_sH = src2->w * src2->cin2; p.x2_sN = src2->h * src2->w * src2->cin2;
    p.Cin = d->cin + src2->cin2;
    p.w2 = src2->w2;
    p.Ktot = src2->w2 ? d->cin : p.Cin;       // row stride of w: its own part only, or both parts side by side
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int tile = choose_tile(p.M, p.Cout);
  if (src2 != nullptr && tile != 0 && tile != 2 && tile != 7) tile = 0;
  static int impl_env = -1;   // OSD_CONV_IMPL=regstage selects the first-generation register-staged kernel (A/B testing)
  if (impl_env < 0) {
    const char* e = getenv("OSD_CONV_IMPL");
    impl_env = (e && !strcmp(e, "regstage")) ? 1 : 0;
  }
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad dtype %d", d->dtype);
  int impl = impl_env, variant = 0;
  if (d->algo > 0) {
    const int a = d->algo - 1;
    if (a >= 64) return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad algo %d", d->algo);
    impl = a >> 5;                 // 0 = LDS-DMA kernel, 1 = register-staged kernel
    variant = (a >> 3) & 3;
    tile = a & 7;
    if (impl == 1 && variant == 2 && tile <= 1) {      // algos 49 / 50: the pixel-stationary pointwise kernel (conv_px.hip)
      if (d->dtype != OSD_BF16 || src2 != nullptr) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the pixel-stationary 1x1 kernel is bf16 only, one source");
      return osd_conv_px_launch(p, s, tile == 1);
    }
    if (impl == 1 && variant == 2 && tile == 2) {      // algo 51 (round 6): the prediction convs' patch kernel (conv_pred.hip)
      if (d->dtype != OSD_BF16 || src2 != nullptr) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the prediction-conv kernel is bf16 only, one source");
      return osd_conv_pred_launch(p, s);
    }
    if (impl == 1 && variant == 3 && tile <= 3) {      // algos 57 - 60 (round 6): small LDS-DMA tiles with deep rings (conv_igemm_dma.hip)
      if (src2 != nullptr) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the deep-ring tiles take one source");
      return osd_conv_dma_deep(d->dtype, tile == 0 ? 832 : tile == 1 ? 5 : tile == 2 ? 8 : 816, p, s);      // 64x32x8, 64x64x5, 64x64x8, 32x64x8
    }
    if (impl == 1 && variant == 1 && tile == 0)        // algo 41 was the persistent pointwise kernel of round 4 (conv_pw.hip: retired in
      return osd_fail(OSD_ERR_UNSUPPORTED, "conv: algo 41 (conv_pw) was retired in round 5");      // round 5, never the tuner's pick inside the step; git history)
    // the ping-pong 256x256 kernel and the row-reuse kernel without the software pipeline were retired in round 5: tile id 5 now
    // names the 128-pixel x 256-channel LDS-DMA tile, tile 6 / variant 0 conv_sp's 128 x 128 tile
    if (tile > 7 || (impl == 1 && (variant != 0 || tile > 3)))
      return osd_fail(OSD_ERR_UNSUPPORTED, "conv: algo %d not built", d->algo);
    if (p.Cout > 16 && tile == 3 && p.Cout > 64) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: skinny tile on a wide conv");
  }
  if (src2 != nullptr && (impl != 0 || tile == 5 || tile == 6))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv: a second source runs on the LDS-DMA kernel, tiles 0 / 2 / 7");
  if (impl == 0 && tile == 6) {
    if (d->dtype != OSD_BF16) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the row-reuse 3x3 kernel is bf16 only");
    return osd_conv_sp_launch(p, s, variant == 2, variant == 3, variant == 0);
  }
  if (impl == 0) return osd_conv_dma_dispatch(d->dtype, tile, variant, p, s);
  if (mask) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: mask epilogue only in the LDS-DMA kernel");
  if (d->dtype == OSD_F32) {
    return dispatch_tile<float, 64>(tile, p, s);
  } else if (d->dtype == OSD_BF16) {
    if (d->cin % 64 == 0) return dispatch_tile<__bf16, 128>(tile, p, s);
    return dispatch_tile<__bf16, 64>(tile, p, s);
  }
  return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad dtype %d", d->dtype);
}

// One launch over n_seg (x, y) pairs with the same conv geometry (channels, taps, stride, pad) but their own batch / spatial
// size AND their own weights / bias: the five FPN levels of an FCOS tower conv (fcos.py:83-99 applies the same modules to
// every level: they repeat one weight pointer), both towers at once, or the same layer of the target and the query backbone
// (generalized_rcnn.py:270-272: two R-50-FPN with separate parameters walk the same graph).  LDS-DMA kernels only.
struct ConvGnArgs {      // osd_conv2d_fwd_multi_gn's extra arguments (per segment; a null gn_us[i] leaves segment i out)
  const void* const* us; const float* const* abs; const float* const* gammas; float* const* wss; float* const* pws;
  int n, groups;
};

static int conv_fwd_multi(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys,
                          const void* const* residuals, const void* const* masks,
                          const float* const* act_scale_devs, const int32_t* ns, const int32_t* hs,
                          const int32_t* ws, const void* const* wts, const float* const* biases, const ConvGnArgs* gn, void* stream) {
  if (!d || !xs || !ys || !ns || !hs || !ws || !wts || !biases || n_seg < 1 || n_seg > kConvMaxSeg)
    return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: bad arguments (1..%d segments)", kConvMaxSeg);
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: bad dtype %d", d->dtype);
  if (d->relu_in) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_grouped: the relu_in prologue is not supported");
  if (d->cout % 4 != 0 || d->out_stride % 4 != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: cout/out_stride must be multiples of 4");
  if (d->res_mode != OSD_RES_NONE && d->res_mode != OSD_RES_SAME && d->res_mode != OSD_RES_UP2X && d->res_mode != OSD_RES_DOWN2X)
    return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: bad res_mode %d", d->res_mode);
  if (d->res_mode != OSD_RES_NONE && (!residuals || d->res_stride % 4 != 0))
    return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: residual requested without residual tensors");
  if (d->w_rows < d->cout) return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: w_rows < cout");
  const int epc = d->dtype == OSD_BF16 ? 8 : 4;
  if (d->cin % epc) return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: cin must keep 16-byte alignment");
  ConvKParams p;
  p.w = wts[0]; p.bias = biases[0];
  p.Cin = d->cin; p.sW = d->cin; p.Cout = d->cout;
  p.R = d->r; p.S = d->s; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
  p.w_rows = d->w_rows; p.Ktot = d->r * d->s * d->cin; p.out_stride = d->out_stride;
  p.res_mode = d->res_mode; p.res_h = 0; p.res_w = 0; p.res_stride = d->res_stride;
  p.act = d->act; p.act_scale = d->act_scale; p.act_scale_dev = nullptr; p.relu_in = 0;
  p.tilesM = p.tilesN = p.KT = 0;
  p.n_seg = n_seg;
  p.x2 = nullptr; p.w2 = nullptr; p.cin1 = p.x2_sN = p.x2_sH = p.x2_sW = p.st2 = 0;
  p.gn_n = gn ? gn->n : 0; p.gn_groups = gn ? gn->groups : 0;
  if (gn && (gn->groups < 1 || gn->n < 1 || !gn->wss))
    return osd_fail(OSD_ERR_INVALID_ARG, "conv_multi_gn: bad statistics arguments");
  long long mtot = 0;
  for (int i = 0; i < kConvMaxSeg; ++i) {
    const int j = i < n_seg ? i : 0;
    ConvSeg& sg = p.seg[i];
    if (!xs[j] || !ys[j] || !wts[j] || !biases[j] || ns[j] <= 0 || hs[j] <= 0 || ws[j] <= 0)
      return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: bad segment %d", j);
    sg.x = xs[j]; sg.y = ys[j]; sg.w = wts[j]; sg.bias = biases[j];
    sg.res = d->res_mode != OSD_RES_NONE ? residuals[j] : nullptr;
    if (d->res_mode != OSD_RES_NONE && !sg.res) return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: null residual %d", j);
    sg.mask = masks ? masks[j] : nullptr;
    sg.act_scale_dev = act_scale_devs ? act_scale_devs[j] : nullptr;
    sg.H = hs[j]; sg.W = ws[j];
    sg.Ho = (hs[j] + 2 * d->pad_h - d->r) / d->stride_h + 1;
    sg.Wo = (ws[j] + 2 * d->pad_w - d->s) / d->stride_w + 1;
    const long long M = (long long)ns[j] * sg.Ho * sg.Wo;
    if (sg.Ho <= 0 || sg.Wo <= 0 || M > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: bad M in segment %d", j);
    if (d->res_mode == OSD_RES_UP2X && ((sg.Ho | sg.Wo) & 1))
      return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: the nearest-2x addend needs an even output size (segment %d: %d x %d)", j, sg.Ho, sg.Wo);
    sg.M = (int)M; sg.sH = ws[j] * d->cin; sg.sN = hs[j] * ws[j] * d->cin; sg.tile_begin = 0;
    sg.gn.u = nullptr; sg.gn.ab = nullptr; sg.gn.gamma = nullptr; sg.gn.ws = nullptr; sg.gn.pw = nullptr;
    if (gn && gn->us && gn->us[j]) {          // backward statistics
      if (!gn->abs || !gn->gammas || !gn->pws || !gn->abs[j] || !gn->gammas[j] || !gn->wss[j] || !gn->pws[j] || ns[j] != gn->n)
        return osd_fail(OSD_ERR_INVALID_ARG, "conv_multi_gn: segment %d: null statistics argument or a batch other than gn_n", j);
      sg.gn.u = gn->us[j]; sg.gn.ab = gn->abs[j]; sg.gn.gamma = gn->gammas[j]; sg.gn.ws = gn->wss[j]; sg.gn.pw = gn->pws[j];
    } else if (gn && gn->wss[j]) {            // forward statistics: sums of the outputs and of their squares
      if (ns[j] != gn->n) return osd_fail(OSD_ERR_INVALID_ARG, "conv_multi_gn: segment %d: a batch other than gn_n", j);
      sg.gn.ws = gn->wss[j];
    }
    if (i < n_seg) mtot += M;
  }
  if (mtot > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: too many pixels");
  // single-problem fields mirror segment 0 (unused by the kernel when n_seg > 0)
  p.x = p.seg[0].x; p.y = p.seg[0].y; p.res = p.seg[0].res; p.mask = p.seg[0].mask;
  p.H = p.seg[0].H; p.W = p.seg[0].W; p.Ho = p.seg[0].Ho; p.Wo = p.seg[0].Wo; p.HoWo = p.Ho * p.Wo;
  p.sN = p.seg[0].sN; p.sH = p.seg[0].sH; p.M = (int)mtot;
  int tile = choose_tile((int)mtot, p.Cout), variant = 0;
  if (d->algo == 51) {      // the prediction convs' patch kernel (conv_pred.hip): all levels of a tower output in one launch
    if (d->dtype != OSD_BF16 || gn) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_grouped: the prediction-conv kernel is bf16 only, no statistics");
    return osd_conv_pred_launch(p, reinterpret_cast<hipStream_t>(stream));
  }
  if (d->algo > 0) {
    const int a = d->algo - 1;
    if (a >= 32) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_grouped: algo %d is not an LDS-DMA algorithm", d->algo);
    variant = (a >> 3) & 3;
    tile = a & 7;
    if (tile > 7) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_grouped: algo %d not built", d->algo);
    if (tile == 3 && p.Cout > 64) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_grouped: skinny tile on a wide conv");
  }
  if (gn && !(tile == 6 && variant == 1))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_multi_gn: only the software-pipelined 3x3 kernel (algo 15) gathers GroupNorm statistics");
  if (tile == 6) {
    if (d->dtype != OSD_BF16) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_grouped: the row-reuse 3x3 kernel is bf16 only");
    return osd_conv_sp_launch(p, reinterpret_cast<hipStream_t>(stream), variant == 2, variant == 3, variant == 0);
  }
  return osd_conv_dma_dispatch(d->dtype, tile, variant, p, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int osd_conv2d_fwd_multi(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys,
                                    const void* const* residuals, const void* const* masks,
                                    const float* const* act_scale_devs, const int32_t* ns, const int32_t* hs,
                                    const int32_t* ws, const void* const* wts, const float* const* biases, void* stream) {
  return conv_fwd_multi(d, n_seg, xs, ys, residuals, masks, act_scale_devs, ns, hs, ws, wts, biases, nullptr, stream);
}

// osd_conv2d_fwd_multi for a data-gradient conv whose outputs dt feed a GroupNorm + ReLU backward: the epilogue also gathers,
// for every segment with gn_us[i] != nullptr, the sums osd_groupnorm_relu_bwd_levels_fused reads (conv_params.h: ConvGnb)
extern "C" int osd_conv2d_fwd_multi_gn(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys, const int32_t* ns,
                                       const int32_t* hs, const int32_t* ws, const void* const* wts, const float* const* biases,
                                       const void* const* gn_us, const float* const* gn_abs, const float* const* gn_gammas,
                                       float* const* gn_wss, float* const* gn_pws, int gn_n, int gn_groups, void* stream) {
  ConvGnArgs gn{gn_us, gn_abs, gn_gammas, gn_wss, gn_pws, gn_n, gn_groups};
  if (d && (d->res_mode != OSD_RES_NONE || d->act != OSD_ACT_NONE))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_multi_gn: no residual / activation");
  if (!gn_wss) return osd_fail(OSD_ERR_INVALID_ARG, "conv_multi_gn: gn_wss is null");
  return conv_fwd_multi(d, n_seg, xs, ys, nullptr, nullptr, nullptr, ns, hs, ws, wts, biases, &gn, stream);
}

// the same with ONE weight / bias pointer for all pairs (the FPN levels of one conv)
extern "C" int osd_conv2d_fwd_grouped(const osd_conv_desc* d, int n_seg, const void* const* xs, void* const* ys,
                                      const void* const* residuals, const void* const* masks,
                                      const float* const* act_scale_devs, const int32_t* ns, const int32_t* hs,
                                      const int32_t* ws, const void* w, const float* bias, void* stream) {
  if (n_seg < 1 || n_seg > kConvMaxSeg || !w || !bias)
    return osd_fail(OSD_ERR_INVALID_ARG, "conv_grouped: bad arguments (1..%d segments)", kConvMaxSeg);
  const void* wts[kConvMaxSeg];
  const float* bs[kConvMaxSeg];
  for (int i = 0; i < n_seg; ++i) { wts[i] = w; bs[i] = bias; }
  return osd_conv2d_fwd_multi(d, n_seg, xs, ys, residuals, masks, act_scale_devs, ns, hs, ws, wts, bs, stream);
}
