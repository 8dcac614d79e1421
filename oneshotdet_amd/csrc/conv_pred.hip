// conv_pred — the FCOS prediction convs' FORWARD (fcos.py:50-61, 91-97: cls_logits + centerness as one 2-output conv, bbox_pred with
// exp(scale_l * x)): 3x3 / stride 1 / pad 1, 256 -> (2 | 4) channels over the five FPN levels of a tower output, bf16, gfx950.
// Algo 51 of osd_conv2d_fwd / osd_conv2d_fwd_grouped (round 6).
//
// Why a kernel of its own: with 4 output channels the implicit GEMM has N = 16 (one MFMA column tile, 12 of them padding) and is
// bound by what a CU takes in, not by MFMAs.  The 256 x 16 LDS-DMA tile re-reads every pixel's 512-byte channel run once per filter
// tap through the texture path: 136,512 pixels x 2,304 x 2 B = 629 MB of L2 -> LDS traffic per launch for 70 MB of input, ~65 us
// at the ~10 TB/s the chip's CUs take in together (profiles/r6_train_bf16_conv_layers.md: 69.1 / 64.3 us).  Here a workgroup owns an
// 8 x 32 patch of output pixels of ONE image, stages the 10 x 34 input patch (halo included) once per 64-channel slab in LDS and
// takes all nine taps from it: 174 KB per 256 outputs instead of 1,180 KB.
//   * 256 threads = 4 waves; wave w owns output rows 2w, 2w + 1 of the patch = 4 MFMA pixel fragments of 16 consecutive pixels;
//   * MFMA operands as everywhere in this library: the weight tile [16 co][32 k] is A, a pixel fragment [32 k][16 px] is B, so
//     lanes 0..15 end up with channels 0..3 of one pixel each: one 8-byte NHWC store per pixel (lanes 16..63 hold padding rows);
//   * K order: 64-channel slab, tap, 32-channel half — another fp32 summation order than the tile kernels (same bar as every
//     other algorithm: the fp32 reference within the bf16 tolerance, tests/test_gpu_kernels.py);
//   * staging is global -> registers -> LDS (16 bytes per lane, unconditional loads from a clamped address + a select: no branch
//     around a load), the next slab's loads in flight while the current one is multiplied; ONE LDS stage of 60.5 KB, so that two
//     workgroups share a CU and one's memory latency hides under the other's MFMAs (a slab's 72 MFMAs per wave are ~0.6 us, an HBM
//     round trip 1 - 2 us: with two stages and one workgroup per CU the kernel ran at 58 us against 66 for the tile kernel);
//   * LDS rows are 128 bytes (one pixel's / one (tap, co) row's slab) with the 16-byte chunk index XORed by bits 1..3 of the row
//     (the LDS-DMA kernels' key): the 16 lanes of a fragment read (consecutive pixels, one chunk) spread over the banks; the
//     tap-shifted fragment bases are not multiples of 16 here, so some reads are 2-way conflicts — the kernel is bound by its
//     input stream, not by LDS.
#include "osd_common.h"
#include "conv_params.h"
#include "conv_epilogue.h"

namespace {

constexpr int PT_R = 8, PT_C = 32;                 // output pixels per workgroup: rows x columns
constexpr int PR = PT_R + 2, PC = PT_C + 2;        // input patch with its halo
constexpr int NPX = PR * PC;                       // 340 patch pixels
constexpr int SLAB = 64;                           // channels per slab = one 128-byte LDS row
constexpr int XB = NPX * 128;                      // patch slab bytes
constexpr int WB = 9 * 16 * 128;                   // weight slab bytes: [tap][co 16][64 ci]
constexpr int PSTAGE = XB + WB;                    // 61,952
constexpr int PRED_LDS = PSTAGE;                   // ONE stage: two workgroups per CU (the next slab waits in registers)
constexpr int NXI = (NPX * 8 + 255) / 256;         // 11 patch chunk loads per thread and slab
constexpr int NWI = (9 * 16 * 8 + 255) / 256;      // 5 weight chunk loads per thread and slab

__device__ __forceinline__ int pswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }      // the 128-byte-row key of conv_igemm_dma.hip

__global__ void __launch_bounds__(256) conv_pred_kernel(ConvKParams p) {
  typedef __bf16 T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const ConvView q = conv_select_view(p, t);       // t: tile index inside the selected level
  const int H = q.H, W = q.W, Cin = p.Cin;
  const int tc = (W + PT_C - 1) / PT_C, tr = (H + PT_R - 1) / PT_R;
  const int img = t / (tr * tc);
  const int rem = t - img * (tr * tc);
  const int y0 = (rem / tc) * PT_R, x0 = (rem % tc) * PT_C;
  typedef const __attribute__((address_space(1))) T* gptr;
  gptr xg = (gptr)(q.x) + (size_t)img * H * W * Cin;
  gptr wg = (gptr)(q.w);

  // ---- staging coordinates: patch chunk c = tid + 256 i -> (patch pixel, 16-byte chunk); element offset of its pixel in the image
  // (0 = a valid address for pixels outside the map: the loaded value is replaced by zeros) ----
  int x_off[NXI], x_dst[NXI];
  unsigned x_ok = 0u;
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int c = tid + 256 * i, px = c >> 3, ch = c & 7;
    const int py = px / PC, pxx = px - py * PC;
    const int yy = y0 - 1 + py, xx = x0 - 1 + pxx;
    const bool ok = px < NPX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    x_off[i] = ok ? (yy * W + xx) * Cin + ch * 8 : ch * 8;
    x_dst[i] = px < NPX ? pswz(px, ch) : -1;
    if (ok) x_ok |= 1u << i;
  }
  int w_off[NWI], w_dst[NWI];
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = c & 7;       // row = tap * 16 + co
    const int tap = row >> 4, co = row & 15;
    const bool in = row < 9 * 16;
    w_off[i] = in && co < p.w_rows ? (co * p.Ktot + tap * Cin + ch * 8) : -1;      // packed [w_rows][R][S][Cin]
    w_dst[i] = in ? XB + pswz(row, ch) : -1;
  }
  // (the kernarg pointers arrive as integers: explicit global address space, or hipcc emits flat loads; and the out-of-map zeroing is
  // an AND with a per-lane mask — a select between two 16-byte values made hipcc build the pair in scratch and index it)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) u32x4* gptr4;
  u32x4 xr[NXI], wr[NWI];
  auto load_slab = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NXI; ++i) xr[i] = *reinterpret_cast<gptr4>(xg + x_off[i] + kc);
#pragma unroll
    for (int i = 0; i < NWI; ++i) wr[i] = *reinterpret_cast<gptr4>(wg + (w_off[i] >= 0 ? w_off[i] + kc : 0));
  };
  auto store_slab = [&](int buf) {
    char* s = smem + buf * PSTAGE;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const unsigned m = ((x_ok >> i) & 1u) ? 0xffffffffu : 0u;
      const u32x4 v = xr[i] & m;
      if (x_dst[i] >= 0) *reinterpret_cast<u32x4*>(s + x_dst[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const unsigned m = w_off[i] >= 0 ? 0xffffffffu : 0u;
      const u32x4 v = wr[i] & m;
      if (w_dst[i] >= 0) *reinterpret_cast<u32x4*>(s + w_dst[i]) = v;
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fkq = lane >> 4;
  // fragment f of wave w: output row 2 w + (f >> 1), columns 16 (f & 1) + frow -> patch pixel (row + kr) * PC + col + ks for tap (kr, ks)
  int fpx[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) fpx[f] = (2 * wave + (f >> 1)) * PC + 16 * (f & 1) + frow;

  const int nslab = Cin / SLAB;
  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int sl = 0; sl < nslab; ++sl) {
    if (sl + 1 < nslab) load_slab((sl + 1) * SLAB);      // in flight while this slab is multiplied
    const char* xs = smem;
    const char* ws = xs + XB;
    // fragments one tap ahead of the MFMAs that consume them (two register sets): read in place, every MFMA waited for its own
    // ds_read_b128 — 72 exposed LDS latencies per slab and wave, 13 of the kernel's 19 us per tile
    bf16x8 wf[2][2], xf[2][2][4];
    auto read_tap = [&](int tap, int set) {
      const int kr = tap / 3, ks = tap % 3;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        wf[set][kb] = *reinterpret_cast<const bf16x8*>(ws + pswz(tap * 16 + frow, kb * 4 + fkq));
#pragma unroll
        for (int f = 0; f < 4; ++f)
          xf[set][kb][f] = *reinterpret_cast<const bf16x8*>(xs + pswz(fpx[f] + kr * PC + ks, kb * 4 + fkq));
      }
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) read_tap(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int f = 0; f < 4; ++f)
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap & 1][kb], xf[tap & 1][kb][f], acc[f], 0, 0, 0);
    }
    __syncthreads();                                     // every wave has read this slab
    if (sl + 1 < nslab) {
      store_slab(0);
      __syncthreads();
    }
  }

  // ---- epilogue: lanes 0..15 hold channels 0..3 of one pixel per fragment ----
  if (lane < 16) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(q.bias);
    const float escale = p.act == OSD_ACT_EXP_SCALE ? (q.scale_dev ? *q.scale_dev : p.act_scale) : 1.f;
    T* __restrict__ yg = reinterpret_cast<T*>(q.y) + (size_t)img * H * W * p.out_stride;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int yy = y0 + 2 * wave + (f >> 1), xx = x0 + 16 * (f & 1) + frow;
      if (yy < H && xx < W) {
        float v[4] = {acc[f][0] + bv[0], acc[f][1] + bv[1], acc[f][2] + bv[2], acc[f][3] + bv[3]};
        if (p.act == OSD_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == OSD_ACT_EXP_SCALE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = expf(v[e] * escale);
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4*>(yg + ((size_t)yy * W + xx) * p.out_stride) = o;
      }
    }
  }
}

}  // namespace

int osd_conv_pred_launch(const ConvKParams& pin, hipStream_t stream) {
  ConvKParams p = pin;
  if (p.R != 3 || p.S != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): 3x3 stride 1 pad 1 only");
  if (p.Cout > 4 || p.out_stride < 4 || p.out_stride % 4 != 0 || p.w_rows > 16)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): at most 4 output channels (stored as 4)");
  if (p.Cin % SLAB != 0 || p.sW != p.Cin) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): dense NHWC input with cin %% 64 == 0");
  if (p.res_mode != OSD_RES_NONE || p.mask != nullptr || p.relu_in || p.x2 != nullptr || p.gn_groups > 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): no residual / mask / relu_in / second source / statistics");
  long long tiles = 0;
  if (p.n_seg > 0) {
    for (int i = 0; i < p.n_seg; ++i) {
      ConvSeg& g = p.seg[i];
      if (g.mask != nullptr || g.res != nullptr) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): no residual / mask (segment %d)", i);
      if (g.Ho != g.H || g.Wo != g.W || g.sH != g.W * p.Cin || (long long)g.M * p.Cin >= 0x7fffffffLL)
        return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): segment %d is not a dense map under 2^31 elements", i);
      g.tile_begin = (int)tiles;
      tiles += (long long)(g.M / (g.H * g.W)) * cdiv(g.H, PT_R) * cdiv(g.W, PT_C);
    }
  } else {
    if (p.Ho != p.H || p.Wo != p.W || p.sH != p.W * p.Cin || (long long)p.M * p.Cin >= 0x7fffffffLL)
      return osd_fail(OSD_ERR_UNSUPPORTED, "conv(pred): not a dense map under 2^31 elements");
    tiles = (long long)(p.M / (p.H * p.W)) * cdiv(p.H, PT_R) * cdiv(p.W, PT_C);
  }
  if (tiles <= 0 || tiles > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv(pred): bad grid");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pred_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PRED_LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_pred_kernel, dim3((unsigned)tiles), dim3(256), PRED_LDS, stream, p);
  return osd_check_launch("conv_pred");
}
