// conv_igemm_xr — 3x3 / stride 1 / pad 1 implicit-GEMM convolution (bf16, 256 x 256 tile, 8 waves) that moves each input
// pixel row L2 -> LDS ONCE per (filter row, channel slab) instead of once per tap.
//
// Why: the LDS-DMA kernel (conv_igemm_dma.hip) is bound by the number of distinct 128-byte lines it moves L2 -> LDS
// (DESIGN.md 4.1b': ~15 B/clk/CU beside the MFMA / ds_read load; the same instruction stream reading one cached line
// runs 25-40 % faster).  Half of those lines are the pixel operand, and for a 3x3 filter the three taps of one filter row
// read the SAME pixels shifted by one: tap s of output pixel m is input pixel m + s - 1 of the same image line.
//
// How: the K loop runs (filter row r, 64-channel slab c, tap s) with s innermost.  Per (r, c) group the 256 pixel rows
// of the tile are fetched once into a PADDED LDS image: every image line (W pixels, W | 256, so tiles start at x = 0)
// is followed by 16 zero rows (and the first line preceded by 16).  Tap s then reads LDS row  base + x + s - 1:
// x = 0 / s = 0 and x = W-1 / s = 2 land in a zero row — the horizontal zero padding costs nothing, no halo pixels are
// needed, and no per-lane select touches the fragments.  The pad rows are written once per workgroup.  Vertical
// padding / image ends / the M tail read the zero page at DMA time as in conv_igemm_dma.hip.  The weight operand keeps
// the 2-deep ring of 32 KB stages (one per tap); the pixel operand has its own 2-deep ring of groups.
// DMA instructions per wave per 3 stages: 4 (pixels) + 12 (weights) instead of 24.
//
// LDS: 2 x 336 rows x 128 B (pixels, W >= 64) + 2 x 32 KB (weights) = 148 KB.  XOR swizzle, swapped MFMA operands,
// one barrier per stage and the epilogue (conv_epilogue.h) are those of the DMA kernel; the accumulation order over K
// differs ((r, c, s) instead of (r, s, c)), so results equal the DMA kernel's to fp32 rounding, not bit for bit.
#include "osd_common.h"
#include "conv_params.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(256))) unsigned g_xr_zero_page[64];

template <int N> __device__ __forceinline__ void xr_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void xr_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// XOR swizzle key of a padded pixel-image row (period 16).  The DMA kernel's key ((row >> 1) & 7) is conflict-free only for
// fragments that start on a multiple of 16; the taps s = 0 / 2 read 16 consecutive rows starting at -1 / +1.  ds_read_b128
// serves lanes in groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (16 lanes per LDS cycle: rows f..f+3, f+12..f+15 of
// one k chunk with rows f+4..f+11 of the next chunk), so with a = key(even rows), b = key(odd rows) the three shifts need
// {a0,a1,a6,a7, a2^1..a5^1}, the same with a2 <-> a6 exchanged, and likewise for b with b1 <-> b5: solved by a2 = a6,
// b1 = b5 -> a = 2,3,0,4,7,6,0,4; b = 2,0,4,7,6,0,3,4 (checked exhaustively by tools/lds_swizzle_check.py --xr).
__device__ __forceinline__ int xr_key(int row) { return (int)((0x4430066774400322ull >> ((row & 15) * 4)) & 7); }

constexpr int XR_BM = 256, XR_BN = 256, XR_KB = 128 /* bytes per row per slab */, XR_BKE = 64, XR_EPC = 8;
constexpr int XR_WM = 2, XR_WN = 4, XR_TM = 8, XR_TN = 4;
constexpr int XR_AROWS = 336;                        // 256 + 16 * (256 / 64 + 1)
constexpr int XR_ABYTES = XR_AROWS * XR_KB;          // 43,008
constexpr int XR_BBYTES = XR_BN * XR_KB;             // 32,768
constexpr int XR_LDS = 2 * XR_ABYTES + 2 * XR_BBYTES;

__global__ void __launch_bounds__(512) conv_xr_kernel(ConvKParams p) {
  typedef __bf16 T;
  constexpr int TM = XR_TM, TN = XR_TN, KB = XR_KB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / XR_WN, wn = wave % XR_WN;

  int t;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = t % p.tilesN;
  int tile_m = t / p.tilesN;
  const ConvView q = conv_select_view(p, tile_m);
  const int q_H = q.H, q_W = q.W, q_M = q.M, q_sN = q.sN, q_sH = q.sH, q_HoWo = q.HoWo;
  const int m0 = tile_m * XR_BM, n0 = tile_n * XR_BN;
  const int logw = __builtin_ctz((unsigned)q_W);       // W is a power of two in {64, 128, 256} (checked by the launcher)

  const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(q.w);
  const T* zero = reinterpret_cast<const T*>(g_xr_zero_page) + (lane & 15) * XR_EPC;

  // ---- zero both pixel images once: the pad rows stay zero for the whole K loop ----
  {
    uint4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * XR_ABYTES / 16; i += 512) *reinterpret_cast<uint4*>(smem + i * 16) = z;
  }

  // ---- per-lane DMA sources.  One wave-instruction = 8 rows x 128 B; wave w owns pixel instructions 4w .. 4w+3 and
  // weight instructions 4w .. 4w+3 (32 each per stage / group) ----
  const int lrow = lane >> 3, lpos = lane & 7;
  const T* a_base[4];
  int a_ho[4];
  unsigned a_dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;          // tile row of this lane
    const int m = m0 + row;
    const int R = 16 + row + 16 * (row >> logw);        // padded LDS row
    a_dst[i] = (unsigned)((16 + (wave * 4 + i) * 8 + 16 * (((wave * 4 + i) * 8) >> logw)) * KB);   // wave-uniform
    a_base[i] = zero;
    a_ho[i] = -0x40000000;
    if (m < q_M) {
      const int n_img = m / q_HoWo;
      const int rem = m - n_img * q_HoWo;
      const int ho = rem >> logw;
      const int wo = rem & (q_W - 1);
      a_base[i] = xg + (size_t)n_img * q_sN + wo * p.sW + ((lpos ^ xr_key(R)) * XR_EPC);
      a_ho[i] = ho - 1;                                 // input line of filter row 0
    }
  }
  const T* b_row[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    b_ok[i] = n0 + row < p.w_rows;
    b_row[i] = b_ok[i] ? wg + (size_t)(n0 + row) * p.Ktot + ((lpos ^ ((row >> 1) & 7)) * XR_EPC) : zero;
  }

  const unsigned a_lds[2] = {lds0, lds0 + (unsigned)XR_ABYTES};
  const unsigned b_lds[2] = {lds0 + 2u * XR_ABYTES, lds0 + 2u * XR_ABYTES + (unsigned)XR_BBYTES};

  // pixel instruction i of group (kr, kc) into pixel image `buf`
  auto issue_a = [&](int buf, int i, int kr, int kc) {
    const int hi = a_ho[i] + kr;
    const bool ok = (unsigned)hi < (unsigned)q_H;
    const T* src = ok ? a_base[i] + (hi * q_sH + kc) : zero;
    xr_dma16(src, a_lds[buf] + a_dst[i]);
  };
  // weight instruction i of stage (kr, ks, kc) into weight stage `buf`
  auto issue_b = [&](int buf, int i, int koff) {
    const T* src = b_ok[i] ? b_row[i] + koff : zero;
    xr_dma16(src, b_lds[buf] + (unsigned)((wave * 4 + i) * 1024));
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fkq = lane >> 4;
  // weight fragment offsets (as in the DMA kernel) and the per-lane part of the pixel fragment offsets for the 3 taps
  int w_off[TN][2];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = (wn * TN + i) * 16 + frow;
      w_off[i][kb] = row * KB + (((kb * 4 + fkq) ^ ((row >> 1) & 7)) << 4);
    }
  int x_lane[3][2];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int e = frow + s - 1;                       // -1 .. 16: row relative to the fragment's first padded row
      x_lane[s][kb] = e * KB + (((kb * 4 + fkq) ^ xr_key(e)) << 4);          // fragment bases are multiples of 16 rows
    }
  int x_frag[TM];                                       // wave-uniform: first padded row of fragment j (a multiple of 16)
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int r0 = (wm * TM + j) * 16;
    x_frag[j] = (16 + r0 + 16 * (r0 >> logw)) * KB;
  }

  const int nslab = p.Cin / XR_BKE;
  const int G = 3 * nslab;                              // groups = (filter row, slab)
  const int KT = 3 * G;

  // MFMAs of one stage: tap `s` of pixel image `ab`, weight stage `bb`.  The DMA of the next weight stage (4 per wave)
  // and a share of the next pixel group (2 per wave in the s = 0 and s = 1 stages) are spread over the MFMA row groups.
  auto compute_stage = [&](int ab, int bb, auto s_tag, auto fb_tag, int koff_next, auto fa_tag, int nkr, int nkc) {
    constexpr int s = decltype(s_tag)::value;
    constexpr bool fetch_b = decltype(fb_tag)::value, fetch_a = decltype(fa_tag)::value;   // compile time: no branches between the MFMAs
    constexpr int a_first = 2 * s;                      // pixel instructions 0,1 in the s = 0 stage, 2,3 in the s = 1 stage
    const char* xs = smem + ab * XR_ABYTES;
    const char* ws = smem + 2 * XR_ABYTES + bb * XR_BBYTES;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      uint4 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const uint4*>(ws + w_off[i][kb]);
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const uint4*>(xs + x_frag[j] + x_lane[s][kb]);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf[i]),
                                                              *reinterpret_cast<const bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
        // slots 0..3 of the first k half: the 4 weight instructions; slots 0..1 of the second: 2 pixel instructions
        if constexpr (fetch_b) {
          if (kb == 0) issue_b(bb ^ 1, i, koff_next);
        }
        if constexpr (fetch_a && s < 2) {
          if (kb == 1 && i < 2) issue_a(ab ^ 1, a_first + i, nkr, nkc);
        }
      }
    }
  };

  // ---- prologue: pixel group 0 and weight stage 0 ----
  __syncthreads();                                      // the zero fill is complete before any DMA lands on real rows
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_a(0, i, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_b(0, i, 0);

  int ab = 0, bb = 0;
  int kr = 0, kc = 0;                                   // current group: filter row, channel offset (elements)
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  for (int g = 0; g + 1 < G; ++g) {                     // every group but the last: all prefetches on
    int nkr = kr, nkc = kc + XR_BKE;                    // next group
    if (nkc >= p.Cin) { nkc = 0; ++nkr; }
    const int kbase = kr * 3 * p.Cin + kc;              // K offset of tap 0 of this group; tap s adds s * Cin
    xr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(ab, bb, T0(), std::true_type(), kbase + p.Cin, std::true_type(), nkr, nkc);
    bb ^= 1;
    xr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(ab, bb, T1(), std::true_type(), kbase + 2 * p.Cin, std::true_type(), nkr, nkc);
    bb ^= 1;
    xr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(ab, bb, T2(), std::true_type(), nkr * 3 * p.Cin + nkc, std::false_type(), nkr, nkc);   // tap 0 of the next group
    bb ^= 1;
    ab ^= 1;
    kr = nkr;
    kc = nkc;
  }
  {                                                     // last group: only the two remaining weight stages are fetched
    const int kbase = kr * 3 * p.Cin + kc;
    xr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(ab, bb, T0(), std::true_type(), kbase + p.Cin, std::false_type(), 0, 0);
    bb ^= 1;
    xr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(ab, bb, T1(), std::true_type(), kbase + 2 * p.Cin, std::false_type(), 0, 0);
    bb ^= 1;
    xr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(ab, bb, T2(), std::false_type(), 0, std::false_type(), 0, 0);
  }
  (void)KT;

  xr_wait_vmcnt<0>();
  __syncthreads();
  conv_epilogue<T, TM, TN>(acc, p, q, smem, wave, wm, wn, lane, m0, n0);
}

}  // namespace

int osd_conv_xr_launch(const ConvKParams& pin, hipStream_t stream) {
  ConvKParams p = pin;
  if (p.R != 3 || p.S != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(xr): 3x3 stride 1 pad 1 only");
  if (p.Cin % XR_BKE != 0 || p.sW != p.Cin) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(xr): dense NHWC input with cin %% 64 == 0");
  if (p.relu_in) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(xr): relu_in prologue not supported");
  auto width_ok = [](int w, int wo, int h, int ho) { return (w == 64 || w == 128 || w == 256) && wo == w && ho == h; };
  p.tilesM = cdiv(p.M, XR_BM);
  if (p.n_seg > 0) {
    p.tilesM = 0;
    for (int i = 0; i < p.n_seg; ++i) {
      if (!width_ok(p.seg[i].W, p.seg[i].Wo, p.seg[i].H, p.seg[i].Ho) || p.seg[i].sH != p.seg[i].W * p.Cin)
        return osd_fail(OSD_ERR_UNSUPPORTED, "conv(xr): segment %d width %d (needs 64, 128 or 256)", i, p.seg[i].W);
      p.seg[i].tile_begin = p.tilesM;
      p.tilesM += cdiv(p.seg[i].M, XR_BM);
    }
  } else if (!width_ok(p.W, p.Wo, p.H, p.Ho) || p.sH != p.W * p.Cin) {
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(xr): width %d (needs 64, 128 or 256)", p.W);
  }
  p.tilesN = cdiv(p.Cout, XR_BN);
  p.KT = p.Ktot / XR_BKE;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, XR_LDS);
    attr_done = true;
  }
  const long long nblocks = (long long)p.tilesM * p.tilesN;
  if (nblocks <= 0 || nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv(xr): bad grid");
  hipLaunchKernelGGL(conv_xr_kernel, dim3((unsigned)nblocks), dim3(512), XR_LDS, stream, p);
  return osd_check_launch("conv_igemm_xr");
}
