// conv_pw — persistent pointwise (1x1, stride 1) convolution for the HBM-bound bottleneck convs (bf16, gfx950): algo id 41.
//     y[m][co] = act( mask( sum_k x[m][k] w[co][k] + bias[co] + res[m][co] ) )         m = (n, h, w), dense NHWC
// The bottleneck 1x1 convs (conv1 / conv3 of resnet.py:295-315 and their data gradients) move 60 - 240 MB per launch for 13 - 27
// GFLOP: at the HBM roofline they take 12 - 45 us, the one-tile-per-workgroup kernel (conv_igemm_dma.hip) needs 22 - 56.  Its
// cycle stamps (tools/cd_stamps.py) say why: a workgroup runs operand prologue -> K loop -> epilogue one after the other
// (1.4k + 5.9k + 4.7k cycles on 25600 x 1024 x 256), only the epilogue talks to HBM, and with two workgroups per CU plus the gap
// until the dispatcher refills a slot the memory system idles half of the time.  Here:
//   * PERSISTENT workgroups (2 per CU, 4 waves, 128 x 128 tile, 64 x 64 per wave): workgroup w walks the tiles
//     [w T / G, (w + 1) T / G) in (pixel tile, channel tile) order — consecutive tiles share their pixel operand (L2);
//   * ONE operand ring across tiles (2 x 32 KB stages of 64 K elements, `buffer_load ... lds`): the first stage of the next tile
//     is fetched while the last stage of this one computes and stays in flight through the epilogue — no operand prologue;
//   * the tile's WHOLE residual / mask / bias operand (32 + 32 + 8 VGPRs) is requested before the tile's second-to-last K stage:
//     it streams in under the MFMAs, and the epilogue (accumulators staged through the ring buffer the last stage has just
//     freed, 16-pixel passes) only adds, converts and stores;
//   * `s_waitcnt vmcnt` counts loads AND stores in issue order on gfx9, so every wait in the loop allows exactly the younger
//     instructions of that point to stay in flight (the epilogue's 8 stores, the 8 - 18 operand loads); bounds are the buffer
//     descriptors' (rows past M read zeros / drop their stores: no branches, a fixed instruction count per stage).
// Same K order and MFMA operand roles as conv_dma_kernel: results are bit-identical to it.
#include "osd_common.h"
#include "conv_params.h"
#include <type_traits>

namespace {

constexpr int PW_BM = 128, PW_BN = 128, PW_KB = 128, PW_BKE = 64;       // tile, bytes / elements of K per stage row
constexpr int PW_WM = 2, PW_WN = 2, PW_NWV = 4, PW_TM = 4, PW_TN = 4;
constexpr int PW_STAGE = (PW_BM + PW_BN) * PW_KB;                        // 32 KB
constexpr int PW_LDS = 2 * PW_STAGE;
constexpr int PW_PA = PW_BM / 8 / PW_NWV, PW_PB = PW_BN / 8 / PW_NWV, PW_LPS = PW_PA + PW_PB;      // DMA instructions per wave and stage
constexpr int PW_NPASS = 4, PW_ITER = 2, PW_CSW = 64 * 4 + 16;           // epilogue: 16-pixel passes, 2 x 16-byte chunks per lane, staging row stride
constexpr int PW_STORES = PW_NPASS * PW_ITER;
constexpr unsigned PW_OOB = 0x80000000u;

typedef unsigned int pw_u32x4 __attribute__((ext_vector_type(4)));
typedef int pw_i32x4 __attribute__((ext_vector_type(4)));

// s_waitcnt vmcnt(N) as the BUILTIN, not inline asm: the compiler's own wait insertion reads it (it then knows which of the
// loads / stores IT issued are complete; the LDS-DMA instructions of the inline asm are invisible to it, which only makes its
// model conservative).  With asm waits it guarded every reuse of a store's data registers in the next tile with `vmcnt(0)` —
// right behind the freshly issued stage fetch.  gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]
template <int N> __device__ __forceinline__ void pw_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ pw_i32x4 pw_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  pw_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

__device__ __forceinline__ void pw_dma16(pw_i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "s"(rsrc)
               : "memory");
}

__device__ __forceinline__ int pw_swz(int row) { return (row >> 1) & 7; }       // conv_dma's swizzle for 128-byte rows

template <bool HR, bool HM>
__global__ void __launch_bounds__(64 * PW_NWV, 2) conv_pw_kernel(ConvKParams p) {
  constexpr int RLOADS = 2 + (HR ? PW_STORES : 0) + (HM ? PW_STORES : 0);      // bias (2 x 16 B) + the tile's residual / mask chunks
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / PW_WN, wn = wave % PW_WN;
  const int T = p.tilesM * p.tilesN, G = gridDim.x;
  const int t_first = (int)((long long)blockIdx.x * T / G), t_last = (int)((long long)(blockIdx.x + 1) * T / G);
  if (t_first >= t_last) return;
  const int KT = p.KT, M = p.M, Cin = p.Cin, Ktot = p.Ktot;

  const pw_i32x4 xrs = pw_rsrc(p.x, (unsigned)M * (unsigned)Cin * 2u);
  const pw_i32x4 wrs = pw_rsrc(p.w, (unsigned)p.w_rows * (unsigned)Ktot * 2u);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((unsigned)M * (unsigned)p.out_stride * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, HR ? (int)((unsigned)M * (unsigned)p.res_stride * 2u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.mask), 0, HM ? (int)((unsigned)M * (unsigned)p.out_stride * 2u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.Cout * 4, 0x00020000);

  // ---- per-lane DMA offsets: instruction i of a wave covers tile rows (wave * 4 + i) * 8 .. + 7, lane -> (row lrow, chunk lpos);
  // the LDS destination is linear, so the swizzle is applied on the source side (chunk lpos of the row holds source chunk
  // lpos ^ swz(row)); swz depends on the instruction only through its parity
  const int lrow = lane >> 3, lpos = lane & 7;
  unsigned va[2], vb[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int ch = lpos ^ ((4 * par + (lrow >> 1)) & 7);
    va[par] = (unsigned)(lrow * Cin + ch * 8) * 2u;
    vb[par] = (unsigned)(lrow * Ktot + ch * 8) * 2u;
  }
  // fetch state: the stage the next issue_stage() fetches — tile f_t (pixel tile f_tm, channel tile f_tn), K stage f_kt
  int f_t = t_first, f_tm = t_first / p.tilesN, f_tn = t_first % p.tilesN, f_kt = 0;
  auto issue_stage = [&](int buf) {
    const unsigned st = lds0 + buf * PW_STAGE;
    const bool live = f_t < t_last;                     // past the last tile: zero fetches (the instruction count per stage stays fixed)
    const int m0 = f_tm * PW_BM, n0 = f_tn * PW_BN, k0 = f_kt * PW_BKE;
#pragma unroll
    for (int i = 0; i < PW_PA; ++i) {
      const int r0 = m0 + (wave * PW_PA + i) * 8;
      const unsigned off = va[i & 1] + (unsigned)(r0 * Cin + k0) * 2u;
      pw_dma16(xrs, (live && r0 + lrow < M) ? off : PW_OOB, st + (unsigned)((wave * PW_PA + i) * 1024));
    }
#pragma unroll
    for (int i = 0; i < PW_PB; ++i) {
      const int r0 = n0 + (wave * PW_PB + i) * 8;
      const unsigned off = vb[i & 1] + (unsigned)(r0 * Ktot + k0) * 2u;
      pw_dma16(wrs, (live && r0 + lrow < p.w_rows) ? off : PW_OOB, st + (unsigned)(PW_BM * PW_KB + (wave * PW_PB + i) * 1024));
    }
    if (++f_kt == KT) {
      f_kt = 0;
      ++f_t;
      if (++f_tn == p.tilesN) { f_tn = 0; ++f_tm; }
    }
  };

  f32x4 acc[PW_TN][PW_TM];
  const int frow = lane & 15, fkq = lane >> 4;
  auto mfma_stage = [&](int buf) {
    const char* xs = smem + buf * PW_STAGE;
    const char* ws = xs + PW_BM * PW_KB;
#pragma unroll
    for (int kb = 0; kb < PW_KB / 64; ++kb) {
      uint4 wf[PW_TN], xf[PW_TM];
#pragma unroll
      for (int i = 0; i < PW_TN; ++i) {
        const int row = (wn * PW_TN + i) * 16 + frow;
        wf[i] = *reinterpret_cast<const uint4*>(ws + row * PW_KB + (((kb * 4 + fkq) ^ pw_swz(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < PW_TM; ++j) {
        const int row = (wm * PW_TM + j) * 16 + frow;
        xf[j] = *reinterpret_cast<const uint4*>(xs + row * PW_KB + (((kb * 4 + fkq) ^ pw_swz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < PW_TN; ++i)
#pragma unroll
        for (int j = 0; j < PW_TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf[i]), *reinterpret_cast<const bf16x8*>(&xf[j]),
                                                              acc[i][j], 0, 0, 0);
    }
  };

  // ---- the tile's epilogue operands: bias of my 8 channels, residual / mask chunks of my 8 (row, chunk) slots ----
  const int cc = lane & 7, erow = lane >> 3;            // epilogue: lane -> (row within an 8-row group, 8-channel chunk)
  pw_u32x4 rr[HR ? PW_STORES : 1], mm[HM ? PW_STORES : 1];
  f32x4 bias_lo, bias_hi;
  int c_m0 = 0, c_n0 = 0;                               // the tile being computed
  auto issue_operands = [&]() {
    const int c = c_n0 + wn * 64 + cc * 8;
    bias_lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, c * 4, 0, 0));
    bias_hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, c * 4 + 16, 0, 0));
#pragma unroll
    for (int s = 0; s < PW_STORES; ++s) {
      const int mrow = c_m0 + wm * 64 + (s >> 1) * 16 + (s & 1) * 8 + erow;
      if constexpr (HR) rr[s] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (mrow * p.res_stride + c) * 2, 0, 0);
      if constexpr (HM) mm[s] = __builtin_amdgcn_raw_buffer_load_b128(mrs, (mrow * p.out_stride + c) * 2, 0, 0);
    }
  };

  // one K stage: wait until it has landed (ALLOW younger instructions may stay in flight), publish it, fetch the stage after
  // it into the other buffer, optionally request the epilogue operands, then the MFMAs
#ifdef OSD_PW_STAMPS      // diagnostic build: per-wave cycle sums into the buffer passed as act_scale_dev (tools/pw_stamps.py)
  unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_epi = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#define PW_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define PW_T(var)
#endif
  auto k_stage = [&](int buf, auto allow_tag, auto ops_tag) {
    PW_T(ta);
    pw_wait_vmcnt<decltype(allow_tag)::value>();
    PW_T(tb);
    __builtin_amdgcn_s_barrier();
    PW_T(tc);
    issue_stage(buf ^ 1);
    if constexpr (decltype(ops_tag)::value) issue_operands();
    PW_T(td);
    mfma_stage(buf);
#ifdef OSD_PW_STAMPS
    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[3][3]));
    PW_T(te);
    st_wait += tb - ta; st_bar += tc - tb; st_issue += td - tc; st_mfma += te - td;
#endif
  };
  using A0 = std::integral_constant<int, 0>;
  using AR = std::integral_constant<int, RLOADS>;

  issue_stage(0);
  pw_wait_vmcnt<0>();           // once per workgroup: afterwards a tile's first stage was fetched under the previous tile's epilogue
  int cur = 0;
  int c_tm = t_first / p.tilesN, c_tn = t_first % p.tilesN;
  for (int t = t_first; t < t_last; ++t) {
    c_m0 = c_tm * PW_BM; c_n0 = c_tn * PW_BN;
#pragma unroll
    for (int i = 0; i < PW_TN; ++i)
#pragma unroll
      for (int j = 0; j < PW_TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // in flight at the first stage's wait: [this stage][the previous tile's stores] — all of it is waited for (the stage was
    // fetched an epilogue ago, stores are acknowledged within a few hundred cycles; and the compiler then knows the stores' data
    // registers are free); at the last stage's wait: [this stage][operands]: the operands stay in flight
    if (KT == 1) {
      k_stage(cur, A0(), std::true_type()); cur ^= 1;
    } else if (KT == 2) {
      k_stage(cur, A0(), std::true_type()); cur ^= 1;
      k_stage(cur, AR(), std::false_type()); cur ^= 1;
    } else {
      k_stage(cur, A0(), std::false_type()); cur ^= 1;
      for (int kt = 1; kt < KT - 2; ++kt) { k_stage(cur, A0(), std::false_type()); cur ^= 1; }
      k_stage(cur, A0(), std::true_type()); cur ^= 1;
      k_stage(cur, AR(), std::false_type()); cur ^= 1;
    }
    // ---- epilogue: the buffer of the last stage (cur ^ 1) is the staging area once every wave has read its last fragments;
    // the other one holds the next tile's first stage, in flight.  Per wave: a private 16 x 272-byte region
    PW_T(tf);
    __builtin_amdgcn_s_barrier();
    char* stage = smem + (cur ^ 1) * PW_STAGE + wave * (16 * PW_CSW);
    const int c = c_n0 + wn * 64 + cc * 8;
    const float bv[8] = {bias_lo[0], bias_lo[1], bias_lo[2], bias_lo[3], bias_hi[0], bias_hi[1], bias_hi[2], bias_hi[3]};
#pragma unroll
    for (int ps = 0; ps < PW_NPASS; ++ps) {
#pragma unroll
      for (int i = 0; i < PW_TN; ++i)
        *reinterpret_cast<f32x4*>(stage + (lane & 15) * PW_CSW + (i * 16 + (lane >> 4) * 4) * 4) = acc[i][ps];
#pragma unroll
      for (int it = 0; it < PW_ITER; ++it) {
        const int s = ps * PW_ITER + it;
        const char* src = stage + (it * 8 + erow) * PW_CSW + cc * 32;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 16);
        float v[8] = {a0[0] + bv[0], a0[1] + bv[1], a0[2] + bv[2], a0[3] + bv[3], a1[0] + bv[4], a1[1] + bv[5], a1[2] + bv[6], a1[3] + bv[7]};
        if constexpr (HR) {
          const bf16x8 r = __builtin_bit_cast(bf16x8, rr[s]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
        }
        if constexpr (HM) {
          const bf16x8 mk = __builtin_bit_cast(bf16x8, mm[s]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
        }
        if (p.act == OSD_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        const int mrow = c_m0 + wm * 64 + ps * 16 + it * 8 + erow;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pw_u32x4, o), yrs, (mrow * p.out_stride + c) * 2, 0, 0);
      }
    }
    if (++c_tn == p.tilesN) { c_tn = 0; ++c_tm; }
#ifdef OSD_PW_STAMPS
    PW_T(tg);
    st_epi += tg - tf;
#endif
  }
#ifdef OSD_PW_STAMPS
  if (lane == 0 && p.act_scale_dev != nullptr) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.act_scale_dev)) + ((size_t)blockIdx.x * PW_NWV + wave) * 8;
    o[0] = st_begin; o[1] = __builtin_amdgcn_s_memtime() - st_begin; o[2] = st_wait; o[3] = st_bar; o[4] = st_issue; o[5] = st_mfma; o[6] = st_epi;
    o[7] = (unsigned long long)(t_last - t_first);
  }
#endif
  pw_wait_vmcnt<0>();           // the zero fetches past the last tile: nothing may land in this LDS once the workgroup has left
}

}  // namespace

int osd_conv_pw_launch(const ConvKParams& pin, hipStream_t stream) {
  ConvKParams p = pin;
  if (p.R != 1 || p.S != 1 || p.sh != 1 || p.sw != 1 || p.ph != 0 || p.pw != 0 || p.x2 != nullptr || p.relu_in || p.n_seg > 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_pw: a plain 1x1 / stride 1 conv of one tensor");
  if (p.sW != p.Cin || p.sH != p.W * p.Cin || p.sN != p.H * p.W * p.Cin)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_pw: the input must be dense NHWC");
  if (p.Cin % PW_BKE || p.Cout % PW_BN || p.w_rows < p.Cout || p.Ktot != p.Cin || p.out_stride % 8 || p.Cout % 8)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_pw: cin in 64s, cout in 128s, 16-byte output rows");
  if (p.res_mode != OSD_RES_NONE && (p.res_mode != OSD_RES_SAME || p.res_stride % 8))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_pw: residual of the output's own size only");
  if (p.act != OSD_ACT_NONE && p.act != OSD_ACT_RELU) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_pw: activation none / relu");
  const long long lim = 0x7fffffffLL;
  if ((long long)p.M * p.Cin * 2 >= lim || (long long)p.M * p.out_stride * 2 >= lim || (long long)p.w_rows * p.Ktot * 2 >= lim ||
      (p.res_mode != OSD_RES_NONE && (long long)p.M * p.res_stride * 2 >= lim))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_pw: tensors below 2 GiB (32-bit buffer offsets)");
  p.tilesM = cdiv(p.M, PW_BM);
  p.tilesN = p.Cout / PW_BN;
  p.KT = p.Cin / PW_BKE;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  }
  const long long tiles = (long long)p.tilesM * p.tilesN;
  if (tiles > lim) return osd_fail(OSD_ERR_INVALID_ARG, "conv_pw: bad grid");
  const unsigned grid = (unsigned)(tiles < 2LL * cus ? tiles : 2LL * cus);
  const bool hr = p.res_mode == OSD_RES_SAME, hm = p.mask != nullptr;
#define OSD_PW_LAUNCH(HR, HM)                                                                                              \
  do {                                                                                                                     \
    auto kern = conv_pw_kernel<HR, HM>;                                                                                    \
    static bool attr = false;                                                                                              \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS); attr = true; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * PW_NWV), PW_LDS, stream, p);                                            \
  } while (0)
  if (hr && hm) OSD_PW_LAUNCH(true, true);
  else if (hr) OSD_PW_LAUNCH(true, false);
  else if (hm) OSD_PW_LAUNCH(false, true);
  else OSD_PW_LAUNCH(false, false);
#undef OSD_PW_LAUNCH
  return osd_check_launch("conv_pw");
}
