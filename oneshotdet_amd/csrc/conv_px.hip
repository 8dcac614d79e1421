// conv_px — pixel-stationary pointwise (1x1, stride 1) convolution for the EXPANDING bottleneck convs (bf16, gfx950): algo id 49.
//     y[m][co] = act( mask( sum_k x[m][k] w[co][k] + bias[co] + res[m][co] ) )       m = (n, h, w), dense NHWC, K = Cin <= 256
// conv3 of a bottleneck (resnet.py:295-315: K = 64 / 128 / 256 -> N = 4 K, + identity + ReLU) and the data gradient of conv1 (the
// same shape, + the skip path's gradient + the ReLU mask) move 118 - 340 MB per launch for 13 GFLOP: HBM work.  The tile kernels
// (conv_igemm_dma.hip, conv_pw.hip) take 35 us for the 118 MB of layer3 (3.3 TB/s) whatever the tile, because every output tile
// streams BOTH operands through the CU's texture path again: 128 KB of operands for 32 KB of output on the 128 x 128 tile, and that
// path accepts one 1 KB wave-instruction per 25 - 30 cycles.  With K <= 256 a wave can KEEP its pixels instead:
//   * a wave owns 32 pixels and holds their K values as MFMA B fragments in registers (2 x K / 32 x 4 VGPRs = 64 at K = 256),
//     loaded once per 128-pixel block straight from global memory in fragment shape (the pixel operand is the SMALL tensor here:
//     13 MB against 105 MB of residual + output);
//   * a workgroup (4 waves, 128 pixels) walks work units (pixel block, 64-channel chunk) — [w U / G, (w + 1) U / G) of the
//     launch's U units, so the chip's 2 x CUs workgroups carry the same load to one unit — and streams only the WEIGHT chunk
//     (64 rows x K: 32 KB at K = 256) through a two-slot LDS ring by LDS-DMA, a unit ahead;
//   * the unit's residual / mask / bias operands are requested a unit ahead as well (two register sets), so every memory
//     latency has a whole unit (MFMAs + epilogue of the unit before) of cover; one barrier per unit;
//   * bytes through the texture path per 128 x 64 outputs: 32 KB weights + 16 residual + 16 output = 64 KB against 96 KB.
// Same MFMA operand roles and K order as conv_dma_kernel: the outputs are bit-identical to it (tested).
#include "osd_common.h"
#include "conv_params.h"
#include <type_traits>

namespace {

constexpr int PX_BP = 128;                                               // pixels per block: 4 waves x 32 or 8 waves x 16
constexpr int PX_CH = 64;                                                // channels per work unit
constexpr int PX_CSW = PX_CH * 4;                                        // epilogue staging row stride (bytes): no pad, 16-byte chunks XOR-swizzled by the row
constexpr int PX_STGS = 16 * 1024;                                       // staging, all waves: 4 x 16 rows or 8 x 8 rows (K = 256: 2 x 32 KB + 16 KB = half of the CU's LDS)
constexpr unsigned PX_OOB = 0x80000000u;

typedef unsigned int px_u32x4 __attribute__((ext_vector_type(4)));
typedef int px_i32x4 __attribute__((ext_vector_type(4)));

// EVERY vector-memory instruction of this kernel is inline asm — LDS-DMA, register loads and stores alike — and so is every
// `s_waitcnt vmcnt`.  A first version used the buffer-load / store builtins for the register operands: hipcc's wait insertion then
// counts only the operations IT issued, so with LDS-DMA instructions (invisible to it) interleaved among them its counted waits
// come out too small — `vmcnt(6)` right behind the weight DMA of the next unit, i.e. a full L2 round trip plus the previous unit's
// store acknowledgements in EVERY unit (9.5k cycles per unit for 1k cycles of MFMAs; measured 31.5 us against the tile kernel's
// 30.4).  With nothing visible to it hipcc inserts no vmcnt at all and the counts below are exact.  A load's destination is tied
// to its wait by "+v" operands (the consumers cannot be scheduled above the wait, CDNA guide 5.7 item 1 form ii); a store ends
// with `s_nop 1` so that its data registers may be overwritten at once.
template <int N> __device__ __forceinline__ void px_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void px_load16(px_u32x4& dst, px_i32x4 rsrc, int voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void px_store16(px_u32x4 v, px_i32x4 rsrc, int voff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
// "these registers are written by loads that the wait before this statement covers"
__device__ __forceinline__ void px_tie(px_u32x4& a, px_u32x4& b, px_u32x4& c, px_u32x4& d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ void px_tie(px_u32x4& a, px_u32x4& b) { asm volatile("" : "+v"(a), "+v"(b)::"memory"); }

__device__ __forceinline__ px_i32x4 px_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  px_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

__device__ __forceinline__ void px_dma16(px_i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "s"(rsrc)
               : "memory");
}

// KS = K / 32 MFMA k steps (2, 4, 8); HR / HM: residual / mask operand; PXW: pixels per wave (32: 4 waves, 16: 8 waves per workgroup)
template <int KS, bool HR, bool HM, int PXW>
__global__ void __launch_bounds__(64 * (PX_BP / PXW), PXW == 16 ? 4 : 2) conv_px_kernel(ConvKParams p) {      // two workgroups per CU either way
  constexpr int PX_NWV = PX_BP / PXW, PX_PXW = PXW, NF = PXW / 16;      // waves, pixels per wave, pixel fragments per wave
  constexpr int PX_STG = PX_STGS / PX_NWV;                              // one wave's staging region: 16 or 8 rows
  constexpr int K = KS * 32;
  constexpr int NKST = (K + 63) / 64;                   // 128-byte K stages of the weight image (K = 64: one)
  constexpr int SLOT = NKST * PX_CH * 128;              // one weight chunk in LDS: [K stage][64 rows][128 B]
  constexpr int NDMA = NKST * PX_CH / 8 / PX_NWV;       // weight DMA instructions per wave and unit (8 rows x 128 B each)
  constexpr int NRS = PXW / 8;                          // 8-row x 64-channel slabs of a wave's unit: residual / mask loads, stores
  constexpr int NOPS = 2 + (HR ? NRS : 0) + (HM ? NRS : 0); // operand loads per wave and unit: bias (2 x 16 B), residual / mask chunks
  constexpr int NXL = NF * KS;                          // pixel-fragment loads per wave and block
  constexpr int NST = NRS;                              // stores per wave and unit
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunks = p.Cout / PX_CH;
  const int U = p.tilesM * chunks, G = gridDim.x;
  const int u_first = (int)((long long)blockIdx.x * U / G), u_last = (int)((long long)(blockIdx.x + 1) * U / G);
  if (u_first >= u_last) return;
  const int M = p.M;

  const px_i32x4 xrs = px_rsrc(p.x, (unsigned)M * (unsigned)K * 2u);
  const px_i32x4 wrs = px_rsrc(p.w, (unsigned)p.w_rows * (unsigned)p.Ktot * 2u);
  const px_i32x4 yrs = px_rsrc(p.y, (unsigned)M * (unsigned)p.out_stride * 2u);
  const px_i32x4 rrs = px_rsrc(HR ? p.res : p.x, HR ? (unsigned)M * (unsigned)p.res_stride * 2u : 0u);
  const px_i32x4 mrs = px_rsrc(HM ? p.mask : p.x, HM ? (unsigned)M * (unsigned)p.out_stride * 2u : 0u);
  const px_i32x4 brs = px_rsrc(p.bias, (unsigned)p.Cout * 4u);

  // ---- weight chunk of unit u -> slot: instruction i of a wave covers rows (wave * NDMA + i) * 8 .. + 7 of the [K stage][64 rows]
  // image (row index r = stage * 64 + channel); lane -> (row lrow, 16-byte chunk lpos), swizzled on the source side like conv_dma
  const int lrow = lane >> 3, lpos = lane & 7;
  auto issue_weights = [&](int u, int slot) {
    const bool live = u < u_last;
    const int n0 = (u % chunks) * PX_CH;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int r = (wave * NDMA + i) * 8 + lrow;       // row of the slot image
      const int st = r >> 6, ch = r & 63;               // K stage, channel of the chunk
      const int kc = (lpos ^ ((ch >> 1) & 7)) * 8 + st * 64;
      const bool ok = live && kc < K;                   // (K = 64 with 128-byte rows never happens: K % 64 == 0; K = 32 * odd is refused)
      px_dma16(wrs, ok ? (unsigned)((n0 + ch) * p.Ktot + kc) * 2u : PX_OOB, lds0 + (unsigned)(slot * SLOT + (wave * NDMA + i) * 1024));
    }
  };

  // ---- pixel fragments of a block: fragment j, k step ks <- x[m0 + wave * 32 + j * 16 + (lane & 15)][ks * 32 + (lane >> 4) * 8 .. + 7]
  px_u32x4 xf[NF][KS];
  const int frow = lane & 15, fkq = lane >> 4;
  auto load_pixels = [&](int blk) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int m = blk * PX_BP + wave * PX_PXW + j * 16 + frow;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        px_load16(xf[j][ks], xrs, (m * K + ks * 32 + fkq * 8) * 2);
    }
  };

  // ---- epilogue operands of a unit: bias of my 8 channels, residual / mask chunks of my 4 (row, chunk) slots
  const int cc = lane & 7, erow = lane >> 3;
  struct Ops { px_u32x4 rr[HR ? NRS : 1], mm[HM ? NRS : 1]; px_u32x4 blo, bhi; };
  auto issue_operands = [&](int u, Ops& o) {
    const int c = (u % chunks) * PX_CH + cc * 8;
    const int m0 = (u / chunks) * PX_BP + wave * PX_PXW;
    px_load16(o.blo, brs, c * 4);
    px_load16(o.bhi, brs, c * 4 + 16);
#pragma unroll
    for (int s = 0; s < NRS; ++s) {
      const int mrow = m0 + s * 8 + erow;
      if constexpr (HR) px_load16(o.rr[s], rrs, (mrow * p.res_stride + c) * 2);
      if constexpr (HM) px_load16(o.mm[s], mrs, (mrow * p.out_stride + c) * 2);
    }
  };

  f32x4 acc[4][NF];
  auto mfma_unit = [&](int slot) {
    const char* ws = smem + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + frow;
        wf[i] = *reinterpret_cast<const uint4*>(ws + (ks >> 1) * (PX_CH * 128) + row * 128 + ((((ks & 1) * 4 + fkq) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf[i]), __builtin_bit_cast(bf16x8, xf[j][ks]),
                                                              acc[i][j], 0, 0, 0);
    }
  };

  char* stage = smem + 2 * SLOT + wave * PX_STG;        // this wave's private staging region
  auto epilogue = [&](int u, Ops& o) {
    const int c = (u % chunks) * PX_CH + cc * 8;
    const int m0 = (u / chunks) * PX_BP + wave * PX_PXW;
    // the caller has waited for this unit's operand loads: tie the registers to that wait
    px_tie(o.blo, o.bhi);
    if constexpr (HR) { if constexpr (NRS == 4) px_tie(o.rr[0], o.rr[1], o.rr[2], o.rr[3]); else px_tie(o.rr[0], o.rr[1]); }
    if constexpr (HM) { if constexpr (NRS == 4) px_tie(o.mm[0], o.mm[1], o.mm[2], o.mm[3]); else px_tie(o.mm[0], o.mm[1]); }
    const f32x4 blo = __builtin_bit_cast(f32x4, o.blo), bhi = __builtin_bit_cast(f32x4, o.bhi);
    const float bv[8] = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
#pragma unroll
    for (int ps = 0; ps < NF; ++ps) {
      // staging image [16 pixels][64 channels] fp32 with 256-byte rows; chunk c of row r sits at c ^ (r & 7): the 8 lanes a
      // ds_write_b128 services together (8 consecutive rows, one column) and the 16 lanes of a ds_read_b128 group (4 rows x 4
      // chunks) then touch every bank once (checked against the gfx950 lane groups).  Eight waves have 8 rows each: the lanes
      // that hold pixels it * 8 .. + 7 stage them for pass it (no vector-memory load is visible to hipcc in this kernel, so the
      // divergent region costs no `vmcnt(0)` at its merge)
      if constexpr (PX_NWV == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<f32x4*>(stage + (lane & 15) * PX_CSW + (((i * 4 + (lane >> 4)) ^ (lane & 7)) << 4)) = acc[i][ps];
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if constexpr (PX_NWV == 8) {
          if (((lane >> 3) & 1) == it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              *reinterpret_cast<f32x4*>(stage + (lane & 7) * PX_CSW + (((i * 4 + (lane >> 4)) ^ (lane & 7)) << 4)) = acc[i][ps];
          }
        }
        const int s = ps * 2 + it;
        const char* srow = stage + ((PX_NWV == 4 ? it * 8 : 0) + erow) * PX_CSW;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(srow + (((cc * 2) ^ (erow & 7)) << 4));
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(srow + (((cc * 2 + 1) ^ (erow & 7)) << 4));
        float v[8] = {a0[0] + bv[0], a0[1] + bv[1], a0[2] + bv[2], a0[3] + bv[3], a1[0] + bv[4], a1[1] + bv[5], a1[2] + bv[6], a1[3] + bv[7]};
        if constexpr (HR) {
          const bf16x8 r = __builtin_bit_cast(bf16x8, o.rr[s]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
        }
        if constexpr (HM) {
          const bf16x8 mk = __builtin_bit_cast(bf16x8, o.mm[s]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? v[e] : 0.f;
        }
        if (p.act == OSD_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (__bf16)v[e];
        const int mrow = m0 + s * 8 + erow;
        px_store16(__builtin_bit_cast(px_u32x4, ov), yrs, (mrow * p.out_stride + c) * 2);
      }
    }
  };

  // One unit.  Issue order per unit (what the counted waits below rely on; gfx9 counts loads, LDS-DMA and stores together, in order):
  //   [weights(u + 1): NDMA] [operands(u + 1): NOPS] MFMA(u) [pixels of the next block, if unit u + 1 starts one: NXL]
  //   epilogue(u): [stores(u): NST]
  // * top of unit u, "my part of weights(u) has landed": younger = operands(u), pixels (if u started a block), stores(u - 1);
  // * before MFMA(u) where u starts a block, "pixels have landed": younger = stores(u - 1), weights(u + 1), operands(u + 1);
  // * epilogue(u), "operands(u) have landed": younger = [pixels], stores(u - 1), weights(u + 1), operands(u + 1), [pixels of u + 1's block].
  // The first unit of a workgroup has no stores(u - 1) ahead of it and waits for everything instead (once per workgroup).
#ifdef OSD_PX_STAMPS      // diagnostic build: per-wave cycle sums into the buffer passed as act_scale_dev (tools/px_stamps.py)
  unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_ops = 0, st_epi = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#define PX_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define PX_T(var)
#endif
  auto unit = [&](int u, int slot, Ops& ocur, Ops& onxt, bool first) {
    const bool starts_block = first || (u % chunks) == 0;
    const bool next_starts_block = (u + 1 < u_last) && ((u + 1) % chunks) == 0;
    PX_T(ta);
    if (first) px_wait_vmcnt<0>();
    else if (starts_block) px_wait_vmcnt<NOPS + NXL + NST>();
    else px_wait_vmcnt<NOPS + NST>();
    PX_T(tb);
    __builtin_amdgcn_s_barrier();                       // everyone's part of weights(u) is visible; slot ^ 1 is no longer read
    PX_T(tc);
    issue_weights(u + 1, slot ^ 1);
    issue_operands(u + 1 < u_last ? u + 1 : u, onxt);   // (past the end: a harmless re-read, the instruction count stays fixed)
    if (starts_block && !first) px_wait_vmcnt<NST + NDMA + NOPS>();      // the block's pixel fragments (requested a unit ago)
    if (starts_block) {
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        if constexpr (NF == 2) px_tie(xf[0][ks], xf[0][ks + 1], xf[1][ks], xf[1][ks + 1]); else px_tie(xf[0][ks], xf[0][ks + 1]);
      }
    }
    PX_T(td);
    mfma_unit(slot);
#ifdef OSD_PX_STAMPS
    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[3][NF - 1]));
#endif
    PX_T(te);
    if (next_starts_block) {
      // the registers are free (every MFMA that reads them has been issued, and an MFMA reads its operands at issue): request
      // the next block's pixels now, under this unit's epilogue
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        if constexpr (NF == 2) px_tie(xf[0][ks], xf[0][ks + 1], xf[1][ks], xf[1][ks + 1]); else px_tie(xf[0][ks], xf[0][ks + 1]);
      }
      load_pixels((u + 1) / chunks);
      if (first) px_wait_vmcnt<0>(); else px_wait_vmcnt<NST + NDMA + NOPS + NXL>();      // operands(u)
    } else {
      if (first) px_wait_vmcnt<0>(); else px_wait_vmcnt<NST + NDMA + NOPS>();            // operands(u)
    }
    PX_T(tf);
    epilogue(u, ocur);
#ifdef OSD_PX_STAMPS
    PX_T(tg);
    st_wait += tb - ta; st_bar += tc - tb; st_issue += td - tc; st_mfma += te - td; st_ops += tf - te; st_epi += tg - tf;
#endif
  };

  Ops oa, ob;
  issue_weights(u_first, 0);
  load_pixels(u_first / chunks);
  issue_operands(u_first, oa);
  int u = u_first, slot = 0;
  bool first = true;
  while (u < u_last) {
    unit(u, slot, oa, ob, first);
    first = false; ++u; slot ^= 1;
    if (u >= u_last) break;
    unit(u, slot, ob, oa, false);
    ++u; slot ^= 1;
  }
  px_wait_vmcnt<0>();           // the zero fetches past the last unit: nothing may land in this LDS once the workgroup has left
#ifdef OSD_PX_STAMPS
  if (lane == 0 && p.act_scale_dev != nullptr) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.act_scale_dev)) + ((size_t)blockIdx.x * 8 + wave) * 10;
    o[0] = st_begin; o[1] = __builtin_amdgcn_s_memtime() - st_begin; o[2] = st_wait; o[3] = st_bar; o[4] = st_issue; o[5] = st_mfma; o[6] = st_ops; o[7] = st_epi;
    o[8] = (unsigned long long)(u_last - u_first);
    o[9] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (8 << 6) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg(((3 - 1) << 11) | (0 << 6) | 20) << 8);
  }
#endif
}

}  // namespace

int osd_conv_px_launch(const ConvKParams& pin, hipStream_t stream, bool wide_waves) {
  ConvKParams p = pin;
  if (p.R != 1 || p.S != 1 || p.sh != 1 || p.sw != 1 || p.ph != 0 || p.pw != 0 || p.x2 != nullptr || p.relu_in || p.n_seg > 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_px: a plain 1x1 / stride 1 conv of one tensor");
  if (p.sW != p.Cin || p.sH != p.W * p.Cin || p.sN != p.H * p.W * p.Cin)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_px: the input must be dense NHWC");
  if ((p.Cin != 64 && p.Cin != 128 && p.Cin != 256) || p.Cout % PX_CH || p.w_rows < p.Cout || p.Ktot != p.Cin || p.out_stride % 8)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_px: cin 64 / 128 / 256 (the pixel operand lives in registers), cout in 64s, 16-byte output rows");
  if (p.res_mode != OSD_RES_NONE && (p.res_mode != OSD_RES_SAME || p.res_stride % 8))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_px: residual of the output's own size only");
  if (p.act != OSD_ACT_NONE && p.act != OSD_ACT_RELU) return osd_fail(OSD_ERR_UNSUPPORTED, "conv_px: activation none / relu");
  const long long lim = 0x7fffffffLL;
  if ((long long)p.M * p.Cin * 2 >= lim || (long long)p.M * p.out_stride * 2 >= lim || (long long)p.w_rows * p.Ktot * 2 >= lim ||
      (p.res_mode != OSD_RES_NONE && (long long)p.M * p.res_stride * 2 >= lim))
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv_px: tensors below 2 GiB (32-bit buffer offsets)");
  p.tilesM = cdiv(p.M, PX_BP);
  p.tilesN = p.Cout / PX_CH;
  p.KT = p.Cin / 32;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  }
  const long long units = (long long)p.tilesM * p.tilesN;
  if (units > lim) return osd_fail(OSD_ERR_INVALID_ARG, "conv_px: bad grid");
  const unsigned grid = (unsigned)(units < 2LL * cus ? units : 2LL * cus);
  const bool hr = p.res_mode == OSD_RES_SAME, hm = p.mask != nullptr;
  // eight waves of 16 pixels (algo 49: twice the waves per CU to overlap fetch issue, MFMAs and epilogues) or four of 32 (algo 50)
  const bool wide = wide_waves;
#define OSD_PX_LAUNCH2(KS, HR, HM, PXW)                                                                                    \
  do {                                                                                                                     \
    auto kern = conv_px_kernel<KS, HR, HM, PXW>;                                                                           \
    constexpr int lds = 2 * (((KS * 32 + 63) / 64) * PX_CH * 128) + PX_STGS;      /* 81,920 B at K = 256: two workgroups per CU */ \
    static bool attr = false;                                                                                              \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (PX_BP / PXW)), lds, stream, p);                                        \
  } while (0)
#define OSD_PX_LAUNCH(KS, HR, HM)                                                                                          \
  do {                                                                                                                     \
    if (wide) OSD_PX_LAUNCH2(KS, HR, HM, 32); else OSD_PX_LAUNCH2(KS, HR, HM, 16);                                         \
  } while (0)
#define OSD_PX_K(KS)                                                                                                       \
  do {                                                                                                                     \
    if (hr && hm) OSD_PX_LAUNCH(KS, true, true);                                                                           \
    else if (hr) OSD_PX_LAUNCH(KS, true, false);                                                                           \
    else if (hm) OSD_PX_LAUNCH(KS, false, true);                                                                           \
    else OSD_PX_LAUNCH(KS, false, false);                                                                                  \
  } while (0)
  if (p.Cin == 256) OSD_PX_K(8);
  else if (p.Cin == 128) OSD_PX_K(4);
  else OSD_PX_K(2);
#undef OSD_PX_K
#undef OSD_PX_LAUNCH
#undef OSD_PX_LAUNCH2
  return osd_check_launch("conv_px");
}
