// conv_igemm_sp — the row-reuse 3x3 convolution (conv_igemm_xr.hip: bf16, 3x3 / stride 1 / pad 1, 256 x 256 tile, 8 waves,
// padded pixel image fetched once per filter row) with the operand fragments SOFTWARE-PIPELINED through registers.
//
// Why: in conv_xr_kernel every half stage (32 of K) opens with 12 ds_read_b128 per wave and the MFMAs wait for them — all 8
// waves at once, twice per stage: ~900 of a stage's ~3,000 cycles are the matrix pipe waiting for LDS (the loop WITHOUT any
// global traffic runs at 0.59 of the MFMA peak).  Here the fragments of half stage u + 1 are read while the MFMAs
// of half stage u run:
//   * the 4 weight fragments have two register sets; the 8 pixel fragments are REPLACED one by one — fragment j of the next
//     half stage is read as soon as the 4 MFMAs that use the current fragment j have been issued (MFMA order: pixel fragment
//     outer, weight fragment inner) — so the pipeline costs 16 more VGPRs, not 48;
//   * reading half stage u + 1 during u needs its LDS bytes visible one half stage early, so the ONE barrier per stage moves
//     from the stage boundary to the middle of the stage: barrier B_w sits between the two halves of weight stage w; before
//     it every wave has waited for its own LDS-DMA of weight stage w + 1 (and, in a group's last tap, of the next pixel image)
//     and for its last reads of stage w's buffers; after it stage w + 1 may be read and weight stage w + 2 is fetched into the
//     buffer stage w just vacated.  Same 2-deep rings, same LDS budget, every DMA still has a full stage of flight time.
//   * LDS-DMA by `buffer_load_dwordx4 ... lds` with hardware bounds checking: a lane whose row is zero padding (vertical
//     border, M tail, Cout tail) presents an out-of-range offset and the hardware writes zeros — no zero page, no 64-bit
//     per-lane pointers (8 VGPRs and two VALU selects per instruction less).
// The accumulation order over K is conv_xr_kernel's, so the two kernels' outputs are bit-identical (tested).
//
// GENW (round 4): ANY map width.  The padded image above needs tiles that start at x = 0 and hold whole lines (W in 64 / 128 /
// 256).  The general form keeps the tile's 256 pixels CONSECUTIVE in LDS (row 16 + r <- pixel m0 + r, no pad rows) plus one
// 8-row halo instruction on either side (rows 8..15 <- pixels m0 - 8 .. m0 - 1, rows 272..279 <- m0 + 256 .. m0 + 263), so tap s
// of pixel r still reads row 16 + r + s - 1 and every fragment base stays a multiple of 16 (the same conflict-free key table).
// What the zero pad rows did — x = -1 and x = W read zeros — is done in registers: a lane whose pixel sits at x = 0 (x = W - 1)
// zeroes its tap-0 (tap-2) fragment right before the MFMAs that consume it (4 v_cndmask per fragment, long after the read has
// landed, so the software pipelining of the reads is untouched).  Same K order, and a masked value is the same exact zero a pad
// row held: on the widths both forms accept the outputs are bit-identical (tested).  All lanes' vertical validity, the image
// borders and the M tail stay hardware bounds checks of the DMA.  Every wave issues FIVE pixel instructions per image (the
// halo one is all out-of-range lanes — zeros into the unused rows 0..7 — on waves 1..6), so the counted waits are uniform.
//
// TMX (round 4): pixel fragments per wave.  8 = the 256-pixel tile above; 4 = a 128-pixel x 256-channel tile on the same eight
// waves (64 x 64 per wave: 8 instead of 6 LDS fragment reads per 16 MFMAs, half the pixel DMA per workgroup, the same weight
// stages) for the launches whose pixel count gives the 256-pixel tile 25-100 workgroups on 256 CUs (layer3 / layer4 / P5-P7
// at bs = 8): twice the workgroups.  Same K order and per-pixel arithmetic: bit-identical outputs (tested).
#include "osd_common.h"
#include "conv_params.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void sp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void sp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes, lane l lands at lds_dst + 16 l; source = buffer base + voff (bytes),
// zeros when voff is outside the buffer
__device__ __forceinline__ void sp_dma16(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "s"(rsrc)
      : "memory");
}

__device__ __forceinline__ i32x4 sp_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));       // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// swizzle key of a padded pixel-image row: see conv_igemm_xr.hip (conflict-free for the three tap shifts)
__device__ __forceinline__ int sp_key(int row) { return (int)((0x4430066774400322ull >> ((row & 15) * 4)) & 7); }

constexpr int SP_BM = 256, SP_BN = 256, SP_KB = 128, SP_BKE = 64, SP_EPC = 8;
constexpr int SP_WN = 4, SP_TM = 8, SP_TN = 4;
constexpr int SP_AROWS = 336;
constexpr int SP_ABYTES = SP_AROWS * SP_KB;
constexpr int SP_BBYTES = SP_BN * SP_KB;
constexpr int SP_LDS = 2 * SP_ABYTES + 2 * SP_BBYTES;
// per tile (round 6): the pixel images and weight stages of a BM x BN tile, or its epilogue's staging (8 waves x 2 regions x rows x
// (64 channels x 4 B + 16)) — whichever is larger.  The 128 x 128 tile needs 76 KB, so TWO of its workgroups share a CU (117 VGPRs): its
// launches are the latency-sized ones (a tower layer over P5 - P7: 134 workgroups waiting for cold weights), and one's waits now sit
// under the other's MFMAs where the step leaves them fewer CUs than workgroups
constexpr int sp_arows(int bm) { return bm + 16 + 16 * (bm / 64); }
constexpr int sp_lds_bytes(int tm, int wn) {
  const int bm = (8 / wn) * tm * 16, bn = wn * SP_TN * 16;
  const int ring = 2 * sp_arows(bm) * SP_KB + 2 * bn * SP_KB;
  const int npass = tm >= 8 ? tm / 2 : (tm >= 2 ? 2 : 1);
  const int epi = 8 * 2 * ((tm / npass) * 16) * (SP_TN * 16 * 4 + 16);
  return ring > epi ? ring : epi;
}
static_assert(sp_arows(SP_BM) == SP_AROWS && sp_lds_bytes(8, 4) == SP_LDS, "the 256 x 256 tile's LDS plan");
constexpr unsigned SP_OOB = 0x80000000u;            // beyond any buffer of < 2 GiB: the DMA writes zeros

// GNB: the epilogue also gathers GroupNorm statistics (1: backward or forward per segment, 2: forward only; conv_epilogue.h); GENW: any
// width; TMX: 16-pixel fragments per wave; WNX (round 5): wave columns — 4 = a 256-channel tile on 2 x 4 waves, 2 = a 128-channel
// tile on 4 x 2 waves (layer2's 128 -> 128 3x3 convs, resnet.py:295-315: the 256-channel tile would spend half its MFMAs and half
// its weight stages on channels that do not exist)
template <int GNB, bool GENW, int TMX, int WNX = 4>
__global__ void __launch_bounds__(512) conv_sp_kernel(ConvKParams p) {
  typedef __bf16 T;
  constexpr int TM = TMX, TN = SP_TN, KB = SP_KB;
  constexpr int WN = WNX, WM = 8 / WNX;                 // wave grid: WM rows of pixels x WN columns of 64 channels
  constexpr int BM = WM * TM * 16;                      // pixels per tile
  constexpr int BN = WN * TN * 16;                      // channels per tile
  constexpr int NBW = BN / 64;                          // weight DMA instructions (8 rows each) per wave and stage
  constexpr int BBYTES = BN * KB;                       // one weight stage
  constexpr int ABYTES = sp_arows(BM) * KB;             // one pixel image (SP_ABYTES for the 256-pixel tiles)
  static_assert(BM <= SP_BM && BN <= SP_BN && BM % 64 == 0, "tile geometry");
  constexpr int PA = BM / 64;                           // pixel DMA instructions (8 rows each) per wave and image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int t;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = t % p.tilesN;
  int tile_m = t / p.tilesN;
  const ConvView q = conv_select_view(p, tile_m);
  const int q_H = q.H, q_W = q.W, q_M = q.M, q_HoWo = q.HoWo;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  [[maybe_unused]] const int logw = __builtin_ctz((unsigned)q_W);       // !GENW: W in {64, 128, 256} (checked by the launcher)
  const int Cin = p.Cin;

  const i32x4 xrs = sp_make_rsrc(q.x, (unsigned)q_M * (unsigned)Cin * 2u);            // H = Ho, W = Wo: M pixels x Cin
  const i32x4 wrs = sp_make_rsrc(q.w, (unsigned)p.w_rows * (unsigned)p.Ktot * 2u);

  // ---- zero both pixel images once: the pad rows stay zero for the whole K loop (GENW: no pad rows; every row that is read
  // is rewritten by each image's DMA) ----
  if constexpr (!GENW) {
    uint4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * ABYTES / 16; i += 512) *reinterpret_cast<uint4*>(smem + i * 16) = z;
  }

  // ---- per-lane DMA sources (byte offsets).  One wave-instruction = 8 rows x 128 B; wave w owns pixel instructions
  // 4w .. 4w+3 and weight instructions 4w .. 4w+3; instruction i covers tile rows (4w + i) * 8 + lrow.  Few registers:
  //   pixels:  ((n_img H + ho - 1) W + wo) Cin = (m - W) Cin is LINEAR in the pixel index m, so instruction i's offset is
  //            a_par[i & 1] + i * 8 * Cin * 2 (the swizzled chunk depends on the row's parity group only); whether the input
  //            line ho - 1 + kr exists is one bit per (instruction, filter row) in a_valid (M tail: all clear)
  //   weights: b_par[i & 1] + i * 8 * Ktot * 2; rows past w_rows are past the end of the buffer by themselves
  const int lrow = lane >> 3, lpos = lane & 7;
  unsigned a_par[2], b_par[2], a_valid = 0u;          // a_valid bit i * 3 + kr; GENW: bits 12..14 = the halo instruction
  unsigned a_dst[4];
  [[maybe_unused]] unsigned h_off = 0u, h_dst = 0u;   // GENW: the halo instruction's source offset (filter row 0, slab 0), LDS row
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave * PA + i) * 8 + lrow;
    const int m = m0 + row;
    int R, ad;
    if constexpr (GENW) {
      R = 16 + row;
      ad = (16 + (wave * PA + i) * 8) * KB;
    } else {
      R = 16 + row + 16 * (row >> logw);
      ad = (16 + (wave * PA + i) * 8 + 16 * (((wave * PA + i) * 8) >> logw)) * KB;
    }
    a_dst[i] = (unsigned)__builtin_amdgcn_readfirstlane(ad);
    if (i < 2) {
      a_par[i] = (unsigned)((m - i * 8 - q_W) * Cin + ((lpos ^ sp_key(R)) * SP_EPC)) * 2u;      // instruction 0's row + parity chunk
      const int brow = (wave * NBW + i) * 8 + lrow;     // weight instruction i of this wave: NBW per wave whatever the pixel tile
      b_par[i] = (unsigned)((n0 + brow - i * 8) * p.Ktot + ((lpos ^ ((brow >> 1) & 7)) * SP_EPC)) * 2u;
    }
    if (m < q_M) {
      int ho;
      if constexpr (GENW) ho = (int)((unsigned)(m % q_HoWo) / (unsigned)q_W);
      else ho = (m % q_HoWo) >> logw;
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
        if ((unsigned)(ho - 1 + kr) < (unsigned)q_H) a_valid |= 1u << (i * 3 + kr);
    }
  }
  // GENW: which of this lane's 8 fragment pixels (tile row (wm * 8 + j) * 16 + (lane & 15)) sit on the left (bit j) / right
  // (bit 8 + j) border of their line: the tap-0 / tap-2 fragment of such a pixel is zeroed in registers
  [[maybe_unused]] unsigned edge = 0u;
  if constexpr (GENW) {
    const int hrow = wave == 0 ? -8 + lrow : BM + lrow;         // halo rows: waves 0 and 7; the other waves fetch nothing
    const int hm = m0 + hrow;
    h_dst = (unsigned)__builtin_amdgcn_readfirstlane(wave == 0 ? 8 * KB : (wave == 7 ? (16 + BM) * KB : 0));
    h_off = (unsigned)((hm - q_W) * Cin + ((lpos ^ sp_key(16 + hrow)) * SP_EPC)) * 2u;
    if ((wave == 0 || wave == 7) && hm >= 0 && hm < q_M) {
      const int ho = (int)((unsigned)(hm % q_HoWo) / (unsigned)q_W);
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
        if ((unsigned)(ho - 1 + kr) < (unsigned)q_H) a_valid |= 1u << (12 + kr);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const unsigned x = (unsigned)(m0 + (wm * TM + j) * 16 + (lane & 15)) % (unsigned)q_W;
      if (x == 0u) edge |= 1u << j;
      if (x == (unsigned)q_W - 1u) edge |= 1u << (8 + j);
    }
  }
  const int line_bytes = q_W * Cin * 2;

  const unsigned a_lds[2] = {lds0, lds0 + (unsigned)ABYTES};
  const unsigned b_lds[2] = {lds0 + 2u * ABYTES, lds0 + 2u * ABYTES + (unsigned)BBYTES};

  // pixel instruction i of group (kr, kc) into pixel image `buf`
  auto issue_a = [&](int buf, int i, int kr, int kc) {
    const unsigned off = ((a_valid >> (i * 3 + kr)) & 1u) ? a_par[i & 1] + (unsigned)(i * 16 * Cin + kr * line_bytes + kc * 2) : SP_OOB;
    sp_dma16(xrs, off, a_lds[buf] + a_dst[i]);
  };
  // GENW: the halo instruction of group (kr, kc) — issued by every wave AHEAD of its four tile instructions
  [[maybe_unused]] auto issue_halo = [&](int buf, int kr, int kc) {
    const unsigned off = ((a_valid >> (12 + kr)) & 1u) ? h_off + (unsigned)(kr * line_bytes + kc * 2) : SP_OOB;
    sp_dma16(xrs, off, a_lds[buf] + h_dst);
  };
  constexpr int NIMG = GENW ? PA + 1 : PA;              // pixel DMA instructions per wave and image
  // weight instruction i of the stage whose K offset is koff (elements) into weight stage `buf`
  auto issue_b = [&](int buf, int i, int koff) {
    sp_dma16(wrs, b_par[i & 1] + (unsigned)((i * 8 * p.Ktot + koff) * 2), b_lds[buf] + (unsigned)((wave * NBW + i) * 1024));
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fkq = lane >> 4;
  // fragment offsets of the first K half; the second half is the same address with bit 6 flipped (the 16-byte chunk index
  // (kb * 4 + fkq) ^ key differs in bit 2 only, and every base added later is a multiple of 128)
  const int w_off0 = (wn * TN * 16 + frow) * KB + ((fkq ^ ((frow >> 1) & 7)) << 4);      // tile i adds i * 16 rows (same key)
  int x_lane[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int e = frow + s - 1;
    x_lane[s] = e * KB + ((fkq ^ sp_key(e)) << 4);
  }
  int x_frag[TM];                                       // wave-uniform: first padded row of fragment j (a multiple of 16)
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int r0 = (wm * TM + j) * 16;
    if constexpr (GENW) x_frag[j] = (16 + r0) * KB;
    else x_frag[j] = (16 + r0 + 16 * (r0 >> logw)) * KB;
  }

  const int nslab = Cin / SP_BKE;
  const int G = 3 * nslab;                              // groups = (filter row, 64-channel slab)

  // fragment registers: the pixel set is replaced in place, the weight sets alternate
  uint4 xf[TM], wf0[TN], wf1[TN];
#ifdef OSD_SP_STAMPS      // diagnostic build: per-wave cycle stamps into the buffer passed as act_scale_dev (tools/sp_stamps.py)
  unsigned long long st_lgkm = 0, st_vm = 0, st_bar = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif
  f32x4 bq[2];

  // One half stage: MFMAs on (wcur, xf) while the NEXT half stage's fragments are read — weights into wnxt up front, pixel
  // fragment j into xf[j] right behind the 4 MFMAs that consumed it.  ab_n / s_n / kb_n / bb_n: pixel image, tap, K half and
  // weight stage of the NEXT half stage.  DMA: FB -> the 4 weight instructions of the stage at K offset koff into weight
  // buffer dma_bb; FA -> the 4 pixel instructions of group (nkr, nkc) into image dma_ab; both spread over the MFMA groups.
  auto half_stage = [&](auto cur_tag, auto bar_tag, uint4 (&wcur)[TN], uint4 (&wnxt)[TN], int ab_n, auto sn_tag, auto kbn_tag, int bb_n,
                        auto has_next, auto fb_tag, int dma_bb, int koff, auto fa_tag, int dma_ab, int nkr, int nkc) {
    // cur_tag: the tap (0 / 1 / 2) of THIS half stage's pixel fragments (GENW: taps 0 and 2 are masked at the line borders)
    [[maybe_unused]] constexpr int s_cur = decltype(cur_tag)::value;
    // BAR: 0 = no barrier in this half stage; 1 / 2 = the stage's barrier, with `vmcnt(0)` / `vmcnt(4)` before it.  The
    // barrier sits BEHIND the first MFMA group: the last fragment read of the previous half stage was issued just before that
    // group, so the lgkmcnt(0) the barrier needs (all my reads of the buffers it releases are complete) has 4 MFMAs of cover.
    constexpr int BAR = decltype(bar_tag)::value;
    constexpr bool dma_early = true;
    constexpr int s_n = decltype(sn_tag)::value, kb_n = decltype(kbn_tag)::value;
    constexpr bool NEXT = decltype(has_next)::value, FB = decltype(fb_tag)::value, FA = decltype(fa_tag)::value;
    const char* xs = smem + ab_n * ABYTES;
    const char* ws = smem + 2 * ABYTES + bb_n * BBYTES;
#ifndef OSD_SP_BAR_AT
#define OSD_SP_BAR_AT 1
#endif
    constexpr int BAR_AT = OSD_SP_BAR_AT;               // MFMA groups of this half stage issued ahead of its barrier
    if constexpr (BAR != 0 && BAR_AT == 0) {
      sp_wait_lgkm0();
      if constexpr (BAR == 2) sp_wait_vmcnt<NIMG>(); else sp_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (NEXT && (BAR == 0 || BAR_AT == 0)) {
#pragma unroll
      for (int i = 0; i < TN; ++i) wnxt[i] = *reinterpret_cast<const uint4*>(ws + (w_off0 ^ (kb_n * 64)) + i * 16 * KB);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      if constexpr (GENW && s_cur != 1) {               // x = -1 / x = W of this pixel's line: the zero a pad row would hold
        const bool z = (edge >> (j + (s_cur == 2 ? 8 : 0))) & 1u;
        xf[j].x = z ? 0u : xf[j].x; xf[j].y = z ? 0u : xf[j].y; xf[j].z = z ? 0u : xf[j].z; xf[j].w = z ? 0u : xf[j].w;
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wcur[i]),
                                                            *reinterpret_cast<const bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
      if constexpr (BAR != 0 && BAR_AT != 0) {
        if (j == BAR_AT - 1) {
          __builtin_amdgcn_sched_barrier(0);
#ifdef OSD_SP_STAMPS
          const unsigned long long ta = __builtin_amdgcn_s_memtime();
          sp_wait_lgkm0();
          const unsigned long long tb = __builtin_amdgcn_s_memtime();
          if constexpr (BAR == 2) sp_wait_vmcnt<NIMG>(); else sp_wait_vmcnt<0>();
          const unsigned long long tc = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_s_barrier();
          const unsigned long long td = __builtin_amdgcn_s_memtime();
          st_lgkm += tb - ta; st_vm += tc - tb; st_bar += td - tc;
#else
          sp_wait_lgkm0();
          if constexpr (BAR == 2) sp_wait_vmcnt<NIMG>(); else sp_wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
#endif
          if constexpr (NEXT) {
#pragma unroll
            for (int i = 0; i < TN; ++i) wnxt[i] = *reinterpret_cast<const uint4*>(ws + (w_off0 ^ (kb_n * 64)) + i * 16 * KB);
          }
        }
      }
      if constexpr (NEXT) xf[j] = *reinterpret_cast<const uint4*>(xs + x_frag[j] + (x_lane[s_n] ^ (kb_n * 64)));
#ifndef OSD_SP_NO_DMA
      // (measured: the two waves of a SIMD issuing their DMA in opposite halves of the half stage — two copies of the K loop —
      // is 5 % SLOWER than all waves issuing at the same place)
      if constexpr (FB) {
        if constexpr (TM >= 8) {
          if ((j < 4) == dma_early && (j & 3) < NBW) issue_b(dma_bb, j & 3, koff);
        } else {
          if (j < NBW) issue_b(dma_bb, j, koff);          // TM = 4: one weight instruction per MFMA group
        }
      }
      if constexpr (FA) {
        if constexpr (GENW) {
          if (j == 0) issue_halo(dma_ab, nkr, nkc);
        }
        if (j < PA) issue_a(dma_ab, j, nkr, nkc);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);                // keep the replacement read behind its fragment's last MFMA
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using Y = std::true_type;
  using N_ = std::false_type;

  // ---- prologue: pixel image 0, weight stages 0 and 1; then the first fragments ----
  __syncthreads();                                      // the zero fill is complete before any DMA lands on real rows
  if constexpr (GENW) issue_halo(0, 0, 0);
#pragma unroll
  for (int i = 0; i < PA; ++i) issue_a(0, i, 0, 0);
#pragma unroll
  for (int i = 0; i < NBW; ++i) issue_b(0, i, 0);
#pragma unroll
  for (int i = 0; i < NBW; ++i) issue_b(1, i, Cin);     // tap 1 of group 0
  sp_wait_vmcnt<NBW>();                                 // image 0 and weight stage 0 have landed (mine)
  __builtin_amdgcn_s_barrier();                         // ... and everyone's
  {
    const char* xs = smem;
    const char* ws = smem + 2 * ABYTES;
#pragma unroll
    for (int i = 0; i < TN; ++i) wf0[i] = *reinterpret_cast<const uint4*>(ws + w_off0 + i * 16 * KB);
#pragma unroll
    for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const uint4*>(xs + x_frag[j] + x_lane[0]);
  }

#ifdef OSD_SP_STAMPS
  const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
#endif
  // (round 6, measured and removed — tools/sp_lib_bench.py, tower P3 + P4 launch, two interleaved rounds: default 125 - 127 us; `s_setprio 1`
  // for waves 4 - 7 (MI355X_MICROARCH.md, two waves per SIMD, item 4) 131 - 133; for waves 0 - 3 129 - 132; SIMD partners issuing their weight DMA
  // in opposite halves of the half stage through a wave-uniform run-time branch 216 - 218: the branch un-pins the schedule)
  // Group loop.  Weight stage index w = 3 g + s alternates buffers, and with 3 stages per group a group flips the parity:
  // (ab, bb) are the image / weight buffers of the group's tap 0.
  int ab = 0, bb = 0;
  int kr = 0, kc = 0;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  {
    for (int g = 0; g + 1 < G; ++g) {                     // every group but the last
      // next group: the three FILTER ROWS of one 64-channel slab one after the other, then the next slab (round 5; until then
      // all slabs of a filter row first).  The image of filter row kr + 1 is the image of kr shifted by one map line: two thirds
      // of its lines were fetched a group ago and are still in the XCD's L2 (~64 KB per workgroup between the two reads; in
      // the old order a whole pass over the other slabs lay between them, ~6 MB per XCD against 4 MB of L2).  PMC FETCH_SIZE of
      // the tower launch: 145.5 MB for 67 MB of operands before (profiles/r5_pmc_traffic_by_kernel.txt)
#ifdef OSD_SP_ROW_MAJOR_K
      int nkr = kr, nkc = kc + SP_BKE;
      if (nkc >= Cin) { nkc = 0; ++nkr; }
#else
      int nkr = kr + 1, nkc = kc;
      if (nkr >= 3) { nkr = 0; nkc += SP_BKE; }
#endif
      const int kbase = kr * 3 * Cin + kc;                // K offset of tap 0 of this group; tap s adds s * Cin
      const int knext = nkr * 3 * Cin + nkc;              // ... of the next group
      // tap 0, first half: reads its own second half and fetches the NEXT group's pixel image (its buffer was vacated at the
      // previous group's last barrier)
      half_stage(I0(), B0(), wf0, wf1, ab, I0(), I1(), bb, Y(), N_(), 0, 0, Y(), ab ^ 1, nkr, nkc);
      // second half, barrier B(g, 0) inside (the wait leaves the 4 image DMAs in flight): weight stage (g, 1) visible, buffer bb
      // vacated -> fetch weight stage (g, 2) into it
      half_stage(I0(), B2(), wf1, wf0, ab, I1(), I0(), bb ^ 1, Y(), Y(), bb, kbase + 2 * Cin, N_(), 0, 0, 0);
      // tap 1
      half_stage(I1(), B0(), wf0, wf1, ab, I1(), I1(), bb ^ 1, Y(), N_(), 0, 0, N_(), 0, 0, 0);
      // B(g, 1): weight stage (g, 2) visible, buffer bb ^ 1 vacated -> weight stage (g + 1, 0)
      half_stage(I1(), B1(), wf1, wf0, ab, I2(), I0(), bb, Y(), Y(), bb ^ 1, knext, N_(), 0, 0, 0);
      // tap 2
      half_stage(I2(), B0(), wf0, wf1, ab, I2(), I1(), bb, Y(), N_(), 0, 0, N_(), 0, 0, 0);
      // B(g, 2): next image + weight stage (g + 1, 0) visible; image ab and weight buffer bb vacated -> weight stage (g + 1, 1)
      half_stage(I2(), B1(), wf1, wf0, ab ^ 1, I0(), I0(), bb ^ 1, Y(), Y(), bb, knext + Cin, N_(), 0, 0, 0);
      ab ^= 1;
      bb ^= 1;
      kr = nkr;
      kc = nkc;
    }
    {
      // last group: no next image; weight stage (g, 1) is in flight, (g, 2) is fetched below
      const int kbase = kr * 3 * Cin + kc;
      half_stage(I0(), B0(), wf0, wf1, ab, I0(), I1(), bb, Y(), N_(), 0, 0, N_(), 0, 0, 0);
      half_stage(I0(), B1(), wf1, wf0, ab, I1(), I0(), bb ^ 1, Y(), Y(), bb, kbase + 2 * Cin, N_(), 0, 0, 0);
      half_stage(I1(), B0(), wf0, wf1, ab, I1(), I1(), bb ^ 1, Y(), N_(), 0, 0, N_(), 0, 0, 0);
      half_stage(I1(), B1(), wf1, wf0, ab, I2(), I0(), bb, Y(), N_(), 0, 0, N_(), 0, 0, 0);
      // the epilogue's bias values (8 consecutive channels per lane, the same in all of its passes) travel while the last
      // MFMAs run: loaded at the start of the epilogue they cost it a full memory latency
      {
        const int c = n0 + wn * (TN * 16) + (lane % (TN * 16 / SP_EPC)) * SP_EPC;
        const float* bsrc = q.bias + (c + SP_EPC <= p.w_rows ? c : 0);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bsrc), b1 = *reinterpret_cast<const f32x4*>(bsrc + 4);
        bq[0] = b0; bq[1] = b1;
      }
      half_stage(I2(), B0(), wf0, wf1, ab, I2(), I1(), bb, Y(), N_(), 0, 0, N_(), 0, 0, 0);
      half_stage(I2(), B0(), wf1, wf0, ab, I0(), I0(), bb, N_(), N_(), 0, 0, N_(), 0, 0, 0);
    }

  }

#ifdef OSD_SP_STAMPS
  const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef OSD_SP_STAMPS
  const unsigned long long st_t3 = __builtin_amdgcn_s_memtime();
#endif
#ifdef OSD_SP_NO_EPI      // diagnostic build: keep the accumulators live, store nothing
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(acc[i][j]));
  return;
#endif
  const float pre_bias[8] = {bq[0][0], bq[0][1], bq[0][2], bq[0][3], bq[1][0], bq[1][1], bq[1][2], bq[1][3]};
  conv_epilogue<T, TM, TN, true, GNB, (TM <= 4)>(acc, p, q, smem, wave, wm, wn, lane, m0, n0, pre_bias);   // 8 waves x 2 x 8,704 B <= SP_LDS
#ifdef OSD_SP_STAMPS
  if (lane == 0 && p.act_scale_dev != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_t4 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.act_scale_dev)) + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = st_t0; o[1] = st_t1 - st_t0; o[2] = st_t2 - st_t1; o[3] = st_t3 - st_t2; o[4] = st_t4 - st_t3; o[5] = st_lgkm; o[6] = st_vm; o[7] = st_bar;
  }
#endif
}

}  // namespace

int osd_conv_sp_launch(const ConvKParams& pin, hipStream_t stream, bool general_width, bool half_tile, bool small_tile) {
  ConvKParams p = pin;
  if (p.R != 3 || p.S != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): 3x3 stride 1 pad 1 only");
  if (p.Cin % SP_BKE != 0 || p.sW != p.Cin) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): dense NHWC input with cin %% 64 == 0");
  if (p.relu_in) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): relu_in prologue not supported");
  if ((long long)p.w_rows * p.Ktot * 2 >= 0x7fffffffLL) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): weights over 2 GiB");
  // any width: dense maps (Wo == W, Ho == H) under 2 GiB.  The padded-image form needs W in 64 / 128 / 256 (tiles start at x = 0
  // and hold whole lines); every other width — or all of them with general_width — runs the consecutive-rows form (GENW)
  bool genw = general_width;
  auto ok = [&](int w, int wo, int h, int ho, int m, int sh) {
    if (!(w == 64 || w == 128 || (w == 256 && !half_tile && !small_tile))) genw = true;      // whole lines per tile, tiles start at x = 0
    return w >= 1 && wo == w && ho == h && sh == w * p.Cin && (long long)m * p.Cin * 2 < 0x7fffffffLL;
  };
  // round 5: convs with <= 128 output channels run a 256-pixel x 128-channel tile on 4 x 2 waves (64 x 64 per wave, as the
  // 128-pixel tile's waves): no MFMAs and no weight stages for channels that do not exist
  // small_tile (round 5, algo 7): 128 pixels x 128 channels (4 x 2 waves of 32 x 64) for the launches whose WEIGHT stream bounds a
  // workgroup — layer4's 3x3 (6,400 pixels, K = 4,608): a 256-channel tile streams 2.4 MB of weights through its CU's texture path
  // whatever its pixel count, 100 workgroups on 256 CUs; 128-channel tiles halve that per workgroup and double the workgroups
  const bool narrow = p.Cout <= 128 || small_tile;
  if (narrow && half_tile) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): the 128-pixel tile is built for 256-channel tiles only");
  if (small_tile && p.gn_groups > 0) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): no GroupNorm statistics on the small tile");
  const int BMx = (half_tile || small_tile) ? SP_BM / 2 : SP_BM;
  const int BNx = narrow ? SP_BN / 2 : SP_BN;
  p.tilesM = cdiv(p.M, BMx);
  if (p.n_seg > 0) {
    p.tilesM = 0;
    for (int i = 0; i < p.n_seg; ++i) {
      if (!ok(p.seg[i].W, p.seg[i].Wo, p.seg[i].H, p.seg[i].Ho, p.seg[i].M, p.seg[i].sH))
        return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): segment %d (width %d) is not a dense map under 2 GiB", i, p.seg[i].W);
      p.seg[i].tile_begin = p.tilesM;
      p.tilesM += cdiv(p.seg[i].M, BMx);
    }
  } else if (!ok(p.W, p.Wo, p.H, p.Ho, p.M, p.sH)) {
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): not a dense map under 2 GiB (width %d)", p.W);
  }
  p.tilesN = cdiv(p.Cout, BNx);
  p.KT = p.Ktot / SP_BKE;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(8, 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<1, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(8, 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<2, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(8, 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(8, 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(4, 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(4, 4));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, false, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(4, 2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, true, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(4, 2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, false, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(2, 2));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sp_kernel<0, true, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, sp_lds_bytes(2, 2));
    attr_done = true;
  }
  const long long nblocks = (long long)p.tilesM * p.tilesN;
  if (nblocks <= 0 || nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv(sp): bad grid");
  if (p.gn_groups > 0) {
    // GroupNorm-backward statistics in the epilogue: whole tiles only (fast path), a wave's 128 rows inside one image, one
    // 16-byte channel chunk inside one group
    if (genw || half_tile) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): GroupNorm statistics need map widths 64 / 128 / 256 and the 256-pixel tile");
    if (p.n_seg <= 0 || p.Cout % SP_BN != 0 || p.out_stride % 8 != 0 || p.Cout % p.gn_groups != 0 || (p.Cout / p.gn_groups) % 8 != 0 ||
        p.res_mode != OSD_RES_NONE || p.act != OSD_ACT_NONE)
      return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): GroupNorm statistics need a plain multi-segment conv with Cout %% 256 == 0 and groups of whole 16-byte chunks");
    for (int i = 0; i < p.n_seg; ++i) {
      if (!p.seg[i].gn.u && !p.seg[i].gn.ws) continue;
      if (p.seg[i].M % SP_BM != 0 || (p.seg[i].Ho * p.seg[i].Wo) % (SP_BM / 2) != 0 || p.seg[i].M / (p.seg[i].Ho * p.seg[i].Wo) != p.gn_n)
        return osd_fail(OSD_ERR_UNSUPPORTED, "conv(sp): GroupNorm statistics need images of whole 128-pixel runs (segment %d)", i);
    }
    bool backward = false;
    for (int i = 0; i < p.n_seg; ++i) backward = backward || p.seg[i].gn.u != nullptr;
    if (backward) hipLaunchKernelGGL((conv_sp_kernel<1, false, 8>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(8, 4), stream, p);
    else hipLaunchKernelGGL((conv_sp_kernel<2, false, 8>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(8, 4), stream, p);
  } else if (small_tile) {
    if (genw) hipLaunchKernelGGL((conv_sp_kernel<0, true, 2, 2>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(2, 2), stream, p);
    else hipLaunchKernelGGL((conv_sp_kernel<0, false, 2, 2>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(2, 2), stream, p);
  } else if (narrow) {
    if (genw) hipLaunchKernelGGL((conv_sp_kernel<0, true, 4, 2>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(4, 2), stream, p);
    else hipLaunchKernelGGL((conv_sp_kernel<0, false, 4, 2>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(4, 2), stream, p);
  } else if (half_tile) {
    if (genw) hipLaunchKernelGGL((conv_sp_kernel<0, true, 4>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(4, 4), stream, p);
    else hipLaunchKernelGGL((conv_sp_kernel<0, false, 4>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(4, 4), stream, p);
  } else if (genw) {
    hipLaunchKernelGGL((conv_sp_kernel<0, true, 8>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(8, 4), stream, p);
  } else {
    hipLaunchKernelGGL((conv_sp_kernel<0, false, 8>), dim3((unsigned)nblocks), dim3(512), sp_lds_bytes(8, 4), stream, p);
  }
  return osd_check_launch("conv_igemm_sp");
}
