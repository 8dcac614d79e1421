// conv_wgrad — weight gradient of an NHWC convolution on MFMA (gfx950):
//     dW[co][r][s][ci] += sum_m dY[m][co] * X[m @ tap(r,s)][ci]          m = (n, ho, wo)
// GEMM view: rows = output channels, columns = (tap, input channel), reduction over PIXELS.  Both operands are stored
// pixel-major (NHWC), i.e. the reduction index is the slow index, so the MFMA fragments need a transpose:
//   * bf16: tiles are staged as [32 pixels][128 channels] (256-byte rows) by LDS-DMA and read with
//     ds_read_b64_tr_b16 (hardware transpose): a lane receives 4 consecutive pixels of one channel per read.  The
//     k slots of v_mfma_f32_16x16x32_bf16 are mapped to tile rows (4g+j | 16+4g+j-4) for both operands, so one
//     32-lane half reads 8 consecutive rows; with the 16-byte-chunk swizzle chunk ^= 2*(row&7) (applied on the DMA
//     source, the destination is lane-linear) every transposed read is bank-conflict free;
//   * fp32: tiles [32 pixels][64 channels] (256-byte rows), v_mfma_f32_16x16x4_f32 takes ONE float per lane so the
//     fragments are plain ds_read_b32 of [pixel][channel]; swizzle chunk ^= 4*(row&3).
// Pixels are split over `splits` workgroups per output tile (the reduction is 1e5..1e6 long while there are only
// tens of output tiles); partial tiles are accumulated with fp32 atomics (no-return global_atomic_add_f32).
// A 3-deep LDS ring with counted vmcnt and one raw barrier per stage, as in conv_igemm_dma.hip.
#include "wgrad_params.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned g_wzero[64];

__device__ __forceinline__ void wg_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// Output tile = (WCO x WCI) sub-tiles of TWS channels (TWS = one 256-byte row: 128 bf16 / 64 fp32): 1 x 1 on 4 waves
// (2 x 2, 64 x 64 channels each) or 2 x 2 on 8 waves (2 x 4: 128 co x 64 ci each — half the operand bytes per MFMA and
// twice the MFMAs per barrier).  LDS stage = [dY sub-tiles | X sub-tiles], each [BKP pixels][256 B].
template <typename T, int BKP, int NST, int WCO, int WCI, int WM, int WN, bool ILV = false>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN == 4 && WCO + WCI == 3) ? 2 : 1) conv_wgrad_kernel(WgradParams gp) {
  constexpr int NW = WM * WN;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int TWS = 256 / (int)sizeof(T);   // channels per sub-tile row (256 bytes)
  constexpr int TCO = WCO * TWS, TCI = WCI * TWS;
  constexpr int OPB = BKP * 256;              // bytes of one operand sub-tile (BKP pixels per stage)
  constexpr int STAGE = (WCO + WCI) * OPB;
  constexpr int PPS = BKP / 4;                // 1 KiB DMA pieces (4 rows) per sub-tile
  constexpr int IPA = WCO * PPS / NW;         // DMA instructions per wave per stage, dY
  constexpr int IPB = WCI * PPS / NW;         // ... X
  static_assert((WCO * PPS) % NW == 0 && (WCI * PPS) % NW == 0, "DMA pieces must divide over the waves");
  constexpr int LPS = IPA + IPB;
  constexpr int TA = TCO / WM / 16;           // 16-wide MFMA tiles per wave along co
  constexpr int TB = TCI / WN / 16;           // ... along ci
  static_assert((TCO / WM) % 16 == 0 && (TCI / WN) % 16 == 0, "wave tiling");
  static_assert(TWS % (TCO / WM) == 0 || (TCO / WM) == TWS, "a wave's co range must stay inside one sub-tile");
  static_assert(TWS % (TCI / WN) == 0 || (TCI / WN) == TWS, "a wave's ci range must stay inside one sub-tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective remap (blocks b, b+8 share an XCD): all output tiles of one pixel split — they re-read the same
  // dY / X rows — become consecutive logical ids and therefore land on ONE XCD's L2 instead of being fetched by all 8
  int bid;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < kMaxSeg; ++i)
    if (i < gp.n_seg && bid >= gp.seg[i].block_begin) sidx = i;
  // flatten the chosen segment into the single-problem view the rest of the kernel uses.  The entry is read straight
  // from the kernarg segment (scalar loads at a dynamic offset): indexing the by-value struct with a runtime index would
  // make hipcc copy the whole table to scratch
  struct {
    const void* x; const void* dy; float* dw; const float* scale; float* db;
    int H, W, Cin, Ho, Wo, Cout, HoWo, R, S, sh, sw, ph, pw, dy_stride, M, tilesCo, tilesCi, rows_per_split, Ktot, owner;
  } p;
  typedef const __attribute__((address_space(4))) char* kptr;
  typedef unsigned long long u64;
  kptr sb = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WgradParams, seg) + sidx * (int)sizeof(WgradSeg);
#define OSD_WSEG(type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(sb + offsetof(WgradSeg, field)))
  p.x = (const void*)OSD_WSEG(u64, x); p.dy = (const void*)OSD_WSEG(u64, dy);
  p.dw = (float*)OSD_WSEG(u64, dw); p.scale = (const float*)OSD_WSEG(u64, scale); p.db = (float*)OSD_WSEG(u64, db);
  p.H = OSD_WSEG(int, H); p.W = OSD_WSEG(int, W); p.Ho = OSD_WSEG(int, Ho); p.Wo = OSD_WSEG(int, Wo); p.HoWo = p.Ho * p.Wo;
  p.M = OSD_WSEG(int, M); p.rows_per_split = OSD_WSEG(int, rows_per_split);
  p.Cin = OSD_WSEG(int, Cin); p.Cout = OSD_WSEG(int, Cout); p.R = OSD_WSEG(int, R); p.S = OSD_WSEG(int, S);
  p.sh = OSD_WSEG(int, sh); p.sw = OSD_WSEG(int, sw); p.ph = OSD_WSEG(int, ph); p.pw = OSD_WSEG(int, pw);
  p.dy_stride = OSD_WSEG(int, dy_stride); p.tilesCo = OSD_WSEG(int, tilesCo); p.tilesCi = OSD_WSEG(int, tilesCi);
  p.Ktot = OSD_WSEG(int, Ktot); p.owner = OSD_WSEG(int, owner);
  const int slot_id = bid;            // logical id over the whole launch: the partial tile's slot in ordered mode
  bid -= OSD_WSEG(int, block_begin);
#undef OSD_WSEG
  const int co_tile = bid % p.tilesCo;
  bid /= p.tilesCo;
  const int ntile = p.R * p.S * p.tilesCi;
  const int nt = bid % ntile;
  const int split = bid / ntile;
  const int tap = nt / p.tilesCi, ci_tile = nt % p.tilesCi;
  const int fr = tap / p.S, fs = tap % p.S;
  const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
  const int p_lo = split * p.rows_per_split;
  const int p_hi = min(p.M, p_lo + p.rows_per_split);
  if (p_lo >= p_hi) return;
  const int KT = (p_hi - p_lo + BKP - 1) / BKP;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dyg = reinterpret_cast<const T*>(p.dy);
  const T* zero = reinterpret_cast<const T*>(g_wzero) + (lane & 15) * EPC;

  // ---- per-lane DMA coordinates: piece = wave * IP + i -> (sub-tile, 4-row group); 4 tile rows per instruction ----
  const int lrow = lane >> 4, lpos = lane & 15;
  int a_row[IPA], a_col[IPA], a_dst[IPA];    // dY: tile row, channel (swizzled source chunk), LDS byte offset in the stage
  int b_row[IPB], b_col[IPB], b_dst[IPB];    // X
  int b_n[IPB], b_ho[IPB], b_wo[IPB];        // output-pixel coordinates of the lane's row (advanced by BKP per stage)
  bool a_cok[IPA], b_cok[IPB];
#pragma unroll
  for (int i = 0; i < IPA; ++i) {
    const int piece = wave * IPA + i, sub = piece / PPS, row = (piece % PPS) * 4 + lrow;
    a_row[i] = row;
    a_col[i] = co0 + sub * TWS + (lpos ^ wg_swz<T>(row)) * EPC;
    a_cok[i] = a_col[i] < p.Cout;
    a_dst[i] = sub * OPB + (piece % PPS) * 1024;
  }
#pragma unroll
  for (int i = 0; i < IPB; ++i) {
    const int piece = wave * IPB + i, sub = piece / PPS, row = (piece % PPS) * 4 + lrow;
    b_row[i] = row;
    b_col[i] = ci0 + sub * TWS + (lpos ^ wg_swz<T>(row)) * EPC;
    b_cok[i] = b_col[i] < p.Cin;
    b_dst[i] = (WCO + sub) * OPB + (piece % PPS) * 1024;
    const int m = p_lo + row;
    const int n_img = m / p.HoWo;
    const int rem = m - n_img * p.HoWo;
    b_n[i] = n_img; b_ho[i] = rem / p.Wo; b_wo[i] = rem - (rem / p.Wo) * p.Wo;
  }

  // a lane's pixel row advances by BKP output pixels per stage: (n, ho, wo) += (dn, dho, dwo) with at most one carry per
  // digit — uniform increments and selects instead of per-lane wrap loops (those cost ~12 divergent branches per stage)
  const int dwo = BKP % p.Wo, q_rows = BKP / p.Wo;
  const int dho = q_rows % p.Ho, dn = q_rows / p.Ho;
  const T* a_ptr[IPA];                       // dY source of the lane's row in the NEXT stage to issue
  const size_t a_step = (size_t)BKP * p.dy_stride;
#pragma unroll
  for (int i = 0; i < IPA; ++i) a_ptr[i] = dyg + (size_t)(p_lo + a_row[i]) * p.dy_stride + a_col[i];
  int stage_m = p_lo;   // first pixel of the NEXT stage to issue
  auto issue_a = [&](int i, unsigned st) {
    const bool ok = (stage_m + a_row[i] < p_hi) && a_cok[i];
#ifdef OSD_WG_CHEAP_ADDR        // diagnostic: every piece reads the zero page (no address arithmetic, always cached)
    wg_dma16(zero, st + a_dst[i]);
#else
    wg_dma16(ok ? a_ptr[i] : zero, st + a_dst[i]);
#endif
#ifndef OSD_WG_SAME_ADDR        // diagnostic: every stage re-reads the first stage's rows (full address arithmetic, cached data)
    a_ptr[i] += a_step;
#endif
  };
  auto issue_b = [&](int i, unsigned st) {
    const int hi = b_ho[i] * p.sh - p.ph + fr, wi = b_wo[i] * p.sw - p.pw + fs;
    const bool ok = (stage_m + b_row[i] < p_hi) && b_cok[i] && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
    const int off = ((b_n[i] * p.H + hi) * p.W + wi) * p.Cin + b_col[i];      // < 2^31 elements (checked by the host)
#ifdef OSD_WG_CHEAP_ADDR
    wg_dma16(zero, st + b_dst[i]);
#else
    wg_dma16(ok ? xg + off : zero, st + b_dst[i]);
#endif
    int wo = b_wo[i] + dwo, ho = b_ho[i] + dho;
    const bool c1 = wo >= p.Wo;
    wo -= c1 ? p.Wo : 0;
    ho += c1 ? 1 : 0;
    const bool c2 = ho >= p.Ho;
    ho -= c2 ? p.Ho : 0;
#ifndef OSD_WG_SAME_ADDR
    b_wo[i] = wo; b_ho[i] = ho; b_n[i] += dn + (c2 ? 1 : 0);
#else
    asm volatile("" ::"v"(wo), "v"(ho), "v"(c2 ? 1 : 0));
#endif
  };
  auto issue_stage = [&](int buf) {
    const unsigned st = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < IPA; ++i) issue_a(i, st);
#pragma unroll
    for (int i = 0; i < IPB; ++i) issue_b(i, st);
    stage_m += BKP;
  };

  f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the wave's first channel inside the block tile -> (sub-tile, column inside the sub-tile)
  constexpr int WCOL_A = TCO / WM, WCOL_B = TCI / WN;
  const int a_sub = (wm * WCOL_A) / TWS, a_c0 = (wm * WCOL_A) % TWS;
  const int b_sub = (wn * WCOL_B) / TWS, b_c0 = (wn * WCOL_B) % TWS;

  // ILV: the next stage's DMA pieces are issued BETWEEN the rows of MFMAs (one piece every few rows) instead of in a burst
  // after the barrier: a piece costs the issuing wave 60-180 cycles during which its MFMA stream stands still; spread
  // out, the other wave of the SIMD is (mostly) in an MFMA row at that moment instead of in its own burst
  auto compute_stage = [&](int buf, int nbuf, bool issue) {
    const unsigned nst = lds0 + nbuf * STAGE;
    const char* sa = smem + buf * STAGE + a_sub * OPB;
    const char* sb = smem + buf * STAGE + (WCO + b_sub) * OPB;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
     for (int k32 = 0; k32 < BKP / 32; ++k32) {
      // one 32-deep k step; k slot (g, j) <-> tile row (j < 4 ? 4g + j : 16 + 4g + j - 4)
      const int g = lane >> 4, t = lane & 15, q = t >> 2, pp = t & 3;
      const int r1 = k32 * 32 + 4 * g + q, r2 = k32 * 32 + 16 + 4 * g + q;
      typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
      typedef __attribute__((address_space(3))) bf4* lds_bf4_ptr;
      bf16x8 af[TA], bfr[TB];
      const int h8 = (pp & 1) * 8;
#pragma unroll
      for (int i = 0; i < TA; ++i) {
        const int chunk_a = ((a_c0 + i * 16) >> 3) + (pp >> 1);
        const bf4 a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf4_ptr)(sa + r1 * 256 + ((chunk_a ^ wg_swz<T>(r1)) << 4) + h8));
        const bf4 a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf4_ptr)(sa + r2 * 256 + ((chunk_a ^ wg_swz<T>(r2)) << 4) + h8));
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[i][e] = a_lo[e]; af[i][e + 4] = a_hi[e]; }
      }
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int chunk_b = ((b_c0 + j * 16) >> 3) + (pp >> 1);
        const bf4 b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf4_ptr)(sb + r1 * 256 + ((chunk_b ^ wg_swz<T>(r1)) << 4) + h8));
        const bf4 b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf4_ptr)(sb + r2 * 256 + ((chunk_b ^ wg_swz<T>(r2)) << 4) + h8));
#pragma unroll
        for (int e = 0; e < 4; ++e) { bfr[j][e] = b_lo[e]; bfr[j][e + 4] = b_hi[e]; }
      }
#pragma unroll
      for (int i = 0; i < TA; ++i) {
#pragma unroll
        for (int j = 0; j < TB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if constexpr (ILV) {
          constexpr int SLOTS = (BKP / 32) * TA;
          const int slot = k32 * TA + i;                         // compile-time after unrolling
          const int p0 = slot * LPS / SLOTS, p1 = (slot + 1) * LPS / SLOTS;
#pragma unroll
          for (int pc = 0; pc < LPS; ++pc)
            if (pc >= p0 && pc < p1 && issue) {
              __builtin_amdgcn_sched_barrier(0);
              if (pc < IPA) issue_a(pc, nst);
              else issue_b(pc - IPA, nst);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
     }
      if constexpr (ILV) { if (issue) stage_m += BKP; }
    } else {
      const int k = lane >> 4, e16 = lane & 15;
#pragma unroll
      for (int kb = 0; kb < BKP / 4; ++kb) {
        const int row = kb * 4 + k;
        const int sw = wg_swz<T>(row);
        float af[TA], bfr[TB];
#pragma unroll
        for (int i = 0; i < TA; ++i) {
          const int col_a = a_c0 + i * 16 + e16;
          af[i] = *reinterpret_cast<const float*>(sa + row * 256 + (((col_a >> 2) ^ sw) << 4) + (col_a & 3) * 4);
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const int col_b = b_c0 + j * 16 + e16;
          bfr[j] = *reinterpret_cast<const float*>(sb + row * 256 + (((col_b >> 2) ^ sw) << 4) + (col_b & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
          for (int j = 0; j < TB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // bias gradient: the blocks that own (tap 0, ci tile 0) also sum their dY tile over its pixels, one column per thread
  const bool do_bias = (p.db != nullptr) && (nt == 0) && (tid < TCO);
  float bsum = 0.f;
  auto bias_stage = [&](int buf) {
    const int sub = tid / TWS, cc = tid % TWS;
    const char* sa = smem + buf * STAGE + sub * OPB;
    const int chunk = cc / EPC, within = cc % EPC;
#pragma unroll 8
    for (int row = 0; row < BKP; ++row)
      bsum += to_f32(*reinterpret_cast<const T*>(sa + row * 256 + ((chunk ^ wg_swz<T>(row)) << 4) + within * (int)sizeof(T)));
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) issue_stage(s);
  int cur = 0, nxt = NST - 1;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + NST - 2 < KT) wg_wait_vmcnt<LPS * (NST - 2)>();
    else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const bool more = kt + NST - 1 < KT;
#ifndef OSD_WG_NO_DMA            // diagnostic builds: timing of the loop without one of its parts (results are garbage)
    if (!ILV && more) issue_stage(nxt);
#endif
    if (do_bias) bias_stage(cur);
    compute_stage(cur, nxt, more);
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }

  if (gp.partials != nullptr) {
    // ordered mode: the raw partial tile (and the bias partial of the tap-0 / ci-tile-0 workgroups) goes to this
    // workgroup's slot; channels past Cout / Cin hold zeros (their operands were the zero page)
    float* __restrict__ slot = gp.partials + (size_t)slot_id * (TCO * TCI + TCO);
    float* __restrict__ base = slot + (wm * WCOL_A + (lane >> 4) * 4) * TCI + wn * WCOL_B + (lane & 15);
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TB; ++j) base[(i * 16 + e) * TCI + j * 16] = acc[i][j][e];
    if (do_bias) slot[TCO * TCI + tid] = bsum;
    return;
  }
  if (do_bias && co0 + tid < p.Cout) wg_atomic_add(p.db + co0 + tid, bsum);
#ifdef OSD_WG_NO_ATOMICS
  if (acc[0][0][0] != 12345.678f) return;
#endif
  // ---- accumulate the partial tile into dW (fp32 atomics; rows = co, 16 consecutive ci per 16 lanes) ----
  // Whole-tile fast path: the per-row FrozenBN scales are loaded up front and the 64 atomics of a wave follow in ONE
  // basic block.  With a load or a bounds branch between two atomics hipcc puts `s_waitcnt vmcnt(0)` in front of every
  // atomic (vmcnt counts the atomics too), which serialises them at ~300 ns each: a 20 us tail on EVERY workgroup and
  // the whole duration of the small launches.
  float* __restrict__ dw = p.dw;
  const int co_w = co0 + wm * WCOL_A + (lane >> 4) * 4;           // first co row of this lane (tile i adds 16 * i)
  const int ci_w = ci0 + wn * WCOL_B + (lane & 15);               // first ci of this lane (tile j adds 16 * j)
  if (co0 + TCO <= p.Cout && ci0 + TCI <= p.Cin) {
    float scv[TA][4];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) scv[i][e] = p.scale ? p.scale[co_w + i * 16 + e] : 1.f;
    float* base = dw + (size_t)co_w * p.Ktot + tap * p.Cin + ci_w;
    if (p.owner != 0) {                                   // the only writer of this tile in this launch: plain load + store
      wg_owner_add<TA, TB>(base, p.Ktot, acc, scv);
      return;
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* row = base + (size_t)(i * 16 + e) * p.Ktot;
#pragma unroll
        for (int j = 0; j < TB; ++j) wg_atomic_add(row + j * 16, acc[i][j][e] * scv[i][e]);
      }
    return;
  }
  // ragged tile (channel counts that are not multiples of the tile): per-element bounds checks
#pragma unroll
  for (int i = 0; i < TA; ++i) {
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int ci = ci_w + j * 16;
      if (ci >= p.Cin) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co_w + i * 16 + e;
        if (co < p.Cout) {
          const float sc = p.scale ? p.scale[co] : 1.f;
          wg_atomic_add(dw + (size_t)co * p.Ktot + tap * p.Cin + ci, acc[i][j][e] * sc);
        }
      }
    }
  }
}

// ---- ordered mode, second pass: dW += scale * sum of the partial tiles, in a fixed order ----
// grid (output tiles of all segments, TCO / 16): a workgroup owns 16 rows of one output tile of one LEADER segment (the
// first segment that names a dW; the FPN levels of a conv repeat the pointer) and adds, element by element, the slots of
// that segment's splits, then of the next segment with the same dW, ... — always in that order.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgradParams gp, int TCO, int TCI) {
  typedef const __attribute__((address_space(4))) char* kptr;
  kptr kb = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WgradParams, seg);
#define OSD_RSEG(idx, type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(kb + (idx) * (int)sizeof(WgradSeg) + offsetof(WgradSeg, field)))
  typedef unsigned long long u64;
  int t = blockIdx.x, s = 0, tiles = 0;
  for (; s < gp.n_seg; ++s) {
    tiles = OSD_RSEG(s, int, tilesCo) * OSD_RSEG(s, int, tilesCi) * OSD_RSEG(s, int, R) * OSD_RSEG(s, int, S);
    if (t < tiles) break;
    t -= tiles;
  }
  if (s >= gp.n_seg) return;
  const u64 dwp = OSD_RSEG(s, u64, dw);
  for (int q = 0; q < s; ++q)
    if (OSD_RSEG(q, u64, dw) == dwp) return;            // not the leader of its dW
  float* __restrict__ dw = (float*)dwp;
  const float* __restrict__ scale = (const float*)OSD_RSEG(s, u64, scale);
  float* __restrict__ db = (float*)OSD_RSEG(s, u64, db);
  const int Cout = OSD_RSEG(s, int, Cout), Cin = OSD_RSEG(s, int, Cin), Ktot = OSD_RSEG(s, int, Ktot);
  const int tilesCo = OSD_RSEG(s, int, tilesCo), tilesCi = OSD_RSEG(s, int, tilesCi);
  const int co_tile = t % tilesCo, nt = t / tilesCo;
  const int tap = nt / tilesCi, ci_tile = nt % tilesCi;
  const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
  const size_t slot_floats = (size_t)TCO * TCI + TCO;
  const int r0 = blockIdx.y * 16;
  for (int idx = threadIdx.x; idx < 16 * TCI; idx += 256) {
    const int r = r0 + idx / TCI, c = idx % TCI;
    float sum = 0.f;
    for (int m = s; m < gp.n_seg; ++m) {
      if (OSD_RSEG(m, u64, dw) != dwp) continue;
      const int begin = OSD_RSEG(m, int, block_begin);
      const int end = m + 1 < gp.n_seg ? OSD_RSEG(m + 1, int, block_begin) : gp.n_blocks;
      const int splits = (end - begin) / tiles;
      const float* __restrict__ src = gp.partials + (size_t)(begin + t) * slot_floats + (size_t)r * TCI + c;
      const size_t step = (size_t)tiles * slot_floats;
      int sp = 0;
      for (; sp + 4 <= splits; sp += 4) {                // four independent loads in flight; the ADD order stays fixed
        const float a0 = src[(size_t)sp * step], a1 = src[(size_t)(sp + 1) * step];
        const float a2 = src[(size_t)(sp + 2) * step], a3 = src[(size_t)(sp + 3) * step];
        sum = (((sum + a0) + a1) + a2) + a3;
      }
      for (; sp < splits; ++sp) sum += src[(size_t)sp * step];
    }
    const int co = co0 + r, ci = ci0 + c;
    if (co < Cout && ci < Cin) dw[(size_t)co * Ktot + tap * Cin + ci] += sum * (scale ? scale[co] : 1.f);
  }
  if (db != nullptr && nt == 0 && blockIdx.y == 0 && (int)threadIdx.x < TCO && co0 + (int)threadIdx.x < Cout) {
    float sum = 0.f;
    for (int m = s; m < gp.n_seg; ++m) {
      if (OSD_RSEG(m, u64, dw) != dwp) continue;
      const int begin = OSD_RSEG(m, int, block_begin);
      const int end = m + 1 < gp.n_seg ? OSD_RSEG(m + 1, int, block_begin) : gp.n_blocks;
      const int splits = (end - begin) / tiles;
      for (int sp = 0; sp < splits; ++sp)
        sum += gp.partials[(size_t)(begin + co_tile + (size_t)sp * tiles) * slot_floats + (size_t)TCO * TCI + threadIdx.x];
    }
    db[co0 + threadIdx.x] += sum;
  }
#undef OSD_RSEG
}

// ---- 3x3 / stride 1 / pad 1: one workgroup per FILTER ROW (three taps), bf16 ----
// The kernel above gives every tap its own workgroup, so nine workgroups fetch the same dY tile and nine one-pixel-shifted
// copies of the same X rows; diagnostic builds put that operand traffic (L2 -> LDS by DMA, 7-12 TB/s chip-wide) at 26-39 % of
// the tower launch.  Here a stage is a run of SP = min(BKP, Wo) output pixels of ONE image row: the dY tile [BKP][128 co]
// and ONE X tile [BKP + 4][128 ci] holding input row ho - 1 + fr from column wo0 - 1 on; tap s of the filter row pairs dY
// row r with X row r + s (the k slot <-> tile row mapping of the transposed reads is free, and a row offset keeps both the
// 16-byte alignment and the 8-row bank pattern), so three taps' MFMAs run on one dY fragment set: 17 (33) KB per 3.1
// (6.3) MFLOP instead of 16 KB per 1.05.  The image border is the DMA's zero page as before; rows narrower than BKP
// (P6, P7) are padded with zero rows (1.5 % of the tower's pixels).  128 x 128 channels x 3 taps on 8 waves (2 x 4: 64 co x
// 32 ci per wave and tap, 96 accumulator registers); requires Cout, Cin multiples of 128 and Wo a multiple of BKP or
// a divisor of it.
template <int BKP, int NST>
__global__ void __launch_bounds__(512) conv_wgrad_xr_kernel(WgradParams gp) {
  typedef __bf16 T;
  constexpr int NW = 8, WN = 4;
  constexpr int XR = BKP + 4;                      // X tile rows (BKP + 2 needed; whole 4-row DMA pieces)
  constexpr int AP = BKP / 4, BP = XR / 4;         // 1 KiB DMA pieces per stage: dY, X
  constexpr int IPA = (AP + NW - 1) / NW, IPB = (BP + NW - 1) / NW;
  constexpr int LPS = IPA + IPB;                   // DMA instructions per stage of wave 0; the other waves issue one less
  static_assert(AP % NW == 0 && BP % NW == 1, "the one surplus X piece belongs to wave 0");
  constexpr int STAGE = (BKP + XR) * 256;
  constexpr int TA = 4, TB = 2;                    // 16-wide MFMA tiles per wave: 64 co x 32 ci (per tap)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int bid;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int sidx = 0;
#pragma unroll
  for (int i = 1; i < kMaxSeg; ++i)
    if (i < gp.n_seg && bid >= gp.seg[i].block_begin) sidx = i;
  typedef const __attribute__((address_space(4))) char* kptr;
  typedef unsigned long long u64;
  kptr sb = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WgradParams, seg) + sidx * (int)sizeof(WgradSeg);
#define OSD_WSEG(type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(sb + offsetof(WgradSeg, field)))
  const T* __restrict__ xg = (const T*)OSD_WSEG(u64, x);
  const T* __restrict__ dyg = (const T*)OSD_WSEG(u64, dy);
  float* __restrict__ dw = (float*)OSD_WSEG(u64, dw);
  const float* __restrict__ scale = (const float*)OSD_WSEG(u64, scale);
  float* __restrict__ db = (float*)OSD_WSEG(u64, db);
  const int H = OSD_WSEG(int, H), W = OSD_WSEG(int, W), M = OSD_WSEG(int, M), Cin = OSD_WSEG(int, Cin);
  const int rows_per_split = OSD_WSEG(int, rows_per_split), dy_stride = OSD_WSEG(int, dy_stride);
  const int tilesCo = OSD_WSEG(int, tilesCo), tilesCi = OSD_WSEG(int, tilesCi), Ktot = OSD_WSEG(int, Ktot);
  bid -= OSD_WSEG(int, block_begin);
#undef OSD_WSEG
  const int SP = W < BKP ? W : BKP;                // valid pixels per stage (stride 1, pad 1: Ho == H, Wo == W)
  const int co_tile = bid % tilesCo;
  bid /= tilesCo;
  const int ntile = 3 * tilesCi;
  const int nt = bid % ntile, split = bid / ntile;
  const int fr = nt / tilesCi, ci_tile = nt % tilesCi;
  const int co0 = co_tile * 128, ci0 = ci_tile * 128;
  const int p_lo = split * rows_per_split, p_hi = min(M, p_lo + rows_per_split);
  if (p_lo >= p_hi) return;
  const int KT = (p_hi - p_lo) / SP;               // M and rows_per_split are multiples of SP
  const T* zero = reinterpret_cast<const T*>(g_wzero) + (lane & 15) * 8;

  // ---- DMA coordinates.  Piece q = wave + 8 i covers tile rows 4 q .. 4 q + 3; lane -> (row, 16-byte chunk) ----
  const int lrow = lane >> 4, lpos = lane & 15;
  const T* a_ptr[IPA];
  bool a_ok[IPA];
  unsigned a_dst[IPA];
#pragma unroll
  for (int i = 0; i < IPA; ++i) {
    const int q = wave + NW * i, row = q * 4 + lrow;
    a_ok[i] = row < SP;
    a_ptr[i] = dyg + (size_t)(p_lo + row) * dy_stride + co0 + (lpos ^ wg_swz<T>(row)) * 8;
    a_dst[i] = q * 1024;
  }
  const size_t a_step = (size_t)SP * dy_stride;
  int b_j[IPB], b_col[IPB];
  bool b_real[IPB];
  unsigned b_dst[IPB];
#pragma unroll
  for (int i = 0; i < IPB; ++i) {
    const int q = wave + NW * i, row = q * 4 + lrow;
    b_real[i] = q < BP;
    b_j[i] = row;
    b_col[i] = ci0 + (lpos ^ wg_swz<T>(row)) * 8;
    b_dst[i] = BKP * 256 + q * 1024;
  }
  // the stage's position (uniform over the workgroup): image, output row, first output column
  int s_n, s_ho, s_wo;
  {
    const int HW = H * W;
    s_n = p_lo / HW;
    const int rem = p_lo - s_n * HW;
    s_ho = rem / W;
    s_wo = rem - s_ho * W;
  }
  auto issue_stage = [&](int buf) {
    const unsigned st = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < IPA; ++i) {
#ifdef OSD_WG_CHEAP_ADDR
      wg_dma16(zero, st + a_dst[i]);
#else
      wg_dma16(a_ok[i] ? a_ptr[i] : zero, st + a_dst[i]);
#endif
      a_ptr[i] += a_step;
    }
    const int hi = s_ho - 1 + fr;
    const bool row_ok = (unsigned)hi < (unsigned)H;
    const int base = ((s_n * H + hi) * W + s_wo - 1) * Cin;      // element offset of X tile row 0 (may point before the row)
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      const int wi = s_wo - 1 + b_j[i];
      const bool ok = row_ok && b_j[i] < SP + 2 && (unsigned)wi < (unsigned)W;
#ifdef OSD_WG_CHEAP_ADDR
      if (b_real[i]) wg_dma16(zero, st + b_dst[i]);
#else
      if (b_real[i]) wg_dma16(ok ? xg + (base + b_j[i] * Cin + b_col[i]) : zero, st + b_dst[i]);      // wave-uniform branch
#endif
    }
    s_wo += SP;
    if (s_wo >= W) { s_wo = 0; if (++s_ho >= H) { s_ho = 0; ++s_n; } }
  };

  f32x4 acc[3][TA][TB];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[s][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_c0 = wm * 64, b_c0 = wn * 32;

  auto compute_stage = [&](int buf) {
    const char* sa = smem + buf * STAGE;
    const char* sx = sa + BKP * 256;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
    typedef __attribute__((address_space(3))) bf4* lds_bf4_ptr;
    const int g = lane >> 4, t = lane & 15, q = t >> 2, pp = t & 3;
    const int h8 = (pp & 1) * 8;
#pragma unroll
    for (int k32 = 0; k32 < BKP / 32; ++k32) {
      const int r1 = k32 * 32 + 4 * g + q, r2 = r1 + 16;
      bf16x8 af[TA];
#pragma unroll
      for (int i = 0; i < TA; ++i) {
        const int chunk = ((a_c0 + i * 16) >> 3) + (pp >> 1);
        const bf4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(sa + r1 * 256 + ((chunk ^ wg_swz<T>(r1)) << 4) + h8));
        const bf4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(sa + r2 * 256 + ((chunk ^ wg_swz<T>(r2)) << 4) + h8));
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[i][e] = lo[e]; af[i][e + 4] = hi[e]; }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        bf16x8 bfr[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const int chunk = ((b_c0 + j * 16) >> 3) + (pp >> 1);
          const int x1 = r1 + s, x2 = r2 + s;
          const bf4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(sx + x1 * 256 + ((chunk ^ wg_swz<T>(x1)) << 4) + h8));
          const bf4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(sx + x2 * 256 + ((chunk ^ wg_swz<T>(x2)) << 4) + h8));
#pragma unroll
          for (int e = 0; e < 4; ++e) { bfr[j][e] = lo[e]; bfr[j][e + 4] = hi[e]; }
        }
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
          for (int j = 0; j < TB; ++j)
            acc[s][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[s][i][j], 0, 0, 0);
      }
    }
  };

  const bool do_bias = (db != nullptr) && (nt == 0) && (tid < 128);
  float bsum = 0.f;
  auto bias_stage = [&](int buf) {
    const char* sa = smem + buf * STAGE;
    const int chunk = tid / 8, within = tid % 8;
#pragma unroll 8
    for (int row = 0; row < BKP; ++row)
      bsum += to_f32(*reinterpret_cast<const T*>(sa + row * 256 + ((chunk ^ wg_swz<T>(row)) << 4) + within * 2));
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) issue_stage(s);
  int cur = 0, nxt = NST - 1;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + NST - 2 >= KT) wg_wait_vmcnt<0>();
    else if (wave == 0) wg_wait_vmcnt<LPS * (NST - 2)>();
    else wg_wait_vmcnt<(LPS - 1) * (NST - 2)>();
    __builtin_amdgcn_s_barrier();
#ifndef OSD_WG_NO_DMA            // diagnostic builds (results are garbage)
    if (kt + NST - 1 < KT) issue_stage(nxt);
#endif
    if (do_bias) bias_stage(cur);
    compute_stage(cur);
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }

  if (do_bias) wg_atomic_add(db + co0 + tid, bsum);
#ifdef OSD_WG_NO_ATOMICS
  if (acc[0][0][0][0] != 12345.678f) return;
#endif
  // ---- three partial tiles (one per tap of the filter row) into dW; whole tiles by construction ----
  const int co_w = co0 + wm * 64 + (lane >> 4) * 4;
  const int ci_w = ci0 + wn * 32 + (lane & 15);
  float scv[TA][4];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) scv[i][e] = scale ? scale[co_w + i * 16 + e] : 1.f;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float* base = dw + (size_t)co_w * Ktot + (fr * 3 + s) * Cin + ci_w;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* row = base + (size_t)(i * 16 + e) * Ktot;
#pragma unroll
        for (int j = 0; j < TB; ++j) wg_atomic_add(row + j * 16, acc[s][i][j][e] * scv[i][e]);
      }
  }
}

// ---- skinny weight gradient for the prediction convs (Cout <= 8): dW[co][tap][ci] with one thread per ci ----
template <typename T>
__global__ void __launch_bounds__(256) conv_wgrad_skinny_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                float* __restrict__ dw, int H, int W, int Cin, int Ho,
                                                                int Wo, int Cout, int R, int S, int stride, int pad,
                                                                int dy_stride, int M, int rows_per_block) {
  const int ci = blockIdx.y * blockDim.x + threadIdx.x;
  const int p_lo = blockIdx.x * rows_per_block, p_hi = min(M, p_lo + rows_per_block);
  if (ci >= Cin || p_lo >= p_hi) return;
  float acc[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  const int HoWo = Ho * Wo;
  for (int m = p_lo; m < p_hi; ++m) {
    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / Wo, wo = rem - (rem / Wo) * Wo;
    float g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) g[c] = c < Cout ? to_f32(dy[(size_t)m * dy_stride + c]) : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t >= R * S) break;
      const int r = t / S, s = t - (t / S) * S;
      const int hi = ho * stride - pad + r, wi = wo * stride - pad + s;
      if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
      const float xv = to_f32(x[((size_t)(n * H + hi) * W + wi) * Cin + ci]);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][t] += g[c] * xv;
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c >= Cout) break;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      if (t < R * S) atomicAdd(dw + ((size_t)c * R * S + t) * Cin + ci, acc[c][t]);
  }
}

// ---- read-once weight gradient of the FCOS prediction convs (3x3 / stride 1 / pad 1, Cout <= 4: cls_logits + centerness
// fused, bbox_pred; fcos.py:50-61) over all FPN levels in one launch ----
// The MFMA kernel above spends a 128-channel output tile on 2 or 4 channels (11-19 TFLOP/s, 0.25 ms per step).  Here a wave
// owns an INPUT pixel q: it reads x[q][256 channels] once (4 channels per lane) and adds dy[q - tap][co] * x[q][c] into
// acc[co][tap][c] for the nine output pixels whose 3x3 window covers q (lanes 0..8 fetch those nine dy vectors, broadcast
// with v_readlane).  A workgroup reduces its four waves through LDS and adds its 36 x 256 partial with one atomic per value.
constexpr int kPredLevels = 6;
struct PredWgradLevels {
  const void* x[kPredLevels]; const void* dy[kPredLevels];
  int H[kPredLevels], W[kPredLevels], npix[kPredLevels], begin[kPredLevels];   // npix = n * H * W; begin = first workgroup
  int n_levels;
};

template <typename T>
__global__ void __launch_bounds__(256) pred_wgrad_kernel(PredWgradLevels L, float* __restrict__ part, float* __restrict__ part_b,
                                                         int C, int dy_stride, int pix_per_block) {
  __shared__ float red[64 * 144];
  const int b = blockIdx.x, cg = blockIdx.y;          // cg: group of 256 input channels
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < kPredLevels; ++i)
    if (i < L.n_levels && b >= L.begin[i]) lvl = i;
  const void* xv = L.x[0]; const void* dyv = L.dy[0];
  int H = L.H[0], W = L.W[0], npix = L.npix[0], beg = L.begin[0];
#pragma unroll
  for (int i = 1; i < kPredLevels; ++i)
    if (lvl == i) { xv = L.x[i]; dyv = L.dy[i]; H = L.H[i]; W = L.W[i]; npix = L.npix[i]; beg = L.begin[i]; }
  const T* __restrict__ x = reinterpret_cast<const T*>(xv);
  const T* __restrict__ dy = reinterpret_cast<const T*>(dyv);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (b - beg) * pix_per_block, q1 = min(npix, q0 + pix_per_block);
  const int c0 = cg * 256 + lane * 4;
  const int HW = H * W;
  float acc[4][9][4];
#pragma unroll
  for (int co = 0; co < 4; ++co)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[co][t][k] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int tr = lane / 3, ts = lane - tr * 3;         // the tap this lane fetches dy for (lanes 0..8)
  // U pixels per wave and iteration: all 2 x U loads are issued before the first use (one load in flight per wave made the
  // kernel latency bound: 211 us for 70 MB)
  constexpr int U = 8;
  for (int qb = q0 + wave * U; qb < q1; qb += 4 * U) {
    float xk[U][4], d[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = min(qb + u, q1 - 1);                 // clamped: a repeated pixel is masked below
      const int img = q / HW, rem = q - img * HW, qy = rem / W, qx = rem - qy * W;
      if constexpr (sizeof(T) == 2) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(x + (size_t)q * C + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) xk[u][k] = (float)v[k];
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)q * C + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) xk[u][k] = v[k];
      }
      // output pixel whose tap (tr, ts) reads input pixel q: (qy - tr + 1, qx - ts + 1)
      const int py = qy - tr + 1, px = qx - ts + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) d[u][k] = 0.f;
      if (lane < 9 && qb + u < q1 && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) {
        const T* dp = dy + ((size_t)(img * H + py) * W + px) * dy_stride;
        if constexpr (sizeof(T) == 2) {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>(dp);
#pragma unroll
          for (int k = 0; k < 4; ++k) d[u][k] = (float)v[k];
        } else {
          const f32x4 v = *reinterpret_cast<const f32x4*>(dp);
#pragma unroll
          for (int k = 0; k < 4; ++k) d[u][k] = v[k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (lane == 4) {                                   // the centre tap is the pixel itself: bias gradient (cg 0 only)
#pragma unroll
        for (int k = 0; k < 4; ++k) bsum[k] += d[u][k];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int co = 0; co < 4; ++co) {
          const float dv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d[u][co]), t));
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[co][t][k] += dv * xk[u][k];
        }
      }
    }
  }
  // fold the four waves: waves 1..3 hand their accumulators to wave 0 through LDS, one after the other
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int k = 0; k < 4; ++k) red[((co * 9 + t) * 4 + k) * 64 + lane] = acc[co][t][k];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[co][t][k] += red[((co * 9 + t) * 4 + k) * 64 + lane];
    }
    __syncthreads();
  }
  // the workgroup's partial: part[block][co][tap][C] (plain coalesced stores; 500 workgroups adding atomically into the same
  // 36 KB serialise at the memory-side atomic units: 57-114 us of a 170 us kernel), folded in fixed order by
  // pred_wgrad_reduce_kernel
  if (wave == 0) {
    float* pp = part + ((size_t)blockIdx.x * gridDim.y + cg) * 36 * 256 + lane * 4;
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t)
        *reinterpret_cast<f32x4*>(pp + (co * 9 + t) * 256) = f32x4{acc[co][t][0], acc[co][t][1], acc[co][t][2], acc[co][t][3]};
  }
  if (cg == 0) {                                         // bias partials: lane 4 of every wave -> part_b[block][wave][4]
    if (lane == 4) {
#pragma unroll
      for (int co = 0; co < 4; ++co) part_b[((size_t)blockIdx.x * 4 + wave) * 4 + co] = bsum[co];
    }
  }
}

// dw[co][tap][c] += sum over workgroups of part[block][cg][co][tap][c & 255]; db[co] += sum of part_b.  blockIdx.y = one of
// `chunks` slices of the workgroup list: each thread sums its slice with four independent loads in flight, then ONE atomic
// per (element, slice) — 16 adders per address instead of 500.
__global__ void pred_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_b, float* __restrict__ dw,
                                         float* __restrict__ db, int blocks, int groups, int C, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = cout * 9 * C;
  const int per = (blocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(blocks, b0 + per);
  if (i < n) {
    const int c = i % C, ct = i / C;                     // ct = co * 9 + tap
    const int cg = c >> 8, cl = c & 255;
    const size_t stride = (size_t)groups * 36 * 256;
    const float* p = part + ((size_t)cg * 36 + ct) * 256 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      s0 += p[(size_t)b * stride]; s1 += p[(size_t)(b + 1) * stride]; s2 += p[(size_t)(b + 2) * stride]; s3 += p[(size_t)(b + 3) * stride];
    }
    for (; b < b1; ++b) s0 += p[(size_t)b * stride];
    if (b0 < b1) atomicAdd(dw + i, (s0 + s1) + (s2 + s3));
  } else if (db != nullptr && i < n + cout) {
    const int co = i - n;
    float s = 0.f;
    for (int b = b0 * 4; b < b1 * 4; ++b) s += part_b[(size_t)b * 4 + co];
    if (b0 < b1) atomicAdd(db + co, s);
  }
}

// ---- bias gradient: db[c] += sum over pixels of dy[m][c] ----
template <typename T>
__global__ void __launch_bounds__(256) bias_grad_kernel(const T* __restrict__ dy, float* __restrict__ db, int M, int C,
                                                        int stride, int rows_per_block) {
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  const int p_lo = blockIdx.x * rows_per_block, p_hi = min(M, p_lo + rows_per_block);
  if (c >= C || p_lo >= p_hi) return;
  float s = 0.f;
  for (int m = p_lo; m < p_hi; ++m) s += to_f32(dy[(size_t)m * stride + c]);
  atomicAdd(db + c, s);
}

}  // namespace

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int osd_bias_grad(const void* dy, float* db, int m, int c, int stride, int dtype, void* stream);

struct WgradProblem {      // host-side description of one segment
  const osd_conv_desc* d;    // geometry: n, h, w, cin, cout, r, s, strides, pads, out_stride
  int n, h, w;
  const void* x; const void* dy; const float* scale; float* dw; float* db;
};

// Filter-row kernel (conv_wgrad_xr_kernel): algo = 1 + 128 + x + 16 * target_code, x: 0 = 32-pixel stages x 6, 1 = 64 x 4, 2 = 32 x 8.
static int wgrad_xr_launch(int n_seg, const WgradProblem* pr, hipStream_t s) {
  if (n_seg < 1 || n_seg > kMaxSeg) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: 1..%d segments", kMaxSeg);
  const osd_conv_desc* d0 = pr[0].d;
  static const int kTargets[8] = {512, 256, 128, 64, 1024, 768, 1536, 2048};
  const int a = d0->algo - 1 - 128;
  const int x = a & 15, target = kTargets[(a >> 4) & 7];
  if (a < 0 || a >= 128 || x > 2) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: unknown algo %d", d0->algo);
  const int bkp = x == 1 ? 64 : 32;
  WgradParams p;
  p.n_seg = n_seg;
  long long work = 0;
  for (int i = 0; i < n_seg; ++i) {
    const osd_conv_desc* d = pr[i].d;
    WgradSeg& g = p.seg[i];
    const int h = pr[i].h, w = pr[i].w;
    if (d->dtype != OSD_BF16 || d->r != 3 || d->s != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1 ||
        d->cout % 128 || d->cin % 128 || d->out_stride % 8 || !(w % bkp == 0 || bkp % w == 0))
      return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad (filter-row kernel): bf16 3x3 / 1 / 1, channels in 128s, width %d vs %d-pixel stages", w, bkp);
    const long long M = (long long)pr[i].n * h * w;
    if (M <= 0 || M > 0x7fffffffLL || !pr[i].x || !pr[i].dy || !pr[i].dw) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad segment %d", i);
    if (M * d->cin > 0x7fffffffLL) return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: input of segment %d has more than 2^31 elements", i);
    g.x = pr[i].x; g.dy = pr[i].dy; g.dw = pr[i].dw; g.scale = pr[i].scale; g.db = pr[i].db;
    g.H = h; g.W = w; g.Ho = h; g.Wo = w; g.M = (int)M;
    g.Cin = d->cin; g.Cout = d->cout; g.R = 3; g.S = 3; g.sh = g.sw = 1; g.ph = g.pw = 1;
    g.dy_stride = d->out_stride; g.Ktot = 9 * d->cin;
    g.tilesCo = d->cout / 128; g.tilesCi = d->cin / 128;
    work += M * g.tilesCo * g.tilesCi * 3;
  }
  long long rows = (work + target - 1) / target;
  rows = (rows + 63) / 64 * 64;                  // whole stages for every width (stages are min(bkp, w) pixels, all divide 64)
  if (rows < 128) rows = 128;
  long long nblocks = 0;
  for (int i = 0; i < n_seg; ++i) {
    WgradSeg& g = p.seg[i];
    int sp = (int)((g.M + rows - 1) / rows);
    if (sp < 1) sp = 1;
    g.rows_per_split = cdiv(cdiv(g.M, sp), 64) * 64;
    const int splits = cdiv(g.M, g.rows_per_split);
    g.block_begin = (int)nblocks;
    nblocks += (long long)g.tilesCo * g.tilesCi * 3 * splits;
    if (nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad grid");
  }
  // owner mode needs the segment's dW (and db: its adds stay atomic) to be named by no other segment of the launch; never in
  // ordered mode (the partial tiles go to the scratch slots) or team mode (several teams add into one dW).  OSD_WGRAD_NO_OWNER=1: A/B
  static int no_owner = -1;
  if (no_owner < 0) { const char* e = getenv("OSD_WGRAD_NO_OWNER"); no_owner = (e && atoi(e) != 0) ? 1 : 0; }
  for (int i = 0; i < n_seg; ++i) {
    if (no_owner || d0->ordered_ws != nullptr || p.sk_units > 0) p.seg[i].owner = 0;
    for (int j = 0; j < n_seg && p.seg[i].owner; ++j)
      if (j != i && p.seg[j].dw == p.seg[i].dw) p.seg[i].owner = 0;
  }
  for (int i = n_seg; i < kMaxSeg; ++i) p.seg[i] = p.seg[0];
  p.n_blocks = (int)nblocks;
  p.partials = nullptr;         // the filter-row kernel has no ordered mode
#define OSD_WGX(BK, NS)                                                                                               \
  do {                                                                                                                \
    auto kern = conv_wgrad_xr_kernel<BK, NS>;                                                                         \
    constexpr int lds = NS * (BK + BK + 4) * 256;                                                                     \
    static bool attr = false;                                                                                         \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(512), lds, s, p);                                         \
  } while (0)
  if (x == 0) OSD_WGX(32, 6);
  else if (x == 1) OSD_WGX(64, 4);
  else OSD_WGX(32, 8);
#undef OSD_WGX
  return osd_check_launch("conv_wgrad_xr");
}

// dtype and algo (0 = default, else 1 + variant + 16 * target_code) come from the first problem's descriptor.
// variant 0..2: 128 x 128 channel tile on 4 waves, pixels per stage x ring depth = 32x3 / 64x2 / 32x4 (bf16; fp32 always 32x3);
// 13, 3: the software-pipelined 256 x 256 kernel of conv_wgrad_sk.hip with this launcher's splits / in team mode; 12: retired; variants 4..7: 256-wide tiles on 8 waves; 8, 9: 256 x 256 with a 5- / 4-deep ring of 32-pixel stages
// (the whole 160 KB / 128 KB of LDS as prefetch distance: one workgroup per CU has nothing else to hide the operand
// latency behind); 10..12: 128 x 256 / 256 x 128 tiles on FOUR waves (64 x 128 per wave, 72 KB: two workgroups per CU)
// Variant ids are NOT stable across rounds (round 3 re-used 3 for team mode and 13 for the software-pipelined kernel, retired 12):
// ids pinned from outside through OSD_WGRAD_VARIANT (the untuned algo-0 default) or an old OSD_DUMP_ALGOS dump may name a kernel
// that does not cover a launch.  On the algo-0 path such a launch falls back to variant 0, which covers everything; an explicit
// `algo` keeps returning OSD_ERR_UNSUPPORTED (the tuner skips it).
static int wgrad_launch_v(int n_seg, const WgradProblem* pr, hipStream_t s, int variant_override);
static int wgrad_launch(int n_seg, const WgradProblem* pr, hipStream_t s) {
  int rc = wgrad_launch_v(n_seg, pr, s, -1);
  if (rc == OSD_ERR_UNSUPPORTED && n_seg >= 1 && pr[0].d->algo == 0) rc = wgrad_launch_v(n_seg, pr, s, 0);
  return rc;
}

static int wgrad_launch_v(int n_seg, const WgradProblem* pr, hipStream_t s, int variant_override) {
  if (n_seg < 1 || n_seg > kMaxSeg) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: 1..%d segments", kMaxSeg);
  const osd_conv_desc* d0 = pr[0].d;
  if (d0->algo > 128) return wgrad_xr_launch(n_seg, pr, s);
  static int env_target = -1, env_variant = -1;
  if (env_target < 0) { const char* e = getenv("OSD_WGRAD_BLOCKS"); env_target = e ? atoi(e) : 512; }
  if (env_variant < 0) { const char* e = getenv("OSD_WGRAD_VARIANT"); env_variant = e ? atoi(e) : 0; }
  // (3072 / 4096 were tried in place of 256 / 768: the big launches pick them and get 6-15 % faster in isolation —
  // tower 857 -> 796 us — but the training step does not: 639.6 / 642.9 vs 638.3 / 634.9 images/s in one-box A/B, the
  // extra workgroups compete with the main chain they run beside)
  static const int kTargets[8] = {512, 256, 128, 64, 1024, 768, 1536, 2048};
  int target = env_target, variant = variant_override >= 0 ? variant_override : env_variant, code = 0;
  if (d0->algo > 0) {
    const int a = d0->algo - 1;
    if (a >= 128) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: unknown algo %d", d0->algo);
    variant = a & 15;
    code = a >> 4;
    target = kTargets[code];
  }
  // exact-fp32 MFMA runs at 1/16 of the bf16 rate and its channel tile is 64 wide (4x the output tiles): a workgroup's
  // fixed costs and its atomic epilogue weigh 16x less, so the same codes mean 8x the workgroups (finer pixel splits
  // also even out the FPN levels of a multi-segment launch: P3 has 228x the pixels of P7)
  if (d0->dtype == OSD_F32) target *= 8;
  // channel tile (co x ci) in 256-byte sub-tiles: variants 4 / 5 = 2 x 2, 6 = 1 x 2, 7 = 2 x 1 (bf16, 8 waves); else 1 x 1
  const bool bf = d0->dtype == OSD_BF16;
  if (!bf && variant != 0) variant = 0;
  const int sub_co = bf && (variant == 3 || variant == 4 || variant == 5 || variant == 7 || variant == 8 || variant == 9 || variant == 11 || variant == 13 || variant == 14) ? 2 : 1;
  const int sub_ci = bf && (variant == 3 || variant == 4 || variant == 5 || variant == 6 || variant == 8 || variant == 9 || variant == 10 || variant == 13) ? 2 : 1;
  const int tw = bf ? 128 : 64;
  const int epc = d0->dtype == OSD_BF16 ? 8 : 4;
  WgradParams p;
  p.n_seg = n_seg;
  long long work = 0;        // sum over segments of pixels x output tiles
  for (int i = 0; i < n_seg; ++i) {
    const osd_conv_desc* d = pr[i].d;
    WgradSeg& g = p.seg[i];
    if (d->dtype != d0->dtype) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: segments of different dtype");
    if (d->cin % epc || d->out_stride % epc) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: channel counts must keep 16-byte alignment");
    const int ho = (pr[i].h + 2 * d->pad_h - d->r) / d->stride_h + 1, wo = (pr[i].w + 2 * d->pad_w - d->s) / d->stride_w + 1;
    const long long M = (long long)pr[i].n * ho * wo;
    if (M <= 0 || M > 0x7fffffffLL || !pr[i].x || !pr[i].dy || !pr[i].dw) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad segment %d", i);
    if ((long long)pr[i].n * pr[i].h * pr[i].w * d->cin > 0x7fffffffLL)
      return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: input of segment %d has more than 2^31 elements", i);
    g.x = pr[i].x; g.dy = pr[i].dy; g.dw = pr[i].dw; g.scale = pr[i].scale; g.db = pr[i].db;
    g.H = pr[i].h; g.W = pr[i].w; g.Ho = ho; g.Wo = wo; g.M = (int)M;
    g.Cin = d->cin; g.Cout = d->cout; g.R = d->r; g.S = d->s;
    g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w;
    g.dy_stride = d->out_stride; g.Ktot = d->r * d->s * d->cin;
    g.tilesCo = cdiv(d->cout, tw * sub_co); g.tilesCi = cdiv(d->cin, tw * sub_ci);
    work += M * g.tilesCo * g.tilesCi * d->r * d->s;
  }
  // pixel splits: enough workgroups to fill the chip, every workgroup about the same number of pixels (>= 128)
  long long rows = (work + target - 1) / target;
  rows = (rows + 31) / 32 * 32;
  if (rows < 128) rows = 128;
  long long nblocks = 0;
  for (int i = 0; i < n_seg; ++i) {
    WgradSeg& g = p.seg[i];
    int sp = (int)((g.M + rows - 1) / rows);
    if (sp < 1) sp = 1;
    // (round 6, measured and dropped: keeping a segment up to 25 % over the target at ONE split so that it owns its tiles — the
    // layer3 stage launch then runs 108 workgroups instead of its best 216: 480 vs 405 us; the tuner's target codes decide alone)
    g.rows_per_split = cdiv(cdiv(g.M, sp), 32) * 32;
    const int splits = cdiv(g.M, g.rows_per_split);
    g.owner = splits == 1 ? 1 : 0;
    g.block_begin = (int)nblocks;
    nblocks += (long long)g.tilesCo * g.tilesCi * g.R * g.S * splits;
    if (nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad grid");
  }
  p.sk_units = p.sk_teams = p.sk_total = 0;
  if (bf && variant == 3) {
    // team mode (conv_wgrad_sk.hip): all segments are instances of ONE conv shape (the FPN levels of a conv, the convs of a
    // tower), so every pixel range has the same `units` output tiles x taps; teams of `units` workgroups share the launch's
    // concatenated 64-pixel stages equally.  The target code is the number of rounds: (code + 1) workgroups per CU.
    if (d0->ordered_ws != nullptr) return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: team mode adds its partial tiles atomically (not for ordered mode)");
    if (code > 3) return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: team mode knows target codes 0..3 (rounds)");
    const WgradSeg& g0 = p.seg[0];
    long long total = 0;
    for (int i = 0; i < n_seg; ++i) {
      WgradSeg& g = p.seg[i];
      if (g.Cout != g0.Cout || g.Cin != g0.Cin || g.R != g0.R || g.S != g0.S)
        return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: team mode needs segments of one conv shape (segment %d differs)", i);
      g.stage_begin = (int)total;
      total += cdiv(g.M, 64);
    }
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
      const char* e = getenv("OSD_WGRAD_TEAM_CUS");      // A/B: teams sized for fewer CUs (leave some to the other streams' kernels)
      if (e && atoi(e) > 0) cus = atoi(e);
    }
    const int units = g0.tilesCo * g0.tilesCi * g0.R * g0.S;
    long long teams = (long long)cus * (code + 1) / units;
    if (teams < 1) return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: team mode needs at most one output tile x tap per CU (%d)", units);
    if (teams > total) teams = total;
    if (total > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad grid");
    p.sk_units = units; p.sk_teams = (int)teams; p.sk_total = (int)total;
    nblocks = teams * units;
  }
  // owner mode needs the segment's dW (and db: its adds stay atomic) to be named by no other segment of the launch; never in
  // ordered mode (the partial tiles go to the scratch slots) or team mode (several teams add into one dW).  OSD_WGRAD_NO_OWNER=1: A/B
  static int no_owner = -1;
  if (no_owner < 0) { const char* e = getenv("OSD_WGRAD_NO_OWNER"); no_owner = (e && atoi(e) != 0) ? 1 : 0; }
  for (int i = 0; i < n_seg; ++i) {
    if (no_owner || d0->ordered_ws != nullptr || p.sk_units > 0) p.seg[i].owner = 0;
    for (int j = 0; j < n_seg && p.seg[i].owner; ++j)
      if (j != i && p.seg[j].dw == p.seg[i].dw) p.seg[i].owner = 0;
  }
  for (int i = n_seg; i < kMaxSeg; ++i) p.seg[i] = p.seg[0];
  p.n_blocks = (int)nblocks;
  p.partials = nullptr;
  const int tco = tw * sub_co, tci = tw * sub_ci;
  if (d0->ordered_ws != nullptr) {     // ordered mode: the caller's scratch buffer, passed with this call
    const long long need = nblocks * ((long long)tco * tci + tco) * 4;
    if (need > (long long)d0->ordered_ws_bytes)
      return osd_fail(OSD_ERR_WORKSPACE, "wgrad (ordered mode): the launch needs %lld bytes of partial tiles, ordered_ws_bytes is %lld",
                      need, (long long)d0->ordered_ws_bytes);
    p.partials = static_cast<float*>(d0->ordered_ws);
  }
  const osd_conv_desc* d = d0;
  // variant: 0 = 32 px x 3 stages, 1 = 64 px x 2, 2 = 32 px x 4 (bf16; fp32 always 32 x 3)
#define OSD_WG_LAUNCH(TT, BK, NS, WC, WMM, WNN) OSD_WG_LAUNCH3(TT, BK, NS, WC, WC, WMM, WNN, false)
#define OSD_WG_LAUNCH2(TT, BK, NS, WCOO, WCII, WMM, WNN) OSD_WG_LAUNCH3(TT, BK, NS, WCOO, WCII, WMM, WNN, false)
#define OSD_WG_LAUNCH3(TT, BK, NS, WCOO, WCII, WMM, WNN, IL)                                                        \
  do {                                                                                                               \
    auto kern = conv_wgrad_kernel<TT, BK, NS, WCOO, WCII, WMM, WNN, IL>;                                                 \
    constexpr int lds = NS * (WCOO + WCII) * BK * 256;                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * WMM * WNN), lds, s, p);                             \
  } while (0)
  if (d->dtype == OSD_F32) {
    OSD_WG_LAUNCH(float, 32, 3, 1, 2, 2);
  } else {
    switch (variant) {
      case 1: OSD_WG_LAUNCH(__bf16, 64, 2, 1, 2, 2); break;
      case 2: OSD_WG_LAUNCH(__bf16, 32, 4, 1, 2, 2); break;
      case 4: OSD_WG_LAUNCH(__bf16, 32, 3, 2, 2, 4); break;
      // long stages on 8 waves (64 MFMAs per wave between barriers, as the forward kernel's 256 x 256 tile): 2-deep ring
      case 5: OSD_WG_LAUNCH(__bf16, 64, 2, 2, 2, 4); break;          // 256 co x 256 ci
      case 6: OSD_WG_LAUNCH2(__bf16, 64, 2, 1, 2, 2, 4); break;      // 128 co x 256 ci
      case 7: OSD_WG_LAUNCH2(__bf16, 64, 2, 2, 1, 4, 2); break;      // 256 co x 128 ci
      case 8: OSD_WG_LAUNCH(__bf16, 32, 5, 2, 2, 4); break;          // 256 x 256, 5 x 32 KB: all of the LDS
      case 9: OSD_WG_LAUNCH(__bf16, 32, 4, 2, 2, 4); break;          // 256 x 256, 4 x 32 KB
      case 10: OSD_WG_LAUNCH2(__bf16, 32, 3, 1, 2, 2, 2); break;     // 128 co x 256 ci on 4 waves, 3 x 24 KB
      case 11: OSD_WG_LAUNCH2(__bf16, 32, 3, 2, 1, 2, 2); break;     // 256 co x 128 ci on 4 waves
      // 14, 15: variants 11 and 0 with the DMA pieces issued between the MFMA rows
      case 12: return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: variant 12 is retired");
      case 3:       // conv_wgrad_sk.hip in team mode (set up above)
      case 13: {    // conv_wgrad_sk.hip, the splits of this launcher: 256 x 256 on eight waves, software-pipelined loop
        for (int i = 0; i < n_seg; ++i) {      // the index arithmetic of its DMA is exact (and fits 32-bit byte offsets) below these bounds
          const WgradSeg& g = p.seg[i];
          if (g.Wo < 2 || g.Ho < 2 || (long long)g.M * g.Wo >= 0xffffffffLL || (long long)g.M * g.dy_stride * 2 >= 0x7fffffffLL ||
              (long long)(g.M / (g.Ho * g.Wo)) * g.H * g.W * g.Cin * 2 >= 0x7fffffffLL)
            return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: the pipelined variants need output maps of at least 2 x 2 and operands below 2 GiB (segment %d)", i);
        }
        const int rc = osd_wgrad_sk_launch(p, s);
        if (rc) return rc;
        break;
      }
      case 14: OSD_WG_LAUNCH3(__bf16, 32, 3, 2, 1, 2, 2, true); break;
      case 15: OSD_WG_LAUNCH3(__bf16, 32, 3, 1, 1, 2, 2, true); break;
      default: OSD_WG_LAUNCH(__bf16, 32, 3, 1, 2, 2); break;
    }
  }
#undef OSD_WG_LAUNCH
#undef OSD_WG_LAUNCH2
#undef OSD_WG_LAUNCH3
  if (p.partials != nullptr) {
    int rc = osd_check_launch("conv_wgrad");
    if (rc) return rc;
    long long tiles = 0;
    for (int i = 0; i < n_seg; ++i) tiles += (long long)p.seg[i].tilesCo * p.seg[i].tilesCi * p.seg[i].R * p.seg[i].S;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)tiles, tco / 16), dim3(256), 0, s, p, tco, tci);
    return osd_check_launch("conv_wgrad(reduce)");
  }
  return osd_check_launch("conv_wgrad");
}

extern "C" int osd_conv2d_wgrad(const osd_conv_desc* d, const void* x, const void* dy, const float* scale, float* dw,
                                float* db, void* stream) {
  if (!d || !x || !dy || !dw) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: null argument");
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad dtype");
  const int epc = d->dtype == OSD_BF16 ? 8 : 4;
  const long long M = (long long)d->n * d->ho * d->wo;
  if (M <= 0 || M > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad M");
  if (d->in_stride_w != d->cin || d->in_stride_h != d->w * d->cin)
    return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: dense NHWC input required");
  hipStream_t s = OSD_STREAM(stream);
  if (d->cout <= 8 && (d->out_stride % epc != 0 || d->cin % epc != 0)) {   // prediction convs whose dy is not padded
    if (d->r * d->s > 9) return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad: skinny kernel supports up to 9 taps");
    const int rows = 512;
    dim3 grid(cdiv((int)M, rows), cdiv(d->cin, 256));
    if (d->dtype == OSD_F32)
      hipLaunchKernelGGL(conv_wgrad_skinny_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, dw, d->h,
                         d->w, d->cin, d->ho, d->wo, d->cout, d->r, d->s, d->stride_h, d->pad_h, d->out_stride, (int)M, rows);
    else
      hipLaunchKernelGGL(conv_wgrad_skinny_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, (const __bf16*)dy, dw,
                         d->h, d->w, d->cin, d->ho, d->wo, d->cout, d->r, d->s, d->stride_h, d->pad_h, d->out_stride, (int)M,
                         rows);
    int rc = osd_check_launch("conv_wgrad_skinny");
    if (rc || !db) return rc;
    return osd_bias_grad(dy, db, (int)M, d->cout, d->out_stride, d->dtype, stream);
  }
  const WgradProblem pr = {d, d->n, d->h, d->w, x, dy, scale, dw, db};
  return wgrad_launch(1, &pr, s);
}

// several (x, dy) pairs with the same conv geometry except batch / spatial size, sharing dW (weights shared over FPN levels)
extern "C" int osd_conv2d_wgrad_grouped(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                                        const int32_t* ns, const int32_t* hs, const int32_t* ws, const float* scale,
                                        float* dw, float* db, void* stream) {
  if (!d || !xs || !dys || !ns || !hs || !ws || !dw || n_seg < 1 || n_seg > kMaxSeg)
    return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_grouped: bad arguments");
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad dtype");
  WgradProblem pr[kMaxSeg];
  for (int i = 0; i < n_seg; ++i) pr[i] = WgradProblem{d, ns[i], hs[i], ws[i], xs[i], dys[i], scale, dw, db};
  return wgrad_launch(n_seg, pr, OSD_STREAM(stream));
}

static int pred_blocks(int n_seg, const int32_t* ns, const int32_t* hs, const int32_t* ws, int ppb) {
  long long b = 0;
  for (int i = 0; i < n_seg; ++i) b += cdiv((int)((long long)ns[i] * hs[i] * ws[i]), ppb);
  return (int)b;
}

// ---- the prediction convs' weight gradient as a 1x1 GEMM (default) ----
// dW[co][r][s][ci] = sum over input pixels q of x[q][ci] * dy[q - (r - 1, s - 1)][co]: with the nine shifted dy vectors of a
// pixel laid side by side, G[q][tap * 4 + co] (36 of 40 columns; zero outside the map), this is the weight gradient of a 1x1
// conv with 36 output channels — the MFMA kernel above with ONE output row-tile, x read once — whose result T[tap * 4 + co][ci]
// a last tiny launch adds into dw[co][tap][ci] (and the centre tap's column sums into db).  The gather moves 8 + 80 bytes per
// pixel (a sixth of x's bytes): a thread owns one 16-byte chunk = two taps x four channels, 8 lanes cover a pixel's row (chunks 5-7
// and the second half of chunk 4 are zero columns).
// Round 5: the same G gives the DATA gradient dX[q][ci] = sum over (tap, co) of G[q][tap * 4 + co] * W[co][tap][ci] — a 1x1 conv
// with K = 64 over G (osd_pred_dy_gather + osd_pred_dgrad_pack + osd_conv2d_fwd) instead of a 3x3 conv over dy whose 4 input
// channels are padded to 64 per tap (K = 576 for 36 real: 118 - 132 us on the critical chain for a 70 MB write).
constexpr int kPredG = 64;             // columns of G (36 used; round 5: 64 instead of 40 so that G is also the input of the DATA gradient as a 1x1 conv with K = 64)
struct PredGatherLevels { const void* dy[kPredLevels]; int H[kPredLevels], W[kPredLevels], npix[kPredLevels], begin[kPredLevels]; int n_levels; };

template <typename T>
__global__ void __launch_bounds__(256) pred_dy_gather_kernel(PredGatherLevels L, T* __restrict__ g, int dy_stride) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;          // (pixel over all levels, chunk)
  const int c8 = (int)(idx & 7);
  const int q_all = (int)(idx >> 3);
  int lvl = -1;
#pragma unroll
  for (int i = 0; i < kPredLevels; ++i)
    if (i < L.n_levels && q_all >= L.begin[i] && q_all < L.begin[i] + L.npix[i]) lvl = i;
  if (lvl < 0) return;
  const void* dyv = L.dy[0];
  int H = L.H[0], W = L.W[0], beg = L.begin[0];
#pragma unroll
  for (int i = 1; i < kPredLevels; ++i)
    if (lvl == i) { dyv = L.dy[i]; H = L.H[i]; W = L.W[i]; beg = L.begin[i]; }
  const T* __restrict__ dy = reinterpret_cast<const T*>(dyv);
  const int q = q_all - beg, HW = H * W;
  const int img = q / HW, rem = q - img * HW, qy = rem / W, qx = rem - qy * W;
  T out[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int t = 2 * c8 + h;
    const int py = qy - t / 3 + 1, px = qx - t % 3 + 1;
    const bool ok = t < 9 && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
    const T* dp = dy + ((size_t)(img * H + (ok ? py : qy)) * W + (ok ? px : qx)) * dy_stride;
#pragma unroll
    for (int co = 0; co < 4; ++co) out[h * 4 + co] = ok ? dp[co] : from_f32<T>(0.f);
  }
  T* o = g + (size_t)q_all * kPredG + c8 * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = out[j];
}

// dw[co][tap][ci] += T[tap * 4 + co][ci]; db[co] += column sums of the centre tap (tap 4 reads dy[q][co] itself)
__global__ void __launch_bounds__(256) pred_wgrad_scatter_kernel(const float* __restrict__ t, const float* __restrict__ dbg, float* __restrict__ dw,
                                                                float* __restrict__ db, int cout, int cin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = cout * 9 * cin;
  if (i < n) {
    const int ci = i % cin, tap = (i / cin) % 9, co = i / (9 * cin);
    dw[i] += t[(size_t)(tap * 4 + co) * cin + ci];
  }
  if (db != nullptr && i < cout) db[i] += dbg[4 * 4 + i];
}

// pixels per workgroup: a multiple of the 32 pixels a workgroup reads per iteration, at least 256, and large enough that
// all workgroups are resident at once (2 per CU: the kernel is VALU bound, a second round of a few workgroups costs a
// whole round's time)
static int pred_ppb(int n_seg, const int32_t* ns, const int32_t* hs, const int32_t* ws) {
  long long tot = 0;
  for (int i = 0; i < n_seg; ++i) tot += (long long)ns[i] * hs[i] * ws[i];
  int ppb = (int)((tot + 479) / 480 + 31) / 32 * 32;
  if (ppb < 256) ppb = 256;
  while (pred_blocks(n_seg, ns, hs, ws, ppb) > 512) ppb += 32;
  return ppb;
}


static int pred_gather_launch(int dtype, int n_seg, const void* const* dys, const int32_t* ns, const int32_t* hs, const int32_t* ws,
                              int dy_stride, void* g, hipStream_t st) {
  PredGatherLevels L;
  L.n_levels = n_seg;
  long long tot = 0;
  for (int i = 0; i < kPredLevels; ++i) {
    const int j = i < n_seg ? i : 0;
    const long long npix = (long long)ns[j] * hs[j] * ws[j];
    L.dy[i] = dys[j]; L.H[i] = hs[j]; L.W[i] = ws[j]; L.npix[i] = i < n_seg ? (int)npix : 0; L.begin[i] = (int)tot;
    if (i < n_seg) tot += npix;
  }
  if (tot * 8 > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "pred_dy_gather: too many pixels");
  const unsigned gblocks = (unsigned)((tot * 8 + 255) / 256);
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(pred_dy_gather_kernel<float>, dim3(gblocks), dim3(256), 0, st, L, (float*)g, dy_stride);
  else
    hipLaunchKernelGGL(pred_dy_gather_kernel<__bf16>, dim3(gblocks), dim3(256), 0, st, L, (__bf16*)g, dy_stride);
  return osd_check_launch("pred_dy_gather");
}

// the weight gradient from G [pixels of all levels][64] (gathered by the caller): T[tap * 4 + co][ci] by the MFMA weight-gradient
// kernel on the 1x1 problem (x, G), then the scatter into dw / db.  workspace: [column sums: 256 B][T: 64 x cin fp32]
static int pred_wgrad_from_g(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* g, const int32_t* ns, const int32_t* hs,
                             const int32_t* ws, float* dw, float* db, void* workspace, hipStream_t st) {
  const size_t esz = d->dtype == OSD_BF16 ? 2 : 4;
  float* dbg = static_cast<float*>(workspace);
  float* tbuf = dbg + 64;
  if (hipMemsetAsync(workspace, 0, 256 + (size_t)kPredG * d->cin * 4, st) != hipSuccess)
    return osd_fail(OSD_ERR_LAUNCH, "wgrad_pred: memset failed");
  osd_conv_desc d1[kPredLevels];
  WgradProblem pr[kPredLevels];
  long long tot = 0;
  for (int i = 0; i < n_seg; ++i) {
    const long long npix = (long long)ns[i] * hs[i] * ws[i];
    if (npix > 0x7fffffffLL / d->cin) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred: bad segment %d", i);
    d1[i] = *d;
    d1[i].cout = 36; d1[i].r = 1; d1[i].s = 1; d1[i].pad_h = 0; d1[i].pad_w = 0; d1[i].out_stride = kPredG;
    static int code = -1;                    // split-target code (OSD_PRED_WGRAD_CODE: A/B); 512 workgroups measured best (65 us; 128: 102, 1,024: 90)
    if (code < 0) { const char* e = getenv("OSD_PRED_WGRAD_CODE"); code = e ? atoi(e) & 7 : 0; }
    d1[i].algo = 1 + 0 + 16 * code;          // 128 x 128 tile, 32-pixel stages
    pr[i] = WgradProblem{&d1[i], ns[i], hs[i], ws[i], xs[i], static_cast<const char*>(g) + (size_t)tot * kPredG * esz, nullptr, tbuf, db ? dbg : nullptr};
    tot += npix;
  }
  int rc = wgrad_launch(n_seg, pr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(pred_wgrad_scatter_kernel, dim3(cdiv(d->cout * 9 * d->cin, 256)), dim3(256), 0, st, (const float*)tbuf, (const float*)dbg,
                     dw, db, d->cout, d->cin);
  return osd_check_launch("pred_wgrad_scatter");
}

// Wd[ci][tap * 4 + co] = w[co][tap][ci]: the prediction conv's weights as the [cin rows][64] matrix of the 1x1 data-gradient conv
// over G (columns >= 36, and co >= cout, are zero)
template <typename T>
__global__ void __launch_bounds__(256) pred_dgrad_pack_kernel(const float* __restrict__ w, T* __restrict__ wd, int cout, int cin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cin * kPredG) return;
  const int ci = i / kPredG, col = i - ci * kPredG;
  const int tap = col >> 2, co = col & 3;
  float v = 0.f;
  if (tap < 9 && co < cout) v = w[((size_t)co * 9 + tap) * cin + ci];
  wd[i] = from_f32<T>(v);
}

extern "C" int64_t osd_conv2d_wgrad_pred_workspace_bytes(int n_seg, const int32_t* ns, const int32_t* hs, const int32_t* ws,
                                                         int cin) {
  if (n_seg < 1 || !ns || !hs || !ws || cin <= 0) return 0;
  const long long blocks = pred_blocks(n_seg, ns, hs, ws, pred_ppb(n_seg, ns, hs, ws));
  long long tot = 0;
  for (int i = 0; i < n_seg; ++i) tot += (long long)ns[i] * hs[i] * ws[i];
  const long long readonce = blocks * ((cin + 255) / 256) * 36 * 256 * 4 + blocks * 16 * 4 + 256;
  const long long gather = tot * kPredG * 4 + (long long)kPredG * cin * 4 + 1024;      // G [pixels][64] (widest dtype) + T [64][cin] + column sums
  return (int64_t)(readonce > gather ? readonce : gather);
}

// The prediction convs' weight + bias gradient (3x3 / stride 1 / pad 1, Cout <= 4, Cin a multiple of 256) over n_seg FPN
// levels: the read-once kernel + a fixed-order reduction of the per-workgroup partials (deterministic, no atomics)
extern "C" int osd_conv2d_wgrad_pred(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                                     const int32_t* ns, const int32_t* hs, const int32_t* ws, float* dw, float* db,
                                     void* workspace, void* stream) {
  if (!d || !xs || !dys || !ns || !hs || !ws || !dw || !workspace || n_seg < 1 || n_seg > kPredLevels)
    return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred: bad arguments (1..%d levels)", kPredLevels);
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred: bad dtype");
  if (d->cout < 1 || d->cout > 4 || d->r != 3 || d->s != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 ||
      d->pad_w != 1 || d->cin % 256 != 0 || d->out_stride % 4 != 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad_pred: 3x3 / stride 1 / pad 1, cout <= 4, cin %% 256 == 0, dy rows of >= 4 channels");
  for (int i = 0; i < n_seg; ++i)
    if (ns[i] <= 0 || hs[i] <= 0 || ws[i] <= 0 || !xs[i] || !dys[i]) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred: bad segment %d", i);
  static int use_gemm = -1;     // OSD_PRED_WGRAD_READONCE=1: the first-generation read-once kernel (A/B)
  if (use_gemm < 0) { const char* e = getenv("OSD_PRED_WGRAD_READONCE"); use_gemm = (e && e[0] == '1') ? 0 : 1; }
  if (use_gemm) {
    hipStream_t st = OSD_STREAM(stream);
    // workspace: [column sums: 256 B][T: 64 x cin fp32][G: pixels x 64]
    char* gbase = reinterpret_cast<char*>(static_cast<float*>(workspace) + 64 + (size_t)kPredG * d->cin);
    int rc = pred_gather_launch(d->dtype, n_seg, dys, ns, hs, ws, d->out_stride, gbase, st);
    if (rc) return rc;
    return pred_wgrad_from_g(d, n_seg, xs, gbase, ns, hs, ws, dw, db, workspace, st);
  }
  PredWgradLevels L;
  L.n_levels = 0;
  const int ppb = pred_ppb(n_seg, ns, hs, ws);
  int blocks = 0;
  for (int i = 0; i < n_seg; ++i) {
    const long long npix = (long long)ns[i] * hs[i] * ws[i];
    if (npix <= 0 || npix > 0x7fffffffLL / d->cin || !xs[i] || !dys[i])
      return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred: bad segment %d", i);
    const int k = L.n_levels++;
    L.x[k] = xs[i]; L.dy[k] = dys[i]; L.H[k] = hs[i]; L.W[k] = ws[i]; L.npix[k] = (int)npix; L.begin[k] = blocks;
    blocks += cdiv((int)npix, ppb);
  }
  for (int k = L.n_levels; k < kPredLevels; ++k) { L.x[k] = L.x[0]; L.dy[k] = L.dy[0]; L.H[k] = 1; L.W[k] = 1; L.npix[k] = 0; L.begin[k] = 0x7fffffff; }
  const int groups = d->cin / 256;
  float* part = static_cast<float*>(workspace);
  float* part_b = part + (size_t)blocks * groups * 36 * 256;
  dim3 grid(blocks, groups);
  hipStream_t st = OSD_STREAM(stream);
  if (d->dtype == OSD_F32)
    hipLaunchKernelGGL(pred_wgrad_kernel<float>, grid, dim3(256), 0, st, L, part, part_b, d->cin, d->out_stride, ppb);
  else
    hipLaunchKernelGGL(pred_wgrad_kernel<__bf16>, grid, dim3(256), 0, st, L, part, part_b, d->cin, d->out_stride, ppb);
  int rc = osd_check_launch("pred_wgrad");
  if (rc) return rc;
  const int n = d->cout * 9 * d->cin + d->cout;
  hipLaunchKernelGGL(pred_wgrad_reduce_kernel, dim3(cdiv(n, 256), 16), dim3(256), 0, st, (const float*)part, (const float*)part_b, dw, db,
                     blocks, groups, d->cin, d->cout);
  return osd_check_launch("pred_wgrad(reduce)");
}

static int pred_check(const char* who, const osd_conv_desc* d, int n_seg, const int32_t* ns, const int32_t* hs, const int32_t* ws) {
  if (!d || !ns || !hs || !ws || n_seg < 1 || n_seg > kPredLevels) return osd_fail(OSD_ERR_INVALID_ARG, "%s: bad arguments (1..%d levels)", who, kPredLevels);
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "%s: bad dtype", who);
  if (d->cout < 1 || d->cout > 4 || d->r != 3 || d->s != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1 ||
      d->cin % 256 != 0 || d->out_stride % 4 != 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "%s: 3x3 / stride 1 / pad 1, cout <= 4, cin %% 256 == 0, dy rows of >= 4 channels", who);
  for (int i = 0; i < n_seg; ++i)
    if (ns[i] <= 0 || hs[i] <= 0 || ws[i] <= 0) return osd_fail(OSD_ERR_INVALID_ARG, "%s: bad segment %d", who, i);
  return OSD_OK;
}

extern "C" int osd_pred_dy_gather(const osd_conv_desc* d, int n_seg, const void* const* dys, const int32_t* ns, const int32_t* hs,
                                  const int32_t* ws, void* g, void* stream) {
  if (!dys || !g) return osd_fail(OSD_ERR_INVALID_ARG, "pred_dy_gather: null argument");
  int rc = pred_check("pred_dy_gather", d, n_seg, ns, hs, ws);
  if (rc) return rc;
  for (int i = 0; i < n_seg; ++i)
    if (!dys[i]) return osd_fail(OSD_ERR_INVALID_ARG, "pred_dy_gather: null segment %d", i);
  return pred_gather_launch(d->dtype, n_seg, dys, ns, hs, ws, d->out_stride, g, OSD_STREAM(stream));
}

extern "C" int osd_pred_dgrad_pack(int dtype, const float* w, int cout, int cin, void* wd, void* stream) {
  if (!w || !wd || cout < 1 || cout > 4 || cin < 1) return osd_fail(OSD_ERR_INVALID_ARG, "pred_dgrad_pack: bad arguments (cout 1..4)");
  if (dtype != OSD_F32 && dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "pred_dgrad_pack: bad dtype");
  const unsigned blocks = (unsigned)cdiv(cin * kPredG, 256);
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(pred_dgrad_pack_kernel<float>, dim3(blocks), dim3(256), 0, OSD_STREAM(stream), w, (float*)wd, cout, cin);
  else
    hipLaunchKernelGGL(pred_dgrad_pack_kernel<__bf16>, dim3(blocks), dim3(256), 0, OSD_STREAM(stream), w, (__bf16*)wd, cout, cin);
  return osd_check_launch("pred_dgrad_pack");
}

extern "C" int osd_conv2d_wgrad_pred_gathered(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* g, const int32_t* ns,
                                              const int32_t* hs, const int32_t* ws, float* dw, float* db, void* workspace, void* stream) {
  if (!xs || !g || !dw || !workspace) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred_gathered: null argument");
  int rc = pred_check("wgrad_pred_gathered", d, n_seg, ns, hs, ws);
  if (rc) return rc;
  for (int i = 0; i < n_seg; ++i)
    if (!xs[i]) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_pred_gathered: null segment %d", i);
  return pred_wgrad_from_g(d, n_seg, xs, g, ns, hs, ws, dw, db, workspace, OSD_STREAM(stream));
}

// n_seg convs of IDENTICAL geometry (same x / dy shapes, different tensors and different weights: the repeated bottleneck
// blocks of a ResNet stage) in one launch, each with its own dW / scale / db: the output tiles of all of them share the
// workgroup budget, so each needs 1/n_seg of the pixel splits — and of the atomic traffic — of a launch of its own
extern "C" int osd_conv2d_wgrad_batched(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                                        const float* const* scales, float* const* dws, float* const* dbs, void* stream) {
  if (!d || !xs || !dys || !dws || n_seg < 1 || n_seg > kMaxSeg)
    return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_batched: bad arguments");
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad dtype");
  if (d->in_stride_w != d->cin || d->in_stride_h != d->w * d->cin)
    return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad_batched: dense NHWC input required");
  WgradProblem pr[kMaxSeg];
  for (int i = 0; i < n_seg; ++i)
    pr[i] = WgradProblem{d, d->n, d->h, d->w, xs[i], dys[i], scales ? scales[i] : nullptr, dws[i], dbs ? dbs[i] : nullptr};
  return wgrad_launch(n_seg, pr, OSD_STREAM(stream));
}

// (x, dy) pairs with their own batch / spatial size AND their own dW / scale / db, same conv geometry — e.g. the four
// convs of an FCOS tower x five FPN levels in one launch (pairs that share a dW simply repeat its pointer)
extern "C" int osd_conv2d_wgrad_multi(const osd_conv_desc* d, int n_seg, const void* const* xs, const void* const* dys,
                                      const int32_t* ns, const int32_t* hs, const int32_t* ws, const float* const* scales,
                                      float* const* dws, float* const* dbs, void* stream) {
  if (!d || !xs || !dys || !ns || !hs || !ws || !dws || n_seg < 1 || n_seg > kMaxSeg)
    return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_multi: bad arguments (1..%d segments)", kMaxSeg);
  if (d->dtype != OSD_F32 && d->dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad dtype");
  WgradProblem pr[kMaxSeg];
  for (int i = 0; i < n_seg; ++i)
    pr[i] = WgradProblem{d, ns[i], hs[i], ws[i], xs[i], dys[i], scales ? scales[i] : nullptr, dws[i], dbs ? dbs[i] : nullptr};
  return wgrad_launch(n_seg, pr, OSD_STREAM(stream));
}

// the general form: every pair has its OWN conv descriptor (channels, kernel, stride, pad, n, h, w, out_stride): all weight
// gradients of a ResNet stage — 1x1 and 3x3, stride 1 and 2, the FPN laterals — in one launch.  dtype and algo are taken
// from descs[0].  The workgroup budget is shared out in proportion to pixels x output tiles.
extern "C" int osd_conv2d_wgrad_mixed(int n_seg, const osd_conv_desc* descs, const void* const* xs, const void* const* dys,
                                      const float* const* scales, float* const* dws, float* const* dbs, void* stream) {
  if (!descs || !xs || !dys || !dws || n_seg < 1 || n_seg > kMaxSeg)
    return osd_fail(OSD_ERR_INVALID_ARG, "wgrad_mixed: bad arguments (1..%d segments)", kMaxSeg);
  if (descs[0].dtype != OSD_F32 && descs[0].dtype != OSD_BF16) return osd_fail(OSD_ERR_INVALID_ARG, "wgrad: bad dtype");
  WgradProblem pr[kMaxSeg];
  for (int i = 0; i < n_seg; ++i) {
    const osd_conv_desc* d = descs + i;
    if (d->in_stride_w != d->cin || d->in_stride_h != d->w * d->cin)
      return osd_fail(OSD_ERR_UNSUPPORTED, "wgrad_mixed: dense NHWC input required (segment %d)", i);
    pr[i] = WgradProblem{d, d->n, d->h, d->w, xs[i], dys[i], scales ? scales[i] : nullptr, dws[i], dbs ? dbs[i] : nullptr};
  }
  return wgrad_launch(n_seg, pr, OSD_STREAM(stream));
}

extern "C" int osd_bias_grad(const void* dy, float* db, int m, int c, int stride, int dtype, void* stream) {
  if (!dy || !db) return osd_fail(OSD_ERR_INVALID_ARG, "bias_grad: null argument");
  if (m == 0 || c == 0) return OSD_OK;
  const int rows = m > 65536 ? 2048 : 512;
  dim3 grid(cdiv(m, rows), cdiv(c, 256));
  if (dtype == OSD_F32)
    hipLaunchKernelGGL(bias_grad_kernel<float>, grid, dim3(256), 0, OSD_STREAM(stream), (const float*)dy, db, m, c, stride, rows);
  else if (dtype == OSD_BF16)
    hipLaunchKernelGGL(bias_grad_kernel<__bf16>, grid, dim3(256), 0, OSD_STREAM(stream), (const __bf16*)dy, db, m, c, stride,
                       rows);
  else
    return osd_fail(OSD_ERR_INVALID_ARG, "bias_grad: bad dtype");
  return osd_check_launch("bias_grad");
}
