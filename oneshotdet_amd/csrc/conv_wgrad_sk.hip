// conv_wgrad_sk — the software-pipelined weight-gradient kernel (bf16, gfx950), variants 13 and 3 of osd_conv2d_wgrad*.
//     dW[co][r][s][ci] += sum_m dY[m][co] * X[m @ tap(r,s)][ci]          m = (n, ho, wo)
// Same GEMM view, LDS image (pixel-major 256-byte rows, 16-byte-chunk swizzle chunk ^= 2*(row&7), transposed fragment reads
// with ds_read_b64_tr_b16) and 256 co x 256 ci tile on eight waves (2 x 4: 128 co x 64 ci per wave) as variant 5 of
// conv_wgrad.hip.  What differs:
//   * the loop is software-pipelined the way conv_igemm_sp.hip is: 64-pixel stages in a 2-deep ring, two 32-deep k steps per
//     stage; the operand fragments of k step u + 1 are read while the MFMAs of step u run (the dY fragments are replaced one
//     by one behind their last MFMA, the X fragments alternate between two sets); the ONE barrier of a stage sits behind the
//     first MFMA group of its second k step — it publishes the next stage and frees this stage's buffer, whose last
//     fragments are in registers by then — and the DMA of the stage after next goes there, one 1 KiB piece per MFMA group;
//   * the DMA is bounds-checked buffer loads (`buffer_load_dwordx4 ... offen lds`; zeros for rows past the work item's end,
//     channels past Cout / Cin and pixels outside the map) with per-lane state of two parity offsets per operand; the input
//     pixel of output pixel m at tap (fr, fs) and its validity come from (n, ho, wo) = m by multiply-high reciprocals, per
//     piece (any stride / padding).  Stages past the end of a work item are all-zero fetches, so every stage of the loop is
//     the same code (no peeled tails);
//   * TEAM MODE (variant 3; WgradParams.sk_units > 0): the launch's pixels — all segments, concatenated in 64-pixel stages —
//     are cut into sk_teams equal ranges, and a team of sk_units workgroups (one per output tile and tap) walks one range in
//     lockstep: the nine taps of a pixel range run side by side on one XCD (they read the same dY rows and one-pixel-shifted
//     copies of the same X rows out of its L2, as the splits of the other variants do), every workgroup does the same
//     number of stages (no tail of small workgroups: P3 has 228x the pixels of P7), and a partial tile is added to dW only
//     where the range ends or the next segment names another dW — the FPN levels of a conv are one accumulation.  The
//     atomic traffic falls from one 256 KB tile per split (1536 splits in the tower launch) to one per workgroup (~252).
#include "wgrad_params.h"
#include <type_traits>

namespace {

constexpr int WCO = 2, WCI = 2, WM = 2, WN = 4;     // 256-byte sub-tiles per operand, waves along co / ci
constexpr int NW = WM * WN, EPC = 8, TWS = 128;     // waves, elements per 16-byte chunk, channels per sub-tile row
constexpr int TCO = WCO * TWS, TCI = WCI * TWS;
constexpr int TA = TCO / WM / 16, TB = TCI / WN / 16;
constexpr int WCOL_A = TCO / WM, WCOL_B = TCI / WN;
constexpr int SKU = 64;                             // team mode: pixels per unit of WgradSeg.stage_begin / WgradParams.sk_total
typedef __bf16 T;
typedef int sk_i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;               // a byte offset past every buffer (< 2 GiB, checked by the launcher)

__device__ __forceinline__ sk_i32x4 sk_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  sk_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

__device__ __forceinline__ void sk_dma16(sk_i32x4 rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "s"(rsrc)
               : "memory");
}

// BKP pixels per stage, NST stages in the ring: 64 x 2 (two k steps per stage) is what is launched.  32 x 5 (one k step per
// stage, twice the pixels of prefetch distance, all 160 KB of LDS) compiles from the same source but spills inside the loop
// and measured 1.47 ms on the tower launch where 64 x 2 takes 0.63; an L2 prefetch (one line per lane, two
// stages ahead of the DMA) cost 5 % instead of gaining: the memory system is not what the loop waits for
template <int BKP, int NST>
__global__ void __launch_bounds__(64 * NW, 1) conv_wgrad_sk_kernel(WgradParams gp) {
  constexpr int OPB = BKP * 256, STAGE = (WCO + WCI) * OPB;
  constexpr int PPS = BKP / 4;                        // 1 KiB DMA pieces (4 rows) per sub-tile
  constexpr int IPA = WCO * PPS / NW, IPB = WCI * PPS / NW, LPS = IPA + IPB;
  static_assert((BKP == 64 || BKP == 32) && PPS % IPA == 0 && PPS % IPB == 0 && IPA % 2 == 0 && IPB % 2 == 0 && LPS <= TA,
                "a wave's pieces: inside one sub-tile, at most one per MFMA group");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective remap (blocks b, b+8 share an XCD): consecutive logical ids — the output tiles and taps of one pixel
  // split / one team, which read the same dY / X rows — land on ONE XCD's L2
  int bid;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  typedef const __attribute__((address_space(4))) char* kptr;
  typedef unsigned long long u64;
  const kptr seg0 = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WgradParams, seg);
#define OSD_KSEG(idx, type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(seg0 + (idx) * (int)sizeof(WgradSeg) + offsetof(WgradSeg, field)))

  // ---- the work of this workgroup: one output tile and tap, and a run of 64-pixel stages [g0, g1) on the launch's
  // concatenated pixel axis (team mode) or one split of one segment ----
  const bool team = gp.sk_units > 0;
  int sidx = 0, unit, g0 = 0, g1 = 0, split = 0;
  if (team) {
    const int tm = bid / gp.sk_units;
    unit = bid - tm * gp.sk_units;
    g0 = (int)((long long)tm * gp.sk_total / gp.sk_teams);
    g1 = (int)((long long)(tm + 1) * gp.sk_total / gp.sk_teams);
    for (int i = 1; i < gp.n_seg; ++i)
      if (g0 >= OSD_KSEG(i, int, stage_begin)) sidx = i;
  } else {
    for (int i = 1; i < gp.n_seg; ++i)
      if (bid >= OSD_KSEG(i, int, block_begin)) sidx = i;
    const int local = bid - OSD_KSEG(sidx, int, block_begin);
    const int per_split = OSD_KSEG(sidx, int, tilesCo) * OSD_KSEG(sidx, int, tilesCi) * OSD_KSEG(sidx, int, R) * OSD_KSEG(sidx, int, S);
    split = local / per_split;
    unit = local - split * per_split;
  }
  const int slot_id = bid;            // logical id over the whole launch: the partial tile's slot in ordered mode

  // fragment coordinates (the same for every work item)
  const int a_sub = (wm * WCOL_A) / TWS, a_c0 = (wm * WCOL_A) % TWS;
  const int b_sub = (wn * WCOL_B) / TWS, b_c0 = (wn * WCOL_B) % TWS;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
  typedef __attribute__((address_space(3))) bf4* lds_bf4_ptr;
  const int g = lane >> 4, t16 = lane & 15, q4 = t16 >> 2, pp = t16 & 3;
  const int h8 = (pp & 1) * 8;
  // one MFMA operand fragment: k slot (g, j) <-> tile row (j < 4 ? 4g + j : 16 + 4g + j - 4) of k step k32.  Addresses: fragment
  // idx of an operand starts 2 * idx chunks further; chunk bits 1..3 only meet the swizzle by XOR, so its address is the
  // address of fragment 0 XOR (idx << 5) — ONE address register per operand (fa0 / fb0, inside a stage), the k step, the second
  // row half and the fragment index are immediates (the ring's buffers start on multiples of 256 bytes)
  const int rr = 4 * g + q4;
  const unsigned fa0 = a_sub * OPB + rr * 256 + ((((a_c0 >> 3) + (pp >> 1)) ^ wg_swz<T>(rr)) << 4) + h8;
  const unsigned fb0 = (WCO + b_sub) * OPB + rr * 256 + ((((b_c0 >> 3) + (pp >> 1)) ^ wg_swz<T>(rr)) << 4) + h8;
  static_assert((WCOL_A % TWS == 0 || TA * 2 <= 8) && ((WCOL_B % TWS) % 64 == 0 && TB * 2 <= 8), "fragment index bits stay clear of the wave's first chunk");
  auto read_frag = [&](unsigned stage_base, unsigned f0, int idx, int k32) -> bf16x8 {
#ifdef OSD_WG_NO_READS           // diagnostic: no LDS reads in the loop
    if (gp.n_seg >= 0) { bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (__bf16)(float)(idx + k32); return z; }
#endif
    const unsigned a = ((stage_base + f0) ^ (unsigned)(idx << 5)) + (unsigned)(k32 * 32 * 256);
    const bf4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)a);
    const bf4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(a + 16 * 256));
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[e + 4] = hi[e]; }
    return v;
  };
  // DMA coordinates of this wave: its first piece per operand -> sub-tile and first tile row
  const int a_q0 = wave * IPA, b_q0 = wave * IPB;
  const int a_sub_d = a_q0 / PPS, a_row0 = (a_q0 % PPS) * 4;
  const int b_sub_d = b_q0 / PPS, b_row0 = (b_q0 % PPS) * 4;
  const int plrow = lane >> 4, plpos = lane & 15;

  f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  for (;;) {
    // ---- this segment: geometry, tensors, and the pixels [p_lo, p_hi) of it that belong to the work item ----
    const kptr sb = seg0 + sidx * (int)sizeof(WgradSeg);
#define OSD_WSEG(type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(sb + offsetof(WgradSeg, field)))
    const void* px = (const void*)OSD_WSEG(u64, x);
    const void* pdy = (const void*)OSD_WSEG(u64, dy);
    float* pdw = (float*)OSD_WSEG(u64, dw);
    const float* pscale = (const float*)OSD_WSEG(u64, scale);
    float* pdb = (float*)OSD_WSEG(u64, db);
    const int H = OSD_WSEG(int, H), W = OSD_WSEG(int, W), Ho = OSD_WSEG(int, Ho), Wo = OSD_WSEG(int, Wo), M = OSD_WSEG(int, M);
    const int Cin = OSD_WSEG(int, Cin), Cout = OSD_WSEG(int, Cout), S = OSD_WSEG(int, S);
    const int sh = OSD_WSEG(int, sh), sw = OSD_WSEG(int, sw), ph = OSD_WSEG(int, ph), pw = OSD_WSEG(int, pw);
    const int dy_stride = OSD_WSEG(int, dy_stride), tilesCo = OSD_WSEG(int, tilesCo), tilesCi = OSD_WSEG(int, tilesCi);
    const int Ktot = OSD_WSEG(int, Ktot);
    const int co_tile = unit % tilesCo, nt = unit / tilesCo;
    const int tap = nt / tilesCi, ci_tile = nt % tilesCi;
    const int fr = tap / S, fs = tap % S;
    const int co0 = co_tile * TCO, ci0 = ci_tile * TCI;
    int p_lo, p_hi;
    bool last;                         // no further segment in this work item
    if (team) {
      const int sbeg = OSD_WSEG(int, stage_begin);
      const int kts = (M + SKU - 1) / SKU;
      p_lo = (max(g0, sbeg) - sbeg) * SKU;
      p_hi = min(M, (min(g1, sbeg + kts) - sbeg) * SKU);
      last = g1 <= sbeg + kts || sidx + 1 >= gp.n_seg;
    } else {
      const int rows = OSD_WSEG(int, rows_per_split);
      p_lo = split * rows;
      p_hi = min(M, p_lo + rows);
      last = true;
    }
#undef OSD_WSEG
    const int KT = p_lo < p_hi ? (p_hi - p_lo + BKP - 1) / BKP : 0;

    if (KT > 0) {
      const sk_i32x4 dyrs = sk_make_rsrc(pdy, (unsigned)M * (unsigned)dy_stride * 2u);
      const sk_i32x4 xrs = sk_make_rsrc(px, (unsigned)(M / (Ho * Wo)) * (unsigned)(H * W) * (unsigned)Cin * 2u);
      // exact floor(m / Wo) and floor(r / Ho) by multiply-high with floor(2^32 / d) + 1: exact while m < 2^32 / d (the launcher
      // checks M * Wo < 2^32 and d >= 2)
      const unsigned magic_w = 0xffffffffu / (unsigned)Wo + 1u, magic_h = 0xffffffffu / (unsigned)Ho + 1u;
      // per-lane state: dY — byte offset of (row plrow + 4 * parity, my 16-byte chunk) for instruction parity 0 / 1 (instruction
      // i adds (i & ~1) * 4 rows); X — my chunk's byte offset inside a pixel, per parity (the pixel itself is computed per piece)
      unsigned pa[2], pb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int swz = plpos ^ wg_swz<T>(i * 4 + plrow);          // row0 is a multiple of 8: the swizzle sees (i & 1) * 4 + plrow
        const int ca = co0 + a_sub_d * TWS + swz * EPC, cb = ci0 + b_sub_d * TWS + swz * EPC;
        pa[i] = ca < Cout ? (unsigned)((plrow + i * 4) * dy_stride + ca) * 2u : OOB;
        pb[i] = cb < Cin ? (unsigned)cb * 2u : OOB;
      }
      const int hoff = fr - ph, woff = fs - pw;
      // issue piece pc (0 .. IPA - 1: dY, then X) of the stage whose first pixel is m_first into the buffer at `st`
      auto issue_piece = [&](int pc, unsigned st, int m_first) {
        if (pc < IPA) {
          const int i = pc;
          const int row0 = a_row0 + i * 4;                              // tile row of lane group 0 (uniform)
          const bool rowok = plrow < p_hi - m_first - row0;             // rows of this piece inside the work item
          const unsigned off = pa[i & 1] + (unsigned)((m_first + a_row0 + (i & ~1) * 4) * dy_stride) * 2u;
          sk_dma16(dyrs, (rowok && pa[i & 1] != OOB) ? off : OOB, st + (unsigned)(a_sub_d * OPB + (a_q0 % PPS + i) * 1024));
        } else {
          const int i = pc - IPA;
          const int row0 = b_row0 + i * 4;
          const bool rowok = plrow < p_hi - m_first - row0;
          const unsigned m = (unsigned)(m_first + row0 + plrow);        // my output pixel -> (n, ho, wo) -> the input pixel of this tap
          const unsigned r = __umulhi(m, magic_w);
          const unsigned n_img = __umulhi(r, magic_h);
          const int wi = (int)(m - r * (unsigned)Wo) * sw + woff;
          const int hi = (int)(r - n_img * (unsigned)Ho) * sh + hoff;
          const bool ok = (int)rowok & (int)(pb[i & 1] != OOB) & (int)((unsigned)hi < (unsigned)H) & (int)((unsigned)wi < (unsigned)W);
          const unsigned off = pb[i & 1] + ((n_img * (unsigned)H + (unsigned)hi) * (unsigned)W + (unsigned)wi) * (unsigned)(Cin * 2);
          sk_dma16(xrs, ok ? off : OOB, st + (unsigned)((WCO + b_sub_d) * OPB + (b_q0 % PPS + i) * 1024));
        }
      };
      // bias gradient: the workgroups of ci tile 0 also sum their dY tile over its pixels, one column per thread.  The nine taps
      // of a pixel range see the same dY tile, so they take turns (stage s of the segment belongs to tap s mod 9): a
      // workgroup that summed every stage ran a third slower than its team, and a team moves in lockstep (tower launch: 812 us
      // with the bias gradient on tap 0, 606 without it, 669 with turns).  A 16-byte-per-thread form of the sum (all 512 threads,
      // partial sums in registers or in LDS) needs 8 - 30 more registers and made hipcc spill inside the MFMA loop: 1,035 us.
      // Ordered mode keeps the sum with tap 0, whose slot the reduction reads
      const int ntaps = gp.partials != nullptr ? 1 : OSD_KSEG(sidx, int, R) * S;
      const bool do_bias = (pdb != nullptr) && (ci_tile == 0) && (gp.partials != nullptr ? tap == 0 : true) && (tid < TCO);
      int bias_turn = (p_lo / BKP + ntaps - (gp.partials != nullptr ? 0 : tap)) % ntaps;      // 0: this stage is mine
      auto bias_stage = [&](int buf) {
        const int sub = tid / TWS, cc = tid % TWS;
        const char* sa = smem + buf * STAGE + sub * OPB;
        const int chunk = cc / EPC, within = cc % EPC;
#pragma unroll 8
        for (int row = 0; row < BKP; ++row)
          bsum += to_f32(*reinterpret_cast<const T*>(sa + row * 256 + ((chunk ^ wg_swz<T>(row)) << 4) + within * (int)sizeof(T)));
      };

      bf16x8 af[TA], bf0[TB], bf1[TB];
      int next_m = p_lo;                                    // first pixel of the next stage to fetch
      auto k_step = [&](bf16x8 (&bcur)[TB], bf16x8 (&bnxt)[TB], int nbuf, auto nk_tag, auto bar_tag, int dbuf) {
        constexpr int nk = decltype(nk_tag)::value;
        constexpr bool BAR = decltype(bar_tag)::value;      // the second k step of a stage: barrier, then fragments of the NEXT stage + DMA
        const unsigned nst = lds0 + nbuf * STAGE;
        const unsigned dst = lds0 + dbuf * STAGE;
        if constexpr (!BAR) {
#pragma unroll
          for (int j = 0; j < TB; ++j) bnxt[j] = read_frag(nst, fb0, j, nk);
        }
#pragma unroll
        for (int i = 0; i < TA; ++i) {
#pragma unroll
          for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bcur[j], acc[i][j], 0, 0, 0);
          if constexpr (BAR) {
            if (i == 0) {
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // all my reads of the buffer the barrier releases
              wg_wait_vmcnt<(NST - 2) * LPS>();                        // my pieces of the next stage have landed
#ifndef OSD_WG_NO_BAR
              __builtin_amdgcn_s_barrier();
#endif
#pragma unroll
              for (int j = 0; j < TB; ++j) bnxt[j] = read_frag(nst, fb0, j, nk);
            }
          }
          af[i] = read_frag(nst, fa0, i, nk);
#if defined(OSD_WG_SAME_ADDR)    // diagnostic builds: timing of the loop without one of its parts (results are garbage):
          if constexpr (BAR) { if (i < LPS) issue_piece(i, dst, p_lo); }        // ... every stage re-fetches the first one (always in the L2)
#elif !defined(OSD_WG_NO_DMA)    // ... no fetches after the prologue
          if constexpr (BAR) { if (i < LPS) issue_piece(i, dst, next_m); }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (BAR) next_m += BKP;
      };
      using K0 = std::integral_constant<int, 0>;
      using K1 = std::integral_constant<int, 1>;
      // prologue: the whole ring in flight, stage 0 landed and visible, its fragments in registers.  (A previous segment's
      // loop has left nothing in flight and every wave has passed its last barrier before any wave's first piece here can
      // land: each wave waited for its own fetches, and the buffers are only read before that barrier.)
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int st = 0; st < NST; ++st) {
#pragma unroll
        for (int pc = 0; pc < LPS; ++pc) issue_piece(pc, lds0 + st * STAGE, next_m);
        next_m += BKP;
      }
      wg_wait_vmcnt<(NST - 1) * LPS>();
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < TA; ++i) af[i] = read_frag(lds0, fa0, i, 0);
#pragma unroll
      for (int j = 0; j < TB; ++j) bf0[j] = read_frag(lds0, fb0, j, 0);
      int cur = 0;
      if constexpr (BKP == 64) {
        for (int kt = 0; kt < KT; ++kt) {
          if (do_bias && bias_turn == 0) bias_stage(cur);
          bias_turn = bias_turn + 1 == ntaps ? 0 : bias_turn + 1;
          k_step(bf0, bf1, cur, K1(), std::false_type(), 0);
          k_step(bf1, bf0, cur ^ 1, K0(), std::true_type(), cur);
          cur ^= 1;
        }
      } else {
        // one k step per stage; two stages per trip (the X fragment sets swap roles)
        int kt = 0;
        for (; kt + 1 < KT; kt += 2) {
          const int n1 = cur + 1 == NST ? 0 : cur + 1, n2 = n1 + 1 == NST ? 0 : n1 + 1;
          if (do_bias && ntaps == 1) bias_stage(cur);
          k_step(bf0, bf1, n1, K0(), std::true_type(), cur);
          if (do_bias && ntaps == 1) bias_stage(n1);
          k_step(bf1, bf0, n2, K0(), std::true_type(), n1);
          cur = n2;
        }
        if (kt < KT) {
          if (do_bias && ntaps == 1) bias_stage(cur);
          k_step(bf0, bf1, cur + 1 == NST ? 0 : cur + 1, K0(), std::true_type(), cur);
        }
      }
      wg_wait_vmcnt<0>();      // the zero fetches past the end: nothing may land in this LDS once the loop is left
    }

    // ---- where the accumulation ends: the work item's last segment, or the next one names another gradient ----
    bool flush = last;
    if (!last) {
      flush = (float*)OSD_KSEG(sidx + 1, u64, dw) != pdw || (const float*)OSD_KSEG(sidx + 1, u64, scale) != pscale ||
              (float*)OSD_KSEG(sidx + 1, u64, db) != pdb;
    }
    if (flush) {
      const bool do_bias = (pdb != nullptr) && (ci_tile == 0) && (gp.partials != nullptr ? tap == 0 : true) && (tid < TCO);
      if (gp.partials != nullptr) {
        // ordered mode (never in team mode): the raw partial tile (and the bias partial of the tap-0 / ci-tile-0 workgroups)
        // goes to this workgroup's slot; channels past Cout / Cin hold zeros (their operands were zero fetches)
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
      if (do_bias && co0 + tid < Cout) wg_atomic_add(pdb + co0 + tid, bsum);
#ifdef OSD_WG_NO_ATOMICS
      if (acc[0][0][0] != 12345.678f) return;
#endif
      // accumulate the partial tile into dW (fp32 atomics; rows = co, 16 consecutive ci per 16 lanes).  Whole-tile fast path:
      // the per-row FrozenBN scales are loaded up front and the 64 atomics of a wave follow in ONE basic block (a load or a
      // bounds branch between two atomics makes hipcc put `s_waitcnt vmcnt(0)` in front of every atomic)
      const int co_w = co0 + wm * WCOL_A + (lane >> 4) * 4;           // first co row of this lane (tile i adds 16 * i)
      const int ci_w = ci0 + wn * WCOL_B + (lane & 15);               // first ci of this lane (tile j adds 16 * j)
      if (co0 + TCO <= Cout && ci0 + TCI <= Cin) {
        float scv[TA][4];
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) scv[i][e] = pscale ? pscale[co_w + i * 16 + e] : 1.f;
        float* base = pdw + (size_t)co_w * Ktot + tap * Cin + ci_w;
        if (!team && OSD_KSEG(sidx, int, owner) != 0) {
          wg_owner_add<TA, TB>(base, Ktot, acc, scv);       // the only writer of this tile: plain load + store
        } else {
#pragma unroll
          for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float* row = base + (size_t)(i * 16 + e) * Ktot;
#pragma unroll
              for (int j = 0; j < TB; ++j) wg_atomic_add(row + j * 16, acc[i][j][e] * scv[i][e]);
            }
        }
      } else {
        // ragged tile (channel counts that are not multiples of the tile): per-element bounds checks
#pragma unroll
        for (int i = 0; i < TA; ++i) {
#pragma unroll
          for (int j = 0; j < TB; ++j) {
            const int ci = ci_w + j * 16;
            if (ci >= Cin) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int co = co_w + i * 16 + e;
              if (co < Cout) wg_atomic_add(pdw + (size_t)co * Ktot + tap * Cin + ci, acc[i][j][e] * (pscale ? pscale[co] : 1.f));
            }
          }
        }
      }
      if (last) return;
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      bsum = 0.f;
    }
    ++sidx;
  }
#undef OSD_KSEG
}

}  // namespace

template <int BKP, int NST> static void sk_launch(const WgradParams& p, hipStream_t s) {
  constexpr int lds = NST * (WCO + WCI) * BKP * 256;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_sk_kernel<BKP, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((conv_wgrad_sk_kernel<BKP, NST>), dim3((unsigned)p.n_blocks), dim3(64 * NW), lds, s, p);
}

int osd_wgrad_sk_launch(const WgradParams& p, hipStream_t s) {
  sk_launch<64, 2>(p, s);
  return 0;
}
