// conv_igemm_p8 — third-generation implicit-GEMM convolution for the large bf16 layers on gfx950: the 256 x 256 tile of
// conv_igemm_dma.hip re-scheduled as a ping-pong between two wave groups (the "8-phase" GEMM schedule of the CDNA
// guide, written for an im2col pixel operand).
//
// The 2-barrier-per-stage structure tops out near 900 TFLOP/s: every wave leaves the barrier together, so the LDS pipe
// (fragment reads) and the matrix pipe take turns.  Here
//   * 8 waves = 2 pixel halves (wm) x 4 channel quarters (wn); a wave owns 128 pixels x 64 channels = 4 quadrants of
//     64 x 32; one PHASE = the fragment reads of one quadrant + 16 MFMAs (16x16x32 bf16, K = 64);
//   * the two wave groups (wm = 0 / 1; they share every SIMD pairwise) run ONE BARRIER APART: while one group issues
//     MFMAs at raised priority, the other does its ds_reads and its LDS-DMA issue, then they swap;
//   * a K tile (64 k) is staged as four 16 KiB half tiles (pixel halves A0/A1, channel halves B0/B1), two LDS buffers;
//     each phase issues ONE half tile (2 DMA wave-instructions per wave: one beside the fragment reads, one from the
//     MFMA shadow) two phases after the region's last read, and there is ONE counted `s_waitcnt vmcnt(3)` per K tile
//     (phase 3) — never 0 inside the loop;
//   * read order per K tile: (A0,B0) -> B1 -> A1 -> B0 again; the A fragments of a pixel half stay in registers for two
//     phases, so a K tile costs 28 ds_read_b128 per wave for 64 MFMAs.
// Hazards (LDS-DMA is ordered for a ds_read only by the issuing waves' vmcnt followed by a barrier the reader passed):
// with the groups one barrier apart every wave has executed the phase-3 wait before barrier instance 8t+8, group 0
// reads K tile t+1 after 8t+8 and group 1 after 8t+9.  A region is re-staged two phases (four barrier instances)
// after its last read by either group (schedule 2: one phase, but only from the matrix section, i.e. after the barrier
// that the other group's consuming MFMAs precede).
//
// Same operand layout, swizzle, zero page, XCD remap, grouped-launch segments and epilogue as conv_igemm_dma.hip.
#include "osd_common.h"
#include "conv_params.h"
#include "conv_epilogue.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned g_zero_page_p8[64];

template <int N> __device__ __forceinline__ void p8_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void p8_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

struct KState { int kr, ks, kc; };

#ifdef OSD_P8_NO_DMA      // diagnostic: timing of the loop without its LDS-DMA issue (results are garbage)
#define P8_LOOP_DMA(x)
#else
#define P8_LOOP_DMA(x) x
#endif

#ifdef OSD_P8_STAMPS
// diagnostic build only (never the shipped library): cycles spent per section, summed over the K loop, for waves 0 and 4
// of workgroup 0: [wave/4][0..3] = loads+issue, wait at the first barrier, MFMA issue, wait at the second barrier
__device__ unsigned long long g_p8_stamps[2][4];
#define P8_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define P8_ACC(slot, a, b) st_acc[slot] += (b) - (a)
#else
#define P8_STAMP(var)
#define P8_ACC(slot, a, b)
#endif

__global__ void __launch_bounds__(512) conv_p8_kernel(ConvKParams p) {
  typedef __bf16 T;
  constexpr int KB = 128;                 // bytes of K per tile row
  constexpr int BKE = 64;                 // bf16 elements of K per tile
  constexpr int EPC = 8;
  constexpr int BM = 256, BN = 256;
  constexpr int HALF = 128 * KB;          // one half tile: 128 rows x 128 B
  constexpr int BUF = 4 * HALF;           // A0 | A1 | B0 | B1
  constexpr int TM = 8, TN = 4;           // 16-pixel / 16-channel MFMA tiles per wave

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  int t;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = t % p.tilesN;
  int tile_m = t / p.tilesN;
  const ConvView q = conv_select_view(p, tile_m);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(q.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(q.w);
  const T* zero = reinterpret_cast<const T*>(g_zero_page_p8) + (lane & 7) * EPC;

  // ---- per-lane DMA coordinates.  A wave moves pieces 2*wave, 2*wave+1 (8 rows x 128 B each) of every half tile.
  // half-tile row r -> tile row:  A_h: (r >> 6) * 128 + h * 64 + (r & 63)     B_h: (r >> 5) * 64 + h * 32 + (r & 31)
  const int lrow = lane >> 3, lpos = lane & 7;
  const T* a_base[2][2];
  int a_hi0[2][2], a_wi0[2][2];
  const T* b_base[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + lrow;
      const int sw = (r >> 1) & 7;
      const int arow = (r >> 6) * 128 + h * 64 + (r & 63);
      const int m = m0 + arow;
      a_base[h][i] = zero;
      a_hi0[h][i] = -0x40000000;
      a_wi0[h][i] = 0;
      if (m < q.M) {
        const int n_img = m / q.HoWo;
        const int rem = m - n_img * q.HoWo;
        const int ho = rem / q.Wo;
        const int wo = rem - ho * q.Wo;
        a_base[h][i] = xg + (size_t)n_img * q.sN + (lpos ^ sw) * EPC;
        a_hi0[h][i] = ho * p.sh - p.ph;
        a_wi0[h][i] = wo * p.sw - p.pw;
      }
      const int brow = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
      b_base[h][i] = brow < p.w_rows ? wg + (size_t)brow * p.Ktot + (lpos ^ sw) * EPC : nullptr;
    }
  }

  // one DMA piece (i = 0 / 1) of a pixel half tile / a channel half tile
  auto issue_a = [&](int h, int buf, const KState& k, bool valid, int i) {
    const int hi = a_hi0[h][i] + k.kr, wi = a_wi0[h][i] + k.ks;
    const bool ok = valid && ((unsigned)hi < (unsigned)q.H) && ((unsigned)wi < (unsigned)q.W);
    const T* src = ok ? a_base[h][i] + (hi * q.sH + wi * p.sW + k.kc) : zero;
#if defined(OSD_P8_CHEAP_ADDR)      // diagnostic: every piece reads the zero page (no misses, 1 line per piece)
    src = zero;
#elif defined(OSD_P8_SAME_ADDR)     // diagnostic: real address arithmetic, but always K tile 0's rows (cached)
    src = ok ? a_base[h][i] + (hi * q.sH + wi * p.sW) : zero;
#endif
    p8_dma16(src, lds0 + buf * BUF + h * HALF + (wave * 2 + i) * 1024);
  };
  auto issue_b = [&](int h, int buf, int ktile, bool valid, int i) {
    const T* src = (valid && b_base[h][i]) ? b_base[h][i] + ktile * BKE : zero;
#if defined(OSD_P8_CHEAP_ADDR)
    src = zero;
#elif defined(OSD_P8_SAME_ADDR)
    src = (valid && b_base[h][i]) ? b_base[h][i] : zero;
#endif
    p8_dma16(src, lds0 + buf * BUF + (2 + h) * HALF + (wave * 2 + i) * 1024);
  };
  auto advance = [&](KState& k) {
    k.kc += BKE;
    if (k.kc >= p.Cin) {
      k.kc = 0;
      if (++k.ks >= p.S) { k.ks = 0; ++k.kr; }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fkq = lane >> 4;
  // fragment reads: 16 rows x 16 B per ds_read_b128, chunk ^ ((row >> 1) & 7) as written by the DMA
  auto read_a = [&](uint4 (&xa)[2][4], int buf, int h) {
    const char* base = smem + buf * BUF + h * HALF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wm * 64 + j * 16 + frow;
        xa[kk][j] = *reinterpret_cast<const uint4*>(base + row * KB + (((kk * 4 + fkq) ^ ((row >> 1) & 7)) << 4));
      }
  };
  auto read_b = [&](uint4 (&wb)[2][2], int buf, int h) {
    const char* base = smem + buf * BUF + (2 + h) * HALF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wn * 32 + i * 16 + frow;
        wb[kk][i] = *reinterpret_cast<const uint4*>(base + row * KB + (((kk * 4 + fkq) ^ ((row >> 1) & 7)) << 4));
      }
  };
  // 16 MFMAs of quadrant (pixel half ph, channel half ch), bracketed by the two barriers of a phase
#ifdef OSD_P8_STAMPS
  unsigned long long st_acc[4] = {0, 0, 0, 0};
  unsigned long long st_phase = __builtin_amdgcn_s_memtime();
#endif
  // one phase's matrix section: 16 MFMAs of quadrant (pixel half ph, channel half ch) between the two barriers, with
  // the phase's two DMA pieces issued from the MFMA shadow (an MFMA holds the issue port for half of its 16 cycles; the
  // same two instructions issued beside the ds_reads cost the loading group ~220 cycles per phase and made it the
  // critical path)
  auto quadrant = [&](const uint4 (&xa)[2][4], const uint4 (&wb)[2][2], int ph, int ch, auto&& dma, int inside = 3) {
    P8_STAMP(s0);
    P8_ACC(0, st_phase, s0);
    __builtin_amdgcn_s_barrier();
    P8_STAMP(s1);
    P8_ACC(1, s0, s1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[ch * 2 + i][ph * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              *reinterpret_cast<const bf16x8*>(&wb[kk][i]), *reinterpret_cast<const bf16x8*>(&xa[kk][j]),
              acc[ch * 2 + i][ph * 4 + j], 0, 0, 0);
        if (kk == 0 && ((inside >> i) & 1)) {
          __builtin_amdgcn_sched_barrier(0);
          dma(i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    P8_STAMP(s2);
    P8_ACC(2, s1, s2);
    __builtin_amdgcn_s_barrier();
#ifdef OSD_P8_STAMPS
    st_phase = __builtin_amdgcn_s_memtime();
    P8_ACC(3, s2, st_phase);
#endif
  };

  const int KT = p.KT;
#if !defined(OSD_P8_SCHEDULE) || OSD_P8_SCHEDULE == 1
  // ---- schedule 1 ("split", default): regions re-staged TWO phases after their last read, so a piece may be issued from
  // the fragment-read section; piece 0 of a phase goes there, piece 1 into the MFMA section (balances the two sections:
  // 843 TFLOP/s on the 3x3 256->256 P3 conv; schedule 2 below, both pieces in the MFMA section one phase after the last
  // read, 772; both pieces in the read section 834)
  KState k1{0, 0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(0, 0, k1, true, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_b(1, 0, 0, true, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(1, 0, k1, true, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_b(0, 0, 0, true, i);
  advance(k1);                                  // k1 = K tile 1
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(0, 1, k1, 1 < KT, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_b(1, 1, 1, 1 < KT, i);
  KState k2 = k1;
  advance(k2);                                  // k2 = K tile 2
  p8_wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();

  // per K tile kt: phase 0 stages A1 of kt+1, phase 1 B0 of kt+1, phase 2 A0 of kt+2, phase 3 B1 of kt+2; the wait in
  // phase 3's read section leaves A0(kt+2) and the first piece of B1(kt+2) in flight (3 loads)
  uint4 xa[2][4], wb0[2][2], wb1[2][2];
  for (int kt = 0; kt < KT; ++kt) {
    const int b = kt & 1;
    const bool v1 = kt + 1 < KT, v2 = kt + 2 < KT;
    auto d0 = [&](int i) { P8_LOOP_DMA(issue_a(1, b ^ 1, k1, v1, i)); };
    auto d1 = [&](int i) { P8_LOOP_DMA(issue_b(0, b ^ 1, kt + 1, v1, i)); };
    auto d2 = [&](int i) { P8_LOOP_DMA(issue_a(0, b, k2, v2, i)); };
    auto d3 = [&](int i) { P8_LOOP_DMA(issue_b(1, b, kt + 2, v2, i)); };
    read_a(xa, b, 0);
    read_b(wb0, b, 0);
    d0(0);
    quadrant(xa, wb0, 0, 0, d0, 2);
    read_b(wb1, b, 1);
    d1(0);
    quadrant(xa, wb1, 0, 1, d1, 2);
    read_a(xa, b, 1);
    d2(0);
    quadrant(xa, wb1, 1, 1, d2, 2);
    read_b(wb0, b, 0);
    d3(0);
    p8_wait_vmcnt<3>();
    quadrant(xa, wb0, 1, 0, d3, 2);
    advance(k1);
    advance(k2);
  }
#else
  // ---- prologue: K tile 0 completely, then A0 / B1 / A1 of K tile 1 (issue order = retirement order of the waits)
  KState k1{0, 0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(0, 0, k1, true, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_b(1, 0, 0, true, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(1, 0, k1, true, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_b(0, 0, 0, true, i);
  advance(k1);                                  // k1 = K tile 1
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(0, 1, k1, 1 < KT, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_b(1, 1, 1, 1 < KT, i);
#pragma unroll
  for (int i = 0; i < 2; ++i) issue_a(1, 1, k1, 1 < KT, i);
  KState k2 = k1;
  advance(k2);                                  // k2 = K tile 2
  p8_wait_vmcnt<6>();                           // K tile 0 has landed (this wave's part)
  __builtin_amdgcn_s_barrier();                 // ... every wave's part
  if (wm == 1) __builtin_amdgcn_s_barrier();    // group 1 runs one barrier behind group 0 from here on

  // Per K tile kt (buffer b = kt & 1) the matrix sections issue, one phase after the region's last read:
  //   phase 0: B0 of kt+1 (buffer b^1)   phase 1: A0 of kt+2 (buffer b)   phase 2: B1 of kt+2   phase 3: A1 of kt+2
  // and the fragment-read section of phase 3 holds the K loop's only wait: vmcnt(4) leaves A0/B1 of kt+2 in flight and
  // retires everything of K tile kt+1.  Every wave executes it before its first phase-3 barrier, i.e. before barrier
  // instance 8kt+8; group 0 reads K tile kt+1 after instance 8kt+8, group 1 after 8kt+9.
  uint4 xa[2][4], wb0[2][2], wb1[2][2];
  for (int kt = 0; kt < KT; ++kt) {
    const int b = kt & 1;
    const bool v1 = kt + 1 < KT, v2 = kt + 2 < KT;
    // phase 0: quadrant (0, 0)
    read_a(xa, b, 0);
    read_b(wb0, b, 0);
    quadrant(xa, wb0, 0, 0, [&](int i) { P8_LOOP_DMA(issue_b(0, b ^ 1, kt + 1, v1, i)); });
    // phase 1: quadrant (0, 1)
    read_b(wb1, b, 1);
    quadrant(xa, wb1, 0, 1, [&](int i) { P8_LOOP_DMA(issue_a(0, b, k2, v2, i)); });
    // phase 2: quadrant (1, 1)
    read_a(xa, b, 1);
    quadrant(xa, wb1, 1, 1, [&](int i) { P8_LOOP_DMA(issue_b(1, b, kt + 2, v2, i)); });
    // phase 3: quadrant (1, 0); K tile kt+1 must have landed before anyone passes this phase's barriers
    read_b(wb0, b, 0);
    p8_wait_vmcnt<4>();
    quadrant(xa, wb0, 1, 0, [&](int i) { P8_LOOP_DMA(issue_a(1, b, k2, v2, i)); });
    advance(k2);
  }
#endif
#ifdef OSD_P8_STAMPS
  if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0)
    for (int i = 0; i < 4; ++i) g_p8_stamps[wave >> 2][i] = st_acc[i];
#endif
  if (wm == 0) __builtin_amdgcn_s_barrier();    // pairs with group 1's last barrier
  p8_wait_vmcnt<0>();                           // the tail's dummy DMA writes must not land on the staging area
  __syncthreads();
  conv_epilogue<T, TM, TN>(acc, p, q, smem, wave, wm, wn, lane, m0, n0);
}

}  // namespace

int osd_conv_p8_launch(const ConvKParams& pin, hipStream_t stream) {
  ConvKParams p = pin;
  constexpr int BM = 256, BN = 256, BKE = 64;
  if (p.Cin % BKE != 0) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(p8): cin %d not a multiple of %d", p.Cin, BKE);
  if (p.relu_in) return osd_fail(OSD_ERR_UNSUPPORTED, "conv(p8): relu_in prologue not supported");
  p.tilesM = cdiv(p.M, BM);
  if (p.n_seg > 0) {
    p.tilesM = 0;
    for (int i = 0; i < p.n_seg; ++i) {
      p.seg[i].tile_begin = p.tilesM;
      p.tilesM += cdiv(p.seg[i].M, BM);
    }
  }
  p.tilesN = cdiv(p.Cout, BN);
  p.KT = p.Ktot / BKE;
  constexpr int lds = 2 * 4 * 128 * 128;       // 128 KiB ring; the epilogue staging (8 waves x 32 rows x 272 B) fits inside
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const long long nblocks = (long long)p.tilesM * p.tilesN;
  if (nblocks <= 0 || nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv(p8): bad grid");
  hipLaunchKernelGGL(conv_p8_kernel, dim3((unsigned)nblocks), dim3(512), lds, stream, p);
  return osd_check_launch("conv_igemm_p8");
}

#ifdef OSD_P8_STAMPS
extern "C" int osd_debug_p8_stamps(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_p8_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
