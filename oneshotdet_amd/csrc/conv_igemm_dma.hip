// conv_igemm_dma — second-generation implicit-GEMM convolution for gfx950: same GEMM view, tile geometry, swizzled
// LDS image and swapped MFMA operands as conv_igemm.hip, but
//   * operands go HBM/L2 -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction) with no VGPR
//     staging and no ds_write pass.  The DMA destination is lane-linear (M0 base + 16*lane), so the XOR swizzle is
//     applied to the per-lane SOURCE chunk (linear dest + permuted source + same permutation on the ds_read, rule 21
//     of the CDNA guide); the im2col gather is simply the per-lane source address; zero padding, the M tail and the
//     Cout tail read from a zero page in device memory;
//   * an NST-deep LDS ring with NST-1 stages in flight, ONE raw s_barrier per K stage, counted `s_waitcnt vmcnt(N)`
//     (never 0 in steady state).  The DMA is issued from inline asm so hipcc does not fence every ds_read with
//     vmcnt(0) (it does for the builtin when it cannot prove the LDS addresses disjoint);
//   * the epilogue stages the fp32 accumulator tile through LDS and writes NHWC rows with 16-byte-per-lane, fully
//     coalesced stores (and 16-byte residual loads): one output pixel's BN channels are one contiguous run.
#include "osd_common.h"
#include <stdlib.h>
#include "conv_params.h"
#include "conv_epilogue.h"
#include <type_traits>
#ifndef OSD_DMA_FRONT
#define OSD_DMA_FRONT 1   // issue the next stage's DMA in the first half of the MFMA groups (more time to land)
#endif

namespace {

__device__ __attribute__((aligned(256))) unsigned g_zero_page[64];   // 256 B of zeros: source of every padded chunk

template <int KB> __device__ __forceinline__ int swz_g(int row) {
  if constexpr (KB == 64) return (-(row >> 2)) & 3;
  else return (row >> 1) & 7;
}
template <int KB> __device__ __forceinline__ int swz_off(int row, int chunk) {
  return row * KB + ((chunk ^ swz_g<KB>(row)) << 4);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one LDS-DMA wave-instruction: every lane moves 16 bytes from its own global address to lds_dst + 16*lane
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ uint4 relu_frag(uint4 v, float) {
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = fmaxf(f[i], 0.f);
  return v;
}
__device__ __forceinline__ uint4 relu_frag(uint4 v, __bf16) {
  unsigned* u = reinterpret_cast<unsigned*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] &= ~(((u[i] >> 15) & 0x00010001u) * 0xFFFFu);
  return v;
}

template <typename T, int BM, int BN, int KB, int WM, int WN, int NST, bool RELU_IN, bool SRC2>
__global__ void __launch_bounds__(64 * WM * WN) conv_dma_kernel(ConvKParams p) {
  // p lives in the kernarg segment; the per-segment fields are read through the q_* locals below
  constexpr int NWV = WM * WN;            // waves per workgroup: 4 (256 threads) or 8 (512 threads, the 256x256 tile)
  constexpr int CH = KB / 16;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BKE = KB / (int)sizeof(T);
  constexpr int RPI = 1024 / KB;          // tile rows covered by one wave-instruction
  constexpr int IA = BM / RPI, IB = BN / RPI;
  static_assert(IA % NWV == 0, "pixel tile must give every wave the same number of DMA instructions");
  constexpr int PA = IA / NWV;
  constexpr int PB = IB >= NWV ? IB / NWV : 1;
  static_assert(IB >= NWV ? (IB % NWV == 0) : (NWV % IB == 0), "weight tile / wave split");
  constexpr int LPS = PA + PB;            // DMA instructions per wave per stage
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int STAGE = (BM + BN) * KB;
  static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;   // LDS byte offset of the dynamic segment (low 32 bits of the flat address)

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
  const void* q_x = q.x;
  const int q_H = q.H, q_W = q.W, q_Wo = q.Wo, q_M = q.M, q_sN = q.sN, q_sH = q.sH;
  const int q_HoWo = q.HoWo;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(q_x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(q.w);
  const T* zero = reinterpret_cast<const T*>(g_zero_page) + (lane & 15) * EPC;

  // ---- per-lane DMA source coordinates ----
  const int lrow = lane / CH, lpos = lane % CH;
  const T* a_base[PA];
  const T* a_base2[SRC2 ? PA : 1];      // second source (1x1 convs only): the lane's pixel in x2, channel chunk included
  int a_hi0[PA], a_wi0[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave * PA + i) * RPI + lrow;
    const int m = m0 + row;
    a_base[i] = zero;
    if constexpr (SRC2) a_base2[i] = zero;
    a_hi0[i] = -0x40000000;
    a_wi0[i] = 0;
    if (m < q_M) {
      const int n_img = m / q_HoWo;
      const int rem = m - n_img * q_HoWo;
      const int ho = rem / q_Wo;
      const int wo = rem - ho * q_Wo;
      a_base[i] = xg + (size_t)n_img * q_sN + (lpos ^ swz_g<KB>(row)) * EPC;
      a_hi0[i] = ho * p.sh - p.ph;
      a_wi0[i] = wo * p.sw - p.pw;
      if constexpr (SRC2)
        a_base2[i] = reinterpret_cast<const T*>(p.x2) + (size_t)n_img * p.x2_sN + (size_t)(ho * p.st2) * p.x2_sH +
                     (size_t)(wo * p.st2) * p.x2_sW + (lpos ^ swz_g<KB>(row)) * EPC;
    }
  }
  const T* b_ptr[PB];
  const T* b_ptr2[SRC2 ? PB : 1];       // two sources with separate weight buffers: the row's start in the second one
  int b_step[PB];
  int b_instr[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int jb = IB >= NWV ? wave * PB + i : wave % IB;   // tiles narrower than NWV instructions: waves duplicate a load
    const int row = jb * RPI + lrow;
    const bool ok = n0 + row < p.w_rows;
    b_instr[i] = jb;
    b_ptr[i] = ok ? wg + (size_t)(n0 + row) * p.Ktot + (lpos ^ swz_g<KB>(row)) * EPC : zero;
    b_step[i] = ok ? BKE : 0;
    if constexpr (SRC2)
      b_ptr2[i] = (ok && p.w2) ? reinterpret_cast<const T*>(p.w2) + (size_t)(n0 + row) * (p.Cin - p.cin1) + (lpos ^ swz_g<KB>(row)) * EPC
                               : zero;
  }

  int kr = 0, ks = 0, kc = 0;
  // one DMA instruction of the stage being fetched: j < PA -> pixel operand, else weight operand
  auto issue_one = [&](int buf, int j) {
    const unsigned xs = lds0 + buf * STAGE;
    const unsigned ws = xs + BM * KB;
    if (j < PA) {
      const int hi = a_hi0[j] + kr, wi = a_wi0[j] + ks;
      const bool ok = ((unsigned)hi < (unsigned)q_H) && ((unsigned)wi < (unsigned)q_W);
#ifdef OSD_CD_CHEAP_ADDR      // diagnostic builds (results are garbage): every piece reads the always-cached zero page
      const T* src = zero;
#else
      const T* src = ok ? a_base[j] + (hi * q_sH + wi * p.sW + kc) : zero;
#endif
      if constexpr (SRC2) {
        if (kc >= p.cin1) src = a_base2[j] + (a_base2[j] == zero ? 0 : kc - p.cin1);      // uniform branch: K past source 1
      }
      dma16(src, xs + (wave * PA + j) * 1024);
    } else {
      const int i = j - PA;
#ifdef OSD_CD_CHEAP_ADDR
      dma16(zero, ws + b_instr[i] * 1024);
#else
      dma16(b_ptr[i], ws + b_instr[i] * 1024);
#endif
      b_ptr[i] += b_step[i];
    }
  };
  auto advance_k = [&]() {
    kc += BKE;
    if constexpr (SRC2) {
      if (p.w2 != nullptr && kc == p.cin1) {      // the K range of the second source starts: its own weight rows from here on
#pragma unroll
        for (int i = 0; i < PB; ++i) b_ptr[i] = b_ptr2[i];
      }
    }
    if (kc >= p.Cin) {
      kc = 0;
      if (++ks >= p.S) { ks = 0; ++kr; }
    }
  };
  auto issue_stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < LPS; ++j) issue_one(buf, j);
    advance_k();
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fkq = lane >> 4;

  // MFMAs of stage `buf`; when `fetch` is set the LPS DMA instructions of the next stage are spread between the MFMA
  // row groups instead of being issued as one burst after the barrier (all waves leave the barrier together, so a
  // burst would leave the matrix pipe idle on every SIMD at the same time)
  constexpr int SLOTS = (KB / 64) * TN;
  // (measured: spreading helps the 8-wave 256x256 tile, 877 -> 939 TFLOP/s; the 4-wave tiles run two workgroups per CU
  // that cover each other's bursts and are faster with the burst, 760 vs 635 TFLOP/s)
  constexpr bool SPREAD = NWV == 8;
  auto compute_stage = [&](int buf, int nbuf, auto fetch_tag) {
    constexpr bool fetch = decltype(fetch_tag)::value && SPREAD;
    if constexpr (decltype(fetch_tag)::value && !SPREAD) issue_stage(nbuf);
    const char* xs = smem + buf * STAGE;
    const char* ws = xs + BM * KB;
    int issued = 0;
#pragma unroll
    for (int kb = 0; kb < KB / 64; ++kb) {
      uint4 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wf[i] = *reinterpret_cast<const uint4*>(ws + swz_off<KB>((wn * TN + i) * 16 + frow, kb * 4 + fkq));
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        xf[j] = *reinterpret_cast<const uint4*>(xs + swz_off<KB>((wm * TM + j) * 16 + frow, kb * 4 + fkq));
        if constexpr (RELU_IN) xf[j] = relu_frag(xf[j], T());   // P7 = conv(relu(P6)), fpn.py:98
      }
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
        // DMA instructions due by the end of this slot
        const int slot = kb * TN + i;
        const int due = OSD_DMA_FRONT ? min(LPS, ((slot + 1) * LPS * 2) / SLOTS) : ((slot + 1) * LPS) / SLOTS;
        if constexpr (fetch) {
#pragma unroll
          for (int j = 0; j < LPS; ++j)
            if (j >= issued && j < due) issue_one(nbuf, j);
        }
        issued = due;
      }
    }
    if constexpr (fetch) advance_k();
  };

  // ---- main loop: NST-1 stages in flight, one barrier per stage ----
  const int KT = p.KT;
#ifdef OSD_CD_STAMPS      // diagnostic build: per-wave cycle stamps into the buffer passed as act_scale_dev (tools/cd_stamps.py)
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long st_t1 = 0;
#endif
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) issue_stage(s);
  int cur = 0, nxt = NST - 1;
  int kt = 0;
  for (; kt + NST - 1 < KT; ++kt) {                        // steady state: fetch stage kt+NST-1 while computing stage kt
    wait_vmcnt<LPS * (NST - 2)>();                         // stage kt has landed; later stages stay in flight
    __builtin_amdgcn_s_barrier();                          // every wave's part of stage kt is visible; buffer `nxt` is free
#ifdef OSD_CD_STAMPS
    if (kt == 0) st_t1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef OSD_CD_NO_DMA             // diagnostic: the loop without its operand fetch
    compute_stage(cur, nxt, std::false_type());
#else
    compute_stage(cur, nxt, std::true_type());
#endif
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
  for (; kt < KT; ++kt) {                                  // drain: nothing left to fetch
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#ifdef OSD_CD_STAMPS
    if (st_t1 == 0) st_t1 = __builtin_amdgcn_s_memtime();
#endif
    compute_stage(cur, nxt, std::false_type());
    cur = cur + 1 == NST ? 0 : cur + 1;
  }

  // ---- epilogue (conv_epilogue.h): LDS-staged, 16-byte-per-lane NHWC stores with bias / residual / mask / activation
#ifdef OSD_CD_STAMPS
  const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef OSD_CD_NO_EPI             // diagnostic: no output (keeps the accumulators alive)
  if (acc[0][0][0] != 12345.678f) return;
#endif
  conv_epilogue<T, TM, TN, false, 0, (TM * TN <= 16)>(acc, p, q, smem, wave, wm, wn, lane, m0, n0);
#ifdef OSD_CD_STAMPS
  if (p.act != OSD_ACT_EXP_SCALE && p.act_scale_dev != nullptr) {
    const unsigned long long st_t3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_t4 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.act_scale_dev)) + ((size_t)blockIdx.x * NWV + wave) * 8;
      o[0] = st_t0; o[1] = st_t1 - st_t0; o[2] = st_t2 - st_t1; o[3] = st_t3 - st_t2; o[4] = st_t4 - st_t3;
      o[5] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (8 << 6) | 20);      // HW_ID: cu id bits [11:8]
    }
  }
#endif
}

template <typename T, int BM, int BN, int KB, int WM, int WN, int NST, bool RELU_IN = false, bool SRC2 = false>
int launch_dma(const ConvKParams& pin, hipStream_t stream) {
  ConvKParams p = pin;
  p.tilesM = cdiv(p.M, BM);
  if (p.n_seg > 0) {
    p.tilesM = 0;
    for (int i = 0; i < p.n_seg; ++i) {
      p.seg[i].tile_begin = p.tilesM;
      p.tilesM += cdiv(p.seg[i].M, BM);
    }
  }
  p.tilesN = cdiv(p.Cout, BN);
  constexpr int BKE = KB / (int)sizeof(T);
  if (p.Cin % BKE != 0) return osd_fail(OSD_ERR_UNSUPPORTED, "conv: cin %d not a multiple of %d", p.Cin, BKE);
  p.KT = (p.x2 != nullptr ? p.Cin : p.Ktot) / BKE;      // two sources (1x1): K = both parts, Ktot = row stride of w
  constexpr int ring = NST * (BM + BN) * KB;
  constexpr int TMx = BM / WM / 16, TNx = BN / WN / 16;
  constexpr int npass = TMx >= 8 ? TMx / 2 : (TMx >= 2 ? 2 : 1);
  constexpr int stagec = WM * WN * ((TMx / npass) * 16) * (TNx * 16 * 4 + 16);
  constexpr int lds = ring > stagec ? ring : stagec;
  auto kern = conv_dma_kernel<T, BM, BN, KB, WM, WN, NST, RELU_IN, SRC2>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const long long nblocks = (long long)p.tilesM * p.tilesN;
  if (nblocks <= 0 || nblocks > 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad grid");
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * WM * WN), lds, stream, p);
  return osd_check_launch("conv_igemm_dma");
}

template <typename T, int KB, int NST>
int dispatch_tile_dma(int tile, const ConvKParams& p, hipStream_t s) {
  if (p.x2 != nullptr) {        // two pixel sources (1x1, single problem): three tiles are built for it
    if (p.relu_in || p.n_seg > 0 || p.R != 1 || p.S != 1 || p.ph != 0 || p.pw != 0)
      return osd_fail(OSD_ERR_UNSUPPORTED, "conv: a second source needs a plain 1x1 conv");
    switch (tile) {
      case 0: return launch_dma<T, 128, 128, KB, 2, 2, NST, false, true>(p, s);
      case 2: return launch_dma<T, 64, 64, KB, 2, 2, NST, false, true>(p, s);
      case 7:
        if constexpr (NST * 384 * KB <= 155648) return launch_dma<T, 256, 128, KB, 4, 2, NST, false, true>(p, s);
        else return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the 256x128 tile does not fit LDS with this ring");
    }
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv: tile %d is not built for two sources (0, 2, 7 are)", tile);
  }
  if (p.relu_in) return launch_dma<T, 64, 64, KB, 2, 2, NST, true>(p, s);   // only the tiny P7 conv uses it
  switch (tile) {
    case 0: return launch_dma<T, 128, 128, KB, 2, 2, NST>(p, s);
    case 1: return launch_dma<T, 128, 64, KB, 4, 1, NST>(p, s);
    case 2: return launch_dma<T, 64, 64, KB, 2, 2, NST>(p, s);
    case 3: return launch_dma<T, 256, 16, KB, 4, 1, NST>(p, s);
    case 4:   // 256 x 256, 8 waves: half the operand bytes per MFMA of the 128 x 128 tile (the L1/TA path is the bound there)
      if constexpr (NST * 512 * KB <= 131072) return launch_dma<T, 256, 256, KB, 2, 4, NST>(p, s);
      else return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the 256x256 tile does not fit LDS with this ring");
    case 5:   // 128 pixels x 256 channels, 8 waves (2 x 4, 64 x 64 per wave; round 5, the id of the retired ping-pong kernel): the
              // reducing 1x1 convs of the bottlenecks (K = 4 N: 1024 -> 256, 2048 -> 512) stream the whole pixel operand ONCE per
              // tile row when one tile spans all of N = 256, where the 256 x 128 tile reads it twice; at M = 25,600 still 200
              // workgroups.  Operand bytes per 128 pixels: 256 + 512 KB against 2 x (256 + 256) for the 128 x 128 tile
      if constexpr (NST * 384 * KB <= 155648) return launch_dma<T, 128, 256, KB, 2, 4, NST>(p, s);
      else return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the 128x256 tile does not fit LDS with this ring");
    case 7:   // 256 pixels x 128 channels, 8 waves (4 x 2, 64 x 64 per wave): twice the workgroups of the 256 x 256 tile for
              // the layers whose pixel count gives that tile only 50-100 workgroups on 256 CUs (layer3 / layer4 at bs = 8)
      if constexpr (NST * 384 * KB <= 155648) return launch_dma<T, 256, 128, KB, 4, 2, NST>(p, s);
      else return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the 256x128 tile does not fit LDS with this ring");
  }
  return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad tile id %d", tile);
}

}  // namespace

// variant: 0 = deep ring (bf16 KB128 x3 / fp32 KB64 x4), 1 = shallow ring (x2: half the LDS, twice the blocks per CU),
//          2 = short stages (bf16 KB64 x4 / fp32 KB64 x3), 3 = short stages x 2 (bf16; fp32 = 1): a quarter of the deep
//          ring's LDS — for the HBM-bound 1x1 convs whose K is 2-4 stages long, where workgroups in flight per CU (latency
//          of the first operand fetch and of the residual read in the epilogue) matter and the ring depth does not
int osd_conv_dma_dispatch(int dtype, int tile, int variant, const ConvKParams& p, hipStream_t s) {
  if (dtype == OSD_F32) {
    if (variant == 1 || variant == 3) return dispatch_tile_dma<float, 64, 2>(tile, p, s);
    if (variant == 2) return dispatch_tile_dma<float, 64, 3>(tile, p, s);
    return dispatch_tile_dma<float, 64, 4>(tile, p, s);
  }
  // variant 3: short stages (32 K elements), three deep: 48 KB for the 128 x 128 tile = three workgroups per CU.  (Two deep
  // until round 4: the third stage costs no occupancy on any tile and is faster on every 1x1 shape of the step, 1 - 8 %:
  // tools/pw_bench.py)
  if (variant == 3) return dispatch_tile_dma<__bf16, 64, 3>(tile, p, s);
  if (p.Cin % 64 != 0 || variant == 2) return dispatch_tile_dma<__bf16, 64, 4>(tile, p, s);
  if (variant == 1) return dispatch_tile_dma<__bf16, 128, 2>(tile, p, s);
  return dispatch_tile_dma<__bf16, 128, 3>(tile, p, s);
}

// Round 6: the 64 x 64 tile with a DEEP ring (algos 58 / 59: five / eight 16 KB stages = 64 / 112 KB of operands in flight per
// workgroup).  For the latency-sized launches of the query backbone (M = 128 .. 2,048 pixels at bs 8: 16 - 64 workgroups that each
// stream a 64-row slice of a weight matrix which is COLD inside the step — ~1 GB of parameter state passes between two uses of a
// weight).  Round 4 dropped this form on a warm benchmark loop (14.5 vs 15.0 us); with cold weights the two-stage ring the warm
// timing prefers is the SLOWEST candidate (31 us against 20 - 23 for the deep rings, tools/small_m_cold.py), so the tuner now times
// these shapes cold (tuner._time_launches_cold).
int osd_conv_dma_deep(int dtype, int nst, const ConvKParams& p, hipStream_t s) {
  if (dtype != OSD_BF16 || p.Cin % 64 != 0 || p.x2 != nullptr || p.relu_in)
    return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the deep-ring 64 x 64 tile is bf16, cin in 64s, one source, no relu_in");
  if (nst == 8) return launch_dma<__bf16, 64, 64, 128, 2, 2, 8>(p, s);
  if (nst == 5) return launch_dma<__bf16, 64, 64, 128, 2, 2, 5>(p, s);
  // algos 57 / 60: 64 x 32 (pixels x channels; 4 x 1 waves) and 32 x 64 (2 x 2 waves) tiles, eight stages — twice the workgroups of the
  // 64 x 64 tile; the 64 x 32 one streams HALF the (cold) weight rows per workgroup: the per-CU intake from HBM (~23 GB/s) is what
  // bounds these launches, so more CUs pulling is what helps
  if (nst == 832) return launch_dma<__bf16, 64, 32, 128, 4, 1, 8>(p, s);
  if (nst == 816) return launch_dma<__bf16, 32, 64, 128, 2, 2, 8>(p, s);
  return osd_fail(OSD_ERR_INVALID_ARG, "conv: deep ring of %d stages not built", nst);
}
