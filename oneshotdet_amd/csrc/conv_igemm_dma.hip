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
#include "conv_params.h"
#include <type_traits>
#include <cstddef>
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

template <typename T, int BM, int BN, int KB, int WM, int WN, int NST, bool RELU_IN>
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
  // grouped launch: the pixel tile selects its (x, y) pair; everything per-pair is uniform (SGPRs)
  const void* q_x = p.x; void* q_y = p.y; const void* q_res = p.res; const void* q_mask = p.mask;
  const float* q_scale_dev = p.act_scale_dev;
  int q_H = p.H, q_W = p.W, q_Ho = p.Ho, q_Wo = p.Wo, q_M = p.M, q_sN = p.sN, q_sH = p.sH;
  if (p.n_seg > 0) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < kConvMaxSeg; ++i)
      if (i < p.n_seg && tile_m >= p.seg[i].tile_begin) si = i;
    // read the chosen entry straight from the kernarg segment (constant address space, scalar loads): indexing the
    // by-value struct dynamically would make the compiler copy all of it to scratch
    typedef const __attribute__((address_space(4))) char* kptr;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvKParams, seg) + si * (int)sizeof(ConvSeg);
#define OSD_KSEG(type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(base + offsetof(ConvSeg, field)))
    typedef unsigned long long u64;
    q_x = (const void*)OSD_KSEG(u64, x); q_y = (void*)OSD_KSEG(u64, y); q_res = (const void*)OSD_KSEG(u64, res);
    q_mask = (const void*)OSD_KSEG(u64, mask); q_scale_dev = (const float*)OSD_KSEG(u64, act_scale_dev);
    q_H = OSD_KSEG(int, H); q_W = OSD_KSEG(int, W); q_Ho = OSD_KSEG(int, Ho); q_Wo = OSD_KSEG(int, Wo);
    q_M = OSD_KSEG(int, M); q_sN = OSD_KSEG(int, sN); q_sH = OSD_KSEG(int, sH);
    tile_m -= OSD_KSEG(int, tile_begin);
#undef OSD_KSEG
  }
  const int q_HoWo = q_Ho * q_Wo;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(q_x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const T* zero = reinterpret_cast<const T*>(g_zero_page) + (lane & 15) * EPC;

  // ---- per-lane DMA source coordinates ----
  const int lrow = lane / CH, lpos = lane % CH;
  const T* a_base[PA];
  int a_hi0[PA], a_wi0[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave * PA + i) * RPI + lrow;
    const int m = m0 + row;
    a_base[i] = zero;
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
    }
  }
  const T* b_ptr[PB];
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
  }

  int kr = 0, ks = 0, kc = 0;
  // one DMA instruction of the stage being fetched: j < PA -> pixel operand, else weight operand
  auto issue_one = [&](int buf, int j) {
    const unsigned xs = lds0 + buf * STAGE;
    const unsigned ws = xs + BM * KB;
    if (j < PA) {
      const int hi = a_hi0[j] + kr, wi = a_wi0[j] + ks;
      const bool ok = ((unsigned)hi < (unsigned)q_H) && ((unsigned)wi < (unsigned)q_W);
      const T* src = ok ? a_base[j] + (hi * q_sH + wi * p.sW + kc) : zero;
      dma16(src, xs + (wave * PA + j) * 1024);
    } else {
      const int i = j - PA;
      dma16(b_ptr[i], ws + b_instr[i] * 1024);
      b_ptr[i] += b_step[i];
    }
  };
  auto advance_k = [&]() {
    kc += BKE;
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
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) issue_stage(s);
  int cur = 0, nxt = NST - 1;
  int kt = 0;
  for (; kt + NST - 1 < KT; ++kt) {                        // steady state: fetch stage kt+NST-1 while computing stage kt
    wait_vmcnt<LPS * (NST - 2)>();                         // stage kt has landed; later stages stay in flight
    __builtin_amdgcn_s_barrier();                          // every wave's part of stage kt is visible; buffer `nxt` is free
    compute_stage(cur, nxt, std::true_type());
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
  for (; kt < KT; ++kt) {                                  // drain: nothing left to fetch
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute_stage(cur, nxt, std::false_type());
    cur = cur + 1 == NST ? 0 : cur + 1;
  }

  // ---- epilogue: each wave stages its accumulator sub-tile through a private LDS region (fp32, two passes of TM/2
  // pixel tiles) and writes NHWC runs of TN*16 channels with 16-byte-per-lane accesses; all residual loads of a pass
  // are issued before any arithmetic so ~8 x 16 B per lane are in flight (the K=64 convs are HBM-bound here) ----
  __syncthreads();
  T* __restrict__ yg = reinterpret_cast<T*>(q_y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(q_res);
  constexpr int WC = TN * 16;                    // channels of a wave tile
  constexpr int CSW = WC * 4 + 16;               // staging row stride (bytes); +16 keeps ds_write_b128 conflict free
  constexpr int NPASS = TM >= 8 ? TM / 2 : (TM >= 2 ? 2 : 1);
  constexpr int TMP = TM / NPASS;                // 16-pixel tiles per pass
  constexpr int ROWS = TMP * 16;
  constexpr int CPR = WC / EPC;                  // 16-byte output chunks per row
  constexpr int ITER = ROWS * CPR / 64;
  static_assert((ROWS * CPR) % 64 == 0 && ITER >= 1, "epilogue chunking");
  char* stage = smem + wave * (ROWS * CSW);
  const bool vec_ok = (p.out_stride % EPC == 0) && (p.res_mode == OSD_RES_NONE || p.res_stride % EPC == 0);
  const int cbase = n0 + wn * WC;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int jj = 0; jj < TMP; ++jj) {
      const int j = ps * TMP + jj;
#pragma unroll
      for (int i = 0; i < TN; ++i)
        *reinterpret_cast<f32x4*>(stage + (jj * 16 + (lane & 15)) * CSW + (i * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
    }
    const int mbase = m0 + (wm * TM + ps * TMP) * 16;
    float v[ITER][EPC];
    bool live[ITER];
    int nval[ITER];
    size_t ooff[ITER];
    // pass 1: addresses + residual loads (all issued back to back)
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPR, cc = idx % CPR;
      const int m = mbase + row, c = cbase + cc * EPC;
      live[it] = (m < q_M) && (c < p.Cout);
      nval[it] = min(EPC, p.Cout - c);
      ooff[it] = (size_t)m * p.out_stride + c;
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[it][e] = 0.f;
      if (live[it] && p.res_mode != OSD_RES_NONE) {
        size_t res_off;
        if (p.res_mode == OSD_RES_SAME) {
          res_off = (size_t)m * p.res_stride + c;
        } else {
          const int n_img = m / q_HoWo;
          const int rem = m - n_img * q_HoWo;
          const int ho = rem / q_Wo, wo = rem - (rem / q_Wo) * q_Wo;
          res_off = ((size_t)(n_img * p.res_h + (ho >> 1)) * p.res_w + (wo >> 1)) * p.res_stride + c;
        }
        if constexpr (sizeof(T) == 2) {
          if (vec_ok && nval[it] == EPC) {
            const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(rg + res_off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = (float)r8[e];
          } else {
            for (int e = 0; e < nval[it]; e += 4) {
              const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(rg + res_off + e);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[it][e + k] = (float)r4[k];
            }
          }
        } else {
          const float4 r4 = *reinterpret_cast<const float4*>(rg + res_off);
          v[it][0] = r4.x; v[it][1] = r4.y; v[it][2] = r4.z; v[it][3] = r4.w;
        }
      }
    }
    // pass 2: accumulator + bias + residual, activation, store
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (!live[it]) continue;
      const int idx = it * 64 + lane;
      const int row = idx / CPR, cc = idx % CPR;
      const int c = cbase + cc * EPC;
      const char* src = stage + row * CSW + cc * EPC * 4;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(src + e * 4);
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + c + e);
        v[it][e] = (a4[0] + b4.x) + v[it][e];
        v[it][e + 1] = (a4[1] + b4.y) + v[it][e + 1];
        v[it][e + 2] = (a4[2] + b4.z) + v[it][e + 2];
        v[it][e + 3] = (a4[3] + b4.w) + v[it][e + 3];
      }
      if (q_mask) {     // ReLU backward of the producer layer: zero where its forward output was not positive
        const T* mk = reinterpret_cast<const T*>(q_mask) + ooff[it];
        if constexpr (sizeof(T) == 2) {
          if (vec_ok && nval[it] == EPC) {
            const bf16x8 m8 = *reinterpret_cast<const bf16x8*>(mk);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = (float)m8[e] > 0.f ? v[it][e] : 0.f;
          } else {
            for (int e = 0; e < nval[it]; ++e) v[it][e] = (float)mk[e] > 0.f ? v[it][e] : 0.f;
          }
        } else {
          const float4 m4 = *reinterpret_cast<const float4*>(mk);
          v[it][0] = m4.x > 0.f ? v[it][0] : 0.f; v[it][1] = m4.y > 0.f ? v[it][1] : 0.f;
          v[it][2] = m4.z > 0.f ? v[it][2] : 0.f; v[it][3] = m4.w > 0.f ? v[it][3] : 0.f;
        }
      }
      if (p.act == OSD_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
      } else if (p.act == OSD_ACT_EXP_SCALE) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[it][e] = expf(v[it][e] * (q_scale_dev ? *q_scale_dev : p.act_scale));
      }
      T* dst = yg + ooff[it];
      if constexpr (sizeof(T) == 2) {
        if (vec_ok && nval[it] == EPC) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[it][e];
          *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
          for (int e = 0; e < nval[it]; e += 4) {
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (__bf16)v[it][e + k];
            *reinterpret_cast<bf16x4*>(dst + e) = o;
          }
        }
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
      }
    }
  }
}

template <typename T, int BM, int BN, int KB, int WM, int WN, int NST, bool RELU_IN = false>
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
  p.KT = p.Ktot / BKE;
  constexpr int ring = NST * (BM + BN) * KB;
  constexpr int TMx = BM / WM / 16, TNx = BN / WN / 16;
  constexpr int npass = TMx >= 8 ? TMx / 2 : (TMx >= 2 ? 2 : 1);
  constexpr int stagec = WM * WN * ((TMx / npass) * 16) * (TNx * 16 * 4 + 16);
  constexpr int lds = ring > stagec ? ring : stagec;
  auto kern = conv_dma_kernel<T, BM, BN, KB, WM, WN, NST, RELU_IN>;
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
  if (p.relu_in) return launch_dma<T, 64, 64, KB, 2, 2, NST, true>(p, s);   // only the tiny P7 conv uses it
  switch (tile) {
    case 0: return launch_dma<T, 128, 128, KB, 2, 2, NST>(p, s);
    case 1: return launch_dma<T, 128, 64, KB, 4, 1, NST>(p, s);
    case 2: return launch_dma<T, 64, 64, KB, 2, 2, NST>(p, s);
    case 3: return launch_dma<T, 256, 16, KB, 4, 1, NST>(p, s);
    case 4:   // 256 x 256, 8 waves: half the operand bytes per MFMA of the 128 x 128 tile (the L1/TA path is the bound there)
      if constexpr (NST * 512 * KB <= 131072) return launch_dma<T, 256, 256, KB, 2, 4, NST>(p, s);
      else return osd_fail(OSD_ERR_UNSUPPORTED, "conv: the 256x256 tile does not fit LDS with this ring");
  }
  return osd_fail(OSD_ERR_INVALID_ARG, "conv: bad tile id %d", tile);
}

}  // namespace

// variant: 0 = deep ring (bf16 KB128 x3 / fp32 KB64 x4), 1 = shallow ring (x2: half the LDS, twice the blocks per CU),
//          2 = short stages (bf16 KB64 x4 / fp32 KB64 x3)
int osd_conv_dma_dispatch(int dtype, int tile, int variant, const ConvKParams& p, hipStream_t s) {
  if (dtype == OSD_F32) {
    if (variant == 1) return dispatch_tile_dma<float, 64, 2>(tile, p, s);
    if (variant == 2) return dispatch_tile_dma<float, 64, 3>(tile, p, s);
    return dispatch_tile_dma<float, 64, 4>(tile, p, s);
  }
  if (p.Cin % 64 != 0 || variant == 2) return dispatch_tile_dma<__bf16, 64, 4>(tile, p, s);
  if (variant == 1) return dispatch_tile_dma<__bf16, 128, 2>(tile, p, s);
  return dispatch_tile_dma<__bf16, 128, 3>(tile, p, s);
}
