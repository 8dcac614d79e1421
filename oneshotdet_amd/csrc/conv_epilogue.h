// Pieces shared by the LDS-DMA convolution kernels (conv_igemm_dma.hip, conv_igemm_p8.hip): the per-segment view of a
// grouped launch and the LDS-staged NHWC epilogue.
#pragma once
#include "osd_common.h"
#include "conv_params.h"
#include <cstddef>
#include <type_traits>

namespace {

#ifdef OSD_EPI_NO_PIPE      // diagnostic build (OSD_BUILD_TAG=nopipe OSD_BUILD_FLAGS=-DOSD_EPI_NO_PIPE): same-box A/B of the pipelined epilogue
constexpr bool kEpiPipe = false;
#else
constexpr bool kEpiPipe = true;
#endif

// everything that differs between the (x, y) pairs of a grouped launch; uniform per workgroup (SGPRs)
struct ConvView {
  const void* x; void* y; const void* res; const void* mask; const float* scale_dev; const void* w; const float* bias;
  int H, W, Ho, Wo, M, sN, sH, HoWo, res_h, res_w;
  ConvGnb gn;
};

// tile_m: global pixel-tile index of this workgroup; on return it is the index inside the selected segment
__device__ __forceinline__ ConvView conv_select_view(const ConvKParams& p, int& tile_m) {
  ConvView q;
  q.x = p.x; q.y = p.y; q.res = p.res; q.mask = p.mask; q.scale_dev = p.act_scale_dev; q.w = p.w; q.bias = p.bias;
  q.H = p.H; q.W = p.W; q.Ho = p.Ho; q.Wo = p.Wo; q.M = p.M; q.sN = p.sN; q.sH = p.sH;
  q.res_h = p.res_h; q.res_w = p.res_w;
  q.gn.u = nullptr; q.gn.ab = nullptr; q.gn.gamma = nullptr; q.gn.ws = nullptr; q.gn.pw = nullptr;
  if (p.n_seg > 0) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < kConvMaxSeg; ++i)
      if (i < p.n_seg && tile_m >= p.seg[i].tile_begin) si = i;
    // read the chosen entry straight from the kernarg segment (constant address space, scalar loads): indexing the
    // by-value struct dynamically would make the compiler copy all of it to scratch
    typedef const __attribute__((address_space(4))) char* kptr;
    typedef unsigned long long u64;
    kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvKParams, seg) + si * (int)sizeof(ConvSeg);
#define OSD_KSEG(type, field) (*reinterpret_cast<const __attribute__((address_space(4))) type*>(base + offsetof(ConvSeg, field)))
    q.x = (const void*)OSD_KSEG(u64, x); q.y = (void*)OSD_KSEG(u64, y); q.res = (const void*)OSD_KSEG(u64, res);
    q.mask = (const void*)OSD_KSEG(u64, mask); q.scale_dev = (const float*)OSD_KSEG(u64, act_scale_dev);
    q.w = (const void*)OSD_KSEG(u64, w); q.bias = (const float*)OSD_KSEG(u64, bias);
    q.H = OSD_KSEG(int, H); q.W = OSD_KSEG(int, W); q.Ho = OSD_KSEG(int, Ho); q.Wo = OSD_KSEG(int, Wo);
    q.M = OSD_KSEG(int, M); q.sN = OSD_KSEG(int, sN); q.sH = OSD_KSEG(int, sH);
    if (p.gn_groups > 0) {
      q.gn.u = (const void*)OSD_KSEG(u64, gn.u); q.gn.ab = (const float*)OSD_KSEG(u64, gn.ab);
      q.gn.gamma = (const float*)OSD_KSEG(u64, gn.gamma); q.gn.ws = (float*)OSD_KSEG(u64, gn.ws); q.gn.pw = (float*)OSD_KSEG(u64, gn.pw);
    }
    tile_m -= OSD_KSEG(int, tile_begin);
#undef OSD_KSEG
    // nearest-2x top-down add: the addend is exactly half size; every-other-pixel identity: exactly double size (checked by the host)
    if (p.res_mode == OSD_RES_DOWN2X) { q.res_h = q.Ho << 1; q.res_w = q.Wo << 1; }
    else { q.res_h = q.Ho >> 1; q.res_w = q.Wo >> 1; }
  }
  q.HoWo = q.Ho * q.Wo;
  return q;
}

// Epilogue: each wave stages its fp32 accumulator sub-tile (TN x TM MFMA tiles: acc[i][j] = channels (wn*TN+i)*16..,
// pixels (wm*TM+j)*16..) through a private LDS region and writes NHWC runs of TN*16 channels with 16-byte-per-lane
// accesses; all residual loads of a pass are issued before any arithmetic.  The caller has already made the LDS ring
// reusable (barrier, no DMA in flight).
template <typename T, int TM, int TN, bool TWO_REGIONS = false, int GNB = 0, bool PIPE = false>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[TN][TM], const ConvKParams& p, const ConvView& q, char* smem,
                                              int wave, int wm, int wn, int lane, int m0, int n0,
                                              const float* pre_bias = nullptr) {
  // PIPE (round 4; wave tiles of <= 64 x 64, where the registers are there): the residual / mask operands of pass ps + 1 are
  // loaded while pass ps is computed and stored (two register sets), and pass 0's are issued before anything else — the
  // HBM-bound 1x1 convs of the bottlenecks (K = 128 .. 512: a K loop of 2 - 8 stages, then an epilogue that reads as many
  // bytes as it writes) no longer expose one memory latency per pass.  The pipelined form is compiled per (residual, mask)
  // combination, WITHOUT branches around its loads: hipcc places `s_waitcnt vmcnt(0)` at a branch merge that follows a load,
  // which would wait for the prefetch right where it is issued.
  // GNB (0 = off; 1 = backward or forward statistics, chosen per segment at run time; 2 = forward statistics ONLY — round 5: the
  // backward form's per-lane state (a, b, S, Su, two sets of u chunks: ~48 registers) is then not even allocated, and the
  // instantiation that gathers sum / sum of squares of the tower convs' outputs no longer spills in its epilogue):
  // the fast path also gathers the GroupNorm-backward statistics of ConvGnb (conv_params.h) from the values it stores —
  // AFTER their rounding to T, the numbers a separate pass over the stored tensor would read.  The launcher guarantees that the
  // segments that ask for it consist of whole tiles and that a wave's rows stay inside one image
  // TWO_REGIONS: the fast path's passes alternate between two private staging regions per wave (the caller's LDS must hold
  // 2 x waves x ROWS x CSW bytes), so the staging writes of pass p + 1 need not wait for the reads of pass p
  // pre_bias (optional): the fast path's EPC bias values of this lane (channels n0 + wn * TN * 16 + (lane % CPR) * EPC ..),
  // loaded by the caller ahead of time
  constexpr int EPC = 16 / (int)sizeof(T);
  const int q_M = q.M, q_HoWo = q.HoWo, q_Wo = q.Wo;
  const void* q_y = q.y; const void* q_res = q.res; const void* q_mask = q.mask; const float* q_scale_dev = q.scale_dev;
  // explicit global address space: the per-segment pointers come out of the kernarg table as integers, so hipcc would
  // otherwise treat them as generic and emit flat_load / flat_store (slower, and they tie up lgkmcnt as well)
#define OSD_G __attribute__((address_space(1)))
  OSD_G T* yg = (OSD_G T*)(const_cast<void*>(q_y));
  const OSD_G T* rg = (const OSD_G T*)(q_res);
  const OSD_G T* mkg = (const OSD_G T*)(q_mask);
  const OSD_G float* biasg = (const OSD_G float*)(q.bias);
  constexpr int WC = TN * 16;                    // channels of a wave tile
  constexpr int CSW = WC * 4 + 16;               // staging row stride (bytes); +16 keeps ds_write_b128 conflict free
  constexpr int NPASS = TM >= 8 ? TM / 2 : (TM >= 2 ? 2 : 1);
  constexpr int TMP = TM / NPASS;                // 16-pixel tiles per pass
  constexpr int ROWS = TMP * 16;
  constexpr int CPR = WC / EPC;                  // 16-byte output chunks per row
  constexpr int ITER = ROWS * CPR / 64;
  static_assert((ROWS * CPR) % 64 == 0 && ITER >= 1, "epilogue chunking");
  char* stage = smem + wave * (ROWS * CSW) * (TWO_REGIONS ? 2 : 1);
  const bool vec_ok = (p.out_stride % EPC == 0) && (p.res_mode == OSD_RES_NONE || p.res_stride % EPC == 0);
  const int cbase = n0 + wn * WC;

  // ---- fast path (wave-uniform): the wave's whole sub-tile is inside the output and every access is a full 16-byte
  // chunk.  No per-lane bounds branches, so each phase is one basic block: all residual and mask loads of a pass are
  // issued back to back (hipcc places `s_waitcnt vmcnt(0)` at every branch merge that follows a load, which made the
  // general path below wait for each load — and, in pass 2, for the previous STORE — one at a time), then the arithmetic,
  // then the stores with nothing to wait for between them.
  if (vec_ok && m0 + (wm * TM + TM) * 16 <= q_M && cbase + WC <= p.Cout) {
    typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type Vec;     // one 16-byte chunk
    const int cc = lane % CPR;                 // 64 % CPR == 0: a lane keeps its channel chunk in every iteration
    const int c = cbase + cc * EPC;
    float bv[EPC];
    if (pre_bias) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) bv[e] = pre_bias[e];
    } else {
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const f32x4 b4 = *(const OSD_G f32x4*)(biasg + c + e);
        bv[e] = b4[0]; bv[e + 1] = b4[1]; bv[e + 2] = b4[2]; bv[e + 3] = b4[3];
      }
    }
    const float escale = p.act == OSD_ACT_EXP_SCALE ? (q_scale_dev ? *(const OSD_G float*)q_scale_dev : p.act_scale) : 1.f;
    if constexpr (PIPE && kEpiPipe && GNB == 0) {
      auto piped = [&](auto res_tag, auto mask_tag) {
        constexpr bool HR = decltype(res_tag)::value, HM = decltype(mask_tag)::value;
        Vec rr[2][ITER], mm[2][ITER];
        auto prefetch = [&](int ps, Vec (&r)[ITER], Vec (&m)[ITER]) {
          const int mrow = m0 + (wm * TM + ps * TMP) * 16 + lane / CPR;
#pragma unroll
          for (int it = 0; it < ITER; ++it) {
            if constexpr (HR) r[it] = *(const OSD_G Vec*)(rg + (size_t)(mrow + it * (64 / CPR)) * p.res_stride + c);
            if constexpr (HM) m[it] = *(const OSD_G Vec*)(mkg + (size_t)(mrow + it * (64 / CPR)) * p.out_stride + c);
          }
        };
        prefetch(0, rr[0], mm[0]);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          char* stage_p = stage + (TWO_REGIONS ? (ps & 1) * (ROWS * CSW) : 0);
#pragma unroll
          for (int jj = 0; jj < TMP; ++jj) {
            const int j = ps * TMP + jj;
#pragma unroll
            for (int i = 0; i < TN; ++i)
              *reinterpret_cast<f32x4*>(stage_p + (jj * 16 + (lane & 15)) * CSW + (i * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
          }
          if (ps + 1 < NPASS) prefetch(ps + 1, rr[(ps + 1) & 1], mm[(ps + 1) & 1]);      // (ps is a compile-time constant: unrolled)
          const int mrow = m0 + (wm * TM + ps * TMP) * 16 + lane / CPR;
          float v[ITER][EPC];
#pragma unroll
          for (int it = 0; it < ITER; ++it) {
            const char* src = stage_p + (it * (64 / CPR) + lane / CPR) * CSW + cc * EPC * 4;
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
              const f32x4 a4 = *reinterpret_cast<const f32x4*>(src + e * 4);
              v[it][e] = a4[0] + bv[e]; v[it][e + 1] = a4[1] + bv[e + 1];
              v[it][e + 2] = a4[2] + bv[e + 2]; v[it][e + 3] = a4[3] + bv[e + 3];
            }
          }
          if constexpr (HR) {
#pragma unroll
            for (int it = 0; it < ITER; ++it)
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[it][e] += (float)rr[ps & 1][it][e];
          }
          if constexpr (HM) {
#pragma unroll
            for (int it = 0; it < ITER; ++it)
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[it][e] = (float)mm[ps & 1][it][e] > 0.f ? v[it][e] : 0.f;
          }
          if (p.act == OSD_ACT_RELU) {
#pragma unroll
            for (int it = 0; it < ITER; ++it)
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
          } else if (p.act == OSD_ACT_EXP_SCALE) {
#pragma unroll
            for (int it = 0; it < ITER; ++it)
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[it][e] = expf(v[it][e] * escale);
          }
#pragma unroll
          for (int it = 0; it < ITER; ++it) {
            Vec o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              if constexpr (sizeof(T) == 2) o[e] = (__bf16)v[it][e];
              else o[e] = v[it][e];
            }
            *(OSD_G Vec*)(yg + (size_t)(mrow + it * (64 / CPR)) * p.out_stride + c) = o;
          }
        }
      };
      const bool hr = p.res_mode == OSD_RES_SAME, hm = q_mask != nullptr;
      if ((p.res_mode == OSD_RES_NONE || p.res_mode == OSD_RES_SAME) && (hr || hm)) {
        if (hr && hm) piped(std::true_type(), std::true_type());
        else if (hr) piped(std::true_type(), std::false_type());
        else piped(std::false_type(), std::true_type());
        return;
      }
    }
    // GroupNorm-backward statistics: z = a u + b decides dz = z > 0 ? dt : 0; with xhat = xa u + xb per (image, channel) every
    // sum the backward needs follows from two per-channel sums, S = sum dz and Su = sum dz * u, so only a, b live in the loop
    [[maybe_unused]] bool gn_on = false;
    [[maybe_unused]] float ga[EPC], gb[EPC], gS[EPC], gSu[EPC];
    [[maybe_unused]] const OSD_G T* ug = (const OSD_G T*)(q.gn.u);
    [[maybe_unused]] int gn_img = 0;
    typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type VecU;
    [[maybe_unused]] VecU uu[ITER], uu_next[ITER];
    // GroupNorm FORWARD statistics (gn.u == nullptr, gn.ws != nullptr): sum and sum of squares of the stored values per (image,
    // slab, group) — what gnl_stats_kernel computes in its own pass over the conv's output
    [[maybe_unused]] bool gnf_on = false;
    [[maybe_unused]] float fs = 0.f, fss = 0.f;
    if constexpr (GNB != 0) {
      gn_on = (GNB == 1) && q.gn.u != nullptr;      // GNB == 2: constant false, the backward form compiles away
      gnf_on = !gn_on && q.gn.ws != nullptr;
      if (gnf_on) gn_img = (m0 + wm * TM * 16) / q_HoWo;
      if (gn_on) {
        gn_img = (m0 + wm * TM * 16) / q_HoWo;
        const OSD_G float* abg = (const OSD_G float*)(q.gn.ab) + (size_t)gn_img * p.Cout + c;
        const size_t plane = (size_t)p.gn_n * p.Cout;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const f32x4 a4 = *(const OSD_G f32x4*)(abg + e), b4 = *(const OSD_G f32x4*)(abg + plane + e);
#pragma unroll
          for (int k = 0; k < 4; ++k) { ga[e + k] = a4[k]; gb[e + k] = b4[k]; gS[e + k] = 0.f; gSu[e + k] = 0.f; }
        }
        // u of pass 0; the loads of pass ps + 1 are issued before the stores of pass ps, so that waiting for them (vmcnt counts
        // loads and stores in order) never waits for a store
        const int mrow0 = m0 + (wm * TM) * 16 + lane / CPR;
#pragma unroll
        for (int it = 0; it < ITER; ++it)
#ifdef OSD_GNB_NO_ULOAD      // diagnostic: every load reads the first row
          uu[it] = *(const OSD_G VecU*)(ug + c);
#else
          uu[it] = *(const OSD_G VecU*)(ug + (size_t)(mrow0 + it * (64 / CPR)) * p.out_stride + c);
#endif
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      char* stage_p = stage + (TWO_REGIONS ? (ps & 1) * (ROWS * CSW) : 0);
#pragma unroll
      for (int jj = 0; jj < TMP; ++jj) {
        const int j = ps * TMP + jj;
#pragma unroll
        for (int i = 0; i < TN; ++i)
          *reinterpret_cast<f32x4*>(stage_p + (jj * 16 + (lane & 15)) * CSW + (i * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
      }
      const int mrow = m0 + (wm * TM + ps * TMP) * 16 + lane / CPR;     // iteration `it` adds it * (64 / CPR) rows
      Vec rr[ITER], mm[ITER];
      if (p.res_mode == OSD_RES_SAME) {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          rr[it] = *(const OSD_G Vec*)(rg + (size_t)(mrow + it * (64 / CPR)) * p.res_stride + c);
      } else if (p.res_mode != OSD_RES_NONE) {      // UP2X: (ho / 2, wo / 2) of a half-size map; DOWN2X: (2 ho, 2 wo) of a double-size one
        const bool down = p.res_mode == OSD_RES_DOWN2X;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int m = min(mrow + it * (64 / CPR), q_M - 1);      // (rows past M are computed and dropped: keep their addend address inside the map)
          const int n_img = m / q_HoWo;
          const int rem = m - n_img * q_HoWo;
          const int ho = rem / q_Wo, wo = rem - (rem / q_Wo) * q_Wo;
          const int hr = down ? (ho << 1) : (ho >> 1), wr = down ? (wo << 1) : (wo >> 1);
          rr[it] = *(const OSD_G Vec*)(rg + ((size_t)(n_img * q.res_h + hr) * q.res_w + wr) * p.res_stride + c);
        }
      }
      if (q_mask) {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
          mm[it] = *(const OSD_G Vec*)(mkg + (size_t)(mrow + it * (64 / CPR)) * p.out_stride + c);
      }
      if constexpr (GNB != 0) {
        if (gn_on && ps + 1 < NPASS) {
#pragma unroll
          for (int it = 0; it < ITER; ++it)
#ifdef OSD_GNB_NO_ULOAD
            uu_next[it] = *(const OSD_G VecU*)(ug + c);
#else
            uu_next[it] = *(const OSD_G VecU*)(ug + (size_t)(mrow + ROWS + it * (64 / CPR)) * p.out_stride + c);
#endif
        }
      }
      float v[ITER][EPC];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const char* src = stage_p + (it * (64 / CPR) + lane / CPR) * CSW + cc * EPC * 4;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(src + e * 4);
          v[it][e] = a4[0] + bv[e]; v[it][e + 1] = a4[1] + bv[e + 1];
          v[it][e + 2] = a4[2] + bv[e + 2]; v[it][e + 3] = a4[3] + bv[e + 3];
        }
      }
      if (p.res_mode != OSD_RES_NONE) {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[it][e] += (float)rr[it][e];
      }
      if (q_mask) {     // ReLU backward of the producer layer: zero where its forward output was not positive
#pragma unroll
        for (int it = 0; it < ITER; ++it)
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[it][e] = (float)mm[it][e] > 0.f ? v[it][e] : 0.f;
      }
      if (p.act == OSD_ACT_RELU) {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
      } else if (p.act == OSD_ACT_EXP_SCALE) {
#pragma unroll
        for (int it = 0; it < ITER; ++it)
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[it][e] = expf(v[it][e] * escale);
      }
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        Vec o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          if constexpr (sizeof(T) == 2) o[e] = (__bf16)v[it][e];
          else o[e] = v[it][e];
        }
#ifdef OSD_EPI_NOSTORE      // diagnostic build: everything but the global stores
        asm volatile("" ::"v"(o));
#else
        *(OSD_G Vec*)(yg + (size_t)(mrow + it * (64 / CPR)) * p.out_stride + c) = o;
#endif
        if constexpr (GNB != 0) {
          if (gnf_on) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float x = (float)o[e]; fs += x; fss = fmaf(x, x, fss); }
          }
          if (gn_on) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const float uval = (float)uu[it][e];
              const float dz = fmaf(uval, ga[e], gb[e]) > 0.f ? (float)o[e] : 0.f;
              gS[e] += dz;
              gSu[e] = fmaf(dz, uval, gSu[e]);
            }
          }
        }
      }
      if constexpr (GNB != 0) {
        if (gn_on) {
#pragma unroll
          for (int it = 0; it < ITER; ++it) uu[it] = uu_next[it];
        }
      }
    }
    if constexpr (GNB != 0) {
      if (gnf_on) {
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1) { fs += __shfl_xor(fs, off); fss += __shfl_xor(fss, off); }
        if (lane < CPR) {
          const int slab = (blockIdx.x * 8 + wave) & (kGnSlabs - 1);
          const int cpg = p.Cout / p.gn_groups;
          OSD_G float* wsg = (OSD_G float*)(q.gn.ws) + (((size_t)gn_img * kGnSlabs + slab) * p.gn_groups + c / cpg) * 2;
          (void)__hip_atomic_fetch_add(wsg, fs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          (void)__hip_atomic_fetch_add(wsg + 1, fss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (gn_on) {
        // lanes cc, cc + CPR, ... hold the same channel chunk: fold them, then lanes 0 .. CPR - 1 turn the wave's S, Su into the
        // sums of the backward pass and add them to this (image, slab)
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) { gS[e] += __shfl_xor(gS[e], off); gSu[e] += __shfl_xor(gSu[e], off); }
        }
#ifdef OSD_GNB_NO_ATOMICS      // diagnostic
        if (lane < CPR && gS[0] == 12345.678f) {
#else
        if (lane < CPR) {
#endif
          const OSD_G float* abg = (const OSD_G float*)(q.gn.ab) + (size_t)gn_img * p.Cout + c;
          const size_t plane = (size_t)p.gn_n * p.Cout;
          const OSD_G float* gmg = (const OSD_G float*)(q.gn.gamma) + c;
          const int slab = (blockIdx.x * 8 + wave) & (kGnSlabs - 1);
          const int cpg = p.Cout / p.gn_groups;
          OSD_G float* wsg = (OSD_G float*)(q.gn.ws) + (((size_t)gn_img * kGnSlabs + slab) * p.gn_groups + c / cpg) * 2;
          OSD_G float* pwg = (OSD_G float*)(q.gn.pw) + (((size_t)gn_img * kGnSlabs + slab) * 2) * p.Cout + c;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float dg = fmaf(abg[2 * plane + e], gSu[e], abg[3 * plane + e] * gS[e]);     // sum dz * xhat
            s1 = fmaf(gmg[e], gS[e], s1);
            s2 = fmaf(gmg[e], dg, s2);
            (void)__hip_atomic_fetch_add(pwg + e, dg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(pwg + p.Cout + e, gS[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          (void)__hip_atomic_fetch_add(wsg, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          (void)__hip_atomic_fetch_add(wsg + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    return;
  }

  // ---- general path: ragged M / Cout tails, 8-byte granularity
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
          const bool down = p.res_mode == OSD_RES_DOWN2X;
          res_off = ((size_t)(n_img * q.res_h + (down ? (ho << 1) : (ho >> 1))) * q.res_w + (down ? (wo << 1) : (wo >> 1))) * p.res_stride + c;
        }
        if constexpr (sizeof(T) == 2) {
          if (vec_ok && nval[it] == EPC) {
            const bf16x8 r8 = *(const OSD_G bf16x8*)(rg + res_off);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = (float)r8[e];
          } else {
            for (int e = 0; e < nval[it]; e += 4) {
              const bf16x4 r4 = *(const OSD_G bf16x4*)(rg + res_off + e);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[it][e + k] = (float)r4[k];
            }
          }
        } else {
          const f32x4 r4 = *(const OSD_G f32x4*)(rg + res_off);
          v[it][0] = r4[0]; v[it][1] = r4[1]; v[it][2] = r4[2]; v[it][3] = r4[3];
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
        const f32x4 b4 = *(const OSD_G f32x4*)(biasg + c + e);
        v[it][e] = (a4[0] + b4[0]) + v[it][e];
        v[it][e + 1] = (a4[1] + b4[1]) + v[it][e + 1];
        v[it][e + 2] = (a4[2] + b4[2]) + v[it][e + 2];
        v[it][e + 3] = (a4[3] + b4[3]) + v[it][e + 3];
      }
      if (q_mask) {     // ReLU backward of the producer layer: zero where its forward output was not positive
        const OSD_G T* mk = mkg + ooff[it];
        if constexpr (sizeof(T) == 2) {
          if (vec_ok && nval[it] == EPC) {
            const bf16x8 m8 = *(const OSD_G bf16x8*)(mk);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = (float)m8[e] > 0.f ? v[it][e] : 0.f;
          } else {
            for (int e = 0; e < nval[it]; ++e) v[it][e] = (float)mk[e] > 0.f ? v[it][e] : 0.f;
          }
        } else {
          const f32x4 m4 = *(const OSD_G f32x4*)(mk);
          v[it][0] = m4[0] > 0.f ? v[it][0] : 0.f; v[it][1] = m4[1] > 0.f ? v[it][1] : 0.f;
          v[it][2] = m4[2] > 0.f ? v[it][2] : 0.f; v[it][3] = m4[3] > 0.f ? v[it][3] : 0.f;
        }
      }
      if (p.act == OSD_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[it][e] = fmaxf(v[it][e], 0.f);
      } else if (p.act == OSD_ACT_EXP_SCALE) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[it][e] = expf(v[it][e] * (q_scale_dev ? *(const OSD_G float*)q_scale_dev : p.act_scale));
      }
      OSD_G T* dst = yg + ooff[it];
      if constexpr (sizeof(T) == 2) {
        if (vec_ok && nval[it] == EPC) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[it][e];
          *(OSD_G bf16x8*)(dst) = o;
        } else {
          for (int e = 0; e < nval[it]; e += 4) {
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (__bf16)v[it][e + k];
            *(OSD_G bf16x4*)(dst + e) = o;
          }
        }
      } else {
        *(OSD_G f32x4*)(dst) = f32x4{v[it][0], v[it][1], v[it][2], v[it][3]};
      }
    }
  }
}
#undef OSD_G

}  // namespace
