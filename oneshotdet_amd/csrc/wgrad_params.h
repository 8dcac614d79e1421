// Shared by conv_wgrad.hip (launchers, the ring kernels) and conv_wgrad_sk.hip (the software-pipelined kernel): the kernel
// argument table of a weight-gradient launch and two device helpers.
#pragma once
#include "osd_common.h"
#include <cstddef>

constexpr int kMaxSeg = 24;

// One launch can reduce over several (x, dy) pairs that share the weights (the FPN levels of the FCOS towers): the
// pixel splits are distributed over the segments, every workgroup works inside one segment, and all of them add into
// the same dW, so the atomic traffic is paid once instead of once per level.
struct WgradSeg {
  const void* x;
  const void* dy;
  float* dw;
  const float* scale;   // optional per-Cout factor (folded FrozenBN scale: d/dw of conv(x, w*scale))
  float* db;            // optional bias gradient: db[co] += sum over pixels of dy (done by the tap-0 / ci-tile-0 blocks)
  int H, W, Ho, Wo, M, rows_per_split;
  int Cin, Cout, R, S, sh, sw, ph, pw, dy_stride, tilesCo, tilesCi, Ktot;
  int block_begin;      // first (logical) workgroup of this segment; its workgroups: tilesCo x R*S*tilesCi x splits
  int stage_begin;      // team mode (conv_wgrad_sk.hip): first 64-pixel stage of this segment on the launch's concatenated pixel axis
  int owner;            // round 6: 1 = every output tile of this segment belongs to ONE workgroup of the launch (one pixel split, no
                        // other segment names its dW): the tile is added by plain load + store instead of memory-side atomics
};

// Every segment is a complete problem (its own tensors, geometry and outputs; segments that share a dW simply repeat the
// pointer): the FPN levels of one conv, the repeated blocks of a stage, or all weight gradients of a stage at once.
struct WgradParams {
  WgradSeg seg[kMaxSeg];
  int n_seg;
  int n_blocks;       // logical workgroups of the launch (= partial tiles in ordered mode)
  int sk_units;       // team mode of conv_wgrad_sk.hip: output tiles x taps per pixel range (0: every workgroup is one split of one segment)
  int sk_teams;       // ... teams (each walks 1 / sk_teams of the concatenated pixel axis), sk_total stages in all
  int sk_total;
  float* partials;    // ordered mode (osd_conv_desc.ordered_ws): every workgroup STORES its partial tile into slot
                      // [logical id][TCO * TCI + TCO] instead of adding it atomically; wgrad_reduce_kernel sums the slots
                      // in a fixed order: bit-reproducible dW, plain stores instead of memory-side atomics
};

// fp32 add into GLOBAL memory.  The dW / db pointers come out of the kernarg table as integers, so plain atomicAdd sees a
// generic pointer and emits flat_atomic_add_f32 (aperture check, counted on both lgkmcnt and vmcnt)
typedef __attribute__((address_space(1))) float wg_gfloat;
__device__ __forceinline__ void wg_atomic_add(float* ptr, float v) {
  (void)__hip_atomic_fetch_add((wg_gfloat*)ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Owner mode (WgradSeg.owner): dW[rows i*16 + e][cols j*16] += acc * scale by plain loads and stores — this workgroup is the only
// writer of the tile in this launch.  Memory-side float atomics run at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md): a 256 x 256
// fp32 tile per workgroup on every CU at once is ~50 us of epilogue, plain traffic ~20; and the layer4 + FPN stage launch wrote
// 329 MB of partial tiles for 60 MB of dW (profiles/r5_pmc_traffic_by_kernel.txt).  One row group (16 co rows) at a time: 4 x TB
// loads in flight, then their stores; the sched_barrier keeps hipcc from hoisting the next group's loads (registers).
template <int TA, int TB>
__device__ __forceinline__ void wg_owner_add(float* base, int Ktot, const f32x4 (&acc)[TA][TB], const float (&scv)[TA][4]) {
#pragma unroll
  for (int i = 0; i < TA; ++i) {
    float old[4][TB];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < TB; ++j) old[e][j] = ((const wg_gfloat*)(base + (size_t)(i * 16 + e) * Ktot))[j * 16];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < TB; ++j) ((wg_gfloat*)(base + (size_t)(i * 16 + e) * Ktot))[j * 16] = old[e][j] + acc[i][j][e] * scv[i][e];
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int N> __device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


template <typename T> __device__ __forceinline__ int wg_swz(int row) {
  if constexpr (sizeof(T) == 2) return 2 * (row & 7);
  else return 4 * (row & 3);
}

// conv_wgrad_sk.hip: the splits of `p` as filled by wgrad_launch (variant 13) or team mode (p.sk_* set: variant 3)
int osd_wgrad_sk_launch(const WgradParams& p, hipStream_t s);
