// groupnorm_onepass — GroupNorm + ReLU of one tower layer over all FPN levels (fcos.py:29-37, 97-120) and its backward with ONE
// pass over HBM each (bf16, gfx950).  The two-launch forms of backward.hip read the conv output twice (statistics, then apply) and,
// backward, (u, dt) twice: 215 MB / 363 MB per tower layer at bs = 8 for 140 MB / 210 MB of tensors.  Here a workgroup loads its
// 128 pixels x c channels ONCE into registers, publishes its partial sums, waits for the other workgroups of its (level, image),
// sums everyone's partials in a fixed order (deterministic: no floating-point atomics on the statistics) and applies from the
// registers.
//
// Inter-workgroup hand-off (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility"): partial sums are stored `sc1`
// (relaxed agent-scope atomic stores), every storing wave waits `vmcnt(0)`, a workgroup barrier, then ONE lane adds 1 to the job's
// arrival counter (agent-scope atomic); consumers poll that counter with `sc1` loads from one lane, pass a workgroup barrier, and
// read the partials with `sc1` loads only.  Partial slots are whole 256-byte lines written by one wave instruction each and are
// never read before the job's barrier.
// Forward progress: a workgroup takes a TICKET (atomic counter) when it starts and its job follows from the ticket, so the
// workgroups that hold slots on the chip are always the ones with the lowest tickets; a job's workgroups have consecutive tickets
// and a job is at most ~100 workgroups (512 resident slots), so the lowest incomplete job can always become fully resident —
// whatever order the hardware starts blockIdx in, and beside other streams' kernels (they finish without us).  The launcher checks
// the premise (gn1p_resident_slots: the occupancy API's workgroups per CU x CUs must hold two whole jobs, else UNSUPPORTED and the
// caller runs the two-launch kernels), and the spin is bounded all the same: after ~1 s a workgroup sets the error word (sync[2])
// AND POISONS what it writes — every output element, the saved statistics (forward) / every du (backward) become NaN, so a step
// that ran on incomplete sums cannot pass for a valid one (round 6; until then it "went on with what it had").  Hosts read the
// error word where they synchronise anyway (ops.gn_onepass_check: bench.py after the timed region, TrainEngine.state_dict).
// Determinism: the group statistics and every output are bit-reproducible (fixed slot order, double sums); the backward's d gamma /
// d beta are folded per job in part order but ADDED across jobs with float atomics, so their last bits depend on timing.
// The sync words are zero before the first launch and every launch leaves them zero (the last workgroup to leave a job clears its
// counters, the last one of the launch clears the ticket counter): no memset per launch.  One sync buffer per concurrent stream.
#include "osd_common.h"
#include <stdlib.h>
#include <mutex>

#define OSD_STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

constexpr int kL = 8;            // levels per launch
constexpr int kThreads = 512;
constexpr int kSyncStride = 32;  // ints per job in `sync` (128 bytes: a line of its own); [0] arrivals, [16] departures
constexpr int kSpinLimit = 1 << 20;
constexpr int kMaxParts = 192;   // workgroups of one (level, image): see gn1p_fill

struct Gn1pParams {
  const void* x[kL];       // forward: u (conv output); backward: u
  const void* dy[kL];      // backward: dt
  void* y[kL];             // forward: t = relu(gn(u)); backward: du
  int hw[kL], parts[kL], ticket_begin[kL + 1], slot_begin[kL];
  int n_levels, n, c, groups, total;
  int spin_limit;          // polls before a waiting workgroup gives up (kSpinLimit; the self-test passes a small one)
  float eps;
  const float* gamma;
  const float* beta;
  float* ab;
  float* ws;
  int* sync;
  float* dgamma;
  float* dbeta;
};

__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
typedef unsigned gn_u32x4 __attribute__((ext_vector_type(4)));
// The tensors stay PACKED in registers (8 bf16 = 4 dwords) from the load to the store and are widened element by element where they
// are used (bf16 -> fp32 is a shift / a mask): as `bf16x8` values hipcc splits them into one register per element and keeps the
// fp32 conversions of pass 1 alive for pass 2 (256 VGPRs and spills instead of ~100)
__device__ __forceinline__ float bf_at(const gn_u32x4& v, int e) {
  const unsigned w = v[e >> 1];
  return __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
}
__device__ __forceinline__ gn_u32x4 bf_pack(const float (&f)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)f[e];
  return __builtin_bit_cast(gn_u32x4, o);
}
__device__ __forceinline__ void keep_packed(gn_u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// per-level fields are read from the kernel-argument segment by a wave-uniform index (scalar loads): indexing the by-value struct
// makes hipcc hold all eight entries of every array in SGPRs and spill them into VGPR lanes
typedef const __attribute__((address_space(4))) Gn1pParams* gn1p_kargs;
__device__ __forceinline__ gn1p_kargs gn1p_args() { return (gn1p_kargs)__builtin_amdgcn_kernarg_segment_ptr(); }

// what a workgroup works on: ticket -> (level, image, part); job = (level, image)
struct Gn1pJob { int lvl, img, part, parts, job, slot; };

__device__ __forceinline__ Gn1pJob gn1p_take(const Gn1pParams& P, int* sh) {
  if (threadIdx.x == 0) sh[0] = __hip_atomic_fetch_add(P.sync + 0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int t = __builtin_amdgcn_readfirstlane(sh[0]);
  const gn1p_kargs K = gn1p_args();
  Gn1pJob j;
  j.lvl = 0;
  for (int l = 1; l < P.n_levels; ++l)
    if (t >= K->ticket_begin[l]) j.lvl = l;
  const int r = t - K->ticket_begin[j.lvl];
  j.parts = K->parts[j.lvl];
  j.img = r / j.parts;
  j.part = r - j.img * j.parts;
  j.job = j.lvl * P.n + j.img;
  j.slot = K->slot_begin[j.lvl] + j.img * j.parts;      // first partial slot of the job
  return j;
}

// all partials of this workgroup are stored: signal, then wait for the job's other workgroups.  -> true when the wait timed out
// (the job's sums are incomplete: the caller poisons everything it writes)
__device__ __forceinline__ bool gn1p_arrive_and_wait(const Gn1pParams& P, const Gn1pJob& j, int* sh) {
  wait_vm0();
  __syncthreads();
  if (threadIdx.x == 0) {
    int* a = P.sync + kSyncStride * (1 + j.job);
    int spins = 0, bad = 0;
    // the add's own return value is the first poll: the last workgroup to arrive (the one everybody waits for) goes straight on
    if (__hip_atomic_fetch_add(a, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 < j.parts)
    while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < j.parts) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > P.spin_limit) { __hip_atomic_store(P.sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break; }
    }
    sh[2] = bad;
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(sh[2]) != 0;
}

// -> true for the LAST workgroup to leave the job (it has cleared the job's counters)
__device__ __forceinline__ bool gn1p_depart(const Gn1pParams& P, const Gn1pJob& j, int* sh) {
  if (threadIdx.x == 0) {
    int* a = P.sync + kSyncStride * (1 + j.job);
    const int old = __hip_atomic_fetch_add(a + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == j.parts - 1;
    if (last) {
      __hip_atomic_store(a, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a + 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sh[1] = last ? 1 : 0;
  }
  __syncthreads();
  return sh[1] != 0;
}

__device__ __forceinline__ void gn1p_exit(const Gn1pParams& P) {
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(P.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == P.total - 1) {         // every ticket has been taken, every job left: zero for the next launch
      __hip_atomic_store(P.sync + 0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// sum over the job's partial slots, value v of each (nv values per slot, nv <= 128), in slot order: threads (q, v) take slots
// q, q + Q, ... in double, then the Q partial sums are added in order -> tot[v] (LDS, double).  Eight loads are in flight per
// thread: one after the other (each an L2 / fabric round trip of 1 - 2 us) the 115 slots of a P3 job took ~30 us per workgroup
__device__ __forceinline__ void gn1p_sum_slots(const float* slots, int parts, int nv, double* scratch, double* tot) {
  const int Q = kThreads / nv;
  const int v = threadIdx.x % nv, q = threadIdx.x / nv;
  double s = 0.0;
  if (q < Q) {
    for (int k0 = q; k0 < parts; k0 += 8 * Q) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = ld_sc1(slots + (size_t)min(k0 + i * Q, parts - 1) * nv + v);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (k0 + i * Q < parts) s += (double)t[i];
    }
    scratch[q * nv + v] = s;
  }
  __syncthreads();
  if (threadIdx.x < nv) {
    double t = 0.0;
    for (int k = 0; k < Q; ++k) t += scratch[k * nv + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

// d gamma / d beta of a job: workgroup `part` folds values [v0, v1) of the 2 c per-workgroup partials (every partial of the job is
// visible behind the arrival barrier) — part order, so the job's sums do not depend on timing — and adds them to the launch's
// accumulators.  All loads of a workgroup in flight at once (~5 values x ~115 parts on 512 threads); scr: >= per * parts floats of LDS
__device__ __forceinline__ void gn1p_fold_slice(const Gn1pParams& P, const Gn1pJob& j, const float* pws, float* scr) {
  const int nvals = 2 * P.c;
  const int per = (nvals + j.parts - 1) / j.parts;
  const int v0 = j.part * per, cnt = min(nvals, v0 + per) - v0;
  if (cnt <= 0) return;                                 // uniform per workgroup
  for (int i = threadIdx.x; i < cnt * j.parts; i += kThreads) {
    const int vi = i / j.parts, k = i - vi * j.parts;
    scr[i] = ld_sc1(pws + (size_t)k * nvals + v0 + vi);
  }
  __syncthreads();
  // 8 threads per value: thread (vi, r) sums parts r, r + 8, ...; then the 8 sums in order (fixed order: deterministic)
  const int r = threadIdx.x & 7;
  for (int vb = 0; vb < cnt; vb += kThreads / 8) {      // uniform trip count
    const int vi = vb + (threadIdx.x >> 3);
    float t = 0.f;
    if (vi < cnt)
      for (int k = r; k < j.parts; k += 8) t += scr[vi * j.parts + k];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) t += __shfl_xor(t, o, 8);
    if (vi < cnt && r == 0) {
      const int v = v0 + vi;
      atomicAdd(v < P.c ? P.dgamma + v : P.dbeta + (v - P.c), t);
    }
  }
}

// ---- forward: t = relu(a u + b), a = gamma rstd, b = beta - mean a; ab planes (a, b, rstd, -mean rstd) for the backward pass ----
template <int U>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) gn1p_fwd_kernel(Gn1pParams P) {
  typedef __bf16 T;
  constexpr int E = 8;
  __shared__ int sh[4];
  __shared__ float red[2][kThreads];
  __shared__ double scratch[kThreads];
  __shared__ double tot[128];
  const Gn1pJob j = gn1p_take(P, sh);
  const int c = P.c, groups = P.groups, cch = c / E, PL = kThreads / cch;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int cpgc = (c / groups) / E;                 // 16-byte chunks per group (>= 1, checked by the launcher)
  const gn1p_kargs K = gn1p_args();
  const int hw = K->hw[j.lvl];
  // wave-uniform image base + 32-bit byte offset per lane (one register per access instead of a 64-bit pointer each)
  const char* x = reinterpret_cast<const char*>(K->x[j.lvl]) + (size_t)j.img * hw * c * sizeof(T);
  char* y = reinterpret_cast<char*>(K->y[j.lvl]) + (size_t)j.img * hw * c * sizeof(T);
  const int p0 = j.part * (U * PL) + pl;
  const unsigned rowb = (unsigned)c * (unsigned)sizeof(T), cofs = (unsigned)cc * 16u;
  gn_u32x4 v[U];
#pragma unroll
  for (int k = 0; k < U; ++k) v[k] = *reinterpret_cast<const gn_u32x4*>(x + ((unsigned)min(p0 + k * PL, hw - 1) * rowb + cofs));
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const float m = (p0 + k * PL < hw) ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { const float t = bf_at(v[k], e) * m; s += t; ss += t * t; }
  }
#pragma unroll
  for (int k = 0; k < U; ++k) keep_packed(v[k]);
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  const int nv = 2 * groups;
  float* slots = P.ws + (size_t)j.slot * nv;
  if (threadIdx.x < nv) {
    const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
    float t = 0.f;
    for (int l = 0; l < PL; ++l)
      for (int k = 0; k < cpgc; ++k) t += red[which][l * cch + g * cpgc + k];
    st_sc1(slots + (size_t)j.part * nv + threadIdx.x, t);
  }
  const bool timed_out = gn1p_arrive_and_wait(P, j, sh);
  gn1p_sum_slots(slots, j.parts, nv, scratch, tot);
  const int g = cc / cpgc;
  const double cnt = (double)hw * (c / groups);
  const double mean = tot[2 * g] / cnt;
  double var = tot[2 * g + 1] / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  // a timed-out wait: NaN in the saved planes and in every output element (`poison`, below)
  const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + (double)P.eps));
  float av[E], bv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int ch = cc * E + e;
    av[e] = P.gamma[ch] * frstd;
    bv[e] = P.beta[ch] - fmean * av[e];
  }
  const unsigned poison = (unsigned)__builtin_amdgcn_readfirstlane(timed_out ? -1 : 0);
  if (j.part == 0 && pl == 0) {
    const int n = P.n;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int ch = cc * E + e;
      P.ab[(((size_t)j.lvl * 4 + 0) * n + j.img) * c + ch] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, av[e]) | poison);
      P.ab[(((size_t)j.lvl * 4 + 1) * n + j.img) * c + ch] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, bv[e]) | poison);
      P.ab[(((size_t)j.lvl * 4 + 2) * n + j.img) * c + ch] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, frstd) | poison);
      P.ab[(((size_t)j.lvl * 4 + 3) * n + j.img) * c + ch] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, -fmean * frstd) | poison);
    }
  }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    keep_packed(v[k]);
    if (p0 + k * PL < hw) {
      float o[E];
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = fmaxf(fmaf(bf_at(v[k], e), av[e], bv[e]), 0.f);
      gn_u32x4 po = bf_pack(o);
      po |= poison;                                  // 0, or all ones (bf16 NaN pairs) after a timed-out wait: an SGPR operand, no register
      *reinterpret_cast<gn_u32x4*>(y + ((unsigned)(p0 + k * PL) * rowb + cofs)) = po;
    }
  }
  gn1p_depart(P, j, sh);
  gn1p_exit(P);
}

// ---- backward: dz = [a u + b > 0] dt, xhat = xa u + xb; du = rstd (dz gamma - mean(dz gamma) - xhat mean(dz gamma xhat));
// d gamma += sum dz xhat, d beta += sum dz (per-workgroup partials, folded by the last workgroup to leave the job) ----
template <int U>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) gn1p_bwd_kernel(Gn1pParams P) {
  typedef __bf16 T;
  constexpr int E = 8;
  __shared__ int sh[4];
  __shared__ float red[2][kThreads];
  __shared__ double scratch[kThreads];
  __shared__ double tot[128];
  extern __shared__ __attribute__((aligned(16))) float redc[];      // [2][PL][c]
  const Gn1pJob j = gn1p_take(P, sh);
  const int c = P.c, groups = P.groups, cch = c / E, PL = kThreads / cch, n = P.n;
  const int cc = threadIdx.x % cch, pl = threadIdx.x / cch;
  const int cpgc = (c / groups) / E;
  const gn1p_kargs K = gn1p_args();
  const int hw = K->hw[j.lvl];
  const size_t base = (size_t)j.img * hw * c * sizeof(T);
  const char* u = reinterpret_cast<const char*>(K->x[j.lvl]) + base;
  const char* dt = reinterpret_cast<const char*>(K->dy[j.lvl]) + base;
  char* du = reinterpret_cast<char*>(K->y[j.lvl]) + base;
  const int p0 = j.part * (U * PL) + pl;
  const unsigned rowb = (unsigned)c * (unsigned)sizeof(T), cofs = (unsigned)cc * 16u;
  gn_u32x4 uu[U], gg[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const unsigned off = (unsigned)min(p0 + k * PL, hw - 1) * rowb + cofs;
    uu[k] = *reinterpret_cast<const gn_u32x4*>(u + off);
    gg[k] = *reinterpret_cast<const gn_u32x4*>(dt + off);
  }
  float av[E], bv[E], gm[E];
  const float* abp = P.ab + ((size_t)j.lvl * 4 * n + j.img) * c + cc * E;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    av[e] = abp[e];
    bv[e] = abp[(size_t)n * c + e];
    gm[e] = P.gamma[cc * E + e];
  }
  const float xa = abp[(size_t)2 * n * c], xb = abp[(size_t)3 * n * c];      // rstd, -mean rstd: one group per chunk
  float s1 = 0.f, s2 = 0.f, dg[E], db[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { dg[e] = 0.f; db[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const bool ok = p0 + k * PL < hw;
    // one pixel at a time: pixel k's registers pass through an asm that also takes pixel k - 1's sums, so none of its widening can be
    // scheduled early (interleaved, hipcc holds the fp32 values of all U pixels: 16 more registers per pixel)
    asm volatile("" : "+v"(uu[k]), "+v"(gg[k]), "+v"(s1), "+v"(s2));
    asm volatile("" : "+v"(dg[0]), "+v"(dg[1]), "+v"(dg[2]), "+v"(dg[3]), "+v"(dg[4]), "+v"(dg[5]), "+v"(dg[6]), "+v"(dg[7]));
    asm volatile("" : "+v"(db[0]), "+v"(db[1]), "+v"(db[2]), "+v"(db[3]), "+v"(db[4]), "+v"(db[5]), "+v"(db[6]), "+v"(db[7]));
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float uv = bf_at(uu[k], e);
      const float z = fmaf(uv, av[e], bv[e]);
      const float dz = (ok && z > 0.f) ? bf_at(gg[k], e) : 0.f;
      const float xhat = fmaf(uv, xa, xb);
      s1 += dz * gm[e];
      s2 += dz * gm[e] * xhat;
      dg[e] += dz * xhat;
      db[e] += dz;
    }
  }
#pragma unroll
  for (int k = 0; k < U; ++k) { keep_packed(uu[k]); keep_packed(gg[k]); }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  {
    float* r0 = redc + ((size_t)0 * PL + pl) * c + cc * E;
    float* r1 = redc + ((size_t)1 * PL + pl) * c + cc * E;
    *reinterpret_cast<f32x4*>(r0) = f32x4{dg[0], dg[1], dg[2], dg[3]};
    *reinterpret_cast<f32x4*>(r0 + 4) = f32x4{dg[4], dg[5], dg[6], dg[7]};
    *reinterpret_cast<f32x4*>(r1) = f32x4{db[0], db[1], db[2], db[3]};
    *reinterpret_cast<f32x4*>(r1 + 4) = f32x4{db[4], db[5], db[6], db[7]};
  }
  __syncthreads();
  const int nv = 2 * groups;
  float* slots = P.ws + (size_t)j.slot * nv;
  float* pws = P.ws + (size_t)P.total * nv + (size_t)j.slot * 2 * c;      // d gamma / d beta partials of the job: [part][2][c]
  if (threadIdx.x < nv) {
    const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
    float t = 0.f;
    for (int l = 0; l < PL; ++l)
      for (int k = 0; k < cpgc; ++k) t += red[which][l * cch + g * cpgc + k];
    st_sc1(slots + (size_t)j.part * nv + threadIdx.x, t);
  }
  for (int i = threadIdx.x; i < 2 * c; i += kThreads) {
    const int which = i / c, ch = i - which * c;
    float t = 0.f;
    for (int l = 0; l < PL; ++l) t += redc[((size_t)which * PL + l) * c + ch];
    st_sc1(pws + (size_t)j.part * 2 * c + i, t);
  }
  const bool timed_out = gn1p_arrive_and_wait(P, j, sh);
  gn1p_sum_slots(slots, j.parts, nv, scratch, tot);
  const int g = cc / cpgc;
  const float inv_m = 1.f / ((float)hw * (c / groups));
  // a timed-out wait: c1 = NaN makes every du of this workgroup NaN
  const float c1 = timed_out ? __builtin_nanf("") : (float)tot[2 * g] * inv_m, c2 = (float)tot[2 * g + 1] * inv_m;
  unsigned chain = 0u;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    asm volatile("" : "+v"(uu[k]), "+v"(gg[k]) : "v"(chain));      // as in pass 1: behind the wait, and behind pixel k - 1
    if (p0 + k * PL < hw) {
      float o[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float uv = bf_at(uu[k], e);
        const float z = fmaf(uv, av[e], bv[e]);
        const float dz = z > 0.f ? bf_at(gg[k], e) : 0.f;
        const float xhat = fmaf(uv, xa, xb);
        o[e] = xa * (dz * gm[e] - c1 - xhat * c2);
      }
      const gn_u32x4 po = bf_pack(o);
      chain = po[3];
      *reinterpret_cast<gn_u32x4*>(du + ((unsigned)(p0 + k * PL) * rowb + cofs)) = po;
    }
  }
  __syncthreads();                                      // redc is free: every thread has published its partials long ago
  gn1p_fold_slice(P, j, pws, redc);
  gn1p_depart(P, j, sh);
  gn1p_exit(P);
}

// pixels per thread.  Backward: 8 (128 VGPRs, 3 spilled: two workgroups per CU); OSD_GN1P_U_BWD picks 4 / 6 / 7.  Forward (4 registers
// per pixel): 20 = 320 pixels per workgroup, the most that fits 128 VGPRs: the bs = 8 launch at 800 x 1024 is 440 workgroups, ONE
// round (all resident at once, two per CU) — a second round of a few workgroups costs a whole workgroup lifetime (load, hand-off,
// store: 36.9 us with 16 pixels per thread = 584 workgroups, 30.9 with 18 or 20).  The value must NOT follow the batch size: the
// partition of an image into partial sums fixes the summation order, and image i of a batch equals its single-image run bit for
// bit (tests/test_gpu_parity.py::test_full_size_batch8_properties).  OSD_GN1P_U_FWD = 8 / 16 / 18 for experiments (tools/gn_bench.py)
int gn1p_u_bwd() {
  static int ub = 0;
  if (ub == 0) {
    const char* e = getenv("OSD_GN1P_U_BWD");
    ub = e ? atoi(e) : 8;
    if (ub != 4 && ub != 6 && ub != 7 && ub != 8) ub = 8;
  }
  return ub;
}

int gn1p_u_fwd(int, const int32_t*, int, int) {
  static int uf = 0;
  if (uf == 0) {
    const char* e = getenv("OSD_GN1P_U_FWD");
    uf = e ? atoi(e) : 20;
    if (uf != 8 && uf != 16 && uf != 18 && uf != 20) uf = 20;
  }
  return uf;
}

int gn1p_fill(Gn1pParams& P, const char* who, int n_levels, const void* const* xs, const void* const* dys, void* const* ys,
              const int32_t* hws, int n, int c, int groups, int dtype, int U) {
  if (dtype != OSD_BF16) return osd_fail(OSD_ERR_UNSUPPORTED, "%s: bf16 only (fp32 runs the two-launch form)", who);
  if (n_levels < 1 || n_levels > kL || !xs || !ys || !hws || n < 1) return osd_fail(OSD_ERR_INVALID_ARG, "%s: bad arguments", who);
  if (c % 8 != 0 || c > 512 || kThreads % (c / 8) != 0 || groups < 1 || groups > 64 || c % groups != 0 || (c / groups) % 8 != 0)
    return osd_fail(OSD_ERR_UNSUPPORTED, "%s: unsupported shape c=%d groups=%d", who, c, groups);
  const int px = U * (kThreads / (c / 8));      // pixels per workgroup
  long long total = 0;
  for (int l = 0; l < kL; ++l) {
    const int j = l < n_levels ? l : 0;
    if (hws[j] < 1 || (long long)hws[j] * c * 2 >= 0x7fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "%s: level %d is empty or an image of it exceeds 2 GiB", who, j);
    P.x[l] = xs[j]; P.dy[l] = dys ? dys[j] : nullptr; P.y[l] = ys[j]; P.hw[l] = hws[j];
    P.parts[l] = (hws[j] + px - 1) / px;
    // forward progress: a job's workgroups must be able to be resident together — beside the job of another one-pass launch on
    // another stream (the two towers) — on 2 x 256 slots; bigger maps take the two-launch kernels (the caller falls back)
    if (P.parts[l] > kMaxParts)
      return osd_fail(OSD_ERR_UNSUPPORTED, "%s: level %d needs %d workgroups per image (> %d): use the two-launch form", who, j, P.parts[l], kMaxParts);
    if (2 * c + P.parts[l] > 2 * (kThreads / (c / 8)) * c)
      return osd_fail(OSD_ERR_UNSUPPORTED, "%s: level %d has more pixels per image than the fold's LDS scratch covers", who, j);
    P.ticket_begin[l] = (int)total;
    P.slot_begin[l] = (int)total;
    if (l < n_levels) total += (long long)n * P.parts[l];
    if (total > 0x3fffffffLL) return osd_fail(OSD_ERR_INVALID_ARG, "%s: too many workgroups", who);
  }
  P.ticket_begin[kL] = (int)total;
  P.n_levels = n_levels; P.n = n; P.c = c; P.groups = groups; P.total = (int)total;
  P.spin_limit = kSpinLimit;
  return OSD_OK;
}

// Forward progress rests on residency: the workgroups of the lowest incomplete job of THIS launch and of one more one-pass launch
// on another stream (the two towers) must fit the chip together.  The occupancy API's answer for the kernel about to be launched
// (registers, LDS, 512 threads) x the device's CU count is checked against 2 x the largest job: a part with fewer CUs, a
// partitioned device or a build whose register use halves the residency gets OSD_ERR_UNSUPPORTED here and the caller's two-launch
// kernels — not a 1 s stall.  (What the API cannot see — a CU mask set from outside — is what the poisoned timeout is for.)
int gn1p_check_residency(const Gn1pParams& P, const char* who, const void* kernel, size_t dyn_lds) {
  struct Entry { const void* k; int dev; size_t lds; int slots; };
  static std::mutex mu;
  static Entry cache[16];
  static int n_cache = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return osd_fail(OSD_ERR_LAUNCH, "%s: hipGetDevice failed", who);
  int slots = -1;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n_cache; ++i)
      if (cache[i].k == kernel && cache[i].dev == dev && cache[i].lds == dyn_lds) slots = cache[i].slots;
    if (slots < 0) {
      int cus = 0, per_cu = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, dyn_lds) != hipSuccess)
        return osd_fail(OSD_ERR_LAUNCH, "%s: occupancy query failed", who);
      slots = cus * per_cu;
      if (n_cache < 16) cache[n_cache++] = Entry{kernel, dev, dyn_lds, slots};
    }
  }
  int maxparts = 0;
  for (int l = 0; l < P.n_levels; ++l) maxparts = P.parts[l] > maxparts ? P.parts[l] : maxparts;
  if (2 * maxparts > slots)
    return osd_fail(OSD_ERR_UNSUPPORTED, "%s: a job of %d workgroups x 2 launches does not fit the %d workgroups this device holds at once: "
                    "use the two-launch form", who, maxparts, slots);
  return OSD_OK;
}

}  // namespace

extern "C" int64_t osd_groupnorm_onepass_workspace_bytes(int n_levels, const int32_t* hws, int n, int c, int groups, int backward) {
  if (n_levels < 1 || n_levels > kL || !hws || n < 1 || c < 8 || c % 8 != 0 || c > 512 || kThreads % (c / 8) != 0 || groups < 1) return -1;
  const int px = (backward ? gn1p_u_bwd() : gn1p_u_fwd(n_levels, hws, n, c)) * (kThreads / (c / 8));
  long long total = 0;
  for (int l = 0; l < n_levels; ++l) total += (long long)n * ((hws[l] + px - 1) / px);
  return total * (2LL * groups + (backward ? 2LL * c : 0LL)) * 4;
}

extern "C" int64_t osd_groupnorm_onepass_sync_bytes(int n_levels, int n) {
  if (n_levels < 1 || n_levels > kL || n < 1) return -1;
  return (int64_t)kSyncStride * (1 + (int64_t)n_levels * n) * 4;
}

// launch the forward kernel for U pixels per thread; `drop` workgroups (0, or 1 in the self-test) are not started
static int gn1p_launch_fwd(const Gn1pParams& P, int U, int drop, void* stream) {
  const void* kern = U == 8 ? reinterpret_cast<const void*>(gn1p_fwd_kernel<8>) : U == 18 ? reinterpret_cast<const void*>(gn1p_fwd_kernel<18>)
                   : U == 20 ? reinterpret_cast<const void*>(gn1p_fwd_kernel<20>) : reinterpret_cast<const void*>(gn1p_fwd_kernel<16>);
  const int rc = gn1p_check_residency(P, "groupnorm_fwd_onepass", kern, 0);
  if (rc) return rc;
  const dim3 grid((unsigned)(P.total - drop)), block(kThreads);
  switch (U) {
    case 8: hipLaunchKernelGGL(gn1p_fwd_kernel<8>, grid, block, 0, OSD_STREAM(stream), P); break;
    case 18: hipLaunchKernelGGL(gn1p_fwd_kernel<18>, grid, block, 0, OSD_STREAM(stream), P); break;
    case 20: hipLaunchKernelGGL(gn1p_fwd_kernel<20>, grid, block, 0, OSD_STREAM(stream), P); break;
    default: hipLaunchKernelGGL(gn1p_fwd_kernel<16>, grid, block, 0, OSD_STREAM(stream), P); break;
  }
  return osd_check_launch("gn1p_fwd");
}

extern "C" int osd_groupnorm_relu_fwd_levels_onepass(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                                     const float* gamma, const float* beta, float* ab, float* ws, int32_t* sync,
                                                     int n, int c, int groups, float eps, int dtype, void* stream) {
  if (!gamma || !beta || !ab || !ws || !sync) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_fwd_onepass: null argument");
  Gn1pParams P;
  if (n_levels < 1 || n_levels > kL || !hws || n < 1 || c < 8 || c % 8 != 0 || c > 512 || kThreads % (c / 8) != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_fwd_onepass: bad arguments");
  const int U = gn1p_u_fwd(n_levels, hws, n, c);
  int rc = gn1p_fill(P, "groupnorm_fwd_onepass", n_levels, xs, nullptr, ys, hws, n, c, groups, dtype, U);
  if (rc) return rc;
  P.eps = eps; P.gamma = gamma; P.beta = beta; P.ab = ab; P.ws = ws; P.sync = sync; P.dgamma = nullptr; P.dbeta = nullptr;
  return gn1p_launch_fwd(P, U, 0, stream);
}

// DIAGNOSTIC (tests/test_gpu_groupnorm_onepass.py): the forward launch with its LAST workgroup never started and a short spin —
// the job that workgroup belongs to can never complete, so its other workgroups must time out, set the error word of `sync` and
// write NaN.  `sync` is left dirty (counters of the incomplete job): pass a buffer of its own and throw it away afterwards.
extern "C" int osd_groupnorm_onepass_selftest_timeout(int n_levels, const void* const* xs, void* const* ys, const int32_t* hws,
                                                      const float* gamma, const float* beta, float* ab, float* ws, int32_t* sync,
                                                      int n, int c, int groups, float eps, int dtype, int spin_limit, void* stream) {
  if (!gamma || !beta || !ab || !ws || !sync) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_onepass_selftest: null argument");
  Gn1pParams P;
  if (n_levels < 1 || n_levels > kL || !hws || n < 1 || c < 8 || c % 8 != 0 || c > 512 || kThreads % (c / 8) != 0)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_onepass_selftest: bad arguments");
  const int U = gn1p_u_fwd(n_levels, hws, n, c);
  int rc = gn1p_fill(P, "groupnorm_onepass_selftest", n_levels, xs, nullptr, ys, hws, n, c, groups, dtype, U);
  if (rc) return rc;
  if (P.total < 2 || P.parts[n_levels - 1] < 2)
    return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_onepass_selftest: the last level must be cut into >= 2 workgroups per image");
  P.eps = eps; P.gamma = gamma; P.beta = beta; P.ab = ab; P.ws = ws; P.sync = sync; P.dgamma = nullptr; P.dbeta = nullptr;
  P.spin_limit = spin_limit > 0 ? spin_limit : 1024;
  return gn1p_launch_fwd(P, U, 1, stream);
}

extern "C" int osd_groupnorm_relu_bwd_levels_onepass(int n_levels, const void* const* us, const void* const* dts, void* const* dus,
                                                     const int32_t* hws, const float* ab, const float* gamma, const float* beta,
                                                     float* ws, int32_t* sync, float* dgamma, float* dbeta, int n, int c, int groups,
                                                     int dtype, void* stream) {
  if (!gamma || !beta || !ab || !ws || !sync || !dgamma || !dbeta || !dts) return osd_fail(OSD_ERR_INVALID_ARG, "groupnorm_bwd_onepass: null argument");
  Gn1pParams P;
  const int U = gn1p_u_bwd();
  int rc = gn1p_fill(P, "groupnorm_bwd_onepass", n_levels, us, dts, dus, hws, n, c, groups, dtype, U);
  if (rc) return rc;
  P.eps = 0.f; P.gamma = gamma; P.beta = beta; P.ab = const_cast<float*>(ab); P.ws = ws; P.sync = sync; P.dgamma = dgamma; P.dbeta = dbeta;
  const size_t lds = (size_t)2 * (kThreads / (c / 8)) * c * sizeof(float);
  {
    const void* kern = U == 4 ? reinterpret_cast<const void*>(gn1p_bwd_kernel<4>) : U == 6 ? reinterpret_cast<const void*>(gn1p_bwd_kernel<6>)
                     : U == 8 ? reinterpret_cast<const void*>(gn1p_bwd_kernel<8>) : reinterpret_cast<const void*>(gn1p_bwd_kernel<7>);
    rc = gn1p_check_residency(P, "groupnorm_bwd_onepass", kern, lds);
    if (rc) return rc;
  }
  const dim3 grid((unsigned)P.total), block(kThreads);
  switch (U) {
    case 4: hipLaunchKernelGGL(gn1p_bwd_kernel<4>, grid, block, lds, OSD_STREAM(stream), P); break;
    case 6: hipLaunchKernelGGL(gn1p_bwd_kernel<6>, grid, block, lds, OSD_STREAM(stream), P); break;
    case 8: hipLaunchKernelGGL(gn1p_bwd_kernel<8>, grid, block, lds, OSD_STREAM(stream), P); break;
    default: hipLaunchKernelGGL(gn1p_bwd_kernel<7>, grid, block, lds, OSD_STREAM(stream), P); break;
  }
  return osd_check_launch("gn1p_bwd");
}
