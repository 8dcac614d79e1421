// ISA-level experiment for DESIGN.md 6d: why does the conv kernels' "MFMAs only" diagnostic build run at
// ~1.49 PFLOP/s when the bare loop of peak_probe.hip reaches ~2.2?  The same per-wave tile as conv_sp_kernel / conv_wgrad_sk
// (64 output channels x 128 pixels per wave = 4 weight fragments x 8 pixel fragments, 32 accumulator tiles = 128 VGPRs), built
// up one ingredient at a time:
//   bare      4 accumulators, one operand pair (peak_probe.hip's loop)
//   outer     the 4 x 8 outer product on 12 distinct fragment registers, pixel-fragment-outer order, nothing else in the loop
//   sched     + the __builtin_amdgcn_sched_barrier(0) per 4-MFMA group the kernels use to pin their software pipeline
//   lds       + the operand fragments read from LDS one half stage ahead (12 ds_read_b128 per 32 MFMAs, replaced in place)
//   bar       + one s_barrier per 64 MFMAs (the kernels' one barrier per K stage)
//   o32 / l32 the same tile on v_mfma_f32_32x32x16_bf16 (2 x 4 tiles of 32 x 32: half the MFMA instructions per FLOP), registers
//             only / with its 6 ds_read_b128 per 8 MFMAs
// each with 2 waves per SIMD (512 threads, one workgroup per CU) and 1 wave per SIMD (256 threads).
// build:  hipcc --offload-arch=gfx950 -O3 -o mfma_structure_probe mfma_structure_probe.hip      run: ./mfma_structure_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint4 seed4(const unsigned* seed, int t, int k) {
  return uint4{seed[(t + k) & 1023], seed[(t + k + 1) & 1023], seed[(t + k + 2) & 1023], seed[(t + k + 3) & 1023]};
}
#define BF(x) (*reinterpret_cast<const bf16x8*>(&(x)))

// in-kernel clock (MI355X_MICROARCH.md 'DVFS give-back' item 6): delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz).
// The stamps go to a buffer of their own that nothing else reads.
struct Stamp { unsigned long long c0, r0; };
__device__ __forceinline__ Stamp stamp_begin() { return Stamp{__builtin_amdgcn_s_memtime(), __builtin_amdgcn_s_memrealtime()}; }
__device__ __forceinline__ void stamp_end(const Stamp& s, unsigned long long* st) {
  if (threadIdx.x == 0) {
    st[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - s.c0;
    st[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - s.r0;
  }
}

__global__ void __launch_bounds__(512) k_bare(const unsigned* __restrict__ seed, float* __restrict__ out, int iters, unsigned long long* __restrict__ st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 ra = seed4(seed, t, 0), rb = seed4(seed, t, 5);
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const Stamp sp = stamp_begin();
  for (int i = 0; i < iters * 8; ++i) {      // 32 MFMAs per `iter`, like the others
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(ra), BF(rb), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(ra), BF(rb), c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(ra), BF(rb), c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(ra), BF(rb), c3, 0, 0, 0);
  }
  stamp_end(sp, st);
  out[t] = c0[0] + c1[1] + c2[2] + c3[3];
}

// MODE 0 = outer, 1 = + sched_barrier, 2 = + LDS fragment reads one half stage ahead, 3 = + s_barrier per 64 MFMAs
template <int MODE>
__global__ void __launch_bounds__(512) k_outer16(const unsigned* __restrict__ seed, float* __restrict__ out, int iters, unsigned long long* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (MODE >= 2) {
    for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) *reinterpret_cast<uint4*>(smem + i * 16) = seed4(seed, i, 3);
    __syncthreads();
  }
  uint4 wf0[4], wf1[4], xf[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { wf0[i] = seed4(seed, t, 7 * i); wf1[i] = seed4(seed, t, 7 * i + 3); }
#pragma unroll
  for (int j = 0; j < 8; ++j) xf[j] = seed4(seed, t, 11 * j + 1);
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // conflict-free ds_read_b128 pattern of the conv kernels: row = lane & 15 (128-byte rows), 16-byte chunk = (lane >> 4) ^ key
  const int frag = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);
  auto half = [&](uint4 (&wc)[4], uint4 (&wn)[4], int base, bool bar) {
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wn[i] = *reinterpret_cast<const uint4*>(smem + (base + i * 2048 + frag));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(wc[i]), BF(xf[j]), acc[i][j], 0, 0, 0);
      if (MODE >= 3 && bar && j == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (MODE >= 2) xf[j] = *reinterpret_cast<const uint4*>(smem + (base + 8192 + j * 2048 + frag));
      if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
    }
  };
  int base = 0;
  const Stamp sp = stamp_begin();
  for (int it = 0; it < iters; it += 2) {
    half(wf0, wf1, base, false);
    half(wf1, wf0, base + 64, true);
    base ^= 24576;
  }
  stamp_end(sp, st);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][(i + j) & 3];
  out[t] = s;
}

template <int MODE>      // 0 = registers only, 2 = + LDS reads (4 pixel + 2 weight fragments per k step of 16)
__global__ void __launch_bounds__(512) k_outer32(const unsigned* __restrict__ seed, float* __restrict__ out, int iters, unsigned long long* __restrict__ st) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (MODE >= 2) {
    for (int i = threadIdx.x; i < 65536 / 16; i += blockDim.x) *reinterpret_cast<uint4*>(smem + i * 16) = seed4(seed, i, 3);
    __syncthreads();
  }
  uint4 wf0[2], wf1[2], xf[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) { wf0[i] = seed4(seed, t, 7 * i); wf1[i] = seed4(seed, t, 7 * i + 3); }
#pragma unroll
  for (int j = 0; j < 4; ++j) xf[j] = seed4(seed, t, 11 * j + 1);
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int frag = (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4);
  auto kstep = [&](uint4 (&wc)[2], uint4 (&wn)[2], int base) {      // 8 MFMAs of 32 x 32 x 16 = the FLOPs of 16 of 16 x 16 x 32
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) wn[i] = *reinterpret_cast<const uint4*>(smem + (base + i * 4096 + frag));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(wc[i]), BF(xf[j]), acc[i][j], 0, 0, 0);
      if (MODE >= 2) {
        xf[j] = *reinterpret_cast<const uint4*>(smem + (base + 8192 + j * 4096 + frag));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  int base = 0;
  const Stamp sp = stamp_begin();
  for (int it = 0; it < iters; ++it) {       // 32 MFMA-equivalents of 16 x 16 x 32 per `iter` = 16 of 32 x 32 x 16
    kstep(wf0, wf1, base);
    kstep(wf1, wf0, base + 32);
    base ^= 24576;
  }
  stamp_end(sp, st);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][(i + j) & 15];
  out[t] = s;
}

template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<unsigned> h(1024);
  unsigned* seed; float* out; unsigned long long* st;
  CHECK(hipMalloc(&seed, 4096)); CHECK(hipMalloc(&out, (size_t)cus * 512 * 4)); CHECK(hipMalloc(&st, (size_t)cus * 16));
  const int iters = 40000;       // x 32 MFMAs (16 x 16 x 32) per wave: ~10-20 ms per launch
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer16<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer16<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  printf("{\"device\": \"%s\", \"cus\": %d, \"unit\": \"TFLOP/s | in-kernel GHz | shader cycles per 16x16x32-equivalent MFMA per SIMD\", \"rows\": {\n", prop.gcnArchName, cus);
  const char* names[] = {"bare", "outer", "outer+sched_barrier", "outer+sched+lds_reads", "outer+sched+lds_reads+barrier", "outer32x32", "outer32x32+lds_reads"};
  bool first = true;
  for (int zero = 0; zero < 2; ++zero) {
    for (int i = 0; i < 1024; ++i) h[i] = zero ? 0u : (0x3f803f80u ^ (unsigned)(rand() & 0x007f007f) ^ ((unsigned)(rand() & 1) << 15) ^ ((unsigned)(rand() & 1) << 31));
    CHECK(hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice));
    for (int v = 0; v < 7; ++v) {
      for (int threads : {512, 256}) {
        auto launch = [&]() {
          switch (v) {
            case 0: hipLaunchKernelGGL(k_bare, dim3(cus), dim3(threads), 0, 0, seed, out, iters, st); break;
            case 1: hipLaunchKernelGGL(k_outer16<0>, dim3(cus), dim3(threads), 0, 0, seed, out, iters, st); break;
            case 2: hipLaunchKernelGGL(k_outer16<1>, dim3(cus), dim3(threads), 0, 0, seed, out, iters, st); break;
            case 3: hipLaunchKernelGGL(k_outer16<2>, dim3(cus), dim3(threads), 65536, 0, seed, out, iters, st); break;
            case 4: hipLaunchKernelGGL(k_outer16<3>, dim3(cus), dim3(threads), 65536, 0, seed, out, iters, st); break;
            case 5: hipLaunchKernelGGL(k_outer32<0>, dim3(cus), dim3(threads), 0, 0, seed, out, iters, st); break;
            case 6: hipLaunchKernelGGL(k_outer32<2>, dim3(cus), dim3(threads), 65536, 0, seed, out, iters, st); break;
          }
        };
        const double ms = time_ms(launch, 12);
        std::vector<unsigned long long> hs((size_t)cus * 2);
        CHECK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (int b = 0; b < cus; ++b) { ghz.push_back((double)hs[b * 2] / (double)hs[b * 2 + 1] * 0.1); cyc.push_back((double)hs[b * 2]); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double fl = (double)cus * (threads / 64) * iters * 32.0 * 16384.0;
        const double per_mfma = cyc[cus / 2] / ((double)iters * 32.0 * (threads / 256));      // waves per SIMD x MFMAs per wave
        printf("%s  \"%s, %s operands, %d waves/SIMD\": [%.0f, %.3f, %.2f]", first ? "" : ",\n", names[v], zero ? "zero" : "random", threads / 256,
               fl / (ms * 1e-3) / 1e12, ghz[cus / 2], per_mfma);
        first = false;
      }
    }
  }
  printf("\n}}\n");
  return 0;
}
