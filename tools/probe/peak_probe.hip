// Standalone gfx950 yardsticks for DESIGN.md / SURVEY.md 8(d): what the MFMA pipe and HBM deliver on THIS box.
//   mfma   back-to-back v_mfma on register operands (random data), 4 independent accumulators per wave,
//          1 / 2 waves per SIMD: the ceiling of any MFMA-bound kernel at the clock the chip holds under that load
//   copy   16 B per lane streaming copy of 1 GiB: the ceiling of any HBM-bound kernel (read + write)
// build:  hipcc --offload-arch=gfx950 -O3 -o peak_probe peak_probe.hip      run: ./peak_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void mfma_bf16_16x16x32(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 ra = {seed[t & 1023], seed[(t + 1) & 1023], seed[(t + 2) & 1023], seed[(t + 3) & 1023]};
  uint4 rb = {seed[(t + 5) & 1023], seed[(t + 6) & 1023], seed[(t + 7) & 1023], seed[(t + 8) & 1023]};
  const bf16x8 a = *reinterpret_cast<bf16x8*>(&ra), b = *reinterpret_cast<bf16x8*>(&rb);
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  out[t] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void mfma_bf16_32x32x16(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 ra = {seed[t & 1023], seed[(t + 1) & 1023], seed[(t + 2) & 1023], seed[(t + 3) & 1023]};
  uint4 rb = {seed[(t + 5) & 1023], seed[(t + 6) & 1023], seed[(t + 7) & 1023], seed[(t + 8) & 1023]};
  const bf16x8 a = *reinterpret_cast<bf16x8*>(&ra), b = *reinterpret_cast<bf16x8*>(&rb);
  f32x16 c0 = {0}, c1 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
  }
  out[t] = c0[0] + c1[1];
}

__global__ void mfma_f32_16x16x4(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const float a = __uint_as_float((seed[t & 1023] & 0x007fffffu) | 0x3f000000u), b = __uint_as_float((seed[(t + 9) & 1023] & 0x007fffffu) | 0x3f000000u);
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
  }
  out[t] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
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
  printf("device: %s, %d CUs, clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  std::vector<unsigned> h(1024);
  unsigned x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x7f7f7f7fu) | 0x3c003c00u; }   // bf16 pairs of moderate size
  unsigned* seed; float* out;
  CHECK(hipMalloc(&seed, 4096)); CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 512));
  CHECK(hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice));
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 256 * wps, blocks = cus;                    // one workgroup per CU, wps waves per SIMD
    double ms = time_ms([&] { hipLaunchKernelGGL(mfma_bf16_16x16x32, dim3(blocks), dim3(threads), 0, 0, seed, out, iters); }, 5);
    double flops = (double)blocks * (threads / 64) * iters * 4.0 * (2.0 * 16 * 16 * 32);
    printf("mfma bf16 16x16x32, %d wave(s)/SIMD: %.1f TFLOP/s\n", wps, flops / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(mfma_bf16_32x32x16, dim3(blocks), dim3(threads), 0, 0, seed, out, iters); }, 5);
    flops = (double)blocks * (threads / 64) * iters * 2.0 * (2.0 * 32 * 32 * 16);
    printf("mfma bf16 32x32x16, %d wave(s)/SIMD: %.1f TFLOP/s\n", wps, flops / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32_16x16x4, dim3(blocks), dim3(threads), 0, 0, seed, out, iters); }, 5);
    flops = (double)blocks * (threads / 64) * iters * 4.0 * (2.0 * 16 * 16 * 4);
    printf("mfma f32  16x16x4,  %d wave(s)/SIMD: %.1f TFLOP/s\n", wps, flops / ms / 1e9);
  }
  const size_t bytes = (size_t)1 << 30;
  uint4 *a, *b;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes));
  for (int g : {cus * 4, cus * 8, cus * 16}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(g), dim3(256), 0, 0, a, b, bytes / 16); }, 10);
    printf("copy 1 GiB, %d workgroups: %.2f TB/s (read + write)\n", g, 2.0 * bytes / ms / 1e9);
  }
  return 0;
}
