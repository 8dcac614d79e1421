// Shared helpers for the gfx950 kernels of liboneshotdet_hip.so (see include/oneshotdet_hip.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/oneshotdet_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

extern thread_local char g_osd_err[512];

static inline int osd_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_osd_err, sizeof(g_osd_err), fmt, ap);
  va_end(ap);
  return code;
}

static inline int osd_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return osd_fail(OSD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return OSD_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
};
template <> struct ElemTraits<__bf16> {
  static constexpr int kPerChunk = 8;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
