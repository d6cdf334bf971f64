// Shared device/host helpers for liblfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfi.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- host side -------------------------------------------------------------
void lfi_set_error(const char* fmt, ...);

#define LFI_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      lfi_set_error(__VA_ARGS__);              \
      return LFI_ERR_ARG;                      \
    }                                          \
  } while (0)

#define LFI_LAUNCH_CHECK(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      lfi_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return LFI_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

static inline int lfi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device side -----------------------------------------------------------
// MFMA lane maps (guide §3, checked on the device by lfi_selftest_mfma):
//   16x16x4 f32 : A(i = l&15, k = l>>4)  B(k = l>>4, j = l&15)  D reg r -> (row (l>>4)*4 + r, col l&15)
//   32x32x2 f32 : A(i = l&31, k = l>>5)  B(k = l>>5, j = l&31)  D reg r -> (row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One 16 x 16 output tile: acc += A(16 x K) * B(K x 16).
//   A is in LDS, k-major: element (i, k) at a_lds[k * lda + i]           (i = 0..15)
//   B is in global memory, row-major over k: element (k, j) at b[k * ldb + j], only j < nvalid is read
// K need not be a multiple of 4 (out-of-range k contributes zero). All 64 lanes must call.
__device__ __forceinline__ f32x4 tile16_lds_glb(f32x4 acc, const float* a_lds, int lda, const float* __restrict__ b,
                                                long ldb, int K, int nvalid, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  const bool jok = i < nvalid;
  int k = 0;
  for (; k + 4 <= K; k += 4) {
    const int kk = k + kq;
    float av = a_lds[kk * lda + i];
    float bv = jok ? b[(long)kk * ldb + i] : 0.0f;
    acc = mfma16(av, bv, acc);
  }
  if (k < K) {
    const int kk = k + kq;
    const bool kok = kk < K;
    float av = kok ? a_lds[kk * lda + i] : 0.0f;
    float bv = (kok && jok) ? b[(long)kk * ldb + i] : 0.0f;
    acc = mfma16(av, bv, acc);
  }
  return acc;
}

// 32 x 32 variant: A(i, k) at a_lds[k * lda + i] (i = 0..31), B(k, j) at b[k * ldb + j], j < nvalid.
__device__ __forceinline__ f32x16 tile32_lds_glb(f32x16 acc, const float* a_lds, int lda, const float* __restrict__ b,
                                                 long ldb, int K, int nvalid, int lane) {
  const int i = lane & 31, kq = lane >> 5;
  const bool jok = i < nvalid;
  int k = 0;
  for (; k + 2 <= K; k += 2) {
    const int kk = k + kq;
    float av = a_lds[kk * lda + i];
    float bv = jok ? b[(long)kk * ldb + i] : 0.0f;
    acc = mfma32(av, bv, acc);
  }
  if (k < K) {
    const int kk = k + kq;
    const bool kok = kk < K;
    float av = kok ? a_lds[kk * lda + i] : 0.0f;
    float bv = (kok && jok) ? b[(long)kk * ldb + i] : 0.0f;
    acc = mfma32(av, bv, acc);
  }
  return acc;
}

// Sum over the 64 lanes of a wave (result in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
