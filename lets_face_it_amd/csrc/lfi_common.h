// Shared device/host helpers for liblfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfi.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- host side -------------------------------------------------------------
void lfi_set_error(const char* fmt, ...);
extern unsigned long long* g_lfi_stamps;  // diagnostics only (lfi_debug_set_stamps), null in normal operation

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

// ---- internal (not part of the C ABI): the sampler's fused per-frame conditioning (lfi_sample.hip), called by lfi_flow.hip
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_sample_cond_ok(int D, int G, int K1);
extern "C" __attribute__((visibility("hidden"))) long lfi_internal_sample_cond_bytes(int B, int Ks, int G, int K1);
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_sample_cond_prepare(const float* wct, long ldw, int col0, int K1, const float* wc,
                                                                                      int Ks, int G, void* frags, void* stream);
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_sample_cond(const float* faces, long ld_faces, long off, int K1, int B, int Ks, int G,
                                                                              const float* pre, const float* b_ih, void* frags, float* gic,
                                                                              float slope, long faces_floats, unsigned* reset, int reset_words,
                                                                              int have_xfrag, void* stream);
extern "C" __attribute__((visibility("hidden"))) void* lfi_internal_sample_cond_xfrag_ptr(void* frags, int Ks, int G, int K1);

// ---- internal: the flow's thin weight-gradient products in one pass over the backward stash (lfi_wgrad.hip), called by
// lfi_flow_param_grads (lfi_flow.hip)
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_flow_wgrad_ok(int B, int N, int C, int Ch, int Cout, int H, int G, int ldc,
                                                                                 int ldo);
extern "C" __attribute__((visibility("hidden"))) long lfi_internal_flow_wgrad_work_floats(int B, int N, int Ks, int role);
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_flow_wgrad(int role, int B, int N, int Ks, int C, int Ch, int Cout, int I,
                                                                              int ldc, int ldo, const void* dgh, const void* dgi,
                                                                              const float* h, const float* dlin, const float* sY,
                                                                              const float* sA, const float* dy, float* part, float* w_hh,
                                                                              float* w_ih, float* w_fl, float* b_fl, float* dW,
                                                                              int accumulate, void* stream);

// ---- device side -----------------------------------------------------------
// Operand planes (lfi_planes_from_f32, lfi_pgemm.hip): byte offset inside a 1-KB block (32 rows x 16 columns, bf16, row-major
// 32-byte rows) of the 16-byte chunk ch (columns 8 ch .. + 7) of row r. The two chunks of a row trade places in rows 8-15 and
// 24-31: the four 16-lane groups of a ds_read_b128 over (row l & 31, chunk l >> 5) then cover all 64 LDS banks, and the rows stay
// 32 contiguous bytes, which is what the transposing read (ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes) wants when the
// same block is used with its ROWS as the contraction index.
__device__ __forceinline__ int lfi_u_plane_offset(int r, int ch) { return r * 32 + ((ch ^ ((r >> 3) & 1)) << 4); }
// MFMA lane maps (guide §3, checked on the device by lfi_selftest_mfma):
//   16x16x4 f32 : A(i = l&15, k = l>>4)  B(k = l>>4, j = l&15)  D reg r -> (row (l>>4)*4 + r, col l&15)
//   32x32x2 f32 : A(i = l&31, k = l>>5)  B(k = l>>5, j = l&31)  D reg r -> (row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Gate non-linearities on the hardware transcendental units (v_exp_f32 = 2^x, v_rcp_f32; ~1 ulp each): the libm expf /
// tanhf calls cost 15-50 VALU instructions apiece and the gate epilogues run hundreds of them per lane. Absolute error
// ~1e-7 on values in (-1, 1) (tanh near 0 loses relative, not absolute, accuracy), saturating correctly at +-inf.
// -DLFI_EXACT_GATES=1 (measurement builds, tools/build_variant.sh): libm expf / tanhf and a true division;
// =2: the hardware exp2 with ONE Newton step on the reciprocal (rcp error ~1 ulp -> ~0.5 ulp).
#if defined(LFI_EXACT_GATES) && LFI_EXACT_GATES == 1
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#elif defined(LFI_EXACT_GATES) && LFI_EXACT_GATES == 2
__device__ __forceinline__ float rcp_newton_(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
}
__device__ __forceinline__ float sigmoidf_(float x) { return rcp_newton_(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * rcp_newton_(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
#else
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
#endif

// NACC 16 x 16 output tiles that share the A operand: acc[c] += A(16 x K) * B_c(K x 16), c < NACC.
//   A is in LDS, k-major: element (i, k) at a_lds[k * lda + i]                       (i = 0..15)
//   B_c is in global memory, row-major over k: element (k, j) at b[k * ldb + c * cstride + j]; lanes whose column is
//   outside the matrix pass jok = false and contribute zeros.
// The cells are latency-bound, not FLOP-bound (4-8 waves per CU, weights streamed from L2): operands are fetched
// 8 k-steps (32 k) at a time, two chunks in flight (the loads of chunk t+1 are issued before the MFMAs of chunk t),
// so a chain of K/4 dependent MFMAs pays ~K/32 L2 round trips instead of K/4. A single chain is split in two
// (even/odd k-steps) because v_mfma_f32_16x16x4_f32 has a 40-cycle dependent latency against a 32-cycle issue.
// K need not be a multiple of 4. All 64 lanes must call.
template <int NACC>
__device__ __forceinline__ void mma16_pf(f32x4 (&acc)[NACC], const float* a_lds, int lda, const float* __restrict__ b,
                                         int ldb, int cstride, int K, bool jok, int lane) {
  constexpr int U = NACC == 1 ? 8 : 4;  // k-steps per chunk (two chunks in flight)
  const int i = lane & 15, kq = lane >> 4;
  float a0[U], a1[U], b0[U][NACC], b1[U][NACC];
  f32x4 alt = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int k0, float (&a)[U], float (&bb)[U][NACC]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + 4 * u + kq;
      const bool ok = kk < K;
      a[u] = ok ? a_lds[kk * lda + i] : 0.0f;
#pragma unroll
      for (int c = 0; c < NACC; ++c) bb[u][c] = (ok && jok) ? b[kk * ldb + c * cstride + i] : 0.0f;  // 32-bit offsets from one base
    }
  };
  auto mma = [&](const float (&a)[U], const float (&bb)[U][NACC]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NACC == 1) {
        if (u & 1) alt = mfma16(a[u], bb[u][0], alt);
        else acc[0] = mfma16(a[u], bb[u][0], acc[0]);
      } else {
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[c] = mfma16(a[u], bb[u][c], acc[c]);
      }
    }
  };
  if (K <= 0) return;
  load(0, a0, b0);
  for (int k0 = 0; k0 < K; k0 += 8 * U) {
    const bool more1 = k0 + 4 * U < K;
    if (more1) load(k0 + 4 * U, a1, b1);
    mma(a0, b0);
    if (k0 + 8 * U < K) load(k0 + 8 * U, a0, b0);
    if (more1) mma(a1, b1);
  }
  if (NACC == 1) acc[0] += alt;
}

// One tile: acc += A(16 x K) * B(K x 16) with only columns j < nvalid of B read.
__device__ __forceinline__ f32x4 tile16_lds_glb(f32x4 acc, const float* a_lds, int lda, const float* __restrict__ b,
                                                int ldb, int K, int nvalid, int lane) {
  f32x4 a1[1] = {acc};
  mma16_pf<1>(a1, a_lds, lda, b, ldb, 0, K, (lane & 15) < nvalid, lane);
  return a1[0];
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
