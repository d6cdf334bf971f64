// Error reporting, version, MFMA lane-map self-test, column sums and the flat-buffer optimiser.
#include <stdarg.h>
#include <stdio.h>

#include "lfi_common.h"

static thread_local char g_err[512] = "";

void lfi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

unsigned long long* g_lfi_stamps = nullptr;   // diagnostics only, process-global by necessity (include/lfi.h says so): backward passes run on autograd's thread

extern "C" const char* lfi_last_error(void) { return g_err; }
extern "C" int lfi_version(void) { return 100; }

// ------------------------------------------------------------------ a stream confined to part of every XCD
// hipExtStreamCreateWithCUMask on this card (tools/probes/cu_mask_map_probe.hip): bits 8g .. 8g+7 of the mask are CU slot g of the
// eight XCDs (a group with any bit set enables all eight), so `cus_per_xcd` leading groups give a stream that owns that many CUs of
// every XCD - its kernels leave the other CUs (and their share of each L2) to whatever runs beside them.
extern "C" int lfi_stream_create_partial(int cus_per_xcd, void** stream) {
  LFI_REQUIRE(stream != nullptr, "lfi_stream_create_partial: null output pointer");
  int dev = 0, cus = 0;
  LFI_REQUIRE(hipGetDevice(&dev) == hipSuccess &&
                  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess,
              "lfi_stream_create_partial: no device");
  // The mask layout below (bit 8 c + x = CU c of XCD x) was probed on the 256-CU / 8-XCD MI355X only (tools/cu_mask_probe.py):
  // any other part gets an error, and the caller falls back to an ordinary second stream, instead of a partition nobody measured
  LFI_REQUIRE(cus == 256 && cus_per_xcd >= 1 && cus_per_xcd <= cus / 8,
              "lfi_stream_create_partial: %d CUs per XCD asked of a device with %d CUs (the CU-mask layout is known for 256 CUs in 8 "
              "XCDs only)", cus_per_xcd, cus);
  uint32_t mask[64] = {0};
  for (int b = 0; b < 8 * cus_per_xcd; ++b) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask);
  if (e != hipSuccess) {
    lfi_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return LFI_ERR_LAUNCH;
  }
  *stream = (void*)st;
  return LFI_OK;
}
extern "C" int lfi_stream_destroy(void* stream) {
  if (!stream) return LFI_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    lfi_set_error("hipStreamDestroy: %s", hipGetErrorString(e));
    return LFI_ERR_LAUNCH;
  }
  return LFI_OK;
}

// ------------------------------------------------------------------ MFMA lane-map self-test
// A(i,k) = 1 + i + 100 k ; B(k,j) = (k == kk) * (1 + 1000 j) summed over k gives D(i,j) = sum_k A(i,k) B(k,j):
// with small integers every product is exact, and an asymmetric B exposes a transposed accumulator map.
__global__ void selftest_mfma_kernel(int* out) {
  const int lane = threadIdx.x;
  int bad = 0;
  {  // 16x16x4
    const int i = lane & 15, k = lane >> 4;
    const float a = (float)(1 + i + 100 * k);   // A(i, k)
    const float b = (float)(1 + 7 * k + 31 * i);  // B(k, j = i)
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma16(a, b, c);
    for (int r = 0; r < 4; ++r) {
      const int row = (lane >> 4) * 4 + r, col = lane & 15;
      float ref = 0.f;
      for (int kk = 0; kk < 4; ++kk) ref += (float)(1 + row + 100 * kk) * (float)(1 + 7 * kk + 31 * col);
      if (c[r] != ref) ++bad;
    }
  }
  {  // 32x32x2
    const int i = lane & 31, k = lane >> 5;
    const float a = (float)(1 + i + 100 * k);
    const float b = (float)(1 + 7 * k + 31 * i);
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = mfma32(a, b, c);
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
      float ref = 0.f;
      for (int kk = 0; kk < 2; ++kk) ref += (float)(1 + row + 100 * kk) * (float)(1 + 7 * kk + 31 * col);
      if (c[r] != ref) ++bad;
    }
  }
  atomicAdd(out, bad);
}

extern "C" int lfi_selftest_mfma(int* out, void* stream) {
  LFI_REQUIRE(out, "lfi_selftest_mfma: null output");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(int), st);
  if (e != hipSuccess) {
    lfi_set_error("lfi_selftest_mfma: memset failed: %s", hipGetErrorString(e));
    return LFI_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, st, out);
  LFI_LAUNCH_CHECK("lfi_selftest_mfma");
  return LFI_OK;
}

// ------------------------------------------------------------------ column sums (bias gradients)
// One pass: block (cx, ry) sums rows [ry*rpb, ry*rpb + rpb) of 64 columns, four row lanes per column, into
// dst[batch][ry][col]. Two passes (rows -> <= 256 row blocks -> 1) with fixed order: bitwise reproducible.
namespace {
constexpr int CS_MAX_BLOCKS = 256;

int cs_rows_per_block(int rows) {
  const int r = (rows + CS_MAX_BLOCKS - 1) / CS_MAX_BLOCKS;
  return r < 64 ? 64 : r;
}

__global__ __launch_bounds__(256) void colsum_pass(const float* __restrict__ X, long ldx, long strideX, int rows, int cols,
                                                   float* __restrict__ dst, long ldd, long strideD, int rpb, float scale,
                                                   int accumulate) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rq = threadIdx.x >> 6;
  const int rb = blockIdx.y, batch = blockIdx.z;
  const float* x = X + batch * strideX;
  const int r0 = rb * rpb, r1 = min(rows, r0 + rpb);
  float s0 = 0.0f, s1 = 0.0f;
  if (c < cols) {
    int r = r0 + rq;
    for (; r + 4 < r1; r += 8) {  // two independent chains per thread
      s0 += x[(long)r * ldx + c];
      s1 += x[(long)(r + 4) * ldx + c];
    }
    if (r < r1) s0 += x[(long)r * ldx + c];
  }
  red[rq][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (rq == 0 && c < cols) {
    const int l = threadIdx.x;
    float v = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) * scale;
    float* o = dst + batch * strideD + (long)rb * ldd + c;
    *o = accumulate ? *o + v : v;
  }
}
}  // namespace

extern "C" long lfi_colsum_work_floats(int rows, int cols, int batch) {
  return (long)batch * lfi_cdiv(rows, cs_rows_per_block(rows)) * cols;
}

extern "C" int lfi_colsum_f32(const float* X, long ldx, long strideX, int rows, int cols, int batch, float* out,
                              long strideOut, float scale, int accumulate, float* work, void* stream) {
  LFI_REQUIRE(X && out && work, "lfi_colsum_f32: null pointer");
  LFI_REQUIRE(rows >= 1 && cols >= 1 && batch >= 1 && batch <= 65535, "lfi_colsum_f32: bad dims");
  hipStream_t st = (hipStream_t)stream;
  const int rpb = cs_rows_per_block(rows);
  const int nrb = lfi_cdiv(rows, rpb);
  if (nrb == 1) {
    hipLaunchKernelGGL(colsum_pass, dim3(lfi_cdiv(cols, 64), 1, batch), dim3(256), 0, st, X, ldx, strideX, rows, cols, out,
                       (long)0, strideOut, rows, scale, accumulate);
    LFI_LAUNCH_CHECK("lfi_colsum_f32");
    return LFI_OK;
  }
  hipLaunchKernelGGL(colsum_pass, dim3(lfi_cdiv(cols, 64), nrb, batch), dim3(256), 0, st, X, ldx, strideX, rows, cols, work,
                     (long)cols, (long)nrb * cols, rpb, 1.0f, 0);
  LFI_LAUNCH_CHECK("lfi_colsum_f32 pass 1");
  hipLaunchKernelGGL(colsum_pass, dim3(lfi_cdiv(cols, 64), 1, batch), dim3(256), 0, st, (const float*)work, (long)cols,
                     (long)nrb * cols, nrb, cols, out, (long)0, strideOut, nrb, scale, accumulate);
  LFI_LAUNCH_CHECK("lfi_colsum_f32 pass 2");
  return LFI_OK;
}

// ------------------------------------------------------------------ column fold / gather
namespace {
__global__ __launch_bounds__(256) void cols_fold_kernel(const float* __restrict__ src, long lds, long rows, const int* __restrict__ a,
                                                        const int* __restrict__ b, int ncols, float* __restrict__ dst, long ldd) {
  const long total = rows * ncols;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / ncols;
    const int j = (int)(idx - r * ncols);
    const int aj = a[j];
    float v = aj >= 0 ? src[r * lds + aj] : 0.0f;  // a[j] < 0: padding column
    if (b) {
      const int bj = b[j];
      if (bj >= 0) v += src[r * lds + bj];
    }
    dst[r * ldd + j] = v;
  }
}
}  // namespace

extern "C" int lfi_cols_fold(const float* src, long lds, long rows, const int* a, const int* b, int ncols, float* dst,
                             long ldd, void* stream) {
  LFI_REQUIRE(src && a && dst && rows > 0 && ncols > 0, "lfi_cols_fold: bad arguments");
  const long total = rows * ncols;
  const int blocks = (int)(lfi_cdiv(total, 256) < 4096 ? lfi_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(cols_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, lds, rows, a, b, ncols, dst, ldd);
  LFI_LAUNCH_CHECK("lfi_cols_fold");
  return LFI_OK;
}

// ------------------------------------------------------------------ optimiser
namespace {
constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ g, long n, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  // eight loads in flight per thread, added in the order the one-at-a-time loop adds them (17 M gradients took 33 us at the end
  // of the step, one outstanding 4-byte load per wave)
  for (; i + 7 * stride < n; i += 8 * stride) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = g[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u] * (double)v[u];
  }
  for (; i < n; i += stride) {
    const double v = (double)g[i];
    s += v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_stage2(const double* __restrict__ part, int nparts, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, const double* __restrict__ sumsq, float clip,
                                                        float gmul, float step_size, float beta1, float beta2, float eps,
                                                        float inv_sqrt_bc2, const float* __restrict__ hyper) {
  if (hyper) { step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; }   // device-resident (lfi_adam_clip_step_dev: captured steps)
  float coef = gmul;
  if (clip > 0.0f) {
    const double total = sqrt(sumsq[0]) * (double)fabsf(gmul);
    const double c = (double)clip / (total + 1e-6);
    if (c < 1.0) coef *= (float)c;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}
// torch.optim.Adam's two optional terms (single-tensor form, torch/optim/adam.py): weight_decay folds `wd * p` into the clipped
// gradient BEFORE the moments; amsgrad keeps the running maximum of the second moment and divides by its root instead. Template
// switches: the plain kernel above stays the instruction stream every earlier parity figure was measured on.
template <bool WD, bool AMS>
__global__ __launch_bounds__(256) void adam_ex_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ vmax, long n,
                                                           const double* __restrict__ sumsq, float clip, float gmul, float step_size,
                                                           float beta1, float beta2, float eps, float inv_sqrt_bc2, float weight_decay,
                                                           const float* __restrict__ hyper) {
  if (hyper) { step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; }
  float coef = gmul;
  if (clip > 0.0f) {
    const double total = sqrt(sumsq[0]) * (double)fabsf(gmul);
    const double c = (double)clip / (total + 1e-6);
    if (c < 1.0) coef *= (float)c;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i];
    float gi = g[i] * coef;
    if (WD) gi = gi + weight_decay * pi;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float vd = vi;
    if (AMS) {
      vd = fmaxf(vmax[i], vi);
      vmax[i] = vd;
    }
    const float denom = sqrtf(vd) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}
// clip coefficient shared by every flat-buffer optimiser: torch.nn.utils.clip_grad_norm_ (coef = clip / (norm + 1e-6), applied when < 1)
__device__ __forceinline__ float clip_coef(const double* __restrict__ sumsq, float clip, float gmul) {
  float coef = gmul;
  if (clip > 0.0f) {
    const double total = sqrt(sumsq[0]) * (double)fabsf(gmul);
    const double c = (double)clip / (total + 1e-6);
    if (c < 1.0) coef *= (float)c;
  }
  return coef;
}

// torch.optim.SGD (single-tensor form): g += wd p; buf = g on the first step, else momentum buf + (1 - dampening) g;
// nesterov: g += momentum buf, else g = buf; p -= lr g
__global__ __launch_bounds__(256) void sgd_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                                                       const double* __restrict__ sumsq, float clip, float gmul, float lr,
                                                       float momentum, float dampening, float weight_decay, int nesterov, int first) {
  const float coef = clip_coef(sumsq, clip, gmul);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i];
    float gi = g[i] * coef;
    if (weight_decay != 0.0f) gi = gi + weight_decay * pi;
    if (momentum != 0.0f) {
      const float bi = first ? gi : momentum * buf[i] + (1.0f - dampening) * gi;
      buf[i] = bi;
      gi = nesterov ? gi + momentum * bi : bi;
    }
    p[i] = pi - lr * gi;
  }
}

// torch.optim.RMSprop (single-tensor form): g += wd p; sq = alpha sq + (1 - alpha) g^2; centered: gavg = lerp(gavg, g, 1 - alpha),
// avg = sqrt(sq - gavg^2) + eps, else avg = sqrt(sq) + eps; momentum: buf = momentum buf + g / avg, p -= lr buf; else p -= lr g / avg
__global__ __launch_bounds__(256) void rmsprop_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                           float* __restrict__ buf, float* __restrict__ gavg, long n,
                                                           const double* __restrict__ sumsq, float clip, float gmul, float lr, float alpha,
                                                           float eps, float weight_decay, float momentum) {
  const float coef = clip_coef(sumsq, clip, gmul);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i];
    float gi = g[i] * coef;
    if (weight_decay != 0.0f) gi = gi + weight_decay * pi;
    const float si = alpha * sq[i] + (1.0f - alpha) * gi * gi;
    sq[i] = si;
    float avg;
    if (gavg) {
      const float ga = gavg[i] + (1.0f - alpha) * (gi - gavg[i]);
      gavg[i] = ga;
      avg = sqrtf(si - ga * ga) + eps;
    } else {
      avg = sqrtf(si) + eps;
    }
    if (momentum > 0.0f) {
      const float bi = momentum * buf[i] + gi / avg;
      buf[i] = bi;
      p[i] = pi - lr * bi;
    } else {
      p[i] = pi - lr * (gi / avg);
    }
  }
}
}  // namespace

extern "C" int lfi_sgd_clip_step(float* p, const float* g, float* buf, long n, const double* sumsq, float clip, float gmul, float lr,
                                 float momentum, float dampening, float weight_decay, int nesterov, int step_count, void* stream) {
  LFI_REQUIRE(p && g && n >= 0 && step_count >= 1, "lfi_sgd_clip_step: bad arguments");
  LFI_REQUIRE(momentum == 0.0f || buf, "lfi_sgd_clip_step: momentum needs its buffer");
  LFI_REQUIRE(clip <= 0.0f || sumsq, "lfi_sgd_clip_step: clipping needs sumsq");
  LFI_REQUIRE(!nesterov || (momentum > 0.0f && dampening == 0.0f), "lfi_sgd_clip_step: nesterov needs momentum > 0 and dampening 0");
  int blocks = (int)min(2048L, (long)lfi_cdiv(n > 0 ? n : 1, 256));
  hipLaunchKernelGGL(sgd_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, sumsq, clip, gmul, lr, momentum,
                     dampening, weight_decay, nesterov ? 1 : 0, step_count == 1 ? 1 : 0);
  LFI_LAUNCH_CHECK("lfi_sgd_clip_step");
  return LFI_OK;
}

extern "C" int lfi_rmsprop_clip_step(float* p, const float* g, float* sq, float* buf, float* gavg, long n, const double* sumsq,
                                     float clip, float gmul, float lr, float alpha, float eps, float weight_decay, float momentum,
                                     void* stream) {
  LFI_REQUIRE(p && g && sq && n >= 0, "lfi_rmsprop_clip_step: bad arguments");
  LFI_REQUIRE(momentum <= 0.0f || buf, "lfi_rmsprop_clip_step: momentum needs its buffer");
  LFI_REQUIRE(clip <= 0.0f || sumsq, "lfi_rmsprop_clip_step: clipping needs sumsq");
  int blocks = (int)min(2048L, (long)lfi_cdiv(n > 0 ? n : 1, 256));
  hipLaunchKernelGGL(rmsprop_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, sq, buf, gavg, n, sumsq, clip, gmul, lr,
                     alpha, eps, weight_decay, momentum);
  LFI_LAUNCH_CHECK("lfi_rmsprop_clip_step");
  return LFI_OK;
}

extern "C" int lfi_grad_sumsq(const float* g, long n, double* sumsq, double* work, void* stream) {
  LFI_REQUIRE(g && sumsq && work && n >= 0, "lfi_grad_sumsq: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)min((long)SUMSQ_BLOCKS, (long)lfi_cdiv(n > 0 ? n : 1, 256));
  hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, st, g, n, work);
  LFI_LAUNCH_CHECK("lfi_grad_sumsq stage 1");
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, work, blocks, sumsq);
  LFI_LAUNCH_CHECK("lfi_grad_sumsq stage 2");
  return LFI_OK;
}

extern "C" int lfi_adam_clip_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float clip,
                                  float gmul, float lr, float beta1, float beta2, float eps, int step_count, void* stream) {
  LFI_REQUIRE(p && g && m && v && n >= 0 && step_count >= 1, "lfi_adam_clip_step: bad arguments");
  LFI_REQUIRE(clip <= 0.0f || sumsq, "lfi_adam_clip_step: clipping needs sumsq");
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)beta1, step_count), bc2 = 1.0 - pow((double)beta2, step_count);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int blocks = (int)min(2048L, (long)lfi_cdiv(n > 0 ? n : 1, 256));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, sumsq, clip, gmul, step_size, beta1,
                     beta2, eps, inv_sqrt_bc2, (const float*)nullptr);
  LFI_LAUNCH_CHECK("lfi_adam_clip_step");
  return LFI_OK;
}

extern "C" int lfi_adam_clip_step_ex(float* p, const float* g, float* m, float* v, float* vmax, long n, const double* sumsq, float clip,
                                     float gmul, float lr, float beta1, float beta2, float eps, float weight_decay, int step_count,
                                     const float* hyper, void* stream) {
  LFI_REQUIRE(p && g && m && v && n >= 0 && (hyper || step_count >= 1), "lfi_adam_clip_step_ex: bad arguments");
  LFI_REQUIRE(clip <= 0.0f || sumsq, "lfi_adam_clip_step_ex: clipping needs sumsq");
  LFI_REQUIRE(weight_decay >= 0.0f, "lfi_adam_clip_step_ex: weight_decay < 0 (torch.optim.Adam raises too)");
  float step_size = 0.0f, inv_sqrt_bc2 = 0.0f;
  if (!hyper) {
    const double bc1 = 1.0 - pow((double)beta1, step_count), bc2 = 1.0 - pow((double)beta2, step_count);
    step_size = (float)((double)lr / bc1);
    inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
  const int blocks = (int)min(2048L, (long)lfi_cdiv(n > 0 ? n : 1, 256));
  const bool wd = weight_decay != 0.0f, ams = vmax != nullptr;
#define LFI_ADAM_EX(W, A)                                                                                                          \
  hipLaunchKernelGGL((adam_ex_clip_kernel<W, A>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, n, sumsq, clip, \
                     gmul, step_size, beta1, beta2, eps, inv_sqrt_bc2, weight_decay, hyper)
  if (wd && ams) LFI_ADAM_EX(true, true);
  else if (wd) LFI_ADAM_EX(true, false);
  else if (ams) LFI_ADAM_EX(false, true);
  else LFI_ADAM_EX(false, false);
#undef LFI_ADAM_EX
  LFI_LAUNCH_CHECK("lfi_adam_clip_step_ex");
  return LFI_OK;
}

extern "C" int lfi_adam_clip_step_dev(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float clip,
                                      float gmul, float beta1, float beta2, float eps, const float* hyper, void* stream) {
  LFI_REQUIRE(p && g && m && v && n >= 0 && hyper, "lfi_adam_clip_step_dev: bad arguments");
  LFI_REQUIRE(clip <= 0.0f || sumsq, "lfi_adam_clip_step_dev: clipping needs sumsq");
  int blocks = (int)min(2048L, (long)lfi_cdiv(n > 0 ? n : 1, 256));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, clip, gmul, 0.0f, beta1,
                     beta2, eps, 0.0f, hyper);
  LFI_LAUNCH_CHECK("lfi_adam_clip_step_dev");
  return LFI_OK;
}
