// Callers either side of the flow (SURVEY.md par. 8f): the GPU-resident window sampler that stands in for MimicryDataset
// (code/glow_pytorch/mimicry_data_module.py:33-78) and the jerk metric of MimicryLogger
// (code/glow_pytorch/mimicry_logger.py:187-196 -> glow/utils.py:53-58). Both are pure HBM streaming: coalesced row copies
// and a fixed-order two-stage reduction (no float atomics, reproducible).
#include "lfi_common.h"

namespace {

// dst[b, t, :] = src[starts[b] + t, :]   (one training window per sample: T consecutive frames of one recording bin)
// A window is ONE contiguous run of T * dim floats in the source: a workgroup copies it with 16-byte accesses when both
// sides are 16-byte aligned (dim % 4 == 0 is not required: alignment is tested per window), else 4-byte.
__global__ __launch_bounds__(256) void gather_sequences_kernel(const float* __restrict__ src, long rows, int dim,
                                                               const long* __restrict__ starts, int T,
                                                               float* __restrict__ dst) {
  const int b = blockIdx.x;
  long s = starts[b];
  if (s < 0) s = 0;                      // host validates the index table; clamp anyway: never read outside src
  if (s + T > rows) s = rows - T;
  const long n = (long)T * dim;
  const float* p = src + s * dim;
  float* q = dst + (long)b * n;
  if ((((unsigned long long)p | (unsigned long long)q) & 15ull) == 0ull) {
    const long n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float4* q4 = reinterpret_cast<float4*>(q);
    for (long i = threadIdx.x; i < n4; i += 256) q4[i] = p4[i];
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) q[i] = p[i];
  } else {
    for (long i = threadIdx.x; i < n; i += 256) q[i] = p[i];
  }
}

// calc_jerk (glow/utils.py:53-58): mean |x[t+3] - 3 x[t+2] + 3 x[t+1] - x[t]| over (B, T-3, C). Stage 1: per-block partial
// sums in fp64; stage 2: one block adds them in index order and divides.
__global__ __launch_bounds__(256) void jerk_stage1(const float* __restrict__ x, long B, int T, int C, double* __restrict__ part) {
  __shared__ double red[256];
  const long per = (long)(T - 3) * C, total = B * per;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / per, r = i - b * per;     // r = t * C + c
    const float* p = x + b * (long)T * C + r;
    // the reference differences three times in fp32: d1, d2 = d1' - d1, d3 = d2' - d2
    const float x0 = p[0], x1 = p[C], x2 = p[2 * (long)C], x3 = p[3 * (long)C];
    const float d10 = x1 - x0, d11 = x2 - x1, d12 = x3 - x2;
    const float d20 = d11 - d10, d21 = d12 - d11;
    acc += (double)fabsf(d21 - d20);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(64) void jerk_stage2(const double* __restrict__ part, int nparts, double count, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += part[i];
  out[0] = (float)(s / count);
}

}  // namespace

extern "C" int lfi_gather_sequences(const float* src, long rows, int dim, const long* starts, int B, int T, float* dst,
                                    void* stream) {
  LFI_REQUIRE(src && starts && dst, "lfi_gather_sequences: null pointer");
  LFI_REQUIRE(rows > 0 && dim > 0 && B > 0 && T > 0 && T <= rows, "lfi_gather_sequences: bad dims (rows %ld, dim %d, B %d, T %d)",
              rows, dim, B, T);
  hipLaunchKernelGGL(gather_sequences_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, src, rows, dim, starts, T, dst);
  LFI_LAUNCH_CHECK("lfi_gather_sequences");
  return LFI_OK;
}

extern "C" int lfi_jerk_mean(const float* x, int B, int T, int C, float* out, double* work, void* stream) {
  LFI_REQUIRE(x && out && work, "lfi_jerk_mean: null pointer");
  LFI_REQUIRE(B > 0 && C > 0 && T > 3, "lfi_jerk_mean: needs at least 4 frames (B %d, T %d, C %d)", B, T, C);
  const long total = (long)B * (T - 3) * C;
  const int blocks = (int)(lfi_cdiv(total, 256) < 1024 ? lfi_cdiv(total, 256) : 1024);
  hipLaunchKernelGGL(jerk_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)B, T, C, work);
  hipLaunchKernelGGL(jerk_stage2, dim3(1), dim3(64), 0, (hipStream_t)stream, work, blocks, (double)total, out);
  LFI_LAUNCH_CHECK("lfi_jerk_mean");
  return LFI_OK;
}
