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


// dst[r][0 .. cols) = src[r][0 .. cols), dst[r][cols .. ldd) = 0: zero-padded copy of a matrix whose rows are not 4-float
// granular (50-d faces, 27-d speech), so that the input projection takes the vector-load GEMM paths.
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, long rows, int cols, long lds,
                                                       float* __restrict__ dst, long ldd) {
  const long total = rows * ldd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ldd;
    const int c = (int)(i - r * ldd);
    dst[i] = c < cols ? src[r * lds + c] : 0.0f;
  }
}

// Dropout multipliers of the window encoders (nn.Dropout on ones(B, hist), glow/models.py:56-58): out[i] = 1 / keep with
// probability keep, else 0. Counter-based Philox4x32-10 keyed on (seed, call offset): element i is output (i & 3) of counter
// (i >> 2, segment, offset), so a launch is reproducible from (seed, offset) alone and all modalities share one launch.
struct MaskSegs { float* out[4]; long n[4]; float keep[4]; int count; };
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// key: a DEVICE pair (seed, offset) when non-null (lfi_dropout_masks_dev: a captured hipGraph replays the launch with the
// values the host put there before the replay), else the two arguments
__global__ __launch_bounds__(256) void dropout_masks_kernel(MaskSegs m, unsigned long long seed, unsigned long long offset,
                                                            const unsigned long long* __restrict__ key) {
  const int seg = blockIdx.y;
  if (seg >= m.count) return;
  if (key) { seed = key[0]; offset = key[1]; }
  const long n4 = (m.n[seg] + 3) >> 2;
  const float keep = m.keep[seg], inv = 1.0f / keep;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
    unsigned o[4];
    philox4x32_10((unsigned)q, (unsigned)(q >> 32) ^ ((unsigned)seg << 28), (unsigned)offset, (unsigned)(offset >> 32),
                  (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long i = 4 * q + e;
      // 24 random bits -> u in [0, 1): exact in fp32, P(u < keep) = ceil(keep * 2^24) / 2^24
      if (i < m.n[seg]) m.out[seg][i] = (float)(o[e] >> 8) * (1.0f / 16777216.0f) < keep ? inv : 0.0f;
    }
  }
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

extern "C" int lfi_pad_rows(const float* src, long rows, int cols, long lds, float* dst, long ldd, void* stream) {
  LFI_REQUIRE(rows >= 0 && cols > 0 && lds >= cols && ldd >= cols, "lfi_pad_rows: bad dims rows=%ld cols=%d lds=%ld ldd=%ld", rows, cols, lds, ldd);
  if (rows == 0) return LFI_OK;
  LFI_REQUIRE(src && dst, "lfi_pad_rows: null pointer");
  const long blocks = (rows * ldd + 255) / 256;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, src, rows, cols, lds, dst, ldd);
  LFI_LAUNCH_CHECK("lfi_pad_rows");
  return LFI_OK;
}

static int dropout_masks_launch(int count, float* const* out, const long* n, const float* keep, unsigned long long seed,
                                unsigned long long offset, const unsigned long long* key, void* stream);
extern "C" int lfi_dropout_masks(int count, float* const* out, const long* n, const float* keep, unsigned long long seed,
                                 unsigned long long offset, void* stream) {
  return dropout_masks_launch(count, out, n, keep, seed, offset, nullptr, stream);
}
extern "C" int lfi_dropout_masks_dev(int count, float* const* out, const long* n, const float* keep,
                                     const unsigned long long* seed_offset, void* stream) {
  LFI_REQUIRE(seed_offset, "lfi_dropout_masks_dev: null key");
  return dropout_masks_launch(count, out, n, keep, 0ull, 0ull, seed_offset, stream);
}
// step_params (device, 32 bytes): [0] mask seed, [1] mask call offset (u64 each), then step_size, 1 / sqrt(1 - beta2^t) (fp32): what
// changes from one optimiser step to the next when the step itself is a captured hipGraph. One 1-thread launch, values by value.
namespace {
__global__ void set_step_params_kernel(unsigned long long* p, unsigned long long seed, unsigned long long offset, float step_size,
                                       float inv_sqrt_bc2) {
  p[0] = seed; p[1] = offset;
  float* f = reinterpret_cast<float*>(p + 2);
  f[0] = step_size; f[1] = inv_sqrt_bc2;
}
}  // namespace
extern "C" int lfi_set_step_params(void* params, unsigned long long seed, unsigned long long offset, float step_size,
                                   float inv_sqrt_bc2, void* stream) {
  LFI_REQUIRE(params && (reinterpret_cast<uintptr_t>(params) & 7) == 0, "lfi_set_step_params: null / misaligned buffer");
  hipLaunchKernelGGL(set_step_params_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(params), seed,
                     offset, step_size, inv_sqrt_bc2);
  LFI_LAUNCH_CHECK("lfi_set_step_params");
  return LFI_OK;
}
static int dropout_masks_launch(int count, float* const* out, const long* n, const float* keep, unsigned long long seed,
                                unsigned long long offset, const unsigned long long* key, void* stream) {
  LFI_REQUIRE(count >= 0 && count <= 4, "lfi_dropout_masks: %d segments (at most 4)", count);
  if (count == 0) return LFI_OK;
  LFI_REQUIRE(out && n && keep, "lfi_dropout_masks: null pointer");
  MaskSegs m = {};
  long nmax = 0;
  for (int i = 0; i < count; ++i) {
    LFI_REQUIRE(n[i] >= 0 && (n[i] == 0 || out[i]) && keep[i] > 0.0f && keep[i] <= 1.0f, "lfi_dropout_masks: segment %d: n=%ld keep=%g", i, n[i], (double)keep[i]);
    m.out[i] = out[i]; m.n[i] = n[i]; m.keep[i] = keep[i];
    nmax = n[i] > nmax ? n[i] : nmax;
  }
  m.count = count;
  if (nmax == 0) return LFI_OK;
  const long blocks = ((nmax + 3) / 4 + 255) / 256;
  hipLaunchKernelGGL(dropout_masks_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), count), dim3(256), 0, (hipStream_t)stream, m, seed, offset, key);
  LFI_LAUNCH_CHECK("lfi_dropout_masks");
  return LFI_OK;
}


// ---------------------------------------------------------------------------------------------- range guard of the sampler
// max |v| over up to 8 fp32 arrays in ONE launch, as the BIT PATTERN of the maximum: |v| of finite floats order like their bits,
// +inf's pattern is above every finite one and a NaN's above +inf's, so the result reads back as NaN / inf when any input held
// one. (SeqGlow.inference's fp16-piece arithmetic wants its operands inside fp16's range; the engine reads this word through pinned
// memory on a side stream instead of one blocking reduction per tensor.)
namespace {
struct AbsMaxSegs { const float* p[8]; long n[8]; int count; };
__global__ __launch_bounds__(256) void absmax_kernel(AbsMaxSegs sg, unsigned* __restrict__ out) {
  unsigned m = 0u;
  for (int i = 0; i < sg.count; ++i) {
    const unsigned* __restrict__ p = reinterpret_cast<const unsigned*>(sg.p[i]);
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < sg.n[i]; j += (long)gridDim.x * 256) {
      const unsigned b = p[j] & 0x7fffffffu;
      m = b > m ? b : m;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
}  // namespace

extern "C" int lfi_absmax_f32(int count, const float* const* ptrs, const long* n, unsigned* out_bits, void* stream) {
  LFI_REQUIRE(count >= 0 && count <= 8, "lfi_absmax_f32: %d arrays (at most 8)", count);
  LFI_REQUIRE(out_bits && (count == 0 || (ptrs && n)), "lfi_absmax_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipError_t me = hipMemsetAsync(out_bits, 0, sizeof(unsigned), st);
  LFI_REQUIRE(me == hipSuccess, "lfi_absmax_f32: hipMemsetAsync: %s", hipGetErrorString(me));
  AbsMaxSegs sg = {};
  long total = 0;
  for (int i = 0; i < count; ++i) {
    LFI_REQUIRE(n[i] >= 0 && (n[i] == 0 || ptrs[i]), "lfi_absmax_f32: array %d: n = %ld", i, n[i]);
    sg.p[i] = ptrs[i]; sg.n[i] = n[i]; total += n[i];
  }
  sg.count = count;
  if (total == 0) return LFI_OK;
  const int blocks = (int)(lfi_cdiv(total, 256 * 8) < 2048 ? (lfi_cdiv(total, 256 * 8) > 0 ? lfi_cdiv(total, 256 * 8) : 1) : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, sg, out_bits);
  LFI_LAUNCH_CHECK("lfi_absmax_f32");
  return LFI_OK;
}
