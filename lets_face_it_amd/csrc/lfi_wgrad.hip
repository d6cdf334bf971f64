// The flow's thin weight-gradient products in ONE pass over the backward walk's stash (round 6), gfx950.
//
// What it replaces (lfi_flow_param_grads, lfi_flow.hip): per flow step k four products whose contraction runs over the F = N B
// frames and whose outputs are a handful of tiles - autograd of the coupling net's GRU cell and LinearZeros
// (glow/models.py:204-214, glow/modules.py:93-95) and of the invertible 1x1 convolution (glow/modules.py:147-177):
//     w_hh[k]         (G x H)    = sum_f dgh[k][f]^T  h[k][f - B]        (f >= B: the cell's previous state)
//     w_ih[k][:, :Ch] (G x Ch)   = sum_f dgi[k][f]^T  z1[k][f]
//     w_fl[k]         (Cout x H) = sum_f dlin[k][f]^T h[k][f]
//     dW[k]           (C x C)    = sum_f a[k][f]^T    dy[k][f]
//     b_fl[k]         (Cout)     = sum_f dlin[k][f]
// As four batched split-K launches of the generic kernel they read h three + one times and every gradient row once per 128-column
// output tile, at L2 hit rates of 19 - 41 % (profiles/round5_pmc_summary.md): 1.16 GB per step for 0.66 GB of distinct bytes, plus
// a column-sum pass over dlin and five split-K reduce launches. Here workgroup (k, 16-sample batch tile, range of timesteps) walks
// its timesteps ONCE: per timestep it brings the 16 rows of dgh | dgi (bf16, as the walk leaves them in two-product mode), h, dlin,
// z1, a, dy into LDS (h split into bf16 hi + lo on the way, the A operands dlin and a rounded to bf16 - the two-product form of
// every backward class), and its 8 waves accumulate all four outputs in registers: 72 tiles of
// v_mfma_f32_32x32x16_bf16, 9 per wave, two products each (a_hi b_lo + a_hi b_hi). Because the workgroup keeps ONE batch tile and
// walks n, the h tile of timestep n - 1 that w_hh needs is the one the previous iteration loaded for w_fl: h is read once.
// Partials go to a workspace [k][split][PART]; wgrad_reduce_kernel sums the splits in a fixed order (deterministic, no atomics)
// and scatters into the gradient buffers.
//
// Two roles, one launch each (one kernel holding all 72 output tiles needs 144 accumulator + 48 staging registers per lane: it
// spilled at the 256-register cap of a 512-thread workgroup): role 0 = w_hh | w_fl | b_fl (reads dgh, h, dlin; 56 tiles, 7 per
// wave), role 1 = w_ih[:, :Ch] | dW (reads dgi, z1, a, dy; 16 tiles, 2 per wave, two workgroups per CU). Every array is still read
// exactly once per step.
//
// Bound: HBM. Per 16 frames the two roles move 46 KB and issue 144 MFMAs (1 150 cycles per SIMD pair against ~3 500 cycles of a
// CU's share of 8 TB/s); loads run two timesteps ahead in registers.
#include "lfi_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
extern __shared__ __attribute__((aligned(16))) __bf16 wsmem[];

constexpr int WNT = 512;          // threads per workgroup
constexpr int WR = 16;            // frames (one batch tile of one timestep) per stage
constexpr int WH = 128, WG_ = 384;  // the shapes these kernels are built for (final_model.yaml: H = 128, GRU: G = 3 H)
// bf16 pitches of the k-major LDS images ([frame][column], fragments by the transposing ds_read_b64_tr_b16): 64 or 192 mod 256
// bytes, so the four rows a 16-lane group of the transposing read touches fall into four different 64-byte bank groups
constexpr int PG = WG_ + 32;      // 832 B
constexpr int PH = WH + 32;       // 320 B
constexpr int PS = 64 + 32;       // 192 B (images of at most 64 columns)
// role 0 stage: dgh, dlin (rounded) - and a ring of three h tiles (hi | lo) beside the two stages
constexpr int S0_GH = 0, S0_DL = WR * PG, STAGE0 = S0_DL + WR * PS;
constexpr int HSLOT = 2 * WR * PH;
constexpr int LDS0_BF16 = 2 * STAGE0 + 3 * HSLOT;      // 63 488 bytes
// role 1 stage: dgi, a (rounded), dy hi, dy lo, z1 hi, z1 lo
constexpr int S1_GI = 0, S1_SA = WR * PG, S1_DYH = S1_SA + WR * PS, S1_DYL = S1_DYH + WR * PS, S1_ZH = S1_DYL + WR * PS,
              S1_ZL = S1_ZH + WR * PS, STAGE1 = S1_ZL + WR * PS;
constexpr int LDS1_BF16 = 2 * STAGE1;                  // 57 344 bytes
// partial layouts of one (flow step, batch tile, timestep range), floats. Role 0: w_hh | w_fl | b_fl; role 1: w_ih[:, :Ch] | dW
constexpr int P0_HH = 0, P0_FL = WG_ * WH, P0_BF = P0_FL + 64 * WH, PART0 = P0_BF + 64;
constexpr int P1_IZ = 0, P1_DW = WG_ * 32, PART1 = P1_DW + 64 * 64;

struct WgradArgs {
  int B, N, Ks, F, C, Ch, Cout, I, ldc, ldo;
  int nbt, ns, nchunk;           // batch tiles, timestep ranges per (k, tile), timesteps per range - of the role launched
  const __bf16 *dgh, *dgi;       // [Ks][F][G] bf16
  const float *h, *dlin, *sY, *sA, *dy;
  float* part;                   // [Ks][nbt * ns][PART0 | PART1]
  // reduce
  float *w_hh, *w_ih, *w_fl, *b_fl, *dW;
  int accumulate;
};

__device__ __forceinline__ void split2w(float a, float b, unsigned* hi, unsigned* lo) {
  const bf16x2 h = __builtin_convertvector((float2_t){a, b}, bf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float ha = __builtin_bit_cast(float, hb << 16), hbv = __builtin_bit_cast(float, hb & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector((float2_t){a - ha, b - hbv}, bf16x2);
  *hi = hb;
  *lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ unsigned round2w(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((float2_t){a, b}, bf16x2));
}
// four floats -> LDS as bf16 hi (+ lo) at element offset `at`
__device__ __forceinline__ void put_split(__bf16* hi_img, __bf16* lo_img, int at, const f32x4& v) {
  unsigned h0, l0, h1, l1;
  split2w(v[0], v[1], &h0, &l0);
  split2w(v[2], v[3], &h1, &l1);
  *reinterpret_cast<u32x2*>(hi_img + at) = (u32x2){h0, h1};
  *reinterpret_cast<u32x2*>(lo_img + at) = (u32x2){l0, l1};
}
__device__ __forceinline__ void put_round(__bf16* img, int at, const f32x4& v) {
  const u32x2 h = {round2w(v[0], v[1]), round2w(v[2], v[3])};
  *reinterpret_cast<u32x2*>(img + at) = h;
}

// MFMA 32x32x16 operand fragment out of a k-major image of 16 frames: lane l gets column mn + (l & 31), frames 8 (l >> 5) .. + 7
template <int PIT>
__device__ __forceinline__ bf16x8 tfrag(const __bf16* img, int mn, int lane) {
  const int i = lane & 15, q = i >> 2, pp = i & 3;
  const __bf16* ptr = img + (8 * (lane >> 5) + q) * PIT + mn + 16 * ((lane >> 4) & 1) + 4 * pp;
  typedef __attribute__((address_space(3))) bf16x4 lds_v4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)ptr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(ptr + 4 * PIT));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
__device__ __forceinline__ f32x16 mfma2(const bf16x8& a, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc, 0, 0, 0);
}
// accumulator tile -> rows row0 .., columns col0 .. of a row-major matrix
__device__ __forceinline__ void put_tile(const f32x16& acc, float* base, int ld, int row0, int col0, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) base[(row0 + (r & 3) + 8 * (r >> 2) + 4 * half) * ld + col0 + l31] = acc[r];
}

// what one thread has in flight for one timestep: two 16-byte chunks of the 16 x 384 bf16 gradient rows (768 chunks: the second
// slot on threads 0-255 only) and two chunks of four floats of the fp32 arrays
struct StageRegs {
  u32x4 g0, g1;
  f32x4 a, b;
};

// The walk of one workgroup over its timesteps: loads two timesteps ahead in registers, LDS stages double-buffered, one barrier
// per timestep. LOAD(n, regs), STORE(n, buf, regs), MMA(n, buf). The steady loop has no branch between a load and the store that
// consumes it two phases later (every load is unconditional - slots a thread does not own read an address it does own - so the
// s_waitcnt before a store counts the younger stage's loads instead of draining them: vmcnt(0) there would halve the bytes in flight).
#define LFI_WGRAD_WALK(LOAD, STORE, MMA)                     \
  {                                                          \
    StageRegs R0, R1;                                        \
    LOAD(n0, R0);                                            \
    LOAD(min(n0 + 1, n1 - 1), R1);                           \
    STORE(n0, 0, R0);                                        \
    __syncthreads();                                         \
    int n = n0;                                              \
    for (; n + 3 < n1; n += 2) {                             \
      LOAD(n + 2, R0);                                       \
      __builtin_amdgcn_sched_barrier(0);                     \
      MMA(n, 0);                                             \
      __builtin_amdgcn_sched_barrier(0);                     \
      STORE(n + 1, 1, R1);                                   \
      __syncthreads();                                       \
      LOAD(n + 3, R1);                                       \
      __builtin_amdgcn_sched_barrier(0);                     \
      MMA(n + 1, 1);                                         \
      __builtin_amdgcn_sched_barrier(0);                     \
      STORE(n + 2, 0, R0);                                   \
      __syncthreads();                                       \
    }                                                        \
    for (; n < n1; n += 2) {                                 \
      if (n + 2 < n1) LOAD(n + 2, R0);                       \
      MMA(n, 0);                                             \
      if (n + 1 < n1) STORE(n + 1, 1, R1);                   \
      __syncthreads();                                       \
      if (n + 1 < n1) {                                      \
        MMA(n + 1, 1);                                       \
        if (n + 2 < n1) STORE(n + 2, 0, R0);                 \
        __syncthreads();                                     \
      }                                                      \
    }                                                        \
  }

// ------------------------------------------------------------------------------------------------ role 0: w_hh | w_fl | b_fl
__global__ __launch_bounds__(WNT, 1) void flow_wgrad_hh_kernel(WgradArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.y;
  const int bt = blockIdx.x % g.nbt, sp = blockIdx.x / g.nbt;
  const int n0 = sp * g.nchunk, n1 = min(g.N, n0 + g.nchunk);
  const long kF = (long)k * g.F;
  const int b0 = bt * WR;
  float* __restrict__ P = g.part + ((long)k * (g.nbt * g.ns) + blockIdx.x) * PART0;
  if (n0 >= n1) return;   // (an empty range: wgrad_splits never makes one)

  // per-thread load slots, fixed for the whole walk: one pointer per slot at timestep 0 of this batch tile
  const int r0 = tid / 48, c0 = tid - r0 * 48;                    // dgh chunk tid: row, 16-byte chunk
  const int r1 = (tid + WNT) / 48, c1 = (tid + WNT) - r1 * 48;    // dgh chunk tid + 512 (threads 0-255)
  const bool g1ok = tid < WR * 48 - WNT;
  const __bf16* gp0 = g.dgh + (kF + b0 + r0) * WG_ + c0 * 8;
  const __bf16* gp1 = g1ok ? g.dgh + (kF + b0 + r1) * WG_ + c1 * 8 : gp0;
  const int gl0 = S0_GH + r0 * PG + c0 * 8, gl1 = S0_GH + r1 * PG + c1 * 8;
  const int hrow = tid >> 5, hch = tid & 31;                      // h: 16 rows x 32 chunks of four floats
  const float* hp = g.h + (kF + b0 + hrow) * WH + hch * 4;
  const int hl = hrow * PH + hch * 4;
  const int srow = (tid >> 4) & 15, sch = tid & 15;               // dlin: 16 rows x 16 chunk slots, threads 0-255
  const bool sok = tid < 256 && sch < (g.ldo >> 2);
  const float* sp0 = g.dlin + (kF + b0 + srow) * g.ldo + (sch < (g.ldo >> 2) ? sch * 4 : 0);   // (a slot past the row re-reads its first chunk)
  const int sl = srow * PS + sch * 4;
  const long stepG = (long)g.B * WG_, stepH = (long)g.B * WH, stepS = (long)g.B * g.ldo;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};     // column sums of dlin

  auto hslot = [&](int n) { return wsmem + 2 * STAGE0 + (n % 3) * HSLOT; };
#define LOAD0(n_, R)                                                                      \
  {                                                                                       \
    R.g0 = *reinterpret_cast<const u32x4*>(gp0 + (n_) * stepG);                           \
    R.g1 = *reinterpret_cast<const u32x4*>(gp1 + (n_) * stepG);                           \
    R.a = *reinterpret_cast<const f32x4*>(hp + (n_) * stepH);                             \
    R.b = *reinterpret_cast<const f32x4*>(sp0 + (n_) * stepS);                            \
  }
#define STORE0(n_, buf, R)                                                                \
  {                                                                                       \
    __bf16* st = wsmem + (buf) * STAGE0;                                                  \
    *reinterpret_cast<u32x4*>(st + gl0) = R.g0;                                           \
    if (g1ok) *reinterpret_cast<u32x4*>(st + gl1) = R.g1;                                 \
    __bf16* hs = hslot(n_);                                                               \
    put_split(hs, hs + WR * PH, hl, R.a);                                                 \
    if (tid < 256) {                                                                      \
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};                                              \
      const f32x4 dv = sok ? R.b : z4;                                                    \
      bsum += dv;                                                                         \
      put_round(st + S0_DL, sl, dv);                                                      \
    }                                                                                     \
  }
  // accumulators: 6 tiles of w_hh (rows 96 wm + 32 mt, columns 64 wn + 32 nt), 1 of w_fl (row tile wave >> 2, column tile wave & 3)
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 ahh[3][2], afl;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) ahh[i][j][r] = 0.0f;
    afl[r] = 0.0f;
  }
#define MMA0(n_, buf)                                                                     \
  {                                                                                       \
    const __bf16* st = wsmem + (buf) * STAGE0;                                            \
    const __bf16* hcur = hslot(n_);                                                       \
    if ((n_) > 0) {                                                                       \
      const __bf16* hprev = hslot((n_) + 2);                                              \
      bf16x8 bh[2], bl[2];                                                                \
      _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                  \
        bh[nt] = tfrag<PH>(hprev, 64 * wn + 32 * nt, lane);                               \
        bl[nt] = tfrag<PH>(hprev + WR * PH, 64 * wn + 32 * nt, lane);                     \
      }                                                                                   \
      _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) {                                  \
        const bf16x8 a = tfrag<PG>(st + S0_GH, 96 * wm + 32 * mt, lane);                  \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) ahh[mt][nt] = mfma2(a, bh[nt], bl[nt], ahh[mt][nt]); \
      }                                                                                   \
    }                                                                                     \
    {                                                                                     \
      const bf16x8 a = tfrag<PS>(st + S0_DL, 32 * (wave >> 2), lane);                     \
      const bf16x8 bh = tfrag<PH>(hcur, 32 * (wave & 3), lane), bl = tfrag<PH>(hcur + WR * PH, 32 * (wave & 3), lane); \
      afl = mfma2(a, bh, bl, afl);                                                        \
    }                                                                                     \
  }
  if (n0 > 0) {   // the previous range's last h tile (w_hh of timestep n0)
    const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + (n0 - 1) * stepH);
    __bf16* hs = hslot(n0 - 1);
    put_split(hs, hs + WR * PH, hl, hv);
  }
  LFI_WGRAD_WALK(LOAD0, STORE0, MMA0)
#undef LOAD0
#undef STORE0
#undef MMA0

#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) put_tile(ahh[mt][nt], P + P0_HH, WH, 96 * wm + 32 * mt, 64 * wn + 32 * nt, lane);
  put_tile(afl, P + P0_FL, WH, 32 * (wave >> 2), 32 * (wave & 3), lane);
  // column sums of dlin: 16 row slots per column chunk, summed in row order
  float* red = reinterpret_cast<float*>(wsmem);
  __syncthreads();
  if (tid < 256) *reinterpret_cast<f32x4*>(red + srow * 64 + sch * 4) = bsum;
  __syncthreads();
  if (tid < 64) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < WR; ++r) s += red[r * 64 + tid];
    P[P0_BF + tid] = s;
  }
}

// ------------------------------------------------------------------------------------------------ role 1: w_ih[:, :Ch] | dW
__global__ __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(4, 4))) void flow_wgrad_ih_kernel(WgradArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.y;
  const int bt = blockIdx.x % g.nbt, sp = blockIdx.x / g.nbt;
  const int n0 = sp * g.nchunk, n1 = min(g.N, n0 + g.nchunk);
  const long kF = (long)k * g.F;
  const int b0 = bt * WR;
  float* __restrict__ P = g.part + ((long)k * (g.nbt * g.ns) + blockIdx.x) * PART1;
  if (n0 >= n1) return;

  const int r0 = tid / 48, c0 = tid - r0 * 48;
  const int r1 = (tid + WNT) / 48, c1 = (tid + WNT) - r1 * 48;
  const bool g1ok = tid < WR * 48 - WNT;
  const __bf16* gp0 = g.dgi + (kF + b0 + r0) * WG_ + c0 * 8;
  const __bf16* gp1 = g1ok ? g.dgi + (kF + b0 + r1) * WG_ + c1 * 8 : gp0;
  const int gl0 = S1_GI + r0 * PG + c0 * 8, gl1 = S1_GI + r1 * PG + c1 * 8;
  // fp32 arrays, 16 rows x 16 chunk slots each: slot a = a (threads 0-255) | dy (256-511); slot b = z1 (threads 0-255, its first
  // 32 columns at most)
  const int srow = (tid >> 4) & 15, sch = tid & 15, ncc = g.ldc >> 2;
  const bool hi_half = tid >= 256;
  const bool aok = sch < ncc, bok = !hi_half && sch < min(ncc, 8);
  const float* ap = (hi_half ? g.dy : g.sA) + (kF + b0 + srow) * g.ldc + (aok ? sch * 4 : 0);   // (a slot past the row: its first chunk)
  const float* bp = bok ? g.sY + (kF + b0 + srow) * g.ldc + sch * 4 : ap;
  const int sl = srow * PS + sch * 4;
  const long stepG = (long)g.B * WG_, stepS = (long)g.B * g.ldc;

#define LOAD1(n_, R)                                                                      \
  {                                                                                       \
    R.g0 = *reinterpret_cast<const u32x4*>(gp0 + (n_) * stepG);                           \
    R.g1 = *reinterpret_cast<const u32x4*>(gp1 + (n_) * stepG);                           \
    R.a = *reinterpret_cast<const f32x4*>(ap + (n_) * stepS);                             \
    R.b = *reinterpret_cast<const f32x4*>(bp + (n_) * stepS);                             \
  }
#define STORE1(n_, buf, R)                                                                \
  {                                                                                       \
    __bf16* st = wsmem + (buf) * STAGE1;                                                  \
    *reinterpret_cast<u32x4*>(st + gl0) = R.g0;                                           \
    if (g1ok) *reinterpret_cast<u32x4*>(st + gl1) = R.g1;                                 \
    if (hi_half) put_split(st + S1_DYH, st + S1_DYL, sl, R.a);                            \
    else put_round(st + S1_SA, sl, R.a);                                                  \
    if (!hi_half && sch < 8) put_split(st + S1_ZH, st + S1_ZL, sl, R.b);                  \
  }
  // accumulators: w_ih[:, :Ch] row tile `wave`, then row tile 8 + wave on waves 0-3 | dW tile (dwm, dwn) on waves 4-7
  f32x16 ax0, ax1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { ax0[r] = 0.0f; ax1[r] = 0.0f; }
  const bool lowave = wave < 4;
  const int dwm = (wave - 4) >> 1, dwn = (wave - 4) & 1;
#define MMA1(n_, buf)                                                                     \
  {                                                                                       \
    const __bf16* st = wsmem + (buf) * STAGE1;                                            \
    const bf16x8 zh = tfrag<PS>(st + S1_ZH, 0, lane), zl = tfrag<PS>(st + S1_ZL, 0, lane); \
    const bf16x8 a = tfrag<PG>(st + S1_GI, 32 * wave, lane);                              \
    ax0 = mfma2(a, zh, zl, ax0);                                                          \
    if (lowave) {                                                                         \
      const bf16x8 a2 = tfrag<PG>(st + S1_GI, 32 * (8 + wave), lane);                     \
      ax1 = mfma2(a2, zh, zl, ax1);                                                       \
    } else {                                                                              \
      const bf16x8 a2 = tfrag<PS>(st + S1_SA, 32 * dwm, lane);                            \
      const bf16x8 dh = tfrag<PS>(st + S1_DYH, 32 * dwn, lane), dl = tfrag<PS>(st + S1_DYL, 32 * dwn, lane); \
      ax1 = mfma2(a2, dh, dl, ax1);                                                       \
    }                                                                                     \
  }
  LFI_WGRAD_WALK(LOAD1, STORE1, MMA1)
#undef LOAD1
#undef STORE1
#undef MMA1
  put_tile(ax0, P + P1_IZ, 32, 32 * wave, 0, lane);
  if (lowave) put_tile(ax1, P + P1_IZ, 32, 32 * (8 + wave), 0, lane);
  else put_tile(ax1, P + P1_DW, 64, 32 * dwm, 32 * dwn, lane);
}

// out[k][j .. j + 3] = sum over the splits (in split order) of part[k][split][j .. j + 3], scattered to the gradient buffers
template <int ROLE>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs g) {
  constexpr int PART = ROLE == 0 ? PART0 : PART1;
  const int k = blockIdx.y;
  const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= PART) return;
  const int S = g.nbt * g.ns;
  const float* __restrict__ p = g.part + (long)k * S * PART + j;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int i = 0;
  for (; i + 4 <= S; i += 4) {   // four loads in flight, summed in split order
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + (long)i * PART), b = *reinterpret_cast<const f32x4*>(p + (long)(i + 1) * PART);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p + (long)(i + 2) * PART), d = *reinterpret_cast<const f32x4*>(p + (long)(i + 3) * PART);
    s += a; s += b; s += c; s += d;
  }
  for (; i < S; ++i) s += *reinterpret_cast<const f32x4*>(p + (long)i * PART);
  const bool accf = g.accumulate != 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int je = j + e;
    float* dst = nullptr;
    bool acc = accf;
    if (ROLE == 0) {
      if (je < P0_FL) {
        dst = g.w_hh + (long)k * WG_ * WH + je;
      } else if (je < P0_BF) {
        const int row = (je - P0_FL) >> 7, col = (je - P0_FL) & 127;
        if (row < g.Cout) dst = g.w_fl + ((long)k * g.Cout + row) * WH + col;
      } else {
        const int col = je - P0_BF;
        if (col < g.Cout) dst = g.b_fl + (long)k * g.Cout + col;
      }
    } else {
      if (je < P1_DW) {
        const int row = je >> 5, col = je & 31;
        if (col < g.Ch) dst = g.w_ih + ((long)k * WG_ + row) * g.I + col;
      } else {
        const int row = (je - P1_DW) >> 6, col = (je - P1_DW) & 63;
        if (row < g.C && col < g.C) dst = g.dW + ((long)k * g.C + row) * g.C + col;
        acc = false;
      }
    }
    if (dst) *dst = acc ? *dst + s[e] : s[e];
  }
}

// timestep ranges per (flow step, batch tile): enough workgroups for `want` of them, ranges of at least 4 timesteps
int wgrad_splits(int Ks, int nbt, int N, int want) {
  int ns = 1;
  while ((long)Ks * nbt * ns < want && N / (ns * 2) >= 4) ns *= 2;
  return ns;
}
// role 0 holds a CU per workgroup (210 VGPRs x 8 waves): one round of 256; role 1 fits twice per CU
constexpr int WANT0 = 256, WANT1 = 512;
}  // namespace

extern "C" __attribute__((visibility("hidden"))) int lfi_internal_flow_wgrad_ok(int B, int N, int C, int Ch, int Cout, int H, int G,
                                                                                 int ldc, int ldo) {
  return (H == WH && G == WG_ && B % WR == 0 && N >= 1 && C >= 1 && C <= 64 && Cout >= 1 && Cout <= 64 && Ch >= 1 && Ch <= 32 &&
          ldc % 4 == 0 && ldo % 4 == 0 && ldc <= 64 && ldo <= 64) ? 1 : 0;
}

// floats of the partials of role 0 (w_hh | w_fl | b_fl) and role 1 (w_ih[:, :Ch] | dW); the caller places role 1's after role 0's
extern "C" __attribute__((visibility("hidden"))) long lfi_internal_flow_wgrad_work_floats(int B, int N, int Ks, int role) {
  const int nbt = B / WR;
  return role == 0 ? (long)Ks * nbt * wgrad_splits(Ks, nbt, N, WANT0) * PART0 : (long)Ks * nbt * wgrad_splits(Ks, nbt, N, WANT1) * PART1;
}

// role 0: w_hh, w_fl, b_fl from dgh, h, dlin; role 1: w_ih[:, :Ch], dW from dgi, z1 (= sY's first Ch columns), a, dy. Each role is
// two launches on `stream` (walk + reduce); the roles share nothing but read-only inputs and may run on different streams.
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_flow_wgrad(int role, int B, int N, int Ks, int C, int Ch, int Cout, int I,
                                                                              int ldc, int ldo, const void* dgh, const void* dgi,
                                                                              const float* h, const float* dlin, const float* sY,
                                                                              const float* sA, const float* dy, float* part, float* w_hh,
                                                                              float* w_ih, float* w_fl, float* b_fl, float* dW,
                                                                              int accumulate, void* stream) {
  WgradArgs g = {};
  g.B = B; g.N = N; g.Ks = Ks; g.F = B * N; g.C = C; g.Ch = Ch; g.Cout = Cout; g.I = I; g.ldc = ldc; g.ldo = ldo;
  g.nbt = B / WR;
  g.ns = wgrad_splits(Ks, g.nbt, N, role == 0 ? WANT0 : WANT1);
  g.nchunk = (N + g.ns - 1) / g.ns;
  g.dgh = reinterpret_cast<const __bf16*>(dgh); g.dgi = reinterpret_cast<const __bf16*>(dgi);
  g.h = h; g.dlin = dlin; g.sY = sY; g.sA = sA; g.dy = dy; g.part = part;
  g.w_hh = w_hh; g.w_ih = w_ih; g.w_fl = w_fl; g.b_fl = b_fl; g.dW = dW; g.accumulate = accumulate;
  static bool attr_set = false;
  const size_t lds0 = (size_t)LDS0_BF16 * sizeof(__bf16), lds1 = (size_t)LDS1_BF16 * sizeof(__bf16);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flow_wgrad_hh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(flow_wgrad_ih_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (e != hipSuccess) {
      lfi_set_error("lfi_flow_param_grads (one-pass thin products): cannot reserve %zu / %zu bytes of LDS: %s", lds0, lds1, hipGetErrorString(e));
      return LFI_ERR_LAUNCH;
    }
    attr_set = true;
  }
  if (role == 0) {
    hipLaunchKernelGGL(flow_wgrad_hh_kernel, dim3(g.nbt * g.ns, Ks), dim3(WNT), lds0, (hipStream_t)stream, g);
    LFI_LAUNCH_CHECK("lfi_flow_param_grads one-pass thin products (w_hh | w_fl)");
    hipLaunchKernelGGL(wgrad_reduce_kernel<0>, dim3(lfi_cdiv(PART0 / 4, 256), Ks), dim3(256), 0, (hipStream_t)stream, g);
  } else {
    hipLaunchKernelGGL(flow_wgrad_ih_kernel, dim3(g.nbt * g.ns, Ks), dim3(WNT), lds1, (hipStream_t)stream, g);
    LFI_LAUNCH_CHECK("lfi_flow_param_grads one-pass thin products (w_ih | dW)");
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(lfi_cdiv(PART1 / 4, 256), Ks), dim3(256), 0, (hipStream_t)stream, g);
  }
  LFI_LAUNCH_CHECK("lfi_flow_param_grads one-pass thin products reduce");
  return LFI_OK;
}
