// Generic fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Why f32 MFMA: the reference computes this path in fp32 (hparams/final_model.yaml:141 `precision: 32`) and parity
// is gated at 1e-4 relative on per-frame NLL; v_mfma_f32_32x32x2_f32 is bit-for-bit an fp32 fma chain at
// 64 FLOP/clk/SIMD (157 TFLOP/s chip peak), so this kernel is MFMA-bound by construction: per 16-deep k-tile a wave
// issues 32 MFMAs (2048 cycles) against 16 global loads + 16 LDS stores + 32 LDS reads.
//
// Structure: block tile BM x BN (wave tile 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator VGPRs), k-tile 16, both
// operands staged through LDS k-major ([k][m] / [k][n]) so every MFMA operand read is a conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave; register-staged double buffering (global loads of tile t+1
// in flight under the MFMAs of tile t, one barrier per k-tile). Tiles are dealt to XCDs in contiguous runs so the
// blocks that share an A panel share an L2.
#include "lfi_gemm_common.h"

namespace {


// LDS images (floats). A k-contiguous operand keeps its rows: [mn][BKT + 4] (80-byte rows: 16-B aligned for
// ds_write_b128 / ds_read_b128, and 5*row mod 16 distinct over any 16 rows mod 16 -> conflict-free b128 reads). An
// mn-contiguous operand is stored [k][mn + 4] and read with ds_read_b32. Both are filled with 16-byte loads/stores when
// the operand is 16-byte aligned (vec flag), with four guarded scalar loads per float4 otherwise.
// The k index an MFMA slot multiplies is permuted identically for both operands (a sum over k does not care):
// within a group of 8 k, lane half h at MFMA step t (0..3) takes k = 8q + 4h + t, so a k-contiguous operand's four
// steps are ONE float4 read.
constexpr int KROW = BKT + 4;

template <int BMN, bool KC>
struct Stager {
  static constexpr int NF4 = BMN * BKT / 4 / 256;  // float4 per thread per k-tile
  static constexpr int LDS_FLOATS = KC ? BMN * KROW : BKT * (BMN + 4);
  int off[NF4];        // element offset of the float4 from the tile origin (32-bit)
  int kk[NF4];         // first k of the float4 inside the tile (KC) / the k row (!KC)
  int lds[NF4];        // float offset in the LDS image
  unsigned ok;         // bit i: the float4's mn index is inside the matrix
  __device__ __forceinline__ void init(int tid, int mn0, int MN, long ld) {
    ok = 0;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int f = tid + 256 * i;
      int mn, k;
      if (KC) { k = (f % (BKT / 4)) * 4; mn = f / (BKT / 4); } else { mn = (f % (BMN / 4)) * 4; k = f / (BMN / 4); }
      kk[i] = k;
      lds[i] = KC ? mn * KROW + k : k * (BMN + 4) + mn;
      const bool in = mn0 + mn < MN;
      ok |= (in ? 1u : 0u) << i;
      off[i] = in ? (KC ? mn * (int)ld + k : k * (int)ld + mn) : 0;
    }
  }
  // p: tile origin for this k-tile; krem: valid k in this tile; mnrem: valid mn from the tile origin.
  // Loads are predicated, never branched around, and nothing here consumes a loaded value: the loads of tile t+1 stay
  // in flight under the MFMAs of tile t. The tail masking of a partly valid float4 happens in store().
  template <bool VEC>
  __device__ __forceinline__ void load(const float* __restrict__ p, int krem, int mnrem, f32x4 (&r)[NF4]) const {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const bool in = ((ok >> i) & 1u) && kk[i] < krem;
      if (VEC) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r[i] = in ? *reinterpret_cast<const f32x4*>(p + off[i]) : z;
      } else if (KC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[i][j] = (in && kk[i] + j < krem) ? p[off[i] + j] : 0.0f;
      } else {
        const int mn = lds[i] - kk[i] * (BMN + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[i][j] = (in && mn + j < mnrem) ? p[off[i] + j] : 0.0f;
      }
    }
  }
  template <bool VEC>
  __device__ __forceinline__ void store(float* img, int krem, f32x4 (&r)[NF4]) const {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      if (VEC && KC) {  // a 16-byte load may have run past K inside the row: zero those lanes
#pragma unroll
        for (int j = 1; j < 4; ++j) r[i][j] = (kk[i] + j < krem) ? r[i][j] : 0.0f;
      }
      *reinterpret_cast<f32x4*>(img + lds[i]) = r[i];
    }
  }
};

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  static_assert(WM * WN == 4 && BM == WM * 64 && BN == WN * 64, "wave tile is 64 x 64");
  static_assert(BKT == 16, "fragment reads assume two groups of 8 k per tile");
  using SA = Stager<BM, AKC>;
  using SB = Stager<BN, BKC>;
  constexpr int MAIN_FLOATS = 2 * (SA::LDS_FLOATS + SB::LDS_FLOATS);
  constexpr int EPI_FLOATS = (BM == 128 && BN == 128) ? 64 * (128 + 4) : 0;   // wide epilogue: two passes of 64 rows
  __shared__ __attribute__((aligned(16))) float smem_f32[MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS];
  float (*As)[SA::LDS_FLOATS] = reinterpret_cast<float (*)[SA::LDS_FLOATS]>(smem_f32);
  float (*Bs)[SB::LDS_FLOATS] = reinterpret_cast<float (*)[SB::LDS_FLOATS]>(smem_f32 + 2 * SA::LDS_FLOATS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ A = g.A + batch * g.strideA;
  const float* __restrict__ B = g.B + batch * g.strideB;
  const int kbeg = split * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + BKT - 1) / BKT;

  SA sa;
  SB sb;
  sa.init(tid, m0, g.M, g.lda);
  sb.init(tid, n0, g.N, g.ldb);
  const float* __restrict__ tA = AKC ? A + (long)m0 * g.lda + kbeg : A + (long)kbeg * g.lda + m0;  // tile origin, uniform
  const float* __restrict__ tB = BKC ? B + (long)n0 * g.ldb + kbeg : B + (long)kbeg * g.ldb + n0;
  const long stepA = AKC ? BKT : (long)BKT * g.lda, stepB = BKC ? BKT : (long)BKT * g.ldb;
  const int mrem = g.M - m0, nrem = g.N - n0;

  f32x4 ra[SA::NF4], rb[SB::NF4];
  auto load_tiles = [&](int kt) {
    const int krem = kend - (kbeg + kt * BKT);
    sa.template load<VEC>(tA + kt * stepA, krem, mrem, ra);
    sb.template load<VEC>(tB + kt * stepB, krem, nrem, rb);
  };
  auto store_tiles = [&](int kt, int buf) {
    const int krem = kend - (kbeg + kt * BKT);
    sa.template store<VEC>(As[buf], krem, ra);
    sb.template store<VEC>(Bs[buf], krem, rb);
  };

  const int wm = wave / WN, wn = wave % WN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (nkt > 0) {
    load_tiles(0);
    store_tiles(0, 0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tiles(kt + 1);
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int row = wm * 64 + t2 * 32 + l31, col = wn * 64 + t2 * 32 + l31;
        if (AKC) {
          fa[t2] = *reinterpret_cast<const f32x4*>(as + row * KROW + 8 * q + 4 * half);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[t2][t] = as[(8 * q + 4 * half + t) * (BM + 4) + row];
        }
        if (BKC) {
          fb[t2] = *reinterpret_cast<const f32x4*>(bs + col * KROW + 8 * q + 4 * half);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) fb[t2][t] = bs[(8 * q + 4 * half + t) * (BN + 4) + col];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = mfma32(fa[0][t], fb[0][t], acc[0][0]);
        acc[0][1] = mfma32(fa[0][t], fb[1][t], acc[0][1]);
        acc[1][0] = mfma32(fa[1][t], fb[0][t], acc[1][0]);
        acc[1][1] = mfma32(fa[1][t], fb[1][t], acc[1][1]);
      }
    }
    if (kt + 1 < nkt) store_tiles(kt + 1, buf ^ 1);
    __syncthreads();
  }

  if (BM == 128 && BN == 128 && g.vecC) gemm_epilogue_wide<128>(g, acc, smem_f32, 64, m0, n0, wm, wn, l31, half, batch, split, 128);
  else gemm_epilogue(g, acc, m0, n0, wm, wn, l31, half, batch, split);
}

// ---------------------------------------------------------------------------------------------- bf16 x 3
// Same product on the bf16 matrix cores (16x the f32 MFMA rate) without leaving fp32-class accuracy: every fp32 operand
// is split on the fly into hi = bf16(x) and lo = bf16(x - hi) while it is staged into LDS, and each 32x32x16 step
// issues three MFMAs into the same fp32 accumulator: hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-16 relative).
// Block tile 128 x 128, 4 waves x (2 x 2) tiles of 32 x 32, k-tile 32 (two MFMA k-steps), LDS rows of 32 bf16 + 8 pad
// (80 bytes: 16-byte aligned fragment reads, conflict-free by the same 5*row mod 16 argument as the fp32 kernel),
// four images per buffer (A hi, A lo, B hi, B lo), two buffers = 80 KB -> two workgroups per CU.
// Only 16-byte aligned operands take this path (the host falls back to the exact fp32 kernel otherwise).
constexpr int XBK = 32;
constexpr int XROW = XBK + 8;                 // bf16 elements per LDS row
constexpr int XIMG = 128 * XROW;              // bf16 elements per image

__device__ __forceinline__ void split4(const f32x4 v, bf16x4* hi, bf16x4* lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)v[j];
    (*hi)[j] = h;
    (*lo)[j] = (__bf16)(v[j] - (float)h);
  }
}

// fp16 pieces (XT == 4): hi = fp16(x), lo = fp16(x - hi): 11 + 11 mantissa bits, three products are 2^-22 relative - fp32-grade at
// the three-product cost - for operands inside fp16's range (activations and weights of the sampler's per-frame products; a
// value beyond 65504 becomes inf - inf = NaN; gradients, which underflow, never take this path). The LDS images are typed
// __bf16: the same 2-byte slots carry the fp16 bit patterns.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split4h(const f32x4 v, bf16x4* hi, bf16x4* lo) {
  h16x4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = (_Float16)v[j];
    l[j] = (_Float16)(v[j] - (float)h[j]);
  }
  *hi = __builtin_bit_cast(bf16x4, h);
  *lo = __builtin_bit_cast(bf16x4, l);
}

// x = p0 + p1 + p2 with three bf16 pieces (8 + 8 + 8 mantissa bits: all 24 of an fp32 but the last rounding): the six-product
// form of the kernel below (XT == 3) sums p2 b0 + p0 b2 + p1 b1 + p1 b0 + p0 b1 + p0 b0 - what is dropped is 2^-24 relative, the
// fp32 rounding itself - at 6/16 of the f32-input MFMA's cost
__device__ __forceinline__ void split4x3(const f32x4 v, bf16x4* p0, bf16x4* p1, bf16x4* p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)v[j];
    const float r1 = v[j] - (float)h;
    const __bf16 m = (__bf16)r1;
    (*p0)[j] = h;
    (*p1)[j] = m;
    (*p2)[j] = (__bf16)(r1 - (float)m);
  }
}

template <bool KC>
struct XStager {
  // KC: four float4 along k per thread; !KC: one 4(k) x 4(mn) patch per thread (four float4 along mn)
  int off[4];
  int kk[4];      // k of the float4 inside the tile
  int lds[4];     // KC: bf16 offset of the 4 values in the image; !KC: unused
  int mn_l, kg4;  // !KC: first mn of the patch, first k
  bool in[4];
  __device__ __forceinline__ void init(int tid, int mn0, int MN, long ld) {
    if (KC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        const int k = (f & 7) * 4, mn = f >> 3;
        kk[i] = k;
        in[i] = mn0 + mn < MN;
        off[i] = in[i] ? mn * (int)ld + k : 0;
        lds[i] = mn * XROW + k;
      }
    } else {
      kg4 = (tid & 7) * 4;
      mn_l = (tid >> 3) * 4;
      const bool ok = mn0 + mn_l < MN;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kk[j] = kg4 + j;
        in[j] = ok;
        off[j] = ok ? (kg4 + j) * (int)ld + mn_l : 0;
        lds[j] = 0;
      }
    }
  }
  __device__ __forceinline__ void load(const float* __restrict__ p, int krem, f32x4 (&r)[4]) const {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (in[i] && kk[i] < krem) ? *reinterpret_cast<const f32x4*>(p + off[i]) : z;
  }
  // the same patch of an operand that ARRIVES rounded to bf16 (lfi_gemm_desc.a_bf16; !KC only): 8-byte loads, widened exactly - the
  // split below then yields hi = the value, lo = 0
  __device__ __forceinline__ void load16(const __bf16* __restrict__ p, int krem, f32x4 (&r)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 b = {0u, 0u};
      if (in[i] && kk[i] < krem) b = *reinterpret_cast<const uint2*>(p + off[i]);
      r[i] = f32x4{__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                   __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
    }
  }
  template <bool H16 = false>
  __device__ __forceinline__ void store(__bf16* hi_img, __bf16* lo_img, int krem, f32x4 (&r)[4]) const {
    if (KC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 1; j < 4; ++j) r[i][j] = (kk[i] + j < krem) ? r[i][j] : 0.0f;
        bf16x4 h, l;
        if (H16) split4h(r[i], &h, &l);
        else split4(r[i], &h, &l);
        *reinterpret_cast<bf16x4*>(hi_img + lds[i]) = h;
        *reinterpret_cast<bf16x4*>(lo_img + lds[i]) = l;
      }
    } else {
      // r[j][i] = element (k = kg4 + j, mn = mn_l + i): transpose the 4 x 4 patch so each LDS row gets 4 consecutive k
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 col = {r[0][i], r[1][i], r[2][i], r[3][i]};
        bf16x4 h, l;
        if (H16) split4h(col, &h, &l);
        else split4(col, &h, &l);
        *reinterpret_cast<bf16x4*>(hi_img + (mn_l + i) * XROW + kg4) = h;
        *reinterpret_cast<bf16x4*>(lo_img + (mn_l + i) * XROW + kg4) = l;
      }
    }
  }
  __device__ __forceinline__ void store3(__bf16* img0, __bf16* img1, __bf16* img2, int krem, f32x4 (&r)[4]) const {
    if (KC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 1; j < 4; ++j) r[i][j] = (kk[i] + j < krem) ? r[i][j] : 0.0f;
        bf16x4 a, b, c;
        split4x3(r[i], &a, &b, &c);
        *reinterpret_cast<bf16x4*>(img0 + lds[i]) = a;
        *reinterpret_cast<bf16x4*>(img1 + lds[i]) = b;
        *reinterpret_cast<bf16x4*>(img2 + lds[i]) = c;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 col = {r[0][i], r[1][i], r[2][i], r[3][i]};
        bf16x4 a, b, c;
        split4x3(col, &a, &b, &c);
        *reinterpret_cast<bf16x4*>(img0 + (mn_l + i) * XROW + kg4) = a;
        *reinterpret_cast<bf16x4*>(img1 + (mn_l + i) * XROW + kg4) = b;
        *reinterpret_cast<bf16x4*>(img2 + (mn_l + i) * XROW + kg4) = c;
      }
    }
  }
};

// ABF (round 5): A arrives rounded to bf16 (lfi_gemm_desc.a_bf16; mn-contiguous, two-product mode): half the A bytes of the thin
// HBM-bound weight-gradient products, the same two products on the same values as the fp32 form with skip bit 0
template <bool AKC, bool BKC, int XT = 0, bool ABF = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmArgs g) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * 128, n0 = tn * 128;
  const float* __restrict__ A = g.A + batch * g.strideA;
  const float* __restrict__ B = g.B + batch * g.strideB;
  const int kbeg = split * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + XBK - 1) / XBK;

  XStager<AKC> sa;
  XStager<BKC> sb;
  sa.init(tid, m0, g.M, g.lda);
  sb.init(tid, n0, g.N, g.ldb);
  const float* __restrict__ tA = AKC ? A + (long)m0 * g.lda + kbeg : A + (long)kbeg * g.lda + m0;
  const float* __restrict__ tB = BKC ? B + (long)n0 * g.ldb + kbeg : B + (long)kbeg * g.ldb + n0;
  const long stepA = AKC ? XBK : (long)XBK * g.lda, stepB = BKC ? XBK : (long)XBK * g.ldb;
  const __bf16* __restrict__ tA16 = reinterpret_cast<const __bf16*>(g.A) + batch * g.strideA + (long)kbeg * g.lda + m0;   // (ABF)

  f32x4 ra[4], rb[4];
  auto load_tiles = [&](int kt) {
    const int krem = kend - (kbeg + kt * XBK);
    if constexpr (ABF) sa.load16(tA16 + kt * stepA, krem, ra);
    else sa.load(tA + kt * stepA, krem, ra);
    sb.load(tB + kt * stepB, krem, rb);
  };
  constexpr int NIMG = XT == 3 ? 6 : 4;   // images per buffer: A hi, A lo, B hi, B lo - or three pieces of each (six products)
  // six products: ONE LDS buffer (60 KB; with two, 120 KB, only one workgroup fits a CU and the sampler's small per-frame
  // products - 384 and 512 tiles for 256 CUs - ran in two half-empty rounds) at the price of a second barrier per k-tile
  constexpr int NBUF = XT == 3 ? 1 : 2;
  auto store_tiles = [&](int kt, int buf) {
    const int krem = kend - (kbeg + kt * XBK);
    __bf16* base = xsmem + buf * NIMG * XIMG;
    if constexpr (XT == 3) {
      sa.store3(base, base + XIMG, base + 2 * XIMG, krem, ra);
      sb.store3(base + 3 * XIMG, base + 4 * XIMG, base + 5 * XIMG, krem, rb);
    } else {
      sa.template store<XT == 4>(base, base + XIMG, krem, ra);
      sb.template store<XT == 4>(base + 2 * XIMG, base + 3 * XIMG, krem, rb);
    }
  };
  auto mfma3 = [](const bf16x8& x, const bf16x8& y, const f32x16& c) {   // bf16 pieces, or (XT == 4) fp16 pieces in the same slots
    if constexpr (XT == 4)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, x), __builtin_bit_cast(h16x8, y), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  };

  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (nkt > 0) {
    load_tiles(0);
    store_tiles(0, 0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = NBUF == 2 ? (kt & 1) : 0;
    if (kt + 1 < nkt) load_tiles(kt + 1);
    const __bf16* base = xsmem + buf * NIMG * XIMG;
    if constexpr (XT == 3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a0[2], a1[2], a2[2], b0[2], b1[2], b2[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int ro = (wm * 64 + t2 * 32 + l31) * XROW + ks * 16 + half * 8;
          const int co = (wn * 64 + t2 * 32 + l31) * XROW + ks * 16 + half * 8;
          a0[t2] = *reinterpret_cast<const bf16x8*>(base + ro);
          a1[t2] = *reinterpret_cast<const bf16x8*>(base + XIMG + ro);
          a2[t2] = *reinterpret_cast<const bf16x8*>(base + 2 * XIMG + ro);
          b0[t2] = *reinterpret_cast<const bf16x8*>(base + 3 * XIMG + co);
          b1[t2] = *reinterpret_cast<const bf16x8*>(base + 4 * XIMG + co);
          b2[t2] = *reinterpret_cast<const bf16x8*>(base + 5 * XIMG + co);
        }
        // smallest terms first: (2,0) (0,2) (1,1) | (1,0) (0,1) | (0,0)
#define LFI_X6(AP, BP)                                                                                                     \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                          \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AP[mt], BP[nt], acc[mt][nt], 0, 0, 0)
        LFI_X6(a2, b0);
        LFI_X6(a0, b2);
        LFI_X6(a1, b1);
        LFI_X6(a1, b0);
        LFI_X6(a0, b1);
        LFI_X6(a0, b0);
#undef LFI_X6
      }
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int ro = (wm * 64 + t2 * 32 + l31) * XROW + ks * 16 + half * 8;
        const int co = (wn * 64 + t2 * 32 + l31) * XROW + ks * 16 + half * 8;
        ah[t2] = *reinterpret_cast<const bf16x8*>(base + ro);
        al[t2] = *reinterpret_cast<const bf16x8*>(base + XIMG + ro);
        bh[t2] = *reinterpret_cast<const bf16x8*>(base + 2 * XIMG + co);
        bl[t2] = *reinterpret_cast<const bf16x8*>(base + 3 * XIMG + co);
      }
      // three products per accumulator and k-step, in the order lo*hi, hi*lo, hi*hi (each accumulator sees them in that
      // order whichever are switched on: results with skip = 0 are what they always were)
      if (!LFI_GSKIP(1)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma3(al[mt], bh[nt], acc[mt][nt]);
      }
      if (!LFI_GSKIP(2)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma3(ah[mt], bl[nt], acc[mt][nt]);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma3(ah[mt], bh[nt], acc[mt][nt]);
    }
    if (NBUF == 1) __syncthreads();   // every wave has read tile kt before it is overwritten
    if (kt + 1 < nkt) store_tiles(kt + 1, NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }
  // 80 KB of dynamic LDS: the whole 128 x 132 fp32 tile image (67.6 KB) fits, one pass
  if (g.vecC) gemm_epilogue_wide<128, 256, 2, (XT >= 1)>(g, acc, reinterpret_cast<float*>(xsmem), 128, m0, n0, wm, wn, l31, half, batch, split, 128);
  else gemm_epilogue(g, acc, m0, n0, wm, wn, l31, half, batch, split);
}

// Warp-specialised variant. In the kernel above one wave alternates between its 24 MFMAs (768 cycles) and ~200 VALU
// instructions of operand splitting per k-tile, so the matrix pipe idles during the split (31 % of the bf16 peak with two
// workgroups per CU trading places). Here a workgroup has 8 waves with fixed roles: waves 0-3 only read fragments and
// issue MFMAs (each a 64 x 64 patch of the 128 x 128 tile), waves 4-7 only load, split and write the next k-tile. Wave w
// and w + 4 share a SIMD (a workgroup's waves go to SIMDs 0, 2, 1, 3 in turn), where the VALU and matrix pipes run
// concurrently: the split of tile t + 1 executes under the MFMAs of tile t. One barrier per k-tile as before; loads run
// two tiles ahead (two register sets); full k-tiles take a predicate-free path (rows past M / N are clamped: they only
// feed outputs that are never stored), only the last partial k-tile masks.

// ---------------------------------------------------------------------------------------------- bf16 x 3, 256 x 256 tiles
// The 128 x 128 kernel above is bound by operand delivery, not by the matrix pipe: with only the hi*hi MFMA left (1/3 of
// the matrix work) the cond_transform product ran 0.73 ms instead of 0.93 ms. Each workgroup has one k-tile of loads in
// flight (64 KB per CU) against ~1-2 us of L2/HBM latency, and at full MFMA rate a 128 x 128 tile consumes 43 B/clk/CU of
// fp32 operands. This variant halves the bytes per MFMA (256 x 256 block tile: 21 B/clk/CU) and quadruples the depth:
// 1024 threads (16 waves x 64 x 64 patches), k-tile 16, FOUR k-tiles of loads in flight per thread (128 KB per CU),
// register-staged, split to bf16 hi/lo on the way into a double-buffered LDS image (48-byte rows: 16-byte aligned
// fragment reads, conflict-free since 3 is odd).
constexpr int YBK = 16;
constexpr int YROW = YBK + 8;            // bf16 per LDS row
constexpr int YIMG = 256 * YROW;         // bf16 per image (one operand, one plane)
constexpr int YDEPTH = 2;                // k-tiles of loads in flight

constexpr int YPIT = 256 + 32;           // bf16 per k row of a k-major (mn-contiguous operand) image: 576 B, = 64 mod 256

// One float4 per thread and k-tile for either operand orientation.
//  KC (k contiguous in memory): 4 float4 per row of 16 k; LDS image row-major [mn][YROW], fragments by ds_read_b128.
//  !KC (mn contiguous in memory): 64 float4 per k row; LDS image k-major [k][YPIT] exactly as loaded (8-byte writes, 512
//  contiguous bytes per wave), fragments by the transposing ds_read_b64_tr_b16 (two reads of 4 k each per fragment).
template <bool KC>
struct YStager {
  int off, lds, kk;
  __device__ __forceinline__ void init(int tid, int mn0, int MN, long ld) {
    if (KC) {
      const int k4 = (tid & 3) * 4, mn = tid >> 2;
      kk = k4;
      off = (min(mn0 + mn, MN - 1) - mn0) * (int)ld + k4;
      lds = mn * YROW + k4;
    } else {
      const int k = tid >> 6, mn4 = (tid & 63) * 4;
      kk = k;
      off = k * (int)ld + (mn0 + mn4 < MN ? mn4 : 0);   // a float4 wholly outside the matrix re-reads the tile's first columns
      lds = k * YPIT + mn4;
    }
  }
  __device__ __forceinline__ void load(const float* __restrict__ p, int krem, f32x4& r) const {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (krem >= YBK) {
      r = *reinterpret_cast<const f32x4*>(p + off);
    } else {  // last, partial k-tile
      r = kk < krem ? *reinterpret_cast<const f32x4*>(p + off) : z;
      if (KC) {
#pragma unroll
        for (int j = 1; j < 4; ++j) r[j] = (kk + j < krem) ? r[j] : 0.0f;
      }
    }
  }
  __device__ __forceinline__ void store(__bf16* hi_img, __bf16* lo_img, const f32x4& r) const {
    uint2 h, l;
    split2(r[0], r[1], &h.x, &l.x);
    split2(r[2], r[3], &h.y, &l.y);
    *reinterpret_cast<uint2*>(hi_img + lds) = h;
    *reinterpret_cast<uint2*>(lo_img + lds) = l;
  }
};

typedef __bf16 ybf16x4 __attribute__((ext_vector_type(4)));

// MFMA 32x32x16 operand fragment (lane: row/col l & 31, k = 8 (l >> 5) .. + 7) of the 32-wide tile starting at `mn` of an image
template <bool KC>
__device__ __forceinline__ bf16x8 yfrag(const __bf16* img, int mn, int lane) {
  if (KC) {
    return *reinterpret_cast<const bf16x8*>(img + (mn + (lane & 31)) * YROW + (lane >> 5) * 8);
  } else {
    // 16-lane group g = lane >> 4 covers columns mn + 16 (g & 1) .. + 15 and k rows 8 (g >> 1) .. + 7; lane 4q + p of the
    // group supplies the address of row q, columns 4p .. 4p + 3 and receives column (lane & 15) of the four rows
    const int i = lane & 15, q = i >> 2, pp = i & 3;
    const __bf16* ptr = img + (8 * (lane >> 5) + q) * YPIT + mn + 16 * ((lane >> 4) & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) ybf16x4 lds_v4;
    const ybf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)ptr);
    const ybf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(ptr + 4 * YPIT));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

template <bool AKC, bool BKC, int XT = 0>
__global__ __launch_bounds__(1024, 4) void gemm_bf16x3_256_kernel(GemmArgs g) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * 256, n0 = tn * 256;
  const float* __restrict__ A = g.A + batch * g.strideA;
  const float* __restrict__ B = g.B + batch * g.strideB;
  const int kbeg = split * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + YBK - 1) / YBK;

  YStager<AKC> sa;
  YStager<BKC> sb;
  sa.init(tid, m0, g.M, g.lda);
  sb.init(tid, n0, g.N, g.ldb);
  const float* __restrict__ tA = AKC ? A + (long)m0 * g.lda + kbeg : A + (long)kbeg * g.lda + m0;
  const float* __restrict__ tB = BKC ? B + (long)n0 * g.ldb + kbeg : B + (long)kbeg * g.ldb + n0;
  const long stepA = AKC ? YBK : (long)YBK * g.lda, stepB = BKC ? YBK : (long)YBK * g.ldb;

  f32x4 ra[YDEPTH], rb[YDEPTH];
  auto load = [&](int kt, f32x4& xa, f32x4& xb) {
    const int krem = kend - (kbeg + kt * YBK);
    sa.load(tA + kt * stepA, krem, xa);
    sb.load(tB + kt * stepB, krem, xb);
  };
  auto load_full = [&](int kt, f32x4& xa, f32x4& xb) {   // a k-tile known to be complete: no predicates at all
    xa = *reinterpret_cast<const f32x4*>(tA + kt * stepA + sa.off);
    xb = *reinterpret_cast<const f32x4*>(tB + kt * stepB + sb.off);
  };
  auto store = [&](int buf, const f32x4& xa, const f32x4& xb) {
    __bf16* base = xsmem + buf * 4 * YIMG;
    sa.store(base, base + YIMG, xa);
    sb.store(base + 2 * YIMG, base + 3 * YIMG, xb);
  };

  const int wm = wave >> 2, wn = wave & 3;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto mma = [&](int buf) {
    const __bf16* base = xsmem + buf * 4 * YIMG;
    bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      ah[t2] = yfrag<AKC>(base, wm * 64 + t2 * 32, lane);
      al[t2] = yfrag<AKC>(base + YIMG, wm * 64 + t2 * 32, lane);
      bh[t2] = yfrag<BKC>(base + 2 * YIMG, wn * 64 + t2 * 32, lane);
      bl[t2] = yfrag<BKC>(base + 3 * YIMG, wn * 64 + t2 * 32, lane);
    }
    if (!LFI_GSKIP(1)) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
    }
    if (!LFI_GSKIP(2)) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
  };

  // tile t lives in register set t % YDEPTH; at the top of iteration kt tile kt is already in LDS buffer kt & 1.
  // Long K (the common case): prologue and steady state without a single conditional load, so that the compiler's
  // s_waitcnt insertion can count the loads in flight (a conditional load anywhere on the path into the loop makes it fall
  // back to vmcnt(0) before every store: the whole memory latency exposed once per k-tile, measured 6k cycles per
  // iteration instead of 1.5k). The last YDEPTH + 1 .. 2 YDEPTH tiles and short products take the generic loop.
  int kt0 = 0;
  if (nkt > 2 * YDEPTH) {
#pragma unroll
    for (int t = 0; t < YDEPTH; ++t) load_full(t, ra[t], rb[t]);
    store(0, ra[0], rb[0]);
    __syncthreads();
    for (; kt0 + 2 * YDEPTH < nkt; kt0 += YDEPTH) {   // strict: the last (possibly partial) k-tile is left to the drain loop
#pragma unroll
      for (int u = 0; u < YDEPTH; ++u) {
        const int kt = kt0 + u;
        load_full(kt + YDEPTH, ra[u], rb[u]);   // set u held tile kt: converted one iteration ago
        __builtin_amdgcn_sched_barrier(0);      // keep the loads first: the scheduler otherwise sinks them below the store
        mma(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        store((kt + 1) & 1, ra[(u + 1) % YDEPTH], rb[(u + 1) % YDEPTH]);
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < YDEPTH; ++t)
      if (t < nkt) load(t, ra[t], rb[t]);
    if (nkt > 0) store(0, ra[0], rb[0]);
    __syncthreads();
  }
  // drain / short products: loads only while tiles remain
  for (; kt0 < nkt; kt0 += YDEPTH) {
#pragma unroll
    for (int u = 0; u < YDEPTH; ++u) {
      const int kt = kt0 + u;
      if (kt < nkt) {
        if (kt + YDEPTH < nkt) load(kt + YDEPTH, ra[u], rb[u]);
        mma(kt & 1);
        if (kt + 1 < nkt) store((kt + 1) & 1, ra[(u + 1) % YDEPTH], rb[(u + 1) % YDEPTH]);
        __syncthreads();
      }
    }
  }
  // 96 KB of LDS: 64 rows x 260 floats (66.5 KB) per pass of the wide epilogue
  // Two passes of 128 rows (133 KB of LDS) rather than four of 64: half the barriers and operand round trips, and 8 instead
  // of 4 waves in the register phase. Measured (tools/gemm_probe.py, same box): cond_transform forward 0.804 -> 0.767 ms, the
  // in-place dpre product (K = 384, epilogue-heavy: reads G, writes C) 0.561 -> 0.501 ms. Tried on top and dropped: requesting
  // pass p + 1's operand rows (8 float4 per thread) before pass p's register phase - the kernel then spills (64 - 76 bytes
  // of scratch per lane) and is slower (0.537 vs 0.483 ms on dpre); starting the first round of workgroups in 8 phases so
  // that the rounds' epilogue bursts do not coincide (s_memtime-timed start delays of 0 .. 7 x 3000 - 24000 ticks) - no effect
  // for any spacing.
#ifndef LFI_EPI_ROWS
#define LFI_EPI_ROWS 128
#endif
  if (g.vecC) gemm_epilogue_wide<256, 1024, 2, (XT >= 1)>(g, acc, reinterpret_cast<float*>(xsmem), LFI_EPI_ROWS, m0, n0, wm, wn, l31, half, batch, split, 256);
  else gemm_epilogue_n<256>(g, acc, m0, n0, wm, wn, l31, half, batch, split);
}

// The same 256 x 256 kernel for a product whose A operand ARRIVES rounded to bf16 (lfi_gemm_desc.a_bf16: the window encoders'
// bf16 gradient stash, mn-contiguous as every weight-gradient operand) and whose B operand is fp32, mn-contiguous: two products per
// k-step (a_hi b_lo + a_hi b_hi - exactly what the three-product kernel computes with skip bit 0 on the fp32 form of the same
// values), A's 8-byte loads go straight into the hi image (no split, no lo image), B is split as everywhere. Half the A bytes.
__global__ __launch_bounds__(1024, 4) void gemm_bf16a_256_kernel(GemmArgs g) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * 256, n0 = tn * 256;
  const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.A) + batch * g.strideA;
  const float* __restrict__ B = g.B + batch * g.strideB;
  const int kbeg = split * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + YBK - 1) / YBK;

  YStager<false> sa, sb;     // same thread -> (k row, 4 mn) map for both operands; A's 4 values are 8 bytes
  sa.init(tid, m0, g.M, g.lda);
  sb.init(tid, n0, g.N, g.ldb);
  const __bf16* __restrict__ tA = A + (long)kbeg * g.lda + m0;
  const float* __restrict__ tB = B + (long)kbeg * g.ldb + n0;
  const long stepA = (long)YBK * g.lda, stepB = (long)YBK * g.ldb;

  uint2 ra[YDEPTH];
  f32x4 rb[YDEPTH];
  auto load = [&](int kt, uint2& xa, f32x4& xb) {
    const int krem = kend - (kbeg + kt * YBK);
    xa = sa.kk < krem ? *reinterpret_cast<const uint2*>(tA + kt * stepA + sa.off) : uint2{0u, 0u};
    sb.load(tB + kt * stepB, krem, xb);
  };
  auto load_full = [&](int kt, uint2& xa, f32x4& xb) {
    xa = *reinterpret_cast<const uint2*>(tA + kt * stepA + sa.off);
    xb = *reinterpret_cast<const f32x4*>(tB + kt * stepB + sb.off);
  };
  auto store = [&](int buf, const uint2& xa, const f32x4& xb) {
    __bf16* base = xsmem + buf * 4 * YIMG;
    *reinterpret_cast<uint2*>(base + sa.lds) = xa;
    sb.store(base + 2 * YIMG, base + 3 * YIMG, xb);
  };

  const int wm = wave >> 2, wn = wave & 3;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto mma = [&](int buf) {
    const __bf16* base = xsmem + buf * 4 * YIMG;
    bf16x8 ah[2], bh[2], bl[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      ah[t2] = yfrag<false>(base, wm * 64 + t2 * 32, lane);
      bh[t2] = yfrag<false>(base + 2 * YIMG, wn * 64 + t2 * 32, lane);
      bl[t2] = yfrag<false>(base + 3 * YIMG, wn * 64 + t2 * 32, lane);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
  };

  int kt0 = 0;
  if (nkt > 2 * YDEPTH) {
#pragma unroll
    for (int t = 0; t < YDEPTH; ++t) load_full(t, ra[t], rb[t]);
    store(0, ra[0], rb[0]);
    __syncthreads();
    for (; kt0 + 2 * YDEPTH < nkt; kt0 += YDEPTH) {
#pragma unroll
      for (int u = 0; u < YDEPTH; ++u) {
        const int kt = kt0 + u;
        load_full(kt + YDEPTH, ra[u], rb[u]);
        __builtin_amdgcn_sched_barrier(0);
        mma(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        store((kt + 1) & 1, ra[(u + 1) % YDEPTH], rb[(u + 1) % YDEPTH]);
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < YDEPTH; ++t)
      if (t < nkt) load(t, ra[t], rb[t]);
    if (nkt > 0) store(0, ra[0], rb[0]);
    __syncthreads();
  }
  for (; kt0 < nkt; kt0 += YDEPTH) {
#pragma unroll
    for (int u = 0; u < YDEPTH; ++u) {
      const int kt = kt0 + u;
      if (kt < nkt) {
        if (kt + YDEPTH < nkt) load(kt + YDEPTH, ra[u], rb[u]);
        mma(kt & 1);
        if (kt + 1 < nkt) store((kt + 1) & 1, ra[(u + 1) % YDEPTH], rb[(u + 1) % YDEPTH]);
        __syncthreads();
      }
    }
  }
  if (g.vecC) gemm_epilogue_wide<256, 1024, 2, false>(g, acc, reinterpret_cast<float*>(xsmem), LFI_EPI_ROWS, m0, n0, wm, wn, l31, half, batch, split, 256);
  else gemm_epilogue_n<256>(g, acc, m0, n0, wm, wn, l31, half, batch, split);
}

template <int BM, int BN, int WM, int WN, bool VEC>
void launch_gemm_v(const GemmArgs& a, int akc, int bkc, dim3 grid, hipStream_t st) {
  if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, VEC>), grid, dim3(256), 0, st, a);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false, VEC>), grid, dim3(256), 0, st, a);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true, VEC>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false, VEC>), grid, dim3(256), 0, st, a);
}
template <int BM, int BN, int WM, int WN>
void launch_gemm(const GemmArgs& a, int akc, int bkc, dim3 grid, hipStream_t st) {
  if (a.vecA && a.vecB) launch_gemm_v<BM, BN, WM, WN, true>(a, akc, bkc, grid, st);
  else launch_gemm_v<BM, BN, WM, WN, false>(a, akc, bkc, grid, st);
}

int kchunk_for(int K, int splitk) {
  int c = lfi_cdiv(K, splitk);
  return lfi_cdiv(c, BKT) * BKT;
}

}  // namespace

// Tile shape and K split of a product. splitk_req > 0 is taken as given; 0 lets the library fill the chip: the score is
// (share of the last round of co-resident workgroups that is used) x (relative rate of the tile shape) / (padding waste),
// less 3 % per extra split for the reduce pass. shape: 0 = 128 x 128 (two workgroups per CU), 3 = 256 x 256 (bf16x3 only,
// one per CU, ~12 % faster per MFMA on long products).
struct GemmPlan { int shape, splitk; };
GemmPlan gemm_plan(int M, int N, int K, int batch, int splitk_req, bool x3) {
  static int allow256 = -1;
  if (allow256 < 0) {
    const char* e = getenv("LFI_GEMM_256");
    allow256 = (e && e[0] == '0') ? 0 : 1;
  }
  GemmPlan best = {0, splitk_req > 0 ? splitk_req : 1};
  double best_score = -1.0;
  const int cand[6] = {1, 2, 3, 4, 6, 8};
  for (int pass = 0; pass < 2; ++pass) {
    const int shape = pass ? 3 : 0;
    if (shape == 3 && !(x3 && allow256)) continue;
    const int tile = shape == 3 ? 256 : 128;
    const double slots = shape == 3 ? 256.0 : 512.0, rate = shape == 3 ? 1.12 : 1.0;
    const double tiles = (double)lfi_cdiv(M, tile) * lfi_cdiv(N, tile) * batch;
    const double waste = tiles * tile * tile / ((double)M * N * batch);
    for (int ci = 0; ci < 6; ++ci) {
      const int sk = splitk_req > 0 ? splitk_req : cand[ci];
      if (splitk_req <= 0 && sk > 1 && K / sk < 512) break;
      const double wg = tiles * sk;
      const double rounds = (double)(long)((wg + slots - 1) / slots);
      const double score = wg / (rounds * slots) * rate / waste * (1.0 - 0.03 * (sk - 1) * (splitk_req > 0 ? 0.0 : 1.0));
      if (score > best_score + 1e-9) { best_score = score; best.shape = shape; best.splitk = sk; }
      if (splitk_req > 0) break;
    }
  }
  return best;
}

extern "C" long lfi_gemm_work_floats(const lfi_gemm_desc* d) {
  if (!d) return 0;
  if (d->splitk == 0) return 8L * d->batch * d->M * d->N;   // automatic split: room for the largest one
  if (d->splitk <= 1) return 0;
  return (long)d->batch * d->splitk * d->M * d->N;
}


namespace {
// instantiations of the two default bf16x3 kernels by operand orientation and extras (XT, see LFI_GSKIP)
template <int XT>
int launch_x3_256(const GemmArgs& a, int akc, int bkc, dim3 grid, size_t lds, hipStream_t st) {
  // XT = 1 (epilogue column sums, no skip switch) is instantiated for the one orientation that uses it - A k-contiguous, B
  // n-contiguous: the in-place dpre product - and hands every other orientation to XT = 2 (compile time: each instantiation
  // of this kernel costs ~8 s of hipcc)
  if (XT == 1 && !(akc && !bkc)) return launch_x3_256<2>(a, akc, bkc, grid, lds, st);
  static bool attr = false;
  if (!attr) {
    hipError_t e1 = hipSuccess, e2, e3 = hipSuccess, e4 = hipSuccess;
    e2 = hipFuncSetAttribute((const void*)gemm_bf16x3_256_kernel<true, false, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (XT != 1) {
      e1 = hipFuncSetAttribute((const void*)gemm_bf16x3_256_kernel<true, true, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      e3 = hipFuncSetAttribute((const void*)gemm_bf16x3_256_kernel<false, true, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      e4 = hipFuncSetAttribute((const void*)gemm_bf16x3_256_kernel<false, false, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      lfi_set_error("lfi_gemm_f32: cannot reserve %zu bytes of LDS for the 256 x 256 bf16x3 kernel", lds);
      return LFI_ERR_LAUNCH;
    }
    attr = true;
  }
  if constexpr (XT == 1) {
    hipLaunchKernelGGL((gemm_bf16x3_256_kernel<true, false, XT>), grid, dim3(1024), lds, st, a);
  } else {
    if (akc && bkc) hipLaunchKernelGGL((gemm_bf16x3_256_kernel<true, true, XT>), grid, dim3(1024), lds, st, a);
    else if (akc) hipLaunchKernelGGL((gemm_bf16x3_256_kernel<true, false, XT>), grid, dim3(1024), lds, st, a);
    else if (bkc) hipLaunchKernelGGL((gemm_bf16x3_256_kernel<false, true, XT>), grid, dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((gemm_bf16x3_256_kernel<false, false, XT>), grid, dim3(1024), lds, st, a);
  }
  return LFI_OK;
}
template <int XT>
int launch_x3_128(const GemmArgs& a, int akc, int bkc, dim3 grid, size_t lds, hipStream_t st) {
  if (XT == 1 && !(akc && !bkc)) return launch_x3_128<2>(a, akc, bkc, grid, lds, st);   // as launch_x3_256
  static bool attr = false;
  if (!attr) {
    hipError_t e1 = hipSuccess, e2, e3 = hipSuccess, e4 = hipSuccess;
    e2 = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<true, false, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if constexpr (XT != 1) {
      e1 = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<true, true, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      e3 = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<false, true, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      e4 = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<false, false, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      lfi_set_error("lfi_gemm_f32: cannot reserve %zu bytes of LDS for the bf16x3 kernel", lds);
      return LFI_ERR_LAUNCH;
    }
    attr = true;
  }
  if constexpr (XT == 1) {
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false, XT>), grid, dim3(256), lds, st, a);
  } else {
    if (akc && bkc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, XT>), grid, dim3(256), lds, st, a);
    else if (akc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false, XT>), grid, dim3(256), lds, st, a);
    else if (bkc) hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, XT>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, XT>), grid, dim3(256), lds, st, a);
  }
  return LFI_OK;
}

}  // namespace

// Rows of the partial column-sum matrix lfi_gemm_f32 fills when lfi_gemm_desc.colsum_part is set, or 0 when this product does
// not take a path that can (it needs the bf16x3 kernels' wide epilogue, no K split, column-batched or unbatched C).
extern "C" long lfi_gemm_colpart_rows(const lfi_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || !(d->precision & 1) || d->a_bf16) return 0;
  if (d->splitk != 1 || (d->batch > 1 && !(d->strideC > 0 && d->strideC * d->batch <= d->ldc))) return 0;
  auto vec_ok = [](const float* p, long ld, long stride, int kcontig, int mn, int K) {
    if ((reinterpret_cast<uintptr_t>(p) & 15) || (ld & 3) || (stride & 3)) return 0;
    const long need = kcontig ? K : mn;
    return ld >= (need + 3) / 4 * 4 ? 1 : 0;
  };
  if (!vec_ok(d->A, d->lda, d->strideA, d->a_kcontig, d->M, d->K) || !vec_ok(d->B, d->ldb, d->strideB, d->b_kcontig, d->N, d->K)) return 0;
  const bool c_ok = (reinterpret_cast<uintptr_t>(d->C) & 15) == 0 && d->ldc % 4 == 0 && d->strideC % 4 == 0;
  const bool g_ok = d->act != 2 || ((reinterpret_cast<uintptr_t>(d->G) & 15) == 0 && d->ldg % 4 == 0 && d->strideG % 4 == 0);
  if (!(c_ok && g_ok && !(d->act == 2 && d->accumulate != 0))) return 0;
  GemmPlan plan = gemm_plan(d->M, d->N, d->K, d->batch, 1, true);
  if (d->precision & 0x30) plan.shape = (d->precision & 0x10) ? 3 : 0;
  if (d->precision & 12) plan.shape = 0;
  return plan.shape == 3 ? (long)lfi_cdiv(d->M, 256) * (256 / LFI_EPI_ROWS) : (long)lfi_cdiv(d->M, 128);
}

extern "C" int lfi_gemm_f32(const lfi_gemm_desc* d, void* stream) {
  LFI_REQUIRE(d, "lfi_gemm_f32: null descriptor");
  LFI_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0 && d->batch >= 1, "lfi_gemm_f32: bad dims M=%d N=%d K=%d batch=%d",
              d->M, d->N, d->K, d->batch);
  if (d->M == 0 || d->N == 0) return LFI_OK;
  LFI_REQUIRE(d->A && d->B && d->C, "lfi_gemm_f32: null operand");
  LFI_REQUIRE(d->act >= 0 && d->act <= 2, "lfi_gemm_f32: bad act %d", d->act);
  LFI_REQUIRE(d->act != 2 || d->G, "lfi_gemm_f32: act 2 needs G");
  LFI_REQUIRE(d->batch <= 65535, "lfi_gemm_f32: batch %d too large", d->batch);
  auto vec_ok = [](const float* p, long ld, long stride, int kcontig, int mn, int K) {
    if ((reinterpret_cast<uintptr_t>(p) & 15) || (ld & 3) || (stride & 3)) return 0;
    const long need = kcontig ? K : mn;
    return ld >= (need + 3) / 4 * 4 ? 1 : 0;
  };
  const int vecA = vec_ok(d->A, d->lda, d->strideA, d->a_kcontig, d->M, d->K);
  const int vecB = vec_ok(d->B, d->ldb, d->strideB, d->b_kcontig, d->N, d->K);
  const bool use_x3 = (d->precision & 1) && vecA && vecB;
  if (d->a_bf16) {
    LFI_REQUIRE((d->precision & 1) && (d->precision & 0x100) && !(d->precision & 0x200) && !d->a_kcontig && !d->b_kcontig && vecB &&
                (reinterpret_cast<uintptr_t>(d->A) & 7) == 0 && (d->lda & 3) == 0 && (d->strideA & 3) == 0 && !d->colsum_part,
                "lfi_gemm_f32: a_bf16 needs bf16x3 mode with skip bit 0 (two products), both operands mn-contiguous, 8-byte aligned "
                "A rows and a 16-byte aligned B");
  }
  GemmPlan plan = gemm_plan(d->M, d->N, d->K, d->batch, (d->splitk == 0 && !d->work) ? 1 : d->splitk, use_x3 || d->a_bf16);
  if (use_x3 && (d->precision & 0x30)) plan.shape = (d->precision & 0x10) ? 3 : 0;  // tests pin the tile shape (lfi.h)
  const bool x6 = use_x3 && (d->precision & 4) && !d->a_bf16;   // six products: fp32-grade, 128 x 128 kernel only
  const bool x3h = use_x3 && !x6 && (d->precision & 8) && !d->a_bf16;   // three products of fp16 pieces, likewise
  if (x6 || x3h) plan.shape = 0;
  // (a_bf16: the 256 x 256 kernel - the window encoders' long-K dW_hh - unless the plan wants 128 x 128 tiles for a BATCHED product:
  // the flow's thin weight-gradient products, round 5)
  const bool a16_128 = d->a_bf16 && d->batch > 1 && plan.shape == 0;
  if (d->a_bf16 && !a16_128) plan.shape = 3;
  int splitk = plan.splitk < 1 ? 1 : plan.splitk;
  if (splitk > d->K / BKT) splitk = d->K / BKT < 1 ? 1 : d->K / BKT;
  LFI_REQUIRE(splitk == 1 || d->work, "lfi_gemm_f32: splitk needs a workspace");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.A = d->A; a.lda = d->lda; a.B = d->B; a.ldb = d->ldb; a.C = d->C; a.ldc = d->ldc;
  a.bias = d->bias; a.G = d->G; a.ldg = d->ldg;
  a.strideA = d->strideA; a.strideB = d->strideB; a.strideC = d->strideC;
  a.strideBias = d->strideBias; a.strideG = d->strideG;
  a.accumulate = d->accumulate; a.act = d->act; a.slope = d->slope;
  a.splitk = splitk; a.kchunk = splitk > 1 ? kchunk_for(d->K, splitk) : d->K;
  if (a.kchunk & 31) a.kchunk = (a.kchunk + 31) & ~31;  // also a whole number of bf16x3 k-tiles
  // the chunking may leave trailing splits empty: they still write zeros, which keeps the reduce simple
  a.work = d->work;
  a.vecA = vecA;
  a.vecB = vecB;
  a.skip = (d->precision >> 8) & 3;
  a.gm = 0; a.Ap = nullptr; a.Bp = nullptr; a.nkt = a.nktA = a.nktB = 0; a.pstrideA = a.pstrideB = 0;
  a.colpart = d->colsum_part; a.ldpart = d->ld_part;
  {
    // C (or the split-K workspace) and G rows 16-byte aligned; act 2 together with accumulate stays on the narrow path
    const bool partial = splitk > 1;
    const bool c_ok = partial ? (((long)d->M * d->N) % 4 == 0 && d->N % 4 == 0 && (reinterpret_cast<uintptr_t>(d->work) & 15) == 0)
                              : ((reinterpret_cast<uintptr_t>(d->C) & 15) == 0 && d->ldc % 4 == 0 && d->strideC % 4 == 0);
    const bool g_ok = d->act != 2 || ((reinterpret_cast<uintptr_t>(d->G) & 15) == 0 && d->ldg % 4 == 0 && d->strideG % 4 == 0);
    a.vecC = (c_ok && g_ok && !(d->act == 2 && d->accumulate != 0)) ? 1 : 0;
  }
  // tile shape: narrow outputs get the tall tile, short outputs the wide one; bf16x3: 128 x 128 or 256 x 256 by the plan
  int shape = (use_x3 || d->a_bf16) ? plan.shape : 0;
  if (!use_x3 && !d->a_bf16) {
    if (d->N <= 64 && d->M > 128) shape = 1;       // 256 x 64
    else if (d->M <= 64 && d->N > 128) shape = 2;  // 64 x 256
  }
  const int bm = shape == 0 ? 128 : (shape == 1 ? 256 : (shape == 3 ? 256 : 64)),
            bn = shape == 0 ? 128 : (shape == 1 ? 64 : 256);
  a.tiles_m = lfi_cdiv(d->M, bm);
  a.tiles_n = lfi_cdiv(d->N, bn);
  // (16-byte loads need: base pointer 16-byte aligned, leading dimension and batch stride multiples of 4 floats, every k
  // chunk starting on a multiple of 4, and a row stride that covers the last float4 of a row: a load may run up to 3
  // floats past K (k-contiguous; zeroed before it reaches LDS) or past M/N (mn-contiguous; those LDS columns only feed
  // output rows/columns that are never stored) — see vec_ok above.)
  dim3 grid(a.tiles_m * a.tiles_n, d->batch, splitk);
  if (a16_128) {
    const size_t lds = (size_t)2 * 4 * XIMG * sizeof(__bf16);
    static bool attrb = false;
    if (!attrb) {
      if (hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<false, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        lfi_set_error("lfi_gemm_f32: cannot reserve %zu bytes of LDS for the bf16-A 128 x 128 kernel", lds);
        return LFI_ERR_LAUNCH;
      }
      attrb = true;
    }
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, 2, true>), grid, dim3(256), lds, st, a);
  } else if (d->a_bf16) {
    const size_t lds_loop = (size_t)2 * 4 * YIMG * sizeof(__bf16), lds_epi = (size_t)LFI_EPI_ROWS * 260 * sizeof(float);
    const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static bool attra = false;
    if (!attra) {
      if (hipFuncSetAttribute((const void*)gemm_bf16a_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        lfi_set_error("lfi_gemm_f32: cannot reserve %zu bytes of LDS for the bf16-A kernel", lds);
        return LFI_ERR_LAUNCH;
      }
      attra = true;
    }
    hipLaunchKernelGGL(gemm_bf16a_256_kernel, grid, dim3(1024), lds, st, a);
  } else if (use_x3 && shape == 3) {
    // main loop: two buffers of four 256 x 16 bf16 planes (96 KB); the wide epilogue: LFI_EPI_ROWS rows x 260 floats per pass
    const size_t lds_loop = (size_t)2 * 4 * YIMG * sizeof(__bf16), lds_epi = (size_t)LFI_EPI_ROWS * 260 * sizeof(float);
    const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    const int rcl = a.skip ? launch_x3_256<2>(a, d->a_kcontig, d->b_kcontig, grid, lds, st)
                           : (a.colpart ? launch_x3_256<1>(a, d->a_kcontig, d->b_kcontig, grid, lds, st)
                                        : launch_x3_256<0>(a, d->a_kcontig, d->b_kcontig, grid, lds, st));   // (<1>: see launch_x3_256)
    if (rcl) return rcl;
  } else if (use_x3) {
    const size_t lds = x6 ? (size_t)128 * 132 * sizeof(float)   // one buffer of six images (60 KB) < the epilogue's tile image
                          : (size_t)2 * 4 * XIMG * sizeof(__bf16);
    {
      const int rcl = x6 ? launch_x3_128<3>(a, d->a_kcontig, d->b_kcontig, grid, lds, st)
                      : x3h ? launch_x3_128<4>(a, d->a_kcontig, d->b_kcontig, grid, lds, st)
                      : a.skip ? launch_x3_128<2>(a, d->a_kcontig, d->b_kcontig, grid, lds, st)
                             : (a.colpart ? launch_x3_128<1>(a, d->a_kcontig, d->b_kcontig, grid, lds, st)
                                          : launch_x3_128<0>(a, d->a_kcontig, d->b_kcontig, grid, lds, st));
      if (rcl) return rcl;
    }
  } else if (shape == 0) launch_gemm<128, 128, 2, 2>(a, d->a_kcontig, d->b_kcontig, grid, st);
  else if (shape == 1) launch_gemm<256, 64, 4, 1>(a, d->a_kcontig, d->b_kcontig, grid, st);
  else launch_gemm<64, 256, 1, 4>(a, d->a_kcontig, d->b_kcontig, grid, st);
  LFI_LAUNCH_CHECK("lfi_gemm_f32");
  if (splitk > 1) {
    const long mn = (long)d->M * d->N;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool red4 = d->N % 4 == 0 && al16(d->work) && al16(d->C) && d->ldc % 4 == 0 && d->strideC % 4 == 0 &&
                      (!d->G || (al16(d->G) && d->ldg % 4 == 0 && d->strideG % 4 == 0));
    if (red4) {
      dim3 rgrid((unsigned)min((long)lfi_cdiv(mn / 4, 256), 2048L), d->batch);
      hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, rgrid, dim3(256), 0, st, a);
    } else {
      dim3 rgrid((unsigned)min((long)lfi_cdiv(mn, 256), 2048L), d->batch);
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, rgrid, dim3(256), 0, st, a);
    }
    LFI_LAUNCH_CHECK("lfi_gemm_f32 split-k reduce");
  }
  return LFI_OK;
}


