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
#include <stdarg.h>
#include <stdio.h>

#include "lfi_common.h"

namespace {

#ifndef LFI_GEMM_BKT
#define LFI_GEMM_BKT 16
#endif
#ifndef LFI_GEMM_LDSPIPE
#define LFI_GEMM_LDSPIPE 1
#endif
constexpr int BKT = LFI_GEMM_BKT;  // k-tile
constexpr int LPAD = 4;

struct GemmArgs {
  int M, N, K;
  const float* A; long lda;
  const float* B; long ldb;
  float* C; long ldc;
  const float* bias;
  const float* G; long ldg;
  long strideA, strideB, strideC, strideBias, strideG;
  int accumulate, act;
  float slope;
  int splitk, kchunk;
  float* work;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float apply_act(float v, int act, float slope, const float* G, long gidx) {
  if (act == 1) return v > 0.0f ? v : v * slope;
  if (act == 2) return G[gidx] > 0.0f ? v : v * slope;
  return v;
}

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  static_assert(WM * WN == 4 && BM == WM * 64 && BN == WN * 64, "wave tile is 64 x 64");
  __shared__ float As[2][BKT][BM + LPAD];
  __shared__ float Bs[2][BKT][BN + LPAD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (round-robin dispatch); give each XCD a contiguous run of
  // tiles (bijective also when the tile count is not a multiple of 8).
  const int ntile = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // ... and walk the tiles in groups of GM tile-rows, column by column inside a group, so the ~100 tiles an XCD has in
  // flight form a compact GM x 12 patch: every k-slice of an A panel is shared by ~12 tiles and every B slice by GM
  // (row-major order shared A 64 ways but B only 1.5 ways, and B panels streamed from beyond L2).
  constexpr int GM = 8;
  const int per_group = GM * g.tiles_n;
  const int grp = bid / per_group, in_grp = bid - grp * per_group;
  const int rows_here = min(GM, g.tiles_m - grp * GM);
  const int tm = grp * GM + in_grp % rows_here, tn = in_grp / rows_here;
  const int m0 = tm * BM, n0 = tn * BN;
  const int batch = blockIdx.y, split = blockIdx.z;
  const float* __restrict__ A = g.A + batch * g.strideA;
  const float* __restrict__ B = g.B + batch * g.strideB;
  const int kbeg = split * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + BKT - 1) / BKT;

  constexpr int EA = BM * BKT / 256, EB = BN * BKT / 256;
  float ra[EA], rb[EB];

  // Per-thread element pattern, fixed for the whole k loop: element i of the staging pass is (mn_i, k_i) of the tile.
  //   k-contiguous operand: k = tid % 16 (same for every i), mn_i = tid / 16 + 16 i
  //   mn-contiguous operand: mn = tid % BM (same for every i), k_i = tid / BM + (256 / BM) i
  // Addresses are a wave-uniform base (advanced once per k-tile) plus a 32-bit per-lane offset computed once, and the
  // row/column bounds test is hoisted into a bit mask, so the k loop carries no address arithmetic in the VALU.
  int offA[EA], offB[EB];
  unsigned okA = 0, okB = 0;
  int kA[EA], kB[EB];  // k index inside the tile (constant per i)
#pragma unroll
  for (int i = 0; i < EA; ++i) {
    const int idx = tid + 256 * i;
    int m, k;
    if (AKC) { k = idx % BKT; m = idx / BKT; } else { m = idx % BM; k = idx / BM; }
    kA[i] = k;
    const bool ok = m0 + m < g.M;
    okA |= (ok ? 1u : 0u) << i;
    offA[i] = ok ? (AKC ? m * (int)g.lda + k : k * (int)g.lda + m) : 0;
  }
#pragma unroll
  for (int i = 0; i < EB; ++i) {
    const int idx = tid + 256 * i;
    int n, k;
    if (BKC) { k = idx % BKT; n = idx / BKT; } else { n = idx % BN; k = idx / BN; }
    kB[i] = k;
    const bool ok = n0 + n < g.N;
    okB |= (ok ? 1u : 0u) << i;
    offB[i] = ok ? (BKC ? n * (int)g.ldb + k : k * (int)g.ldb + n) : 0;
  }
  const float* __restrict__ tA = AKC ? A + (long)m0 * g.lda + kbeg : A + (long)kbeg * g.lda + m0;  // tile origin, uniform
  const float* __restrict__ tB = BKC ? B + (long)n0 * g.ldb + kbeg : B + (long)kbeg * g.ldb + n0;
  const long stepA = AKC ? BKT : (long)BKT * g.lda, stepB = BKC ? BKT : (long)BKT * g.ldb;

  auto load_tiles = [&](int kt) {
    const int krem = kend - (kbeg + kt * BKT);  // valid k in this tile (>= BKT except for the last one)
    const float* __restrict__ pa = tA + kt * stepA;
    const float* __restrict__ pb = tB + kt * stepB;
#pragma unroll
    for (int i = 0; i < EA; ++i) ra[i] = (((okA >> i) & 1u) && kA[i] < krem) ? pa[offA[i]] : 0.0f;
#pragma unroll
    for (int i = 0; i < EB; ++i) rb[i] = (((okB >> i) & 1u) && kB[i] < krem) ? pb[offB[i]] : 0.0f;
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int idx = tid + 256 * i;
      int m, k;
      if (AKC) { k = idx % BKT; m = idx / BKT; } else { m = idx % BM; k = idx / BM; }
      As[buf][k][m] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int idx = tid + 256 * i;
      int n, k;
      if (BKC) { k = idx % BKT; n = idx / BKT; } else { n = idx % BN; k = idx / BN; }
      Bs[buf][k][n] = rb[i];
    }
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
    store_tiles(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tiles(kt + 1);
#if LFI_GEMM_LDSPIPE
    // fragments of k-step s+1 are read from LDS before the MFMAs of k-step s issue: the waves a SIMD holds run in
    // lock step (same code, launched together), so an LDS round trip that is exposed in one wave is exposed in all
    float fa[2][2], fb[2][2];
    fa[0][0] = As[buf][half][wm * 64 + l31];
    fa[0][1] = As[buf][half][wm * 64 + 32 + l31];
    fb[0][0] = Bs[buf][half][wn * 64 + l31];
    fb[0][1] = Bs[buf][half][wn * 64 + 32 + l31];
#pragma unroll
    for (int s2 = 0; s2 < BKT / 2; ++s2) {
      const int c = s2 & 1, n = c ^ 1;
      if (s2 + 1 < BKT / 2) {
        const int kk = 2 * (s2 + 1) + half;
        fa[n][0] = As[buf][kk][wm * 64 + l31];
        fa[n][1] = As[buf][kk][wm * 64 + 32 + l31];
        fb[n][0] = Bs[buf][kk][wn * 64 + l31];
        fb[n][1] = Bs[buf][kk][wn * 64 + 32 + l31];
      }
      acc[0][0] = mfma32(fa[c][0], fb[c][0], acc[0][0]);
      acc[0][1] = mfma32(fa[c][0], fb[c][1], acc[0][1]);
      acc[1][0] = mfma32(fa[c][1], fb[c][0], acc[1][0]);
      acc[1][1] = mfma32(fa[c][1], fb[c][1], acc[1][1]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the next step's two ds_read2 ...
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // ... ahead of this step's four MFMAs
    }
#else
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 2) {
      const float a0 = As[buf][kk + half][wm * 64 + l31];
      const float a1 = As[buf][kk + half][wm * 64 + 32 + l31];
      const float b0 = Bs[buf][kk + half][wn * 64 + l31];
      const float b1 = Bs[buf][kk + half][wn * 64 + 32 + l31];
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
#endif
    if (kt + 1 < nkt) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue
  const bool partial = g.splitk > 1;
  float* __restrict__ Cb = partial ? g.work + ((long)batch * g.splitk + split) * (long)g.M * g.N : g.C + batch * g.strideC;
  const long ldc = partial ? g.N : g.ldc;
  const float* bias = g.bias ? g.bias + batch * g.strideBias : nullptr;
  const float* G = g.G ? g.G + batch * g.strideG : nullptr;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wn * 64 + nt * 32 + l31;
      if (col >= g.N) continue;
      const float bv = (!partial && bias) ? bias[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row >= g.M) continue;
        float v = acc[mt][nt][r];
        if (!partial) {
          v += bv;
          if (g.accumulate == 2) v += Cb[(long)row * ldc + col];
          v = apply_act(v, g.act, g.slope, G, (long)row * g.ldg + col);
          if (g.accumulate == 1) v += Cb[(long)row * ldc + col];
        }
        Cb[(long)row * ldc + col] = v;
      }
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g) {
  const long mn = (long)g.M * g.N;
  const int batch = blockIdx.y;
  const float* bias = g.bias ? g.bias + batch * g.strideBias : nullptr;
  const float* G = g.G ? g.G + batch * g.strideG : nullptr;
  float* C = g.C + batch * g.strideC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < mn; i += (long)gridDim.x * 256) {
    const int row = (int)(i / g.N), col = (int)(i % g.N);
    const float* w = g.work + (long)batch * g.splitk * mn + i;
    float v = 0.0f;
    for (int s = 0; s < g.splitk; ++s) v += w[(long)s * mn];
    if (bias) v += bias[col];
    if (g.accumulate == 2) v += C[(long)row * g.ldc + col];
    v = apply_act(v, g.act, g.slope, G, (long)row * g.ldg + col);
    if (g.accumulate == 1) v += C[(long)row * g.ldc + col];
    C[(long)row * g.ldc + col] = v;
  }
}

template <int BM, int BN, int WM, int WN>
void launch_gemm(const GemmArgs& a, int akc, int bkc, dim3 grid, hipStream_t st) {
  if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, a);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, a);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, a);
}

int kchunk_for(int K, int splitk) {
  int c = lfi_cdiv(K, splitk);
  return lfi_cdiv(c, BKT) * BKT;
}

}  // namespace

extern "C" long lfi_gemm_work_floats(const lfi_gemm_desc* d) {
  if (!d || d->splitk <= 1) return 0;
  return (long)d->batch * d->splitk * d->M * d->N;
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
  int splitk = d->splitk < 1 ? 1 : d->splitk;
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
  // the chunking may leave trailing splits empty: they still write zeros, which keeps the reduce simple
  a.work = d->work;
  // tile shape: narrow outputs get the tall tile, short outputs the wide one
  int shape = 0;  // 128 x 128
  if (d->N <= 64 && d->M > 128) shape = 1;       // 256 x 64
  else if (d->M <= 64 && d->N > 128) shape = 2;  // 64 x 256
  const int bm = shape == 0 ? 128 : (shape == 1 ? 256 : 64), bn = shape == 0 ? 128 : (shape == 1 ? 64 : 256);
  a.tiles_m = lfi_cdiv(d->M, bm);
  a.tiles_n = lfi_cdiv(d->N, bn);
  dim3 grid(a.tiles_m * a.tiles_n, d->batch, splitk);
  if (shape == 0) launch_gemm<128, 128, 2, 2>(a, d->a_kcontig, d->b_kcontig, grid, st);
  else if (shape == 1) launch_gemm<256, 64, 4, 1>(a, d->a_kcontig, d->b_kcontig, grid, st);
  else launch_gemm<64, 256, 1, 4>(a, d->a_kcontig, d->b_kcontig, grid, st);
  LFI_LAUNCH_CHECK("lfi_gemm_f32");
  if (splitk > 1) {
    const long mn = (long)d->M * d->N;
    dim3 rgrid((unsigned)min((long)lfi_cdiv(mn, 256), 2048L), d->batch);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, rgrid, dim3(256), 0, st, a);
    LFI_LAUNCH_CHECK("lfi_gemm_f32 split-k reduce");
  }
  return LFI_OK;
}
