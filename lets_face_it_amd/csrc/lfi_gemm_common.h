// Pieces shared by the GEMM translation units (lfi_gemm.hip: fp32-operand kernels; lfi_pgemm.hip: kernels on pre-split bf16
// hi / lo operand planes): the launch descriptor, the epilogues, the XCD-aware tile order, the operand split, the split-K
// reduce kernels. Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "lfi_common.h"

namespace {

#ifndef LFI_GEMM_BKT
#define LFI_GEMM_BKT 16
#endif
#ifndef LFI_GEMM_LDSPIPE
#define LFI_GEMM_LDSPIPE 1
#endif
constexpr int BKT = LFI_GEMM_BKT;  // k-tile
// XT (template parameter of the bf16x3 kernels): 0 = lean; 1 = + per-tile column sums in the wide epilogue (colpart);
// 2 = + runtime skipping of the a_lo b_hi / a_hi b_lo products (measurement builds of the precision sweep). Compiled in
// unconditionally, the skip branches alone cost the 256 x 256 kernels 16 - 148 bytes of scratch per lane at their 128-VGPR cap
// (the sampler's F x 8192 x 640 product ran 4x slower): they are their own instantiations.
#define LFI_GSKIP(bit) (XT >= 2 && (g.skip & (bit)))

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
  int vecA, vecB;  // operand rows are 16-byte aligned and 4-float granular: 16-byte global loads are legal
  int vecC;        // the same for C (and G): the epilogue goes through LDS with 16-byte row-wise loads / stores
  // gemm_planes_256_kernel only: operands as pre-split bf16 hi / lo planes in MFMA fragment order (lfi_planes_from_f32)
  const __bf16* Ap; const __bf16* Bp;   // block ((rt * nkt_op + kt) * 2 + plane) * 512 bf16, rt = 32-row tile, kt = 16-k tile
  int nkt, nktA, nktB;                  // k-tiles of this product; k-tiles per row tile in either plane buffer
  long pstrideA, pstrideB;              // bf16 elements between batch entries
  float* colpart; long ldpart;   // wide epilogue only: per (row tile, pass) column sums of the stored result (lfi_gemm_desc)
  int gm;     // tile rows per group of the XCD-contiguous tile walk (gemm_tile_of_block); 0 = 8
  int skip;   // bf16x3 kernels: bit 0 drops the a_lo * b_hi product, bit 1 the a_hi * b_lo product (lfi_gemm_desc.precision
              // bits 8 / 9; tools/precision_sweep.py). 0 = all three products.
  // lfi_pgemm.hip only (everything below is zero elsewhere): the result tile also / instead leaves the kernel as bf16 hi / lo
  // planes for the products that consume it, and the act == 2 operand may arrive as planes (only its sign is used)
  int storeC;                               // 1: fp32 rows to C as everywhere else; 0: plane outputs only
  __bf16* Cr; int nktCr; long colCr;        // planes of the result: rows = C rows, columns = colCr + batch * strideC + C column
  const __bf16* Gr; int nktGr; long colGr;  // act == 2: row planes of G (hi plane read) instead of g.G
  int hiOnly;                               // plane outputs: the hi planes only (their consumer takes them as a rounded A operand)
};

__device__ __forceinline__ float apply_act(float v, int act, float slope, const float* G, long gidx) {
  if (act == 1) return v > 0.0f ? v : v * slope;
  if (act == 2) return G[gidx] > 0.0f ? v : v * slope;
  return v;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) __bf16 xsmem[];
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// (hi, lo) bf16 pairs of (a, b): 6 VALU instructions (cvt_pk, shift, and, 2 sub, cvt_pk)
__device__ __forceinline__ void split2(float a, float b, unsigned* hi, unsigned* lo) {
  const bf16x2 h = __builtin_convertvector((float2_t){a, b}, bf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float ha = __builtin_bit_cast(float, hb << 16), hbv = __builtin_bit_cast(float, hb & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector((float2_t){a - ha, b - hbv}, bf16x2);
  *hi = hb;
  *lo = __builtin_bit_cast(unsigned, l);
}


// Accumulator tile -> C (or the split-K workspace): bias, activation, accumulate. acc[mt][nt] register r holds
// (row (r&3) + 8*(r>>2) + 4*half, col l31) of the 32 x 32 tile (mt, nt) of this wave's 64 x 64 patch.
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                              int l31, int half, int batch, int split) {
  const bool partial = g.splitk > 1;
  float* __restrict__ Cb = partial ? g.work + ((long)batch * g.splitk + split) * (long)g.M * g.N : g.C + batch * g.strideC;
  const long ldc = partial ? g.N : g.ldc;
  const float* bias = g.bias ? g.bias + batch * g.strideBias : nullptr;
  const float* G = g.G ? g.G + batch * g.strideG : nullptr;
  const bool need_c = !partial && g.accumulate != 0, need_g = !partial && g.act == 2;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wn * 64 + nt * 32 + l31;
      if (col >= g.N) continue;
      const float bv = (!partial && bias) ? bias[col] : 0.0f;
      // operands of the epilogue first, all 16 in flight (C may alias G - the in-place dpre product - so the compiler
      // cannot hoist these loads over the stores below by itself: one exposed HBM round trip per element otherwise)
      float cold[16], gold[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, g.M - 1);
        cold[r] = need_c ? Cb[(long)row * ldc + col] : 0.0f;
        gold[r] = need_g ? G[(long)row * g.ldg + col] : 1.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row >= g.M) continue;
        float v = acc[mt][nt][r];
        if (!partial) {
          v += bv;
          if (g.accumulate == 2) v += cold[r];
          if (g.act == 1) v = v > 0.0f ? v : v * g.slope;
          else if (g.act == 2) v = gold[r] > 0.0f ? v : v * g.slope;
          if (g.accumulate == 1) v += cold[r];
        }
        Cb[(long)row * ldc + col] = v;
      }
    }
}

template <int BN>
__device__ __forceinline__ void gemm_epilogue_n(const GemmArgs& g, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                                int l31, int half, int batch, int split) {
  gemm_epilogue(g, acc, m0, n0, wm, wn, l31, half, batch, split);
}

// Wide epilogue. The accumulator layout gives a lane 32-bit elements two rows apart: written straight from registers, a
// 128 x 128 tile costs 64 four-byte store instructions per lane, and an epilogue operand (G of the in-place dpre product, C
// when accumulating) as many four-byte loads - measured: the G loads alone doubled the dpre product's time. Instead the
// tile goes through LDS (free after the main loop): operand tile in with 16-byte row-wise loads, each lane combines its own
// elements in place, result tile out with 16-byte row-wise stores (512 contiguous bytes per 32 lanes).
// `rows` rows of the block tile per pass (the LDS image is rows x (BN + 4) floats); waves whose 64-row patch is in the pass
// take part in the register phase, all 256 threads in the row-wise phases.
// ---- plane output of the wide epilogue (lfi_pgemm.hip). The result tile lies in LDS as fp32 ([rows][BN + 4]); a lane reads 8
// consecutive floats of one row (two conflict-free ds_read_b128), splits them into bf16 hi + lo and stores 16 + 16 bytes so
// that every wave-instruction writes one whole 1-KB block of the destination planes: block (32-row tile, 16-column tile), lane l
// = row l & 31, chunk l >> 5 at lfi_u_plane_offset - the layout of lfi_planes_from_f32, which a consumer reads either way
// (columns or rows as its contraction index). Rows >= M and columns >= N are written as zeros: as a contraction index they
// would otherwise enter the consumer's sums.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 pg_u4(const uint4& v) { return (u32x4){v.x, v.y, v.z, v.w}; }
template <int BN, int NTH>
__device__ __forceinline__ void gemm_emit_planes(const GemmArgs& g, const float* lds, int rows_per_pass, int row0, int n0, int batch) {
  constexpr int WLD = BN + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NTH / 64;
  const long gcol0 = (long)batch * g.strideC + n0;
  if (!g.Cr) return;
  constexpr int KT = BN / 16;
  const int nb = (rows_per_pass / 32) * KT;
  const int uoff = lfi_u_plane_offset(lane & 31, lane >> 5);
  for (int b = wave; b < nb; b += NW) {
    const int rtl = b / KT, ktl = b - rtl * KT;
    if (row0 + rtl * 32 >= g.M || n0 + ktl * 16 >= g.N) continue;
    const int r = rtl * 32 + (lane & 31), c = ktl * 16 + (lane >> 5) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(lds + r * WLD + c), bq = *reinterpret_cast<const f32x4*>(lds + r * WLD + c + 4);
    const bool rok = row0 + r < g.M;
    float v[8] = {a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (rok && n0 + c + e < g.N) ? v[e] : 0.0f;
    uint4 h, l;
    split2(v[0], v[1], &h.x, &l.x); split2(v[2], v[3], &h.y, &l.y);
    split2(v[4], v[5], &h.z, &l.z); split2(v[6], v[7], &h.w, &l.w);
    const long rt = (row0 + rtl * 32) >> 5, kt = (g.colCr + gcol0 + ktl * 16) >> 4;
    char* dst = reinterpret_cast<char*>(g.Cr) + ((rt * g.nktCr + kt) * 2) * 1024 + uoff;
    __builtin_nontemporal_store(pg_u4(h), reinterpret_cast<u32x4*>(dst));
    if (!g.hiOnly) __builtin_nontemporal_store(pg_u4(l), reinterpret_cast<u32x4*>(dst + 1024));
  }
}
// act == 2 operand from planes: the hi plane's bf16 values (only their sign is used) into the LDS operand tile
template <int BN, int NTH>
__device__ __forceinline__ void gemm_sign_tile_from_planes(const GemmArgs& g, float* lds, int rows_per_pass, int row0, int n0, int batch) {
  constexpr int WLD = BN + 4, KT = BN / 16, NW = NTH / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long gcol0 = (long)batch * g.strideC + n0;
  const int nb = (rows_per_pass / 32) * KT;
  for (int b = wave; b < nb; b += NW) {
    const int rtl = b / KT, ktl = b - rtl * KT;
    const int r = rtl * 32 + (lane & 31), c = ktl * 16 + (lane >> 5) * 8;
    uint4 h = {0u, 0u, 0u, 0u};
    if (row0 + rtl * 32 < g.M && n0 + ktl * 16 < g.N) {
      const long rt = (row0 + rtl * 32) >> 5, kt = (g.colGr + gcol0 + ktl * 16) >> 4;
      h = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g.Gr) + ((rt * g.nktGr + kt) * 2) * 1024 +
                                          lfi_u_plane_offset(lane & 31, lane >> 5));
    }
    f32x4 a, bq;
    a[0] = __builtin_bit_cast(float, h.x << 16); a[1] = __builtin_bit_cast(float, h.x & 0xffff0000u);
    a[2] = __builtin_bit_cast(float, h.y << 16); a[3] = __builtin_bit_cast(float, h.y & 0xffff0000u);
    bq[0] = __builtin_bit_cast(float, h.z << 16); bq[1] = __builtin_bit_cast(float, h.z & 0xffff0000u);
    bq[2] = __builtin_bit_cast(float, h.w << 16); bq[3] = __builtin_bit_cast(float, h.w & 0xffff0000u);
    *reinterpret_cast<f32x4*>(lds + r * WLD + c) = a;
    *reinterpret_cast<f32x4*>(lds + r * WLD + c + 4) = bq;
  }
}

// L16: the accumulators are the 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (f32x4 acc[4][4]; tile (mi, ni) register r = row
// 16 mi + 4 (lane >> 4) + r, column 16 ni + (lane & 15) of the wave's 64 x 64 patch) instead of 2 x 2 tiles of the 32 x 32 shape.
template <int BN, int NTH = 256, int MT = 2, bool COLP = false, bool PL = false, bool L16 = false, typename ACC>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmArgs& g, const ACC& acc, float* lds, int rows_per_pass,
                                                   int m0, int n0, int wm, int wn, int l31, int half, int batch, int split, int bm,
                                                   bool has_acc = true) {
  constexpr int WLD = BN + 4;       // LDS row pitch (floats)
  constexpr int F4 = BN / 4;        // float4 per tile row
  constexpr int SWEEP = NTH / F4;   // tile rows per row-wise sweep
  const int tid = threadIdx.x;
  const bool partial = g.splitk > 1;
  float* __restrict__ Cb = partial ? g.work + ((long)batch * g.splitk + split) * (long)g.M * g.N : g.C + batch * g.strideC;
  const long ldc = partial ? g.N : g.ldc;
  const float* bias = g.bias ? g.bias + batch * g.strideBias : nullptr;
  const float* G = g.G ? g.G + batch * g.strideG : nullptr;
  const bool need_c = !partial && g.accumulate != 0, need_g = !partial && g.act == 2;
  const int rrow = tid / F4, c4 = (tid % F4) * 4;  // row-wise phases: SWEEP rows x F4 float4 per sweep
  const int col_g = n0 + c4;
  for (int p0 = 0; p0 < bm; p0 += rows_per_pass) {
    if (PL && need_g && g.Gr) {
      gemm_sign_tile_from_planes<BN, NTH>(g, lds, rows_per_pass, m0 + p0, n0, batch);
      __syncthreads();
    } else if (need_c || need_g) {  // operand tile in (never both: the host keeps act 2 + accumulate on the narrow path)
      const float* src = need_g ? G : Cb;
      const long lds_src = need_g ? g.ldg : ldc;
      for (int r = rrow; r < rows_per_pass; r += SWEEP) {
        const int row = m0 + p0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < g.M && col_g < g.N) {
          if (col_g + 3 < g.N) v = *reinterpret_cast<const f32x4*>(src + (long)row * lds_src + col_g);
          else
            for (int j = 0; j < 4; ++j)
              if (col_g + j < g.N) v[j] = src[(long)row * lds_src + col_g + j];
        }
        *reinterpret_cast<f32x4*>(lds + r * WLD + c4) = v;
      }
      __syncthreads();
    }
    if constexpr (L16) {
      static_assert(!L16 || MT == 2, "16 x 16 accumulator tiles: 64-row patches");
      const int lane = threadIdx.x & 63, c15 = lane & 15, r4 = (lane >> 4) * 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        if (has_acc && wm * 64 + mi * 16 >= p0 && wm * 64 + mi * 16 < p0 + rows_per_pass)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int cl = wn * 64 + ni * 16 + c15;
          const int col = n0 + cl;
          const float bv = (!partial && bias && col < g.N) ? bias[col] : 0.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = wm * 64 - p0 + mi * 16 + r4 + r;
            float v = acc[mi][ni][r];
            if (!partial) {
              v += bv;
              const float o = (need_c || need_g) ? lds[rl * WLD + cl] : 0.0f;
              if (g.accumulate == 2) v += o;
              if (g.act == 1) v = v > 0.0f ? v : v * g.slope;
              else if (g.act == 2) v = o > 0.0f ? v : v * g.slope;
              if (g.accumulate == 1) v += o;
            }
            lds[rl * WLD + cl] = v;
          }
        }
    } else
    if (has_acc) {   // a wave's patch is MT x 32 rows from wm * MT * 32: the row tiles that lie in this pass take part
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (wm * MT * 32 + mt * 32 >= p0 && wm * MT * 32 + mt * 32 < p0 + rows_per_pass)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int cl = wn * 64 + nt * 32 + l31;
          const int col = n0 + cl;
          const float bv = (!partial && bias && col < g.N) ? bias[col] : 0.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wm * MT * 32 - p0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[mt][nt][r];
            if (!partial) {
              v += bv;
              const float o = (need_c || need_g) ? lds[rl * WLD + cl] : 0.0f;
              if (g.accumulate == 2) v += o;
              if (g.act == 1) v = v > 0.0f ? v : v * g.slope;
              else if (g.act == 2) v = o > 0.0f ? v : v * g.slope;
              if (g.accumulate == 1) v += o;
            }
            lds[rl * WLD + cl] = v;
          }
        }
    }
    __syncthreads();
    if (PL && !partial) gemm_emit_planes<BN, NTH>(g, lds, rows_per_pass, m0 + p0, n0, batch);
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};   // (dead code unless COLP)
    for (int r = rrow; r < rows_per_pass && (!PL || partial || g.storeC || COLP); r += SWEEP) {  // result tile out
      const int row = m0 + p0 + r;
      if (row < g.M && col_g < g.N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + r * WLD + c4);
        if (COLP) csum += v;
        if (PL && !partial && !g.storeC) continue;   // plane outputs only (the column sums above still see the tile)
        // written once, read by a later kernel: non-temporal, so the result tile does not push the operand panels the other
        // workgroups are re-reading out of L2 (measured -1.5 .. -2 % on the three cond_transform products)
        // (split-K partial sums are read back at once by the reduce kernel: those stay cacheable)
        if (col_g + 3 < g.N) {
#ifndef LFI_EPI_NT_MODE
#define LFI_EPI_NT_MODE 2   // 0 never, 1 always, 2 final results only
#endif
          if (LFI_EPI_NT_MODE == 0 || (LFI_EPI_NT_MODE == 2 && partial)) *reinterpret_cast<f32x4*>(Cb + (long)row * ldc + col_g) = v;
          else __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Cb + (long)row * ldc + col_g));
        }
        else
          for (int j = 0; j < 4; ++j)
            if (col_g + j < g.N) Cb[(long)row * ldc + col_g + j] = v[j];
      }
    }
    if (COLP && g.colpart && !partial) {
      // column sums of this pass's rows (bias gradients without a second pass over C): SWEEP partial rows through LDS, added
      // in a fixed order; row (tile row * passes + pass) of the partial matrix, columns as in C (batch entries side by side)
      __syncthreads();
      *reinterpret_cast<f32x4*>(lds + rrow * WLD + c4) = csum;
      __syncthreads();
      if (tid < BN && n0 + tid < g.N) {
        float v = 0.0f;
        for (int i = 0; i < SWEEP; ++i) v += lds[i * WLD + tid];
        const int npass = (bm + rows_per_pass - 1) / rows_per_pass;
        g.colpart[((long)(m0 / bm) * npass + p0 / rows_per_pass) * g.ldpart + batch * g.strideC + n0 + tid] = v;
      }
      __syncthreads();
    } else if (p0 + rows_per_pass < bm) __syncthreads();
  }
}

// Epilogue WITHOUT the LDS round trip, for the 16 x 16 x 32 kernels when they compute the product transposed (weights / B fragments
// as the MFMA's A operand): accumulator (i, j) register r is then C[64 wm + 16 i + (lane & 15)][64 wn + 16 j + 4 (lane >> 4) + r] - four
// CONSECUTIVE columns of one row per lane. MODE 3: operand planes
// only, 8 bytes per lane and plane (half a 16-byte chunk of the block format; a wave-instruction writes a contiguous 512-byte half
// block). Bias and LeakyReLU (act 1) in registers; NOT the act-2 mask, accumulate, per-pass column sums or planes + rows together
// (those keep the LDS form). All addressing is a buffer descriptor + a scalar offset + one per-lane VGPR, stores are predicated.
// fp32 ROW outputs were tried the same way (16-byte stores of a lane's four columns, 16 rows per wave-instruction: 64-byte runs)
// and dropped: gic 0.284 -> 0.357 ms, the split-K partials -1 .. -2 %; planes: cond_transform forward 0.505 -> 0.459 ms.
// (the through-LDS epilogue of a 128 x 256 tile measured about half as long as the tile's whole main loop at K = 896)
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gu32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__device__ __forceinline__ void gemm_epilogue_direct16(const GemmArgs& g, const f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lane,
                                                       int batch, int split) {
  const int l15 = lane & 15, g4 = lane >> 4;
  const int r0 = m0 + wm * 64, c0 = n0 + wn * 64;             // this wave's patch (wave-uniform)
  const float* bias = g.bias ? g.bias + batch * g.strideBias : nullptr;
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = c0 + 16 * j + 4 * g4;
    bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias && col < g.N) bv[j] = *reinterpret_cast<const f32x4*>(bias + col);
  }
  static_assert(MODE == 3 || MODE == 4, "plane outputs only");
  // MODE 4 (the in-place dpre product): + the act-2 mask read from the hi plane of Gr at the element's own place (8 bytes per lane:
  // only the signs are used), + per-wave column sums of the stored values (the wave's 64 rows are one "pass" of the 128-row tile:
  // partial row 2 (m0 / 128) + wm of colpart, the layout the through-LDS epilogue fills; sums over rows in another order)
  f32x4 csum[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) csum[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const long gcol0 = g.colCr + (long)batch * g.strideC + c0;      // multiple of 16 (colCr, strideC, c0 are)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 16 * i + l15;
      const bool blk = (r0 + 16 * (i & ~1)) < g.M;                    // the 32-row block exists (rows past M inside it: zeros)
      const long rt = (r0 + 16 * i) >> 5;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(g.Cr) + (rt * g.nktCr + (gcol0 >> 4)) * 2048, 0,
                                                                         0x7fffffff, 0x00020000);
      const unsigned voff = (unsigned)(lfi_u_plane_offset(16 * (i & 1) + l15, g4 >> 1) + (g4 & 1) * 8);
      __amdgpu_buffer_rsrc_t rg = rs;
      if constexpr (MODE == 4) {
        const long ggcol0 = g.colGr + (long)batch * g.strideC + c0;
        rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.Gr)) + (rt * g.nktGr + (ggcol0 >> 4)) * 2048, 0,
                                               0x7fffffff, 0x00020000);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = c0 + 16 * j + 4 * g4;
        f32x4 v = acc[i][j] + bv[j];
        if constexpr (MODE == 4) {
          gu32x2 m2 = {0u, 0u};
          if (blk && c0 + 16 * j < g.N) m2 = __builtin_amdgcn_raw_buffer_load_b64(rg, voff, 2048 * j, 0);
          const float o0 = __builtin_bit_cast(float, m2[0] << 16), o1 = __builtin_bit_cast(float, m2[0] & 0xffff0000u);
          const float o2 = __builtin_bit_cast(float, m2[1] << 16), o3 = __builtin_bit_cast(float, m2[1] & 0xffff0000u);
          v[0] = o0 > 0.0f ? v[0] : v[0] * g.slope; v[1] = o1 > 0.0f ? v[1] : v[1] * g.slope;
          v[2] = o2 > 0.0f ? v[2] : v[2] * g.slope; v[3] = o3 > 0.0f ? v[3] : v[3] * g.slope;
        } else if (g.act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : v[r] * g.slope;
        }
        if (row >= g.M || col >= g.N) v = (f32x4){0.f, 0.f, 0.f, 0.f};   // inside an existing block: zeros (k of a later product)
        if constexpr (MODE == 4) csum[j] += v;
        uint2 h, l;
        split2(v[0], v[1], &h.x, &l.x);
        split2(v[2], v[3], &h.y, &l.y);
        if (blk && c0 + 16 * j < g.N) {
          __builtin_amdgcn_raw_buffer_store_b64((gu32x2){h.x, h.y}, rs, voff, 2048 * j, 2);
          if (!g.hiOnly) __builtin_amdgcn_raw_buffer_store_b64((gu32x2){l.x, l.y}, rs, voff, 2048 * j + 1024, 2);
        }
      }
    }
  }
  if constexpr (MODE == 4) {
    if (g.colpart) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = csum[j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], o, 64);
        }
        const int col = c0 + 16 * j + 4 * g4;
        if (l15 == 0 && col < g.N)
          *reinterpret_cast<f32x4*>(g.colpart + ((long)(m0 >> 7) * 2 + wm) * g.ldpart + (long)batch * g.strideC + col) = v;
      }
    }
  }
}

// XCD-aware, grouped work order shared by both GEMM kernels: workgroups b, b+8, ... share an XCD (round-robin dispatch over
// the linearised grid). Each XCD gets a contiguous run of (split, batch, tile) work items (bijective for any grid size),
// so that all tiles of one K-split / one batch entry - which re-read the same operand panels - meet in ONE 4 MB L2 instead
// of fetching them from HBM once per XCD (the long-K weight-gradient products have 12 tiles per split: spread over the
// XCDs their B panel was fetched up to 6 times). Inside a batch entry tiles are walked in groups of GM tile-rows column by
// column so the ~100 tiles an XCD has in flight form a compact GM x 12 patch.
__device__ __forceinline__ void gemm_tile_of_block(const GemmArgs& g, int* tm, int* tn, int* batch, int* split) {
  const int ntile = g.tiles_m * g.tiles_n;
  const long total = (long)gridDim.x * gridDim.y * gridDim.z;
  long lin = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
  {
    const long q = total >> 3, idx = lin >> 3;
    const int r = (int)(total & 7), xcd = (int)(lin & 7);
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bid = (int)(lin % ntile);
  const long rest = lin / ntile;
  *batch = (int)(rest % gridDim.y);
  *split = (int)(rest / gridDim.y);
  const int GM = g.gm > 0 ? g.gm : 8;
  const int per_group = GM * g.tiles_n;
  const int grp = bid / per_group, in_grp = bid - grp * per_group;
  const int rows_here = min(GM, g.tiles_m - grp * GM);
  *tm = grp * GM + in_grp % rows_here;
  *tn = in_grp / rows_here;
}

#ifndef LFI_EPI_ROWS
#define LFI_EPI_ROWS 128
#endif

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
    int s = 0;
    for (; s + 8 <= g.splitk; s += 8) {   // eight partials in flight, added in split order (the long-K products leave up to 85)
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = w[(long)(s + u) * mn];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < g.splitk; ++s) v += w[(long)s * mn];
    if (bias) v += bias[col];
    if (g.accumulate == 2) v += C[(long)row * g.ldc + col];
    v = apply_act(v, g.act, g.slope, G, (long)row * g.ldg + col);
    if (g.accumulate == 1) v += C[(long)row * g.ldc + col];
    C[(long)row * g.ldc + col] = v;
  }
}

// Four columns per thread (N, ldc, ldg multiples of 4; work, C, G 16-byte aligned): same sums in the same order as above.
__global__ __launch_bounds__(256) void gemm_splitk_reduce4_kernel(GemmArgs g) {
  const long mn = (long)g.M * g.N, mn4 = mn >> 2;
  const int n4 = g.N >> 2;
  const int batch = blockIdx.y;
  const float* bias = g.bias ? g.bias + batch * g.strideBias : nullptr;
  const float* G = g.G ? g.G + batch * g.strideG : nullptr;
  float* C = g.C + batch * g.strideC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < mn4; i += (long)gridDim.x * 256) {
    const int row = (int)(i / n4), col = (int)(i - (long)row * n4) * 4;
    const float* w = g.work + (long)batch * g.splitk * mn + 4 * i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= g.splitk; s += 8) {   // (as above)
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + (long)(s + u) * mn));
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < g.splitk; ++s) v += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + (long)s * mn));
    f32x4* cp = reinterpret_cast<f32x4*>(C + (long)row * g.ldc + col);
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    if (g.accumulate) c0 = *cp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j];
      if (bias) x += bias[col + j];
      if (g.accumulate == 2) x += c0[j];
      x = apply_act(x, g.act, g.slope, G, (long)row * g.ldg + col + j);
      if (g.accumulate == 1) x += c0[j];
      v[j] = x;
    }
    *cp = v;
  }
}

}  // namespace
