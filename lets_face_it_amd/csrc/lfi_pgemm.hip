// bf16 x 3 GEMM on PRE-SPLIT operand planes (include/lfi.h: lfi_planes_from_f32 / lfi_planes_t_from_f32 / lfi_gemm_planes), gfx950.
//
// lfi_gemm.hip's bf16x3 kernels redo the fp32 -> bf16 hi / lo split of every operand element in EVERY workgroup that touches
// it, through VGPRs and ds_write, and a k-tile's phases (global load, convert + LDS store, fragment reads, MFMA) do not overlap.
// Here the operands arrive ALREADY split - once, by a streaming kernel or by the epilogue of the product that made them - and
// in 1-KB blocks that go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR, no VALU, no ds_write).
//
// ONE block format (lfi_planes_from_f32; lfi_common.h, lfi_u_plane_offset), the planes of an fp32 matrix X (rows x cols): per
// 32-row tile rt and 16-column tile ct two 1-KB blocks (hi, lo), index ((rt * nct + ct) * 2 + plane) * 512 bf16, each a row-major
// [32 rows][16 columns] image (32-byte rows) whose two 16-byte chunks trade places in rows 8-15 and 24-31. An operand takes them
// in one of two USES:
//   use 0, "row use": X's rows are the operand's free (mn) index, its columns the contraction index k (an nn.Linear input or
//     weight in its forward product). A k-tile of an mn tile is one block, moved to LDS as it lies; the v_mfma_f32_32x32x16_bf16
//     fragment (lane l: mn = l & 31, k = 8 (l >> 5) .. + 7) is ONE ds_read_b128 at lfi_u_plane_offset(l & 31, l >> 5) - the
//     chunk swap makes its four 16-lane groups cover all 64 banks.
//   use 1, "transposed use": X's ROWS are the contraction index (every weight gradient sums over frames, the rows of every
//     stored activation; the feature gradient sums over cond_transform's output units, the rows of its weight). An operand tile of
//     32 mn x 16 k is rows 16 j .. 16 j + 15 (j = k-tile & 1) of the two blocks (rt = k-tile >> 1, ct = 2 mn-tile, + 1): two
//     contiguous 512-byte runs, which the LDS-DMA's per-lane source address places as two sub-images, the second with its rows
//     0-3 <-> 4-7 swapped; the fragment is TWO ds_read_b64_tr_b16 (the hardware transposing read: 4 rows x 16 columns per 16
//     lanes) at byte offsets toff and toff ^ 128, and the swap puts the two 16-lane groups of a 32-lane half into opposite halves of
//     the 256-byte bank row: conflict-free (cdna_hip_programming.md T10).
// So a matrix that is consumed both ways - c, the gradient of cond_transform's pre-activation, dgi - is written ONCE.
//
// Kernel: 128 x 256 tile (or 256 x 128: template WMT x WNT), TWO workgroups per CU: 512 threads = 8 waves x (64 x 64) patches (wm = wave >> 2, wn = wave & 3),
// ring of 3 slots of one 16-deep k-tile each, [A: 4 mn tiles x {hi, lo} x 1 KB][B: 8 x 2 x 1 KB] = 24 KB (72 KB; the wide
// epilogue's 64-row passes of 66.5 KB fit inside), three DMA pieces per wave and k-tile.
//   phase t:  DMA tile t + 3 -> slot t % 3 (its fragments were read in phase t - 1)
//             MFMAs of tile t from registers, fragments of tile t + 1 read under them
//             s_waitcnt vmcnt(3) (this wave's pieces of tile t + 2 landed; t + 3 in flight), lgkmcnt(0), s_barrier
// Tiles past the end are fetched again from the last real tile (never used): the counted waits then need no tail cases.
// Split K (grid.z): every split walks its own range of k-tiles and leaves a partial tile in the workspace, reduced (with bias /
// activation) by gemm_splitk_reduce4_kernel - the long-K weight-gradient products have too few output tiles to fill the chip.
// Epilogue (gemm_epilogue_wide<.., PL = true>): besides / instead of fp32 rows the result tile can leave as planes of either
// format, and the act == 2 operand (cond_transform's LeakyReLU mask) can be read from the hi plane of its row planes.
// History (rounds 1-2: 256 x 256 one-per-CU and four-wave 128 x 64-patch variants, ingredient-removal builds, stamps): DESIGN.md.
// Round 4: the kernel below (v_mfma_f32_32x32x16_bf16, one k-tile per phase) is the GENERAL member of the family. The step's six
// products run on two siblings further down that take k-tiles in pairs on v_mfma_f32_16x16x32_bf16 - gemm_planes16_kernel (three
// products, either operand by rows or transposed: cond_transform forward, gic; the backward products when they take three) and
// gemm_planes16t_kernel (two products, B transposed: dW_c, dpre, the cond_transform weight gradient, the feature gradient) -
// whenever K has whole pairs of k-tiles in every split; the two whose
// result leaves as planes only write them straight from the accumulators (gemm_epilogue_direct16). LFI_PGEMM_16 / _16T / _DIRECT = 0
// bring this kernel and the through-LDS epilogue back.
#include "lfi_gemm_common.h"
#include <type_traits>

namespace {

constexpr int QRING = 3;
constexpr int QSLOT = 24 * 1024;

typedef __attribute__((address_space(3))) void plds_void;
typedef __attribute__((address_space(1))) const void pglb_void;
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pbf16x4 plds_v4;

template <bool TF>
__device__ __forceinline__ bf16x8 pg_frag(const char* blk, int uoff, int toff) {
  if (!TF) return *reinterpret_cast<const bf16x8*>(blk + uoff);
  const pbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((plds_v4*)(blk + toff));
  const pbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((plds_v4*)(blk + (toff ^ 128)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

#ifdef LFI_PG_EXP_16
// timing-only experiment (garbage results): the same operand registers through two v_mfma_f32_16x16x32_bf16 (same FLOP, same
// issue cycles) - does the chip hold a higher clock on that shape (MI355X_MICROARCH.md, DVFS give-back item 7)?
__device__ __forceinline__ f32x16 pg_mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]};
  q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q0, 0, 0, 0);
  q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q1, 0, 0, 0);
  c[0] = q0[0]; c[1] = q0[1]; c[2] = q0[2]; c[3] = q0[3];
  c[4] = q1[0]; c[5] = q1[1]; c[6] = q1[2]; c[7] = q1[3];
  return c;
}
#define PG_MFMA(a, b, c) pg_mfma16(a, b, c)
#else
#define PG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// WMT x WNT waves of 64 x 64 patches (WMT * WNT = 8): 2 x 4 = the 128 x 256 tile, 4 x 2 = a 256 x 128 tile for products whose N
// fills 128-wide tiles better than 256-wide ones (gic: N = 3 H = 384; the cond_transform weight gradient: N = 896). Same slot
// size (2 (WMT + WNT) mn tiles x 2 planes = 24 blocks), same phases.
template <bool AT, bool BT, bool COLP, int WMT, int WNT>
__global__ __launch_bounds__(512, 4) void gemm_planes_kernel(GemmArgs g) {
  constexpr int XT = 2;   // skip switch compiled in (no register cost here)
  constexpr int NA = 4 * WMT;   // A blocks per slot (2 WMT mn tiles x 2 planes); B: 4 WNT
  static_assert(WMT * WNT == 8 && NA + 4 * WNT == 24, "eight waves, 24 blocks per slot");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * 64 * WMT, n0 = tn * 64 * WNT;
  // two products per k-step with A rounded to bf16 (skip bit 0: a_lo * b_hi never issued): A's lo planes are not fetched (their
  // slot blocks are filled from the hi planes again - an L2 hit, which keeps every wave at three DMA pieces per k-tile and the
  // counted waits as they are) and not read, so the producer of A need not even write them (lfi_pgemm_desc.out_hi_only)
  const bool noalo = LFI_GSKIP(1);
  // this split's k-tiles: [kt0, kt0 + nkt)
  const int ktc = g.kchunk >> 4;
  const int kt0 = split * ktc;
  const int nkt = max(min(g.nkt - kt0, ktc), 0);
  char* lds = reinterpret_cast<char*>(xsmem);
  // the slot's 24 blocks (A: 4 mn tiles x 2 planes, then B: 8 x 2) are dealt to the 8 waves three at a time; a tile's two
  // planes are adjacent in memory and in the slot
  // source of this lane's 16 bytes of a piece (mn tile mt of the workgroup's panel, plane, k-tile kt):
  //   row use:        block (row tile = first + mt, column tile kt), copied as it lies: + 16 lane
  //   transposed use: column tiles first + 2 mt (+ 1 for lanes 32-63), row tile kt >> 1, rows 16 (kt & 1) + (lane & 31) / 2, the
  //                   second sub-image's rows 0-3 <-> 4-7 swapped (see the header)
  const int tsub = lane >> 5, trow = ((lane & 31) >> 1) ^ (tsub << 2);
  const int lpT = tsub * 2048 + trow * 32 + (lane & 1) * 16, lpR = lane * 16;
  const char* baseA = reinterpret_cast<const char*>(g.Ap + batch * g.pstrideA) +
                      (AT ? (long)(tm * 2 * WMT) * 4096 + lpT : (long)(tm * 2 * WMT) * g.nktA * 2048 + lpR);
  const char* baseB = reinterpret_cast<const char*>(g.Bp + batch * g.pstrideB) +
                      (BT ? (long)(tn * 2 * WNT) * 4096 + lpT : (long)(tn * 2 * WNT) * g.nktB * 2048 + lpR);
  const char* src[3];
  int doff[3];
  bool srcB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = wave * 3 + i;                       // block 0 .. 23
    const int isB = j >= NA, jj = isB ? j - NA : j;   // (mn tile, plane) = (jj >> 1, jj & 1)
    const int plane = (!isB && noalo) ? 0 : (jj & 1);
    const long mt = jj >> 1;
    src[i] = (isB ? baseB + (BT ? mt * 4096 : mt * g.nktB * 2048) : baseA + (AT ? mt * 4096 : mt * g.nktA * 2048)) + plane * 1024;
    srcB[i] = isB != 0;
    doff[i] = j * 1024;
  }
  const long rowA = (long)g.nktA * 2048, rowB = (long)g.nktB * 2048;   // bytes per row tile of either plane buffer
  auto dma = [&](int kt, int slot) {
    const int ka = kt0 + max(min(kt, nkt - 1), 0);
    const long koA = AT ? (long)(ka >> 1) * rowA + (ka & 1) * 512 : (long)ka * 2048;
    const long koB = BT ? (long)(ka >> 1) * rowB + (ka & 1) * 512 : (long)ka * 2048;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_global_load_lds((pglb_void*)(src[i] + (srcB[i] ? koB : koA)), (plds_void*)(lds + slot * QSLOT + doff[i]), 16, 0, 0);
  };
  const int wm = wave / WNT, wn = wave % WNT;
  // row use: this lane's chunk of a block. Transposed use: lane 16 g + 4 q + p addresses k row 8 (g >> 1) + q (then + 4), mn columns
  // 4 p .. + 3 of sub-image g & 1 (whose rows 0-3 <-> 4-7 are swapped); a row's chunks are swapped in rows 8-15 as they lie in memory
  const int uoff = lfi_u_plane_offset(lane & 31, lane >> 5);
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3, tkh = tg >> 1, tmh = tg & 1;
  const int toff = tmh * 512 + (((tkh * 8 + tq) ^ (tmh << 2)) * 32) + (((tp >> 1) ^ tkh) << 4) + (tp & 1) * 8;
  const int fa = (wm * 4) * 1024;             // A fragment (mt, plane) in block fa + (mt * 2 + plane) * 1024
  const int fb = NA * 1024 + (wn * 4) * 1024;  // B fragment (nt, plane) in block fb + (nt * 2 + plane) * 1024
  auto fragA = [&](int slot, int off) { return pg_frag<AT>(lds + slot * QSLOT + fa + off, uoff, toff); };
  auto fragB = [&](int slot, int off) { return pg_frag<BT>(lds + slot * QSLOT + fb + off, uoff, toff); };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (nkt > 0) {
    bf16x8 ah[2], al[2], bh[2], bl[2];
    dma(0, 0); dma(1, 1); dma(2, 2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      ah[t2] = fragA(0, (t2 * 2) * 1024);
      if (!noalo) al[t2] = fragA(0, (t2 * 2 + 1) * 1024);
      else al[t2] = ah[t2];
      bh[t2] = fragB(0, (t2 * 2) * 1024); bl[t2] = fragB(0, (t2 * 2 + 1) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    for (int t = 0; t < nkt; ++t) {
      const int nxt = cur == 2 ? 0 : cur + 1;
      bf16x8 nah[2], nal[2], nbh[2], nbl[2];
      dma(t + 3, cur);
      __builtin_amdgcn_sched_barrier(0);
      if (!LFI_GSKIP(1)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[0][nt] = PG_MFMA(al[0], bh[nt], acc[0][nt]);
      }
      if (!LFI_GSKIP(2)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[0][nt] = PG_MFMA(ah[0], bl[nt], acc[0][nt]);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[0][nt] = PG_MFMA(ah[0], bh[nt], acc[0][nt]);
      __builtin_amdgcn_sched_barrier(0);
      nah[0] = fragA(nxt, 0);
      if (!noalo) nal[0] = fragA(nxt, 1024);
      else nal[0] = nah[0];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) { nbh[nt] = fragB(nxt, (nt * 2) * 1024); nbl[nt] = fragB(nxt, (nt * 2 + 1) * 1024); }
      __builtin_amdgcn_sched_barrier(0);
      if (!LFI_GSKIP(1)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[1][nt] = PG_MFMA(al[1], bh[nt], acc[1][nt]);
      }
      if (!LFI_GSKIP(2)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[1][nt] = PG_MFMA(ah[1], bl[nt], acc[1][nt]);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[1][nt] = PG_MFMA(ah[1], bh[nt], acc[1][nt]);
      __builtin_amdgcn_sched_barrier(0);
      nah[1] = fragA(nxt, 2048);
      if (!noalo) nal[1] = fragA(nxt, 3072);
      else nal[1] = nah[1];
      asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i) { ah[i] = nah[i]; al[i] = nal[i]; bh[i] = nbh[i]; bl[i] = nbl[i]; }
      cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetched tail tiles: landed before the epilogue reuses the ring
  }
  __syncthreads();
  // the tile passes through LDS in blocks of rows that fit the ring's 72 KB: 64 rows of 256 columns, 128 rows of 128
  if (g.vecC) gemm_epilogue_wide<64 * WNT, 512, 2, COLP, true>(g, acc, reinterpret_cast<float*>(xsmem), WNT == 4 ? 64 : 128, m0, n0, wm, wn,
                                                             l31, half, batch, split, 64 * WMT);
  else gemm_epilogue_n<64 * WNT>(g, acc, m0, n0, wm, wn, l31, half, batch, split);
}

// The row-use / row-use products (cond_transform forward, gic) on v_mfma_f32_16x16x32_bf16. Same FLOP per cycle as the 32 x 32 x 16
// shape, but the chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS give-back item 7; timing-only build of the
// kernel above with its MFMAs swapped: cond_transform forward 0.574 -> 0.530 ms, gic 0.295 -> 0.271 on one box). The shape wants
// 32 k-values per instruction, a block holds 16: k-tiles are taken in PAIRS, and a ring slot holds, for one pair, ONE plane of
// either operand: sub-slot 2 p = [A: lo blocks of k-tiles 2 p, 2 p + 1 of its 2 WMT mn tiles][B: hi blocks], sub-slot 2 p + 1 =
// [A: hi][B: lo] (24 blocks either way: ring, DMA pieces, counted waits and the one barrier per 16 k are those of the kernel above).
// Lane l of a fragment read (one ds_read_b128) takes row l & 15 of its 16-row tile and k-group g = l >> 4 = chunk g >> 1 of the
// block of k-tile g & 1: the four 16-lane groups of the read then cover all 64 banks (the same k assignment on both operands, so
// the sum is unchanged). Every fragment is read once per pair, as before, and at most three of the four fragment sets (A hi, A lo,
// B hi, B lo: 16 VGPRs each beside the 64 accumulators) are alive at a time:
//   phase 2 p     (A lo, B hi in registers):  acc += a_lo b_hi (16 MFMAs, 32 deep);  A hi read at once, B lo row by row into the
//                                             registers a_lo leaves
//   phase 2 p + 1:                            acc += a_hi b_hi, then a_hi b_lo;      the next pair's B hi read after the first product
//                                             (into b_hi's registers), its A lo row by row into those a_hi leaves
// Needs three products and an even number of k-tiles per split (the host falls back to the kernel above otherwise). Sums differ from
// the 32 x 32 kernel's in the order of their fp32 additions only.
#define PG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// TRP != 0: the product is computed transposed (B fragments as the MFMA's A operand: the same products, the same sums) so that a lane
// holds four consecutive columns of a row and the epilogue needs no LDS (gemm_epilogue_direct16<3>: plane outputs); the launcher takes it whenever the epilogue is one it covers.
#define PG_MFMA16X(a, b, c) (TRP ? PG_MFMA16(b, a, c) : PG_MFMA16(a, b, c))
// AT / BT: that operand in transposed use (a pair of k-tiles is then ONE 32-row block of its buffer; DMA row swap and
// ds_read_b64_tr_b16 as in gemm_planes16t_kernel below) - the three-product form of the weight-gradient and feature-gradient
// products (engine_backward_products = 3: small batches, three_products_everywhere).
template <bool AT, bool BT, bool COLP, int WMT, int WNT, int TRP = 0>
__global__ __launch_bounds__(512, 4) void gemm_planes16_kernel(GemmArgs g) {
  constexpr int NA = 4 * WMT;   // A blocks per sub-slot (2 WMT mn tiles x 2 k-tiles); B: 4 WNT
  static_assert(WMT * WNT == 8 && NA + 4 * WNT == 24, "eight waves, 24 blocks per slot");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * 64 * WMT, n0 = tn * 64 * WNT;
  const int ktc = g.kchunk >> 4;
  const int kt0 = split * ktc;
  const int npair = max(min(g.nkt - kt0, ktc), 0) >> 1;
  char* lds = reinterpret_cast<char*>(xsmem);
  const long rowA = (long)g.nktA * 2048, rowB = (long)g.nktB * 2048;   // bytes per row tile of either plane buffer
  // LDS-DMA through buffer descriptors built from workgroup-uniform values (panel base + the pair's offset: + 4096 per pair in
  // row use - k-tiles are column tiles -, + a row tile per pair in transposed use) + an SGPR offset + ONE per-lane VGPR
  const char* baseA = reinterpret_cast<const char*>(g.Ap + batch * g.pstrideA) +
                      (AT ? (long)(tm * 2 * WMT) * 4096 + (long)(kt0 >> 1) * rowA : (long)(tm * 2 * WMT) * rowA + (long)kt0 * 2048);
  const char* baseB = reinterpret_cast<const char*>(g.Bp + batch * g.pstrideB) +
                      (BT ? (long)(tn * 2 * WNT) * 4096 + (long)(kt0 >> 1) * rowB : (long)(tn * 2 * WNT) * rowB + (long)kt0 * 2048);
  // per-lane source offset inside a block: as it lies (row use), or with rows 16-19 <-> 20-23, 24-27 <-> 28-31 swapped (transposed)
  const int prow = lane >> 1, srow = prow ^ ((prow & 16) ? 4 : 0);
  const int laneR = lane * 16, laneT = srow * 32 + (lane & 1) * 16;
  // this wave's three blocks of a sub-slot: piece i = block 8 i + wave (which operand a piece belongs to is a compile-time fact);
  // sof: the block's lo (A) / hi (B) plane, i.e. its place in sub-slot 0; the odd sub-slots flip A back to hi and B on to lo
  int sof[3], doff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 8 * i + wave;
    const bool isB = 8 * i >= NA;
    const int jj = isB ? j - NA : j;   // (mn tile, half) = (jj >> 1, jj & 1): half = k-tile of the pair (row use) / column tile (transposed)
    const int mt = jj >> 1, h = jj & 1;
    const bool tr = isB ? BT : AT;
    sof[i] = (tr ? mt * 4096 : mt * (int)(isB ? rowB : rowA)) + h * 2048 + (isB ? 0 : 1024);
    doff[i] = j * 1024;
  }
#define PG_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ptr), 0, 0x7fffffff, 0x00020000)
  auto dma = [&](int ph, int slot) {   // sub-slot ph of the ring (past the end: the last pair's again, never used)
    const int pp = max(min(ph >> 1, npair - 1), 0);
    const char* pa = baseA + (AT ? pp * rowA : (long)pp * 4096);
    const char* pb = baseB + (BT ? pp * rowB : (long)pp * 4096);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool isB = 8 * i >= NA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(PG_RSRC(isB ? pb : pa), (plds_void*)(lds + slot * QSLOT + doff[i]), 16,
                                               (isB ? BT : AT) ? laneT : laneR, sof[i] + ((ph & 1) ? (isB ? 1024 : -1024) : 0), 0, 0);
    }
  };
#undef PG_RSRC
  const int wm = wave / WNT, wn = wave % WNT;
  const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int fo = (g4 & 1) * 1024 + lfi_u_plane_offset(lane & 15, g4 >> 1);   // row use: 16-row tile 1 of a block pair: + 512
  const int kr = 16 * (g4 & 1) + 8 * (g4 >> 1) + tq;                           // transposed use: k row of this lane's first read
  const int toff = (kr ^ ((kr & 16) ? 4 : 0)) * 32 + (((tp >> 1) ^ ((kr >> 3) & 1)) << 4) + (tp & 1) * 8;
  // (the ring position is a compile-time constant below: every fragment read is a VGPR address + an immediate)
  const char* ldsA = lds + (wm * 4) * 1024;
  const char* ldsB = lds + NA * 1024 + (wn * 4) * 1024;
  auto fragA = [&](int slot, int i) -> bf16x8 {
    if constexpr (AT) return pg_frag<true>(ldsA + slot * QSLOT + i * 1024, 0, toff);
    else return *reinterpret_cast<const bf16x8*>(ldsA + slot * QSLOT + (i >> 1) * 2048 + (i & 1) * 512 + fo);
  };
  auto fragB = [&](int slot, int i) -> bf16x8 {
    if constexpr (BT) return pg_frag<true>(ldsB + slot * QSLOT + i * 1024, 0, toff);
    else return *reinterpret_cast<const bf16x8*>(ldsB + slot * QSLOT + (i >> 1) * 2048 + (i & 1) * 512 + fo);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (npair > 0) {
    bf16x8 al[4], bh[4];
    dma(0, 0); dma(1, 1); dma(2, 2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) { al[i] = fragA(0, i); bh[i] = fragB(0, i); }
    asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // one pair; SX = ring slot of its sub-slot 2 p (2 p + 1: the next slot; the next pair's first: the one after)
    auto pair = [&](auto SX, int p) {
      constexpr int sx = decltype(SX)::value, sy = (sx + 1) % 3, sn = (sx + 2) % 3;
      bf16x8 ah[4], bl[4], nal[4], nbh[4];
      // ---- phase 2 p: a_lo b_hi
      dma(2 * p + 3, sx);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = PG_MFMA16X(al[i], bh[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) ah[q] = fragA(sy, q);
        }
        bl[i] = fragB(sy, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- phase 2 p + 1: a_hi b_hi, a_hi b_lo
      dma(2 * p + 4, sy);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = PG_MFMA16X(ah[i], bh[j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = PG_MFMA16X(ah[i], bl[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) nbh[q] = fragB(sn, q);
        }
        nal[i] = fragA(sn, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) { al[i] = nal[i]; bh[i] = nbh[i]; }
    };
    for (int p = 0; p < npair; p += 3) {
      pair(std::integral_constant<int, 0>{}, p);
      if (p + 1 >= npair) break;
      pair(std::integral_constant<int, 2>{}, p + 1);
      if (p + 2 >= npair) break;
      pair(std::integral_constant<int, 1>{}, p + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (TRP != 0) {
    // (laundered: otherwise the epilogue's address arithmetic is hoisted above the main loop and lives through it in registers)
    int lanez = lane, m0z = m0, n0z = n0;
    asm volatile("" : "+v"(lanez), "+s"(m0z), "+s"(n0z));
    gemm_epilogue_direct16<TRP>(g, acc, m0z, n0z, wm, wn, lanez, batch, split);
  } else {
    __syncthreads();
    gemm_epilogue_wide<64 * WNT, 512, 2, COLP, true, true>(g, acc, reinterpret_cast<float*>(xsmem), WNT == 4 ? 64 : 128, m0, n0, wm, wn,
                                                           lane & 31, lane >> 5, batch, split, 64 * WMT);
  }
}

// The backward products - TWO products per k (A rounded to bf16: a_hi b_hi + a_hi b_lo), B in transposed use, A by rows (dpre, the
// feature gradient) or transposed (the weight gradients) - on v_mfma_f32_16x16x32_bf16, with the pair / sub-slot structure of the
// kernel above. In the 32 x 32 kernel such a product keeps a k-tile's whole overhead (barrier, eight fragment reads, three DMA
// pieces of which A's lo blocks are dummies) for 8 MFMAs instead of 12; here a pair of k-tiles costs two phases of 16 MFMAs:
//   sub-slot 2 p     = [A: hi blocks of the pair][B: hi blocks]     (24 blocks, three DMA pieces per wave)
//   sub-slot 2 p + 1 = [ -                      ][B: lo blocks]     (4 WNT blocks: NY = WNT / 2 pieces per wave)
//   phase 2 p     (A hi, B hi in registers):  acc += a_hi b_hi;  B lo read row by row under it
//   phase 2 p + 1:                            acc += a_hi b_lo;  the next pair's B hi read at once (b_hi's registers), its A hi row by
//                                             row into those a_hi leaves
// Transposed use of a pair: the pair IS one 32-row block of the buffer (k = its rows), an mn tile of 32 = its column tiles 2 mt,
// 2 mt + 1 = two 16-mn fragments. The LDS-DMA copies a block with rows 16-19 <-> 20-23 and 24-27 <-> 28-31 trading places (per-lane
// source address); lane 16 g + 4 q + p of ds_read_b64_tr_b16 addresses k row 16 (g & 1) + 8 (g >> 1) + q (then + 4), columns 4 p .. + 3:
// the same k assignment as the row-use read (k-tile g & 1, chunk g >> 1), and the two 16-lane groups of a 32-lane half land in
// opposite halves of the 256-byte bank row.
template <bool AT, bool COLP, int WMT, int WNT, int TRP = 0>
__global__ __launch_bounds__(512, 4) void gemm_planes16t_kernel(GemmArgs g) {
  constexpr int NA = 4 * WMT, NY = WNT / 2;   // A blocks per sub-slot; DMA pieces per wave of the lo sub-slot (its 4 WNT B blocks)
  static_assert(WMT * WNT == 8 && NA + 4 * WNT == 24, "eight waves, 24 blocks per slot");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn, batch, split;
  gemm_tile_of_block(g, &tm, &tn, &batch, &split);
  const int m0 = tm * 64 * WMT, n0 = tn * 64 * WNT;
  const int ktc = g.kchunk >> 4;
  const int kt0 = split * ktc;
  const int npair = max(min(g.nkt - kt0, ktc), 0) >> 1;
  char* lds = reinterpret_cast<char*>(xsmem);
  const long rowA = (long)g.nktA * 2048, rowB = (long)g.nktB * 2048;   // bytes per row tile of either plane buffer
  // panel bases at this split's first pair. Row use: + 4096 per pair (k-tiles are column tiles); transposed: + a row tile per pair
  // (64-bit: folded into the descriptor's base, pair by pair)
  const char* baseA = reinterpret_cast<const char*>(g.Ap + batch * g.pstrideA) +
                      (AT ? (long)(tm * 2 * WMT) * 4096 + (long)(kt0 >> 1) * rowA : (long)(tm * 2 * WMT) * rowA + (long)kt0 * 2048);
  const char* baseB = reinterpret_cast<const char*>(g.Bp + batch * g.pstrideB) + (long)(tn * 2 * WNT) * 4096 + (long)(kt0 >> 1) * rowB;
  // per-lane source offset inside a block: as it lies (row use), or with the row swap of the header (transposed use)
  const int prow = lane >> 1, srow = prow ^ ((prow & 16) ? 4 : 0);
  const int laneR = lane * 16, laneT = srow * 32 + (lane & 1) * 16;
  // this wave's pieces: piece i = block 8 i + wave, so that which operand a piece belongs to is a compile-time fact. Sub-slot 2 p:
  // [A (NA)][B hi (4 WNT)]; sub-slot 2 p + 1: B lo, NY pieces, in B's place
  int sofX[3], doffX[3], sofY[2], doffY[2];   // (NY <= 2 used; a template-dependent bound here breaks the host pass of this hipcc)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 8 * i + wave;
    const bool isB = 8 * i >= NA;
    const int jj = isB ? j - NA : j;   // (mn tile, half) = (jj >> 1, jj & 1): half = k-tile of the pair (row use) / column tile (transposed)
    const int mt = jj >> 1, h = jj & 1;
    sofX[i] = (isB || AT) ? mt * 4096 + h * 2048 : mt * (int)rowA + h * 2048;
    doffX[i] = j * 1024;
  }
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    const int jj = 8 * i + wave;                      // block 0 .. 4 WNT - 1 of B
    sofY[i] = (jj >> 1) * 4096 + (jj & 1) * 2048 + 1024;
    doffY[i] = (NA + jj) * 1024;
  }
  // (descriptors are built at the call: base + the pair's offset, all wave-uniform)
#define PG_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ptr), 0, 0x7fffffff, 0x00020000)
  auto dmaX = [&](int pair, int slot) {   // (past the end: the last pair again, never used)
    const int pp = max(min(pair, npair - 1), 0);
    const char* pa = baseA + (AT ? pp * rowA : (long)pp * 4096);
    const char* pb = baseB + pp * rowB;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool isB = 8 * i >= NA;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(PG_RSRC(isB ? pb : pa), (plds_void*)(lds + slot * QSLOT + doffX[i]), 16, (isB || AT) ? laneT : laneR,
                                               sofX[i], 0, 0);
    }
  };
  auto dmaY = [&](int pair, int slot) {
    const int pp = max(min(pair, npair - 1), 0);
    const char* pb = baseB + pp * rowB;
#pragma unroll
    for (int i = 0; i < NY; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(PG_RSRC(pb), (plds_void*)(lds + slot * QSLOT + doffY[i]), 16, laneT, sofY[i], 0, 0);
  };
#undef PG_RSRC
  const int wm = wave / WNT, wn = wave % WNT;
  const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  // row use: one ds_read_b128 (16-row tile 1 of a block pair: + 512). Transposed use: two ds_read_b64_tr_b16 on ONE block (16 mn
  // columns x the pair's 32 k rows), the second at toff ^ 128 (k rows + 4)
  const int fo = (g4 & 1) * 1024 + lfi_u_plane_offset(lane & 15, g4 >> 1);
  const int kr = 16 * (g4 & 1) + 8 * (g4 >> 1) + tq;
  const int toff = (kr ^ ((kr & 16) ? 4 : 0)) * 32 + (((tp >> 1) ^ ((kr >> 3) & 1)) << 4) + (tp & 1) * 8;
  const char* ldsA = lds + (wm * 4) * 1024;
  const char* ldsB = lds + NA * 1024 + (wn * 4) * 1024;
  auto fragA = [&](int slot, int i) -> bf16x8 {
    if constexpr (AT) return pg_frag<true>(ldsA + slot * QSLOT + i * 1024, 0, toff);
    else return *reinterpret_cast<const bf16x8*>(ldsA + slot * QSLOT + (i >> 1) * 2048 + (i & 1) * 512 + fo);
  };
  auto fragB = [&](int slot, int i) -> bf16x8 { return pg_frag<true>(ldsB + slot * QSLOT + i * 1024, 0, toff); };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (npair > 0) {
    bf16x8 ah[4], bh[4];
    dmaX(0, 0); dmaY(0, 1); dmaX(1, 2);
    if (NY == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) { ah[i] = fragA(0, i); bh[i] = fragB(0, i); }
    asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // one pair; SX = ring slot of its sub-slot 2 p (2 p + 1: the next slot; the next pair's first: the one after)
    auto pair = [&](auto SX, int p) {
      constexpr int sx = decltype(SX)::value, sy = (sx + 1) % 3, sn = (sx + 2) % 3;
      bf16x8 bl[4], nah[4], nbh[4];
      // ---- phase 2 p: a_hi b_hi
      dmaY(p + 1, sx);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = PG_MFMA16X(ah[i], bh[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        bl[i] = fragB(sy, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NY == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- phase 2 p + 1: a_hi b_lo
      dmaX(p + 2, sy);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = PG_MFMA16X(ah[i], bl[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) nbh[q] = fragB(sn, q);
        }
        nah[i] = fragA(sn, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) { ah[i] = nah[i]; bh[i] = nbh[i]; }
    };
    for (int p = 0; p < npair; p += 3) {
      pair(std::integral_constant<int, 0>{}, p);
      if (p + 1 >= npair) break;
      pair(std::integral_constant<int, 2>{}, p + 1);
      if (p + 2 >= npair) break;
      pair(std::integral_constant<int, 1>{}, p + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (TRP != 0) {
    int lanez = lane, m0z = m0, n0z = n0;
    asm volatile("" : "+v"(lanez), "+s"(m0z), "+s"(n0z));
    gemm_epilogue_direct16<TRP>(g, acc, m0z, n0z, wm, wn, lanez, batch, split);
  } else {
    __syncthreads();
    gemm_epilogue_wide<64 * WNT, 512, 2, COLP, true, true>(g, acc, reinterpret_cast<float*>(xsmem), WNT == 4 ? 64 : 128, m0, n0, wm, wn,
                                                           lane & 31, lane >> 5, batch, split, 64 * WMT);
  }
}

// fp32 (rows x cols, row pitch ldx) -> bf16 hi / lo planes, zero padded to rows_pad x 16 nkt: block ((rt * nkt + kt) * 2 + plane),
// thread l of a block converts row rt * 32 + (l & 31), columns kt * 16 + 8 (l >> 5) .. + 7 into the chunk at lfi_u_plane_offset.
// One thread per (block pair, lane): 8 floats in (two 16-byte loads when the row allows), 16 + 16 bytes out.
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ X, long ldx, int rows, int cols, long nblk,
                                                             int nkt, int vec, __bf16* __restrict__ out) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nblk * 64; idx += (long)gridDim.x * 256) {
    const int l = (int)(idx & 63);
    const long q = idx >> 6;
    const int kt = (int)(q % nkt);
    const long rt = q / nkt;
    const long row = rt * 32 + (l & 31);
    const int k0 = kt * 16 + 8 * (l >> 5);
    float v[8];
    if (row < rows && vec && k0 + 8 <= cols) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(X + row * ldx + k0), b = *reinterpret_cast<const f32x4*>(X + row * ldx + k0 + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (row < rows && k0 + e < cols) ? X[row * ldx + k0 + e] : 0.0f;
    }
    uint4 h, lo;
    split2(v[0], v[1], &h.x, &lo.x); split2(v[2], v[3], &h.y, &lo.y);
    split2(v[4], v[5], &h.z, &lo.z); split2(v[6], v[7], &h.w, &lo.w);
    char* dst = reinterpret_cast<char*>(out) + q * 2048 + lfi_u_plane_offset(l & 31, l >> 5);
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + 1024) = lo;
  }
}

// the three-product 16 x 16 x 32 kernel: whole pairs of k-tiles in every split, the through-LDS or the direct plane epilogue (LFI_PGEMM_16=0:
// the 32 x 32 x 16 kernel everywhere)
bool planes16_ok(const GemmArgs& a) {
  const char* e = getenv("LFI_PGEMM_16");   // (read per call: the tests compare the two kernels)
  if (e && e[0] == '0') return false;
  return a.vecC && a.skip == 0 && (a.nkt & 1) == 0 && ((a.kchunk >> 4) & 1) == 0;
}

// the two-product 16 x 16 x 32 kernel: B transposed, A either way, skip bit 0 alone (LFI_PGEMM_16T=0: the 32 x 32 x 16 kernel)
bool planes16t_ok(const GemmArgs& a) {
  const char* e = getenv("LFI_PGEMM_16T");
  if (e && e[0] == '0') return false;
  return a.vecC && a.skip == 1 && (a.nkt & 1) == 0 && ((a.kchunk >> 4) & 1) == 0;
}

// the LDS-free epilogue of the 16 x 16 x 32 kernels (gemm_epilogue_direct16): bias / LeakyReLU / plane outputs only; not the act-2
// mask, accumulate, column sums, fp32 rows or split-K partials (LFI_PGEMM_DIRECT=0: always through LDS)
// -> 0: not covered; 3: planes only
int planes16_direct_mode(const GemmArgs& a) {
  const char* e = getenv("LFI_PGEMM_DIRECT");
  if (e && e[0] == '0') return 0;
  if (!(a.vecC && a.act != 2 && a.accumulate == 0 && !a.colpart && (a.N & 3) == 0) || a.splitk > 1) return 0;
  if (a.Cr && !a.storeC) return ((a.strideC & 31) == 0 && (a.colCr & 15) == 0) ? 3 : 0;
  return 0;
}

// MODE 4 of the direct epilogue: plane outputs only, act 2 with the mask in planes, column sums, 128 x 256 tiles
bool planes16_direct_mask_ok(const GemmArgs& a) {
  const char* e = getenv("LFI_PGEMM_DIRECT");
  if (e && e[0] == '0') return false;
  return a.vecC && a.act == 2 && a.Gr && !a.G && a.accumulate == 0 && a.colpart && a.splitk == 1 && a.Cr && !a.storeC && !a.bias &&
         (a.N & 3) == 0 && (a.strideC & 31) == 0 && (a.colCr & 15) == 0 && (a.colGr & 15) == 0 && (a.ldpart & 3) == 0;
}

template <bool COLP, int WMT, int WNT>
int launch_planes(const GemmArgs& a, int at, int bt, dim3 grid, size_t lds, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_planes_kernel<false, false, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_planes_kernel<false, true, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e3 = hipFuncSetAttribute((const void*)gemm_planes_kernel<true, false, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e4 = hipFuncSetAttribute((const void*)gemm_planes_kernel<true, true, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      lfi_set_error("lfi_gemm_planes: cannot reserve %zu bytes of LDS", lds);
      return LFI_ERR_LAUNCH;
    }
    attr = true;
  }
  if (bt && planes16t_ok(a)) {
    static bool attr16t = false;
    if (!attr16t) {
      hipError_t e1 = hipFuncSetAttribute((const void*)gemm_planes16t_kernel<false, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipError_t e2 = hipFuncSetAttribute((const void*)gemm_planes16t_kernel<true, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e1 != hipSuccess || e2 != hipSuccess) {
        lfi_set_error("lfi_gemm_planes: cannot reserve %zu bytes of LDS", lds);
        return LFI_ERR_LAUNCH;
      }
      attr16t = true;
    }
    if (COLP && WMT == 2 && !at && planes16_direct_mask_ok(a)) {   // the in-place dpre product: planes + mask from planes + column sums
      static bool attr4 = false;
      if (!attr4) {
        if (hipFuncSetAttribute((const void*)gemm_planes16t_kernel<false, true, 2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          lfi_set_error("lfi_gemm_planes: cannot reserve %zu bytes of LDS", lds);
          return LFI_ERR_LAUNCH;
        }
        attr4 = true;
      }
      hipLaunchKernelGGL((gemm_planes16t_kernel<false, true, 2, 4, 4>), grid, dim3(512), lds, st, a);
    } else if (at) hipLaunchKernelGGL((gemm_planes16t_kernel<true, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((gemm_planes16t_kernel<false, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
  } else if (!at && !bt && planes16_ok(a) && !COLP && planes16_direct_mode(a)) {
    static bool attrd16 = false;
    if (!attrd16) {
      if (hipFuncSetAttribute((const void*)gemm_planes16_kernel<false, false, false, WMT, WNT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        lfi_set_error("lfi_gemm_planes: cannot reserve %zu bytes of LDS", lds);
        return LFI_ERR_LAUNCH;
      }
      attrd16 = true;
    }
    hipLaunchKernelGGL((gemm_planes16_kernel<false, false, false, WMT, WNT, 3>), grid, dim3(512), lds, st, a);
  } else if (planes16_ok(a)) {   // three products, any use of either operand
    static bool attr16 = false;
    if (!attr16) {
      bool ok = hipFuncSetAttribute((const void*)gemm_planes16_kernel<false, false, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)gemm_planes16_kernel<false, true, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)gemm_planes16_kernel<true, false, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      ok = ok && hipFuncSetAttribute((const void*)gemm_planes16_kernel<true, true, COLP, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      if (!ok) {
        lfi_set_error("lfi_gemm_planes: cannot reserve %zu bytes of LDS", lds);
        return LFI_ERR_LAUNCH;
      }
      attr16 = true;
    }
    if (!at && !bt) hipLaunchKernelGGL((gemm_planes16_kernel<false, false, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
    else if (!at) hipLaunchKernelGGL((gemm_planes16_kernel<false, true, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
    else if (!bt) hipLaunchKernelGGL((gemm_planes16_kernel<true, false, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((gemm_planes16_kernel<true, true, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
  } else if (!at && !bt) hipLaunchKernelGGL((gemm_planes_kernel<false, false, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
  else if (!at) hipLaunchKernelGGL((gemm_planes_kernel<false, true, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
  else if (!bt) hipLaunchKernelGGL((gemm_planes_kernel<true, false, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
  else hipLaunchKernelGGL((gemm_planes_kernel<true, true, COLP, WMT, WNT>), grid, dim3(512), lds, st, a);
  return LFI_OK;
}

// 128 x 256 tiles unless 128-wide column tiles waste at least 10 % less of the matrix pipe on padding columns (lfi_pgemm_desc.tile
// pins it: 1 = 128 x 256, 2 = 256 x 128)
bool planes_tall_tile(const lfi_pgemm_desc* d) {
  if (d->tile == 1) return false;
  if (d->tile == 2) return true;
  const double w256 = (double)lfi_cdiv(d->N, 256) * 256.0 / d->N, w128 = (double)lfi_cdiv(d->N, 128) * 128.0 / d->N;
  return w128 < 0.9 * w256;
}

}  // namespace

extern "C" long lfi_planes_elems(long rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return ((rows + 255) / 256 * 256) * (long)((cols + 15) / 16 * 16) * 2;
}

extern "C" int lfi_planes_from_f32(const float* X, long ldx, long rows, int cols, void* planes, void* stream) {
  LFI_REQUIRE(rows >= 0 && cols >= 0, "lfi_planes_from_f32: bad dims %ld x %d", rows, cols);
  if (rows == 0 || cols == 0) return LFI_OK;
  LFI_REQUIRE(X && planes, "lfi_planes_from_f32: null pointer");
  LFI_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "lfi_planes_from_f32: planes must be 16-byte aligned");
  const int nkt = (cols + 15) / 16;
  const long nblk = (rows + 255) / 256 * 8 * nkt;   // (row tile, k-tile) block pairs, rows padded to whole 256-row panels
  const int vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0 && (ldx & 3) == 0) ? 1 : 0;
  const long threads = nblk * 64;
  hipLaunchKernelGGL(planes_from_f32_kernel, dim3((unsigned)min((threads + 255) / 256, 65535L * 8)), dim3(256), 0, (hipStream_t)stream,
                     X, ldx, (int)rows, cols, nblk, nkt, vec, reinterpret_cast<__bf16*>(planes));
  LFI_LAUNCH_CHECK("lfi_planes_from_f32");
  return LFI_OK;
}

extern "C" long lfi_gemm_planes_work_floats(const lfi_pgemm_desc* d) {
  if (!d || d->splitk <= 1) return 0;
  return (long)d->batch * d->splitk * d->M * d->N;
}

// Rows of the partial column-sum matrix lfi_gemm_planes fills when colsum_part is set (one per 64-row epilogue pass), or 0 when
// this product cannot (split K, or a batch that is not laid side by side in C's columns).
extern "C" long lfi_gemm_planes_colpart_rows(const lfi_pgemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->splitk > 1) return 0;
  if (d->batch > 1 && !(d->strideC > 0 && d->strideC * d->batch <= d->ldc)) return 0;
  if (d->ldc % 4 != 0 || d->strideC % 4 != 0 || (d->act == 2 && d->accumulate != 0)) return 0;
  return planes_tall_tile(d) ? (long)lfi_cdiv(d->M, 256) * 2 : (long)lfi_cdiv(d->M, 128) * 2;   // two passes per tile either way
}

extern "C" int lfi_gemm_planes(const lfi_pgemm_desc* d, void* stream) {
  LFI_REQUIRE(d, "lfi_gemm_planes: null descriptor");
  LFI_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0 && d->batch >= 1 && d->batch <= 65535, "lfi_gemm_planes: bad dims M=%d N=%d K=%d batch=%d",
              d->M, d->N, d->K, d->batch);
  if (d->M == 0 || d->N == 0) return LFI_OK;
  const bool store = d->store_f32 != 0;
  LFI_REQUIRE(d->Ap && d->Bp && (d->C || !store), "lfi_gemm_planes: null operand");
  LFI_REQUIRE(store || d->Cr, "lfi_gemm_planes: store_f32 = 0 and no plane output: the result would go nowhere");
  LFI_REQUIRE(d->K > 0, "lfi_gemm_planes: K = 0");
  LFI_REQUIRE(d->act >= 0 && d->act <= 2 && (d->act != 2 || d->G || d->Gr), "lfi_gemm_planes: bad act %d", d->act);
  const int nkt = (d->K + 15) / 16;
  // (row use: the buffer's column tiles are the product's k-tiles; transposed use: they are its mn tiles, two per 32-wide mn tile)
  LFI_REQUIRE((d->a_fmt ? 2L * d->a_nkt >= (d->M + 15) / 16 : d->a_nkt >= nkt) && (d->b_fmt ? 2L * d->b_nkt >= (d->N + 15) / 16 : d->b_nkt >= nkt),
              "lfi_gemm_planes: plane buffers hold %ld / %ld column tiles per row tile: too few for this product", (long)d->a_nkt, (long)d->b_nkt);
  LFI_REQUIRE(d->a_nkt > 0 && d->b_nkt > 0, "lfi_gemm_planes: a_nkt / b_nkt = column tiles per row tile of the plane buffers");
  LFI_REQUIRE((reinterpret_cast<uintptr_t>(d->Ap) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->Bp) & 15) == 0 &&
              (d->a_stride & 7) == 0 && (d->b_stride & 7) == 0, "lfi_gemm_planes: planes must be 16-byte aligned");
  int splitk = d->splitk < 1 ? 1 : d->splitk;
  if (splitk > nkt) splitk = nkt;
  LFI_REQUIRE(splitk == 1 || d->work, "lfi_gemm_planes: splitk needs a workspace");
  LFI_REQUIRE(splitk == 1 || (!d->Cr && !d->colsum_part && store), "lfi_gemm_planes: plane outputs / column sums need splitk = 1");
  const bool planes_io = d->Cr || d->Gr;
  if (planes_io) {
    LFI_REQUIRE(d->batch == 1 || (d->strideC > 0 && d->strideC * d->batch <= (d->C ? d->ldc : d->strideC * d->batch) && d->strideC % 32 == 0),
                "lfi_gemm_planes: plane outputs need batch entries side by side in C's columns, 32-column granular");
    LFI_REQUIRE(d->cr_col0 % 16 == 0 && d->gr_col0 % 16 == 0, "lfi_gemm_planes: plane outputs must start on block boundaries");
    LFI_REQUIRE((!d->Cr || (reinterpret_cast<uintptr_t>(d->Cr) & 15) == 0) && (!d->Gr || (reinterpret_cast<uintptr_t>(d->Gr) & 15) == 0),
                "lfi_gemm_planes: planes must be 16-byte aligned");
  }
  GemmArgs a = {};
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.C = d->C; a.ldc = d->ldc; a.bias = d->bias; a.G = d->G; a.ldg = d->ldg;
  a.strideC = d->strideC; a.strideBias = d->strideBias; a.strideG = d->strideG;
  a.accumulate = d->accumulate; a.act = d->act; a.slope = d->slope;
  a.splitk = splitk;
  {
    // K split in whole PAIRS of k-tiles where K allows it (the 16 x 16 x 32 kernels take k-tiles two at a time)
    int ktc = splitk > 1 ? lfi_cdiv(nkt, splitk) : nkt;
    if (splitk > 1 && (nkt & 1) == 0 && (ktc & 1) && (long)(splitk - 1) * (ktc + 1) < nkt) ++ktc;
    a.kchunk = ktc * 16;
  }
  a.work = d->work;
  a.Ap = reinterpret_cast<const __bf16*>(d->Ap); a.Bp = reinterpret_cast<const __bf16*>(d->Bp);
  a.nkt = nkt; a.nktA = (int)d->a_nkt; a.nktB = (int)d->b_nkt; a.pstrideA = d->a_stride; a.pstrideB = d->b_stride;
  a.skip = d->skip & 3;
  a.colpart = d->colsum_part; a.ldpart = d->ld_part;
  a.storeC = store ? 1 : 0;
  a.Cr = reinterpret_cast<__bf16*>(d->Cr); a.nktCr = d->cr_nkt; a.colCr = d->cr_col0;
  a.Gr = reinterpret_cast<const __bf16*>(d->Gr); a.nktGr = d->gr_nkt; a.colGr = d->gr_col0;
  {
    // the wide (through-LDS) epilogue needs 16-byte granular fp32 rows; it is also the only one that can emit / read planes
    const bool partial = splitk > 1;
    const bool c_ok = partial ? (((long)d->M * d->N) % 4 == 0 && d->N % 4 == 0 && (reinterpret_cast<uintptr_t>(d->work) & 15) == 0)
                              : (!store || ((reinterpret_cast<uintptr_t>(d->C) & 15) == 0 && d->ldc % 4 == 0 && d->strideC % 4 == 0));
    const bool g_ok = d->act != 2 || d->Gr || ((reinterpret_cast<uintptr_t>(d->G) & 15) == 0 && d->ldg % 4 == 0 && d->strideG % 4 == 0);
    a.vecC = (c_ok && g_ok && !(d->act == 2 && d->accumulate != 0)) ? 1 : 0;
    LFI_REQUIRE(a.vecC || !(planes_io || d->colsum_part), "lfi_gemm_planes: plane outputs / column sums need 16-byte granular C rows");
  }
  a.hiOnly = d->out_hi_only ? 1 : 0;
  const bool tall = planes_tall_tile(d);
  a.tiles_m = lfi_cdiv(d->M, tall ? 256 : 128);
  a.tiles_n = lfi_cdiv(d->N, tall ? 128 : 256);
  {
    // an XCD walks its run of tiles in groups of `gm` tile rows, column by column. PMC (cond_transform forward, 80 MB of
    // operand planes): groups of 8 rows fetch 790 MB per launch into the L2s, one group of 14 per XCD (B once per XCD, but
    // 6.4 MB of A panels per 4 MB L2) 1033 MB; the launch time is the same either way (0.571 / 0.574 ms): L2 misses are
    // served by the Infinity Cache at ~2 TB/s and are not what bounds the kernel. LFI_PGEMM_GM overrides.
    static int gm_env = -1;
    if (gm_env < 0) {
      const char* e = getenv("LFI_PGEMM_GM");
      gm_env = e ? atoi(e) : 0;
    }
    a.gm = gm_env > 0 ? gm_env : 8;
  }
  const size_t lds = (size_t)QRING * QSLOT;   // 72 KB: two workgroups per CU; the epilogue's 64 x 260 floats fit inside
  dim3 grid(a.tiles_m * a.tiles_n, d->batch, splitk);
  const int rc = tall ? (a.colpart ? launch_planes<true, 4, 2>(a, d->a_fmt, d->b_fmt, grid, lds, (hipStream_t)stream)
                                   : launch_planes<false, 4, 2>(a, d->a_fmt, d->b_fmt, grid, lds, (hipStream_t)stream))
                      : (a.colpart ? launch_planes<true, 2, 4>(a, d->a_fmt, d->b_fmt, grid, lds, (hipStream_t)stream)
                                   : launch_planes<false, 2, 4>(a, d->a_fmt, d->b_fmt, grid, lds, (hipStream_t)stream));
  if (rc) return rc;
  LFI_LAUNCH_CHECK("lfi_gemm_planes");
  if (splitk > 1) {
    const long mn = (long)d->M * d->N;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool red4 = d->N % 4 == 0 && al16(d->work) && al16(d->C) && d->ldc % 4 == 0 && d->strideC % 4 == 0 &&
                      (!d->G || (al16(d->G) && d->ldg % 4 == 0 && d->strideG % 4 == 0));
    if (red4) {
      dim3 rgrid((unsigned)min((long)lfi_cdiv(mn / 4, 256), 2048L), d->batch);
      hipLaunchKernelGGL(gemm_splitk_reduce4_kernel, rgrid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
      dim3 rgrid((unsigned)min((long)lfi_cdiv(mn, 256), 2048L), d->batch);
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, rgrid, dim3(256), 0, (hipStream_t)stream, a);
    }
    LFI_LAUNCH_CHECK("lfi_gemm_planes split-k reduce");
  }
  return LFI_OK;
}
