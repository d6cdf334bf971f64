// The per-frame conditioning of the autoregressive sampler as ONE launch (glow/models.py:567-596 inference -> :598-615
// create_conditioning -> FlowStep's cond_transform + the coupling cell's input projection, per generated frame):
//   c    = LeakyReLU(pre_static[frame] + window @ Wct[:, window columns]^T)         (B x Ks D; window = the last hist1 generated frames)
//   gic  = c[:, k D:(k+1) D] @ W_ih[k][:, Ch:]^T + b_ih[k]                           (Ks x B x G)
// Round 3 ran these as two GEMMs per frame (42 us each at batch 1024, two and three times their MFMA time: 33 MB of c written and read
// back per frame, two launches' ramps). Here workgroup (64-row tile, flow step k) computes its 64 x D slice of c, keeps it in LDS as
// fp16 hi / lo pieces and multiplies it by W_ih[k] at once; c never reaches memory. Arithmetic as the per-frame GEMMs it replaces
// (lfi_gemm_desc.precision 9): three v_mfma_f32_16x16x32_f16 products of two-piece fp16 operands, x = hi + lo with hi = fp16(x),
// lo = fp16(x - hi) - 2^-22 relative, fp32 accumulation; the caller keeps the range guard of that path.
// Both products are computed TRANSPOSED (weights as the MFMA's A operand, samples as its B operand): a lane then holds four
// consecutive output columns of one sample - 8-byte LDS writes of the c pieces, 16-byte loads of pre_static and stores of gic.
// All weight operands are converted once per call into MFMA fragment order (1 KB per fragment, one coalesced 16-byte load per lane):
// they are the same for every frame; the window's fragments are rebuilt per frame by a small kernel that replaces the aligned
// gather the GEMM path needed.
#include "lfi_common.h"
#include <type_traits>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
union HFrag { uint4 u; h16x8 v; };

constexpr int SC_D = 512;                   // cond_transform width per flow step (8 waves x 64 columns)
constexpr int SC_PITCH = SC_D * 2 + 16;     // bytes per row of a c plane in LDS: 65 16-byte slots = 1 (mod 16)
constexpr int SC_NT = 512;

// 16-byte k-chunk of lane group q in MFMA step m of a row of nchunk chunks: the chunks of the groups that share a ds_read_b128
// cycle lie half a row apart (a multiple of 256 bytes at D = 512), rows advance by one 16-byte slot: all 64 banks (lfi_encoder.hip,
// enc_chunk16: the same rule)
__host__ __device__ inline int sc_chunk(int m, int q, int nchunk) { return 2 * m + (q >> 1) + (nchunk >> 1) * (q & 1); }

__device__ __forceinline__ void sc_split(float x, _Float16* hi, _Float16* lo) {
  const _Float16 h = (_Float16)x;
  *hi = h;
  *lo = (_Float16)(x - (float)h);
}

// ---- weights -> fragments, once per call
// wf1: cond_transform's window columns as the A operand of phase 1. Fragment (k, wave, m, ci, plane): lane l, element e holds
//      Wct[k D + 64 wave + 16 ci + (l & 15)][col0 + 32 m + 8 (l >> 4) + e]  (0 beyond K1)
// wf2: W_ih[k][:, Ch:] (G x D) as the A operand of phase 2. Fragment (k, wave, m, ni, plane): lane l, element e holds
//      wc[k][NGT 16 wave + 16 ni + (l & 15)][8 sc_chunk(m, l >> 4, 64) + e]
__global__ __launch_bounds__(256) void sc_wfrag1_kernel(const float* __restrict__ wct, long ldw, int col0, int K1, int Ks, int NM1,
                                                        _Float16* __restrict__ dst) {
  const long n = (long)Ks * 8 * NM1 * 4 * 2 * 512;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63), plane = (int)((idx >> 9) & 1);
    long q = idx >> 10;
    const int ci = (int)(q & 3); q >>= 2;
    const int m = (int)(q % NM1); q /= NM1;
    const int w = (int)(q & 7), k = (int)(q >> 3);
    const int kk = 32 * m + 8 * (l >> 4) + e;
    const float v = kk < K1 ? wct[((long)k * SC_D + 64 * w + 16 * ci + (l & 15)) * ldw + col0 + kk] : 0.0f;
    _Float16 hi, lo;
    sc_split(v, &hi, &lo);
    dst[idx] = plane ? lo : hi;
  }
}
__global__ __launch_bounds__(256) void sc_wfrag2_kernel(const float* __restrict__ wc, int G, int Ks, int NGT, _Float16* __restrict__ dst) {
  const long n = (long)Ks * 8 * 16 * NGT * 2 * 512;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63), plane = (int)((idx >> 9) & 1);
    long q = idx >> 10;
    const int ni = (int)(q % NGT); q /= NGT;
    const int m = (int)(q & 15); q >>= 4;
    const int w = (int)(q & 7), k = (int)(q >> 3);
    const int col = NGT * 16 * w + 16 * ni + (l & 15);
    const int kc = 8 * sc_chunk(m, l >> 4, SC_D / 8) + e;
    const float v = wc[((long)k * G + col) * SC_D + kc];
    _Float16 hi, lo;
    sc_split(v, &hi, &lo);
    dst[idx] = plane ? lo : hi;
  }
}
// ---- the frame's window -> fragments (B operand of phase 1). Fragment (16-row tile rt, m, plane): lane l, element e holds
//      window[16 rt + (l & 15)][32 m + 8 (l >> 4) + e] = faces[row][off + kk]  (0 beyond K1 / B)
__global__ __launch_bounds__(256) void sc_xfrag_kernel(const float* __restrict__ faces, long ld_faces, long off, int K1, int B, int NM1,
                                                       int ntile, _Float16* __restrict__ dst) {
  const long n = (long)ntile * NM1 * 2 * 512;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63), plane = (int)((idx >> 9) & 1);
    long q = idx >> 10;
    const int m = (int)(q % NM1);
    const int rt = (int)(q / NM1);
    const int row = 16 * rt + (l & 15), kk = 32 * m + 8 * (l >> 4) + e;
    const float v = (row < B && kk < K1) ? faces[(long)row * ld_faces + off + kk] : 0.0f;
    _Float16 hi, lo;
    sc_split(v, &hi, &lo);
    dst[idx] = plane ? lo : hi;
  }
}

#define SC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
// timing-only removal builds (tools/build_variant.sh; results are garbage): -DSC_NO_W1 / -DSC_NO_W2 re-read the first step's weight
// fragments in every step of phase 1 / 2 (same instructions, L1 hits instead of the L2 stream)
#ifdef SC_NO_W1
#define SC_TIMING_W1(m) 0
#else
#define SC_TIMING_W1(m) (m)
#endif
#ifdef SC_NO_W2
#define SC_TIMING_W2(m) 0
#else
#define SC_TIMING_W2(m) (m)
#endif

struct ScArgs {
  const uint4* wf1;      // [Ks][8][NM1][4][2] fragments
  const uint4* wf2;      // [Ks][8][16][NGT][2]
  const uint4* xf;       // [ntile][NM1][2] window fragments made by sc_xfrag_kernel, or null: the kernel splits the window itself
  const float* faces;    // xf == null: the window rows, faces[row * ld_faces + off + kk] (kk < K1), row pitch and offset 8-byte aligned
  long ld_faces, off, faces_bytes;
  int K1;
  unsigned* reset;       // words the LAST-dispatched... (any one) workgroup zeroes before it ends: the reverse chain's ticket /
  int reset_words;       // progress words for the launch that follows in the stream (instead of a memset node per frame), or null
  const float* pre;      // B x Ks D: the frame's rows of pre_static (bias included)
  const float* b_ih;     // Ks x G
  float* gic;            // Ks x B x G
  int B, Ks, G, NM1;
  float slope;
};

// grid (Ks, row tiles of 16 RT); 512 threads = 8 waves: wave w owns c columns [64 w, 64 w + 64) in phase 1 and gic columns
// [16 NGT w, 16 NGT (w + 1)) in phase 2. RT = 4 (64 samples, the default): one workgroup per CU (133 KB of LDS, ~230 VGPRs). RT = 2
// (32 samples, round 6, LFI_SAMPLE_COND_ROWS=32): 67 KB and <= 128 VGPRs, TWO workgroups per CU - measured slower (see the launch
// site). Same arithmetic per output element either way: bit-identical (tested).
// XF = false (round 5): the window's fp16 pieces are made in registers from the fp32 frames - 16 workgroups repeat the split of
// a row tile, 256 values per lane, instead of a launch of its own per generated frame (12.6 us + its boundary)
template <int NGT, bool XF, int RT>
__global__ __launch_bounds__(SC_NT) __attribute__((amdgpu_waves_per_eu(RT == 2 ? 4 : 2, RT == 2 ? 4 : 2))) void sc_cond_kernel(ScArgs a) {
  constexpr int SC_ROWS = 16 * RT;
  constexpr int SC_PLANE = SC_ROWS * SC_PITCH;
  extern __shared__ __attribute__((aligned(16))) char sc_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups are dealt round-robin over the 8 XCDs: flow step on grid.x, so that the row tiles of one step - which read the same
  // 1.3 MB of weight fragments - share ONE 4 MB L2 (two steps per XCD at Ks = 16) instead of every L2 seeing all 20 MB
  const int k = blockIdx.x, rt = blockIdx.y;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int NM1 = a.NM1;
  const long KD = (long)a.Ks * SC_D;
  const unsigned lane16 = (unsigned)lane * 16u;
  const __amdgpu_buffer_rsrc_t bw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wf1 + (long)(k * 8 + wave) * NM1 * 4 * 2 * 64), 0,
                                                                      NM1 * 4 * 2 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t bx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.xf + (long)(rt * RT) * NM1 * 2 * 64), 0,
                                                                     RT * NM1 * 2 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t bw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wf2 + (long)(k * 8 + wave) * 16 * NGT * 2 * 64), 0,
                                                                      16 * NGT * 2 * 1024, 0x00020000);
  auto ld = [&](__amdgpu_buffer_rsrc_t r, int frag) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, (unsigned)frag * 1024u, 0));
  };

  // ---- phase 1: acc1[ci][ri] = (Wct rows 16 ci ..) x (window rows 16 ri ..)^T, 32 k per step
  f32x4 acc1[4][RT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < RT; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    HFrag wa[4][2], xa[RT][2], wb[4][2], xb[RT][2];   // [tile][plane], two buffers
    // (XF = false) fragment (16-row tile t, step m): lane l, element e = window[16 t + (l & 15)][32 m + 8 (l >> 4) + e]; the frames
    // through a buffer descriptor (reads past the last row's end return 0), 8-byte loads (the window starts (t - hist1) C floats
    // into a 16-byte aligned row: 8-byte aligned for even C)
    const __amdgpu_buffer_rsrc_t bf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.faces), 0,
                                                                       (int)(a.faces_bytes > 0xfffffff0L ? 0xfffffff0L : a.faces_bytes), 0x00020000);
    unsigned frow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const long row = min((long)rt * SC_ROWS + 16 * t + l15, (long)a.B - 1);   // rows past B repeat the last one (never stored)
      frow[t] = (unsigned)((row * a.ld_faces + a.off) * 4 + 32 * g4);
    }
    typedef unsigned sc_u32x2 __attribute__((ext_vector_type(2)));
    auto load1 = [&](int m, HFrag (&w)[4][2], HFrag (&x)[RT][2]) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w[t][pl].u = ld(bw1, (SC_TIMING_W1(m) * 4 + t) * 2 + pl);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        if (XF) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) x[t][pl].u = ld(bx, (t * NM1 + m) * 2 + pl);
        } else {
          const int kk0 = 32 * m + 8 * g4;
#pragma unroll
          for (int c2 = 0; c2 < 4; ++c2) {
            // (bit_cast, not an initialisation: the builtin's 8-byte result is not this vector type, and converting it splats ONE dword)
            const sc_u32x2 v2 = __builtin_bit_cast(sc_u32x2, __builtin_amdgcn_raw_buffer_load_b64(bf, frow[t] + 8u * c2, (unsigned)m * 128u, 0));
            // (through scalars: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index - this
            // hipcc loads from the vector's base address; seen in the IR, round 5)
            const unsigned u0 = v2[0], u1 = v2[1];
            const float v0 = kk0 + 2 * c2 < a.K1 ? __builtin_bit_cast(float, u0) : 0.0f;
            const float v1 = kk0 + 2 * c2 + 1 < a.K1 ? __builtin_bit_cast(float, u1) : 0.0f;
            _Float16 h0, l0, h1, l1;
            sc_split(v0, &h0, &l0);
            sc_split(v1, &h1, &l1);
            x[t][0].v[2 * c2] = h0; x[t][0].v[2 * c2 + 1] = h1;
            x[t][1].v[2 * c2] = l0; x[t][1].v[2 * c2 + 1] = l1;
          }
        }
      }
    };
    auto mma1 = [&](const HFrag (&w)[4][2], const HFrag (&x)[RT][2]) {
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) acc1[ci][ri] = SC_MFMA(w[ci][1].v, x[ri][0].v, acc1[ci][ri]);
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) acc1[ci][ri] = SC_MFMA(w[ci][0].v, x[ri][1].v, acc1[ci][ri]);
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) acc1[ci][ri] = SC_MFMA(w[ci][0].v, x[ri][0].v, acc1[ci][ri]);
      }
    };
    if constexpr (RT == 2) {
      // two workgroups per CU = 128 VGPRs: the weight fragments of a step travel in two halves (c column tiles 0-1, then 2-3), one
      // half in flight under the other's products; the window's fragments once per step. Per accumulator the same three products
      // per step in the same order as below.
      (void)load1; (void)mma1; (void)wb;
      HFrag (&wh0)[2][2] = reinterpret_cast<HFrag (&)[2][2]>(wa[0]);   // wa[0..1] and wa[2..3] serve as the two half buffers
      HFrag (&wh1)[2][2] = reinterpret_cast<HFrag (&)[2][2]>(wa[2]);
      auto loadw = [&](int m, int hf, HFrag (&w)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) w[t][pl].u = ld(bw1, (SC_TIMING_W1(m) * 4 + 2 * hf + t) * 2 + pl);
      };
      auto loadx = [&](int m, HFrag (&x)[RT][2]) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) x[t][pl].u = ld(bx, (t * NM1 + m) * 2 + pl);
      };
      auto mmah = [&](int hf, const HFrag (&w)[2][2], const HFrag (&x)[RT][2]) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int ci = 2 * hf + c2;
#pragma unroll
          for (int ri = 0; ri < RT; ++ri) acc1[ci][ri] = SC_MFMA(w[c2][1].v, x[ri][0].v, acc1[ci][ri]);
#pragma unroll
          for (int ri = 0; ri < RT; ++ri) acc1[ci][ri] = SC_MFMA(w[c2][0].v, x[ri][1].v, acc1[ci][ri]);
#pragma unroll
          for (int ri = 0; ri < RT; ++ri) acc1[ci][ri] = SC_MFMA(w[c2][0].v, x[ri][0].v, acc1[ci][ri]);
        }
      };
      static_assert(XF || RT != 2, "the 32-sample tile takes the window's fragments from sc_xfrag_kernel / the reverse chain");
      loadx(0, xa);
      loadw(0, 0, wh0);
      for (int m = 0; m < NM1; m += 2) {
        loadw(m, 1, wh1);
        if (m + 1 < NM1) loadx(m + 1, xb);
        __builtin_amdgcn_sched_barrier(0);
        mmah(0, wh0, xa);
        if (m + 1 < NM1) loadw(m + 1, 0, wh0);
        __builtin_amdgcn_sched_barrier(0);
        mmah(1, wh1, xa);
        if (m + 1 < NM1) {
          loadw(m + 1, 1, wh1);
          if (m + 2 < NM1) loadx(m + 2, xa);
          __builtin_amdgcn_sched_barrier(0);
          mmah(0, wh0, xb);
          if (m + 2 < NM1) loadw(m + 2, 0, wh0);
          __builtin_amdgcn_sched_barrier(0);
          mmah(1, wh1, xb);
        }
      }
    } else {
    load1(0, wa, xa);
    int m = 0;
    for (; m + 2 < NM1; m += 2) {
      load1(m + 1, wb, xb);
      __builtin_amdgcn_sched_barrier(0);
      mma1(wa, xa);
      load1(m + 2, wa, xa);
      __builtin_amdgcn_sched_barrier(0);
      mma1(wb, xb);
    }
    if (m + 1 < NM1) {
      load1(m + 1, wb, xb);
      __builtin_amdgcn_sched_barrier(0);
      mma1(wa, xa);
      mma1(wb, xb);
    } else {
      mma1(wa, xa);
    }
    }
  }
  // the first W_ih fragments travel while the c pieces are made
  HFrag w2a[NGT][2], w2b[NGT][2];
  auto load2 = [&](int m, HFrag (&w)[NGT][2]) {
#pragma unroll
    for (int t = 0; t < NGT; ++t)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) w[t][pl].u = ld(bw2, (SC_TIMING_W2(m) * NGT + t) * 2 + pl);
  };
  load2(0, w2a);
  // ---- c = LeakyReLU(acc1 + pre_static) -> fp16 hi / lo pieces in LDS, row-major [64][D] per plane
  // accumulator (ci, ri) register r = c column 64 wave + 16 ci + 4 g4 + r of sample 16 ri + l15
  // The frame's rows of pre_static come out of HBM (33 MB per frame, read once): all 16 loads of a lane are issued here in one
  // batch, from a clamped row so that none sits under a branch - one round trip between the phases where there were four (a
  // wait for four loads in front of each row tile's pieces), with nothing for the matrix pipe to do meanwhile.
  f32x4 ps[RT][4];
#pragma unroll
  for (int ri = 0; ri < RT; ++ri) {
    const long row = min((long)rt * SC_ROWS + 16 * ri + l15, (long)a.B - 1);   // (rows past B: values never stored)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
      ps[ri][ci] = *reinterpret_cast<const f32x4*>(a.pre + row * KD + (long)k * SC_D + 64 * wave + 16 * ci + 4 * g4);
  }
#pragma unroll
  for (int ri = 0; ri < RT; ++ri) {
    const int rl = 16 * ri + l15;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      h16x4 hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc1[ci][ri][r] + ps[ri][ci][r];
        v = v > 0.0f ? v : v * a.slope;
        _Float16 h, l;
        sc_split(v, &h, &l);
        hi[r] = h; lo[r] = l;
      }
      char* dst = sc_smem + rl * SC_PITCH + (64 * wave + 16 * ci + 4 * g4) * 2;
      *reinterpret_cast<h16x4*>(dst) = hi;
      *reinterpret_cast<h16x4*>(dst + SC_PLANE) = lo;
    }
  }
  __syncthreads();
  // ---- phase 2: acc2[ni][ri] = (W_ih rows 16 ni ..) x (c rows 16 ri ..)^T over D = 512: 16 steps
  f32x4 acc2[NGT][RT];
#pragma unroll
  for (int i = 0; i < NGT; ++i)
#pragma unroll
    for (int j = 0; j < RT; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const char* cbase = sc_smem + l15 * SC_PITCH + 16 * ((g4 >> 1) + (SC_D / 16) * (g4 & 1));   // + 32 m per step, + 16 rows per ri
    h16x8 cah[RT], cal[RT], cbh[RT], cbl[RT];
    auto loadc = [&](int m, h16x8 (&ch)[RT], h16x8 (&cl)[RT]) {
#pragma unroll
      for (int ri = 0; ri < RT; ++ri) {
        ch[ri] = *reinterpret_cast<const h16x8*>(cbase + ri * 16 * SC_PITCH + 32 * m);
        cl[ri] = *reinterpret_cast<const h16x8*>(cbase + ri * 16 * SC_PITCH + 32 * m + SC_PLANE);
      }
    };
    auto mma2 = [&](const HFrag (&w)[NGT][2], const h16x8 (&ch)[RT], const h16x8 (&cl)[RT]) {
#pragma unroll
      for (int ni = 0; ni < NGT; ++ni) {
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) acc2[ni][ri] = SC_MFMA(w[ni][1].v, ch[ri], acc2[ni][ri]);
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) acc2[ni][ri] = SC_MFMA(w[ni][0].v, cl[ri], acc2[ni][ri]);
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) acc2[ni][ri] = SC_MFMA(w[ni][0].v, ch[ri], acc2[ni][ri]);
      }
    };
    loadc(0, cah, cal);
#pragma unroll
    for (int m = 0; m < 16; m += 2) {
      load2(m + 1, w2b);
      loadc(m + 1, cbh, cbl);
      __builtin_amdgcn_sched_barrier(0);
      mma2(w2a, cah, cal);
      if (m + 2 < 16) {
        load2(m + 2, w2a);
        loadc(m + 2, cah, cal);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma2(w2b, cbh, cbl);
    }
  }
  // ---- gic[k][row][16 NGT wave + 16 ni + 4 g4 + r] = acc2 + b_ih
  // (the biases first, all of them: a load between two groups of stores makes the second group wait for the first to drain)
  f32x4 bv[NGT];
#pragma unroll
  for (int ni = 0; ni < NGT; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(a.b_ih + (long)k * a.G + NGT * 16 * wave + 16 * ni + 4 * g4);
#pragma unroll
  for (int ni = 0; ni < NGT; ++ni)
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
      acc2[ni][ri] += bv[ni];
      asm volatile("" : "+v"(acc2[ni][ri]));   // here, not inside the row predicate of a store (where the compiler sinks it, and with
    }                                          // it a wait that also drains the stores in front: they count in vmcnt too)
#pragma unroll
  for (int ni = 0; ni < NGT; ++ni) {
    const int col = NGT * 16 * wave + 16 * ni + 4 * g4;
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
      const long row = (long)rt * SC_ROWS + 16 * ri + l15;
      if (row < a.B) *reinterpret_cast<f32x4*>(a.gic + ((long)k * a.B + row) * a.G + col) = acc2[ni][ri];
    }
  }
  // the reverse chain that follows in the stream starts from zeroed ticket / abort / progress words: one workgroup of this launch
  // clears them (the previous frame's chain has completed: stream order), instead of a memset node per frame
  if (a.reset && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = tid; i < a.reset_words; i += SC_NT) a.reset[i] = 0u;
}

}  // namespace

// shapes the fused conditioning takes (the caller falls back to its two GEMMs otherwise)
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_sample_cond_ok(int D, int G, int K1) {
  const char* e = getenv("LFI_SAMPLE_FUSED");
  if (e && e[0] == '0') return 0;
  return (D == SC_D && (G == 384 || G == 512) && K1 >= 1 && K1 <= 512) ? 1 : 0;
}
// bytes of fragment storage: [weights of phase 1][weights of phase 2][the frame's window]
extern "C" __attribute__((visibility("hidden"))) long lfi_internal_sample_cond_bytes(int B, int Ks, int G, int K1) {
  const long NM1 = (K1 + 31) / 32, NGT = G / 128, ntile = ((long)B + 63) / 64 * 4;
  return ((long)Ks * 8 * NM1 * 4 * 2 + (long)Ks * 8 * 16 * NGT * 2 + ntile * NM1 * 2) * 1024 + 256;
}
// where the window's fragments live inside `frags` (written by sc_xfrag_kernel or by the reverse chain's step-0 workgroups)
extern "C" __attribute__((visibility("hidden"))) void* lfi_internal_sample_cond_xfrag_ptr(void* frags, int Ks, int G, int K1) {
  const long NM1 = (K1 + 31) / 32, NGT = G / 128;
  return reinterpret_cast<_Float16*>(frags) + (long)Ks * 8 * NM1 * 4 * 2 * 512 + (long)Ks * 8 * 16 * NGT * 2 * 512;
}
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_sample_cond_prepare(const float* wct, long ldw, int col0, int K1, const float* wc,
                                                                                      int Ks, int G, void* frags, void* stream) {
  const int NM1 = (K1 + 31) / 32, NGT = G / 128;
  _Float16* f1 = reinterpret_cast<_Float16*>(frags);
  _Float16* f2 = f1 + (long)Ks * 8 * NM1 * 4 * 2 * 512;
  const long n1 = (long)Ks * 8 * NM1 * 4 * 2 * 512, n2 = (long)Ks * 8 * 16 * NGT * 2 * 512;
  hipLaunchKernelGGL(sc_wfrag1_kernel, dim3((unsigned)min((n1 + 255) / 256, 65535L * 4)), dim3(256), 0, (hipStream_t)stream, wct, ldw, col0, K1,
                     Ks, NM1, f1);
  hipLaunchKernelGGL(sc_wfrag2_kernel, dim3((unsigned)min((n2 + 255) / 256, 65535L * 4)), dim3(256), 0, (hipStream_t)stream, wc, G, Ks, NGT, f2);
  LFI_LAUNCH_CHECK("sampler conditioning: weight fragments");
  return LFI_OK;
}
// faces_floats: floats from `faces` to the end of its buffer (the window loads are bounds-checked against it); reset / reset_words: the
// reverse chain's state words to clear for the launch that follows (or null: the caller memsets them)
extern "C" __attribute__((visibility("hidden"))) int lfi_internal_sample_cond(const float* faces, long ld_faces, long off, int K1, int B, int Ks, int G,
                                                                              const float* pre, const float* b_ih, void* frags, float* gic,
                                                                              float slope, long faces_floats, unsigned* reset, int reset_words,
                                                                              int have_xfrag, void* stream) {
  const int NM1 = (K1 + 31) / 32, NGT = G / 128;
  const int ntile = (B + 63) / 64 * 4;
  _Float16* f1 = reinterpret_cast<_Float16*>(frags);
  _Float16* f2 = f1 + (long)Ks * 8 * NM1 * 4 * 2 * 512;
  _Float16* fx = f2 + (long)Ks * 8 * 16 * NGT * 2 * 512;
  // The window's fp16 pieces come from round 4's fragment kernel in front of this one (default). LFI_SAMPLE_XFRAG=0 splits them
  // inside the conditioning kernel instead (needs 8-byte aligned window rows; bit-identical, tested): one launch less per frame,
  // but 16 workgroups repeat each row tile's split with 8-byte loads - measured 60.0 against 57.0 - 57.4 ms per 1024 x 300 call
  // (profiles/round5_sampler_ab.md): kept as the switch only.
  const char* xe = getenv("LFI_SAMPLE_XFRAG");
  const bool inreg = (xe && xe[0] == '0') && ld_faces % 2 == 0 && off % 2 == 0 && (reinterpret_cast<uintptr_t>(faces) & 7) == 0 &&
                     faces_floats > 0;
  if (!inreg && !have_xfrag) {   // (have_xfrag: the previous frame's reverse chain left this frame's window fragments - lfi_flow.hip, RevChain.xf)
    const long nx = (long)ntile * NM1 * 2 * 512;
    hipLaunchKernelGGL(sc_xfrag_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, (hipStream_t)stream, faces, ld_faces, off, K1, B, NM1, ntile,
                       fx);
  }
  ScArgs a = {};
  a.wf1 = reinterpret_cast<const uint4*>(f1); a.wf2 = reinterpret_cast<const uint4*>(f2); a.xf = inreg ? nullptr : reinterpret_cast<const uint4*>(fx);
  a.faces = faces; a.ld_faces = ld_faces; a.off = off; a.faces_bytes = faces_floats * 4; a.K1 = K1;
  a.reset = reset; a.reset_words = reset_words;
  a.pre = pre; a.b_ih = b_ih; a.gic = gic; a.B = B; a.Ks = Ks; a.G = G; a.NM1 = NM1; a.slope = slope;
  // samples per workgroup: 64 (one workgroup per CU). LFI_SAMPLE_COND_ROWS=32: the two-per-CU tile VERDICT r5 asked for (67 KB of LDS,
  // 126 VGPRs, no spills) - built, bit-identical, and SLOWER: 44.3 against 40.6 ms of per-frame graphs per 1024 x 300 call, 48.96 against
  // 45.13 ms per call, same box, alternating (profiles/round6_sampler_ab.md): every workgroup streams its flow step's 1.3 MB of weight
  // fragments whatever its height, so half the rows per workgroup is twice the L2 -> CU traffic per frame (0.67 GB), and the second
  // resident workgroup does not buy that back. Kept as the switch only.
  // (the in-register window split, LFI_SAMPLE_XFRAG=0, and the LSTM cell's four gate blocks - 82 spilled VGPRs at the 128-register
  // cap - keep the 64-sample tile)
  const char* re = getenv("LFI_SAMPLE_COND_ROWS");
  const int RT = ((re && re[0] == '3') && !inreg && NGT == 3) ? 2 : 4;
  const size_t lds = (size_t)2 * 16 * RT * SC_PITCH;
  static bool attr = false;
  if (!attr) {
    const size_t l4 = (size_t)2 * 64 * SC_PITCH, l2 = (size_t)2 * 32 * SC_PITCH;
    bool ok = true;
#define SC_ATTR(N)                                                                                                                             \
  ok = ok && hipFuncSetAttribute((const void*)sc_cond_kernel<N, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4) == hipSuccess &&  \
       hipFuncSetAttribute((const void*)sc_cond_kernel<N, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4) == hipSuccess
    SC_ATTR(3); SC_ATTR(4);
    ok = ok && hipFuncSetAttribute((const void*)sc_cond_kernel<3, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) == hipSuccess;
#undef SC_ATTR
    if (!ok) {
      lfi_set_error("sampler conditioning: cannot reserve %zu bytes of LDS", l4);
      return LFI_ERR_LAUNCH;
    }
    attr = true;
  }
  const dim3 grid(Ks, (B + 16 * RT - 1) / (16 * RT));
#define SC_LAUNCH(N)                                                                                                \
  do {                                                                                                              \
    if (inreg) hipLaunchKernelGGL((sc_cond_kernel<N, false, 4>), grid, dim3(SC_NT), lds, (hipStream_t)stream, a);   \
    else hipLaunchKernelGGL((sc_cond_kernel<N, true, 4>), grid, dim3(SC_NT), lds, (hipStream_t)stream, a);          \
  } while (0)
  if (RT == 2) hipLaunchKernelGGL((sc_cond_kernel<3, true, 2>), grid, dim3(SC_NT), lds, (hipStream_t)stream, a);
  else if (NGT == 3) SC_LAUNCH(3);
  else SC_LAUNCH(4);
#undef SC_LAUNCH
  LFI_LAUNCH_CHECK("sampler conditioning");
  return LFI_OK;
}
