// The conditional-Glow flow itself: FlowStep.normal_flow / reverse_flow (glow/models.py:311-373), f_seq.forward
// (:204-214), ActNorm2d / InvertibleConv1x1 / LinearZeros (glow/modules.py), SeqGlow.loss (:563-565), and their
// gradients, as one "cell" kernel per (timestep n, flow step k).
//
// Dependency structure. Cell (n, k) needs the flow variable from (n, k-1) and the coupling net's recurrent state from
// (n-1, k) (the GRUCell hidden state is carried over timesteps, glow/models.py:193-214), so the (n, k) grid is walked
// in anti-diagonals: all cells with n + k = d are independent and run in ONE launch (up to min(N, Ks) cells x B/16
// batch tiles of workgroups); N + Ks - 1 launches replace the reference's N * Ks * ~340 ATen calls. Everything that
// does not depend on the recurrence (cond_transform, the W_ih[:, Ch:] c product) is hoisted into big GEMMs by the
// caller (gic), and InvertibleConv1x1's W = P L U is built once per optimiser step (lfi_flow_prep) rather than per
// call (glow/modules.py:163-173).
//
// Cell kernel: 16 samples x all channels per workgroup, 4 waves; the four small GEMMs of a cell
// (a W, [z1|h] [Wz|Whh]^T, h' Wfl^T and their transposes in backward) run on v_mfma_f32_16x16x4_f32 with the A
// operand staged k-major in LDS and the weights streamed from L2 as coalesced row segments; all elementwise math
// (actnorm, gates, LinearZeros scaling, affine coupling, log-det) is fused around them in fp32.
#include <math.h>
#include <stdlib.h>

#include "lfi_common.h"

namespace {

constexpr int MB = 16;        // samples per workgroup
constexpr int NT = 512;       // threads per workgroup (8 waves: one 16-wide hidden tile each at H = 128)
constexpr int NW = NT / 64;
constexpr int LT = MB + 1;    // k-major LDS leading dimension. (ds_read_b32 / ds_write_b32 bank = dword address mod 32, 32 lanes per LDS cycle:
                              // with a pitch of 17 a column-of-k store (one row, 32 consecutive k) is conflict-free and the MFMA A-operand read
                              // of two consecutive k rows x 16 lanes puts ONE lane of the second row on the first row's bank 0 - the extra cycle
                              // SQ_LDS_BANK_CONFLICT counts on nearly every such read (36 - 40 % of the walks' LDS cycles). Round 4 tried a
                              // pitch of 16 with 2 floats after every fourth row (reads conflict-free, stores 2 - 4-way): the counter stayed at
                              // 36 - 40 % - it is not these reads that it counts - and the backward cell's Q3 went from 2.5 k to 5.4 k cycles:
                              // reverted, profiles/round4_walk_ab.md.)
constexpr float LOG2PI_F = 1.8378770664093453f;
constexpr float LN2_F = 0.6931471805599453f;

struct FlowK {
  // dims
  int B, N, C, H, D, Ks, affine, lstm;
  float eps;
  int Ch, C2, Cout, G, I, F, nbt;
  int ldc, ldo;   // row strides of the (rows x C) and (rows x Cout) stash arrays: C and Cout rounded up to 4 floats, so that the
                  // deferred weight-gradient products over them read 16-byte aligned rows (bf16x3 / vector-load paths)
  // params
  lfi_flow_params p;
  // prep
  const float *W, *Wt, *Winv, *wz_t, *whh_t, *wfl_t, *wc, *ldconst;
  // prep, zero-padded images for the register-resident cell kernels (k rows padded to 4, columns to 16)
  const float *pW, *pWt, *pwz, *pwh, *pwfl, *bwfl, *bwh, *bwz, *pWinv;
  // backward recurrent weights pre-split into bf16 hi / lo 32-k fragments (bf16 x 3 walk): [Ks][NG][H16/32][J][4 lane groups],
  // one uint4 per entry and plane; the lo plane follows the hi plane of an image
  const uint4 *xbwh, *xbwz;
  // reverse (sampling) cell weights pre-split into fp16 hi / lo 32-k fragments (flow_prep_x3h_kernel): images of pwz, pwh, pwfl and
  // pWinv, entry (32-k block b, lane group kq, column) = x3h_pack of the two f32x4 entries the cell used to load and split itself;
  // one uint4 per entry and plane, the lo plane follows the hi plane of a flow step's image. Null unless lfi_flow_prep made them.
  const uint4 *hwz, *hwh, *hwfl, *hWinv;
  int dgi_hi_only;         // the dgi planes' hi halves only (their consumers take them as a rounded A operand: two products)
  int g16;                 // backward walk (planes mode): the dgi | dgh ROWS of the backward stash are bf16 arrays of the same shapes -
                           // their readers, the thin weight-gradient products, round that operand to bf16 anyway (two products):
                           // lfi_flow_dims.gemm_precision bit 16, honoured by lfi_flow_seq_bwd_planes and lfi_flow_param_grads alike
  __bf16* bDgiR;           // backward walk (bf16x3): dgi also as operand planes of the (Ks F x G) matrix (lfi_flow_seq_bwd_planes)
  int C16, Ch16, H16, Co16, NG;
  // forward stash
  float *sA, *sY, *sX, *sH, *sG, *sO, *sL, *sC;   // sC: LSTM cell state (lstm only)
  // backward stash
  float *bDlin, *bDgi, *bDgh, *bDy, *bDx, *bDh, *bPlfl, *bPan, *bDc;   // bDc: carried d cell state (lstm only)
  float* bPbias;   // [Ks][nbt][2][G]: per-workgroup sums over timesteps and the tile's rows of dgi | dgh (persistent walk only)
  // sequence inputs
  const float* x0; int T, start;
  const float* gic;
  float gscale;
  unsigned long long* stamps;  // diagnostics only (lfi_debug_set_stamps): s_memtime at phase boundaries, else null
  int stamp_k;                 // flow step whose workgroup (tile 0) stamps (LFI_STAMP_K, default Ks / 2)
  int pipe_fence;              // 1: consumers run an agent-scope acquire after the poll and read the tile with plain loads
                               // 0: no fence, every load of a handed-off tile is an sc1 load (L1 bypass)
  unsigned* pipe;              // persistent-pipeline state (flow_pipe_*_kernel): [0] ticket, [1] abort, [4 + k * nbt + bt] progress
};

// Everything one forward cell touches, resolved to pointers for its (k, frame block).
struct CellIO {
  int k, rows;            // flow step, valid rows in this call (<= B)
  const float* x_in; long ldx;   // rows x C
  const float* h_prev;    // rows x H or null (zeros)
  const float* c_prev;    // LSTM cell state, rows x H or null (zeros); unused for GRU
  float* c_out;           // LSTM: new cell state (required when lstm)
  const float* gic;       // rows x G
  float *a_out, *y_out, *x_out, *h_out, *g_out, *o_out, *l_out;  // nullable stashes; x_out/h_out required
  long ldxo;              // leading dimension of x_out
  long ld_c, ld_o;        // leading dimensions of a_out / y_out and of o_out
  int l_accumulate;       // l_out += instead of =
  int stamp_base;         // diagnostics (lfi_debug_set_stamps): slot of this cell's first phase stamp + 1, 0 = none (rev_fast_cell)
  int state_l2;           // reverse cell: read h_prev / c_prev with L1-bypassing (sc1) loads - the persistent reverse walk re-reads
                          // the state its own workgroup stored one timestep earlier, with no kernel boundary in between
};

extern __shared__ __attribute__((aligned(16))) float flow_smem[];

__device__ __forceinline__ int rup16(int x) { return (x + 15) & ~15; }

// ---- shared phase: coupling net given z1 (Zt) and h_prev (Ht) in LDS -> new hidden (Hn, LDS) and o (Orm, LDS)
__device__ __forceinline__ void coupling_net_phase(const FlowK& f, const CellIO& io, int b0, const float* Zt, const float* Ht,
                                                   float* Hn, float* Orm, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int k = io.k, H = f.H, G = f.G, Ch = f.Ch, Cout = f.Cout;
  const float* wz = f.wz_t + (long)k * Ch * G;
  const float* wh = f.whh_t + (long)k * H * G;
  const float* bhh = f.p.b_hh + (long)k * G;
  const int nht = (H + 15) >> 4;
  if (f.lstm) {
    // torch.nn.LSTMCell (gate order i, f, g, o) from zero (h, c) at the first modelled frame (glow/models.py:181-185,
    // 209-213; the reference's own call crashes there, SURVEY.md finding 2: semantics = zero initial state)
    for (int t = wave; t < nht; t += NW) {
      const int j = t * 16 + l15;
      const bool jok = j < H;
      f32x4 gz[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16_pf<4>(gz, Zt, LT, wz + t * 16, G, H, Ch, jok, lane);
      f32x4 gh[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      mma16_pf<4>(gh, Ht, LT, wh + t * 16, G, H, H, jok, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r;
        const int row = b0 + i;
        float hnew = 0.0f;
        if (row < io.rows && jok) {
          const float* gc = io.gic + (long)row * G;
          const float ii = sigmoidf_(gz[0][r] + gh[0][r] + gc[j] + bhh[j]);
          const float ff = sigmoidf_(gz[1][r] + gh[1][r] + gc[H + j] + bhh[H + j]);
          const float gg = tanhf_(gz[2][r] + gh[2][r] + gc[2 * H + j] + bhh[2 * H + j]);
          const float oo = sigmoidf_(gz[3][r] + gh[3][r] + gc[3 * H + j] + bhh[3 * H + j]);
          const float cp = io.c_prev ? io.c_prev[(long)row * H + j] : 0.0f;
          const float c2 = ff * cp + ii * gg;
          hnew = oo * tanhf_(c2);
          io.h_out[(long)row * H + j] = hnew;
          io.c_out[(long)row * H + j] = c2;
          if (io.g_out) {
            float* gs = io.g_out + (long)row * 4 * H;
            *reinterpret_cast<f32x4*>(gs + 4 * j) = (f32x4){ii, ff, gg, oo};   // gate-interleaved stash: one 16-byte store
          }
        }
        if (jok) Hn[j * LT + i] = hnew;
      }
    }
  } else
  for (int t = wave; t < nht; t += NW) {
    const int j = t * 16 + l15;
    const bool jok = j < H;
    // input side (z1 part; the conditioning part was hoisted into gic): r, z, n chains share the A operand
    f32x4 gz[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma16_pf<3>(gz, Zt, LT, wz + t * 16, G, H, Ch, jok, lane);
    // hidden side
    f32x4 gh[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma16_pf<3>(gh, Ht, LT, wh + t * 16, G, H, H, jok, lane);
    const f32x4 ar = gz[0] + gh[0], au = gz[1] + gh[1], ain = gz[2], ahn = gh[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lq * 4 + r;
      const int row = b0 + i;
      float hnew = 0.0f;
      if (row < io.rows && jok) {
        const float* gc = io.gic + (long)row * G;
        const float rr = sigmoidf_(ar[r] + gc[j] + bhh[j]);
        const float uu = sigmoidf_(au[r] + gc[H + j] + bhh[H + j]);
        const float ghn = ahn[r] + bhh[2 * H + j];
        const float nn = tanhf_(ain[r] + gc[2 * H + j] + rr * ghn);
        const float hp = Ht[j * LT + i];
        hnew = (1.0f - uu) * nn + uu * hp;
        io.h_out[(long)row * H + j] = hnew;
        if (io.g_out) {
          float* gs = io.g_out + (long)row * 4 * H;
          *reinterpret_cast<f32x4*>(gs + 4 * j) = (f32x4){rr, uu, nn, ghn};
        }
      }
      if (jok) Hn[j * LT + i] = hnew;
    }
  }
  __syncthreads();
  // o = (h' Wfl^T + b) * exp(3 logs)    (LinearZeros, glow/modules.py:93-95)
  const float* wf = f.wfl_t + (long)k * H * Cout;
  const float* bfl = f.p.b_fl + (long)k * Cout;
  const float* lfl = f.p.l_fl + (long)k * Cout;
  const int not_ = (Cout + 15) >> 4;
  const int ldo = Cout + 1;
  for (int t = wave; t < not_; t += NW) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = tile16_lds_glb(acc, Hn, LT, wf + t * 16, Cout, H, min(16, Cout - t * 16), lane);
    const int col = t * 16 + l15;
    if (col < Cout) {
      const float bb = bfl[col], sc = expf(3.0f * lfl[col]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lq * 4 + r;
        const int row = b0 + i;
        const float o = (acc[r] + bb) * sc;
        Orm[i * ldo + col] = o;
        if (io.o_out && row < io.rows) io.o_out[(long)row * io.ld_o + col] = o;
      }
    }
  }
  __syncthreads();
}

// LDS carve for the cell kernels (floats). Every k-major block is [dim][LT].
struct Carve {
  int At, Ht, Zt, Hn, Yrm, Orm, Lg, total;
};
__host__ __device__ inline Carve carve_fwd(int C, int H, int Ch, int C2, int Cout) {
  Carve c;
  int o = 0;
  c.At = o; o += C * LT;
  c.Ht = o; o += H * LT;
  c.Zt = o; o += (Ch > 0 ? Ch : 1) * LT;
  c.Hn = o; o += H * LT;
  c.Yrm = o; o += MB * (C + 1);
  c.Orm = o; o += MB * (Cout + 1);
  c.Lg = o; o += MB * (C2 + 1);
  c.total = o;
  return c;
}

// ------------------------------------------------------------------------------------------- forward cell
__device__ void cell_forward(const FlowK& f, const CellIO& io, int b0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, k = io.k;
  const Carve cv = carve_fwd(C, H, Ch, C2, Cout);
  float* At = flow_smem + cv.At;
  float* Ht = flow_smem + cv.Ht;
  float* Zt = flow_smem + cv.Zt;
  float* Hn = flow_smem + cv.Hn;
  float* Yrm = flow_smem + cv.Yrm;
  float* Orm = flow_smem + cv.Orm;
  float* Lg = flow_smem + cv.Lg;
  const int ldy = C + 1, ldo = Cout + 1, ldl = C2 + 1;

  // P0: actnorm (glow/modules.py:45-52), stage a and h_prev k-major
  const float* anb = f.p.an_bias + (long)k * C;
  const float* anl = f.p.an_logs + (long)k * C;
  for (int idx = tid; idx < MB * C; idx += NT) {
    const int i = idx / C, c = idx - i * C;
    const int row = b0 + i;
    float a = 0.0f;
    if (row < io.rows) {
      a = (io.x_in[(long)row * io.ldx + c] + anb[c]) * expf(anl[c]);
      if (io.a_out) io.a_out[(long)row * io.ld_c + c] = a;
    }
    At[c * LT + i] = a;
  }
  for (int idx = tid; idx < MB * H; idx += NT) {
    const int i = idx / H, j = idx - i * H;
    const int row = b0 + i;
    Ht[j * LT + i] = (io.h_prev && row < io.rows) ? io.h_prev[(long)row * H + j] : 0.0f;
  }
  __syncthreads();

  // P1: y = a W   (InvertibleConv1x1.forward, glow/modules.py:186; row-vector convention)
  {
    const float* W = f.W + (long)k * C * C;
    const int nt = (C + 15) >> 4;
    for (int t = wave; t < nt; t += NW) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile16_lds_glb(acc, At, LT, W + t * 16, C, C, min(16, C - t * 16), lane);
      const int c = t * 16 + l15;
      if (c < C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lq * 4 + r;
          const int row = b0 + i;
          const float v = acc[r];
          Yrm[i * ldy + c] = v;
          if (c < Ch) Zt[c * LT + i] = v;
          if (io.y_out && row < io.rows) io.y_out[(long)row * io.ld_c + c] = v;
        }
      }
    }
  }
  __syncthreads();

  // P2 + P3: coupling net
  coupling_net_phase(f, io, b0, Zt, Ht, Hn, Orm, tid);

  // P4: coupling (glow/models.py:330-341) and pass-through half
  for (int idx = tid; idx < MB * C2; idx += NT) {
    const int i = idx / C2, jj = idx - i * C2;
    const int row = b0 + i;
    const float z2 = Yrm[i * ldy + Ch + jj];
    float z2n, lg = 0.0f;
    if (f.affine) {
      const float shift = Orm[i * ldo + 2 * jj];
      const float sraw = sigmoidf_(Orm[i * ldo + 2 * jj + 1] + 2.0f);
      const float sc = fmaxf(sraw, f.eps);
      z2n = (z2 + shift) * sc;
      lg = logf(sc);
    } else {
      z2n = z2 + Orm[i * ldo + jj];
    }
    Lg[i * ldl + jj] = lg;
    if (row < io.rows) io.x_out[(long)row * io.ldxo + Ch + jj] = z2n;
  }
  for (int idx = tid; idx < MB * Ch; idx += NT) {
    const int i = idx / Ch, c = idx - i * Ch;
    const int row = b0 + i;
    if (row < io.rows) io.x_out[(long)row * io.ldxo + c] = Yrm[i * ldy + c];
  }
  __syncthreads();
  if (tid < MB && io.l_out) {
    const int row = b0 + tid;
    if (row < io.rows) {
      float s = 0.0f;
      for (int jj = 0; jj < C2; ++jj) s += Lg[tid * ldl + jj];
      if (io.l_accumulate) io.l_out[row] += s; else io.l_out[row] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------- reverse cell
// FlowStep.reverse_flow (glow/models.py:345-373): coupling^-1 -> invconv^-1 -> actnorm^-1.
__device__ void cell_reverse(const FlowK& f, const CellIO& io, int b0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, k = io.k;
  const Carve cv = carve_fwd(C, H, Ch, C2, Cout);
  float* Yt = flow_smem + cv.At;   // y = [z1 | z2] k-major for the W^-1 product
  float* Ht = flow_smem + cv.Ht;
  float* Zt = flow_smem + cv.Zt;
  float* Hn = flow_smem + cv.Hn;
  float* Yrm = flow_smem + cv.Yrm;
  float* Orm = flow_smem + cv.Orm;
  float* Lg = flow_smem + cv.Lg;
  const int ldy = C + 1, ldo = Cout + 1, ldl = C2 + 1;

  for (int idx = tid; idx < MB * C; idx += NT) {
    const int i = idx / C, c = idx - i * C;
    const int row = b0 + i;
    const float v = row < io.rows ? io.x_in[(long)row * io.ldx + c] : 0.0f;
    Yrm[i * ldy + c] = v;
    if (c < Ch) { Zt[c * LT + i] = v; Yt[c * LT + i] = v; }
  }
  for (int idx = tid; idx < MB * H; idx += NT) {
    const int i = idx / H, j = idx - i * H;
    const int row = b0 + i;
    Ht[j * LT + i] = (io.h_prev && row < io.rows) ? io.h_prev[(long)row * H + j] : 0.0f;
  }
  __syncthreads();
  coupling_net_phase(f, io, b0, Zt, Ht, Hn, Orm, tid);
  for (int idx = tid; idx < MB * C2; idx += NT) {
    const int i = idx / C2, jj = idx - i * C2;
    const float z2n = Yrm[i * ldy + Ch + jj];
    float z2, lg = 0.0f;
    if (f.affine) {
      const float shift = Orm[i * ldo + 2 * jj];
      const float sraw = sigmoidf_(Orm[i * ldo + 2 * jj + 1] + 2.0f);
      const float sc = fmaxf(sraw, f.eps);
      z2 = z2n / sc;
      z2 = z2 - shift;
      lg = -logf(sc);
    } else {
      z2 = z2n - Orm[i * ldo + jj];
    }
    Lg[i * ldl + jj] = lg;
    Yt[(Ch + jj) * LT + i] = z2;
  }
  __syncthreads();
  {
    const float* Wi = f.Winv + (long)k * C * C;
    const float* anb = f.p.an_bias + (long)k * C;
    const float* anl = f.p.an_logs + (long)k * C;
    const int nt = (C + 15) >> 4;
    for (int t = wave; t < nt; t += NW) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile16_lds_glb(acc, Yt, LT, Wi + t * 16, C, C, min(16, C - t * 16), lane);
      const int c = t * 16 + l15;
      if (c < C) {
        const float es = expf(-anl[c]), bb = anb[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = b0 + lq * 4 + r;
          if (row < io.rows) io.x_out[(long)row * io.ldxo + c] = acc[r] * es - bb;  // scale then center (modules.py:76-79)
        }
      }
    }
  }
  if (tid < MB && io.l_out) {
    const int row = b0 + tid;
    if (row < io.rows) {
      float s = 0.0f;
      for (int jj = 0; jj < C2; ++jj) s += Lg[tid * ldl + jj];
      if (io.l_accumulate) io.l_out[row] += s; else io.l_out[row] = s;
    }
  }
}

__global__ __launch_bounds__(NT) void flow_diag_fwd_kernel(FlowK f, int d, int klo) {
  const int k = klo + blockIdx.y, n = d - k;
  const long fr = (long)n * f.B;  // first frame of this timestep
  CellIO io;
  io.k = k; io.rows = f.B;
  if (k == 0) { io.x_in = f.x0 + (long)(f.start + n) * f.C; io.ldx = (long)f.T * f.C; }
  else { io.x_in = f.sX + ((long)(k - 1) * f.F + fr) * f.ldc; io.ldx = f.ldc; }
  io.h_prev = n > 0 ? f.sH + ((long)k * f.F + fr - f.B) * f.H : nullptr;
  io.c_prev = (f.lstm && n > 0) ? f.sC + ((long)k * f.F + fr - f.B) * f.H : nullptr;
  io.c_out = f.lstm ? f.sC + ((long)k * f.F + fr) * f.H : nullptr;
  io.gic = f.gic + ((long)k * f.F + fr) * f.G;
  io.a_out = f.sA + ((long)k * f.F + fr) * f.ldc;
  io.y_out = f.sY + ((long)k * f.F + fr) * f.ldc;
  io.x_out = f.sX + ((long)k * f.F + fr) * f.ldc; io.ldxo = f.ldc;
  io.ld_c = f.ldc; io.ld_o = f.ldo;
  io.h_out = f.sH + ((long)k * f.F + fr) * f.H;
  io.g_out = f.sG + ((long)k * f.F + fr) * 4 * f.H;
  io.o_out = f.sO + ((long)k * f.F + fr) * f.ldo;
  io.l_out = f.sL + (long)k * f.F + fr;
  io.l_accumulate = 0;
  cell_forward(f, io, blockIdx.x * MB);
}

template <bool REVERSE>
__global__ __launch_bounds__(NT) void flow_step_kernel(FlowK f, CellIO io) {
  if (REVERSE) cell_reverse(f, io, blockIdx.x * MB);
  else cell_forward(f, io, blockIdx.x * MB);
}

// nll[f] = -(logdet + sum_c -0.5 (z^2 + log 2pi)) / ln 2   (SeqGlow.loss, glow/models.py:563-565)
// 64 frames per workgroup: their z rows (C floats at stride ldc) come in through LDS with coalesced loads (and leave to the
// caller's z the same way); thread i then sums frame i's row in column order - one thread per frame reading its row straight from
// memory took 30 us of the step's critical path between the two walks for 3.7 MB. C > NLL_CMAX: that form (STAGED = false).
constexpr int NLL_FR = 64, NLL_CMAX = 128;
template <bool STAGED>
__global__ __launch_bounds__(256) void flow_nll_kernel(FlowK f, float* __restrict__ z, float* __restrict__ nll) {
  extern __shared__ float nll_rows[];   // [NLL_FR][C + 1]
  const int nfr = STAGED ? NLL_FR : 256;
  const long fr0 = (long)blockIdx.x * nfr;
  const float* zbase = f.sX + (long)(f.Ks - 1) * f.F * f.ldc;
  if (STAGED) {
    const long left = f.F - fr0;
    const int tot = (int)(left < NLL_FR ? left : NLL_FR) * f.C;
    for (int e = threadIdx.x; e < tot; e += 256) {
      const int r = e / f.C, c = e - r * f.C;
      const float v = zbase[(fr0 + r) * f.ldc + c];
      nll_rows[r * (f.C + 1) + c] = v;
      if (z) z[fr0 * f.C + e] = v;
    }
    __syncthreads();
  }
  const long fr = fr0 + threadIdx.x;
  if (threadIdx.x >= nfr || fr >= f.F) return;
  float ld = f.ldconst[0];
  for (int k = 0; k < f.Ks; ++k) ld += f.sL[(long)k * f.F + fr];
  const float* zz = STAGED ? nll_rows + threadIdx.x * (f.C + 1) : zbase + fr * f.ldc;
  float lp = 0.0f;
  for (int c = 0; c < f.C; ++c) {
    const float v = zz[c];
    lp += -0.5f * (v * v + LOG2PI_F);
    if (!STAGED && z) z[fr * f.C + c] = v;
  }
  // a persistent walk that gave up (bounded spin timed out, abort word set) must not pass for a result: poison it
  const bool aborted = f.pipe && f.pipe[1] != 0u;
  nll[fr] = aborted ? __builtin_nanf("") : -(ld + lp) / LN2_F;
}

// ------------------------------------------------------------------------------------------- backward cell
struct CarveB {
  int Dl, Gi, Gh, Dy, Cy, Pl, total;
};
__host__ __device__ inline CarveB carve_bwd(int C, int H, int Cout, int G) {
  CarveB c;
  int o = 0;
  c.Dl = o; o += Cout * LT;
  c.Gi = o; o += G * LT;
  c.Gh = o; o += G * LT;
  c.Dy = o; o += C * LT;
  c.Cy = o; o += H * LT;
  c.Pl = o; o += MB * (Cout + 1);
  c.total = o;
  return c;
}

__global__ __launch_bounds__(NT) void flow_diag_bwd_kernel(FlowK f, int d, int klo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int k = klo + blockIdx.y, n = d - k;
  const int b0 = blockIdx.x * MB;
  const int B = f.B, C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, G = f.G, I = f.I;
  const long LC = f.ldc, LO = f.ldo;   // stash row strides
  const long fr = (long)n * B;
  const long kf = (long)k * f.F + fr;
  const CarveB cv = carve_bwd(C, H, Cout, G);
  float* Dl = flow_smem + cv.Dl;
  float* Gi = flow_smem + cv.Gi;
  float* Gh = flow_smem + cv.Gh;
  float* Dy = flow_smem + cv.Dy;
  float* Cy = flow_smem + cv.Cy;
  float* Pl = flow_smem + cv.Pl;
  const int ldp = Cout + 1;
  const bool last = k == f.Ks - 1;
  const float gz = f.gscale / LN2_F;   // d loss / d z = z * gz   (prior term)
  const float dl = -f.gscale / LN2_F;  // d loss / d logdet
  const float* dxo = last ? f.sX + kf * LC : f.bDx + ((long)(k + 1) * f.F + fr) * LC;
  const float dxs = last ? gz : 1.0f;

  // Q0: coupling backward + LinearZeros scale
  const float* Y = f.sY + kf * LC;
  const float* O = f.sO + kf * LO;
  const float* lfl = f.p.l_fl + (long)k * Cout;
  for (int idx = tid; idx < MB * C2; idx += NT) {
    const int i = idx / C2, jj = idx - i * C2;
    const int row = b0 + i;
    float dz2 = 0.0f, d0 = 0.0f, d1 = 0.0f, p0 = 0.0f, p1 = 0.0f;
    if (row < B) {
      const float dz2n = dxo[(long)row * LC + Ch + jj] * dxs;
      if (f.affine) {
        const float oe = O[(long)row * LO + 2 * jj], oo = O[(long)row * LO + 2 * jj + 1];
        const float sraw = sigmoidf_(oo + 2.0f);
        const float sc = fmaxf(sraw, f.eps);
        const float z2 = Y[(long)row * LC + Ch + jj];
        dz2 = dz2n * sc;
        const float dsc = dz2n * (z2 + oe) + dl / sc;
        const float dsr = sraw >= f.eps ? dsc : 0.0f;
        d0 = dz2;                             // d o_even (shift)
        d1 = dsr * sraw * (1.0f - sraw);      // d o_odd
        p0 = d0 * oe * 3.0f; p1 = d1 * oo * 3.0f;
      } else {
        const float oe = O[(long)row * LO + jj];
        dz2 = dz2n; d0 = dz2n; p0 = d0 * oe * 3.0f;
      }
    }
    Dy[(Ch + jj) * LT + i] = dz2;
    if (row < B) f.bDy[kf * LC + (long)row * LC + Ch + jj] = dz2;
    if (f.affine) {
      const int c0 = 2 * jj, c1 = 2 * jj + 1;
      const float dl0 = d0 * expf(3.0f * lfl[c0]), dl1 = d1 * expf(3.0f * lfl[c1]);
      Dl[c0 * LT + i] = dl0; Dl[c1 * LT + i] = dl1;
      Pl[i * ldp + c0] = p0; Pl[i * ldp + c1] = p1;
      if (row < B) { f.bDlin[kf * LO + (long)row * LO + c0] = dl0; f.bDlin[kf * LO + (long)row * LO + c1] = dl1; }
    } else {
      const float dl0 = d0 * expf(3.0f * lfl[jj]);
      Dl[jj * LT + i] = dl0;
      Pl[i * ldp + jj] = p0;
      if (row < B) f.bDlin[kf * LO + (long)row * LO + jj] = dl0;
    }
  }
  __syncthreads();
  if (tid < Cout) {
    float s = 0.0f;
    for (int i = 0; i < MB; ++i) s += Pl[i * ldp + tid];
    f.bPlfl[(((long)k * f.N + n) * f.nbt + blockIdx.x) * Cout + tid] = s;
  }

  // Q1: d h' = dlin Wfl + dh from the next timestep; GRU cell backward
  const int nht = (H + 15) >> 4;
  {
    const float* wfl = f.p.w_fl + (long)k * Cout * H;
    const float* dhf = (n < f.N - 1) ? f.bDh + ((long)k * f.F + fr + B) * H : nullptr;
    const float* gs_base = f.sG + kf * 4 * H;
    const float* hp_base = n > 0 ? f.sH + ((long)k * f.F + fr - B) * H : nullptr;
    if (f.lstm) {
      const float* dcf = (n < f.N - 1) ? f.bDc + ((long)k * f.F + fr + B) * H : nullptr;
      const float* cp_base = n > 0 ? f.sC + ((long)k * f.F + fr - B) * H : nullptr;
      const float* c_base = f.sC + kf * H;
      float* dco = f.bDc + kf * H;
      for (int t = wave; t < nht; t += NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile16_lds_glb(acc, Dl, LT, wfl + t * 16, H, Cout, min(16, H - t * 16), lane);
        const int j = t * 16 + l15;
        if (j < H) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = lq * 4 + r;
            const int row = b0 + i;
            float dai = 0.f, daf = 0.f, dag = 0.f, dao = 0.f;
            if (row < B) {
              const float dhn = acc[r] + (dhf ? dhf[(long)row * H + j] : 0.0f);
              const float* gs = gs_base + (long)row * 4 * H;
              const f32x4 g4 = *reinterpret_cast<const f32x4*>(gs + 4 * j);
              const float ii = g4[0], ff = g4[1], gg = g4[2], oo = g4[3];
              const float tc = tanhf_(c_base[(long)row * H + j]);
              const float cp = cp_base ? cp_base[(long)row * H + j] : 0.0f;
              const float dc2 = dhn * oo * (1.0f - tc * tc) + (dcf ? dcf[(long)row * H + j] : 0.0f);
              dai = dc2 * gg * ii * (1.0f - ii);
              daf = dc2 * cp * ff * (1.0f - ff);
              dag = dc2 * ii * (1.0f - gg * gg);
              dao = dhn * tc * oo * (1.0f - oo);
              dco[(long)row * H + j] = dc2 * ff;
              float* gi = f.bDgi + kf * G + (long)row * G;
              gi[j] = dai; gi[H + j] = daf; gi[2 * H + j] = dag; gi[3 * H + j] = dao;
              float* gh = f.bDgh + kf * G + (long)row * G;
              gh[j] = dai; gh[H + j] = daf; gh[2 * H + j] = dag; gh[3 * H + j] = dao;
            }
            Gi[j * LT + i] = dai; Gi[(H + j) * LT + i] = daf; Gi[(2 * H + j) * LT + i] = dag; Gi[(3 * H + j) * LT + i] = dao;
            Gh[j * LT + i] = dai; Gh[(H + j) * LT + i] = daf; Gh[(2 * H + j) * LT + i] = dag; Gh[(3 * H + j) * LT + i] = dao;
            Cy[j * LT + i] = 0.0f;
          }
        }
      }
    } else
    for (int t = wave; t < nht; t += NW) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile16_lds_glb(acc, Dl, LT, wfl + t * 16, H, Cout, min(16, H - t * 16), lane);
      const int j = t * 16 + l15;
      if (j < H) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lq * 4 + r;
          const int row = b0 + i;
          float dar = 0.f, dau = 0.f, dan = 0.f, danr = 0.f, cy = 0.f;
          if (row < B) {
            const float dhn = acc[r] + (dhf ? dhf[(long)row * H + j] : 0.0f);
            const float* gs = gs_base + (long)row * 4 * H;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gs + 4 * j);
            const float rr = g4[0], uu = g4[1], nn = g4[2], ghn = g4[3];
            const float hp = hp_base ? hp_base[(long)row * H + j] : 0.0f;
            const float du = dhn * (hp - nn);
            const float dn = dhn * (1.0f - uu);
            cy = dhn * uu;
            dan = dn * (1.0f - nn * nn);
            dau = du * uu * (1.0f - uu);
            dar = dan * ghn * rr * (1.0f - rr);
            danr = dan * rr;
            float* gi = f.bDgi + kf * G + (long)row * G;
            gi[j] = dar; gi[H + j] = dau; gi[2 * H + j] = dan;
            float* gh = f.bDgh + kf * G + (long)row * G;
            gh[j] = dar; gh[H + j] = dau; gh[2 * H + j] = danr;
          }
          Gi[j * LT + i] = dar; Gi[(H + j) * LT + i] = dau; Gi[(2 * H + j) * LT + i] = dan;
          Gh[j * LT + i] = dar; Gh[(H + j) * LT + i] = dau; Gh[(2 * H + j) * LT + i] = danr;
          Cy[j * LT + i] = cy;
        }
      }
    }
  }
  __syncthreads();

  // Q2: d h_prev = dgh Whh + carry (to timestep n-1);  d z1 = dgi W_ih[:, :Ch] + pass-through
  {
    const float* whh = f.p.w_hh + (long)k * G * H;
    if (n > 0) {
      float* dho = f.bDh + kf * H;
      for (int t = wave; t < nht; t += NW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile16_lds_glb(acc, Gh, LT, whh + t * 16, H, G, min(16, H - t * 16), lane);
        const int j = t * 16 + l15;
        if (j < H) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = lq * 4 + r;
            const int row = b0 + i;
            if (row < B) dho[(long)row * H + j] = acc[r] + Cy[j * LT + i];
          }
        }
      }
    }
    const float* wih = f.p.w_ih + (long)k * G * I;
    const int nzt = (Ch + 15) >> 4;
    for (int t = NW - 1 - wave; t < nzt; t += NW) {  // start from the other end so these tiles land on the less loaded waves
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile16_lds_glb(acc, Gi, LT, wih + t * 16, I, G, min(16, Ch - t * 16), lane);
      const int c = t * 16 + l15;
      if (c < Ch) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lq * 4 + r;
          const int row = b0 + i;
          float v = 0.0f;
          if (row < B) {
            v = acc[r] + dxo[(long)row * LC + c] * dxs;
            f.bDy[kf * LC + (long)row * LC + c] = v;
          }
          Dy[c * LT + i] = v;
        }
      }
    }
  }
  __syncthreads();

  // Q3: d a = dy W^T ; actnorm backward ; d x_in to flow step k-1
  {
    const float* Wt = f.Wt + (long)k * C * C;
    const float* anl = f.p.an_logs + (long)k * C;
    const float* A = f.sA + kf * LC;
    float* dxi = k > 0 ? f.bDx + kf * LC : nullptr;
    float* pan = f.bPan + (((long)k * f.N + n) * f.nbt + blockIdx.x) * 2 * C;
    const int nt = (C + 15) >> 4;
    for (int t = wave; t < nt; t += NW) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile16_lds_glb(acc, Dy, LT, Wt + t * 16, C, C, min(16, C - t * 16), lane);
      const int c = t * 16 + l15;
      float sl = 0.0f, sb = 0.0f;
      if (c < C) {
        const float es = expf(anl[c]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = b0 + lq * 4 + r;
          if (row < B) {
            const float da = acc[r];
            sl += da * A[(long)row * LC + c];
            sb += da * es;
            if (dxi) dxi[(long)row * LC + c] = da * es;
          }
        }
      }
      sl += __shfl_xor(sl, 16, 64); sl += __shfl_xor(sl, 32, 64);
      sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
      if (lq == 0 && c < C) { pan[c] = sl; pan[C + c] = sb; }
    }
  }
}

// ------------------------------------------------------------------------------------------- register-resident cells
// Same cells for the common sizes (C <= 64, H <= 128): the generic kernels above stream every weight chunk from L2 inside
// the dependent MFMA chains (4 phases x ~10 chunk round trips per cell: ~54 % of a wave's life is s_waitcnt, rocprof
// PMC). Weights do not depend on the data, so here each wave issues the loads of ITS slice of a phase's weights one phase
// ahead, into registers (<= 136 VGPRs), from zero-padded images made by lfi_flow_prep, and the k loops run MFMA-paced
// from registers + LDS. Image layout = MFMA B-fragment order in blocks of 16 k: element (k, column) of a K x J operand
// sits at (((k / 16) * 4 + k % 4) * J16 + column) * 4 + (k / 4) % 4, so the four k-steps of a block are ONE 16-byte load
// per lane and a wave-load is four 256-byte segments (dword-per-lane loads spent 12k cycles per cell in issue alone,
// s_memtime stamps). K and J are padded to 16 with zeros: no bounds checks. Elementwise phases use a fixed
// (row = tid / 32, column = tid % 32 [+ 32]) thread map: no integer divisions, 128-byte row segments.
#define LFI_STAMP(slot)                                                                                  \
  do {                                                                                                   \
    if (f.stamps && tid == 0 && bt == 0) f.stamps[cell * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

constexpr int FB_C = 4;   // blocks of 16 k over C    <= 64
constexpr int FB_Z = 2;   //                 over Ch   <= 32
constexpr int FB_H = 8;   //                 over H    <= 128
constexpr int FB_O = 4;   //                 over Cout <= 64

__host__ __device__ inline bool flow_fast_ok(int C, int H, int Cout) { return C <= 64 && H <= 128 && Cout <= 64; }
__host__ __device__ inline long flow_img_index(int k, int col, int J) {
  return ((long)((k >> 4) * 4 + (k & 3)) * J + col) * 4 + ((k >> 2) & 3);
}

// Workgroup -> (cell, batch tile). Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share one), and every
// cell of a diagonal needs its own 270 KB of weights: give each XCD a contiguous run of (cell, tile) pairs so that a cell's
// 16 batch tiles (and the same flow step on the next diagonal) hit the same 4 MB L2 instead of all 8 L2s holding all 16
// steps' weights (4.3 MB: thrashing). Bijective for any grid size; speed only, never correctness.
__device__ __forceinline__ void flow_cell_of_block(int nbt, int* cell, int* bt) {
  const int total = gridDim.x * gridDim.y;
  int bid = blockIdx.x + gridDim.x * blockIdx.y;
  const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  *cell = bid / nbt;
  *bt = bid - *cell * nbt;
}

struct CarveF {
  int At, Ht, Zt, Hn, Yrm, Orm, total;
};
__host__ __device__ inline CarveF carve_fast_fwd(int C, int C16, int H16, int Ch16, int Cout) {
  CarveF c;
  int o = 0;
  c.At = o; o += C16 * LT;
  c.Ht = o; o += H16 * LT;
  c.Zt = o; o += Ch16 * LT;
  c.Hn = o; o += H16 * LT;
  c.Yrm = o; o += MB * (C + 1);
  c.Orm = o; o += MB * (Cout + 1);
  c.total = o;
  return c;
}

// sum over nb blocks of 16 k: A(16 x 16 nb) from LDS (k-major: a_lane = a_lds + kq * LT + l15, element k at + k * LT) times
// the register-resident B slice w[b] (components e: k = 16 b + 4 e + kq). Two interleaved chains (40-cycle dependent latency
// against a 32-cycle issue).
template <int MAXB>
__device__ __forceinline__ f32x4 mma16_reg(const float* a_lane, const f32x4 (&w)[MAXB], int nb) {
  f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
    if (b < nb) {
      const float* ab = a_lane + b * 16 * LT;
      const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
      e = mfma16(a0, w[b][0], e);
      o = mfma16(a1, w[b][1], o);
      e = mfma16(a2, w[b][2], e);
      o = mfma16(a3, w[b][3], o);
    }
  return e + o;
}

// this lane's slice of one 16-column tile of an image: nb float4 (k blocks), image row pitch J (columns, multiple of 16)
template <int MAXB>
__device__ __forceinline__ void load_frag(f32x4 (&w)[MAXB], const float* __restrict__ img, int J, int col, int kq, int nb,
                                          bool on) {
  const f32x4* p = reinterpret_cast<const f32x4*>(img) + (long)kq * J + col;
#pragma unroll
  for (int b = 0; b < MAXB; ++b)
    if (on && b < nb) w[b] = p[(long)b * 4 * J];
}

// ---- bf16 x 3 form of the recurrent products (persistent walk, engine_precision bf16x3). The f32-input MFMA the cells use
// everywhere else runs at 1/16 of the bf16 rate, and with flow step k's weights resident the recurrent cell's
// (z1, h) x (W_ih[:, :Ch], W_hh) product is what a pipeline step waits for (46 % of a forward step, tools/pipe_stamps.py).
// Same split as the GEMMs: x = hi + lo in bf16, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 into fp32 accumulators.
// No new weight images: two consecutive 16-k blocks of the f32 fragment registers (components e: k = 16 b + 4 e + kq) are
// split in registers once per launch into one 32-k bf16 fragment, slot i of lane group kq standing for
// k = 32 B + 16 (i >> 2) + 4 (i & 3) + kq - any bijection does as long as the A operand uses the same one, and this one
// makes the A side exactly the LDS reads the f32 path already does.
typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fbf16x2 __attribute__((ext_vector_type(2)));
typedef float ffloat2 __attribute__((ext_vector_type(2)));
struct X3Frag { fbf16x8 hi, lo; };
__device__ __forceinline__ void x3_split2(float a, float b, unsigned* hi, unsigned* lo) {
  const fbf16x2 h = __builtin_convertvector((ffloat2){a, b}, fbf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float ha = __builtin_bit_cast(float, hb << 16), hbv = __builtin_bit_cast(float, hb & 0xffff0000u);
  const fbf16x2 l = __builtin_convertvector((ffloat2){a - ha, b - hbv}, fbf16x2);
  *hi = hb;
  *lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ X3Frag x3_pack(const f32x4& b0, const f32x4& b1) {
  uint4 h, l;
  x3_split2(b0[0], b0[1], &h.x, &l.x);
  x3_split2(b0[2], b0[3], &h.y, &l.y);
  x3_split2(b1[0], b1[1], &h.z, &l.z);
  x3_split2(b1[2], b1[3], &h.w, &l.w);
  X3Frag r;
  r.hi = __builtin_bit_cast(fbf16x8, h);
  r.lo = __builtin_bit_cast(fbf16x8, l);
  return r;
}
// A fragment of 32 k from a k-major LDS operand: the eight reads of two f32 blocks
__device__ __forceinline__ X3Frag x3_a(const float* ab) {
  f32x4 b0 = {ab[0], ab[4 * LT], ab[8 * LT], ab[12 * LT]};
  const float* a1 = ab + 16 * LT;
  f32x4 b1 = {a1[0], a1[4 * LT], a1[8 * LT], a1[12 * LT]};
  return x3_pack(b0, b1);
}
__device__ __forceinline__ f32x4 x3_mma(const X3Frag& a, const X3Frag& w, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, w.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, w.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, w.hi, acc, 0, 0, 0);
  return acc;
}

// ---- the same with fp16 pieces (11 + 11 mantissa bits: 2^-22 relative, fp32-grade) for the SAMPLER's reverse cells. Their
// operands - h in (-1, 1), flow activations, trained weights - sit far inside fp16's range (a value beyond 65504 turns into
// inf - inf = NaN: loud, as the exact path is at 3e38; tiny values lose nothing that matters: fp16's subnormal spacing, 6e-8,
// is the absolute error of an fp32 near 1). Gradients do not qualify (1e-10 underflows), so every backward product and the
// training walks keep bf16 pieces. Same MFMA rate, same register footprint as bf16 x 3.
typedef _Float16 fh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fh16x2 __attribute__((ext_vector_type(2)));
struct X3FragH { fh16x8 hi, lo; };
__device__ __forceinline__ void x3h_split2(float a, float b, unsigned* hi, unsigned* lo) {
  const fh16x2 h = __builtin_convertvector((ffloat2){a, b}, fh16x2);
  const ffloat2 hf = __builtin_convertvector(h, ffloat2);
  const fh16x2 l = __builtin_convertvector((ffloat2){a - hf[0], b - hf[1]}, fh16x2);
  *hi = __builtin_bit_cast(unsigned, h);
  *lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ X3FragH x3h_pack(const f32x4& b0, const f32x4& b1) {
  uint4 h, l;
  x3h_split2(b0[0], b0[1], &h.x, &l.x);
  x3h_split2(b0[2], b0[3], &h.y, &l.y);
  x3h_split2(b1[0], b1[1], &h.z, &l.z);
  x3h_split2(b1[2], b1[3], &h.w, &l.w);
  X3FragH r;
  r.hi = __builtin_bit_cast(fh16x8, h);
  r.lo = __builtin_bit_cast(fh16x8, l);
  return r;
}
__device__ __forceinline__ X3FragH x3h_a(const float* ab) {
  f32x4 b0 = {ab[0], ab[4 * LT], ab[8 * LT], ab[12 * LT]};
  const float* a1 = ab + 16 * LT;
  f32x4 b1 = {a1[0], a1[4 * LT], a1[8 * LT], a1[12 * LT]};
  return x3h_pack(b0, b1);
}
__device__ __forceinline__ f32x4 x3h_mma(const X3FragH& a, const X3FragH& w, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, w.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, w.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, w.hi, acc, 0, 0, 0);
  return acc;
}

// this lane's fragments of one 16-column tile of a pre-split backward image: nb2 32-k blocks of gate g
template <int MAXB2>
__device__ __forceinline__ void x3_load(X3Frag (&w)[MAXB2], const uint4* __restrict__ img, long per, int g, int nB, int J, int col,
                                        int kq, int nb2, bool on) {
  const uint4* p = img + (((long)g * nB) * J + col) * 4 + kq;
#pragma unroll
  for (int b = 0; b < MAXB2; ++b)
    if (on && b < nb2) {
      w[b].hi = __builtin_bit_cast(fbf16x8, p[(long)b * J * 4]);
      w[b].lo = __builtin_bit_cast(fbf16x8, p[(long)b * J * 4 + per]);
    }
}
// sum over NG gate blocks of nb2 32-k blocks each (A: k-major LDS operand, gate stride blk floats)
template <int NG, int MAXB2>
__device__ __forceinline__ f32x4 x3_mma_gates(const float* a_lane, int blk, const X3Frag (&w)[NG][MAXB2], int nb2) {
  f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int b = 0; b < MAXB2; ++b)
      if (b < nb2) {
        const X3Frag a = x3_a(a_lane + g * blk + b * 32 * LT);
        if ((g * MAXB2 + b) & 1) o = x3_mma(a, w[g][b], o);
        else e = x3_mma(a, w[g][b], e);
      }
  return e + o;
}

// The backward cell's MFMA operands d(gate pre-activations) are needed by all eight waves: instead of every wave splitting the
// same fp32 LDS values again (24 blocks x ~30 VALU per wave and timestep - it bound Q2 once the MFMAs were bf16), the wave
// that computes a value stores its bf16 hi and lo ONCE, row-major [16 rows][NG * H16 + 8], the column of hidden unit j of
// gate g at g * H16 + x3_pos(j): within a 32-k block the slot order of x3_a, so a lane's 8 k are one 16-byte read.
__device__ __forceinline__ int x3_pos(int j) { return (j & ~31) | ((j & 3) << 3) | ((j >> 2) & 7); }
__device__ __forceinline__ void x3_put(__bf16* hi_img, __bf16* lo_img, int idx, float v) {
  const __bf16 h = (__bf16)v;
  hi_img[idx] = h;
  lo_img[idx] = (__bf16)(v - (float)h);
}
template <int NG, int MAXB2>
__device__ __forceinline__ f32x4 x3_mma_gates_img(const __bf16* hi_row, const __bf16* lo_row, int H16, const X3Frag (&w)[NG][MAXB2],
                                                  int nb2) {
  f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int b = 0; b < MAXB2; ++b)
      if (b < nb2) {
        X3Frag a;
        a.hi = *reinterpret_cast<const fbf16x8*>(hi_row + g * H16 + b * 32);
        a.lo = *reinterpret_cast<const fbf16x8*>(lo_row + g * H16 + b * 32);
        if ((g * MAXB2 + b) & 1) o = x3_mma(a, w[g][b], o);
        else e = x3_mma(a, w[g][b], e);
      }
  return e + o;
}

// P2 of a register-resident cell: the coupling net's recurrent cell on this wave's 16 hidden units. Zt / Ht: z1 and
// h_prev in LDS (k-major), Hn: new state (LDS), h_out / c_out / g_out: row-0 pointers of the (rows x H) / (rows x 4H) outputs
// (g_out may be null).
// gate math + stores of P2 on this wave's 16 hidden units, given the two accumulated products (az: z1 side, ah: h side)
template <int NG>
__device__ __forceinline__ void fast_cell_p2_gates(const FlowK& f, const float* Ht, float* Hn, const f32x4 (&az)[NG],
                                                   const f32x4 (&ah)[NG], const float (&gc)[4][NG], const float (&bh)[NG],
                                                   const float (&cprev)[4], int j2, int kq, int b0, int rows, float* h_out,
                                                   float* c_out, float* g_out, float* cnew, __bf16* img_hi = nullptr,
                                                   __bf16* img_lo = nullptr, int img_ld = 0, int img_col = 0) {
  const int H = f.H;
  if (j2 < H) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = kq * 4 + r;
      const int row = b0 + i;
      float hnew;
      float gs0, gs1, gs2, gs3;
      if (NG == 3) {  // torch.nn.GRUCell, gate order r, z, n
        const float rr = sigmoidf_(az[0][r] + ah[0][r] + gc[r][0] + bh[0]);
        const float uu = sigmoidf_(az[1][r] + ah[1][r] + gc[r][1] + bh[1]);
        const float ghn = ah[2][r] + bh[2];
        // (explicit fused forms: left to -ffp-contract, "(1 - z) n + z h" fuses either product, and which one depended on the
        // kernel this function was inlined into - the persistent walk and the diagonal walk then differed by an ulp)
        const float nn = tanhf_(__builtin_fmaf(rr, ghn, az[2][r] + gc[r][2]));
        const float hp = Ht[j2 * LT + i];
        hnew = __builtin_fmaf(uu, hp, (1.0f - uu) * nn);
        gs0 = rr; gs1 = uu; gs2 = nn; gs3 = ghn;
      } else {        // torch.nn.LSTMCell, gate order i, f, g, o; zero (h, c) at the first modelled frame
        const float ii = sigmoidf_(az[0][r] + ah[0][r] + gc[r][0] + bh[0]);
        const float ff = sigmoidf_(az[1][r] + ah[1][r] + gc[r][1] + bh[1]);
        const float gg = tanhf_(az[2][r] + ah[2][r] + gc[r][2] + bh[2]);
        const float oo = sigmoidf_(az[NG - 1][r] + ah[NG - 1][r] + gc[r][NG - 1] + bh[NG - 1]);
        const float c2 = __builtin_fmaf(ff, cprev[r], ii * gg);
        hnew = oo * tanhf_(c2);
        if (row < rows) c_out[(long)row * H + j2] = c2;
        if (cnew) cnew[r] = c2;
        gs0 = ii; gs1 = ff; gs2 = gg; gs3 = oo;
      }
      Hn[j2 * LT + i] = hnew;
      if (img_hi) x3_put(img_hi, img_lo, i * img_ld + img_col + x3_pos(j2), hnew);   // bf16 hi / lo image for the next cell's product
      if (row < rows) {
        if (h_out) h_out[(long)row * H + j2] = hnew;   // (null: the caller stores the tile's rows itself, 16 bytes at a time)
        if (g_out) {
          // the four stashed gate values of (row, hidden unit) lie together: ONE 16-byte store here and one 16-byte load in the
          // backward cell instead of four dword accesses each (the walks are bound by vector-memory instruction issue:
          // without the P2 stash stores the forward walk ran 11 % faster)
          *reinterpret_cast<f32x4*>(g_out + (long)row * 4 * H + 4 * j2) = (f32x4){gs0, gs1, gs2, gs3};
        }
      }
    }
  }
}

template <int NG>
__device__ __forceinline__ void fast_cell_p2(const FlowK& f, const float* Zt, const float* Ht, float* Hn,
                                             const f32x4 (&wz)[NG][FB_Z], const f32x4 (&wh)[NG][FB_H], const float (&gc)[4][NG],
                                             const float (&bh)[NG], const float (&cprev)[4], int nbZ, int nbH, int j2, int kq,
                                             int l15, int b0, int rows, float* h_out, float* c_out, float* g_out,
                                             float* cnew = nullptr) {
  f32x4 az[NG], ah[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    az[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float* zl = Zt + kq * LT + l15;
  const float* hl = Ht + kq * LT + l15;
#pragma unroll
  for (int b = 0; b < FB_Z; ++b)
    if (b < nbZ) {
      const float* ab = zl + b * 16 * LT;
      const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = mfma16(a0, wz[g][b][0], az[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = mfma16(a1, wz[g][b][1], az[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = mfma16(a2, wz[g][b][2], az[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = mfma16(a3, wz[g][b][3], az[g]);
    }
#pragma unroll
  for (int b = 0; b < FB_H; ++b)
    if (b < nbH) {
      const float* ab = hl + b * 16 * LT;
      const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = mfma16(a0, wh[g][b][0], ah[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = mfma16(a1, wh[g][b][1], ah[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = mfma16(a2, wh[g][b][2], ah[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = mfma16(a3, wh[g][b][3], ah[g]);
    }
  fast_cell_p2_gates<NG>(f, Ht, Hn, az, ah, gc, bh, cprev, j2, kq, b0, rows, h_out, c_out, g_out, cnew);
}

// The same cell with its A operand (z1 | h_{t-1}) read from bf16 hi / lo LDS images the PRODUCERS wrote (P1 for z1, the previous
// timestep's gate epilogue for h: x3_put, slot order x3_pos): one 16-byte read per 32-k block and plane instead of eight
// 4-byte reads of the k-major fp32 images plus a split redone by all eight waves (stamps: 2.8 k of the 5.8 k cycles of P2).
template <int NG>
__device__ __forceinline__ void fast_cell_p2_x3_img(const FlowK& f, const __bf16* ih, const __bf16* il, int ldx, int Ch16,
                                                    const float* Ht, float* Hn, const X3Frag (&wz)[NG][FB_Z / 2],
                                                    const X3Frag (&wh)[NG][FB_H / 2], const float (&gc)[4][NG],
                                                    const float (&bh)[NG], const float (&cprev)[4], int nbZ2, int nbH2, int j2,
                                                    int kq, int l15, int b0, int rows, float* h_out, float* c_out, float* g_out,
                                                    float* cnew, __bf16* ihn, __bf16* iln) {
  f32x4 az[NG], ah[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    az[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const __bf16* rh = ih + l15 * ldx + 8 * kq;
  const __bf16* rl = il + l15 * ldx + 8 * kq;
#pragma unroll
  for (int b = 0; b < FB_Z / 2; ++b)
    if (b < nbZ2) {
      X3Frag a;
      a.hi = *reinterpret_cast<const fbf16x8*>(rh + b * 32);
      a.lo = *reinterpret_cast<const fbf16x8*>(rl + b * 32);
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = x3_mma(a, wz[g][b], az[g]);
    }
#pragma unroll
  for (int b = 0; b < FB_H / 2; ++b)
    if (b < nbH2) {
      X3Frag a;
      a.hi = *reinterpret_cast<const fbf16x8*>(rh + Ch16 + b * 32);
      a.lo = *reinterpret_cast<const fbf16x8*>(rl + Ch16 + b * 32);
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = x3_mma(a, wh[g][b], ah[g]);
    }
  fast_cell_p2_gates<NG>(f, Ht, Hn, az, ah, gc, bh, cprev, j2, kq, b0, rows, h_out, c_out, g_out, cnew, ihn, iln, ldx, Ch16);
}

// bf16 x 3 form: weights as packed 32-k fragments (x3_pack), nbZ2 / nbH2 = number of 32-k blocks
template <int NG>
__device__ __forceinline__ void fast_cell_p2_x3(const FlowK& f, const float* Zt, const float* Ht, float* Hn,
                                                const X3Frag (&wz)[NG][FB_Z / 2], const X3Frag (&wh)[NG][FB_H / 2],
                                                const float (&gc)[4][NG], const float (&bh)[NG], const float (&cprev)[4], int nbZ2,
                                                int nbH2, int j2, int kq, int l15, int b0, int rows, float* h_out, float* c_out,
                                                float* g_out, float* cnew) {
  f32x4 az[NG], ah[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    az[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float* zl = Zt + kq * LT + l15;
  const float* hl = Ht + kq * LT + l15;
#pragma unroll
  for (int b = 0; b < FB_Z / 2; ++b)
    if (b < nbZ2) {
      const X3Frag a = x3_a(zl + b * 32 * LT);
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = x3_mma(a, wz[g][b], az[g]);
    }
#pragma unroll
  for (int b = 0; b < FB_H / 2; ++b)
    if (b < nbH2) {
      const X3Frag a = x3_a(hl + b * 32 * LT);
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = x3_mma(a, wh[g][b], ah[g]);
    }
  fast_cell_p2_gates<NG>(f, Ht, Hn, az, ah, gc, bh, cprev, j2, kq, b0, rows, h_out, c_out, g_out, cnew);
}

// fp16 x 3 form (x3h_*): the sampler's reverse cells
template <int NG>
__device__ __forceinline__ void fast_cell_p2_x3h(const FlowK& f, const float* Zt, const float* Ht, float* Hn,
                                                 const X3FragH (&wz)[NG][FB_Z / 2], const X3FragH (&wh)[NG][FB_H / 2],
                                                 const float (&gc)[4][NG], const float (&bh)[NG], const float (&cprev)[4], int nbZ2,
                                                 int nbH2, int j2, int kq, int l15, int b0, int rows, float* h_out, float* c_out,
                                                 float* g_out, float* cnew) {
  f32x4 az[NG], ah[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    az[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float* zl = Zt + kq * LT + l15;
  const float* hl = Ht + kq * LT + l15;
#pragma unroll
  for (int b = 0; b < FB_Z / 2; ++b)
    if (b < nbZ2) {
      const X3FragH a = x3h_a(zl + b * 32 * LT);
#pragma unroll
      for (int g = 0; g < NG; ++g) az[g] = x3h_mma(a, wz[g][b], az[g]);
    }
#pragma unroll
  for (int b = 0; b < FB_H / 2; ++b)
    if (b < nbH2) {
      const X3FragH a = x3h_a(hl + b * 32 * LT);
#pragma unroll
      for (int g = 0; g < NG; ++g) ah[g] = x3h_mma(a, wh[g][b], ah[g]);
    }
  fast_cell_p2_gates<NG>(f, Ht, Hn, az, ah, gc, bh, cprev, j2, kq, b0, rows, h_out, c_out, g_out, cnew);
}

// P3: o = (h' Wfl^T + b) exp(3 logs) on this wave's 16 outputs   (LinearZeros, glow/modules.py:93-95); o_out may be null
// (bb, sc: LinearZeros bias and exp(3 logs) of this lane's output column, loaded by the caller OUTSIDE its dependent phases)
__device__ __forceinline__ void fast_cell_p3(const FlowK& f, int k, const float* Hn, float* Orm, const f32x4 (&w3)[FB_H], int nbH,
                                             int col, int kq, int l15, int b0, int rows, float* o_out, long ld_out, float bb, float sc) {
  const int Cout = f.Cout, ldo = Cout + 1;
  const f32x4 acc = mma16_reg<FB_H>(Hn + kq * LT + l15, w3, nbH);
  if (col < Cout) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = kq * 4 + r;
      const int row = b0 + i;
      const float o = (acc[r] + bb) * sc;
      Orm[i * ldo + col] = o;
      if (o_out && row < rows) o_out[(long)row * ld_out + col] = o;
    }
  }
}

template <int NG>
__global__ __launch_bounds__(NT) void flow_diag_fwd_fast_kernel(FlowK f, int d, int klo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int ri = tid >> 5, cl = tid & 31;  // elementwise thread map
  int cell, bt;
  flow_cell_of_block(f.nbt, &cell, &bt);
  const int k = klo + cell, n = d - k;
  const int b0 = bt * MB;
  const int B = f.B, C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, G = f.G;
  const int C16 = f.C16, Ch16 = f.Ch16, H16 = f.H16, Co16 = f.Co16;
  const long LC = f.ldc, LO = f.ldo;   // stash row strides
  const long fr = (long)n * B;
  const long kf = (long)k * f.F + fr;
  const CarveF cv = carve_fast_fwd(C, C16, H16, Ch16, Cout);
  float* At = flow_smem + cv.At;
  float* Ht = flow_smem + cv.Ht;
  float* Zt = flow_smem + cv.Zt;
  float* Hn = flow_smem + cv.Hn;
  float* Yrm = flow_smem + cv.Yrm;
  float* Orm = flow_smem + cv.Orm;
  const int ldy = C + 1, ldo = Cout + 1;
  const int nbC = C16 >> 4, nbZ = Ch16 >> 4, nbH = H16 >> 4;
  LFI_STAMP(0);

  // ---- weights of P1 (this wave's 16 output channels of W) and P2 (its 16 hidden units, NG gates), gic/bias of its rows
  const bool t1 = wave * 16 < C, t2 = wave * 16 < H, t3 = wave * 16 < Cout;
  const int tcol = wave * 16 + l15;
  f32x4 w1[FB_C], wz[NG][FB_Z], wh[NG][FB_H];
  load_frag<FB_C>(w1, f.pW + (long)k * C16 * C16, C16, tcol, kq, nbC, t1);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    load_frag<FB_Z>(wz[g], f.pwz + (long)k * Ch16 * NG * H16, NG * H16, g * H16 + tcol, kq, nbZ, t2);
    load_frag<FB_H>(wh[g], f.pwh + (long)k * H16 * NG * H16, NG * H16, g * H16 + tcol, kq, nbH, t2);
  }
  const int j2 = tcol;                    // hidden unit of this lane in P2
  const bool j2ok = j2 < H;
  float gc[4][NG], bh[NG], cprev[4];
  {
    const float* gicb = f.gic + kf * G;
    const float* bhh = f.p.b_hh + (long)k * G;
    const float* cpb = (NG == 4 && n > 0) ? f.sC + (kf - B) * H : nullptr;
    const int jc = j2ok ? j2 : 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = bhh[g * H + jc];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(b0 + kq * 4 + r, B - 1);
#pragma unroll
      for (int g = 0; g < NG; ++g) gc[r][g] = gicb[(long)row * G + g * H + jc];
      cprev[r] = cpb ? cpb[(long)row * H + jc] : 0.0f;
    }
  }
  LFI_STAMP(1);

  // ---- P0: actnorm (glow/modules.py:45-52); stage a, h_prev k-major; zero the k padding
  {
    const int row = b0 + ri;
    const bool rok = row < B;
    const float* xin = (k == 0) ? f.x0 + ((long)row * f.T + f.start + n) * C : f.sX + (kf - f.F + row) * LC;
    const float* anb = f.p.an_bias + (long)k * C;
    const float* anl = f.p.an_logs + (long)k * C;
    for (int c = cl; c < C16; c += 32) {
      float a = 0.0f;
      if (c < C && rok) {
        a = (xin[c] + anb[c]) * expf(anl[c]);
        f.sA[(kf + row) * LC + c] = a;
      }
      At[c * LT + ri] = a;
    }
    const float* hp = n > 0 ? f.sH + (kf - B + row) * H : nullptr;
    for (int j = cl; j < H16; j += 32) {
      Ht[j * LT + ri] = (hp && rok && j < H) ? hp[j] : 0.0f;
      if (j >= H) Hn[j * LT + ri] = 0.0f;
    }
    for (int c = Ch + cl; c < Ch16; c += 32) Zt[c * LT + ri] = 0.0f;
  }
  __syncthreads();
  LFI_STAMP(2);

  // ---- P1: y = a W   (InvertibleConv1x1.forward, glow/modules.py:186)
  if (t1) {
    const f32x4 acc = mma16_reg<FB_C>(At + kq * LT + l15, w1, nbC);
    const int c = tcol;
    if (c < C) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq * 4 + r;
        const int row = b0 + i;
        const float v = acc[r];
        Yrm[i * ldy + c] = v;
        if (c < Ch) Zt[c * LT + i] = v;
        if (row < B) f.sY[(kf + row) * LC + c] = v;
      }
    }
  }
  // weights of P3 (this wave's 16 outputs of LinearZeros): in flight under P2
  f32x4 w3[FB_H];
  load_frag<FB_H>(w3, f.pwfl + (long)k * H16 * Co16, Co16, tcol, kq, nbH, t3);
  __syncthreads();
  LFI_STAMP(3);

  // ---- P2: recurrent cell of the coupling net (f_seq.forward, glow/models.py:204-214)
  if (t2) fast_cell_p2<NG>(f, Zt, Ht, Hn, wz, wh, gc, bh, cprev, nbZ, nbH, tcol, kq, l15, b0, B,
                           f.sH + kf * H, NG == 4 ? f.sC + kf * H : nullptr, f.sG + kf * 4 * H);
  __syncthreads();
  LFI_STAMP(4);

  // ---- P3: o = (h' Wfl^T + b) exp(3 logs)   (LinearZeros, glow/modules.py:93-95)
  if (t3) {
    const int cj = tcol < Cout ? tcol : 0;
    fast_cell_p3(f, k, Hn, Orm, w3, nbH, tcol, kq, l15, b0, B, f.sO + kf * LO, LO, f.p.b_fl[(long)k * Cout + cj],
                 expf(3.0f * f.p.l_fl[(long)k * Cout + cj]));
  }
  __syncthreads();
  LFI_STAMP(5);

  // ---- P4: coupling (glow/models.py:330-341), pass-through half, log-det of the coupling (wavefront shuffle sum)
  {
    const int row = b0 + ri;
    const bool rok = row < B;
    float lg = 0.0f;
    if (cl < C2) {
      const float z2 = Yrm[ri * ldy + Ch + cl];
      float z2n;
      if (f.affine) {
        const float shift = Orm[ri * ldo + 2 * cl];
        const float sraw = sigmoidf_(Orm[ri * ldo + 2 * cl + 1] + 2.0f);
        const float sc = fmaxf(sraw, f.eps);
        z2n = (z2 + shift) * sc;
        lg = logf(sc);
      } else {
        z2n = z2 + Orm[ri * ldo + cl];
      }
      if (rok) f.sX[(kf + row) * LC + Ch + cl] = z2n;
    }
    if (cl < Ch && rok) f.sX[(kf + row) * LC + cl] = Yrm[ri * ldy + cl];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);  // the 32 lanes of one row
    if (cl == 0 && rok) f.sL[kf + row] = lg;
  }
  LFI_STAMP(6);
}

// ------------------------------------------------------------------------------------------- persistent pipeline
// The diagonal walk above pays one launch + one reload of 270 KB of weights per workgroup for every one of the N + Ks - 1
// diagonals, although a workgroup's weights never change: cell (n, k) of batch tile bt always needs flow step k's. Here
// workgroup (k, bt) is PERSISTENT: it loads step k's weights into registers once, then walks n = 0 .. N-1 for its 16
// samples; the recurrent state h (and the LSTM cell state) never leaves the workgroup (LDS / registers), and the only
// inter-workgroup traffic is the 16 x C output tile handed from (k, bt) to (k + 1, bt): a systolic pipeline over the flow
// steps, N + Ks - 1 cell times end to end, one launch. Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, form R1):
// the producer stores the tile write-through (sc1), every wave drains its stores, workgroup barrier, ONE lane publishes
// the progress counter with an agent-scope atomic store; the consumer polls that one word relaxed, ONE agent-scope acquire,
// barrier, then plain loads. Deadlock-free for ANY grid size and dispatch order: logical (k, bt) ids are dealt by an atomic
// ticket in arrival order and a workgroup only ever waits on a smaller ticket, i.e. on a workgroup that is already
// running (more workgroups than CUs simply run as successive groups of flow steps). Every spin is bounded: on timeout
// the abort word is set, every workgroup leaves its loop, and the host reports LFI_ERR_LAUNCH.
constexpr unsigned PIPE_HDR = 4;                 // ticket, abort, 2 reserved words
#ifndef LFI_PIPE_STRIDE
#define LFI_PIPE_STRIDE 32
#endif
constexpr unsigned PIPE_STRIDE = LFI_PIPE_STRIDE;   // words between two progress words of the persistent walks: one 128-byte line each (the polls of 256 workgroups
                                                    // on eight shared lines queued at one memory channel)
constexpr unsigned PIPE_WALK_HDR = PIPE_STRIDE > PIPE_HDR ? PIPE_STRIDE : PIPE_HDR;
constexpr unsigned PIPE_SPIN_LIMIT = 1u << 23;   // polls (each >= ~0.5 us) before giving up

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through (sc1) store of one payload element
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// load of one handed-off payload element: L1-bypassing (sc1) when the consumer did not fence
__device__ __forceinline__ float ld_tile(const float* p, bool fenced) {
  return fenced ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ONE lane: wait until *flag >= need. false = aborted (timeout here or in another workgroup).
__device__ __forceinline__ bool pipe_wait(const unsigned* flag, unsigned need, unsigned* abort_w) {
  unsigned spins = 0;
  while (ld_agent(flag) < need) {
    if ((++spins & 31u) == 0u) {
      if (ld_agent(abort_w) != 0u) return false;
      if (spins > PIPE_SPIN_LIMIT) {
        st_agent(abort_w, 1u);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(4);
  }
  return true;
}
// consumer side of a hand-off, all threads: thread 0 polls + acquires, the rest learn the outcome through LDS
__device__ __forceinline__ bool pipe_acquire(const unsigned* flag, unsigned need, unsigned* abort_w, int tid, int* s_ok,
                                             bool fence) {
  if (tid == 0) {
    const bool ok = pipe_wait(flag, need, abort_w);
    if (ok && fence) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}
// producer side, all threads: drain this wave's stores, barrier, one lane publishes
__device__ __forceinline__ void pipe_publish(unsigned* flag, unsigned value, int tid, bool signal) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (signal && tid == 0) st_agent(flag, value);
}

// as mma16_reg with the B fragments of this wave in LDS: wl[b * 64] is this lane's float4 of k block b (consecutive lanes
// read consecutive 16 bytes: conflict-free ds_read_b128)
__device__ __forceinline__ f32x4 mma16_lds(const float* a_lane, const f32x4* wl, int nb) {
  f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nb; ++b) {
    const float* ab = a_lane + b * 16 * LT;
    const f32x4 w = wl[b * 64];
    const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
    e = mfma16(a0, w[0], e);
    o = mfma16(a1, w[1], o);
    e = mfma16(a2, w[2], e);
    o = mfma16(a3, w[3], o);
  }
  return e + o;
}
// LDS floats of the pipeline kernel: the cell's operands, then the W fragments of the C16/16 P1 waves and the Wfl fragments
// of the Co16/16 P3 waves (the recurrent weights W_ih[:, :Ch] and W_hh stay in registers: 120 VGPRs at H = 128)
__host__ __device__ inline int pipe_fwd_img_offset(int C, int C16, int H16, int Ch16, int Cout, int Co16) {
  const int base = (carve_fast_fwd(C, C16, H16, Ch16, Cout).total + 3) & ~3;
  return base + (C16 >> 4) * (C16 >> 4) * 256 + (Co16 >> 4) * (H16 >> 4) * 256;
}
// + the bf16 x 3 cell's operand images: two buffers (h of the previous / of this timestep) x {hi, lo} x MB rows of
// Ch16 + H16 + 8 bf16 (2 MB ldx floats)
__host__ __device__ inline int pipe_fwd_lds_floats(int C, int C16, int H16, int Ch16, int Cout, int Co16) {
  return pipe_fwd_img_offset(C, C16, H16, Ch16, Cout, Co16) + 2 * MB * (Ch16 + H16 + 8);
}

// diagnostics (lfi_debug_set_stamps): s_memtime of workgroup (Ks / 2, tile 0) at the phase boundaries of every timestep, in
// slots [4096 + 2048 * backward + 16 * n + phase] of the stamp buffer
#define PIPE_STAMP(dir, slot)                                                                                              \
  do {                                                                                                                     \
    if (f.stamps && tid == 0 && bt == 0 && k == f.stamp_k && n < 128)                                                       \
      f.stamps[4096 + 2048 * (dir) + 16 * n + (slot)] = __builtin_amdgcn_s_memtime();                                      \
  } while (0)

template <int NG, bool X3>
__global__ __launch_bounds__(NT) void flow_pipe_fwd_kernel(FlowK f) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15_0 = lane & 15, kq_0 = lane >> 4;
  const int ri_0 = tid >> 5, cl_0 = tid & 31;  // elementwise thread map
  __shared__ int s_id, s_ok, s_rdy;
  if (tid == 0) s_id = (int)atomicAdd(f.pipe, 1u);
  __syncthreads();
  const int id = s_id;
  const int nbt = f.nbt;
  const int k = id / nbt, bt = id - k * nbt;   // tickets in arrival order: (k - 1, bt) always started before (k, bt)
  if (k >= f.Ks) return;
  unsigned* abort_w = f.pipe + 1;
  unsigned* prog = f.pipe + PIPE_WALK_HDR;
  const bool fenced = f.pipe_fence != 0;
  const int b0 = bt * MB;
  const int B = f.B, C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, G = f.G;
  const int C16 = f.C16, Ch16 = f.Ch16, H16 = f.H16, Co16 = f.Co16;
  const long LC = f.ldc, LO = f.ldo;
  const CarveF cv = carve_fast_fwd(C, C16, H16, Ch16, Cout);
  float* At = flow_smem + cv.At;
  float* Ht = flow_smem + cv.Ht;   // h of the previous timestep
  float* Zt = flow_smem + cv.Zt;
  float* Hn = flow_smem + cv.Hn;   // h of this timestep (the two swap every iteration)
  float* Yrm = flow_smem + cv.Yrm;
  float* Orm = flow_smem + cv.Orm;
  const int ldy = C + 1, ldo = Cout + 1;
  const int nbC = C16 >> 4, nbZ = Ch16 >> 4, nbH = H16 >> 4;

  // ---- this workgroup's weights, once: P1 (16 output channels of W), P2 (16 hidden units, NG gates), P3 (16 outputs)
  const bool t1 = wave * 16 < C, t2 = wave * 16 < H, t3 = wave * 16 < Cout;
  const int tcol_0 = wave * 16 + l15_0;
  f32x4 wz[X3 ? 1 : NG][FB_Z], wh[X3 ? 1 : NG][FB_H];          // exact f32 form
  X3Frag wzx[X3 ? NG : 1][FB_Z / 2], whx[X3 ? NG : 1][FB_H / 2];  // bf16 x 3 form: the same fragments, split once
  {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f32x4 tz[FB_Z], th[FB_H];
#pragma unroll
      for (int b = 0; b < FB_Z; ++b) tz[b] = zero4;
#pragma unroll
      for (int b = 0; b < FB_H; ++b) th[b] = zero4;
      load_frag<FB_Z>(tz, f.pwz + (long)k * Ch16 * NG * H16, NG * H16, g * H16 + tcol_0, kq_0, nbZ, t2);
      load_frag<FB_H>(th, f.pwh + (long)k * H16 * NG * H16, NG * H16, g * H16 + tcol_0, kq_0, nbH, t2);
      if (X3) {
#pragma unroll
        for (int b = 0; b < FB_Z / 2; ++b) wzx[g][b] = x3_pack(tz[2 * b], tz[2 * b + 1]);
#pragma unroll
        for (int b = 0; b < FB_H / 2; ++b) whx[g][b] = x3_pack(th[2 * b], th[2 * b + 1]);
      } else {
#pragma unroll
        for (int b = 0; b < FB_Z; ++b) wz[X3 ? 0 : g][b] = tz[b];
#pragma unroll
        for (int b = 0; b < FB_H; ++b) wh[X3 ? 0 : g][b] = th[b];
      }
    }
  }
  // P1 / P3 weights: each owning wave parks its fragments in LDS (its own region, its own lanes: no barrier needed)
  f32x4* w1s = reinterpret_cast<f32x4*>(flow_smem + ((cv.total + 3) & ~3)) + wave * nbC * 64 + lane;
  f32x4* w3s = reinterpret_cast<f32x4*>(flow_smem + ((cv.total + 3) & ~3)) + nbC * nbC * 64 + wave * nbH * 64 + lane;
  if (t1) {
    const f32x4* p = reinterpret_cast<const f32x4*>(f.pW + (long)k * C16 * C16) + (long)kq_0 * C16 + tcol_0;
    for (int b = 0; b < nbC; ++b) w1s[b * 64] = p[(long)b * 4 * C16];
  }
  if (t3) {
    const f32x4* p = reinterpret_cast<const f32x4*>(f.pwfl + (long)k * H16 * Co16) + (long)kq_0 * Co16 + tcol_0;
    for (int b = 0; b < nbH; ++b) w3s[b * 64] = p[(long)b * 4 * Co16];
  }
  // bf16 x 3: operand images of the recurrent cell (fast_cell_p2_x3_img)
  const int ldxi = Ch16 + H16 + 8;
  __bf16* imgb = reinterpret_cast<__bf16*>(flow_smem + pipe_fwd_img_offset(C, C16, H16, Ch16, Cout, Co16));
  __bf16* ich = imgb;                       // current buffer: z1 of this timestep | h of the previous one (hi; lo at + MB * ldxi)
  __bf16* inh = imgb + 2 * MB * ldxi;       // next buffer: receives h of this timestep
  if (X3)
    for (int q = tid; q < 4 * MB * ldxi; q += NT) imgb[q] = (__bf16)0.0f;
  const int j2 = tcol_0;
  const int jc = j2 < H ? j2 : 0;
  float bh[NG], cprev[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const float* bhh = f.p.b_hh + (long)k * G;
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = bhh[g * H + jc];
  }
  const float flb = tcol_0 < Cout ? f.p.b_fl[(long)k * Cout + tcol_0] : 0.0f;              // LinearZeros bias and scale
  const float fls = tcol_0 < Cout ? expf(3.0f * f.p.l_fl[(long)k * Cout + tcol_0]) : 0.0f;
  // actnorm of this flow step for the (at most two) channels this thread handles in P0
  float anb[2], ans[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = cl_0 + 32 * u;
    anb[u] = c < C ? f.p.an_bias[(long)k * C + c] : 0.0f;
    ans[u] = c < C ? expf(f.p.an_logs[(long)k * C + c]) : 0.0f;
  }
  // zero state and the k padding of the LDS operands (never written again)
  for (int j = cl_0; j < H16; j += 32) {
    Ht[j * LT + ri_0] = 0.0f;
    Hn[j * LT + ri_0] = 0.0f;
  }
  for (int c = Ch + cl_0; c < Ch16; c += 32) Zt[c * LT + ri_0] = 0.0f;
  __syncthreads();

  const int row = b0 + ri_0;          // elementwise phases: this thread's sample
  const bool rok = row < B;
  // Look-ahead acquire. Every cell paid two dependent L2 round trips at its top - the poll of the predecessor's progress
  // word, then (behind a barrier) the sc1 loads of the tile: 3.6 k of a 20 k-cycle timestep (stamps: 0.5 k in step 0, which
  // has no predecessor, 4.1 k in step 1). Now thread 0 peeks at the word for cell n + 1 at the start of P4 of cell n (the
  // load returns under P4 and the store drain); if that cell is already published, every thread fetches its elements of
  // the tile right behind the publish barrier (poll -> barrier -> sc1 loads, the order of the blocking form) and the next
  // iteration starts with them in flight instead of polling. A step settles ~0.15 cell + one hand-off behind its
  // predecessor, where the peek succeeds every time; a miss just takes the blocking path.
  bool have_next = false;
  float xnext[2] = {0.0f, 0.0f};
  // conditioning part of the gates (written by the GEMM before this launch): cell 0's here, cell n + 1's behind P2 of cell n,
  // into the registers P2 has just finished with (at the top of the cell these loads sat in front of the tile's in the
  // in-order vmcnt queue: -2 % on the walk)
  float gc[4][NG];
  {
    const float* gicb = f.gic + (long)k * f.F * G;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rw = min(b0 + kq_0 * 4 + r, B - 1);
#pragma unroll
      for (int g = 0; g < NG; ++g) gc[r][g] = gicb[(long)rw * G + g * H + jc];
    }
  }
  for (int n = 0; n < f.N; ++n) {
    // lane coordinates laundered per iteration: otherwise every per-lane stash address (a dozen arrays x 4 rows, 64-bit) is
    // hoisted out of the timestep loop and the kernel spills ~150 VGPRs
    int l15 = l15_0, kq = kq_0, ri = ri_0, cl = cl_0, tcol = tcol_0;
    asm volatile("" : "+v"(l15), "+v"(kq), "+v"(ri), "+v"(cl), "+v"(tcol));

    const long fr = (long)n * B;
    const long kf = (long)k * f.F + fr;
    PIPE_STAMP(0, 0);
    if (k > 0 && !have_next && !pipe_acquire(prog + ((k - 1) * nbt + bt) * PIPE_STRIDE, (unsigned)n + 1u, abort_w, tid, &s_ok, fenced)) break;

    PIPE_STAMP(0, 1);
    // ---- P0: actnorm (glow/modules.py:45-52); stage a k-major
    {
      const float* xin = (k == 0) ? f.x0 + ((long)row * f.T + f.start + n) * C : f.sX + (kf - f.F + row) * LC;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = cl + 32 * u;
        if (c < C16) {
          float a = 0.0f;
          if (c < C && rok) a = ((k == 0 ? xin[c] : (have_next ? xnext[u] : ld_tile(xin + c, fenced))) + anb[u]) * ans[u];
          At[c * LT + ri] = a;
        }
      }
    }
    __syncthreads();
    PIPE_STAMP(0, 2);

    // ---- P1: y = a W   (InvertibleConv1x1.forward, glow/modules.py:186)
    if (t1) {
      const f32x4 acc = mma16_lds(At + kq * LT + l15, w1s, nbC);
      const int c = tcol;
      if (c < C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = kq * 4 + r;
          const float v = acc[r];
          Yrm[i * ldy + c] = v;
          if (c < Ch) {
            Zt[c * LT + i] = v;
            if (X3) x3_put(ich, ich + MB * ldxi, i * ldxi + x3_pos(c), v);
          }
        }
      }
    }
    __syncthreads();
    PIPE_STAMP(0, 3);

    // ---- P2: recurrent cell of the coupling net (f_seq.forward, glow/models.py:204-214); state stays in LDS / registers
    if (t2) {
      float cnew[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (X3)
        fast_cell_p2_x3_img<NG>(f, ich, ich + MB * ldxi, ldxi, Ch16, Ht, Hn, wzx, whx, gc, bh, cprev, (nbZ + 1) >> 1, (nbH + 1) >> 1,
                                tcol, kq, l15, b0, B, nullptr, NG == 4 ? f.sC + kf * H : nullptr, f.sG + kf * 4 * H, cnew,
                                inh, inh + MB * ldxi);
      else
        fast_cell_p2<NG>(f, Zt, Ht, Hn, wz, wh, gc, bh, cprev, nbZ, nbH, tcol, kq, l15, b0, B, nullptr,
                         NG == 4 ? f.sC + kf * H : nullptr, f.sG + kf * 4 * H, cnew);
#pragma unroll
      for (int r = 0; r < 4; ++r) cprev[r] = cnew[r];
    }
    if (n + 1 < f.N) {
      const float* gicb = f.gic + (kf + B) * G;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rw = min(b0 + kq * 4 + r, B - 1);
#pragma unroll
        for (int g = 0; g < NG; ++g) gc[r][g] = gicb[(long)rw * G + g * H + jc];
      }
    }
    __syncthreads();
    PIPE_STAMP(0, 4);

    // ---- P3: o = (h' Wfl^T + b) exp(3 logs)   (LinearZeros, glow/modules.py:93-95)
    if (t3) {
      // (exact f32 MFMA in every mode: o sets the coupling's scale and shift directly, and the inverse pass - sampling, invert -
      // computes it in exact f32; with three bf16 products here decode(encode(x)) at 96 flow steps went from 7e-4 to 3e-3 of x
      // for 20 us of a 500 us walk)
      const f32x4 acc = mma16_lds(Hn + kq * LT + l15, w3s, nbH);
      if (tcol < Cout) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = kq * 4 + r;
          const float o = (acc[r] + flb) * fls;
          Orm[i * ldo + tcol] = o;
        }
      }
    }
    __syncthreads();
    PIPE_STAMP(0, 5);

    // ---- P4: coupling (glow/models.py:330-341), pass-through half, log-det; the output tile is the hand-off payload
    const bool peek = k > 0 && !fenced && n + 1 < f.N;
    unsigned pk = 0u;
    if (peek && tid == 0) pk = ld_agent(prog + ((k - 1) * nbt + bt) * PIPE_STRIDE);
    // the cell's stash rows a, y, h', o leave from their LDS tiles here, 16 bytes per thread and array (thread (row ri, columns
    // 4 cl .. 4 cl + 3)): one store instruction each instead of the 2 / 4 / 4 / 4 dword stores per lane that P0 .. P3 issued from the
    // MFMA accumulator layout (the walk is bound by vector-memory instruction issue; round 3: 20 % faster without its stash stores)
    if (rok) {
      const int c0 = 4 * cl;
      if (c0 < C) {   // (LC, LO are multiples of 4 floats: the last vector's tail lies in the row padding)
        f32x4 va, vy;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = c0 + e < C ? At[(c0 + e) * LT + ri] : 0.0f;
          vy[e] = c0 + e < C ? Yrm[ri * ldy + c0 + e] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(f.sA + (kf + row) * LC + c0) = va;
        *reinterpret_cast<f32x4*>(f.sY + (kf + row) * LC + c0) = vy;
      }
      if (c0 < Cout) {
        f32x4 vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) vo[e] = c0 + e < Cout ? Orm[ri * ldo + c0 + e] : 0.0f;
        *reinterpret_cast<f32x4*>(f.sO + (kf + row) * LO + c0) = vo;
      }
      for (int h0 = c0; h0 < H; h0 += 128) {   // (H <= 128 on this path: one trip)
        f32x4 vh;
#pragma unroll
        for (int e = 0; e < 4; ++e) vh[e] = h0 + e < H ? Hn[(h0 + e) * LT + ri] : 0.0f;
        if (h0 + 3 < H && (H & 3) == 0) *reinterpret_cast<f32x4*>(f.sH + (kf + row) * H + h0) = vh;
        else
          for (int e = 0; e < 4; ++e)
            if (h0 + e < H) f.sH[(kf + row) * H + h0 + e] = vh[e];
      }
    }
    {
      float lg = 0.0f;
      if (cl < C2) {
        const float z2 = Yrm[ri * ldy + Ch + cl];
        float z2n;
        if (f.affine) {
          const float shift = Orm[ri * ldo + 2 * cl];
          const float sraw = sigmoidf_(Orm[ri * ldo + 2 * cl + 1] + 2.0f);
          const float sc = fmaxf(sraw, f.eps);
          z2n = (z2 + shift) * sc;
          lg = logf(sc);
        } else {
          z2n = z2 + Orm[ri * ldo + cl];
        }
        if (rok) st_sc1(f.sX + (kf + row) * LC + Ch + cl, z2n);
      }
      if (cl < Ch && rok) st_sc1(f.sX + (kf + row) * LC + cl, Yrm[ri * ldy + cl]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);  // the 32 lanes of one row
      if (cl == 0 && rok) f.sL[kf + row] = lg;
    }
    PIPE_STAMP(0, 6);
    // (pipe_publish, with the outcome of the peek riding on its barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (peek && tid == 0) s_rdy = pk >= (unsigned)n + 2u ? 1 : 0;
    __syncthreads();
    if (k + 1 < f.Ks && tid == 0) st_agent(prog + (k * nbt + bt) * PIPE_STRIDE, (unsigned)n + 1u);
    have_next = peek && s_rdy != 0;
    if (have_next) {
      const float* xn = f.sX + (kf + B - f.F + row) * LC;   // cell n + 1 of step k - 1
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = cl + 32 * u;
        xnext[u] = (c < C && rok) ? ld_tile(xn + c, false) : 0.0f;
      }
    }
    PIPE_STAMP(0, 7);
    float* t = Ht; Ht = Hn; Hn = t;
    __bf16* ti = ich; ich = inh; inh = ti;
  }
}

// FlowStep.reverse_flow (glow/models.py:345-373) with explicit state, register-resident weights: the sampler's and
// SeqGlow.invert's cell. coupling^-1 -> invconv^-1 (W^-1 image) -> actnorm^-1.
// wait_flag / pub_flag: hand-off words of the per-frame reverse chain (flow_rev_chain_kernel), or null for a stand-alone launch:
// the input tile is then read with sc1 loads after the producer's progress word is seen, and the output tile is stored sc1,
// drained and published (the hand-off of the persistent walks).
// need / pub_value: the progress value waited for / published (1 for the one-frame chain; timestep + 1 in the persistent reverse
// walk). false = the wait was abandoned (abort word set): nothing was computed.
// XW (with X3): the weights come as the fp16 fragment images lfi_flow_prep left (FlowK.hwz ..): no f32 fragments, no split here.
template <int NG, bool X3 = false, bool XW = false>
__device__ __forceinline__ bool rev_fast_cell(const FlowK& f, const CellIO& io, int b0, const unsigned* wait_flag,
                                              unsigned* abort_w, unsigned* pub_flag, int* s_ok, unsigned need = 1u,
                                              unsigned pub_value = 1u) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int ri = tid >> 5, cl = tid & 31;
  const int k = io.k, rows = io.rows;
  const int C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, G = f.G;
  const int C16 = f.C16, Ch16 = f.Ch16, H16 = f.H16, Co16 = f.Co16;
  const CarveF cv = carve_fast_fwd(C, C16, H16, Ch16, Cout);
  float* Yt = flow_smem + cv.At;   // y = [z1 | z2] k-major for the W^-1 product
  float* Ht = flow_smem + cv.Ht;
  float* Zt = flow_smem + cv.Zt;
  float* Hn = flow_smem + cv.Hn;
  float* Yrm = flow_smem + cv.Yrm;
  float* Orm = flow_smem + cv.Orm;
  const int ldy = C + 1, ldo = Cout + 1;
  const int nbC = C16 >> 4, nbZ = Ch16 >> 4, nbH = H16 >> 4;
  const bool t1 = wave * 16 < C, t2 = wave * 16 < H, t3 = wave * 16 < Cout;
  const int tcol = wave * 16 + l15;
  // phase stamps of the stamping workgroup (flow step LFI_STAMP_K, tile 0): s_memtime at the phase boundaries (tools/rev_stamps.py)
#define REV_STAMP(slot)                                                                                                  \
  do {                                                                                                                   \
    if (f.stamps && io.stamp_base > 0 && tid == 0 && b0 == 0 && k == f.stamp_k)                                           \
      f.stamps[io.stamp_base - 1 + (slot)] = __builtin_amdgcn_s_memtime();                                               \
  } while (0)
  REV_STAMP(0);
  static_assert(!XW || X3, "pre-split weight images are the X3 cell's");
  f32x4 wz[XW ? 1 : NG][XW ? 1 : FB_Z], wh[XW ? 1 : NG][XW ? 1 : FB_H], w3[XW ? 1 : FB_H];
  X3FragH wzx[X3 ? NG : 1][FB_Z / 2];            // three fp16 products (fp32-grade, x3h_*): the z1-side fragments
  X3FragH whx[XW ? NG : 1][XW ? FB_H / 2 : 1];   // XW: the h-side fragments too (otherwise split where they are used)
  X3FragH w3x[X3 ? FB_H / 2 : 1], w1x[X3 ? FB_C / 2 : 1];
  // this lane's entries of a pre-split image of flow step k: nb2 32-k blocks of the 16-column tile at `col` (row pitch J entries)
  auto load_x3h = [&](X3FragH* w, int maxb2, const uint4* img, int K16, int J, int col, int nb2, bool on) {
    const long per = (long)(K16 >> 5) * 4 * J;
    const uint4* p = img + (long)k * 2 * per + (long)kq * J + col;
#pragma unroll
    for (int b = 0; b < maxb2; ++b)
      if (on && b < nb2) {
        w[b].hi = __builtin_bit_cast(fh16x8, p[(long)b * 4 * J]);
        w[b].lo = __builtin_bit_cast(fh16x8, p[(long)b * 4 * J + per]);
      } else {
        w[b].hi = (fh16x8)(_Float16)0.0f;
        w[b].lo = (fh16x8)(_Float16)0.0f;
      }
  };
  float hv[XW ? FB_H / 2 : 1];   // XW: this thread's elements of h_prev (row ri, columns cl + 32 q), staged to LDS further down
  if constexpr (XW) {
    // Vector-memory results come back in issue order: what the work in front of the wait needs first is issued first - h_prev
    // (its LDS image gates the barrier), then the h-side fragments of the product that runs before the wait; the fragments of the
    // phases behind the wait follow and arrive under that product.
    const int row = b0 + ri;
#pragma unroll
    for (int q = 0; q < FB_H / 2; ++q) {
      const int j = cl + 32 * q;
      hv[q] = (io.h_prev && row < rows && j < H) ? ld_tile(io.h_prev + (long)row * H + j, io.state_l2 == 0) : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) load_x3h(whx[g], FB_H / 2, f.hwh, H16, NG * H16, g * H16 + tcol, nbH >> 1, t2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) load_x3h(wzx[g], FB_Z / 2, f.hwz, Ch16, NG * H16, g * H16 + tcol, nbZ >> 1, t2);
    load_x3h(w3x, FB_H / 2, f.hwfl, H16, Co16, tcol, nbH >> 1, t3);
  } else {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int b = 0; b < FB_Z; ++b) wz[g][b] = zero4;
#pragma unroll
      for (int b = 0; b < FB_H; ++b) wh[g][b] = zero4;
      load_frag<FB_Z>(wz[g], f.pwz + (long)k * Ch16 * NG * H16, NG * H16, g * H16 + tcol, kq, nbZ, t2);
      load_frag<FB_H>(wh[g], f.pwh + (long)k * H16 * NG * H16, NG * H16, g * H16 + tcol, kq, nbH, t2);
    }
    load_frag<FB_H>(w3, f.pwfl + (long)k * H16 * Co16, Co16, tcol, kq, nbH, t3);
  }
  float gc[4][NG], bh[NG], cprev[4];
  {
    const float* bhh = f.p.b_hh + (long)k * G;
    const int jc = tcol < H ? tcol : 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) bh[g] = bhh[g * H + jc];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(b0 + kq * 4 + r, rows - 1);
#pragma unroll
      for (int g = 0; g < NG; ++g) gc[r][g] = io.gic[(long)row * G + g * H + jc];
      cprev[r] = (NG == 4 && io.c_prev) ? ld_tile(io.c_prev + (long)row * H + jc, io.state_l2 == 0) : 0.0f;
    }
  }
  // ---- everything that does not depend on the incoming tile runs BEFORE the wait for it (a chain of Ks dependent cells pays
  // whatever follows the wait Ks times per frame; stamps of round 4: staging h_prev, splitting the weight fragments into fp16
  // pieces and the h_prev W_hh half of the recurrent product - 4 of its 5 k-blocks - were 10 k of a cell's 20 k dependent cycles)
  if constexpr (XW) {
#pragma unroll
    for (int q = 0; q < FB_H / 2; ++q) {
      const int j = cl + 32 * q;
      if (j < H16) {
        Ht[j * LT + ri] = hv[q];
        if (j >= H) Hn[j * LT + ri] = 0.0f;
      }
    }
    for (int c = Ch + cl; c < Ch16; c += 32) Zt[c * LT + ri] = 0.0f;
  } else {
    const int row = b0 + ri;
    const bool rok = row < rows;
    for (int j = cl; j < H16; j += 32) {
      Ht[j * LT + ri] = (io.h_prev && rok && j < H) ? ld_tile(io.h_prev + (long)row * H + j, io.state_l2 == 0) : 0.0f;
      if (j >= H) Hn[j * LT + ri] = 0.0f;
    }
    for (int c = Ch + cl; c < Ch16; c += 32) Zt[c * LT + ri] = 0.0f;
  }
  // W^-1 slice of this wave's 16 output channels: in flight under the coupling net
  f32x4 w1[XW ? 1 : FB_C];
  if constexpr (XW) load_x3h(w1x, FB_C / 2, f.hWinv, C16, C16, tcol, nbC >> 1, t1);
  else load_frag<FB_C>(w1, f.pWinv + (long)k * C16 * C16, C16, tcol, kq, nbC, t1);
  // per-column constants of the phases after the wait (LinearZeros bias / scale, ActNorm^-1 scale / bias): loaded here, not between
  // the barriers of the dependent phases (two L2 round trips per cell each)
  const float flb = tcol < Cout ? f.p.b_fl[(long)k * Cout + tcol] : 0.0f;
  const float fls = tcol < Cout ? expf(3.0f * f.p.l_fl[(long)k * Cout + tcol]) : 0.0f;
  const float an_es = tcol < C ? expf(-f.p.an_logs[(long)k * C + tcol]) : 0.0f;
  const float an_bb = tcol < C ? f.p.an_bias[(long)k * C + tcol] : 0.0f;
  // X3: LinearZeros and W^-1 as three fp16 products too (K = H and K = C: 32 and 16 dependent f32-input MFMAs of 32 cycles
  // otherwise, per cell, after the wait); their weight fragments are split here, before it. Whole pairs of 16-k blocks only.
  // (X3 is only instantiated for shapes with whole pairs everywhere: the launcher checks H16, Ch16 and C16)
  if constexpr (X3 && !XW) {
#pragma unroll
    for (int b = 0; b < FB_H / 2; ++b)
      if (b < (nbH >> 1)) w3x[b] = x3h_pack(w3[2 * b], w3[2 * b + 1]);
#pragma unroll
    for (int b = 0; b < FB_C / 2; ++b)
      if (b < (nbC >> 1)) w1x[b] = x3h_pack(w1[2 * b], w1[2 * b + 1]);
  }
  __syncthreads();
  f32x4 az[NG], ah[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    az[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ah[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (t2) {
    const float* hl = Ht + kq * LT + l15;
    if constexpr (XW) {
#pragma unroll
      for (int b = 0; b < FB_H / 2; ++b)
        if (b < ((nbH + 1) >> 1)) {
          const X3FragH a = x3h_a(hl + b * 32 * LT);
#pragma unroll
          for (int g = 0; g < NG; ++g) ah[g] = x3h_mma(a, whx[g][b], ah[g]);
        }
    } else if constexpr (X3) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int b = 0; b < FB_Z / 2; ++b) wzx[g][b] = x3h_pack(wz[g][2 * b], wz[g][2 * b + 1]);
#pragma unroll
      for (int b = 0; b < FB_H / 2; ++b)
        if (b < ((nbH + 1) >> 1)) {
          const X3FragH a = x3h_a(hl + b * 32 * LT);
#pragma unroll
          for (int g = 0; g < NG; ++g) ah[g] = x3h_mma(a, x3h_pack(wh[g][2 * b], wh[g][2 * b + 1]), ah[g]);
        }
    } else {
#pragma unroll
      for (int b = 0; b < FB_H; ++b)
        if (b < nbH) {
          const float* ab = hl + b * 16 * LT;
          const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
#pragma unroll
          for (int g = 0; g < NG; ++g) ah[g] = mfma16(a0, wh[g][b][0], ah[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) ah[g] = mfma16(a1, wh[g][b][1], ah[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) ah[g] = mfma16(a2, wh[g][b][2], ah[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) ah[g] = mfma16(a3, wh[g][b][3], ah[g]);
        }
    }
  }
  if constexpr (XW) {
    // the fragments of the phases AFTER the wait are plain loads now: pin them in front of it (the compiler sinks a load towards its
    // use - behind the wait, where a chain of Ks cells pays its L2 round trip Ks times per frame)
    auto pin = [](X3FragH& w) { asm volatile("" : "+v"(w.hi), "+v"(w.lo)); };
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int b = 0; b < FB_Z / 2; ++b) pin(wzx[g][b]);
#pragma unroll
    for (int b = 0; b < FB_H / 2; ++b) pin(w3x[b]);
#pragma unroll
    for (int b = 0; b < FB_C / 2; ++b) pin(w1x[b]);
  }
  REV_STAMP(1);
  if (wait_flag && !pipe_acquire(wait_flag, need, abort_w, tid, s_ok, false)) return false;
  REV_STAMP(2);
  // ---- R0: stage the tile [z1 | z2']
  {
    const int row = b0 + ri;
    const bool rok = row < rows;
    for (int c = cl; c < C16; c += 32) {
      const float v = (c < C && rok) ? ld_tile(io.x_in + (long)row * io.ldx + c, wait_flag == nullptr) : 0.0f;
      if (c < C) Yrm[ri * ldy + c] = v;
      if (c < Ch) Zt[c * LT + ri] = v;
      if (c < Ch || c >= C) Yt[c * LT + ri] = v;   // z1 rows and the zero k padding; z2 rows come from R3
    }
  }
  __syncthreads();
  REV_STAMP(3);
  if (t2) {   // the z1 half of the product (one k-block at C <= 64), then the gate math
    const float* zl = Zt + kq * LT + l15;
    if constexpr (X3) {
#pragma unroll
      for (int b = 0; b < FB_Z / 2; ++b)
        if (b < ((nbZ + 1) >> 1)) {
          const X3FragH a = x3h_a(zl + b * 32 * LT);
#pragma unroll
          for (int g = 0; g < NG; ++g) az[g] = x3h_mma(a, wzx[g][b], az[g]);
        }
    } else {
#pragma unroll
      for (int b = 0; b < FB_Z; ++b)
        if (b < nbZ) {
          const float* ab = zl + b * 16 * LT;
          const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
#pragma unroll
          for (int g = 0; g < NG; ++g) az[g] = mfma16(a0, wz[g][b][0], az[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) az[g] = mfma16(a1, wz[g][b][1], az[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) az[g] = mfma16(a2, wz[g][b][2], az[g]);
#pragma unroll
          for (int g = 0; g < NG; ++g) az[g] = mfma16(a3, wz[g][b][3], az[g]);
        }
    }
    fast_cell_p2_gates<NG>(f, Ht, Hn, az, ah, gc, bh, cprev, tcol, kq, b0, rows, io.h_out, io.c_out, nullptr, nullptr);
  }
  __syncthreads();
  REV_STAMP(4);
  if (t3) {
    if constexpr (X3) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* hl = Hn + kq * LT + l15;
#pragma unroll
      for (int b = 0; b < FB_H / 2; ++b)
        if (b < (nbH >> 1)) acc = x3h_mma(x3h_a(hl + b * 32 * LT), w3x[b], acc);
      if (tcol < Cout) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Orm[(kq * 4 + r) * ldo + tcol] = (acc[r] + flb) * fls;
      }
    } else {
      fast_cell_p3(f, k, Hn, Orm, w3, nbH, tcol, kq, l15, b0, rows, nullptr, 0, flb, fls);
    }
  }
  __syncthreads();
  REV_STAMP(5);
  // ---- R3: coupling inverse (glow/models.py:356-365)
  {
    const int row = b0 + ri;
    float lg = 0.0f;
    if (cl < C2) {
      const float z2n = Yrm[ri * ldy + Ch + cl];
      float z2;
      if (f.affine) {
        const float shift = Orm[ri * ldo + 2 * cl];
        const float sraw = sigmoidf_(Orm[ri * ldo + 2 * cl + 1] + 2.0f);
        const float sc = fmaxf(sraw, f.eps);
        z2 = z2n / sc;
        z2 = z2 - shift;
        lg = -logf(sc);
      } else {
        z2 = z2n - Orm[ri * ldo + cl];
      }
      Yt[(Ch + cl) * LT + ri] = z2;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
    if (cl == 0 && row < rows && io.l_out) {
      if (io.l_accumulate) io.l_out[row] += lg; else io.l_out[row] = lg;
    }
  }
  __syncthreads();
  REV_STAMP(6);
  // ---- R4: x = (y W^-1) exp(-logs) - bias   (scale then center, glow/modules.py:76-79)
  if (t1) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      const float* yl = Yt + kq * LT + l15;
#pragma unroll
      for (int b = 0; b < FB_C / 2; ++b)
        if (b < (nbC >> 1)) acc = x3h_mma(x3h_a(yl + b * 32 * LT), w1x[b], acc);
    } else {
      acc = mma16_reg<FB_C>(Yt + kq * LT + l15, w1, nbC);
    }
    const int c = tcol;
    if (c < C) {
      const float es = an_es, bb = an_bb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = b0 + kq * 4 + r;
        if (row < rows) {
          if (pub_flag) st_sc1(io.x_out + (long)row * io.ldxo + c, acc[r] * es - bb);
          else io.x_out[(long)row * io.ldxo + c] = acc[r] * es - bb;
        }
      }
    }
  }
  REV_STAMP(7);
  if (pub_flag) pipe_publish(pub_flag, pub_value, tid, true);
  REV_STAMP(8);
#undef REV_STAMP
  return true;
}

template <int NG>
__global__ __launch_bounds__(NT) void flow_step_rev_fast_kernel(FlowK f, CellIO io) {
  rev_fast_cell<NG>(f, io, blockIdx.x * MB, nullptr, nullptr, nullptr, nullptr);
}

// One generated frame of the sampler: all Ks reverse flow steps of all batch tiles in ONE launch instead of Ks launches of
// B / 16 workgroups each (64 of 256 CUs at batch 1024, 270 KB of weights fetched behind every launch boundary). Workgroup
// (k, tile) - ids by ticket, k descending, so a workgroup only waits on one that already runs - requests its weights, its
// part of gic and its recurrent state, then waits for the tile of step k + 1 (the prior noise for k = Ks - 1), runs the
// cell and hands its tile to step k - 1 (step 0 writes the frame). Tiles of one sample block chain strictly, so the two
// ping-pong tile buffers of the per-step launches still do.
struct RevChain {
  const float* noise;     // B x C prior draws of this frame
  float *xa, *xb;         // B x C tile buffers: step k writes (k & 1) ? xa : xb
  float* frame; long ld_frame;   // output rows of this frame in faces (row stride seq_len * C)
  const float* gic;       // [Ks][B][G]
  float *h, *cstate;      // [Ks][B][H] recurrent state, updated in place
  int has_prev;           // 0 at the first generated frame (zero state)
  int frame_no;           // index of the generated frame (diagnostic phase stamps of frames < 128 only)
  unsigned* pipe;         // ticket, abort, progress words (zeroed before every launch)
  // round 5: step 0's workgroups also leave the NEXT frame's window as the fp16 fragments the fused conditioning kernel reads
  // (lfi_sample.hip, sc_xfrag_kernel's format: tile bt, step m, plane: lane l, element e = window[16 bt + (l & 15)][32 m + 8 (l >> 4) + e])
  // - one launch per generated frame less; null: the conditioning call makes them itself
  _Float16* xf;
  const float* faces;     // row 0 of the frames buffer (row pitch ld_frame)
  long xf_off;            // first window column of the next frame in a row: (t + 1 - hist1) * C
  int K1, NM1;
};
template <int NG, bool X3, bool XW = false>
__global__ __launch_bounds__(NT) void flow_rev_chain_kernel(FlowK f, RevChain rc) {
  __shared__ int s_id, s_ok;
  if (threadIdx.x == 0) s_id = (int)atomicAdd(rc.pipe, 1u);
  __syncthreads();
  const int nbt = f.nbt;
  const int kk = s_id / nbt, bt = s_id - kk * nbt;
  if (kk >= f.Ks) return;
  const int k = f.Ks - 1 - kk;
  unsigned* prog = rc.pipe + PIPE_HDR;
  CellIO io = {};
  io.k = k; io.rows = f.B;
  if (k == f.Ks - 1) { io.x_in = rc.noise; io.ldx = f.C; }
  else { io.x_in = ((k + 1) & 1) ? rc.xa : rc.xb; io.ldx = f.C; }
  if (k == 0) { io.x_out = rc.frame; io.ldxo = rc.ld_frame; }
  else { io.x_out = (k & 1) ? rc.xa : rc.xb; io.ldxo = f.C; }
  io.h_prev = rc.has_prev ? rc.h + (long)k * f.B * f.H : nullptr;
  io.h_out = rc.h + (long)k * f.B * f.H;
  if (NG == 4) { io.c_prev = rc.has_prev ? rc.cstate + (long)k * f.B * f.H : nullptr; io.c_out = rc.cstate + (long)k * f.B * f.H; }
  io.gic = rc.gic + (long)k * f.B * f.G;
  io.stamp_base = rc.frame_no < 128 ? 1024 + 16 * rc.frame_no + 1 : 0;
  rev_fast_cell<NG, X3, XW>(f, io, bt * MB, k + 1 < f.Ks ? prog + (k + 1) * nbt + bt : nullptr, rc.pipe + 1,
                    k > 0 ? prog + k * nbt + bt : nullptr, &s_ok);
  if (k == 0 && ld_agent(rc.pipe + 1) != 0u) {   // an abandoned chain must not pass for a frame
    const int row = bt * MB + (int)(threadIdx.x >> 5);
    if (row < f.B)
      for (int c = threadIdx.x & 31; c < f.C; c += 32) rc.frame[(long)row * rc.ld_frame + c] = __builtin_nanf("");
  }
  if (k == 0 && rc.xf) {
    // the frame's rows of this tile are on their way to memory: drain, meet, then read the window back past the L1 (agent scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    typedef _Float16 xh8 __attribute__((ext_vector_type(8)));
    for (int it = threadIdx.x; it < rc.NM1 * 64; it += NT) {
      const int l = it & 63, m = it >> 6;
      const int row = bt * MB + (l & 15), kk0 = 32 * m + 8 * (l >> 4);
      xh8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = 0.0f;
        if (row < f.B && kk0 + e < rc.K1) v = ld_tile(rc.faces + (long)row * rc.ld_frame + rc.xf_off + kk0 + e, false);
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
      }
      _Float16* dst = rc.xf + ((long)(bt * rc.NM1 + m) * 2) * 512 + l * 8;
      *reinterpret_cast<xh8*>(dst) = hi;
      *reinterpret_cast<xh8*>(dst + 512) = lo;
    }
  }
}

// SeqGlow.invert (glow/models.py:617-645): the teacher-forced reverse pass over ALL timesteps in ONE launch - the reverse twin of the
// persistent forward walk. Workgroup (k, tile), ids by ticket with k descending, walks n = 0 .. N-1: it waits for step k + 1's tile
// of timestep n (the latent z_n for k = Ks - 1), runs the reverse cell with its recurrent state carried in h / cstate (its own
// rows, updated in place) and hands its tile to step k - 1 (step 0 writes x_n). Every (n, k) tile has its own slot in `tiles`, so a
// fast producer never overwrites what its consumer has not read and more workgroups than CUs just run as successive groups. The
// coupling log-det of every (k, n, row) goes to its own word of `ldk` (workgroups on different CUs must not read-modify-write one
// accumulator between kernel boundaries); the host call sums them over k.
struct RevWalk {
  const float* z;       // [N][B][C]
  float* tiles;         // [Ks][N * B][C]
  float* out;           // [N][B][C]
  const float* gic;     // [Ks][N * B][G]
  float *h, *cstate;    // [Ks][B][H]
  float* ldk;           // [Ks][N * B]
  unsigned* pipe;       // ticket, abort, 2 reserved, then one progress word per (k, tile): timesteps published
};
template <int NG>
__global__ __launch_bounds__(NT) void flow_rev_walk_kernel(FlowK f, RevWalk rw) {
  __shared__ int s_id, s_ok;
  if (threadIdx.x == 0) s_id = (int)atomicAdd(rw.pipe, 1u);
  __syncthreads();
  const int nbt = f.nbt;
  const int kk = s_id / nbt, bt = s_id - kk * nbt;
  if (kk >= f.Ks) return;
  const int k = f.Ks - 1 - kk;
  unsigned* prog = rw.pipe + PIPE_HDR;
  const long F = f.F, B = f.B;
  bool ok = true;
  for (int n = 0; n < f.N && ok; ++n) {
    CellIO io = {};
    io.k = k; io.rows = f.B; io.ldx = f.C; io.ldxo = f.C;
    io.x_in = (k == f.Ks - 1) ? rw.z + (long)n * B * f.C : rw.tiles + ((long)(k + 1) * F + (long)n * B) * f.C;
    io.x_out = (k == 0) ? rw.out + (long)n * B * f.C : rw.tiles + ((long)k * F + (long)n * B) * f.C;
    io.h_prev = n > 0 ? rw.h + (long)k * B * f.H : nullptr;
    io.h_out = rw.h + (long)k * B * f.H;
    if (NG == 4) { io.c_prev = n > 0 ? rw.cstate + (long)k * B * f.H : nullptr; io.c_out = rw.cstate + (long)k * B * f.H; }
    io.gic = rw.gic + ((long)k * F + (long)n * B) * f.G;
    io.l_out = rw.ldk + (long)k * F + (long)n * B; io.l_accumulate = 0;
    io.state_l2 = 1;
    io.stamp_base = n < 128 ? 1024 + 16 * n + 1 : 0;
    ok = rev_fast_cell<NG, false>(f, io, bt * MB, k + 1 < f.Ks ? prog + (k + 1) * nbt + bt : nullptr, rw.pipe + 1,
                                  k > 0 ? prog + k * nbt + bt : nullptr, &s_ok, (unsigned)n + 1u, (unsigned)n + 1u);
    __syncthreads();   // the cell's last reads of the LDS operands are done before the next timestep stages its own
  }
  if (k == 0 && ld_agent(rw.pipe + 1) != 0u) {   // an abandoned walk must not pass for a reconstruction
    const int row = bt * MB + (int)(threadIdx.x >> 5);
    if (row < f.B)
      for (int n = 0; n < f.N; ++n)
        for (int c = threadIdx.x & 31; c < f.C; c += 32) rw.out[((long)n * B + row) * f.C + c] = __builtin_nanf("");
  }
}

struct CarveFB {
  int Dl, Gi, Dy, Cy, Pl, total;
};
__host__ __device__ inline CarveFB carve_fast_bwd(int C16, int H16, int Co16, int Cout, int NG) {
  CarveFB c;
  int o = 0;
  c.Dl = o; o += Co16 * LT;
  c.Gi = o; o += 2 * NG * H16 * LT;   // Gi then Gh, each [NG][H16][LT]
  c.Dy = o; o += C16 * LT;
  c.Cy = o; o += H16 * LT;
  c.Pl = o; o += MB * (Cout + 1);
  c.total = o;
  return c;
}

// acc + sum over NG gate blocks (each nb blocks of 16 k; a_lane advances by blk floats per gate)
template <int NG, int MAXB>
__device__ __forceinline__ f32x4 mma16_reg_gates(const float* a_lane, int blk, const f32x4 (&w)[NG][MAXB], int nb) {
  f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
      if (b < nb) {
        const float* ab = a_lane + g * blk + b * 16 * LT;
        const float a0 = ab[0], a1 = ab[4 * LT], a2 = ab[8 * LT], a3 = ab[12 * LT];
        e = mfma16(a0, w[g][b][0], e);
        o = mfma16(a1, w[g][b][1], o);
        e = mfma16(a2, w[g][b][2], e);
        o = mfma16(a3, w[g][b][3], o);
      }
  return e + o;
}

template <int NG>
__global__ __launch_bounds__(NT) void flow_diag_bwd_fast_kernel(FlowK f, int d, int klo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int ri = tid >> 5, cl = tid & 31;
  int cell, bt;
  flow_cell_of_block(f.nbt, &cell, &bt);
  const int k = klo + cell, n = d - k;
  const int b0 = bt * MB;
  const int B = f.B, C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, G = f.G;
  const int C16 = f.C16, Ch16 = f.Ch16, H16 = f.H16, Co16 = f.Co16;
  const long LC = f.ldc, LO = f.ldo;   // stash row strides
  const long fr = (long)n * B;
  const long kf = (long)k * f.F + fr;
  const CarveFB cv = carve_fast_bwd(C16, H16, Co16, Cout, NG);
  float* Dl = flow_smem + cv.Dl;
  float* Gi = flow_smem + cv.Gi;
  float* Gh = Gi + NG * H16 * LT;
  float* Dy = flow_smem + cv.Dy;
  float* Cy = flow_smem + cv.Cy;
  float* Pl = flow_smem + cv.Pl;
  const int ldp = Cout + 1;
  const int nbC = C16 >> 4, nbH = H16 >> 4, nbO = Co16 >> 4;
  const bool last = k == f.Ks - 1;
  const float gz = f.gscale / LN2_F;   // d loss / d z = z * gz   (prior term)
  const float dl = -f.gscale / LN2_F;  // d loss / d logdet
  const float* dxo = last ? f.sX + kf * LC : f.bDx + (kf + f.F) * LC;
  const float dxs = last ? gz : 1.0f;

  // which tiles this wave owns: hidden tile `wave` (Q1, Q2), z tile `wave` (Q2, waves < Ch16/16), channel tile `wave` (Q3)
  const bool th = wave * 16 < H, tz = wave * 16 < Ch, tc = wave * 16 < C;
  const int tcol = wave * 16 + l15;
  // ---- weights of Q1 (dlin Wfl: K = Cout) for this wave's hidden tile
  f32x4 wq1[FB_O];
  load_frag<FB_O>(wq1, f.bwfl + (long)k * Co16 * H16, H16, tcol, kq, nbO, th);
  // the z-tile waves stream W_ih[:, :Ch] first (they run dz1 before the barrier), the others W_hh
  f32x4 wq2[NG][FB_H];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (tz) load_frag<FB_H>(wq2[g], f.bwz + ((long)k * NG + g) * H16 * Ch16, Ch16, tcol, kq, nbH, true);
    else load_frag<FB_H>(wq2[g], f.bwh + ((long)k * NG + g) * H16 * H16, H16, tcol, kq, nbH, th);
  }

  // ---- every stash operand of Q1 / Q2 / Q3 (none depends on this cell's arithmetic) is requested now: issued inside
  //      their phases, each of these HBM round trips (~2 us) was exposed behind a barrier
  float sg[4][4], shp[4], sdhf[4], sc2[4], scp[4], sdcf[4], sdxo[4], sa[4];
  {
    const int j = tcol < H ? tcol : 0, cz = tcol < Ch ? tcol : 0, cc = tcol < C ? tcol : 0;
    const bool hasn = n < f.N - 1, hasp = n > 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = min(b0 + kq * 4 + r, B - 1);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(f.sG + (kf + row) * 4 * H + 4 * j);
      sg[r][0] = g4[0]; sg[r][1] = g4[1]; sg[r][2] = g4[2]; sg[r][3] = g4[3];
      shp[r] = hasp ? f.sH[(kf - B + row) * H + j] : 0.0f;
      sdhf[r] = hasn ? f.bDh[(kf + B + row) * H + j] : 0.0f;
      if (NG == 4) {
        sc2[r] = f.sC[(kf + row) * H + j];
        scp[r] = hasp ? f.sC[(kf - B + row) * H + j] : 0.0f;
        sdcf[r] = hasn ? f.bDc[(kf + B + row) * H + j] : 0.0f;
      }
      sdxo[r] = dxo[row * LC + cz] * dxs;
      sa[r] = f.sA[(kf + row) * LC + cc];
    }
  }

  // ---- Q0: coupling backward + LinearZeros scale; zero the k padding of the LDS operands
  {
    const int row = b0 + ri;
    const bool rok = row < B;
    const float* lfl = f.p.l_fl + (long)k * Cout;
    float dz2 = 0.0f, dl0 = 0.0f, dl1 = 0.0f, p0 = 0.0f, p1 = 0.0f;
    if (cl < C2) {
      if (rok) {
        const float dz2n = dxo[(long)row * LC + Ch + cl] * dxs;
        const float* O = f.sO + (kf + row) * LO;
        if (f.affine) {
          const float oe = O[2 * cl], oo = O[2 * cl + 1];
          const float sraw = sigmoidf_(oo + 2.0f);
          const float sc = fmaxf(sraw, f.eps);
          const float z2 = f.sY[(kf + row) * LC + Ch + cl];
          dz2 = dz2n * sc;
          const float dsc = dz2n * (z2 + oe) + dl / sc;
          const float dsr = sraw >= f.eps ? dsc : 0.0f;
          const float d0 = dz2;                             // d o_even (shift)
          const float d1 = dsr * sraw * (1.0f - sraw);      // d o_odd
          p0 = d0 * oe * 3.0f; p1 = d1 * oo * 3.0f;
          dl0 = d0 * expf(3.0f * lfl[2 * cl]); dl1 = d1 * expf(3.0f * lfl[2 * cl + 1]);
          f.bDlin[(kf + row) * LO + 2 * cl] = dl0; f.bDlin[(kf + row) * LO + 2 * cl + 1] = dl1;
        } else {
          const float oe = O[cl];
          dz2 = dz2n; p0 = dz2n * oe * 3.0f;
          dl0 = dz2n * expf(3.0f * lfl[cl]);
          f.bDlin[(kf + row) * LO + cl] = dl0;
        }
        f.bDy[(kf + row) * LC + Ch + cl] = dz2;
      }
      Dy[(Ch + cl) * LT + ri] = dz2;
      if (f.affine) {
        Dl[(2 * cl) * LT + ri] = dl0; Dl[(2 * cl + 1) * LT + ri] = dl1;
        Pl[ri * ldp + 2 * cl] = p0; Pl[ri * ldp + 2 * cl + 1] = p1;
      } else {
        Dl[cl * LT + ri] = dl0;
        Pl[ri * ldp + cl] = p0;
      }
    }
    for (int c = Cout + cl; c < Co16; c += 32) Dl[c * LT + ri] = 0.0f;
    for (int c = C + cl; c < C16; c += 32) Dy[c * LT + ri] = 0.0f;
    for (int j = H + cl; j < H16; j += 32) {
#pragma unroll
      for (int g = 0; g < NG; ++g) { Gi[(g * H16 + j) * LT + ri] = 0.0f; Gh[(g * H16 + j) * LT + ri] = 0.0f; }
    }
  }
  __syncthreads();
  if (tid < Cout) {
    float sum = 0.0f;
    for (int i = 0; i < MB; ++i) sum += Pl[i * ldp + tid];
    f.bPlfl[(((long)k * f.N + n) * f.nbt + bt) * Cout + tid] = sum;
  }

  // ---- Q1: d h' = dlin Wfl + dh carried from timestep n + 1; recurrent cell backward
  if (th) {
    const f32x4 acc = mma16_reg<FB_O>(Dl + kq * LT + l15, wq1, nbO);
    const int j = tcol;
    if (j < H) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq * 4 + r;
        const int row = b0 + i;
        float gi_[4] = {0.f, 0.f, 0.f, 0.f}, gh_[4] = {0.f, 0.f, 0.f, 0.f}, cy = 0.0f;
        if (row < B) {
          const float dhn = acc[r] + sdhf[r];
          const float g0 = sg[r][0], g1 = sg[r][1], g2 = sg[r][2], g3 = sg[r][3];
          if (NG == 3) {
            const float rr = g0, uu = g1, nn = g2, ghn = g3;
            const float hp = shp[r];
            const float du = dhn * (hp - nn);
            const float dn = dhn * (1.0f - uu);
            cy = dhn * uu;
            const float dan = dn * (1.0f - nn * nn);
            const float dau = du * uu * (1.0f - uu);
            const float dar = dan * ghn * rr * (1.0f - rr);
            gi_[0] = dar; gi_[1] = dau; gi_[2] = dan;
            gh_[0] = dar; gh_[1] = dau; gh_[2] = dan * rr;
          } else {
            const float ii = g0, ff = g1, gg = g2, oo = g3;
            const float tcv = tanhf_(sc2[r]);
            const float cp = scp[r];
            const float dcf = sdcf[r];
            const float dc2 = dhn * oo * (1.0f - tcv * tcv) + dcf;
            gi_[0] = dc2 * gg * ii * (1.0f - ii);
            gi_[1] = dc2 * cp * ff * (1.0f - ff);
            gi_[2] = dc2 * ii * (1.0f - gg * gg);
            gi_[NG - 1] = dhn * tcv * oo * (1.0f - oo);
#pragma unroll
            for (int g = 0; g < NG; ++g) gh_[g] = gi_[g];
            f.bDc[(kf + row) * H + j] = dc2 * ff;
          }
          float* go = f.bDgi + (kf + row) * G + j;
          float* ho = f.bDgh + (kf + row) * G + j;
#pragma unroll
          for (int g = 0; g < NG; ++g) { go[g * H] = gi_[g]; ho[g * H] = gh_[g]; }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) { Gi[(g * H16 + j) * LT + i] = gi_[g]; Gh[(g * H16 + j) * LT + i] = gh_[g]; }
        Cy[j * LT + i] = cy;
      }
    }
  }
  // weights of Q3 (dy W^T) for this wave's channel tile: in flight under Q2
  f32x4 wq3[FB_C];
  load_frag<FB_C>(wq3, f.pWt + (long)k * C16 * C16, C16, tcol, kq, nbC, tc);
  __syncthreads();

  // ---- Q2: d z1 = dgi W_ih[:, :Ch] + pass-through (z-tile waves, before the barrier: Q3 needs it);
  //          d h_prev = dgh W_hh + carry (to timestep n - 1; not needed inside this cell)
  auto dh_prev_tile = [&](const f32x4 (&w)[NG][FB_H]) {
    if (n > 0 && th) {
      const f32x4 acc = mma16_reg_gates<NG, FB_H>(Gh + kq * LT + l15, H16 * LT, w, nbH);
      const int j = tcol;
      if (j < H) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = kq * 4 + r;
          const int row = b0 + i;
          if (row < B) f.bDh[(kf + row) * H + j] = acc[r] + Cy[j * LT + i];
        }
      }
    }
  };
  if (tz) {
    const f32x4 acc = mma16_reg_gates<NG, FB_H>(Gi + kq * LT + l15, H16 * LT, wq2, nbH);
    const int c = tcol;
    if (c < Ch) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq * 4 + r;
        const int row = b0 + i;
        float v = 0.0f;
        if (row < B) {
          v = acc[r] + sdxo[r];
          f.bDy[(kf + row) * LC + c] = v;
        }
        Dy[c * LT + i] = v;
      }
    }
    // now fetch this wave's W_hh slice for its d h_prev tile (runs after Q3)
#pragma unroll
    for (int g = 0; g < NG; ++g) load_frag<FB_H>(wq2[g], f.bwh + ((long)k * NG + g) * H16 * H16, H16, tcol, kq, nbH, n > 0 && th);
  } else {
    dh_prev_tile(wq2);
  }
  __syncthreads();

  // ---- Q3: d a = dy W^T ; actnorm backward ; d x_in to flow step k - 1
  if (tc) {
    const f32x4 acc = mma16_reg<FB_C>(Dy + kq * LT + l15, wq3, nbC);
    const int c = tcol;
    float sl = 0.0f, sb = 0.0f;
    if (c < C) {
      const float es = expf(f.p.an_logs[(long)k * C + c]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = b0 + kq * 4 + r;
        if (row < B) {
          const float da = acc[r];
          sl += da * sa[r];
          sb += da * es;
          if (k > 0) f.bDx[(kf + row) * LC + c] = da * es;
        }
      }
    }
    sl += __shfl_xor(sl, 16, 64); sl += __shfl_xor(sl, 32, 64);
    sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
    if (kq == 0 && c < C) {
      float* pan = f.bPan + (((long)k * f.N + n) * f.nbt + bt) * 2 * C;
      pan[c] = sl; pan[C + c] = sb;
    }
  }
  if (tz) dh_prev_tile(wq2);
}

// Backward twin of flow_pipe_fwd_kernel: workgroup (k, bt) keeps flow step k's backward weights, walks n = N-1 .. 0, carries
// d h (and the LSTM's d c) from timestep n + 1 in registers and receives d x of flow step k + 1 through the same write-through
// hand-off (bDx tile + progress counter). The z-tile waves need two weight slices (W_ih[:, :Ch] and W_hh): the second one is
// re-read from L2 every timestep, requested before the wait.
template <int NG, bool X3>
__global__ __launch_bounds__(NT) void flow_pipe_bwd_kernel(FlowK f) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15_0 = lane & 15, kq_0 = lane >> 4;
  const int ri_0 = tid >> 5, cl_0 = tid & 31;
  __shared__ int s_id, s_ok, s_rdy;
  if (tid == 0) s_id = (int)atomicAdd(f.pipe, 1u);
  __syncthreads();
  const int nbt = f.nbt;
  const int kk = s_id / nbt, bt = s_id - kk * nbt;
  if (kk >= f.Ks) return;
  const int k = f.Ks - 1 - kk;   // tickets in arrival order: (k + 1, bt) always started before (k, bt)
  unsigned* abort_w = f.pipe + 1;
  unsigned* prog = f.pipe + PIPE_WALK_HDR;
  const bool fenced = f.pipe_fence != 0;
  const int b0 = bt * MB;
  const int B = f.B, C = f.C, H = f.H, Ch = f.Ch, C2 = f.C2, Cout = f.Cout, G = f.G;
  const int C16 = f.C16, Ch16 = f.Ch16, H16 = f.H16, Co16 = f.Co16;
  const long LC = f.ldc, LO = f.ldo;   // stash row strides
  const CarveFB cv = carve_fast_bwd(C16, H16, Co16, Cout, NG);
  float* Dl = flow_smem + cv.Dl;
  float* Gi = flow_smem + cv.Gi;
  float* Gh = Gi + NG * H16 * LT;
  // bf16 x 3: the Gi / Gh regions hold bf16 hi + lo images instead ([16][ldx], see x3_put); 64 (NG H16 + 8) <= 68 NG H16 bytes
  const int ldx = NG * H16 + 8;
  __bf16* GiH = reinterpret_cast<__bf16*>(Gi);
  __bf16* GiL = GiH + MB * ldx;
  __bf16* GhH = reinterpret_cast<__bf16*>(Gh);
  __bf16* GhL = GhH + MB * ldx;
  float* Dy = flow_smem + cv.Dy;
  float* Cy = flow_smem + cv.Cy;
  float* Pl = flow_smem + cv.Pl;
  const int ldp = Cout + 1;
  const int nbC = C16 >> 4, nbH = H16 >> 4, nbO = Co16 >> 4;
  const bool last = k == f.Ks - 1;
  const float gz = f.gscale / LN2_F;   // d loss / d z = z * gz   (prior term)
  const float dl = -f.gscale / LN2_F;  // d loss / d logdet
  const float dxs = last ? gz : 1.0f;

  // which tiles this wave owns: hidden tile `wave` (Q1, Q2), z tile `wave` (Q2, waves < Ch16/16), channel tile `wave` (Q3)
  const bool th = wave * 16 < H, tz = wave * 16 < Ch, tc = wave * 16 < C;
  const int tcol_0 = wave * 16 + l15_0;
  // ---- weights of Q1 (dlin Wfl: K = Cout) for this wave's hidden tile
  f32x4 wq1[FB_O];
  {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < FB_O; ++b) wq1[b] = zero4;
  }
  load_frag<FB_O>(wq1, f.bwfl + (long)k * Co16 * H16, H16, tcol_0, kq_0, nbO, th);
  // bf16 x 3: Q1's product too (dlin Wfl, K = Cout): the fragments packed into 32-k blocks, the A operand d lin as a bf16
  // hi / lo image [16][ldxd] that Q0 writes (x3_put, slot order x3_pos) behind the regions of the exact path
  X3Frag wq1x[FB_O / 2];
  const int nbO2 = (nbO + 1) >> 1, ldxd = 32 * nbO2 + 8;
  __bf16* DlH = reinterpret_cast<__bf16*>(flow_smem + ((cv.total + 3) & ~3));
  __bf16* DlL = DlH + MB * ldxd;
  if constexpr (X3) {
#pragma unroll
    for (int b2 = 0; b2 < FB_O / 2; ++b2) wq1x[b2] = x3_pack(wq1[2 * b2], wq1[2 * b2 + 1]);
    for (int q = tid; q < 2 * MB * ldxd; q += NT) DlH[q] = (__bf16)0.0f;
  }
  // W_hh slice of this wave's hidden tile: resident, except in the z-tile waves, which need two slices (W_ih[:, :Ch] for
  // d z1, then W_hh) and stream both through the same registers every timestep, as the per-diagonal kernel does
  f32x4 wq2[X3 ? 1 : NG][X3 ? 1 : FB_H];      // exact f32 form
  X3Frag wq2x[X3 ? NG : 1][FB_H / 2];         // bf16 x 3 form: pre-split 32-k fragments (flow_prep_x3_kernel)
  const int nB = H16 >> 5, nbH2 = (nbH + 1) >> 1;
  const long perH = (long)NG * nB * H16 * 4, perZ = (long)NG * nB * Ch16 * 4;   // uint4 entries per plane and flow step
  const uint4* xh = f.xbwh + (long)k * 2 * perH;
  const uint4* xz = f.xbwz + (long)k * 2 * perZ;
  if (!tz) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (X3) x3_load<FB_H / 2>(wq2x[g], xh, perH, g, nB, H16, tcol_0, kq_0, nbH2, th);
      else load_frag<FB_H>(wq2[g], f.bwh + ((long)k * NG + g) * H16 * H16, H16, tcol_0, kq_0, nbH, th);
    }
  }
  // weights of Q3 (dy W^T) for this wave's channel tile
  f32x4 wq3[FB_C];
  load_frag<FB_C>(wq3, f.pWt + (long)k * C16 * C16, C16, tcol_0, kq_0, nbC, tc);
  const float es_an = tcol_0 < C ? expf(f.p.an_logs[(long)k * C + tcol_0]) : 0.0f;
  float dhc[4] = {0.f, 0.f, 0.f, 0.f}, dcc[4] = {0.f, 0.f, 0.f, 0.f};   // d h / d c carried from timestep n + 1 (registers)

  if constexpr (X3) {   // padding columns (hidden units >= H) are never written: zero the images once
    for (int q = tid; q < 2 * MB * ldx; q += NT) { GiH[q] = (__bf16)0.0f; GhH[q] = (__bf16)0.0f; }
    __syncthreads();
  }
  // b_ih / b_hh gradients = column sums of dgi / dgh over all frames: every thread of the Q1 tile sums its own hidden unit
  // over its 4 rows and all timesteps here (fixed order), instead of two 350 MB passes over the stash afterwards
  float bsi[NG], bsh[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { bsi[g] = 0.0f; bsh[g] = 0.0f; }
  // look-ahead acquire, as in the forward walk: thread 0 peeks at flow step k + 1's progress word for cell n - 1 at the start
  // of Q3 of cell n; if it is published, every thread fetches its elements of that d x tile behind the publish barrier
  bool have_next = false;
  float nx_dxo[4] = {0.f, 0.f, 0.f, 0.f}, nx_dz2 = 0.0f;
  for (int n = f.N - 1; n >= 0; --n) {
    // lane coordinates laundered per iteration: otherwise every per-lane stash address (a dozen arrays x 4 rows, 64-bit) is
    // hoisted out of the timestep loop and the kernel spills ~150 VGPRs
    int l15 = l15_0, kq = kq_0, ri = ri_0, cl = cl_0, tcol = tcol_0;
    asm volatile("" : "+v"(l15), "+v"(kq), "+v"(ri), "+v"(cl), "+v"(tcol));

  const long fr = (long)n * B;
  const long kf = (long)k * f.F + fr;
  const float* dxo = last ? f.sX + kf * LC : f.bDx + (kf + f.F) * LC;
  PIPE_STAMP(1, 0);
  // ---- forward-stash operands of Q1 / Q2 / Q3 (written before this launch): in flight under the wait for flow step k + 1
  float sg[4][4], shp[4], sdhf[4], sc2[4], scp[4], sdcf[4], sdxo[4], sa[4];
  {
    const int j = tcol < H ? tcol : 0, cc = tcol < C ? tcol : 0;
    const bool hasp = n > 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = min(b0 + kq * 4 + r, B - 1);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(f.sG + (kf + row) * 4 * H + 4 * j);
      sg[r][0] = g4[0]; sg[r][1] = g4[1]; sg[r][2] = g4[2]; sg[r][3] = g4[3];
      shp[r] = hasp ? f.sH[(kf - B + row) * H + j] : 0.0f;
      sdhf[r] = dhc[r];
      if (NG == 4) {
        sc2[r] = f.sC[(kf + row) * H + j];
        scp[r] = hasp ? f.sC[(kf - B + row) * H + j] : 0.0f;
        sdcf[r] = dcc[r];
      }
      sa[r] = f.sA[(kf + row) * LC + cc];
    }
  }
  if (tz) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (X3) x3_load<FB_H / 2>(wq2x[g], xz, perZ, g, nB, Ch16, tcol, kq, nbH2, true);
      else load_frag<FB_H>(wq2[g], f.bwz + ((long)k * NG + g) * H16 * Ch16, Ch16, tcol, kq, nbH, true);
    }
  }
  // Q0's forward-stash operands (o of the coupling net, z2) for this thread's element
  float q_oe = 0.0f, q_oo = 0.0f, q_z2 = 0.0f;
  if (cl < C2 && b0 + ri < B) {
    const float* O = f.sO + (kf + b0 + ri) * LO;
    if (f.affine) {
      q_oe = O[2 * cl]; q_oo = O[2 * cl + 1];
      q_z2 = f.sY[(kf + b0 + ri) * LC + Ch + cl];
    } else {
      q_oe = O[cl];
    }
  }
  if (!last && !have_next && !pipe_acquire(prog + ((k + 1) * nbt + bt) * PIPE_STRIDE, (unsigned)(f.N - n), abort_w, tid, &s_ok, fenced)) break;
  {
    const int cz = tcol < Ch ? tcol : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = min(b0 + kq * 4 + r, B - 1);
      sdxo[r] = (last ? dxo[row * LC + cz] : (have_next ? nx_dxo[r] : ld_tile(dxo + row * LC + cz, fenced))) * dxs;
    }
  }

  PIPE_STAMP(1, 1);
  // ---- Q0: coupling backward + LinearZeros scale; zero the k padding of the LDS operands
  {
    const int row = b0 + ri;
    const bool rok = row < B;
    const float* lfl = f.p.l_fl + (long)k * Cout;
    float dz2 = 0.0f, dl0 = 0.0f, dl1 = 0.0f, p0 = 0.0f, p1 = 0.0f;
    if (cl < C2) {
      if (rok) {
        const float dz2n = (last ? dxo[(long)row * LC + Ch + cl]
                                 : (have_next ? nx_dz2 : ld_tile(dxo + (long)row * LC + Ch + cl, fenced))) * dxs;
        if (f.affine) {
          const float oe = q_oe, oo = q_oo;
          const float sraw = sigmoidf_(oo + 2.0f);
          const float sc = fmaxf(sraw, f.eps);
          const float z2 = q_z2;
          dz2 = dz2n * sc;
          const float dsc = dz2n * (z2 + oe) + dl / sc;
          const float dsr = sraw >= f.eps ? dsc : 0.0f;
          const float d0 = dz2;                             // d o_even (shift)
          const float d1 = dsr * sraw * (1.0f - sraw);      // d o_odd
          p0 = d0 * oe * 3.0f; p1 = d1 * oo * 3.0f;
          dl0 = d0 * expf(3.0f * lfl[2 * cl]); dl1 = d1 * expf(3.0f * lfl[2 * cl + 1]);
          f.bDlin[(kf + row) * LO + 2 * cl] = dl0; f.bDlin[(kf + row) * LO + 2 * cl + 1] = dl1;
        } else {
          const float oe = q_oe;
          dz2 = dz2n; p0 = dz2n * oe * 3.0f;
          dl0 = dz2n * expf(3.0f * lfl[cl]);
          f.bDlin[(kf + row) * LO + cl] = dl0;
        }
        f.bDy[(kf + row) * LC + Ch + cl] = dz2;
      }
      Dy[(Ch + cl) * LT + ri] = dz2;
      if (f.affine) {
        if constexpr (X3) {
          x3_put(DlH, DlL, ri * ldxd + x3_pos(2 * cl), dl0);
          x3_put(DlH, DlL, ri * ldxd + x3_pos(2 * cl + 1), dl1);
        } else {
          Dl[(2 * cl) * LT + ri] = dl0; Dl[(2 * cl + 1) * LT + ri] = dl1;
        }
        Pl[ri * ldp + 2 * cl] = p0; Pl[ri * ldp + 2 * cl + 1] = p1;
      } else {
        if constexpr (X3) x3_put(DlH, DlL, ri * ldxd + x3_pos(cl), dl0);
        else Dl[cl * LT + ri] = dl0;
        Pl[ri * ldp + cl] = p0;
      }
    }
    if (!X3) for (int c = Cout + cl; c < Co16; c += 32) Dl[c * LT + ri] = 0.0f;
    for (int c = C + cl; c < C16; c += 32) Dy[c * LT + ri] = 0.0f;
    for (int j = H + cl; j < H16; j += 32) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (!X3) { Gi[(g * H16 + j) * LT + ri] = 0.0f; Gh[(g * H16 + j) * LT + ri] = 0.0f; }
    }
  }
  __syncthreads();
  PIPE_STAMP(1, 2);
  if (tid < Cout) {
    float sum = 0.0f;
    for (int i = 0; i < MB; ++i) sum += Pl[i * ldp + tid];
    f.bPlfl[(((long)k * f.N + n) * f.nbt + bt) * Cout + tid] = sum;
  }

  // ---- Q1: d h' = dlin Wfl + dh carried from timestep n + 1; recurrent cell backward
  if (th) {
    f32x4 acc;
    if constexpr (X3) {
      const __bf16* rh = DlH + l15 * ldxd + 8 * kq;
      const __bf16* rl = DlL + l15 * ldxd + 8 * kq;
      f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b2 = 0; b2 < FB_O / 2; ++b2)
        if (b2 < nbO2) {
          X3Frag av;
          av.hi = *reinterpret_cast<const fbf16x8*>(rh + b2 * 32);
          av.lo = *reinterpret_cast<const fbf16x8*>(rl + b2 * 32);
          if (b2 & 1) o = x3_mma(av, wq1x[b2], o);
          else e = x3_mma(av, wq1x[b2], e);
        }
      acc = e + o;
    } else {
      acc = mma16_reg<FB_O>(Dl + kq * LT + l15, wq1, nbO);
    }
    const int j = tcol;
    if (j < H) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq * 4 + r;
        const int row = b0 + i;
        float gi_[4] = {0.f, 0.f, 0.f, 0.f}, gh_[4] = {0.f, 0.f, 0.f, 0.f}, cy = 0.0f;
        if (row < B) {
          const float dhn = acc[r] + sdhf[r];
          const float g0 = sg[r][0], g1 = sg[r][1], g2 = sg[r][2], g3 = sg[r][3];
          if (NG == 3) {
            const float rr = g0, uu = g1, nn = g2, ghn = g3;
            const float hp = shp[r];
            const float du = dhn * (hp - nn);
            const float dn = dhn * (1.0f - uu);
            cy = dhn * uu;
            const float dan = dn * (1.0f - nn * nn);
            const float dau = du * uu * (1.0f - uu);
            const float dar = dan * ghn * rr * (1.0f - rr);
            gi_[0] = dar; gi_[1] = dau; gi_[2] = dan;
            gh_[0] = dar; gh_[1] = dau; gh_[2] = dan * rr;
          } else {
            const float ii = g0, ff = g1, gg = g2, oo = g3;
            const float tcv = tanhf_(sc2[r]);
            const float cp = scp[r];
            const float dcf = sdcf[r];
            const float dc2 = dhn * oo * (1.0f - tcv * tcv) + dcf;
            gi_[0] = dc2 * gg * ii * (1.0f - ii);
            gi_[1] = dc2 * cp * ff * (1.0f - ff);
            gi_[2] = dc2 * ii * (1.0f - gg * gg);
            gi_[NG - 1] = dhn * tcv * oo * (1.0f - oo);
#pragma unroll
            for (int g = 0; g < NG; ++g) gh_[g] = gi_[g];
            dcc[r] = dc2 * ff;
          }
          float* go = f.bDgi + (kf + row) * G + j;
          float* ho = f.bDgh + (kf + row) * G + j;
#pragma unroll
          // (GRU: d r, d z on the hidden side equal the input side's. Storing only the n block of dgh and taking the other two
          // rows of the W_hh gradient from dgi was measured in round 4: 8 of 24 dword stores per lane and cell less, but the
          // gradient then needs two products over the frames instead of one - whole step 7.73 / 7.69 ms against 7.67 / 7.64: removed)
          for (int g = 0; g < NG; ++g) {
            if (X3 && f.g16) {   // (rounded as the products' operand split rounds its hi part: to nearest even)
              reinterpret_cast<__bf16*>(f.bDgi)[(kf + row) * G + j + g * H] = (__bf16)gi_[g];
              reinterpret_cast<__bf16*>(f.bDgh)[(kf + row) * G + j + g * H] = (__bf16)gh_[g];
            } else {
              go[g * H] = gi_[g]; ho[g * H] = gh_[g];
            }
            bsi[g] += gi_[g]; bsh[g] += gh_[g];
          }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if constexpr (X3) {
            const int at = i * ldx + g * H16 + j;   // natural column order (weights: flow_prep_x3_kernel)
            x3_put(GiH, GiL, at, gi_[g]);
            x3_put(GhH, GhL, at, gh_[g]);
          } else {
            Gi[(g * H16 + j) * LT + i] = gi_[g]; Gh[(g * H16 + j) * LT + i] = gh_[g];
          }
        }
        Cy[j * LT + i] = cy;
      }
    }
  }
  __syncthreads();
  PIPE_STAMP(1, 3);

  // ---- Q2: d z1 = dgi W_ih[:, :Ch] + pass-through (z-tile waves, before the barrier: Q3 needs it);
  //          d h_prev = dgh W_hh + carry (to timestep n - 1; not needed inside this cell)
  auto dh_prev_tile = [&]() {
    if (n > 0 && th) {
      f32x4 acc;
      if constexpr (X3) acc = x3_mma_gates_img<NG, FB_H / 2>(GhH + l15 * ldx + 8 * kq, GhL + l15 * ldx + 8 * kq, H16, wq2x, nbH2);
      else acc = mma16_reg_gates<NG, FB_H>(Gh + kq * LT + l15, H16 * LT, wq2, nbH);
      const int j = tcol;
      if (j < H) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = kq * 4 + r;
          const int row = b0 + i;
          (void)row;
          dhc[r] = acc[r] + Cy[j * LT + i];
        }
      }
    }
  };
  if (tz) {
    f32x4 acc;
    if constexpr (X3) acc = x3_mma_gates_img<NG, FB_H / 2>(GiH + l15 * ldx + 8 * kq, GiL + l15 * ldx + 8 * kq, H16, wq2x, nbH2);
    else acc = mma16_reg_gates<NG, FB_H>(Gi + kq * LT + l15, H16 * LT, wq2, nbH);
    const int c = tcol;
    if (c < Ch) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kq * 4 + r;
        const int row = b0 + i;
        float v = 0.0f;
        if (row < B) {
          v = acc[r] + sdxo[r];
          f.bDy[(kf + row) * LC + c] = v;
        }
        Dy[c * LT + i] = v;
      }
    }
    // now fetch this wave's W_hh slice for its d h_prev tile (runs after Q3)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (X3) x3_load<FB_H / 2>(wq2x[g], xh, perH, g, nB, H16, tcol, kq, nbH2, n > 0 && th);
      else load_frag<FB_H>(wq2[g], f.bwh + ((long)k * NG + g) * H16 * H16, H16, tcol, kq, nbH, n > 0 && th);
    }
  } else {
    dh_prev_tile();
  }
  __syncthreads();
  PIPE_STAMP(1, 4);

  // ---- Q3: d a = dy W^T ; actnorm backward ; d x_in to flow step k - 1
  const bool peek = !last && !fenced && n > 0;
  unsigned pk = 0u;
  if (peek && tid == 0) pk = ld_agent(prog + ((k + 1) * nbt + bt) * PIPE_STRIDE);
  if constexpr (X3) {
    // dgi of this cell as operand planes (include/lfi.h) for the two products that consume it - dpre = dgi W_c sums over gate
    // columns, dW_c = dgi^T c over frames: one set of planes serves both - straight from the bf16 hi / lo LDS images Q1 left: one
    // 16-byte LDS read and one 16-byte store per 8 values and plane, by the waves that own no channel tile and idle through Q3
    if (f.bDgiR && !tc) {
      const int ntc = (C + 15) >> 4;                 // channel-tile waves are waves 0 .. ntc - 1
      const int ne = NT - ntc * 64, et = tid - ntc * 64;
      const long gr0 = kf + b0;                      // row of the (Ks F x G) matrix: flow step k, frame n B + b0 (+ i)
      const int nktG = G >> 4;
      char* rbase = reinterpret_cast<char*>(f.bDgiR) + ((gr0 >> 5) * nktG * 2) * 1024;
      const int r0 = (int)(gr0 & 16);                // this tile's 16 rows inside the 32-row plane tile
      // u = 32 ct + 2 i + hp: the chunk that lies hp-th in row r0 + i of block (row tile, ct): consecutive threads write consecutive
      // 16 bytes, 32 threads the 512 contiguous bytes this tile owns of a block (a row's two chunks in one store instruction:
      // the per-half mapping left every store at a 32-byte stride and the backward walk 0.11 ms longer)
      for (int u = et; u < 2 * G; u += ne) {
        const int hp = u & 1, i = (u >> 1) & 15, ct = u >> 5;
        const int h = hp ^ (((r0 + i) >> 3) & 1);    // which 8 columns lie there (lfi_u_plane_offset)
        const int so = i * ldx + ct * 16 + h * 8;
        char* dst = rbase + (long)ct * 2048 + (r0 + i) * 32 + hp * 16;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(GiH + so);
        if (!f.dgi_hi_only) *reinterpret_cast<uint4*>(dst + 1024) = *reinterpret_cast<const uint4*>(GiL + so);
      }
    }
  }
  if (tc) {
    const f32x4 acc = mma16_reg<FB_C>(Dy + kq * LT + l15, wq3, nbC);
    const int c = tcol;
    float sl = 0.0f, sb = 0.0f;
    if (c < C) {
      const float es = es_an;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = b0 + kq * 4 + r;
        if (row < B) {
          const float da = acc[r];
          sl += da * sa[r];
          sb += da * es;
          if (k > 0) st_sc1(f.bDx + (kf + row) * LC + c, da * es);
        }
      }
    }
    sl += __shfl_xor(sl, 16, 64); sl += __shfl_xor(sl, 32, 64);
    sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
    if (kq == 0 && c < C) {
      float* pan = f.bPan + (((long)k * f.N + n) * f.nbt + bt) * 2 * C;
      pan[c] = sl; pan[C + c] = sb;
    }
  }
  PIPE_STAMP(1, 5);
  // (pipe_publish, with the outcome of the peek riding on its barrier)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (peek && tid == 0) s_rdy = pk >= (unsigned)(f.N - n) + 1u ? 1 : 0;
  __syncthreads();
  if (k > 0 && tid == 0) st_agent(prog + (k * nbt + bt) * PIPE_STRIDE, (unsigned)(f.N - n));
  have_next = peek && s_rdy != 0;
  if (have_next) {
    const float* dxn = f.bDx + (kf - B + f.F) * LC;   // cell n - 1 of flow step k + 1
    const int cz = tcol < Ch ? tcol : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) nx_dxo[r] = ld_tile(dxn + (long)min(b0 + kq * 4 + r, B - 1) * LC + cz, false);
    nx_dz2 = (cl < C2 && b0 + ri < B) ? ld_tile(dxn + (long)(b0 + ri) * LC + Ch + cl, false) : 0.0f;
  }
  PIPE_STAMP(1, 6);
  // the z-tile waves' own d h_prev tile is only needed by the next timestep of THIS workgroup: after the hand-off, off the
  // pipeline's latency path (it reads Gh / Cy, which the next timestep rewrites only behind its first barrier)
  if (tz) dh_prev_tile();
  PIPE_STAMP(1, 7);
  }  // timestep loop
  if (th) {   // (an abandoned walk leaves these incomplete: flow_pipe_poison_kernel overwrites them with NaN)
    float* pb = f.bPbias + ((long)k * nbt + bt) * 2 * G;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float vi = bsi[g], vh = bsh[g];
      vi += __shfl_xor(vi, 16, 64); vi += __shfl_xor(vi, 32, 64);
      vh += __shfl_xor(vh, 16, 64); vh += __shfl_xor(vh, 32, 64);
      if ((lane >> 4) == 0 && tcol_0 < H) { pb[g * H + tcol_0] = vi; pb[G + g * H + tcol_0] = vh; }
    }
  }
}

// A backward walk that gave up must not pass for a result: NaN into one frame of every flow step's stashed gradients, which
// every parameter gradient (and, through dgi, the encoder gradients) sums over.
__global__ __launch_bounds__(64) void flow_pipe_poison_kernel(FlowK f) {
  if (f.pipe[1] == 0u) return;
  const float nan = __builtin_nanf("");
  for (long i = threadIdx.x; i < (long)f.Ks * 2 * f.G; i += 64) f.bPbias[(i / (2 * f.G)) * f.nbt * 2 * f.G + i % (2 * f.G)] = nan;
  for (int k = threadIdx.x; k < f.Ks; k += 64) {
    for (long fr = 0; fr < f.F; fr += (f.F > 1 ? f.F - 1 : 1)) {   // first and last frame (the W_hh gradient skips timestep 0)
      f.bDlin[((long)k * f.F + fr) * f.ldo] = nan;
      if (f.g16) {   // the dgi | dgh rows are bf16 arrays of the same shapes (an fp32 store here landed on two bf16
                     // elements of another flow step's frame, or past the bf16 region)
        reinterpret_cast<__bf16*>(f.bDgi)[((long)k * f.F + fr) * f.G] = (__bf16)nan;
        reinterpret_cast<__bf16*>(f.bDgh)[((long)k * f.F + fr) * f.G] = (__bf16)nan;
      } else {
        f.bDgi[((long)k * f.F + fr) * f.G] = nan;
        f.bDgh[((long)k * f.F + fr) * f.G] = nan;
      }
      f.bDy[((long)k * f.F + fr) * f.ldc] = nan;
    }
    // dgi as operand planes (the dgi^T c and dpre products read these, not the rows): element (row k F, column 0) of the hi plane
    if (f.bDgiR && f.F % 32 == 0) f.bDgiR[(((long)k * f.F / 32) * (f.G / 16) * 2) * 512] = (__bf16)nan;
  }
}

// Zero-padded fragment-order weight images for the register-resident cells (layout: flow_img_index).
// which: 0 pW (k = input channel, col = output channel of W), 1 pWt, 2 pwz (k = z channel, col = g*H16 + hidden),
// 3 pwh (k = hidden in), 4 pwfl (k = hidden, col = output), 5 bwfl (k = output, col = hidden),
// 6 bwh [g] (k = hidden of gate g, col = hidden), 7 bwz [g] (k = hidden of gate g, col = z channel)
__global__ __launch_bounds__(256) void flow_prep_pad_kernel(FlowK f, float* pW, float* pWt, float* pwz, float* pwh, float* pwfl,
                                                            float* bwfl, float* bwh, float* bwz, float* pWinv) {
  const int k = blockIdx.y, which = blockIdx.z;
  const int C = f.C, H = f.H, Ch = f.Ch, Cout = f.Cout, I = f.I, NG = f.NG;
  const int C16 = f.C16, Ch16 = f.Ch16, H16 = f.H16, Co16 = f.Co16;
  const float* W = f.W + (long)k * C * C;
  const float* wih = f.p.w_ih + (long)k * f.G * I;
  const float* whh = f.p.w_hh + (long)k * f.G * H;
  const float* wfl = f.p.w_fl + (long)k * Cout * H;
  int K, J, nimg = 1;   // rows (padded), columns (padded), images per flow step
  float* dst;
  switch (which) {
    case 0: K = C16; J = C16; dst = pW; break;
    case 1: K = C16; J = C16; dst = pWt; break;
    case 2: K = Ch16; J = NG * H16; dst = pwz; break;
    case 3: K = H16; J = NG * H16; dst = pwh; break;
    case 4: K = H16; J = Co16; dst = pwfl; break;
    case 5: K = Co16; J = H16; dst = bwfl; break;
    case 6: K = H16; J = H16; nimg = NG; dst = bwh; break;
    case 7: K = H16; J = Ch16; nimg = NG; dst = bwz; break;
    default: K = C16; J = C16; dst = pWinv; break;   // 8: reverse weight (only when lfi_flow_prep built it)
  }
  if (which == 8 && !pWinv) return;
  const long per = (long)K * J;
  dst += (long)k * nimg * per;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < nimg * per; idx += (long)gridDim.x * 256) {
    const int g = (int)(idx / per);
    const long q = idx - g * per;
    const int kk = (int)(q / J), col = (int)(q - (long)kk * J);
    float v = 0.0f;
    switch (which) {
      case 0: if (kk < C && col < C) v = W[kk * C + col]; break;
      case 1: if (kk < C && col < C) v = W[col * C + kk]; break;
      case 2: { const int gg = col / H16, j = col - gg * H16; if (kk < Ch && j < H) v = wih[((long)gg * H + j) * I + kk]; } break;
      case 3: { const int gg = col / H16, j = col - gg * H16; if (kk < H && j < H) v = whh[((long)gg * H + j) * H + kk]; } break;
      case 4: if (kk < H && col < Cout) v = wfl[(long)col * H + kk]; break;
      case 5: if (kk < Cout && col < H) v = wfl[(long)kk * H + col]; break;
      case 6: if (kk < H && col < H) v = whh[((long)g * H + kk) * H + col]; break;
      case 7: if (kk < H && col < Ch) v = wih[((long)g * H + kk) * I + col]; break;
      default: if (kk < C && col < C) v = f.Winv[(long)k * C * C + kk * C + col]; break;
    }
    dst[g * per + flow_img_index(kk, col, J)] = v;
  }
}

// bf16 hi / lo fragment images of the backward recurrent weights for the bf16 x 3 walk. which 0: bwh (k = hidden of gate g,
// col = hidden: W_hh[g*H + k][col]); 1: bwz (col = z channel: W_ih[g*H + k][col]). Entry (g, B, col, kq) holds the eight k of
// k = 32 B + 8 kq + i, the natural operand order of v_mfma_f32_16x16x32_bf16: the backward walk's d(gate) LDS images are then
// plain row-major [16 rows][gate columns] and double as the source of the dgi operand planes it emits (lfi_flow_seq_bwd_planes).
__global__ __launch_bounds__(256) void flow_prep_x3_kernel(FlowK f, uint4* xbwh, uint4* xbwz) {
  const int k = blockIdx.y, which = blockIdx.z;
  const int H = f.H, Ch = f.Ch, I = f.I, NG = f.NG, H16 = f.H16;
  const int J = which ? f.Ch16 : H16, nB = H16 >> 5;
  const float* whh = f.p.w_hh + (long)k * f.G * H;
  const float* wih = f.p.w_ih + (long)k * f.G * I;
  const long per = (long)NG * nB * J * 4;              // uint4 entries per flow step and plane
  uint4* hi = (which ? xbwz : xbwh) + (long)k * 2 * per;
  uint4* lo = hi + per;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int kq = (int)(idx & 3);
    long r = idx >> 2;
    const int col = (int)(r % J); r /= J;
    const int B = (int)(r % nB), g = (int)(r / nB);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = 32 * B + 8 * kq + i;
      float x = 0.0f;
      if (kk < H) {
        if (which == 0) { if (col < H) x = whh[((long)g * H + kk) * H + col]; }
        else if (col < Ch) x = wih[((long)g * H + kk) * I + col];
      }
      v[i] = x;
    }
    uint4 h, l;
    x3_split2(v[0], v[1], &h.x, &l.x);
    x3_split2(v[2], v[3], &h.y, &l.y);
    x3_split2(v[4], v[5], &h.z, &l.z);
    x3_split2(v[6], v[7], &h.w, &l.w);
    hi[idx] = h;
    lo[idx] = l;
  }
}

// fp16 hi / lo fragment images of the reverse cell's weights (three fp16 products, x3h_*): image `which` 0: pwz (K = Ch16,
// J = NG H16), 1: pwh (H16, NG H16), 2: pwfl (H16, Co16), 3: pWinv (C16, C16) of flow step blockIdx.y. Entry (b, kq, col) is
// x3h_pack of the padded f32 image's entries (2b, kq, col) and (2b + 1, kq, col) - exactly what rev_fast_cell made of them in
// registers in every workgroup of every generated frame (672 conversions per lane in front of a cell's first product).
__global__ __launch_bounds__(256) void flow_prep_x3h_kernel(FlowK f, uint4* hwz, uint4* hwh, uint4* hwfl, uint4* hWinv) {
  const int k = blockIdx.y, which = blockIdx.z;
  const int K16 = which == 0 ? f.Ch16 : (which == 3 ? f.C16 : f.H16);
  const int J = which < 2 ? f.NG * f.H16 : (which == 2 ? f.Co16 : f.C16);
  const float* src = which == 0 ? f.pwz : (which == 1 ? f.pwh : (which == 2 ? f.pwfl : f.pWinv));
  uint4* dst = which == 0 ? hwz : (which == 1 ? hwh : (which == 2 ? hwfl : hWinv));
  const long per = (long)(K16 >> 5) * 4 * J;            // uint4 entries per flow step and plane
  const f32x4* img = reinterpret_cast<const f32x4*>(src + (long)k * K16 * J);
  uint4* hi = dst + (long)k * 2 * per;
  uint4* lo = hi + per;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < per; idx += (long)gridDim.x * 256) {
    const int col = (int)(idx % J);
    const long r = idx / J;
    const int kq = (int)(r & 3), b = (int)(r >> 2);
    const X3FragH w = x3h_pack(img[((long)(2 * b) * 4 + kq) * J + col], img[((long)(2 * b + 1) * 4 + kq) * J + col]);
    hi[idx] = __builtin_bit_cast(uint4, w.hi);
    lo[idx] = __builtin_bit_cast(uint4, w.lo);
  }
}

// ActNorm2d.forward as a stand-alone module call (glow/modules.py:45-80): out = (x + bias) exp(logs), or its inverse
// x exp(-logs) - bias; dlogdet[0] = +-C * sum(logs) (the x C factor of modules.py:62)
__global__ __launch_bounds__(256) void actnorm_module_kernel(const float* __restrict__ x, long rows, int C, const float* __restrict__ bias,
                                                             const float* __restrict__ logs, int reverse, float* __restrict__ out,
                                                             float* __restrict__ dlogdet) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    out[i] = reverse ? x[i] * expf(-logs[c]) - bias[c] : (x[i] + bias[c]) * expf(logs[c]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && dlogdet) {
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += logs[c];
    dlogdet[0] = (reverse ? -(float)C : (float)C) * s;
  }
}

// ------------------------------------------------------------------------------------------- prep
// One workgroup per flow step: W = P (L*mask + I)(U*mask^T + diag(sign exp(log_s)))  (glow/modules.py:167-173),
// its transpose, and (optionally) the reverse weight U^-1 L^-1 P^-1 with fp64 triangular inverses (:175-177).
__global__ __launch_bounds__(256) void flow_prep_invconv_kernel(FlowK f, float* __restrict__ W, float* __restrict__ Wt,
                                                                float* __restrict__ Winv, double* __restrict__ dscratch,
                                                                float* __restrict__ ldpart) {
  __shared__ int piv;
  __shared__ double logabs;
  const int k = blockIdx.x, C = f.C, tid = threadIdx.x;
  float* Lm = flow_smem;            // C*C
  float* Um = Lm + C * C;           // C*C
  float* Tm = Um + C * C;           // C*C
  const long kc = (long)k * C * C;
  float* Wk = W + kc;
  float* Wtk = Wt + kc;
  if (f.p.inv_w) {
    for (int idx = tid; idx < C * C; idx += 256) {
      const int i = idx / C, j = idx - i * C;
      const float v = f.p.inv_w[kc + idx];
      Wk[idx] = v;
      Wtk[(long)j * C + i] = v;
    }
    // log|det| and inverse by Gauss-Jordan with partial pivoting in fp64 (torch.slogdet / torch.inverse(double))
    double* M = dscratch + (long)k * 2 * C * C;  // C x 2C augmented
    for (int idx = tid; idx < C * C; idx += 256) {
      const int i = idx / C, j = idx - i * C;
      M[(long)i * 2 * C + j] = (double)f.p.inv_w[kc + idx];
      M[(long)i * 2 * C + C + j] = i == j ? 1.0 : 0.0;
    }
    if (tid == 0) logabs = 0.0;
    __syncthreads();
    for (int col = 0; col < C; ++col) {
      if (tid == 0) {
        int best = col;
        double bv = fabs(M[(long)col * 2 * C + col]);
        for (int r = col + 1; r < C; ++r) {
          const double v = fabs(M[(long)r * 2 * C + col]);
          if (v > bv) { bv = v; best = r; }
        }
        piv = best;
        logabs += log(bv);
      }
      __syncthreads();
      const int pr = piv;
      if (pr != col)
        for (int j = tid; j < 2 * C; j += 256) {
          const double t0 = M[(long)col * 2 * C + j];
          M[(long)col * 2 * C + j] = M[(long)pr * 2 * C + j];
          M[(long)pr * 2 * C + j] = t0;
        }
      __syncthreads();
      const double pv = M[(long)col * 2 * C + col];
      __syncthreads();
      for (int j = tid; j < 2 * C; j += 256) M[(long)col * 2 * C + j] /= pv;
      __syncthreads();
      // two-pass elimination: every row's factor M[r][col] is read before any row is updated
      double* fac = dscratch + (long)f.Ks * 2 * C * C + (long)k * C;
      for (int r = tid; r < C; r += 256) fac[r] = (r == col) ? 0.0 : M[(long)r * 2 * C + col];
      __syncthreads();
      for (int idx = tid; idx < C * 2 * C; idx += 256) {
        const int r = idx / (2 * C), j = idx - r * 2 * C;
        M[(long)r * 2 * C + j] -= fac[r] * M[(long)col * 2 * C + j];
      }
      __syncthreads();
    }
    if (Winv)
      for (int idx = tid; idx < C * C; idx += 256) {
        const int i = idx / C, j = idx - i * C;
        Winv[kc + idx] = (float)M[(long)i * 2 * C + C + j];
      }
    if (tid == 0) {
      float s = 0.0f;
      for (int c = 0; c < C; ++c) s += f.p.an_logs[(long)k * C + c];
      ldpart[k] = (float)C * (s + (float)logabs);
    }
    return;
  }
  const float* l = f.p.inv_l + kc;
  const float* u = f.p.inv_u + kc;
  const float* P = f.p.inv_p + kc;
  const float* sg = f.p.inv_sign + (long)k * C;
  const float* ls = f.p.inv_logs + (long)k * C;
  for (int idx = tid; idx < C * C; idx += 256) {
    const int i = idx / C, j = idx - i * C;
    Lm[idx] = i > j ? l[idx] : (i == j ? 1.0f : 0.0f);
    Um[idx] = i < j ? u[idx] : (i == j ? sg[i] * expf(ls[i]) : 0.0f);
  }
  __syncthreads();
  for (int idx = tid; idx < C * C; idx += 256) {  // T = Lm Um
    const int i = idx / C, j = idx - i * C;
    float s = 0.0f;
    for (int q = 0; q < C; ++q) s += Lm[i * C + q] * Um[q * C + j];
    Tm[idx] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < C * C; idx += 256) {  // W = P T
    const int i = idx / C, j = idx - i * C;
    float s = 0.0f;
    for (int q = 0; q < C; ++q) s += P[i * C + q] * Tm[q * C + j];
    Wk[idx] = s;
    Wtk[(long)j * C + i] = s;
  }
  if (tid == 0) {
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += f.p.an_logs[(long)k * C + c] + ls[c];
    ldpart[k] = (float)C * s;
  }
  if (!Winv) return;
  // fp64 triangular inverses by substitution, one column per thread
  double* Li = dscratch + (long)k * 2 * C * C;
  double* Ui = Li + C * C;
  __syncthreads();
  for (int j = tid; j < C; j += 256) {
    // Lm unit lower: solve Lm x = e_j
    for (int i = 0; i < C; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int q = j; q < i; ++q) s -= (double)Lm[i * C + q] * Li[(long)q * C + j];
      Li[(long)i * C + j] = i < j ? 0.0 : s;  // diagonal is 1
    }
    // Um upper: solve Um x = e_j
    for (int i = C - 1; i >= 0; --i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int q = i + 1; q <= j; ++q) s -= (double)Um[i * C + q] * Ui[(long)q * C + j];
      Ui[(long)i * C + j] = i > j ? 0.0 : s / (double)Um[i * C + i];
    }
  }
  __syncthreads();
  // reference order (fp32): w = u^-1 (l^-1 p^-1), p^-1 = p^T for a permutation
  for (int idx = tid; idx < C * C; idx += 256) {
    const int i = idx / C, j = idx - i * C;
    float s = 0.0f;
    for (int q = 0; q < C; ++q) s += (float)Li[(long)i * C + q] * P[j * C + q];
    Tm[idx] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < C * C; idx += 256) {
    const int i = idx / C, j = idx - i * C;
    float s = 0.0f;
    for (int q = 0; q < C; ++q) s += (float)Ui[(long)i * C + q] * Tm[q * C + j];
    Winv[kc + idx] = s;
  }
}

__global__ __launch_bounds__(256) void flow_prep_transpose_kernel(FlowK f, float* __restrict__ wz_t, float* __restrict__ whh_t,
                                                                  float* __restrict__ wfl_t, float* __restrict__ wc,
                                                                  const float* __restrict__ ldpart, float* __restrict__ ldconst) {
  const int k = blockIdx.y;
  const int G = f.G, H = f.H, Ch = f.Ch, Cout = f.Cout, I = f.I, D = f.D;
  const long n1 = (long)Ch * G, n2 = (long)H * G, n3 = (long)H * Cout, n4 = (long)G * D;
  // wc[k][g][d] = w_ih[k][g][Ch + d]: the conditioning half of W_ih as its own 16-byte aligned matrix
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (long)gridDim.x * 256) {
    const int g = (int)(idx / D), dd = (int)(idx - (long)g * D);
    wc[(long)k * n4 + idx] = f.p.w_ih[((long)k * G + g) * I + Ch + dd];
  }
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n1 + n2 + n3; idx += (long)gridDim.x * 256) {
    if (idx < n1) {
      const int c = (int)(idx / G), g = (int)(idx - (long)c * G);
      wz_t[(long)k * n1 + idx] = f.p.w_ih[((long)k * G + g) * I + c];
    } else if (idx < n1 + n2) {
      const long q = idx - n1;
      const int h = (int)(q / G), g = (int)(q - (long)h * G);
      whh_t[(long)k * n2 + q] = f.p.w_hh[((long)k * G + g) * H + h];
    } else {
      const long q = idx - n1 - n2;
      const int h = (int)(q / Cout), col = (int)(q - (long)h * Cout);
      wfl_t[(long)k * n3 + q] = f.p.w_fl[((long)k * Cout + col) * H + h];
    }
  }
  if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.0f;
    for (int q = 0; q < f.Ks; ++q) s += ldpart[q];
    ldconst[0] = s;
  }
}

// Gradients of the LU parameters from dW (glow/modules.py:167-173 differentiated), one workgroup per flow step.
__global__ __launch_bounds__(256) void flow_invconv_bwd_kernel(FlowK f, const float* __restrict__ dW, lfi_flow_grads g,
                                                               float cconst, int accumulate, int stage) {
  const int k = blockIdx.x, C = f.C, tid = threadIdx.x;
  const long kc = (long)k * C * C;
  float* Lm = flow_smem;
  float* Um = Lm + C * C;
  float* Q = Um + C * C;  // P^T dW
  if (f.p.inv_w) {
    // dense: d weight = dW + cconst * W^-T   (d log|det W| / dW = W^-T)
    for (int idx = tid; idx < C * C; idx += 256) {
      const int i = idx / C, j = idx - i * C;
      const float v = dW[kc + idx] + cconst * f.Winv[kc + (long)j * C + i];
      g.inv_w[kc + idx] = accumulate ? g.inv_w[kc + idx] + v : v;
    }
    return;
  }
  const float* l = f.p.inv_l + kc;
  const float* u = f.p.inv_u + kc;
  const float* P = f.p.inv_p + kc;
  const float* sg = f.p.inv_sign + (long)k * C;
  const float* ls = f.p.inv_logs + (long)k * C;
  // (P and dW through LDS first: read from global inside the inner product - 100 dependent L2 round trips per thread - this
  // 16-workgroup kernel took 88 us alone and 260 us next to the HBM streams it shares its stream slot with)
  // (stage = 0, C > 90: five C x C images do not fit 160 KB of LDS - P and dW are then read where they lie, as before round 4)
  float* Pl = Q + C * C;
  float* Dl = Pl + C * C;
  const float* Pm = stage ? Pl : P;
  const float* Dm = stage ? Dl : dW + kc;
  for (int idx = tid; idx < C * C; idx += 256) {
    const int i = idx / C, j = idx - i * C;
    Lm[idx] = i > j ? l[idx] : (i == j ? 1.0f : 0.0f);
    Um[idx] = i < j ? u[idx] : (i == j ? sg[i] * expf(ls[i]) : 0.0f);
    if (stage) {
      Pl[idx] = P[idx];
      Dl[idx] = dW[kc + idx];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < C * C; idx += 256) {
    const int i = idx / C, j = idx - i * C;
    float s = 0.0f;
    for (int q = 0; q < C; ++q) s += Pm[q * C + i] * Dm[q * C + j];
    Q[idx] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < C * C; idx += 256) {
    const int i = idx / C, j = idx - i * C;
    if (i > j) {  // dL = Q Um^T, strictly lower
      float s = 0.0f;
      for (int q = 0; q < C; ++q) s += Q[i * C + q] * Um[j * C + q];
      g.inv_l[kc + idx] = accumulate ? g.inv_l[kc + idx] + s : s;
      if (!accumulate) g.inv_u[kc + idx] = 0.0f;
    } else {      // dU = Lm^T Q, upper incl. diagonal
      float s = 0.0f;
      for (int q = 0; q < C; ++q) s += Lm[q * C + i] * Q[q * C + j];
      if (i < j) {
        g.inv_u[kc + idx] = accumulate ? g.inv_u[kc + idx] + s : s;
        if (!accumulate) g.inv_l[kc + idx] = 0.0f;
      } else {
        const float v = s * sg[i] * expf(ls[i]) + cconst;
        const long o = (long)k * C + i;
        g.inv_logs[o] = accumulate ? g.inv_logs[o] + v : v;
        if (!accumulate) { g.inv_l[kc + idx] = 0.0f; g.inv_u[kc + idx] = 0.0f; }
      }
    }
  }
}

__global__ __launch_bounds__(256) void add_const_kernel(float* __restrict__ x, long n, float v) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += v;
}

// ------------------------------------------------------------------------------------------- actnorm init
__global__ __launch_bounds__(256) void actnorm_stats_kernel(const float* __restrict__ x, int rows, int C, double* __restrict__ sums) {
  // one block; column c handled by threads c, c+256...; fixed-order fp64 sums
  for (int c = threadIdx.x; c < C; c += 256) {
    double s = 0.0, s2 = 0.0;
    for (int r = 0; r < rows; ++r) {
      const double v = (double)x[(long)r * C + c];
      s += v; s2 += v * v;
    }
    sums[c] = s; sums[C + c] = s2;
  }
}
__global__ __launch_bounds__(256) void actnorm_apply_kernel(const double* __restrict__ sums, double count, int C, float scale,
                                                            float* __restrict__ bias, float* __restrict__ logs) {
  for (int c = threadIdx.x; c < C; c += 256) {
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    bias[c] = (float)(-mean);
    logs[c] = (float)log((double)scale / (sqrt(var) + 1e-6));
  }
}

// ------------------------------------------------------------------------------------------- host helpers
// floats of the prep buffer up to the end of the scratch area (published layout + log-det parts + fp64 workspace)
long prep_scratch_end(const lfi_flow_dims* d) {
  const int Ch = d->C / 2, C2 = d->C - Ch, Cout = d->affine ? 2 * C2 : C2, G = (d->lstm ? 4 : 3) * d->H;
  const long cc = (long)d->Ks * d->C * d->C;
  long n = 3 * cc + (long)d->Ks * Ch * G + (long)d->Ks * d->H * G + (long)d->Ks * d->H * Cout + (long)d->Ks * G * d->D + 4;
  n += d->Ks + 4;
  n += 2 * ((long)d->Ks * 2 * d->C * d->C + (long)d->Ks * d->C) + 8;  // doubles, counted as 2 floats each
  return (n + 3) & ~3L;  // 16-byte aligned: the images behind it are read with dwordx4 loads
}
long prep_padded_floats(const lfi_flow_dims* d) {
  const int Ch = d->C / 2, C2 = d->C - Ch, Cout = d->affine ? 2 * C2 : C2, NG = d->lstm ? 4 : 3;
  auto r16 = [](int x) { return (long)((x + 15) & ~15); };
  const long C16 = r16(d->C), Ch16 = Ch ? r16(Ch) : 16, H16 = r16(d->H), Co16 = r16(Cout);
  return d->Ks * (3 * C16 * C16 + Ch16 * NG * H16 + H16 * NG * H16 + 2 * H16 * Co16 + NG * H16 * H16 + NG * H16 * Ch16)
         + d->Ks * (NG * H16 * H16 + NG * H16 * Ch16) + 8    // + the bf16 hi/lo fragment images of bwh / bwz (same byte counts)
         + d->Ks * (Ch16 * NG * H16 + H16 * NG * H16 + H16 * Co16 + C16 * C16) + 8;   // + the fp16 hi/lo images of pwz / pwh / pwfl / pWinv
}

int fill_flow(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, FlowK* f, const char* who) {
  LFI_REQUIRE(d && p, "%s: null dims/params", who);
  LFI_REQUIRE(d->B > 0 && d->N > 0 && d->C >= 2 && d->H > 0 && d->D > 0 && d->Ks > 0, "%s: bad dims", who);
  f->B = d->B; f->N = d->N; f->C = d->C; f->H = d->H; f->D = d->D; f->Ks = d->Ks;
  f->affine = d->affine; f->lstm = d->lstm; f->eps = d->scale_eps;
  f->Ch = d->C / 2; f->C2 = d->C - f->Ch; f->Cout = d->affine ? 2 * f->C2 : f->C2;
  f->G = (d->lstm ? 4 : 3) * d->H; f->I = f->Ch + d->D; f->F = d->N * d->B; f->nbt = lfi_cdiv(d->B, MB);
  f->p = *p;
  f->stamps = g_lfi_stamps;
  {
    const char* e = getenv("LFI_STAMP_K");
    f->stamp_k = e ? atoi(e) : f->Ks / 2;
  }
  f->NG = d->lstm ? 4 : 3;
  f->ldc = (f->C + 3) & ~3; f->ldo = (f->Cout + 3) & ~3;
  f->C16 = (f->C + 15) & ~15; f->Ch16 = (f->Ch + 15) & ~15; f->H16 = (f->H + 15) & ~15; f->Co16 = (f->Cout + 15) & ~15;
  if (f->Ch16 == 0) f->Ch16 = 16;
  if (prep) {
    const long cc = (long)d->Ks * d->C * d->C;
    const float* q = prep;
    f->W = q; q += cc;
    f->Wt = q; q += cc;
    f->Winv = q; q += cc;
    f->wz_t = q; q += (long)d->Ks * f->Ch * f->G;
    f->whh_t = q; q += (long)d->Ks * d->H * f->G;
    f->wfl_t = q; q += (long)d->Ks * d->H * f->Cout;
    f->wc = q; q += (long)d->Ks * f->G * d->D;
    f->ldconst = q;
    // scratch (log-det parts, fp64 inverse workspace), then the zero-padded images of the register-resident cells
    q = prep + prep_scratch_end(d);
    const long Ks = d->Ks;
    f->pW = q; q += Ks * f->C16 * f->C16;
    f->pWt = q; q += Ks * f->C16 * f->C16;
    f->pwz = q; q += Ks * f->Ch16 * f->NG * f->H16;
    f->pwh = q; q += Ks * f->H16 * f->NG * f->H16;
    f->pwfl = q; q += Ks * f->H16 * f->Co16;
    f->bwfl = q; q += Ks * f->Co16 * f->H16;
    f->bwh = q; q += Ks * f->NG * f->H16 * f->H16;
    f->bwz = q; q += Ks * f->NG * f->H16 * f->Ch16;
    f->pWinv = q; q += Ks * f->C16 * f->C16;
    q = reinterpret_cast<const float*>((reinterpret_cast<uintptr_t>(q) + 15) & ~(uintptr_t)15);
    f->xbwh = reinterpret_cast<const uint4*>(q); q += Ks * f->NG * f->H16 * f->H16;
    f->xbwz = reinterpret_cast<const uint4*>(q); q += Ks * f->NG * f->H16 * f->Ch16;
    q = reinterpret_cast<const float*>((reinterpret_cast<uintptr_t>(q) + 15) & ~(uintptr_t)15);
    f->hwz = reinterpret_cast<const uint4*>(q); q += Ks * f->Ch16 * f->NG * f->H16;
    f->hwh = reinterpret_cast<const uint4*>(q); q += Ks * f->H16 * f->NG * f->H16;
    f->hwfl = reinterpret_cast<const uint4*>(q); q += Ks * f->H16 * f->Co16;
    f->hWinv = reinterpret_cast<const uint4*>(q); q += Ks * f->C16 * f->C16;
  }
  return LFI_OK;
}

long stash_offsets(const FlowK& f, long* off) {
  const long KF = (long)f.Ks * f.F;
  long o = 0;
  off[0] = o; o += KF * f.ldc;       // a      (rows of ldc = C rounded up to 4 floats)
  off[1] = o; o += KF * f.ldc;       // y
  off[2] = o; o += KF * f.ldc;       // x_out
  off[3] = o; o += KF * f.H;         // h
  o = (o + 3) & ~3L;
  off[4] = o; o += KF * 4 * f.H;     // gates: [frame][hidden unit][4], read and written 16 bytes at a time
  off[5] = o; o += KF * f.ldo;       // o      (rows of ldo = Cout rounded up to 4 floats)
  off[6] = o; o += KF;               // coupling log-det
  off[7] = o; o += f.lstm ? KF * f.H : 0;  // LSTM cell state
  return o;
}
long bstash_offsets(const FlowK& f, long* off) {
  const long KF = (long)f.Ks * f.F;
  long o = 0;
  off[0] = o; o += KF * f.ldo;       // dlin
  off[1] = o; o += KF * f.G;         // dgi
  off[2] = o; o += KF * f.G;         // dgh
  off[3] = o; o += KF * f.ldc;       // dy
  off[4] = o; o += KF * f.ldc;       // dx
  off[5] = o; o += KF * f.H;         // dh
  off[6] = o; o += (long)f.Ks * f.N * f.nbt * f.Cout;   // partial sums for l_fl
  off[7] = o; o += (long)f.Ks * f.N * f.nbt * 2 * f.C;  // partial sums for actnorm logs | bias
  off[8] = o; o += f.lstm ? KF * f.H : 0;               // carried d cell state (LSTM)
  off[9] = o; o += (long)f.Ks * f.nbt * 2 * f.G;        // partial sums for b_ih | b_hh (persistent walk)
  return o;
}
// persistent-pipeline state appended to either stash: header + one progress word per (flow step, batch tile), padded to 16 bytes
long pipe_words(const FlowK& f) { return ((long)PIPE_WALK_HDR + (long)f.Ks * f.nbt * PIPE_STRIDE + 3) & ~3L; }
long align4(long x) { return (x + 3) & ~3L; }
void bind_stash(FlowK* f, float* stash) {
  long off[8];
  stash_offsets(*f, off);
  f->sA = stash + off[0]; f->sY = stash + off[1]; f->sX = stash + off[2]; f->sH = stash + off[3];
  f->sG = stash + off[4]; f->sO = stash + off[5]; f->sL = stash + off[6]; f->sC = stash + off[7];
}
void bind_bstash(FlowK* f, float* b) {
  long off[10];
  bstash_offsets(*f, off);
  f->bDlin = b + off[0]; f->bDgi = b + off[1]; f->bDgh = b + off[2]; f->bDy = b + off[3];
  f->bDx = b + off[4]; f->bDh = b + off[5]; f->bPlfl = b + off[6]; f->bPan = b + off[7]; f->bDc = b + off[8];
  f->bPbias = b + off[9];
}

// LFI_FLOW_GENERIC=1 keeps the streaming cell kernels (tests cover both paths at sizes where either applies)
bool flow_force_generic() {
  const char* e = getenv("LFI_FLOW_GENERIC");
  return e && e[0] == '1';
}

// LFI_FLOW_PIPE=0 keeps one launch per anti-diagonal instead of the persistent pipeline (tests cover both)
bool flow_pipe_enabled() {
  const char* e = getenv("LFI_FLOW_PIPE");
  return !(e && e[0] == '0');
}
// LFI_PIPE_FENCE=1: consumers of a hand-off run an agent-scope acquire and read the tile with plain loads, instead of the
// fence-free form (every store and load of the tile sc1; MI355X_MICROARCH.md, hand-offs measured without the acquire, row 1)
// LFI_PIPE_X3=0: keep the exact f32 MFMA for the recurrent products of the persistent walk in bf16x3 mode too
bool flow_pipe_x3_enabled() {
  const char* e = getenv("LFI_PIPE_X3");
  return !(e && e[0] == '0');
}
// shapes for which lfi_flow_prep leaves the reverse cell's fp16 fragment images (whole 32-k blocks everywhere: the X3 reverse cell's condition)
bool flow_x3h_images_ok(const FlowK& f) { return !f.lstm && f.H16 % 32 == 0 && f.Ch16 % 32 == 0 && f.C16 % 32 == 0; }
// LFI_SAMPLE_WFRAG16=0: the sampler's reverse cells load the f32 images and split them in registers, as before round 5
bool flow_sample_wfrag16_enabled() {
  const char* e = getenv("LFI_SAMPLE_WFRAG16");
  return !(e && e[0] == '0');
}
// LFI_PIPE_FORCE_ABORT=1 (tests): start the walk with the abort word already set, as if a spin had timed out
bool flow_pipe_force_abort() {
  const char* e = getenv("LFI_PIPE_FORCE_ABORT");
  return e && e[0] == '1';
}
int flow_pipe_fence() {
  const char* e = getenv("LFI_PIPE_FENCE");
  return (e && e[0] == '1') ? 1 : 0;
}

template <typename Kf>
int set_flow_lds(Kf kernel, size_t bytes, const char* who) {
  if (bytes > 160 * 1024) {
    lfi_set_error("%s: needs %zu bytes of LDS (C/H too large)", who, bytes);
    return LFI_ERR_UNSUPPORTED;
  }
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      lfi_set_error("%s: hipFuncSetAttribute(%zu): %s", who, bytes, hipGetErrorString(e));
      return LFI_ERR_LAUNCH;
    }
  }
  return LFI_OK;
}

}  // namespace

// =================================================================================================== C ABI
// Diagnostics: device buffer of >= 16 * Ks 64-bit slots that the register-resident forward cells stamp with s_memtime at
// their phase boundaries (workgroup column 0 only); NULL switches it off. Process-global, not for production use.
extern "C" int lfi_debug_set_stamps(void* device_buffer) {
  g_lfi_stamps = (unsigned long long*)device_buffer;
  return LFI_OK;
}

extern "C" long lfi_flow_prep_floats(const lfi_flow_dims* d) {
  if (!d) return 0;
  const int Ch = d->C / 2, C2 = d->C - Ch, Cout = d->affine ? 2 * C2 : C2, G = (d->lstm ? 4 : 3) * d->H;
  (void)Ch; (void)Cout; (void)G;
  // published layout, scratch (per-step log-det parts, fp64 workspace for the inverses), zero-padded cell images
  return prep_scratch_end(d) + prep_padded_floats(d) + 16;
}

extern "C" int lfi_flow_prep(const lfi_flow_dims* d, const lfi_flow_params* p, float* prep, int with_inverse, void* stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_prep");
  if (rc) return rc;
  LFI_REQUIRE(prep, "lfi_flow_prep: null prep buffer");
  LFI_REQUIRE((p->inv_w != nullptr) || (p->inv_l && p->inv_u && p->inv_logs && p->inv_p && p->inv_sign),
              "lfi_flow_prep: invconv parameters missing");
  hipStream_t st = (hipStream_t)stream;
  float* ldconst = (float*)f.ldconst;
  float* ldpart = ldconst + 4;
  // 8-byte aligned fp64 scratch after the float part
  uintptr_t dp = (uintptr_t)(ldpart + d->Ks + 4);
  dp = (dp + 7) & ~(uintptr_t)7;
  double* dscratch = (double*)dp;
  const size_t lds = (size_t)3 * d->C * d->C * sizeof(float);
  rc = set_flow_lds(flow_prep_invconv_kernel, lds, "lfi_flow_prep");
  if (rc) return rc;
  hipLaunchKernelGGL(flow_prep_invconv_kernel, dim3(d->Ks), dim3(256), lds, st, f, (float*)f.W, (float*)f.Wt,
                     (with_inverse || p->inv_w) ? (float*)f.Winv : nullptr, dscratch, ldpart);
  LFI_LAUNCH_CHECK("lfi_flow_prep invconv");
  hipLaunchKernelGGL(flow_prep_transpose_kernel, dim3(64, d->Ks), dim3(256), 0, st, f, (float*)f.wz_t, (float*)f.whh_t,
                     (float*)f.wfl_t, (float*)f.wc, ldpart, ldconst);
  LFI_LAUNCH_CHECK("lfi_flow_prep transpose");
  if (flow_fast_ok(f.C, f.H, f.Cout)) {
    hipLaunchKernelGGL(flow_prep_pad_kernel, dim3(8, d->Ks, 9), dim3(256), 0, st, f, (float*)f.pW, (float*)f.pWt, (float*)f.pwz,
                       (float*)f.pwh, (float*)f.pwfl, (float*)f.bwfl, (float*)f.bwh, (float*)f.bwz,
                       (with_inverse || p->inv_w) ? (float*)f.pWinv : nullptr);
    LFI_LAUNCH_CHECK("lfi_flow_prep pad");
    if ((d->gemm_precision & 1) && !f.lstm && f.H16 % 32 == 0) {
      hipLaunchKernelGGL(flow_prep_x3_kernel, dim3(16, d->Ks, 2), dim3(256), 0, st, f, const_cast<uint4*>(f.xbwh),
                         const_cast<uint4*>(f.xbwz));
      LFI_LAUNCH_CHECK("lfi_flow_prep x3");
    }
    if ((with_inverse || p->inv_w) && flow_x3h_images_ok(f)) {   // (whatever the precision of this call: the sampler picks its own)
      hipLaunchKernelGGL(flow_prep_x3h_kernel, dim3(8, d->Ks, 4), dim3(256), 0, st, f, const_cast<uint4*>(f.hwz),
                         const_cast<uint4*>(f.hwh), const_cast<uint4*>(f.hwfl), const_cast<uint4*>(f.hWinv));
      LFI_LAUNCH_CHECK("lfi_flow_prep x3h");
    }
  }
  return LFI_OK;
}

extern "C" long lfi_flow_stash_floats(const lfi_flow_dims* d) {
  FlowK f = {};
  lfi_flow_params p = {};
  if (fill_flow(d, &p, nullptr, &f, "lfi_flow_stash_floats")) return -1;
  long off[8];
  return align4(stash_offsets(f, off)) + pipe_words(f);
}
extern "C" long lfi_flow_bstash_floats(const lfi_flow_dims* d) {
  FlowK f = {};
  lfi_flow_params p = {};
  if (fill_flow(d, &p, nullptr, &f, "lfi_flow_bstash_floats")) return -1;
  long off[10];
  return align4(bstash_offsets(f, off)) + pipe_words(f);
}
extern "C" float* lfi_flow_stash_ptr(const lfi_flow_dims* d, float* stash, int which) {
  FlowK f = {};
  lfi_flow_params p = {};
  if (which < 0 || which > 8 || fill_flow(d, &p, nullptr, &f, "lfi_flow_stash_ptr")) return nullptr;
  long off[8];
  const long end = stash_offsets(f, off);
  if (which == 8) return stash + align4(end);
  return stash + off[which];
}
extern "C" float* lfi_flow_bstash_ptr(const lfi_flow_dims* d, float* bstash, int which) {
  FlowK f = {};
  lfi_flow_params p = {};
  if (which < 0 || which > 9 || fill_flow(d, &p, nullptr, &f, "lfi_flow_bstash_ptr")) return nullptr;
  long off[10];
  const long end = bstash_offsets(f, off);
  if (which == 9) return bstash + align4(end);
  return bstash + off[which];
}

extern "C" int lfi_flow_seq_fwd(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* x0, int T,
                                int start, const float* gic, float* stash, float* z, float* nll, void* stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_seq_fwd");
  if (rc) return rc;
  LFI_REQUIRE(prep && x0 && gic && stash && nll, "lfi_flow_seq_fwd: null pointer");
  LFI_REQUIRE(start >= 0 && start + d->N <= T, "lfi_flow_seq_fwd: start %d + N %d > T %d", start, d->N, T);
  bind_stash(&f, stash);
  f.x0 = x0; f.T = T; f.start = start; f.gic = gic;
  hipStream_t st = (hipStream_t)stream;
  const bool fast = flow_fast_ok(f.C, f.H, f.Cout) && !flow_force_generic();
  const Carve cv = carve_fwd(f.C, f.H, f.Ch, f.C2, f.Cout);
  const CarveF cf = carve_fast_fwd(f.C, f.C16, f.H16, f.Ch16, f.Cout);
  const size_t lds = (size_t)(fast ? cf.total : cv.total) * sizeof(float);
  rc = fast ? (f.lstm ? set_flow_lds(flow_diag_fwd_fast_kernel<4>, lds, "lfi_flow_seq_fwd")
                      : set_flow_lds(flow_diag_fwd_fast_kernel<3>, lds, "lfi_flow_seq_fwd"))
            : set_flow_lds(flow_diag_fwd_kernel, lds, "lfi_flow_seq_fwd");
  if (rc) return rc;
  const bool pipe = fast && flow_pipe_enabled();
  if (pipe) {
    long off[8];
    f.pipe = reinterpret_cast<unsigned*>(stash + align4(stash_offsets(f, off)));
    f.pipe_fence = flow_pipe_fence();
    const size_t plds = (size_t)pipe_fwd_lds_floats(f.C, f.C16, f.H16, f.Ch16, f.Cout, f.Co16) * sizeof(float);
    // bf16 x 3 recurrent products (GRU cells, hidden and z widths whose 16-k padding is a whole number of 32-k blocks)
    const bool x3 = (d->gemm_precision & 1) && !f.lstm && (f.H16 % 32 == 0) && (f.Ch16 % 32 == 0) && flow_pipe_x3_enabled();
    rc = f.lstm ? set_flow_lds(flow_pipe_fwd_kernel<4, false>, plds, "lfi_flow_seq_fwd")
                : (x3 ? set_flow_lds(flow_pipe_fwd_kernel<3, true>, plds, "lfi_flow_seq_fwd")
                      : set_flow_lds(flow_pipe_fwd_kernel<3, false>, plds, "lfi_flow_seq_fwd"));
    if (rc) return rc;
    hipError_t me = hipMemsetAsync(f.pipe, 0, (size_t)pipe_words(f) * sizeof(unsigned), st);
    LFI_REQUIRE(me == hipSuccess, "lfi_flow_seq_fwd: hipMemsetAsync: %s", hipGetErrorString(me));
    if (flow_pipe_force_abort()) (void)hipMemsetAsync(f.pipe + 1, 1, sizeof(unsigned), st);
    const dim3 grid(f.Ks * f.nbt);
    if (f.lstm) hipLaunchKernelGGL((flow_pipe_fwd_kernel<4, false>), grid, dim3(NT), plds, st, f);
    else if (x3) hipLaunchKernelGGL((flow_pipe_fwd_kernel<3, true>), grid, dim3(NT), plds, st, f);
    else hipLaunchKernelGGL((flow_pipe_fwd_kernel<3, false>), grid, dim3(NT), plds, st, f);
  }
  for (int dg = 0; !pipe && dg < f.N + f.Ks - 1; ++dg) {
    const int klo = dg - (f.N - 1) > 0 ? dg - (f.N - 1) : 0;
    const int khi = dg < f.Ks - 1 ? dg : f.Ks - 1;
    const dim3 grid(f.nbt, khi - klo + 1);
    if (!fast) hipLaunchKernelGGL(flow_diag_fwd_kernel, grid, dim3(NT), lds, st, f, dg, klo);
    else if (f.lstm) hipLaunchKernelGGL(flow_diag_fwd_fast_kernel<4>, grid, dim3(NT), lds, st, f, dg, klo);
    else hipLaunchKernelGGL(flow_diag_fwd_fast_kernel<3>, grid, dim3(NT), lds, st, f, dg, klo);
  }
  LFI_LAUNCH_CHECK("lfi_flow_seq_fwd");
  if (f.C <= NLL_CMAX)
    hipLaunchKernelGGL(flow_nll_kernel<true>, dim3(lfi_cdiv(f.F, NLL_FR)), dim3(256), (size_t)NLL_FR * (f.C + 1) * sizeof(float), st, f, z, nll);
  else
    hipLaunchKernelGGL(flow_nll_kernel<false>, dim3(lfi_cdiv(f.F, 256)), dim3(256), 0, st, f, z, nll);
  LFI_LAUNCH_CHECK("lfi_flow_seq_fwd nll");
  return LFI_OK;
}

// Can the backward walk of these dims leave dgi as operand planes? It takes the persistent walk with bf16x3 recurrent products
// (whose d(gate) LDS images are the source), gate columns that are the stash's (H a multiple of 32: no padding columns) and
// batch tiles that pair up into whole 32-row plane tiles (B a multiple of 32).
static bool flow_bwd_planes_ok(const lfi_flow_dims* d) {
  if (!d || d->lstm || !(d->gemm_precision & 1) || d->H % 32 != 0 || d->B % 32 != 0) return false;
  const int Ch = d->C / 2, C2 = d->C - Ch, Cout = d->affine ? 2 * C2 : C2;
  return flow_fast_ok(d->C, d->H, Cout) && !flow_force_generic() && flow_pipe_enabled() && flow_pipe_x3_enabled() && 3 * d->H >= 128;
}
extern "C" int lfi_flow_bwd_emits_planes(const lfi_flow_dims* d) { return flow_bwd_planes_ok(d) ? 1 : 0; }

extern "C" int lfi_flow_seq_bwd(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* stash,
                                float gscale, float* bstash, void* stream) {
  return lfi_flow_seq_bwd_planes(d, p, prep, stash, gscale, bstash, nullptr, 0, stream);
}

extern "C" int lfi_flow_seq_bwd_planes(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* stash,
                                       float gscale, float* bstash, void* dgi_rows, int hi_only, void* stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_seq_bwd");
  if (rc) return rc;
  LFI_REQUIRE(prep && stash && bstash, "lfi_flow_seq_bwd: null pointer");
  LFI_REQUIRE(!dgi_rows || flow_bwd_planes_ok(d), "lfi_flow_seq_bwd_planes: these dims / switches cannot emit planes "
              "(lfi_flow_bwd_emits_planes returns 0)");
  LFI_REQUIRE((reinterpret_cast<uintptr_t>(dgi_rows) & 15) == 0, "lfi_flow_seq_bwd_planes: planes must be 16-byte aligned");
  bind_stash(&f, (float*)stash);
  bind_bstash(&f, bstash);
  f.bDgiR = reinterpret_cast<__bf16*>(dgi_rows);
  f.dgi_hi_only = hi_only ? 1 : 0;
  f.g16 = (dgi_rows && ((d->gemm_precision >> 16) & 1)) ? 1 : 0;
  f.gscale = gscale;
  hipStream_t st = (hipStream_t)stream;
  const bool fast = flow_fast_ok(f.C, f.H, f.Cout) && !flow_force_generic();
  const CarveB cv = carve_bwd(f.C, f.H, f.Cout, f.G);
  const CarveFB cf = carve_fast_bwd(f.C16, f.H16, f.Co16, f.Cout, f.NG);
  const size_t lds = (size_t)(fast ? cf.total : cv.total) * sizeof(float);
  rc = fast ? (f.lstm ? set_flow_lds(flow_diag_bwd_fast_kernel<4>, lds, "lfi_flow_seq_bwd")
                      : set_flow_lds(flow_diag_bwd_fast_kernel<3>, lds, "lfi_flow_seq_bwd"))
            : set_flow_lds(flow_diag_bwd_kernel, lds, "lfi_flow_seq_bwd");
  if (rc) return rc;
  const bool pipe = fast && flow_pipe_enabled();
  if (pipe) {
    long off[10];
    f.pipe = reinterpret_cast<unsigned*>(bstash + align4(bstash_offsets(f, off)));
    f.pipe_fence = flow_pipe_fence();
    // (NG * H16 >= 128: the bf16 operand images, 64 (NG H16 + 8) bytes each, must fit the fp32 regions they replace)
    const bool x3 = (d->gemm_precision & 1) && !f.lstm && (f.H16 % 32 == 0) && f.NG * f.H16 >= 128 && flow_pipe_x3_enabled();
    // + the bf16 hi / lo image of d lin (Q1's bf16 x 3 operand): 2 x MB rows of 32 ceil(Co16 / 32) + 8 bf16
    const size_t plds = x3 ? (((size_t)cf.total + 3) & ~(size_t)3) * sizeof(float) + (size_t)2 * MB * (32 * ((f.Co16 + 31) / 32) + 8) * 2 : lds;
    rc = f.lstm ? set_flow_lds(flow_pipe_bwd_kernel<4, false>, plds, "lfi_flow_seq_bwd")
                : (x3 ? set_flow_lds(flow_pipe_bwd_kernel<3, true>, plds, "lfi_flow_seq_bwd")
                      : set_flow_lds(flow_pipe_bwd_kernel<3, false>, plds, "lfi_flow_seq_bwd"));
    if (rc) return rc;
    hipError_t me = hipMemsetAsync(f.pipe, 0, (size_t)pipe_words(f) * sizeof(unsigned), st);
    LFI_REQUIRE(me == hipSuccess, "lfi_flow_seq_bwd: hipMemsetAsync: %s", hipGetErrorString(me));
    if (flow_pipe_force_abort()) (void)hipMemsetAsync(f.pipe + 1, 1, sizeof(unsigned), st);
    const dim3 grid(f.Ks * f.nbt);
    if (f.lstm) hipLaunchKernelGGL((flow_pipe_bwd_kernel<4, false>), grid, dim3(NT), plds, st, f);
    else if (x3) hipLaunchKernelGGL((flow_pipe_bwd_kernel<3, true>), grid, dim3(NT), plds, st, f);
    else hipLaunchKernelGGL((flow_pipe_bwd_kernel<3, false>), grid, dim3(NT), plds, st, f);
    hipLaunchKernelGGL(flow_pipe_poison_kernel, dim3(1), dim3(64), 0, st, f);
  }
  for (int dg = f.N + f.Ks - 2; !pipe && dg >= 0; --dg) {
    const int klo = dg - (f.N - 1) > 0 ? dg - (f.N - 1) : 0;
    const int khi = dg < f.Ks - 1 ? dg : f.Ks - 1;
    const dim3 grid(f.nbt, khi - klo + 1);
    if (!fast) hipLaunchKernelGGL(flow_diag_bwd_kernel, grid, dim3(NT), lds, st, f, dg, klo);
    else if (f.lstm) hipLaunchKernelGGL(flow_diag_bwd_fast_kernel<4>, grid, dim3(NT), lds, st, f, dg, klo);
    else hipLaunchKernelGGL(flow_diag_bwd_fast_kernel<3>, grid, dim3(NT), lds, st, f, dg, klo);
  }
  LFI_LAUNCH_CHECK("lfi_flow_seq_bwd");
  return LFI_OK;
}

extern "C" long lfi_flow_param_grads_work_floats(const lfi_flow_dims* d) {
  FlowK f = {};
  lfi_flow_params p = {};
  if (fill_flow(d, &p, nullptr, &f, "lfi_flow_param_grads_work_floats")) return -1;
  const int splitk = 16;
  long gemm_ws = (long)f.Ks * splitk * ((long)f.G * f.I > (long)f.C * f.C ? (long)f.G * f.I : (long)f.C * f.C);
  long cs = lfi_colsum_work_floats(f.F > f.N * f.nbt ? f.F : f.N * f.nbt, f.G > 2 * f.C ? f.G : 2 * f.C, f.Ks);
  // + a second split-K workspace for the products that run on bias_stream
  // + the per-workgroup partials of the one-pass form of the thin products (lfi_wgrad.hip)
  long fused = 0;
  if (lfi_internal_flow_wgrad_ok(f.B, f.N, f.C, f.Ch, f.Cout, f.H, f.G, f.ldc, f.ldo))
    fused = lfi_internal_flow_wgrad_work_floats(f.B, f.N, f.Ks, 0) + lfi_internal_flow_wgrad_work_floats(f.B, f.N, f.Ks, 1) + 32;
  return (long)f.Ks * f.C * f.C + gemm_ws + cs + 16 + gemm_ws + 16 + fused;
}

// LFI_FLOW_WGRAD_FUSED=0: the thin weight-gradient products as four batched split-K launches + a column-sum pass (rounds 2 - 5)
bool flow_wgrad_fused_enabled() {
  const char* e = getenv("LFI_FLOW_WGRAD_FUSED");
  return !(e && e[0] == '0');
}

extern "C" int lfi_flow_param_grads(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* stash,
                                    const float* bstash, const float* c, long ldc, float gscale, const lfi_flow_grads* g,
                                    int accumulate, float* work, void* stream, void* bias_stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_param_grads");
  if (rc) return rc;
  LFI_REQUIRE(prep && stash && bstash && g && work, "lfi_flow_param_grads: null pointer");
  bind_stash(&f, (float*)stash);
  bind_bstash(&f, (float*)bstash);
  const int Ks = f.Ks, F = f.F, B = f.B, C = f.C, H = f.H, G = f.G, I = f.I, Cout = f.Cout, Ch = f.Ch, D = f.D;
  float* dW = work;                       // [Ks][C][C]
  float* gws = dW + (long)Ks * C * C;     // split-k workspace
  const int splitk = 16;
  const long gws_floats = (long)Ks * splitk * ((long)G * I > (long)C * C ? (long)G * I : (long)C * C);
  float* cws = gws + gws_floats;          // colsum workspace
  const long cws_floats = lfi_colsum_work_floats(F > f.N * f.nbt ? F : f.N * f.nbt, G > 2 * C ? G : 2 * C, Ks);
  float* gws2 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(cws + cws_floats) + 63) & ~(uintptr_t)63);  // dW split-K partials
  void* bs = bias_stream ? bias_stream : stream;

  // gemm_precision bit 16 (with the walk's planes mode): dgi | dgh rows are bf16 (FlowK.g16) - the two products that read them take
  // them as an operand that arrives rounded (lfi_gemm_desc.a_bf16: two-product mode, i.e. skip bit 0 set by the caller)
  const bool g16 = ((d->gemm_precision >> 16) & 1) != 0;
  LFI_REQUIRE(!g16 || (flow_bwd_planes_ok(d) && !c && (d->gemm_precision & 0x100) && !(d->gemm_precision & 0x200)),
              "lfi_flow_param_grads: bf16 gradient rows (gemm_precision bit 16) need the walk's planes mode and two-product thin products");
  lfi_gemm_desc q = {};
  q.batch = Ks; q.accumulate = accumulate; q.splitk = splitk; q.work = gws; q.precision = d->gemm_precision & 0xffff;
  q.a_kcontig = 0; q.b_kcontig = 0; q.K = F;
  // K split (<= 16, what the workspace holds) of a K = F product with Ks x few 128 x 128 output tiles: the one that best
  // fills whole rounds of the 512 co-resident workgroups, less 3 % per extra partial for the reduce pass
  auto fill_split = [&](int M, int N, int K) {
    const double tiles = (double)lfi_cdiv(M, 128) * lfi_cdiv(N, 128) * Ks;
    int best = 1;
    double best_score = -1.0;
    for (int sk = 1; sk <= splitk && K / sk >= 512; ++sk) {
      const double wg = tiles * sk, rounds = (double)(long)((wg + 511.0) / 512.0);
      const double score = wg / (rounds * 512.0) * (1.0 - 0.03 * (sk - 1));
      if (score > best_score + 1e-9) { best_score = score; best = sk; }
    }
    return best;
  };
  // w_ih[k][:, Ch:] (G x D) = dgi[k]^T c[:, kD:(k+1)D]: the one MFMA-bound product (it reads c: before the caller's dpre
  // product overwrites it)
  // (c == NULL: the caller runs this product on operand planes itself - lfi_gemm_planes on the walk's dgi planes and c's)
  if (c) {
    q.M = G; q.N = D; q.splitk = fill_split(G, D, F); q.A = f.bDgi; q.lda = G; q.strideA = (long)F * G; q.B = c; q.ldb = ldc; q.strideB = D;
    q.C = g->w_ih + Ch; q.ldc = I; q.strideC = (long)G * I;
    if ((rc = lfi_gemm_f32(&q, stream))) return rc;
  }
  // Round 6: ONE pass over the backward stash for all of the thin products below, b_fl's column sums included (lfi_wgrad.hip) -
  // when the walk left dgi | dgh as bf16 rows (two-product mode) and the shapes are the register-resident cell's. Two roles:
  // w_hh | w_fl | b_fl in line on `stream` (its workgroups hold a CU each: beside the caller's MFMA-bound dgi^T c product on the
  // other stream both ran at half speed, profiles/round6_thin_products_ab.md), w_ih[:, :Ch] | dW (two workgroups per CU) on
  // bias_stream, where dW's reader, the LU-gradient kernel, follows (LFI_FLOW_WGRAD_IH=main: in line as well).
  const bool fused = g16 && !f.lstm && flow_wgrad_fused_enabled() &&
                     lfi_internal_flow_wgrad_ok(B, f.N, C, Ch, Cout, H, G, f.ldc, f.ldo) != 0;
  if (fused) {
    float* part0 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(gws2 + gws_floats) + 63) & ~(uintptr_t)63);
    float* part1 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(part0 + lfi_internal_flow_wgrad_work_floats(B, f.N, Ks, 0)) + 63) & ~(uintptr_t)63);
    const char* ws = getenv("LFI_FLOW_WGRAD_IH");
    void* s1 = (ws && ws[0] == 'm') ? stream : bs;   // default: side (same-box A/B 6.44 against 6.48 ms per step in line)
    if ((rc = lfi_internal_flow_wgrad(1, B, f.N, Ks, C, Ch, Cout, I, f.ldc, f.ldo, f.bDgh, f.bDgi, f.sH, f.bDlin, f.sY, f.sA, f.bDy, part1,
                                      g->w_hh, g->w_ih, g->w_fl, g->b_fl, dW, accumulate, s1)))
      return rc;
    if (s1 != bs) {   // the LU-gradient kernel below reads dW on bias_stream
      hipEvent_t ev;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { lfi_set_error("lfi_flow_param_grads: hipEventCreate failed"); return LFI_ERR_LAUNCH; }
      (void)hipEventRecord(ev, (hipStream_t)s1);
      (void)hipStreamWaitEvent((hipStream_t)bs, ev, 0);
      (void)hipEventDestroy(ev);
    }
    if ((rc = lfi_internal_flow_wgrad(0, B, f.N, Ks, C, Ch, Cout, I, f.ldc, f.ldo, f.bDgh, f.bDgi, f.sH, f.bDlin, f.sY, f.sA, f.bDy, part0,
                                      g->w_hh, g->w_ih, g->w_fl, g->b_fl, dW, accumulate, stream)))
      return rc;
  } else {
  // The other products are thin (K = F frames, a few output tiles per step), HBM-bound at 3 - 4 TB/s. (Moving them to
  // bias_stream as well, next to the caller's MFMA-bound products, was measured no better than leaving them here: same-box
  // A/B 0.18 ms per step gained with them there, 0.21 ms without; round 4, any one or two of them: 6.86 - 6.91 ms per step
  // whichever way.)
  // w_fl[k] (Cout x H) = dlin[k]^T h[k]
  q.M = Cout; q.N = H; q.splitk = fill_split(Cout, H, F); q.A = f.bDlin; q.lda = f.ldo; q.strideA = (long)F * f.ldo; q.B = f.sH; q.ldb = H; q.strideB = (long)F * H;
  q.C = g->w_fl; q.ldc = H; q.strideC = (long)Cout * H;
  if ((rc = lfi_gemm_f32(&q, stream))) return rc;
  // w_hh[k] (G x H) = dgh[k][n >= 1]^T h[k][n - 1]
  if (f.N > 1) {
    q.K = F - B; q.M = G; q.N = H; q.splitk = fill_split(G, H, F - B); q.A = f.bDgh + (long)B * G; q.lda = G; q.strideA = (long)F * G; q.B = f.sH; q.ldb = H;
    q.strideB = (long)F * H; q.C = g->w_hh; q.ldc = H; q.strideC = (long)G * H;
    if (g16) { q.a_bf16 = 1; q.A = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(f.bDgh) + (long)B * G); }
    if ((rc = lfi_gemm_f32(&q, stream))) return rc;
    q.a_bf16 = 0;
  } else if (!accumulate) {
    (void)hipMemsetAsync(g->w_hh, 0, sizeof(float) * (size_t)Ks * G * H, (hipStream_t)stream);
  }
  q.K = F;
  // w_ih[k][:, :Ch] (G x Ch) = dgi[k]^T z1[k]
  if (Ch > 0) {
    q.M = G; q.N = Ch; q.splitk = fill_split(G, Ch, F); q.A = f.bDgi; q.lda = G; q.strideA = (long)F * G; q.B = f.sY; q.ldb = f.ldc; q.strideB = (long)F * f.ldc;
    q.C = g->w_ih; q.ldc = I; q.strideC = (long)G * I;
    q.a_bf16 = g16 ? 1 : 0;
    if ((rc = lfi_gemm_f32(&q, stream))) return rc;
    q.a_bf16 = 0;
  }
  // dW[k] (C x C) = a[k]^T dy[k]  -> LU parameter gradients, then a 16-workgroup kernel: on bias_stream (with a split-K
  // workspace of their own) they run next to the products above instead of holding the chip for 0.14 ms
  q.accumulate = 0; q.work = gws2;
  q.M = C; q.N = C; q.splitk = fill_split(C, C, F); q.A = f.sA; q.lda = f.ldc; q.strideA = (long)F * f.ldc; q.B = f.bDy; q.ldb = f.ldc; q.strideB = (long)F * f.ldc;
  q.C = dW; q.ldc = C; q.strideC = (long)C * C;
  if ((rc = lfi_gemm_f32(&q, bs))) return rc;
  }
  // constant log-det terms: nll has -(C sum(logs))/ln2 per frame -> d/dlogs = -C/ln2 * gscale * F
  const float cconst = -(float)((double)gscale * (double)F * (double)C / 0.6931471805599453);
  {
    const int stage = (size_t)5 * C * C * sizeof(float) <= 160 * 1024 ? 1 : 0;
    const size_t lds = (size_t)(stage ? 5 : 3) * C * C * sizeof(float);
    rc = set_flow_lds(flow_invconv_bwd_kernel, lds, "lfi_flow_param_grads");
    if (rc) return rc;
    hipLaunchKernelGGL(flow_invconv_bwd_kernel, dim3(Ks), dim3(256), lds, (hipStream_t)bs, f, dW, *g, cconst, accumulate, stage);
    LFI_LAUNCH_CHECK("lfi_flow_param_grads invconv");
  }
  // biases and the per-tile partial sums: column sums over the backward stash only (HBM streams), independent of the
  // products above - on bias_stream when the caller has forked one after the backward walk
  if (!fused && (rc = lfi_colsum_f32(f.bDlin, f.ldo, (long)F * f.ldo, F, Cout, Ks, g->b_fl, Cout, 1.0f, accumulate, cws, bs))) return rc;
  if (flow_fast_ok(f.C, f.H, f.Cout) && !flow_force_generic() && flow_pipe_enabled()) {
    // the persistent backward walk left per-workgroup sums of dgi | dgh: [Ks][nbt][2][G]
    if ((rc = lfi_colsum_f32(f.bPbias, 2 * G, (long)f.nbt * 2 * G, f.nbt, G, Ks, g->b_ih, G, 1.0f, accumulate, cws, bs))) return rc;
    if ((rc = lfi_colsum_f32(f.bPbias + G, 2 * G, (long)f.nbt * 2 * G, f.nbt, G, Ks, g->b_hh, G, 1.0f, accumulate, cws, bs))) return rc;
  } else {
    if ((rc = lfi_colsum_f32(f.bDgh, G, (long)F * G, F, G, Ks, g->b_hh, G, 1.0f, accumulate, cws, bs))) return rc;
    if ((rc = lfi_colsum_f32(f.bDgi, G, (long)F * G, F, G, Ks, g->b_ih, G, 1.0f, accumulate, cws, bs))) return rc;
  }
  const int prow = f.N * f.nbt;
  if ((rc = lfi_colsum_f32(f.bPlfl, Cout, (long)prow * Cout, prow, Cout, Ks, g->l_fl, Cout, 1.0f, accumulate, cws, bs))) return rc;
  if ((rc = lfi_colsum_f32(f.bPan, 2 * C, (long)prow * 2 * C, prow, C, Ks, g->an_logs, C, 1.0f, accumulate, cws, bs))) return rc;
  if ((rc = lfi_colsum_f32(f.bPan + C, 2 * C, (long)prow * 2 * C, prow, C, Ks, g->an_bias, C, 1.0f, accumulate, cws, bs))) return rc;
  hipLaunchKernelGGL(add_const_kernel, dim3(lfi_cdiv((long)Ks * C, 256)), dim3(256), 0, (hipStream_t)bs, g->an_logs, (long)Ks * C, cconst);
  LFI_LAUNCH_CHECK("lfi_flow_param_grads const");
  return LFI_OK;
}

extern "C" int lfi_actnorm_init_stats(const float* x, int rows, int C, double* sums, void* stream) {
  LFI_REQUIRE(x && sums && rows > 0 && C > 0, "lfi_actnorm_init_stats: bad arguments");
  hipLaunchKernelGGL(actnorm_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, rows, C, sums);
  LFI_LAUNCH_CHECK("lfi_actnorm_init_stats");
  return LFI_OK;
}
extern "C" int lfi_actnorm_init_apply(const double* sums, double count, int C, float scale, float* bias, float* logs,
                                      void* stream) {
  LFI_REQUIRE(sums && bias && logs && count > 0 && C > 0, "lfi_actnorm_init_apply: bad arguments");
  hipLaunchKernelGGL(actnorm_apply_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, count, C, scale, bias, logs);
  LFI_LAUNCH_CHECK("lfi_actnorm_init_apply");
  return LFI_OK;
}

extern "C" int lfi_flow_step(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, int k, int rows,
                             const float* x_in, long ldx, const float* h_prev, const float* c_prev, const float* gic_k,
                             float* x_out, long ldxo, float* h_out, float* c_out, float* ldc_acc, int reverse, void* stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_step");
  if (rc) return rc;
  LFI_REQUIRE(prep && x_in && gic_k && x_out && h_out, "lfi_flow_step: null pointer");
  LFI_REQUIRE(k >= 0 && k < d->Ks && rows > 0, "lfi_flow_step: bad k/rows");
  LFI_REQUIRE(!d->lstm || c_out, "lfi_flow_step: the LSTM cell needs c_out");
  CellIO io = {};
  io.k = k; io.rows = rows; io.x_in = x_in; io.ldx = ldx; io.h_prev = h_prev; io.gic = gic_k;
  io.c_prev = d->lstm ? c_prev : nullptr; io.c_out = d->lstm ? c_out : nullptr;
  io.x_out = x_out; io.ldxo = ldxo; io.h_out = h_out; io.l_out = ldc_acc; io.l_accumulate = 1;
  const Carve cv = carve_fwd(f.C, f.H, f.Ch, f.C2, f.Cout);
  size_t lds = (size_t)cv.total * sizeof(float);
  if (reverse && flow_fast_ok(f.C, f.H, f.Cout) && !flow_force_generic()) {
    lds = (size_t)carve_fast_fwd(f.C, f.C16, f.H16, f.Ch16, f.Cout).total * sizeof(float);
    rc = f.lstm ? set_flow_lds(flow_step_rev_fast_kernel<4>, lds, "lfi_flow_step") : set_flow_lds(flow_step_rev_fast_kernel<3>, lds, "lfi_flow_step");
    if (rc) return rc;
    if (f.lstm) hipLaunchKernelGGL(flow_step_rev_fast_kernel<4>, dim3(lfi_cdiv(rows, MB)), dim3(NT), lds, (hipStream_t)stream, f, io);
    else hipLaunchKernelGGL(flow_step_rev_fast_kernel<3>, dim3(lfi_cdiv(rows, MB)), dim3(NT), lds, (hipStream_t)stream, f, io);
    LFI_LAUNCH_CHECK("lfi_flow_step");
    return LFI_OK;
  }
  rc = reverse ? set_flow_lds(flow_step_kernel<true>, lds, "lfi_flow_step")
               : set_flow_lds(flow_step_kernel<false>, lds, "lfi_flow_step");
  if (rc) return rc;
  if (reverse) hipLaunchKernelGGL(flow_step_kernel<true>, dim3(lfi_cdiv(rows, MB)), dim3(NT), lds, (hipStream_t)stream, f, io);
  else hipLaunchKernelGGL(flow_step_kernel<false>, dim3(lfi_cdiv(rows, MB)), dim3(NT), lds, (hipStream_t)stream, f, io);
  LFI_LAUNCH_CHECK("lfi_flow_step");
  return LFI_OK;
}

// SeqGlow.invert (glow/models.py:617-645) as ONE persistent launch (flow_rev_walk_kernel) + the sum of the per-step log-dets.
extern "C" int lfi_flow_seq_rev_ok(const lfi_flow_dims* d) {
  if (!d) return 0;
  const int Cout = d->affine ? 2 * (d->C - d->C / 2) : d->C - d->C / 2;
  const char* e = getenv("LFI_INVERT_WALK");
  return (flow_fast_ok(d->C, d->H, Cout) && !flow_force_generic() && !(e && e[0] == '0')) ? 1 : 0;
}

extern "C" long lfi_flow_seq_rev_work_floats(const lfi_flow_dims* d) {
  if (!d) return 0;
  const long F = (long)d->N * d->B, tiles = (d->B + MB - 1) / MB;
  return (long)d->Ks * F * d->C + (long)d->Ks * F + lfi_colsum_work_floats(d->Ks, (int)F, 1) +
         (((long)PIPE_HDR + d->Ks * tiles + 3) & ~3L) + 16;
}

extern "C" int lfi_flow_seq_rev(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* z,
                                const float* gic, float* x_out, float* logdet, float* h, float* cstate, float* work,
                                void* stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_seq_rev");
  if (rc) return rc;
  LFI_REQUIRE(prep && z && gic && x_out && logdet && h && work, "lfi_flow_seq_rev: null pointer");
  LFI_REQUIRE(!d->lstm || cstate, "lfi_flow_seq_rev: the LSTM cell needs cstate");
  LFI_REQUIRE(lfi_flow_seq_rev_ok(d), "lfi_flow_seq_rev: C <= 64, hidden_channels <= 128 only (lfi_flow_seq_rev_ok); wider flows "
              "walk cell by cell with lfi_flow_step");
  LFI_REQUIRE((long)f.N * f.B < (1L << 31), "lfi_flow_seq_rev: too many frames");
  hipStream_t st = (hipStream_t)stream;
  const long F = f.F;
  RevWalk rw = {};
  rw.z = z; rw.gic = gic; rw.out = x_out; rw.h = h; rw.cstate = cstate;
  rw.tiles = work;
  rw.ldk = rw.tiles + (long)f.Ks * F * f.C;
  float* cws = rw.ldk + (long)f.Ks * F;
  rw.pipe = reinterpret_cast<unsigned*>((reinterpret_cast<uintptr_t>(cws + lfi_colsum_work_floats(f.Ks, (int)F, 1)) + 15) & ~(uintptr_t)15);
  const size_t words = (size_t)(((long)PIPE_HDR + (long)f.Ks * f.nbt + 3) & ~3L);
  hipError_t me = hipMemsetAsync(rw.pipe, 0, words * sizeof(unsigned), st);
  LFI_REQUIRE(me == hipSuccess, "lfi_flow_seq_rev: hipMemsetAsync: %s", hipGetErrorString(me));
  const size_t lds = (size_t)carve_fast_fwd(f.C, f.C16, f.H16, f.Ch16, f.Cout).total * sizeof(float);
  rc = f.lstm ? set_flow_lds(flow_rev_walk_kernel<4>, lds, "lfi_flow_seq_rev") : set_flow_lds(flow_rev_walk_kernel<3>, lds, "lfi_flow_seq_rev");
  if (rc) return rc;
  if (f.lstm) hipLaunchKernelGGL(flow_rev_walk_kernel<4>, dim3(f.Ks * f.nbt), dim3(NT), lds, st, f, rw);
  else hipLaunchKernelGGL(flow_rev_walk_kernel<3>, dim3(f.Ks * f.nbt), dim3(NT), lds, st, f, rw);
  LFI_LAUNCH_CHECK("lfi_flow_seq_rev");
  // logdet[n][b] = sum over the flow steps of the coupling log-dets (the constant ActNorm / invconv part is the caller's)
  return lfi_colsum_f32(rw.ldk, F, 0, f.Ks, (int)F, 1, logdet, 0, 1.0f, 0, cws, stream);
}

// SeqGlow.inference (glow/models.py:567-596): everything that does not depend on generated frames was hoisted by the
// caller into pre_static; per frame two small GEMMs (window part of cond_transform, then W_ih[:, Ch:] c) and Ks
// reverse cells. The growing torch.cat history of the reference (:591, O(T^2) copies) is a preallocated buffer here.
extern "C" long lfi_flow_sample_p1_work_floats(const lfi_flow_dims* d, const lfi_p1enc* e, int hist1) {
  if (!d || !e || e->kind == 0) return 0;
  const long hid4 = (e->hid + 3) & ~3;
  long n = (long)d->B * hid4 + 16;
  if (e->kind == 2 || e->kind == 3) {
    const int ng = e->kind == 3 ? 4 : 3;
    lfi_enc_desc ed = {};
    ed.B = d->B; ed.T = hist1; ed.N = 1; ed.start = hist1 - 1; ed.hist = hist1; ed.hid = e->hid; ed.lstm = e->kind == 3;
    n += (long)d->B * hist1 * ng * e->hid + lfi_encode_windows_work_floats(&ed) + (long)hist1 * d->B * e->hid;
    if (e->kind == 3) n += (long)hist1 * d->B * 5 * e->hid;   // the LSTM encoder keeps its cell state in the gate stash
  }
  return n;
}

extern "C" long lfi_flow_sample_work_floats(const lfi_flow_dims* d) {
  if (!d) return 0;
  const int G = (d->lstm ? 4 : 3) * d->H;
  const long tiles = (d->B + MB - 1) / MB;
  return (long)d->B * d->Ks * d->D + (long)d->Ks * d->B * G + 2L * d->B * d->C + 16
         + (((long)PIPE_HDR + d->Ks * tiles + 3) & ~3L) + 4    // + the hand-off words of the per-frame reverse chain
         + (long)d->B * 64 * ((d->C + 3) & ~3) + 4             // + the aligned copy of the raw prev_p1_face window (hist1 <= 64)
         + lfi_internal_sample_cond_bytes(d->B, d->Ks, G, 512) / 4 + 64;   // + the fused conditioning's fragments (window <= 512 floats)
}

extern "C" int lfi_flow_sample_seq(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* wct,
                                   long E, int hist1, float* pre_static, const float* noise, float* faces, int seq_len,
                                   int start, int nframes, float* h, float* cstate, const lfi_p1enc* p1, float* p1work,
                                   float* work, void* stream) {
  return lfi_flow_sample_seq_from(d, p, prep, wct, E, hist1, pre_static, noise, faces, seq_len, start, nframes, 0, h, cstate, p1,
                                  p1work, work, stream);
}

// A run of `nframes` generated frames that is NOT the first of its sequence: first_frame = how many frames of the sequence earlier
// calls generated (> 0: the recurrent state in h / cstate is theirs and carries on; pre_static / noise / start are this run's own).
// The engine samples a long sequence as a few such runs so that the static part of run i + 1 (window encoders, the
// non-autoregressive cond_transform columns) can be computed on a second stream under the latency-bound chain of run i.
extern "C" int lfi_flow_sample_seq_from(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* wct,
                                        long E, int hist1, float* pre_static, const float* noise, float* faces, int seq_len,
                                        int start, int nframes, int first_frame, float* h, float* cstate, const lfi_p1enc* p1,
                                        float* p1work, float* work, void* stream) {
  FlowK f = {};
  int rc = fill_flow(d, p, prep, &f, "lfi_flow_sample_seq");
  if (rc) return rc;
  LFI_REQUIRE(first_frame >= 0, "lfi_flow_sample_seq_from: negative first_frame");
  LFI_REQUIRE(prep && wct && pre_static && noise && faces && h && work, "lfi_flow_sample_seq: null pointer");
  LFI_REQUIRE(hist1 >= 0 && hist1 <= start && start + nframes <= seq_len, "lfi_flow_sample_seq: bad frame range");
  LFI_REQUIRE((long)hist1 * d->C <= E, "lfi_flow_sample_seq: window wider than the feature vector");
  LFI_REQUIRE(!d->lstm || cstate, "lfi_flow_sample_seq: the LSTM cell needs cstate");
  const int p1kind = p1 ? p1->kind : 0;
  LFI_REQUIRE(p1kind >= 0 && p1kind <= 3, "lfi_flow_sample_seq: bad p1_face encoder kind %d", p1kind);
  LFI_REQUIRE(p1kind == 0 || (p1work && p1->hid > 0), "lfi_flow_sample_seq: encoded p1_face window needs p1work");
  const int p1col = p1 ? p1->col : 0;
  const int B = f.B, C = f.C, H = f.H, D = f.D, Ks = f.Ks, G = f.G;
  float* gic = work + (long)B * Ks * D;        // [Ks][B][G]   (the first B x Ks*D floats: round 2's copy of c, unused now)
  float* xa = gic + (long)Ks * B * G;          // B x C ping
  float* xb = xa + (long)B * C;                // B x C pong
  hipStream_t st = (hipStream_t)stream;
  const bool fast = flow_fast_ok(f.C, f.H, f.Cout) && !flow_force_generic();
  const Carve cv = carve_fwd(f.C, f.H, f.Ch, f.C2, f.Cout);
  const size_t lds = (size_t)(fast ? carve_fast_fwd(f.C, f.C16, f.H16, f.Ch16, f.Cout).total : cv.total) * sizeof(float);
  rc = !fast ? set_flow_lds(flow_step_kernel<true>, lds, "lfi_flow_sample_seq")
             : (f.lstm ? set_flow_lds(flow_step_rev_fast_kernel<4>, lds, "lfi_flow_sample_seq")
                       : set_flow_lds(flow_step_rev_fast_kernel<3>, lds, "lfi_flow_sample_seq"));
  if (rc) return rc;
  // LFI_SAMPLE_CHAIN=0 keeps one launch per flow step
  const char* ce = getenv("LFI_SAMPLE_CHAIN");
  const bool chain = fast && !(ce && ce[0] == '0');
  // (the reverse cells' recurrent products as three fp16 products - fp32-grade - in both bf16 modes of the per-frame GEMMs)
  const bool x3 = (d->gemm_precision & 1) && !f.lstm && (f.H16 % 32 == 0) && (f.Ch16 % 32 == 0) && (f.C16 % 32 == 0) && flow_pipe_x3_enabled();
  unsigned* chain_state = reinterpret_cast<unsigned*>((reinterpret_cast<uintptr_t>(xb + (long)B * C) + 15) & ~(uintptr_t)15);
  const size_t chain_words = (size_t)(((long)PIPE_HDR + (long)Ks * f.nbt + 3) & ~3L);
  // raw prev_p1_face windows start (t - hist1) * C floats into a row: 16-byte aligned only on every other frame at C = 50,
  // which sent half of the window products to the exact-f32 kernel (91 vs 35 us). A gather into an aligned buffer first.
  float* wstage = reinterpret_cast<float*>(chain_state + chain_words);
  const int ldw = (hist1 * C + 3) & ~3;
  const bool stage_win = p1kind == 0 && hist1 <= 64;
  // raw window + fp16 pieces (precision 9) + final widths: cond_transform's window part and the coupling cell's input projection
  // as ONE launch per frame, c never written (lfi_sample.hip); the weights' fragments are made here, once per call
  const int K1 = hist1 * C;
  const bool fused = stage_win && (d->gemm_precision & 0xff) == 9 && lfi_internal_sample_cond_ok(D, G, K1) &&
                     (reinterpret_cast<uintptr_t>(pre_static) & 15) == 0;
  void* cfrags = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(wstage + (long)B * 64 * ((C + 3) & ~3)) + 255) & ~(uintptr_t)255);
  if (fused && nframes > 0) {
    if ((rc = lfi_internal_sample_cond_prepare(wct, E, p1col, K1, f.wc, Ks, G, cfrags, stream))) return rc;
  }
  // LFI_SAMPLE_XF_CHAIN=0 keeps the window-fragment kernel in front of every frame's conditioning
  const char* xce = getenv("LFI_SAMPLE_XF_CHAIN");
  const bool xf_chain = fused && chain && !(xce && xce[0] == '0');
  // the reverse cells' weights as the fp16 fragment images lfi_flow_prep left (no split in every workgroup of every frame)
  const bool xw = x3 && chain && flow_x3h_images_ok(f) && flow_sample_wfrag16_enabled();
  if (chain) {
    rc = f.lstm ? set_flow_lds(flow_rev_chain_kernel<4, false>, lds, "lfi_flow_sample_seq")
                : (xw ? set_flow_lds(flow_rev_chain_kernel<3, true, true>, lds, "lfi_flow_sample_seq")
                      : (x3 ? set_flow_lds(flow_rev_chain_kernel<3, true>, lds, "lfi_flow_sample_seq")
                            : set_flow_lds(flow_rev_chain_kernel<3, false>, lds, "lfi_flow_sample_seq")));
    if (rc) return rc;
  }
  for (int n = 0; n < nframes; ++n) {
    const int t = start + n;
    // c = LeakyReLU(pre_static[n] + window @ Wct[:, :hist1*C]^T), IN PLACE: frame n's rows of pre_static are read by this product
    // alone, so they are its pre-activation addend and its output at once (a 32 MB copy per frame into a separate c otherwise)
    float* cfr = pre_static + (long)n * B * Ks * D;
    lfi_gemm_desc q = {};
    q.batch = 1; q.M = B; q.N = Ks * D; q.K = hist1 * C;
    q.A = faces + (long)(t - hist1) * C; q.lda = (long)seq_len * C; q.a_kcontig = 1;
    q.B = wct + p1col; q.ldb = E; q.b_kcontig = 1;
    q.C = cfr; q.ldc = (long)Ks * D; q.accumulate = 2; q.act = 1; q.slope = 0.01f; q.precision = d->gemm_precision;
    if (p1kind != 0) {
      // features of the window first: e (B x hid4), then c = LeakyReLU(pre_static + e Wct[:, col : col + hid]^T)
      const int hid = p1->hid, hid4 = (hid + 3) & ~3;
      float* ebuf = p1work;                       // B x hid4
      if (p1kind == 1) {
        lfi_gemm_desc m = {};
        m.batch = 1; m.M = B; m.N = hid; m.K = hist1 * C;
        m.A = q.A; m.lda = q.lda; m.a_kcontig = 1;
        m.B = p1->w1; m.ldb = (long)hist1 * C; m.b_kcontig = 1;
        m.C = ebuf; m.ldc = hid4; m.bias = p1->b1; m.act = 1; m.slope = 0.01f; m.precision = d->gemm_precision;
        if ((rc = lfi_gemm_f32(&m, stream))) return rc;
      } else {
        // GRU / LSTM over the window: input projections of its hist1 frames (batched over the step), then the recurrence
        const int ng = p1kind == 3 ? 4 : 3;
        float* xp = ebuf + (long)B * hid4;          // [B][hist1][ng * hid]
        float* ework = xp + (long)B * hist1 * ng * hid;
        lfi_gemm_desc m = {};
        m.batch = hist1; m.M = B; m.N = ng * hid; m.K = C;
        m.A = q.A; m.lda = q.lda; m.a_kcontig = 1; m.strideA = C;
        m.B = p1->w_ih; m.ldb = C; m.b_kcontig = 1;
        m.C = xp; m.ldc = (long)hist1 * ng * hid; m.strideC = ng * hid; m.precision = d->gemm_precision;
        if ((rc = lfi_gemm_f32(&m, stream))) return rc;
        lfi_enc_desc ed = {};
        ed.B = B; ed.T = hist1; ed.N = 1; ed.start = hist1 - 1; ed.hist = hist1; ed.hid = hid;
        ed.ldcond = hid4; ed.col = 0; ed.precision = d->gemm_precision; ed.dup = 0; ed.lstm = p1kind == 3;
        float* hs = ework + lfi_encode_windows_work_floats(&ed);   // unfused path / LSTM: state sequence
        float* gst = p1kind == 3 ? hs + (long)hist1 * B * hid : nullptr;   // LSTM: gate + cell stash, 5 * hid per (step, row)
        if ((rc = lfi_encode_windows_fwd(&ed, xp, p1->w_hh, p1->b_ih, p1->b_hh, nullptr, ebuf, gst, hs, ework, stream)))
          return rc;
      }
      q.K = hid; q.A = ebuf; q.lda = hid4;
    }
    if (fused) {
      // (its first workgroup also clears the reverse chain's ticket / progress words for the launch below: no memset node per frame)
      // (from the run's second frame on the window's fragments are already there: the previous frame's chain left them)
      if ((rc = lfi_internal_sample_cond(faces, (long)seq_len * C, (long)(t - hist1) * C, K1, B, Ks, G, cfr, p->b_ih, cfrags, gic, 0.01f,
                                         (long)B * seq_len * C, chain ? chain_state : nullptr, (int)chain_words,
                                         (xf_chain && n > 0) ? 1 : 0, stream)))
        return rc;
    } else {
      if (stage_win) {
        if ((rc = lfi_gather_windows(faces, B, seq_len, C, 1, t, hist1, 0, nullptr, wstage, ldw, 0, stream))) return rc;
        q.A = wstage; q.lda = ldw;
      }
      if ((rc = lfi_gemm_f32(&q, stream))) return rc;
      // gic[k] = c[:, kD:(k+1)D] @ W_ih[k][:, Ch:]^T + b_ih[k]
      lfi_gemm_desc r = {};
      r.batch = Ks; r.M = B; r.N = G; r.K = D;
      r.A = cfr; r.lda = (long)Ks * D; r.a_kcontig = 1; r.strideA = D;
      r.B = f.wc; r.ldb = D; r.b_kcontig = 1; r.strideB = (long)G * D;
      r.C = gic; r.ldc = G; r.strideC = (long)B * G;
      r.bias = p->b_ih; r.strideBias = G; r.precision = d->gemm_precision;
      if ((rc = lfi_gemm_f32(&r, stream))) return rc;
    }
    // reverse flow: z -> x through steps Ks-1 .. 0
    if (chain) {   // one launch for the whole chain of this frame
      RevChain rcn = {};
      rcn.noise = noise + (long)n * B * C; rcn.xa = xa; rcn.xb = xb;
      rcn.frame = faces + (long)t * C; rcn.ld_frame = (long)seq_len * C;
      rcn.gic = gic; rcn.h = h; rcn.cstate = cstate; rcn.has_prev = first_frame + n > 0 ? 1 : 0; rcn.frame_no = first_frame + n; rcn.pipe = chain_state;
      if (xf_chain && n + 1 < nframes) {
        rcn.xf = reinterpret_cast<_Float16*>(lfi_internal_sample_cond_xfrag_ptr(cfrags, Ks, G, K1));
        rcn.faces = faces; rcn.xf_off = (long)(t + 1 - hist1) * C; rcn.K1 = K1; rcn.NM1 = (K1 + 31) / 32;
      }
      if (!fused) {   // (the fused conditioning kernel has cleared them)
        hipError_t me = hipMemsetAsync(chain_state, 0, chain_words * sizeof(unsigned), st);
        LFI_REQUIRE(me == hipSuccess, "lfi_flow_sample_seq: hipMemsetAsync: %s", hipGetErrorString(me));
      }
      if (f.lstm) hipLaunchKernelGGL((flow_rev_chain_kernel<4, false>), dim3(Ks * f.nbt), dim3(NT), lds, st, f, rcn);
      else if (xw) hipLaunchKernelGGL((flow_rev_chain_kernel<3, true, true>), dim3(Ks * f.nbt), dim3(NT), lds, st, f, rcn);
      else if (x3) hipLaunchKernelGGL((flow_rev_chain_kernel<3, true>), dim3(Ks * f.nbt), dim3(NT), lds, st, f, rcn);
      else hipLaunchKernelGGL((flow_rev_chain_kernel<3, false>), dim3(Ks * f.nbt), dim3(NT), lds, st, f, rcn);
      continue;
    }
    const float* xin = noise + (long)n * B * C;
    long ldx = C;
    for (int k = Ks - 1; k >= 0; --k) {
      CellIO io = {};
      io.k = k; io.rows = B; io.x_in = xin; io.ldx = ldx;
      io.h_prev = first_frame + n > 0 ? h + (long)k * B * H : nullptr;
      io.gic = gic + (long)k * B * G;
      io.h_out = h + (long)k * B * H;
      if (f.lstm) { io.c_prev = first_frame + n > 0 ? cstate + (long)k * B * H : nullptr; io.c_out = cstate + (long)k * B * H; }
      if (k == 0) { io.x_out = faces + (long)t * C; io.ldxo = (long)seq_len * C; }
      else { io.x_out = (k & 1) ? xa : xb; io.ldxo = C; }
      if (!fast) hipLaunchKernelGGL(flow_step_kernel<true>, dim3(f.nbt), dim3(NT), lds, st, f, io);
      else if (f.lstm) hipLaunchKernelGGL(flow_step_rev_fast_kernel<4>, dim3(f.nbt), dim3(NT), lds, st, f, io);
      else hipLaunchKernelGGL(flow_step_rev_fast_kernel<3>, dim3(f.nbt), dim3(NT), lds, st, f, io);
      xin = io.x_out; ldx = io.ldxo;
    }
  }
  LFI_LAUNCH_CHECK("lfi_flow_sample_seq");
  return LFI_OK;
}

// ---------------------------------------------------------------------------------------------- stand-alone module calls
// What the reference's test_modules.py:9-29 pokes directly: ActNorm2d and InvertibleConv1x1 outside any flow.
extern "C" int lfi_actnorm_forward(const float* x, int rows, int C, const float* bias, const float* logs, int reverse, float* out,
                                   float* dlogdet, void* stream) {
  LFI_REQUIRE(x && bias && logs && out && rows > 0 && C > 0, "lfi_actnorm_forward: bad arguments");
  const long total = (long)rows * C;
  const int blocks = (int)(lfi_cdiv(total, 256) < 2048 ? lfi_cdiv(total, 256) : 2048);
  hipLaunchKernelGGL(actnorm_module_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)rows, C, bias, logs, reverse, out,
                     dlogdet);
  LFI_LAUNCH_CHECK("lfi_actnorm_forward");
  return LFI_OK;
}

extern "C" long lfi_invconv_work_floats(int C) {
  if (C <= 0) return 0;
  return 2L * (2L * C * C + C) + (long)C * C + C + 16;   // fp64 scratch (as floats), W^T, a zero ActNorm row
}

extern "C" int lfi_invconv_weights(int C, const float* inv_l, const float* inv_u, const float* inv_logs, const float* inv_p,
                                   const float* inv_sign, const float* inv_w, int with_inverse, float* W, float* Winv,
                                   float* dlogdet, float* work, void* stream) {
  LFI_REQUIRE(C >= 2 && W && dlogdet && work, "lfi_invconv_weights: bad arguments");
  LFI_REQUIRE(inv_w || (inv_l && inv_u && inv_logs && inv_p && inv_sign), "lfi_invconv_weights: invconv parameters missing");
  LFI_REQUIRE(!with_inverse || Winv, "lfi_invconv_weights: with_inverse needs Winv");
  LFI_REQUIRE(((uintptr_t)work & 7) == 0, "lfi_invconv_weights: work must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  double* dscratch = reinterpret_cast<double*>(work);
  float* Wt = work + 2L * (2L * C * C + C);
  float* zeros = Wt + (long)C * C;
  hipError_t me = hipMemsetAsync(zeros, 0, sizeof(float) * (size_t)C, st);
  LFI_REQUIRE(me == hipSuccess, "lfi_invconv_weights: hipMemsetAsync: %s", hipGetErrorString(me));
  FlowK f = {};
  f.C = C; f.Ks = 1;
  f.p.inv_l = inv_l; f.p.inv_u = inv_u; f.p.inv_logs = inv_logs; f.p.inv_p = inv_p; f.p.inv_sign = inv_sign; f.p.inv_w = inv_w;
  f.p.an_logs = zeros;   // the kernel adds the ActNorm part of the constant log-det: none here
  const size_t lds = (size_t)3 * C * C * sizeof(float);
  int rc = set_flow_lds(flow_prep_invconv_kernel, lds, "lfi_invconv_weights");
  if (rc) return rc;
  hipLaunchKernelGGL(flow_prep_invconv_kernel, dim3(1), dim3(256), lds, st, f, W, Wt, (with_inverse || inv_w) ? Winv : nullptr, dscratch,
                     dlogdet);
  LFI_LAUNCH_CHECK("lfi_invconv_weights");
  return LFI_OK;
}
