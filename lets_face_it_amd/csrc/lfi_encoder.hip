// Window encoders: the three per-timestep nn.GRU calls of ModalityEncoder.forward (glow/models.py:55-80) for ALL
// (sample, timestep) windows of a training sequence at once, forward and BPTT.
//
// The reference re-encodes an almost identical window from h0 = 0 at every timestep (3 cuDNN GRU launches per
// timestep, each on B rows). Here every window is one row of a (F = N*B) x hid state matrix and the `hist` GRU steps
// advance ALL F windows together:
//   - the input projection x W_ih^T is hoisted out (one GEMM over the B*T distinct frames, shared by every window
//     that covers a frame; the dropout mask is one scalar per (window, step) so it commutes with the projection),
//   - each step is one MFMA GEMM h_{s-1} (F x hid) @ W_hh^T on lfi_gemm_f32 plus one flat, fully coalesced gate
//     kernel that also writes the stash for BPTT,
//   - stashes are step-major ([s][w][...]) so the deferred weight-gradient GEMMs see plain row ranges
//     (dW_hh = dgh[1:]^T hseq[:-1]).
// Backward mirrors it: per step one gate-derivative kernel and one GEMM dgh_s (F x 3hid) @ W_hh accumulated onto the
// carried gradient.
#include "lfi_common.h"
#include <type_traits>

// (LFI_ENC_EXP_M16: timing-only experiment, garbage results - the recurrences' MFMAs as two v_mfma_f32_16x16x32_bf16 on the same
// registers, to see whether the chip holds a higher clock on that shape here as it does in the planes GEMMs)
#ifdef LFI_ENC_EXP_M16
#define ENC_MFMA(a, b, c, x, y, z) enc_mfma16_probe(a, b, c)
#else
#define ENC_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#endif
namespace {
#ifdef LFI_ENC_EXP_M16
template <typename AV, typename BV>
__device__ __forceinline__ f32x16 enc_mfma16_probe(AV a, BV b, f32x16 c) {
  f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]};
  q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q0, 0, 0, 0);
  q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q1, 0, 0, 0);
  c[0] = q0[0]; c[1] = q0[1]; c[2] = q0[2]; c[3] = q0[3];
  c[4] = q1[0]; c[5] = q1[1]; c[6] = q1[2]; c[7] = q1[3];
  return c;
}
#endif


struct EncArgs {
  int B, T, N, start, hist, hid, ldcond, col, dup;
  int lstm, G;         // LSTM window encoder (gate blocks i, f, g, o); G = (lstm ? 4 : 3) * hid
  int compact;         // fused GRU backward: dgi holds only its n-gate block ([hist][F][hid]); its r and z blocks equal dgh's
  int g16;             // dgi / dgh are bf16 arrays of the same shapes (lfi_enc_desc.bwd_two_products with the wide fused backward)
  int F;
  const float* Xp;     // (B*T) x 3hid
  const float* b_ih;
  const float* b_hh;
  const float* mask;   // F x hist or null
  float* cond;
  float* gates;        // [hist][F][4][hid]
  float* hseq;         // [hist][F][hid]
  const float* dcond; int lddcond;
  float* dgi;          // [hist][F][3hid]
  float* dgh;          // [hist][F][3hid]
  float* bias_part;    // fused backward only: [workgroups * row groups][4][hid] partial sums of dar, dau, dan, dan*r (or null)
  unsigned long long* stamps;  // diagnostics only
};
// phase stamps of the fused forward kernel: compiled in only with -DLFI_ENC_STAMPS (they cost registers: 321 spills)
#ifdef LFI_ENC_STAMPS
#define ENC_STAMP(slot)                                                                                          \
  do {                                                                                                           \
    if (a.stamps && tid == 0 && blockIdx.x == 0 && s >= 4 && s < 12)                                              \
      a.stamps[128 + 8 * (s - 4) + (slot)] = __builtin_amdgcn_s_memtime();                                        \
  } while (0)
#else
#define ENC_STAMP(slot) do { } while (0)
#endif

// One GRU step for every window. gh: F x 3hid = h_{s-1} W_hh^T (no bias), or null at s = 0 (h = 0).
__global__ __launch_bounds__(256) void enc_gate_fwd_kernel(EncArgs a, int s, const float* __restrict__ gh) {
  const int hid = a.hid, G3 = 3 * hid;
  const long total = (long)a.F * hid;
  const int pos0 = a.start - a.hist + 1;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int w = (int)(idx / hid), j = (int)(idx - (long)w * hid);
    const int n = w / a.B, b = w - n * a.B;
    const float* xp = a.Xp + ((long)b * a.T + (pos0 + n + s)) * G3;
    const float mk = a.mask ? a.mask[(long)w * a.hist + s] : 1.0f;
    float ghr = a.b_hh[j], ghu = a.b_hh[hid + j], ghn = a.b_hh[2 * hid + j];
    float hp = 0.0f;
    if (gh) {
      const float* g = gh + (long)w * G3;
      ghr += g[j]; ghu += g[hid + j]; ghn += g[2 * hid + j];
      hp = a.hseq[((long)(s - 1) * a.F + w) * hid + j];
    }
    const float rr = sigmoidf_(mk * xp[j] + a.b_ih[j] + ghr);
    const float uu = sigmoidf_(mk * xp[hid + j] + a.b_ih[hid + j] + ghu);
    const float nn = tanhf_(mk * xp[2 * hid + j] + a.b_ih[2 * hid + j] + rr * ghn);
    const float hnew = (1.0f - uu) * nn + uu * hp;
    const long sw = (long)s * a.F + w;
    a.hseq[sw * hid + j] = hnew;
    if (a.gates) {
      float* gs = a.gates + sw * 4 * hid;
      gs[j] = rr; gs[hid + j] = uu; gs[2 * hid + j] = nn; gs[3 * hid + j] = ghn;
    }
    if (s == a.hist - 1) {  // cat(seq[:, -1], h_n[0]): the same vector twice (glow/models.py:63-64)
      float* c = a.cond + (long)w * a.ldcond + a.col;
      c[j] = hnew;
      if (a.dup) c[hid + j] = hnew;
    }
  }
}

// Gate derivatives of step s. dh_in: F x hid gradient w.r.t. h_s (null at s = hist-1: taken from dcond, both halves);
// writes dgi[s], dgh[s] and the carried part dh_out = dh * z (the GEMM then adds dgh_s W_hh onto it).
__global__ __launch_bounds__(256) void enc_gate_bwd_kernel(EncArgs a, int s, const float* __restrict__ dh_in,
                                                           float* __restrict__ dh_out) {
  const int hid = a.hid, G3 = 3 * hid;
  const long total = (long)a.F * hid;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int w = (int)(idx / hid), j = (int)(idx - (long)w * hid);
    float dhn;
    if (dh_in) dhn = dh_in[idx];
    else {
      const float* dc = a.dcond + (long)w * a.lddcond + a.col;
      dhn = a.dup ? dc[j] + dc[hid + j] : dc[j];
    }
    const long sw = (long)s * a.F + w;
    const float* gs = a.gates + sw * 4 * hid;
    const float rr = gs[j], uu = gs[hid + j], nn = gs[2 * hid + j], ghn = gs[3 * hid + j];
    const float hp = s > 0 ? a.hseq[((long)(s - 1) * a.F + w) * hid + j] : 0.0f;
    const float du = dhn * (hp - nn);
    const float dn = dhn * (1.0f - uu);
    const float dan = dn * (1.0f - nn * nn);
    const float dau = du * uu * (1.0f - uu);
    const float dar = dan * ghn * rr * (1.0f - rr);
    float* gi = a.dgi + sw * G3;
    gi[j] = dar; gi[hid + j] = dau; gi[2 * hid + j] = dan;
    float* gh = a.dgh + sw * G3;
    gh[j] = dar; gh[hid + j] = dau; gh[2 * hid + j] = dan * rr;
    if (dh_out) dh_out[idx] = dhn * uu;
  }
}

// ---- "enc: lstm" modality (glow/models.py:27-33,65-69): nn.LSTM from zero (h, c) over the window, gate blocks i, f, g, o
// (torch.nn.LSTM weight layout). Per step one GEMM gh = h_{s-1} W_hh^T (F x 4hid) + this kernel. The stash row of a
// (step, window) is 5 * hid floats: i, f, g, o, c_s.
__global__ __launch_bounds__(256) void enc_lstm_gate_fwd_kernel(EncArgs a, int s, const float* __restrict__ gh) {
  const int hid = a.hid, G = 4 * hid;
  const long total = (long)a.F * hid;
  const int pos0 = a.start - a.hist + 1;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int w = (int)(idx / hid), j = (int)(idx - (long)w * hid);
    const int n = w / a.B, b = w - n * a.B;
    const float* xp = a.Xp + ((long)b * a.T + (pos0 + n + s)) * G;
    const float mk = a.mask ? a.mask[(long)w * a.hist + s] : 1.0f;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pre[q] = mk * xp[q * hid + j] + a.b_ih[q * hid + j] + a.b_hh[q * hid + j];
    float cp = 0.0f;
    if (gh) {
      const float* g = gh + (long)w * G;
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] += g[q * hid + j];
      cp = a.gates[(((long)(s - 1) * a.F + w) * 5 + 4) * hid + j];
    }
    const float ii = sigmoidf_(pre[0]), ff = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), oo = sigmoidf_(pre[3]);
    const float c = ff * cp + ii * gg;
    const float hnew = oo * tanhf_(c);
    const long sw = (long)s * a.F + w;
    a.hseq[sw * hid + j] = hnew;
    float* gs = a.gates + sw * 5 * hid;
    gs[j] = ii; gs[hid + j] = ff; gs[2 * hid + j] = gg; gs[3 * hid + j] = oo; gs[4 * hid + j] = c;
    if (s == a.hist - 1) {  // cat(seq[:, -1], h_n[0]): the same vector twice (glow/models.py:68-69)
      float* cc = a.cond + (long)w * a.ldcond + a.col;
      cc[j] = hnew;
      if (a.dup) cc[hid + j] = hnew;
    }
  }
}

// Gate derivatives of LSTM step s. dh_in / dc_in: F x hid gradients w.r.t. h_s / c_s carried from step s + 1 (null at the last
// step: d h from dcond, d c = 0). Writes dgi[s] = dgh[s] (the pre-activation is a plain sum of both sides), the carried d c
// and zeroes dh_out, onto which the GEMM dgh_s W_hh then accumulates d h_{s-1}.
__global__ __launch_bounds__(256) void enc_lstm_gate_bwd_kernel(EncArgs a, int s, const float* __restrict__ dh_in,
                                                                const float* __restrict__ dc_in, float* __restrict__ dh_out,
                                                                float* __restrict__ dc_out) {
  const int hid = a.hid, G = 4 * hid;
  const long total = (long)a.F * hid;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int w = (int)(idx / hid), j = (int)(idx - (long)w * hid);
    float dhn;
    if (dh_in) dhn = dh_in[idx];
    else {
      const float* dc = a.dcond + (long)w * a.lddcond + a.col;
      dhn = a.dup ? dc[j] + dc[hid + j] : dc[j];
    }
    const long sw = (long)s * a.F + w;
    const float* gs = a.gates + sw * 5 * hid;
    const float ii = gs[j], ff = gs[hid + j], gg = gs[2 * hid + j], oo = gs[3 * hid + j], c = gs[4 * hid + j];
    const float cp = s > 0 ? a.gates[(((long)(s - 1) * a.F + w) * 5 + 4) * hid + j] : 0.0f;
    const float tc = tanhf_(c);
    const float dc = dhn * oo * (1.0f - tc * tc) + (dc_in ? dc_in[idx] : 0.0f);
    const float d0 = dc * gg * ii * (1.0f - ii);
    const float d1 = dc * cp * ff * (1.0f - ff);
    const float d2 = dc * ii * (1.0f - gg * gg);
    const float d3 = dhn * tc * oo * (1.0f - oo);
    float* gi = a.dgi + sw * G;
    float* gh = a.dgh + sw * G;
    gi[j] = d0; gi[hid + j] = d1; gi[2 * hid + j] = d2; gi[3 * hid + j] = d3;
    gh[j] = d0; gh[hid + j] = d1; gh[2 * hid + j] = d2; gh[3 * hid + j] = d3;
    if (dh_out) { dh_out[idx] = 0.0f; dc_out[idx] = dc * ff; }
  }
}

// dXp[b*T + p][c] = sum_{s} mask[w(n,b)][s] * dgi[s][w][c],  n = p - pos0 - s in [0, N)
// One workgroup per frame row; every (step, window) row of dgi is read exactly once over the grid (HBM-bound stream).
// Branch-free: steps whose window falls outside [0, N) read row 0 with weight 0, so eight 16-byte loads per thread are
// in flight at a time.
__global__ __launch_bounds__(256) void enc_scatter_kernel(EncArgs a, float* __restrict__ dXp) {
  const int G3 = a.G;   // 3 * hid (GRU) or 4 * hid (LSTM)
  const int row = blockIdx.x;  // b*T + p
  const int b = row / a.T, p = row - b * a.T;
  const int pos0 = a.start - a.hist + 1;
  const bool vec = (G3 & 3) == 0 && (a.hid & 3) == 0;
  const int nvec = vec ? G3 >> 2 : 0;
  for (int c4 = threadIdx.x; c4 < nvec; c4 += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < a.hist; s0 += 8) {
      f32x4 v[8];
      float wgt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u;
        const int n = p - pos0 - s;
        const bool ok = s < a.hist && n >= 0 && n < a.N;
        const long w = ok ? (long)n * a.B + b : 0;
        const int sc = ok ? s : 0;
        wgt[u] = ok ? (a.mask ? a.mask[w * a.hist + sc] : 1.0f) : 0.0f;
        const long r = (long)sc * a.F + w;
        if (a.g16) {   // bf16 stash (compact): 4 values = one 8-byte load
          const __bf16* gh16 = reinterpret_cast<const __bf16*>(a.dgh), *gi16 = reinterpret_cast<const __bf16*>(a.dgi);
          const __bf16* src = 4 * c4 < 2 * a.hid ? gh16 + r * G3 + 4 * c4 : gi16 + r * a.hid + (4 * c4 - 2 * a.hid);
          const uint2 w2 = *reinterpret_cast<const uint2*>(src);
          v[u] = f32x4{__builtin_bit_cast(float, w2.x << 16), __builtin_bit_cast(float, w2.x & 0xffff0000u),
                       __builtin_bit_cast(float, w2.y << 16), __builtin_bit_cast(float, w2.y & 0xffff0000u)};
        } else {
          const float* src = !a.compact ? a.dgi + r * G3 + 4 * c4
                                        : (4 * c4 < 2 * a.hid ? a.dgh + r * G3 + 4 * c4 : a.dgi + r * a.hid + (4 * c4 - 2 * a.hid));
          v[u] = *reinterpret_cast<const f32x4*>(src);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += wgt[u] * v[u];
    }
    *reinterpret_cast<f32x4*>(dXp + (long)row * G3 + 4 * c4) = acc;
  }
  if (!vec)
    for (int c = threadIdx.x; c < G3; c += 256) {
      float acc = 0.0f;
      for (int s = 0; s < a.hist; ++s) {
        const int n = p - pos0 - s;
        if (n < 0 || n >= a.N) continue;
        const long w = (long)n * a.B + b;
        const float mk = a.mask ? a.mask[w * a.hist + s] : 1.0f;
        const long r = (long)s * a.F + w;
        acc += mk * (!a.compact ? a.dgi[r * G3 + c] : (c < 2 * a.hid ? a.dgh[r * G3 + c] : a.dgi[r * a.hid + (c - 2 * a.hid)]));
      }
      dXp[(long)row * G3 + c] = acc;
    }
}

// The same sums for the bf16 compact stash with hid % 8 == 0: 8 values per thread through ONE 16-byte load, and as many frame rows
// per workgroup as 256 threads hold (3 hid / 8 threads per row). Same products, same order over s: bit-identical to the kernel above.
typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void enc_scatter16_kernel(EncArgs a, float* __restrict__ dXp, int rows, int cpr, int rpw) {
  const int rl = threadIdx.x / cpr, c = threadIdx.x - rl * cpr;
  const int row = blockIdx.x * rpw + rl;
  if (rl >= rpw || row >= rows) return;
  const int b = row / a.T, p = row - b * a.T;
  const int pos0 = a.start - a.hist + 1;
  const int c8 = 8 * c;
  const bool hh = c8 < 2 * a.hid;
  const __bf16* base = hh ? reinterpret_cast<const __bf16*>(a.dgh) + c8 : reinterpret_cast<const __bf16*>(a.dgi) + (c8 - 2 * a.hid);
  const long ldr = hh ? a.G : a.hid;
  constexpr int UB = 8;   // loads in flight per thread (4 / 8 / 12 and non-temporal loads measured alike: tools/scatter_probe.py)
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < a.hist; s0 += UB) {
    su32x4 v[UB];
    float wgt[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int s = s0 + u;
      const int n = p - pos0 - s;
      const bool ok = s < a.hist && n >= 0 && n < a.N;
      const long w = ok ? (long)n * a.B + b : 0;
      const int sc = ok ? s : 0;
      wgt[u] = ok ? (a.mask ? a.mask[w * a.hist + sc] : 1.0f) : 0.0f;
      const su32x4* src = reinterpret_cast<const su32x4*>(base + ((long)sc * a.F + w) * ldr);
      v[u] = *src;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const unsigned q[4] = {v[u][0], v[u][1], v[u][2], v[u][3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += wgt[u] * __builtin_bit_cast(float, q[e] << 16);
        acc[2 * e + 1] += wgt[u] * __builtin_bit_cast(float, q[e] & 0xffff0000u);
      }
    }
  }
  float* dst = dXp + (long)row * a.G + c8;
  *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

__global__ __launch_bounds__(256) void gather_windows_kernel(const float* __restrict__ X, int B, int T, int dim, int N, int start,
                                                             int hist, int incl, const float* __restrict__ mask,
                                                             float* __restrict__ cond, int ldcond, int col) {
  const int f = blockIdx.x;  // n*B + b
  const int n = f / B, b = f - n * B;
  const int t0 = start + n - hist + incl;
  const int tot = hist * dim;
  // window rows are consecutive frames of one sample: one contiguous run of hist*dim floats
  const float* src = X + ((long)b * T + t0) * dim;
  float* dst = cond + (long)f * ldcond + col;
  if (mask) {  // dropout multiplier per (window, history step) (glow/models.py:56-58)
    const float* mk = mask + (long)f * hist;
    for (int i = threadIdx.x; i < tot; i += 256) dst[i] = src[i] * mk[i / dim];
  } else {
    for (int i = threadIdx.x; i < tot; i += 256) dst[i] = src[i];
  }
}

// FeatureEncoder's optional frame-counter column (glow/models.py:89,116-117,143-144): frame f = n*B + b gets base[b] + offset + 2n
// (SeqGlow.forward / invert: base = batch["frame_nb"], offset = 2 * start, :539-542,557-558; inference: base = 1, offset 0, :572-575)
__global__ __launch_bounds__(256) void fill_frame_nb_kernel(const float* __restrict__ base, float offset, int B, long F,
                                                            float* __restrict__ cond, int ldcond, int col) {
  for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < F; f += (long)gridDim.x * 256) {
    const long n = f / B;
    const int b = (int)(f - n * B);
    cond[f * ldcond + col] = (base ? base[b] : 1.0f) + offset + 2.0f * (float)n;
  }
}

__global__ __launch_bounds__(256) void leaky_grad_kernel(float* __restrict__ d, long ldd, const float* __restrict__ y, long ldy, int rows,
                                                         int cols, float slope) {
  const long total = (long)rows * cols;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / cols;
    const int c = (int)(idx - r * cols);
    if (!(y[r * ldy + c] > 0.0f)) d[r * ldd + c] *= slope;
  }
}


// ---------------------------------------------------------------------------------------------- fused persistent path
// For hid <= 256 the whole window recurrence of a block of windows runs inside ONE workgroup: windows are independent
// of one another, so a workgroup owns R windows, keeps their state h (R x hid) in LDS across all `hist` steps and
// per step does   gh = h_{s-1} W_hh^T  on v_mfma_f32_32x32x2_f32 (A = h from LDS k-major, B = W_hh^T streamed from L2
// straight into the MFMA operand registers, chunk-prefetched) + the gate math on the accumulators. Nothing but the
// stashes the backward pass needs goes to HBM (the unfused path also wrote and re-read gh, F x 3hid, every step, and
// paid one GEMM + one gate launch per step). One launch per modality instead of 2 * hist.
//   workgroup = 4 waves; wave (rg, cg) owns rows [32 rg, 32 rg + 32) x hidden [64 cg, 64 cg + 64) x the 3 gates
//   (6 accumulator tiles); ncg = column groups = next_pow2(ceil(hid / 64)) <= 4, R = 32 * 4 / ncg rows per workgroup;
//   LDS = Kp x (R + 1) floats <= 33 KB and <= 256 VGPRs: two workgroups share a CU and one's epilogue (HBM stores)
//   overlaps the other's MFMAs.
// The weights are re-laid once per call into a zero-padded image (Kp = hid rounded up to 16 rows of k, Jp = 64 ncg
// columns) so the k loop carries no bounds checks: every load in it is unconditional.
#ifndef LFI_ENC_NT
#define LFI_ENC_NT 256   // (512 = one 64-window workgroup per CU: measured slower, 0.90 vs 0.78 ms on the p2_face forward launch - no L1 sharing of the weight stream)
#endif
constexpr int ENC_NT = LFI_ENC_NT;   // threads per workgroup of the fused kernels
constexpr int ENC_NW = ENC_NT / 64;  // waves: ncg column groups x ENC_NW / ncg row groups of 32 windows
constexpr int ENC_KC = 4;  // k-pairs per prefetch chunk (two chunks in flight)
extern __shared__ __attribute__((aligned(16))) float enc_smem[];

struct EncFused {
  int ncg, R, Kp, Jp;
  const float* wpad;   // forward: [Kp][3][Jp] = W_hh^T ; backward: [3][Kp][Jp] = W_hh, zero padded
  const uint4* wfrag;  // bf16x3 mode: the same matrices split into bf16 hi / lo planes in MFMA B-fragment order
};

// bf16x3 variant of the fused path (lfi_enc_desc.precision = 1): the state is split into bf16 hi + lo when it is written
// to LDS (row-major [R][Kp + 8] images, one ds_read_b128 per A fragment) and every 32x32x16 step issues
// lo*hi + hi*lo + hi*hi into the fp32 accumulators (~2^-16 relative per product, 16x the f32-input MFMA rate). The
// weights are split ONCE per call into fragment order: the 8 bf16 a lane feeds to v_mfma_f32_32x32x16_bf16 are 16
// contiguous bytes, a wave's fragment 1 KB: one coalesced global_load_dwordx4 per (k-tile, column tile, plane).
typedef __bf16 ebf16x8 __attribute__((ext_vector_type(8)));
union EncFrag { uint4 u; ebf16x8 v; };
// (hi, lo) bf16 pairs of (a, b), packed low / high half: hi = RNE(v), lo = RNE(v - hi) - as `(__bf16)v` element by element
typedef __bf16 ebf16x2 __attribute__((ext_vector_type(2)));
typedef float efloat2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned* hi, unsigned* lo) {
  const ebf16x2 h = __builtin_convertvector((efloat2){a, b}, ebf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float ha = __builtin_bit_cast(float, hb << 16), hbv = __builtin_bit_cast(float, hb & 0xffff0000u);
  const ebf16x2 l = __builtin_convertvector((efloat2){a - ha, b - hbv}, ebf16x2);
  *hi = hb;
  *lo = __builtin_bit_cast(unsigned, l);
}

// element ((((kt*3 + g)*nct + ct)*2 + plane)*64 + lane)*8 + e  (fwd)  /  ((((g*nkt + kt)*nct + ct)*2 + plane)*64 + lane)*8 + e (bwd)
// holds k = kt*16 + 8*(lane>>5) + e, column j = ct*32 + (lane&31) of W_hh^T[k][g*hid + j] (fwd) / W_hh[g*hid + k][j] (bwd)
__global__ __launch_bounds__(256) void enc_frag_weights_kernel(const float* __restrict__ whh, int hid, int Kp, int Jp, int fwd,
                                                               __bf16* __restrict__ dst) {
  const int nkt = Kp >> 4, nct = Jp >> 5;
  const long n = 3L * Kp * Jp * 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63), plane = (int)((idx >> 9) & 1);
    long q = idx >> 10;
    const int ct = (int)(q % nct); q /= nct;
    int g, kt;
    if (fwd) { g = (int)(q % 3); kt = (int)(q / 3); } else { kt = (int)(q % nkt); g = (int)(q / nkt); }
    const int k = kt * 16 + 8 * (l >> 5) + e, j = ct * 32 + (l & 31);
    float v = 0.0f;
    if (k < hid && j < hid) v = fwd ? whh[((long)g * hid + j) * hid + k] : whh[((long)g * hid + k) * hid + j];
    const __bf16 hi = (__bf16)v;
    dst[idx] = plane ? (__bf16)(v - (float)hi) : hi;
  }
}

// Fragment order of v_mfma_f32_16x16x32_bf16 for the forward recurrence (enc_gru_fwd_r64_kernel<.., M16 = true>; Kp a multiple of 256):
// element ((((m*3 + g)*nct + ct)*2 + plane)*64 + lane)*8 + e holds column j = ct*16 + (lane&15) of W_hh^T[k][g*hid + j] at
// k = 8 enc_chunk16(m, lane>>4, Kp/8) + e: lane group q of MFMA m takes the 16-byte k-chunk 2m + (q>>1) of the first (q even) or
// second (q odd) half of the row - the chunks of the groups that share a ds_read_b128 cycle lie 256 bytes apart in the state
// images, whose rows (Kp + 8 bf16) advance by one 16-byte slot: all 64 banks, no conflict.
__host__ __device__ inline int enc_chunk16(int m, int q, int nchunk) { return 2 * m + (q >> 1) + (nchunk >> 1) * (q & 1); }
__global__ __launch_bounds__(256) void enc_frag_weights16_kernel(const float* __restrict__ whh, int hid, int Kp, int Jp, __bf16* __restrict__ dst) {
  const int nct = Jp >> 4, nchunk = Kp >> 3;
  const long n = 3L * Kp * Jp * 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63), plane = (int)((idx >> 9) & 1);
    long q = idx >> 10;
    const int ct = (int)(q % nct); q /= nct;
    const int g = (int)(q % 3), m = (int)(q / 3);
    const int k = 8 * enc_chunk16(m, l >> 4, nchunk) + e, j = ct * 16 + (l & 15);
    float v = 0.0f;
    if (k < hid && j < hid) v = whh[((long)g * hid + j) * hid + k];
    const __bf16 hi = (__bf16)v;
    dst[idx] = plane ? (__bf16)(v - (float)hi) : hi;
  }
}

// fwd = 1: dst[(k*3 + g)*Jp + j] = whh[(g*hid + j)*hid + k] ; fwd = 0: dst[(g*Kp + k)*Jp + j] = whh[(g*hid + k)*hid + j]
__global__ __launch_bounds__(256) void enc_pad_weights_kernel(const float* __restrict__ whh, int hid, int Kp, int Jp, int fwd,
                                                              float* __restrict__ dst) {
  const long n = 3L * Kp * Jp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int j = (int)(idx % Jp);
    const int q = (int)(idx / Jp);
    const int g = fwd ? q % 3 : q / Kp, k = fwd ? q / 3 : q % Kp;
    float v = 0.0f;
    if (k < hid && j < hid) v = fwd ? whh[((long)g * hid + j) * hid + k] : whh[((long)g * hid + k) * hid + j];
    dst[idx] = v;
  }
}

__device__ __forceinline__ int enc_rowl(int rg, int r, int half) { return rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; }

// Raw buffer access for the gate epilogues: a uniform 128-bit resource (base + size) in SGPRs, one 32-bit byte offset per
// lane and a uniform SGPR offset on top. The plain-pointer form cost ~20 VALU instructions of 64-bit address arithmetic
// per access (60 % of the epilogue's instructions; rocprof: these kernels are VALU-issue bound, not HBM or MFMA bound).
// Accesses past `bytes` read 0 / are dropped by the hardware.
typedef __amdgpu_buffer_rsrc_t enc_rsrc;
__device__ __forceinline__ enc_rsrc enc_buf(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xfffffff0L ? 0xfffffff0L : bytes), 0x00020000);
}
__device__ __forceinline__ float enc_ld(enc_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// read-once stash values in the backward kernel (1.76 GB per p2_face launch): streamed (aux 2 = non-temporal), so that they do
// not sweep the weight fragments every workgroup re-reads each step out of L2
#ifndef LFI_ENC_LDS_AUX
#define LFI_ENC_LDS_AUX 2
#endif
__device__ __forceinline__ float enc_ld_stream(enc_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, LFI_ENC_LDS_AUX));
}
// The stashes are written once and read by a much later kernel (BPTT, the weight-gradient GEMMs): non-temporal (aux 2), so
// 1.5 GB of them per launch do not sweep the weights and the projected inputs out of L2.
#ifndef LFI_ENC_ST_AUX
#define LFI_ENC_ST_AUX 2
#endif
__device__ __forceinline__ void enc_st(float v, enc_rsrc r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, LFI_ENC_ST_AUX);
}

template <bool STASH, bool MASK, bool X3>
__global__ __launch_bounds__(ENC_NT, 2) void enc_gru_fwd_fused_kernel(EncArgs a, EncFused q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cg = wave % q.ncg, rg = wave / q.ncg;
  const int hid = a.hid, G3 = 3 * hid, ldk = q.R + 1, Jp = q.Jp;
  const int jb = cg * 64 + l31;               // hidden index of sub-tile 0; sub-tile 1 is jb + 32
  const int wbase = blockIdx.x * q.R;
  const int pos0 = a.start - a.hist + 1;
  float* Als = enc_smem;  // h_{s-1}: element (row, k) at Als[k * ldk + row], rows k >= hid stay zero
  for (int i = tid; i < q.Kp * ldk; i += ENC_NT) Als[i] = 0.0f;
  // X3: bf16 hi / lo images of the same state, row-major, columns k >= hid stay zero
  const int ldx = q.Kp + 8;
  __bf16* Xhi = reinterpret_cast<__bf16*>(enc_smem + q.Kp * ldk);
  __bf16* Xlo = Xhi + q.R * ldx;
  if (X3)
    for (int i = tid; i < q.R * ldx; i += ENC_NT) { Xhi[i] = (__bf16)0.0f; Xlo[i] = (__bf16)0.0f; }
  // per-row byte offsets (rows past F are clamped to the last window: they recompute and re-store its values):
  // rowx = start of the window's step-0 input projection row in Xp, roww = w * hid * 4
  unsigned* rowx = reinterpret_cast<unsigned*>(X3 ? reinterpret_cast<float*>(Xlo + q.R * ldx) : enc_smem + q.Kp * ldk);
  unsigned* roww = rowx + q.R;
  unsigned* rowm = roww + q.R;   // w * hist * 4 (dropout mask row)
  for (int i = tid; i < q.R; i += ENC_NT) {
    const int w = min(wbase + i, a.F - 1);
    const int n = w / a.B, b = w - n * a.B;
    rowx[i] = (unsigned)(b * a.T + pos0 + n) * (unsigned)(G3 * 4);
    roww[i] = (unsigned)w * (unsigned)(hid * 4);
    rowm[i] = (unsigned)w * (unsigned)(a.hist * 4);
  }

  float bi[2][3], bh[2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int j = jb + 32 * t;
      const bool ok = j < hid;
      bi[t][g] = ok ? a.b_ih[g * hid + j] : 0.0f;
      bh[t][g] = ok ? a.b_hh[g * hid + j] : 0.0f;
    }
  const int boff = half * 3 * Jp + jb;         // per-lane offset into the padded weights (k = 2 kp + half)
  const int aoff = half * ldk + rg * 32 + l31;  // per-lane offset into Als
  const int nkp = q.Kp >> 1;                    // multiple of 2 * ENC_KC
  __syncthreads();

  for (int s = 0; s < a.hist; ++s) {
    ENC_STAMP(0);
    f32x16 acc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][g][r] = 0.0f;
    if (X3 && s > 0) {
      // per 16-deep k-tile: A hi/lo fragments of this wave's 32 rows, B hi/lo fragments of its 2 x 3 column tiles
      const int nkt = q.Kp >> 4, nct = Jp >> 5;
      const __bf16* xh = Xhi + (rg * 32 + l31) * ldx + 8 * half;
      const __bf16* xl = Xlo + (rg * 32 + l31) * ldx + 8 * half;
      ebf16x8 ah0, al0, ah1, al1;
      EncFrag f0[2][3][2], f1[2][3][2];  // [t][g][plane]
      auto load = [&](int kt, ebf16x8& ah, ebf16x8& al, EncFrag (&f)[2][3][2]) {
        ah = *reinterpret_cast<const ebf16x8*>(xh + kt * 16);
        al = *reinterpret_cast<const ebf16x8*>(xl + kt * 16);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint4* __restrict__ wf = q.wfrag + ((long)((kt * 3 + g) * nct + cg * 2 + t) * 2) * 64;  // uniform
            f[t][g][0].u = wf[(unsigned)lane];
            f[t][g][1].u = (wf + 64)[(unsigned)lane];
          }
      };
      auto mma = [&](const ebf16x8& ah, const ebf16x8& al, const EncFrag (&f)[2][3][2]) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[t][g] = ENC_MFMA(al, f[t][g][0].v, acc[t][g], 0, 0, 0);
            acc[t][g] = ENC_MFMA(ah, f[t][g][1].v, acc[t][g], 0, 0, 0);
            acc[t][g] = ENC_MFMA(ah, f[t][g][0].v, acc[t][g], 0, 0, 0);
          }
      };
      // every load in the steady-state loop is unconditional: with a conditional load the compiler cannot count the loads
      // in flight and waits for ALL of them (vmcnt(0)) before the MFMAs, i.e. for the fragments it has just requested
      load(0, ah0, al0, f0);
      int kt = 0;
      for (; kt + 2 < nkt; kt += 2) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        load(kt + 2, ah0, al0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah1, al1, f1);
      }
      if (kt + 1 < nkt) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        mma(ah1, al1, f1);
      } else {
        mma(ah0, al0, f0);
      }
    }
    if (!X3 && s > 0) {
      float a0[ENC_KC], a1[ENC_KC], b0[ENC_KC][6], b1[ENC_KC][6];
      auto load = [&](int kp0, float (&av)[ENC_KC], float (&bv)[ENC_KC][6]) {
        const float* __restrict__ wk = q.wpad + (long)kp0 * 6 * Jp;  // uniform base, 32-bit per-lane offset: saddr loads
        const float* ak = Als + kp0 * 2 * ldk;
#pragma unroll
        for (int u = 0; u < ENC_KC; ++u) {
          av[u] = ak[u * 2 * ldk + aoff];
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const float* __restrict__ wu = wk + (u * 6 + g) * Jp;  // uniform
            bv[u][g] = wu[(unsigned)boff];
            bv[u][3 + g] = (wu + 32)[(unsigned)boff];
          }
        }
      };
      auto mma = [&](const float (&av)[ENC_KC], const float (&bv)[ENC_KC][6]) {
#pragma unroll
        for (int u = 0; u < ENC_KC; ++u)
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            acc[0][g] = mfma32(av[u], bv[u][g], acc[0][g]);
            acc[1][g] = mfma32(av[u], bv[u][3 + g], acc[1][g]);
          }
      };
      load(0, a0, b0);
      int kp = 0;
      for (; kp + 2 * ENC_KC < nkp; kp += 2 * ENC_KC) {   // unconditional loads (see the bf16x3 loop)
        load(kp + ENC_KC, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        load(kp + 2 * ENC_KC, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
      }
      load(kp + ENC_KC, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      mma(a1, b1);
    }
    ENC_STAMP(1);
    __syncthreads();  // every wave has finished reading h_{s-1}
    ENC_STAMP(2);
    // gate math on the accumulators in four straight-line groups (one hidden sub-tile x 8 registers each: 40 loads in
    // flight per lane). The lane coordinates are laundered through an empty asm so that the per-register address arithmetic
    // stays inside the step loop instead of being hoisted into ~100 loop-invariant registers (which spilled).
    int halfv = half, jv = jb;
    asm volatile("" : "+v"(halfv), "+v"(jv));
    const enc_rsrc bx = enc_buf(a.Xp, (long)a.B * a.T * G3 * 4);
    const enc_rsrc bhs = enc_buf(STASH ? a.hseq + (long)s * a.F * hid : nullptr, STASH ? (long)a.F * hid * 4 : 0);
    const enc_rsrc bgs = enc_buf(STASH ? a.gates + (long)s * a.F * 4 * hid : nullptr, STASH ? (long)a.F * hid * 16 : 0);
    const enc_rsrc bms = enc_buf(MASK ? a.mask : nullptr, MASK ? (long)a.F * a.hist * 4 : 0);
    const unsigned sx = (unsigned)s * (unsigned)(G3 * 4), h4 = (unsigned)hid * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = jv + 32 * t;
      if (j < hid) {
        const unsigned j4 = (unsigned)j * 4u;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
          float xr[8], xu[8], xn[8], mk[8], hp[8];
          unsigned wo[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int rl = enc_rowl(rg, rh * 8 + e, halfv);
            const unsigned xo = rowx[rl] + j4;
            wo[e] = roww[rl];
            xr[e] = enc_ld(bx, xo, sx); xu[e] = enc_ld(bx, xo, sx + h4); xn[e] = enc_ld(bx, xo, sx + 2 * h4);
            mk[e] = MASK ? enc_ld(bms, rowm[rl], (unsigned)s * 4u) : 1.0f;
            hp[e] = Als[j * ldk + rl];  // zero at s = 0
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = rh * 8 + e;
            const int rl = enc_rowl(rg, r, halfv);
            const float ghn = acc[t][2][r] + bh[t][2];
            const float rr = sigmoidf_(mk[e] * xr[e] + bi[t][0] + (acc[t][0][r] + bh[t][0]));
            const float uu = sigmoidf_(mk[e] * xu[e] + bi[t][1] + (acc[t][1][r] + bh[t][1]));
            const float nn = tanhf_(mk[e] * xn[e] + bi[t][2] + rr * ghn);
            const float hnew = (1.0f - uu) * nn + uu * hp[e];
            if (STASH) {
              enc_st(hnew, bhs, wo[e] + j4, 0);
              const unsigned go = 4u * wo[e] + j4;
              enc_st(rr, bgs, go, 0); enc_st(uu, bgs, go, h4); enc_st(nn, bgs, go, 2 * h4); enc_st(ghn, bgs, go, 3 * h4);
            }
            Als[j * ldk + rl] = hnew;
            if (X3) {
              const __bf16 hi = (__bf16)hnew;
              Xhi[rl * ldx + j] = hi;
              Xlo[rl * ldx + j] = (__bf16)(hnew - (float)hi);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          ENC_STAMP(3 + t * 2 + rh);
        }
      }
    }
    __syncthreads();
    ENC_STAMP(7);
  }
  // cat(seq[:, -1], h_n[0]): the final state, once or twice (glow/models.py:63-64), straight from LDS
  for (int idx = tid; idx < q.R * hid; idx += ENC_NT) {
    const int rl = idx / hid, j = idx - rl * hid;
    const int w = wbase + rl;
    if (w < a.F) {
      const float v = Als[j * ldk + rl];
      float* c = a.cond + (long)w * a.ldcond + a.col + j;
      c[0] = v;
      if (a.dup) c[hid] = v;
    }
  }
}

// ---- the same forward recurrence (bf16x3 products) with a ROW-LAYOUT gate epilogue.
// What paced enc_gru_fwd_fused_kernel<.., true> was not HBM, L2 or the matrix pipe but the number of vector-memory
// INSTRUCTIONS: in the accumulator layout a lane holds one column of 16 different rows, so every projected input, mask,
// stash value is its own 4-byte access - 9 wave instructions per (row, 32 columns) and step, 37 k cycles of address
// processing per CU and step next to 25 k for the weight fragments, against 18 k of MFMA issue (round-2 notes in DESIGN.md).
// Here each gate's accumulators go through a wave-private LDS tile (32 rows x 64 columns, written in accumulator order,
// read back row-wise) so that a lane owns 4 CONSECUTIVE columns of 8 rows: projected inputs come in and h, r, z, n, W_hn h
// leave as 16-byte accesses (4.5x fewer instructions), the state images are refreshed with 8-byte LDS stores, and h_{s-1}
// waits in that same tile during the MFMA phase (no k-major fp32 state image: 34 KB of LDS less, two workgroups per CU stay).
typedef __attribute__((ext_vector_type(4))) unsigned enc_u32x4;
__device__ __forceinline__ f32x4 enc_ld4(enc_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 enc_ld4s(enc_rsrc r, unsigned voff, unsigned soff) {   // read-once stash values: streamed
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2));
}
__device__ __forceinline__ void enc_st4(f32x4 v, enc_rsrc r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(enc_u32x4, v), r, voff, soff, LFI_ENC_ST_AUX);
}
typedef __attribute__((ext_vector_type(2))) unsigned enc_u32x2;
// four fp32 values rounded to bf16 (round to nearest even, as the GEMMs' operand split rounds the hi part), one 8-byte store
__device__ __forceinline__ void enc_st2h(f32x4 v, enc_rsrc r, unsigned voff, unsigned soff) {
  uint2 h, l;
  split2(v[0], v[1], &h.x, &l.x); split2(v[2], v[3], &h.y, &l.y);
  __builtin_amdgcn_raw_buffer_store_b64((enc_u32x2){h.x, h.y}, r, voff, soff, LFI_ENC_ST_AUX);
}
// fp16 gate stash (lfi_enc_desc.stash_f16): the four values BPTT needs of a hidden unit - r, z, n, W_hn h + b_hn - as four fp16
// side by side (8 bytes per unit instead of 16: r, z, n lie in (-1, 1) and the last one within a few units of 0, so fp16's 11
// bits round them by <= 2^-12 absolute, unbiased - finer than the bf16 rounding the two-product backward products apply to the
// gate DERIVATIVES computed from them). A lane's four consecutive units are 32 contiguous bytes: two 16-byte accesses either way.
typedef _Float16 enc_h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 enc_pack_gates(float r, float z, float n, float g) {
  const enc_h16x4 h = __builtin_convertvector((f32x4){r, z, n, g}, enc_h16x4);   // round to nearest even
  return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ f32x4 enc_unpack_gates(unsigned lo, unsigned hi) {
  const enc_h16x4 h = __builtin_bit_cast(enc_h16x4, (uint2){lo, hi});
  return __builtin_convertvector(h, f32x4);
}
constexpr int ENC_TP = 68;   // floats per row of the transpose tile: 272 B (16-byte aligned rows, 4-bank skew per row)

// Ingredient-removal builds (timing only, results are garbage; tools/enc_probe.py): -DLFI_ENC_EXP_FIXED_W = every k-tile reads the
// weight fragments of k-tile 0 (the L2 -> CU weight stream becomes an L1 hit), -DLFI_ENC_EXP_NO_XP = no projected-input loads
// -DLFI_ENC_EXP_NO_KLOOP = no h W_hh^T product (epilogue + barriers only), -DLFI_ENC_EXP_NO_GATEMATH = sigmoid / tanh replaced by one
// multiply (the epilogue's transcendental VALU work removed, its memory and LDS traffic kept)
#ifdef LFI_ENC_EXP_NO_GATEMATH
#define ENC_SIG(x) (0.25f * (x))
#define ENC_TANH(x) (0.25f * (x))
#else
#define ENC_SIG(x) sigmoidf_(x)
#define ENC_TANH(x) tanhf_(x)
#endif
#ifdef LFI_ENC_EXP_FIXED_W
#define ENC_WKT(kt) 0
#else
#define ENC_WKT(kt) (kt)
#endif
template <bool STASH, bool MASK, bool S16>
__global__ __launch_bounds__(ENC_NT, 2) void enc_gru_fwd_wide_kernel(EncArgs a, EncFused q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cg = wave % q.ncg, rg = wave / q.ncg;
  const int hid = a.hid, G3 = 3 * hid, Jp = q.Jp;
  const int wbase = blockIdx.x * q.R;
  const int pos0 = a.start - a.hist + 1;
  const int ldx = q.Kp + 8;
  // LDS: bf16 hi / lo images of h_{s-1} (row-major [R][Kp + 8]) | per-wave transpose tiles | biases [6][Jp] | per-row tables
  __bf16* Xhi = reinterpret_cast<__bf16*>(enc_smem);
  __bf16* Xlo = Xhi + q.R * ldx;
  float* T = reinterpret_cast<float*>(Xlo + q.R * ldx) + wave * (32 * ENC_TP);
  float* bias = reinterpret_cast<float*>(Xlo + q.R * ldx) + ENC_NW * (32 * ENC_TP);   // [0..2][Jp] = b_ih, [3..5][Jp] = b_hh
  unsigned* rowx = reinterpret_cast<unsigned*>(bias + 6 * Jp);
  unsigned* roww = rowx + q.R;
  float* mk_tab = reinterpret_cast<float*>(roww + q.R);                                 // [R][hist] (MASK only)
  for (int i = tid; i < q.R * ldx; i += ENC_NT) { Xhi[i] = (__bf16)0.0f; Xlo[i] = (__bf16)0.0f; }
  for (int i = tid; i < 6 * Jp; i += ENC_NT) {
    const int g = i / Jp, j = i - g * Jp;
    bias[i] = j < hid ? (g < 3 ? a.b_ih[g * hid + j] : a.b_hh[(g - 3) * hid + j]) : 0.0f;
  }
  for (int i = tid; i < q.R; i += ENC_NT) {
    const int w = min(wbase + i, a.F - 1);   // rows past F recompute and re-store the last window
    const int n = w / a.B, b = w - n * a.B;
    rowx[i] = (unsigned)(b * a.T + pos0 + n) * (unsigned)(G3 * 4);
    roww[i] = (unsigned)w * (unsigned)(hid * 4);
  }
  if (MASK)
    for (int i = tid; i < q.R * a.hist; i += ENC_NT) {
      const int rl = i / a.hist, s = i - rl * a.hist;
      mk_tab[i] = a.mask[(long)min(wbase + rl, a.F - 1) * a.hist + s];
    }
  // row layout of this wave's 32 x 64 tile: lane owns columns c4 .. c4 + 3 of rows 4 i + rsub, i = 0 .. 7
  const int rsub = lane >> 4, c4 = (lane & 15) * 4;
  const int j0 = cg * 64 + c4;            // first hidden index of the lane's four
  const bool jok = j0 < hid;              // hid % 4 == 0: the four are all inside or all outside
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(T + (4 * i + rsub) * ENC_TP + c4) = z;   // h_{-1} = 0
  }
  const enc_rsrc bx = enc_buf(a.Xp, (long)a.B * a.T * G3 * 4);
  const unsigned h4 = (unsigned)hid * 4u;
  __syncthreads();

  for (int s = 0; s < a.hist; ++s) {
    ENC_STAMP(0);
    f32x16 acc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][g][r] = 0.0f;
#ifdef LFI_ENC_EXP_NO_KLOOP
    if (false) {
#else
    if (s > 0) {
#endif
      const int nkt = q.Kp >> 4, nct = Jp >> 5;
      const __bf16* xh = Xhi + (rg * 32 + l31) * ldx + 8 * half;
      const __bf16* xl = Xlo + (rg * 32 + l31) * ldx + 8 * half;
      ebf16x8 ah0, al0, ah1, al1;
      EncFrag f0[2][3][2], f1[2][3][2];  // [t][g][plane]
      auto load = [&](int kt, ebf16x8& ah, ebf16x8& al, EncFrag (&f)[2][3][2]) {
        ah = *reinterpret_cast<const ebf16x8*>(xh + kt * 16);
        al = *reinterpret_cast<const ebf16x8*>(xl + kt * 16);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint4* __restrict__ wf = q.wfrag + ((long)((ENC_WKT(kt) * 3 + g) * nct + cg * 2 + t) * 2) * 64;  // uniform
#ifdef LFI_ENC_EXP_NO_WLOADS
            if (kt > 1) continue;
#endif
            f[t][g][0].u = wf[(unsigned)lane];
            f[t][g][1].u = (wf + 64)[(unsigned)lane];
          }
      };
      auto mma = [&](const ebf16x8& ah, const ebf16x8& al, const EncFrag (&f)[2][3][2]) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[t][g] = ENC_MFMA(al, f[t][g][0].v, acc[t][g], 0, 0, 0);
            acc[t][g] = ENC_MFMA(ah, f[t][g][1].v, acc[t][g], 0, 0, 0);
            acc[t][g] = ENC_MFMA(ah, f[t][g][0].v, acc[t][g], 0, 0, 0);
          }
      };
      load(0, ah0, al0, f0);   // every load in the steady-state loop is unconditional (see enc_gru_fwd_fused_kernel)
      int kt = 0;
      for (; kt + 2 < nkt; kt += 2) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        load(kt + 2, ah0, al0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah1, al1, f1);
      }
      if (kt + 1 < nkt) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        mma(ah1, al1, f1);
      } else {
        mma(ah0, al0, f0);
      }
    }
    ENC_STAMP(1);
    // ---- gate epilogue in the row layout
    int rsv = rsub, cv = c4, jv = j0;
    asm volatile("" : "+v"(rsv), "+v"(cv), "+v"(jv));   // keep the per-row address arithmetic inside the step loop (registers)
    const unsigned sx = (unsigned)s * (unsigned)(G3 * 4), j4 = (unsigned)jv * 4u;
    const unsigned oob = jv < hid ? 0u : 0x80000000u;
    const enc_rsrc bhs = enc_buf(STASH ? a.hseq + (long)s * a.F * hid : nullptr, STASH ? (long)a.F * hid * 4 : 0);
    const enc_rsrc bgs = S16 ? enc_buf(STASH ? reinterpret_cast<const _Float16*>(a.gates) + (long)s * a.F * 4 * hid : nullptr,
                                       STASH ? (long)a.F * hid * 8 : 0)
                             : enc_buf(STASH ? a.gates + (long)s * a.F * 4 * hid : nullptr, STASH ? (long)a.F * hid * 16 : 0);
    float* Trow = T + rsv * ENC_TP + cv;                 // + 4 i * ENC_TP per row
    float* Tacc = T + (4 * half) * ENC_TP + l31;         // accumulator (t, r) at + ((r & 3) + 8 (r >> 2)) * ENC_TP + 32 t
    f32x4 hp[8], xin[8];
    unsigned xo[8], wo[8];
    float mk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = rg * 32 + 4 * i + rsv;
      xo[i] = rowx[rl] + j4 + oob;   // columns past hid: an offset outside every buffer (loads return 0, stores are dropped)
      wo[i] = roww[rl];
      mk[i] = MASK ? mk_tab[rl * a.hist + s] : 1.0f;
      hp[i] = *reinterpret_cast<const f32x4*>(Trow + 4 * i * ENC_TP);   // h_{s-1}, parked here by the previous step
#ifdef LFI_ENC_EXP_NO_XP
      xin[i] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
#else
      xin[i] = enc_ld4(bx, xo[i], sx);
#endif
    }
    auto transpose = [&](int g, f32x4 (&out)[8]) {   // gate g of this wave's tile: accumulator layout -> row layout
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier reads of the tile are done
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Tacc[((r & 3) + 8 * (r >> 2)) * ENC_TP + 32 * t] = acc[t][g][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) out[i] = *reinterpret_cast<const f32x4*>(Trow + 4 * i * ENC_TP);
    };
    const f32x4 bir = *reinterpret_cast<const f32x4*>(bias + 0 * Jp + jv), bhr = *reinterpret_cast<const f32x4*>(bias + 3 * Jp + jv);
    const f32x4 biu = *reinterpret_cast<const f32x4*>(bias + 1 * Jp + jv), bhu = *reinterpret_cast<const f32x4*>(bias + 4 * Jp + jv);
    const f32x4 bin = *reinterpret_cast<const f32x4*>(bias + 2 * Jp + jv), bhn = *reinterpret_cast<const f32x4*>(bias + 5 * Jp + jv);
    f32x4 rr[8], uu[8], gh[8];
    transpose(0, gh);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) rr[i][e] = ENC_SIG(mk[i] * xin[i][e] + bir[e] + (gh[i][e] + bhr[e]));
      if (STASH && !S16) enc_st4(rr[i], bgs, 4u * wo[i] + j4 + oob, 0);
#ifndef LFI_ENC_EXP_NO_XP
      xin[i] = enc_ld4(bx, xo[i], sx + h4);
#endif
    }
    transpose(1, gh);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) uu[i][e] = ENC_SIG(mk[i] * xin[i][e] + biu[e] + (gh[i][e] + bhu[e]));
      if (STASH && !S16) enc_st4(uu[i], bgs, 4u * wo[i] + j4 + oob, h4);
#ifndef LFI_ENC_EXP_NO_XP
      xin[i] = enc_ld4(bx, xo[i], sx + 2 * h4);
#endif
    }
    transpose(2, gh);
    ENC_STAMP(2);
    __syncthreads();   // every wave has finished the MFMA phase: the state images may be overwritten
    ENC_STAMP(3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = rg * 32 + 4 * i + rsv;
      f32x4 ghn, nn, hn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ghn[e] = gh[i][e] + bhn[e];
        nn[e] = ENC_TANH(mk[i] * xin[i][e] + bin[e] + rr[i][e] * ghn[e]);
        hn[e] = (1.0f - uu[i][e]) * nn[e] + uu[i][e] * hp[i][e];
      }
      if (STASH && S16) {   // [window][unit][r, z, n, W_hn h] fp16: this lane's four units = 32 contiguous bytes
        const uint2 g0 = enc_pack_gates(rr[i][0], uu[i][0], nn[0], ghn[0]), g1 = enc_pack_gates(rr[i][1], uu[i][1], nn[1], ghn[1]);
        const uint2 g2 = enc_pack_gates(rr[i][2], uu[i][2], nn[2], ghn[2]), g3 = enc_pack_gates(rr[i][3], uu[i][3], nn[3], ghn[3]);
        const unsigned go = 2u * (wo[i] + j4) + oob;
        __builtin_amdgcn_raw_buffer_store_b128((enc_u32x4){g0.x, g0.y, g1.x, g1.y}, bgs, go, 0, LFI_ENC_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128((enc_u32x4){g2.x, g2.y, g3.x, g3.y}, bgs, go, 16, LFI_ENC_ST_AUX);
        enc_st4(hn, bhs, wo[i] + j4 + oob, 0);
        __builtin_amdgcn_sched_barrier(0);   // one row's packing temporaries at a time (register pressure)
      } else if (STASH) {
        enc_st4(nn, bgs, 4u * wo[i] + j4 + oob, 2 * h4);
        enc_st4(ghn, bgs, 4u * wo[i] + j4 + oob, 3 * h4);
        enc_st4(hn, bhs, wo[i] + j4 + oob, 0);
      }
      if (jok) {
        uint2 h, l;
        split2(hn[0], hn[1], &h.x, &l.x);
        split2(hn[2], hn[3], &h.y, &l.y);
        *reinterpret_cast<uint2*>(Xhi + rl * ldx + jv) = h;
        *reinterpret_cast<uint2*>(Xlo + rl * ldx + jv) = l;
      }
      hp[i] = hn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile's last row-wise reads are done
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Trow + 4 * i * ENC_TP) = hp[i];   // park h_s for the next step
    if (s == a.hist - 1 && jok) {   // cat(seq[:, -1], h_n[0]): the final state, once or twice (glow/models.py:63-64)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int w = wbase + rg * 32 + 4 * i + rsv;
        if (w < a.F) {
          float* c = a.cond + (long)w * a.ldcond + a.col + jv;
          *reinterpret_cast<f32x4*>(c) = hp[i];
          if (a.dup) *reinterpret_cast<f32x4*>(c + hid) = hp[i];
        }
      }
    }
    ENC_STAMP(4);
    __syncthreads();   // the new state images are complete
    ENC_STAMP(5);
  }
}

// ---- the forward recurrence with TWO 32-row tiles per wave (64 windows per row group) and one workgroup per CU.
// What paces enc_gru_fwd_wide_kernel, from ingredient-removal builds (tools/enc_probe.py + tools/build_variant.sh, p2_face shape, ms per
// launch): 0.66 as shipped; 0.62 without any stash; 0.65 with every k-tile reading the SAME weight fragments (L1 hits instead of the
// L2 stream); 0.64 with no weight loads in the k loop at all; 0.65 with sigmoid / tanh replaced by a multiply; 0.17 without the
// h W_hh^T loop. The k loop costs ~0.46 ms wherever its operands come from: 389 G MFMA-pass-FLOP at ~0.9 PFLOP/s, the rate every
// three-product bf16 stream of this code base settles at on this chip (the planes GEMM: 1.0), and two workgroups per CU meet in the
// matrix phase and in the gate epilogue alike (staggering the second one by 6 k .. 30 k cycles, by any pairing rule, changed nothing).
// This tiling keeps the MFMA work and halves everything around it in the k loop: a wave holds the accumulators of 64 windows (2 row
// tiles x 2 column tiles x 3 gates = 192 of the 512 registers a lone wave per SIMD may use), so a weight fragment and its address
// arithmetic feed six MFMAs instead of three. Measured: forward 0.66 -> 0.61 ms, whole step -0.06 .. -0.08 ms on three boxes. (An
// eight-wave variant - 64 windows x 32 hidden units per wave, two waves per SIMD - measured 0.58 - 0.67 ms by stash variant and
// +0.05 ms on the whole step: removed.) The gate epilogue is the row-layout one of the wide kernel, run once per row tile.
// (Round 5: the second row tile's first projected inputs requested one gate early - 32 more live registers in a kernel that already
// spills 10: 19 spilled, fp16-stash launch 0.544 -> 0.563 ms. Removed.)
// M16 (round 4, Kp a multiple of 256): the same products on v_mfma_f32_16x16x32_bf16 - same FLOP per cycle, but the chip holds a
// higher clock on that shape in three-product streams (timing-only build with the MFMAs swapped: forward 0.614 -> 0.554 ms on the
// p2_face shape; the two-product BPTT kernels gain nothing and keep the 32 x 32 shape). A wave's 64 x 64 x 3 gates are 4 x 4 x 3
// accumulator tiles; per 32-deep MFMA step m the state fragments (4 row tiles x hi / lo, one ds_read_b128 each) are read once and
// serve the three gates, the weight fragments of gate g (4 column tiles x hi / lo, 1 KB each from L2) are reloaded for step m + 1
// as soon as step m's MFMAs on them have issued: three stages of 48 MFMAs in flight between a load and its use.
template <bool STASH, bool MASK, bool S16, bool M16 = false>
__global__ __launch_bounds__(ENC_NT, 1) void enc_gru_fwd_r64_kernel(EncArgs a, EncFused q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cg = wave % q.ncg, rg = wave / q.ncg;
  const int hid = a.hid, G3 = 3 * hid, Jp = q.Jp;
  const int R2 = 2 * q.R;                        // windows per workgroup
  const int wbase = blockIdx.x * R2;
  const int pos0 = a.start - a.hist + 1;
  const int ldx = q.Kp + 8;
  // LDS: bf16 hi / lo images of h_{s-1} (row-major [R2][Kp + 8]) | per wave two transpose tiles | biases [6][Jp] | per-row tables
  __bf16* Xhi = reinterpret_cast<__bf16*>(enc_smem);
  __bf16* Xlo = Xhi + R2 * ldx;
  float* Tw = reinterpret_cast<float*>(Xlo + R2 * ldx) + wave * (2 * 32 * ENC_TP);
  float* bias = reinterpret_cast<float*>(Xlo + R2 * ldx) + ENC_NW * (2 * 32 * ENC_TP);   // [0..2][Jp] = b_ih, [3..5][Jp] = b_hh
  unsigned* rowx = reinterpret_cast<unsigned*>(bias + 6 * Jp);
  unsigned* roww = rowx + R2;
  float* mk_tab = reinterpret_cast<float*>(roww + R2);                                   // [R2][hist] (MASK only)
  for (int i = tid; i < R2 * ldx; i += ENC_NT) { Xhi[i] = (__bf16)0.0f; Xlo[i] = (__bf16)0.0f; }
  for (int i = tid; i < 6 * Jp; i += ENC_NT) {
    const int g = i / Jp, j = i - g * Jp;
    bias[i] = j < hid ? (g < 3 ? a.b_ih[g * hid + j] : a.b_hh[(g - 3) * hid + j]) : 0.0f;
  }
  for (int i = tid; i < R2; i += ENC_NT) {
    const int w = min(wbase + i, a.F - 1);   // rows past F recompute and re-store the last window
    const int n = w / a.B, b = w - n * a.B;
    rowx[i] = (unsigned)(b * a.T + pos0 + n) * (unsigned)(G3 * 4);
    roww[i] = (unsigned)w * (unsigned)(hid * 4);
  }
  if (MASK)
    for (int i = tid; i < R2 * a.hist; i += ENC_NT) {
      const int rl = i / a.hist, s = i - rl * a.hist;
      mk_tab[i] = a.mask[(long)min(wbase + rl, a.F - 1) * a.hist + s];
    }
  // row layout of a 32 x 64 tile: lane owns columns c4 .. c4 + 3 of rows 4 i + rsub, i = 0 .. 7
  const int rsub = lane >> 4, c4 = (lane & 15) * 4;
  const int j0 = cg * 64 + c4;
  const bool jok = j0 < hid;
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Tw + rt * (32 * ENC_TP) + (4 * i + rsub) * ENC_TP + c4) = z;   // h_{-1} = 0
  }
  const enc_rsrc bx = enc_buf(a.Xp, (long)a.B * a.T * G3 * 4);
  const unsigned h4 = (unsigned)hid * 4u;
  __syncthreads();

  for (int s = 0; s < a.hist; ++s) {
    constexpr int A32 = M16 ? 1 : 2, A16 = M16 ? 4 : 1;
    f32x16 acc[A32][A32][3];   // [row tile][column tile][gate]
    f32x4 acc16[A16][A16][3];  // M16: [16-row tile][16-column tile][gate]
#pragma unroll
    for (int rt = 0; rt < A32; ++rt)
#pragma unroll
      for (int t = 0; t < A32; ++t)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][t][g][r] = 0.0f;
#pragma unroll
    for (int mi = 0; mi < A16; ++mi)
#pragma unroll
      for (int ni = 0; ni < A16; ++ni)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc16[mi][ni][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (M16) if (s > 0) {
      const int nct = Jp >> 4, nchunk = q.Kp >> 3;
      const int gq = lane >> 4;
      const __bf16* xh = Xhi + (rg * 64 + (lane & 15)) * ldx + 8 * ((gq >> 1) + (nchunk >> 1) * (gq & 1));
      const __bf16* xl = xh + R2 * ldx;
      ebf16x8 ahA[4], alA[4], ahB[4], alB[4];
      EncFrag fb[3][4][2];   // [gate][column tile][plane]
      auto loadA = [&](int m, ebf16x8 (&ah)[4], ebf16x8 (&al)[4]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          ah[mi] = *reinterpret_cast<const ebf16x8*>(xh + mi * 16 * ldx + m * 16);
          al[mi] = *reinterpret_cast<const ebf16x8*>(xl + mi * 16 * ldx + m * 16);
        }
      };
      // (buffer loads: descriptor + per-lane 16 lane + a scalar fragment offset - 192 per-lane 64-bit addresses would not fit)
      const enc_rsrc bw = enc_buf(q.wfrag, 12L * q.Kp * Jp);
      const unsigned lane16 = (unsigned)lane * 16u;
      auto wload = [&](int m, int g, int ni, int plane) {
        const unsigned so = (unsigned)((((m * 3 + g) * nct + cg * 4 + ni) * 2 + plane) * 1024);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(bw, lane16, so, 0));
      };
      // one stage: gate G of MFMA step m; its weight fragments are reloaded for step mn column tile by column tile
      auto stage = [&](auto G, const ebf16x8 (&ah)[4], const ebf16x8 (&al)[4], int mn) {
        constexpr int g = decltype(G)::value;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc16[mi][ni][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], fb[g][ni][0].v, acc16[mi][ni][g], 0, 0, 0);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc16[mi][ni][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], fb[g][ni][1].v, acc16[mi][ni][g], 0, 0, 0);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc16[mi][ni][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], fb[g][ni][0].v, acc16[mi][ni][g], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          fb[g][ni][0].u = wload(mn, g, ni, 0);
          fb[g][ni][1].u = wload(mn, g, ni, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      using G0 = std::integral_constant<int, 0>; using G1 = std::integral_constant<int, 1>; using G2 = std::integral_constant<int, 2>;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          fb[g][ni][0].u = wload(0, g, ni, 0);
          fb[g][ni][1].u = wload(0, g, ni, 1);
        }
      loadA(0, ahA, alA);
#pragma unroll
      for (int m = 0; m < 8; m += 2) {   // (Kp = 256: eight 32-deep steps, straight-line; loads past the end re-read the last step)
        const int m1 = m + 1, m2 = min(m + 2, 7);
        stage(G0{}, ahA, alA, m1);
        loadA(m1, ahB, alB);
        __builtin_amdgcn_sched_barrier(0);
        stage(G1{}, ahA, alA, m1);
        stage(G2{}, ahA, alA, m1);
        stage(G0{}, ahB, alB, m2);
        loadA(m2, ahA, alA);
        __builtin_amdgcn_sched_barrier(0);
        stage(G1{}, ahB, alB, m2);
        stage(G2{}, ahB, alB, m2);
      }
    }
    if constexpr (!M16) if (s > 0) {
      const int nkt = q.Kp >> 4, nct = Jp >> 5;
      const __bf16* xh = Xhi + (rg * 64 + l31) * ldx + 8 * half;
      const __bf16* xl = Xlo + (rg * 64 + l31) * ldx + 8 * half;
      ebf16x8 ah0[2], al0[2], ah1[2], al1[2];
      EncFrag f0[2][3][2], f1[2][3][2];  // [t][g][plane]
      auto load = [&](int kt, ebf16x8 (&ah)[2], ebf16x8 (&al)[2], EncFrag (&f)[2][3][2]) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          ah[rt] = *reinterpret_cast<const ebf16x8*>(xh + rt * 32 * ldx + kt * 16);
          al[rt] = *reinterpret_cast<const ebf16x8*>(xl + rt * 32 * ldx + kt * 16);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint4* __restrict__ wf = q.wfrag + ((long)((ENC_WKT(kt) * 3 + g) * nct + cg * 2 + t) * 2) * 64;  // uniform
            f[t][g][0].u = wf[(unsigned)lane];
            f[t][g][1].u = (wf + 64)[(unsigned)lane];
          }
      };
      auto mma = [&](const ebf16x8 (&ah)[2], const ebf16x8 (&al)[2], const EncFrag (&f)[2][3][2]) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              acc[rt][t][g] = ENC_MFMA(al[rt], f[t][g][0].v, acc[rt][t][g], 0, 0, 0);
              acc[rt][t][g] = ENC_MFMA(ah[rt], f[t][g][1].v, acc[rt][t][g], 0, 0, 0);
              acc[rt][t][g] = ENC_MFMA(ah[rt], f[t][g][0].v, acc[rt][t][g], 0, 0, 0);
            }
      };
      load(0, ah0, al0, f0);   // every load in the steady-state loop is unconditional (see enc_gru_fwd_fused_kernel)
      int kt = 0;
      for (; kt + 2 < nkt; kt += 2) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        load(kt + 2, ah0, al0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah1, al1, f1);
      }
      if (kt + 1 < nkt) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        mma(ah1, al1, f1);
      } else {
        mma(ah0, al0, f0);
      }
    }
    __syncthreads();   // every wave has finished the MFMA phase: the state images may be overwritten
    // ---- gate epilogue in the row layout, one row tile after the other
    const enc_rsrc bhs = enc_buf(STASH ? a.hseq + (long)s * a.F * hid : nullptr, STASH ? (long)a.F * hid * 4 : 0);
    const enc_rsrc bgs = S16 ? enc_buf(STASH ? reinterpret_cast<const _Float16*>(a.gates) + (long)s * a.F * 4 * hid : nullptr,
                                       STASH ? (long)a.F * hid * 8 : 0)
                             : enc_buf(STASH ? a.gates + (long)s * a.F * 4 * hid : nullptr, STASH ? (long)a.F * hid * 16 : 0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      int rsv = rsub, cv = c4, jv = j0;
      asm volatile("" : "+v"(rsv), "+v"(cv), "+v"(jv));   // keep the per-row address arithmetic inside the step loop (registers)
      const unsigned sx = (unsigned)s * (unsigned)(G3 * 4), j4 = (unsigned)jv * 4u;
      const unsigned oob = jv < hid ? 0u : 0x80000000u;
      float* T = Tw + rt * (32 * ENC_TP);
      float* Trow = T + rsv * ENC_TP + cv;                 // + 4 i * ENC_TP per row
      float* Tacc = T + (4 * half) * ENC_TP + l31;         // accumulator (t, r) at + ((r & 3) + 8 (r >> 2)) * ENC_TP + 32 t
      float* Tacc16 = T + (4 * (lane >> 4)) * ENC_TP + (lane & 15);   // M16: tile (mi, ni) register r at + (16 mi + r) * ENC_TP + 16 ni
      const int rbase = rg * 64 + rt * 32;
      f32x4 hp[8], xin[8];
      unsigned xo[8], wo[8];
      float mk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = rbase + 4 * i + rsv;
        xo[i] = rowx[rl] + j4 + oob;   // columns past hid: an offset outside every buffer (loads return 0, stores are dropped)
        wo[i] = roww[rl];
        mk[i] = MASK ? mk_tab[rl * a.hist + s] : 1.0f;
        hp[i] = *reinterpret_cast<const f32x4*>(Trow + 4 * i * ENC_TP);   // h_{s-1}, parked here by the previous step
        xin[i] = enc_ld4(bx, xo[i], sx);
      }
      auto transpose = [&](int g, f32x4 (&out)[8]) {   // gate g of this tile: accumulator layout -> row layout
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier reads of the tile are done
        if constexpr (M16) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
              for (int r = 0; r < 4; ++r) Tacc16[(16 * mi + r) * ENC_TP + 16 * ni] = acc16[2 * rt + mi][ni][g][r];
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Tacc[((r & 3) + 8 * (r >> 2)) * ENC_TP + 32 * t] = acc[rt][t][g][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = *reinterpret_cast<const f32x4*>(Trow + 4 * i * ENC_TP);
      };
      const f32x4 bir = *reinterpret_cast<const f32x4*>(bias + 0 * Jp + jv), bhr = *reinterpret_cast<const f32x4*>(bias + 3 * Jp + jv);
      const f32x4 biu = *reinterpret_cast<const f32x4*>(bias + 1 * Jp + jv), bhu = *reinterpret_cast<const f32x4*>(bias + 4 * Jp + jv);
      const f32x4 bin = *reinterpret_cast<const f32x4*>(bias + 2 * Jp + jv), bhn = *reinterpret_cast<const f32x4*>(bias + 5 * Jp + jv);
      f32x4 rr[8], uu[8], gh[8];
      transpose(0, gh);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[i][e] = ENC_SIG(mk[i] * xin[i][e] + bir[e] + (gh[i][e] + bhr[e]));
        if (STASH && !S16) enc_st4(rr[i], bgs, 4u * wo[i] + j4 + oob, 0);
        xin[i] = enc_ld4(bx, xo[i], sx + h4);
      }
      transpose(1, gh);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) uu[i][e] = ENC_SIG(mk[i] * xin[i][e] + biu[e] + (gh[i][e] + bhu[e]));
        if (STASH && !S16) enc_st4(uu[i], bgs, 4u * wo[i] + j4 + oob, h4);
        xin[i] = enc_ld4(bx, xo[i], sx + 2 * h4);
      }
      transpose(2, gh);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = rbase + 4 * i + rsv;
        f32x4 ghn, nn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ghn[e] = gh[i][e] + bhn[e];
          nn[e] = ENC_TANH(mk[i] * xin[i][e] + bin[e] + rr[i][e] * ghn[e]);
          hn[e] = (1.0f - uu[i][e]) * nn[e] + uu[i][e] * hp[i][e];
        }
        if (STASH && S16) {   // [window][unit][r, z, n, W_hn h] fp16: this lane's four units = 32 contiguous bytes
          const uint2 g0 = enc_pack_gates(rr[i][0], uu[i][0], nn[0], ghn[0]), g1 = enc_pack_gates(rr[i][1], uu[i][1], nn[1], ghn[1]);
          const uint2 g2 = enc_pack_gates(rr[i][2], uu[i][2], nn[2], ghn[2]), g3 = enc_pack_gates(rr[i][3], uu[i][3], nn[3], ghn[3]);
          const unsigned go = 2u * (wo[i] + j4) + oob;
          __builtin_amdgcn_raw_buffer_store_b128((enc_u32x4){g0.x, g0.y, g1.x, g1.y}, bgs, go, 0, LFI_ENC_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128((enc_u32x4){g2.x, g2.y, g3.x, g3.y}, bgs, go, 16, LFI_ENC_ST_AUX);
          enc_st4(hn, bhs, wo[i] + j4 + oob, 0);
        } else if (STASH) {
          enc_st4(nn, bgs, 4u * wo[i] + j4 + oob, 2 * h4);
          enc_st4(ghn, bgs, 4u * wo[i] + j4 + oob, 3 * h4);
          enc_st4(hn, bhs, wo[i] + j4 + oob, 0);
        }
        if (jok) {
          uint2 h, l;
          split2(hn[0], hn[1], &h.x, &l.x);
          split2(hn[2], hn[3], &h.y, &l.y);
          *reinterpret_cast<uint2*>(Xhi + rl * ldx + jv) = h;
          *reinterpret_cast<uint2*>(Xlo + rl * ldx + jv) = l;
        }
        hp[i] = hn;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile's last row-wise reads are done
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Trow + 4 * i * ENC_TP) = hp[i];   // park h_s for the next step
      if (s == a.hist - 1 && jok) {   // cat(seq[:, -1], h_n[0]): the final state, once or twice (glow/models.py:63-64)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int w = wbase + rbase + 4 * i + rsv;
          if (w < a.F) {
            float* c = a.cond + (long)w * a.ldcond + a.col + jv;
            *reinterpret_cast<f32x4*>(c) = hp[i];
            if (a.dup) *reinterpret_cast<f32x4*>(c + hid) = hp[i];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one row tile's epilogue at a time (register pressure)
    }
    __syncthreads();   // the new state images are complete
  }
}

// ---- the forward recurrence with the gate epilogue UNDER the matrix phase (round 5; hid = 256, fragment order of
// enc_frag_weights16_kernel). enc_gru_fwd_r64_kernel runs a step as two serial parts on a lone wave per SIMD: 1152 MFMAs (18.4 k
// cycles of matrix pipe), then ~20 k cycles of transposes through LDS, gate math and stores while the pipe idles (PMC: MFMA busy
// 35 % of the wave's lifetime). Two changes let one wave fill the pipe's shadow with its own VALU work:
//   1. the product is taken TRANSPOSED (weights as the MFMA's A operand, state as B: the same products in the same order, bit for
//      bit): accumulator (mi, ni) register r is window 16 ni + (lane & 15), hidden unit 64 cg + 16 mi + 4 (lane >> 4) + r - four
//      consecutive units of one window per lane, which is the epilogue's own 16-byte layout: no transpose tiles (69 KB of LDS, six
//      write-wait-read round trips per step), and h_{s-1} stays in registers;
//   2. the wave's 64 hidden units are taken as four 16-unit tiles one after the other, each with its whole k sweep (phase p: 8
//      blocks of 36 MFMAs); tile p - 1's epilogue is independent of tile p's MFMAs and is issued between them (sched_group_barrier:
//      one MFMA, then up to LFI_T16_NV VALU). Only the last tile's epilogue is exposed. The state images are double-buffered (step parity)
//      because tile 0's new state is written while other waves still read the old one: one barrier per step.
// Weight fragments: a ring of four blocks, loaded three blocks (~1.7 k cycles) ahead and straight through the step boundary and the
// barrier; every fragment is fetched once per step and wave as before (L2 -> CU stream unchanged), the state fragments are re-read
// from LDS for each of the four tiles. Projected inputs are loaded one phase ahead of their use, three loads per two blocks.
// (timing-only experiment switches, garbage results: -DLFI_T16_NV=n VALU per MFMA in the interleave (0: the compiler's own order),
// -DLFI_T16_NO_XP no projected-input loads, -DLFI_T16_NO_W no weight loads in the loop, -DLFI_T16_NO_MFMA no products,
// -DLFI_T16_NO_EPI no gate math / stores under the products, -DLFI_T16_RING=n blocks of weight prefetch)
#ifndef LFI_T16_NV
#define LFI_T16_NV 2   // (an MFMA of this shape holds the vector issue for 8 of its 16 cycles: two 4-cycle fillers fit; measured level with 3: 0.44 - 0.46 ms)
#endif
#ifndef LFI_T16_RING
#define LFI_T16_RING 12   // weight ring: slots of one (block, gate) = two fragments; a divisor of 96; RING - 1 of them in flight
#endif
template <bool STASH, bool MASK, bool S16>
__global__ __launch_bounds__(ENC_NT, 1) void enc_gru_fwd_t16_kernel(EncArgs a, EncFused q) {
  constexpr int hid = 256, G3 = 768, ldx = 264, R2 = 64, nct = 16, IMG = R2 * ldx;
  constexpr unsigned h4 = hid * 4u;
  const int tid = threadIdx.x, lane = tid & 63, cg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wl = lane & 15, jq = lane >> 4;
  const int wbase = blockIdx.x * R2;
  const int pos0 = a.start - a.hist + 1;
  // LDS: state images [step parity][hi, lo][R2][ldx] bf16 | biases [6][256] | per-row tables | masks [R2][hist]
  __bf16* X = reinterpret_cast<__bf16*>(enc_smem);
  float* bias = reinterpret_cast<float*>(X + 4 * IMG);
  unsigned* rowx = reinterpret_cast<unsigned*>(bias + 6 * hid);
  unsigned* roww = rowx + R2;
  float* mk_tab = reinterpret_cast<float*>(roww + R2);
  {
    uint4* z = reinterpret_cast<uint4*>(X);
    for (int i = tid; i < 4 * IMG / 8; i += ENC_NT) z[i] = uint4{0u, 0u, 0u, 0u};   // h_{-1} = 0
  }
  for (int i = tid; i < 6 * hid; i += ENC_NT) bias[i] = i < 3 * hid ? a.b_ih[i] : a.b_hh[i - 3 * hid];
  for (int i = tid; i < R2; i += ENC_NT) {
    const int w = min(wbase + i, a.F - 1);   // rows past F recompute and re-store the last window
    const int n = w / a.B, b = w - n * a.B;
    rowx[i] = (unsigned)(b * a.T + pos0 + n) * (unsigned)(G3 * 4);
    roww[i] = (unsigned)w * h4;
  }
  if (MASK)
    for (int i = tid; i < R2 * a.hist; i += ENC_NT) {
      const int rl = i / a.hist, s = i - rl * a.hist;
      mk_tab[i] = a.mask[(long)min(wbase + rl, a.F - 1) * a.hist + s];
    }
  const enc_rsrc bx = enc_buf(a.Xp, (long)a.B * a.T * G3 * 4);
  const enc_rsrc bw = enc_buf(q.wfrag, 12L * hid * hid);
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned jb4 = (unsigned)(cg * 64 + 4 * jq) * 4u;   // byte offset of this lane's units in tile 0 of its wave (+ 64 per tile)
  // element offsets into an image: fragment reads (row wl of window tile ni, k chunk of lane group jq) and state writes
  const int sfr = wl * ldx + 8 * ((jq >> 1) + 16 * (jq & 1));
  const int swr = wl * ldx + cg * 64 + 4 * jq;
  // (the wave's column group goes into the per-lane offset, the plane into the instruction's immediate: what is left for the scalar
  // offset is a compile-time constant per load - 192 runtime scalars lived in VGPR lanes and cost a v_readlane each)
  const unsigned lanecg = lane16 + (unsigned)cg * 8192u;
  auto wload = [&](int m, int g, int mi, int plane) {
    const unsigned so = (unsigned)((((m * 3 + g) * nct + mi) * 2) * 1024);
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(bw, lanecg + (unsigned)plane * 1024u, so, 0));
  };
  constexpr int WR = LFI_T16_RING;
  EncFrag wr[WR][2];             // weight ring [slot = (block, gate) mod WR][plane (hi, lo)]
  ebf16x8 sh[2][4], sl[2][4];    // state fragments [buffer][window tile]
  f32x4 acc[2][4][3];            // [phase parity][window tile][gate]
  f32x4 hprev[4][4];             // h_{s-1} of this lane's units [unit tile][window tile]
  f32x4 xin[4][3];               // projected inputs of the tile whose MFMAs run now [window tile][gate]
  f32x4 rr, uu;                  // between the two halves of an epilogue chunk
  unsigned xo[4], wo[4];
  float mk[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) hprev[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < WR - 1; ++j) {
    wr[j][0].u = wload((j / 3) & 7, j % 3, (j / 3) >> 3, 0);
    wr[j][1].u = wload((j / 3) & 7, j % 3, (j / 3) >> 3, 1);
  }
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    xo[ni] = rowx[16 * ni + wl] + jb4;
    wo[ni] = roww[16 * ni + wl];
    mk[ni] = 1.0f;
  }

  for (int s = 0; s < a.hist; ++s) {
    const __bf16* Xr = X + ((s & 1) ^ 1) * (2 * IMG) + sfr;   // h_{s-1}
    __bf16* Xw = X + (s & 1) * (2 * IMG) + swr;               // h_s
    const unsigned sx = (unsigned)s * (unsigned)(G3 * 4);
    const enc_rsrc bhs = enc_buf(STASH ? a.hseq + (long)s * a.F * hid : nullptr, STASH ? (long)a.F * hid * 4 : 0);
    const enc_rsrc bgs = S16 ? enc_buf(STASH ? reinterpret_cast<const _Float16*>(a.gates) + (long)s * a.F * 4 * hid : nullptr,
                                       STASH ? (long)a.F * hid * 8 : 0)
                             : enc_buf(STASH ? a.gates + (long)s * a.F * 4 * hid : nullptr, STASH ? (long)a.F * hid * 16 : 0);
    if (MASK) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) mk[ni] = mk_tab[(16 * ni + wl) * a.hist + s];
    }
    auto sload = [&](int m, ebf16x8 (&h)[4], ebf16x8 (&l)[4]) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        h[ni] = *reinterpret_cast<const ebf16x8*>(Xr + ni * 16 * ldx + m * 16);
        l[ni] = *reinterpret_cast<const ebf16x8*>(Xr + IMG + ni * 16 * ldx + m * 16);
      }
    };
    sload(0, sh[0], sl[0]);
    f32x4 bir, bhr, biu, bhu, bin, bhn;
    auto load_biases = [&](int qt) {   // biases of unit tile qt
      const float* bb = bias + cg * 64 + 16 * qt + 4 * jq;
      bir = *reinterpret_cast<const f32x4*>(bb); bhr = *reinterpret_cast<const f32x4*>(bb + 3 * hid);
      biu = *reinterpret_cast<const f32x4*>(bb + hid); bhu = *reinterpret_cast<const f32x4*>(bb + 4 * hid);
      bin = *reinterpret_cast<const f32x4*>(bb + 2 * hid); bhn = *reinterpret_cast<const f32x4*>(bb + 5 * hid);
    };
    // projected inputs of (unit tile t, window tile ni): gates r, z in the even block of a pair, n in the odd one
    auto xload = [&](int t, int ni, int odd, f32x4 (&x)[3]) {
      const unsigned t64 = (unsigned)t * 64u;
#ifdef LFI_T16_NO_XP
      if (s == 0 && t == 0) x[0] = x[1] = x[2] = f32x4{0.1f, 0.2f, 0.3f, 0.4f} * (float)(t64 + lane);
#else
      if (!odd) {
        x[0] = enc_ld4(bx, xo[ni], sx + t64);
        x[1] = enc_ld4(bx, xo[ni], sx + h4 + t64);
      } else {
        x[2] = enc_ld4(bx, xo[ni], sx + 2 * h4 + t64);
      }
#endif
    };
    // one half of the epilogue chunk (unit tile qt, window tile ni): r, z (even) | n, h, stash, state images (odd)
    auto epi_half = [&](int qt, int ni, int odd, const f32x4 (&ac)[3], const f32x4 (&x)[3]) {
      const unsigned t64 = (unsigned)qt * 64u;   // byte offset of the tile's units in fp32 rows
      if (!odd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rr[e] = ENC_SIG(mk[ni] * x[0][e] + bir[e] + (ac[0][e] + bhr[e]));
          uu[e] = ENC_SIG(mk[ni] * x[1][e] + biu[e] + (ac[1][e] + bhu[e]));
        }
        if (STASH && !S16) {
          enc_st4(rr, bgs, 4u * wo[ni] + jb4, t64);
          enc_st4(uu, bgs, 4u * wo[ni] + jb4, h4 + t64);
        }
      } else {
        f32x4 ghn, nn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ghn[e] = ac[2][e] + bhn[e];
          nn[e] = ENC_TANH(mk[ni] * x[2][e] + bin[e] + rr[e] * ghn[e]);
          hn[e] = (1.0f - uu[e]) * nn[e] + uu[e] * hprev[qt][ni][e];
        }
        if (STASH && S16) {   // [window][unit][r, z, n, W_hn h] fp16: this lane's four units = 32 contiguous bytes
          const uint2 g0 = enc_pack_gates(rr[0], uu[0], nn[0], ghn[0]), g1 = enc_pack_gates(rr[1], uu[1], nn[1], ghn[1]);
          const uint2 g2 = enc_pack_gates(rr[2], uu[2], nn[2], ghn[2]), g3 = enc_pack_gates(rr[3], uu[3], nn[3], ghn[3]);
          const unsigned go = 2u * (wo[ni] + jb4);
          __builtin_amdgcn_raw_buffer_store_b128((enc_u32x4){g0.x, g0.y, g1.x, g1.y}, bgs, go, 2 * t64, LFI_ENC_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128((enc_u32x4){g2.x, g2.y, g3.x, g3.y}, bgs, go, 2 * t64 + 16, LFI_ENC_ST_AUX);
          enc_st4(hn, bhs, wo[ni] + jb4, t64);
        } else if (STASH) {
          enc_st4(nn, bgs, 4u * wo[ni] + jb4, 2 * h4 + t64);
          enc_st4(ghn, bgs, 4u * wo[ni] + jb4, 3 * h4 + t64);
          enc_st4(hn, bhs, wo[ni] + jb4, t64);
        }
        {
          uint2 h, l;
          split2(hn[0], hn[1], &h.x, &l.x);
          split2(hn[2], hn[3], &h.y, &l.y);
          *reinterpret_cast<uint2*>(Xw + ni * 16 * ldx + 16 * qt) = h;
          *reinterpret_cast<uint2*>(Xw + IMG + ni * 16 * ldx + 16 * qt) = l;
        }
        hprev[qt][ni] = hn;
      }
    };
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      if (p > 0) load_biases(p - 1);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int i = p * 8 + m, ni = m >> 1;
        if (p < 4) {
          if (i + 1 < 32) sload((i + 1) & 7, sh[(i + 1) & 1], sl[(i + 1) & 1]);
          const ebf16x8 (&bh)[4] = sh[i & 1];
          const ebf16x8 (&bl)[4] = sl[i & 1];
#ifdef LFI_T16_NO_MFMA
          if (m == 0)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int n4 = 0; n4 < 4; ++n4) acc[p & 1][n4][g] = f32x4{(float)bh[n4][0], (float)bl[n4][1], (float)wr[(i * 3 + g) % WR][0].v[0], (float)wr[(i * 3 + g) % WR][1].v[1]};
#else
#pragma unroll
          for (int g = 0; g < 3; ++g) {
#ifndef LFI_T16_NO_W
            {
              const int jn = (i * 3 + g + WR - 1) % 96, in = jn / 3;   // (past the last block: the next step's first ones)
              wr[jn % WR][0].u = wload(in & 7, jn % 3, in >> 3, 0);
              wr[jn % WR][1].u = wload(in & 7, jn % 3, in >> 3, 1);
            }
#endif
            const ebf16x8 whi = wr[(i * 3 + g) % WR][0].v, wlo = wr[(i * 3 + g) % WR][1].v;
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) {
              const f32x4 c0 = m == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[p & 1][n4][g];
              acc[p & 1][n4][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, bl[n4], c0, 0, 0, 0);
            }
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) acc[p & 1][n4][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, bh[n4], acc[p & 1][n4][g], 0, 0, 0);
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) acc[p & 1][n4][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, bh[n4], acc[p & 1][n4][g], 0, 0, 0);
          }
#endif
        }
#ifdef LFI_T16_NO_EPI
        if (p > 0 && m == 7) {
          const int qt = p - 1;
#pragma unroll
          for (int n4 = 0; n4 < 4; ++n4) {
            const f32x4 hn = acc[qt & 1][n4][0] + acc[qt & 1][n4][1] + acc[qt & 1][n4][2] + xin[n4][0] + xin[n4][1] + xin[n4][2];
            uint2 h, l;
            split2(hn[0], hn[1], &h.x, &l.x);
            split2(hn[2], hn[3], &h.y, &l.y);
            *reinterpret_cast<uint2*>(Xw + n4 * 16 * ldx + 16 * qt) = h;
            *reinterpret_cast<uint2*>(Xw + IMG + n4 * 16 * ldx + 16 * qt) = l;
            hprev[qt][n4] = hn;
          }
        }
#else
        // epilogue chunk (unit tile p - 1, window tile ni): first half in the even block, second half in the odd one
        if (p > 0) epi_half(p - 1, ni, m & 1, acc[(p - 1) & 1][ni], xin[ni]);
#endif
        if (p < 4) xload(p, ni, m & 1, xin[ni]);   // projected inputs of tile p, for its epilogue one phase from now
#if LFI_T16_NV > 0
        if (p > 0 && p < 4) {
#pragma unroll
          for (int k = 0; k < 36; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, LFI_T16_NV, 0);
          }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // h_s is complete in its image, and every wave has read h_{s-1} for the last time
  }
  // cat(seq[:, -1], h_n[0]): the final state, once or twice (glow/models.py:63-64) - outside the step loop, whose body stays one
  // basic block (a branch would cut the scheduling regions the interleave lives in)
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int w = wbase + 16 * ni + wl;
    if (w < a.F) {
      float* c = a.cond + (long)w * a.ldcond + a.col + cg * 64 + 4 * jq;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        *reinterpret_cast<f32x4*>(c + 16 * mi) = hprev[mi][ni];
        if (a.dup) *reinterpret_cast<f32x4*>(c + hid + 16 * mi) = hprev[mi][ni];
      }
    }
  }
}

// BPTT of the same block of windows in one workgroup: dh lives in the accumulator layout of the wave that owns
// (rows, hidden) tile (rg, cg); per step the gate derivatives are taken in registers, written to dgi / dgh (the
// deferred weight-gradient GEMMs read those) and fed gate by gate through LDS as the A operand of
// dh_{s-1} = dgh_s W_hh + dh_s * u   (K = 3 hid, B = W_hh rows streamed from L2).
// Round 6: this kernel is the exact-f32 form only. Its bf16x3 twin (the same layout with three bf16 products per k-step) was the
// fallback behind LFI_ENC_WIDE_BWD=0 and for shapes the row-layout kernels do not take; built with the SLP vectoriser it returned
// different bits on 19 of 19 repeat launches (spill reloads inside its counted-vmcnt product loops, profiles/round5_slp_chase.md;
// the instruction at fault was never found). A kernel that is nondeterministic under a legal compiler flag does not ship: it is
// gone, and those cases run this kernel instead (bit-exact fp32 FMA chains on v_mfma_f32_32x32x2_f32 - slower, never wrong).
// One workgroup per CU's worth of registers (launch bound 1, not 2): at two it fits without the SLP vectoriser (230 VGPRs) but spills
// 26 with it - and the spilling SLP build of THIS form turned out to differ between whole passes as well (p2_face, batch 256:
// profiles/round6_determinism.txt, first run), as its bf16x3 twin had: spill code inside these product loops is what goes wrong, in
// whichever arithmetic. With the whole register file it never spills, in either build.
__global__ __launch_bounds__(ENC_NT, 1) void enc_gru_bwd_fused_kernel(EncArgs a, EncFused q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cg = wave % q.ncg, rg = wave / q.ncg;
  const int hid = a.hid, G3 = 3 * hid, ldk = q.R + 1, Jp = q.Jp;
  const int jb = cg * 64 + l31;
  const bool jok0 = jb < hid, jok1 = jb + 32 < hid;
  const int wbase = blockIdx.x * q.R;
  float* Dls = enc_smem;  // one gate's derivatives: element (row, k) at Dls[k * ldk + row], rows k >= hid stay zero
  for (int i = tid; i < q.Kp * ldk; i += ENC_NT) Dls[i] = 0.0f;
  // roww[row] = w * hid * 4 (rows past F clamped), live[row] = 1 for real windows (bias sums skip the clamped duplicates)
  unsigned* roww = reinterpret_cast<unsigned*>(enc_smem + q.Kp * ldk);
  float* rlive = reinterpret_cast<float*>(roww + q.R);
  for (int i = tid; i < q.R; i += ENC_NT) {
    roww[i] = (unsigned)min(wbase + i, a.F - 1) * (unsigned)(hid * 4);
    rlive[i] = wbase + i < a.F ? 1.0f : 0.0f;
  }
  const int boff = half * Jp + jb;
  const int aoff = half * ldk + rg * 32 + l31;
  const int nkp = q.Kp >> 1;

  f32x16 dh[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int w = min(wbase + enc_rowl(rg, r, half), a.F - 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = jb + 32 * t;
      float v = 0.0f;
      if (j < hid) {
        const float* dc = a.dcond + (long)w * a.lddcond + a.col;
        v = a.dup ? dc[j] + dc[hid + j] : dc[j];
      }
      dh[t][r] = v;
    }
  }
  __syncthreads();
  // bias gradients db_ih = sum (dar, dau, dan), db_hh = sum (dar, dau, dan*r) over windows and steps: every lane sums its own
  // column over its rows and all steps in a fixed order; the column sums of the per-(workgroup, row group) partials finish it
  float bsum[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  for (int s = a.hist - 1; s >= 0; --s) {
    float dg[2][2][16];  // gates 1 and 2; gate 0 goes straight into the LDS operand image (free since the last barrier)
    f32x16 acc[2];
    const float hp_on = s > 0 ? 1.0f : 0.0f;
    const int sp = s > 0 ? s - 1 : 0;
    // uniform buffer resources of this step + 32-bit per-lane byte offsets; lane coordinates laundered so that the
    // per-register address arithmetic is not hoisted out of the step loop (see the forward kernel)
    int halfv = half, jv = jb;
    asm volatile("" : "+v"(halfv), "+v"(jv));
    const enc_rsrc bgs = enc_buf(a.gates + (long)s * a.F * 4 * hid, (long)a.F * hid * 16);
    const enc_rsrc bhp = enc_buf(a.hseq + (long)sp * a.F * hid, (long)a.F * hid * 4);
    // d(pre-activation) on the input side differs from the hidden side only in the n gate (dan vs dan * r): dgi keeps that
    // one block ([hist][F][hid]); the scatter and the bias sums take d r, d z from dgh - a third less to write
    const enc_rsrc bgi = enc_buf(a.dgi + (long)s * a.F * hid, (long)a.F * hid * 4);
    const enc_rsrc bgh = enc_buf(a.dgh + (long)s * a.F * G3, (long)a.F * G3 * 4);
    const unsigned h4 = (unsigned)hid * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = jv + 32 * t;
      if (j < hid) {
        const unsigned j4 = (unsigned)j * 4u;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
          float gr[8], gu[8], gn[8], gg[8], hp[8], live[8];
          unsigned wo[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int rl = enc_rowl(rg, rh * 8 + e, halfv);
            wo[e] = roww[rl];
            live[e] = rlive[rl];
            const unsigned go = 4u * wo[e] + j4;
            gr[e] = enc_ld_stream(bgs, go, 0); gu[e] = enc_ld_stream(bgs, go, h4); gn[e] = enc_ld_stream(bgs, go, 2 * h4);
            gg[e] = enc_ld_stream(bgs, go, 3 * h4);
            hp[e] = enc_ld_stream(bhp, wo[e] + j4, 0);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = rh * 8 + e;
            const float rr = gr[e], uu = gu[e], nn = gn[e], ghn = gg[e];
            const float dhn = dh[t][r];
            const float du = dhn * (hp[e] * hp_on - nn);
            const float dn = dhn * (1.0f - uu);
            const float dan = dn * (1.0f - nn * nn);
            const float dau = du * uu * (1.0f - uu);
            const float dar = dan * ghn * rr * (1.0f - rr);
            const float danr = dan * rr;
            const unsigned o = 3u * wo[e] + j4;
            enc_st(dan, bgi, wo[e] + j4, 0);
            enc_st(dar, bgh, o, 0); enc_st(dau, bgh, o, h4); enc_st(danr, bgh, o, 2 * h4);
            dg[0][t][r] = dau; dg[1][t][r] = danr;
            if (s > 0) {
              const int rl = enc_rowl(rg, r, halfv);
              Dls[j * ldk + rl] = dar;
            }
            acc[t][r] = dhn * uu;
            bsum[0][t] += live[e] * dar; bsum[1][t] += live[e] * dau; bsum[2][t] += live[e] * dan; bsum[3][t] += live[e] * danr;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { dg[0][t][r] = 0.f; dg[1][t][r] = 0.f; acc[t][r] = 0.f; }
      }
    }
    if (s == 0) break;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (g > 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = enc_rowl(rg, r, halfv);
        if (jok0) Dls[jv * ldk + rl] = dg[g - 1][0][r];
        if (jok1) Dls[(jv + 32) * ldk + rl] = dg[g - 1][1][r];
      }
      __syncthreads();
      {
        float a0[ENC_KC], a1[ENC_KC], b0[ENC_KC][2], b1[ENC_KC][2];
        const float* __restrict__ wg = q.wpad + (long)g * q.Kp * Jp;
        auto load = [&](int kp0, float (&av)[ENC_KC], float (&bv)[ENC_KC][2]) {
          const float* __restrict__ wk = wg + (long)kp0 * 2 * Jp;  // uniform base, 32-bit per-lane offset: saddr loads
          const float* ak = Dls + kp0 * 2 * ldk;
#pragma unroll
          for (int u = 0; u < ENC_KC; ++u) {
            av[u] = ak[u * 2 * ldk + aoff];
            const float* __restrict__ wu = wk + u * 2 * Jp;  // uniform
            bv[u][0] = wu[(unsigned)boff];
            bv[u][1] = (wu + 32)[(unsigned)boff];
          }
        };
        auto mma = [&](const float (&av)[ENC_KC], const float (&bv)[ENC_KC][2]) {
#pragma unroll
          for (int u = 0; u < ENC_KC; ++u) {
            acc[0] = mfma32(av[u], bv[u][0], acc[0]);
            acc[1] = mfma32(av[u], bv[u][1], acc[1]);
          }
        };
        load(0, a0, b0);
        int kp = 0;
        for (; kp + 2 * ENC_KC < nkp; kp += 2 * ENC_KC) {
          load(kp + ENC_KC, a1, b1);
          __builtin_amdgcn_sched_barrier(0);
          mma(a0, b0);
          load(kp + 2 * ENC_KC, a0, b0);
          __builtin_amdgcn_sched_barrier(0);
          mma(a1, b1);
        }
        load(kp + ENC_KC, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        mma(a1, b1);
      }
      __syncthreads();  // before the next gate overwrites Dls
    }
    dh[0] = acc[0];
    dh[1] = acc[1];
  }
  if (a.bias_part) {
    float* bp = a.bias_part + ((long)blockIdx.x * (ENC_NW / q.ncg) + rg) * 4 * hid;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v = bsum[qq][t] + __shfl_xor(bsum[qq][t], 32, 64);
        const int j = jb + 32 * t;
        if (half == 0 && j < hid) bp[qq * hid + j] = v;
      }
  }
}

// ---- BPTT with the row-layout epilogue (the counterpart of enc_gru_fwd_wide_kernel; bf16x3 products).
// The accumulator-layout kernel above moves every stash value with its own 4-byte access (gates r, z, n, W_hn h and h_{s-1}
// in; d n, d r, d z, d n * r out: 9 wave instructions per (row, 32 columns) and step) and writes the three gate images for
// the MFMA A operand with 2-byte LDS stores (192 per lane and step). Here d h comes out of the accumulators through the
// wave-private transpose tile, a lane owns 4 consecutive hidden units of 8 windows, everything above is a 16-byte access,
// the images are written 8 bytes at a time, and the tile region doubles as a SECOND image pair while no transpose is in
// flight, so that the r and z gate products run back to back: four workgroup barriers per step instead of six.
// A2 (lfi_enc_desc.bwd_two_products): two bf16 products per k-step instead of three - the A operand, d(gate pre-activations),
// enters rounded to bf16 (its lo image is neither written nor read, the a_lo * w_hi MFMA not issued), as the backward GEMM
// classes do from 8192 frames up (profiles/precision_sweep_b256.md: this recurrence is one of them, class enc_bptt).
template <bool A2, bool S16>
__global__ __launch_bounds__(ENC_NT, 2) void enc_gru_bwd_wide_kernel(EncArgs a, EncFused q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cg = wave % q.ncg, rg = wave / q.ncg;
  const int hid = a.hid, G3 = 3 * hid, Jp = q.Jp;
  const int wbase = blockIdx.x * q.R;
  const int ldx = q.Kp + 8;
  const int img = q.R * ldx;                                   // bf16 elements of one image
  __bf16* Xhi = reinterpret_cast<__bf16*>(enc_smem);
  __bf16* Xlo = Xhi + img;
  float* Treg = reinterpret_cast<float*>(Xlo + img);           // transpose tiles | second image pair (Yhi, Ylo)
  const int tfloats = max(ENC_NW * 32 * ENC_TP, img);          // (2 * img bf16 = img floats)
  float* T = Treg + wave * (32 * ENC_TP);
  __bf16* Yhi = reinterpret_cast<__bf16*>(Treg);
  __bf16* Ylo = Yhi + img;
  unsigned* roww = reinterpret_cast<unsigned*>(Treg + tfloats);
  float* rlive = reinterpret_cast<float*>(roww + q.R);
  for (int i = tid; i < img; i += ENC_NT) { Xhi[i] = (__bf16)0.0f; Xlo[i] = (__bf16)0.0f; }
  for (int i = tid; i < tfloats; i += ENC_NT) Treg[i] = 0.0f;   // (also the k padding of the second image pair)
  for (int i = tid; i < q.R; i += ENC_NT) {
    roww[i] = (unsigned)min(wbase + i, a.F - 1) * (unsigned)(hid * 4);
    rlive[i] = wbase + i < a.F ? 1.0f : 0.0f;
  }
  const int rsub = lane >> 4, c4 = (lane & 15) * 4;
  const int j0 = cg * 64 + c4;
  const bool jok = j0 < hid;
  const unsigned h4 = (unsigned)hid * 4u;
  const enc_rsrc bdc = enc_buf(a.dcond, ((long)a.F * a.lddcond) * 4);
  f32x4 dhu[8];          // d h_s * z_s: the part of d h_{s-1} that does not go through W_hh (row layout)
  f32x4 bsum[4];         // column sums of d r, d z, d n, d n * r over this lane's rows and all steps
#pragma unroll
  for (int i = 0; i < 8; ++i) dhu[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) bsum[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  __syncthreads();

  for (int s = a.hist - 1; s >= 0; --s) {
    int rsv = rsub, cv = c4, jv = j0;
    asm volatile("" : "+v"(rsv), "+v"(cv), "+v"(jv));
    const unsigned j4 = (unsigned)jv * 4u, oob = jv < hid ? 0u : 0x80000000u;
    float* Trow = T + rsv * ENC_TP + cv;
    float* Tacc = T + (4 * half) * ENC_TP + l31;
    // ---- d h_s in the row layout
    f32x4 dh[8];
    if (s == a.hist - 1) {
      const unsigned cb = (unsigned)a.col * 4u + j4 + oob;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned wrow = (roww[rg * 32 + 4 * i + rsv] / h4) * (unsigned)(a.lddcond * 4);
        dh[i] = enc_ld4(bdc, wrow + cb, 0);
        if (a.dup) dh[i] += enc_ld4(bdc, wrow + cb, h4);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Tacc[((r & 3) + 8 * (r >> 2)) * ENC_TP + 32 * t] = acc[t][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) dh[i] = *reinterpret_cast<const f32x4*>(Trow + 4 * i * ENC_TP) + dhu[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();   // B0: every wave is done with its transpose tile (the second image pair overlays those) and with the
                       // previous step's n-gate product (which read the first image pair)
    if (q.Kp > hid) {   // the transposes left accumulator bits in the second pair's k padding: NaN patterns x 0 would poison the product
      const int pad = q.Kp - hid;
      for (int i = tid; i < q.R * pad; i += ENC_NT) {
        const int rl = i / pad, c = hid + (i - rl * pad);
        Yhi[rl * ldx + c] = (__bf16)0.0f; Ylo[rl * ldx + c] = (__bf16)0.0f;
      }
    }
    const int sp = s > 0 ? s - 1 : 0;
    const float hp_on = s > 0 ? 1.0f : 0.0f;
    const enc_rsrc bgs = S16 ? enc_buf(reinterpret_cast<const _Float16*>(a.gates) + (long)s * a.F * 4 * hid, (long)a.F * hid * 8)
                             : enc_buf(a.gates + (long)s * a.F * 4 * hid, (long)a.F * hid * 16);
    const enc_rsrc bhp = enc_buf(a.hseq + (long)sp * a.F * hid, (long)a.F * hid * 4);
    // A2: the gradient stash leaves as bf16 (what its consumers - the dW_hh product's rounded A operand, the window scatter in
    // front of dW_ih's - use of it): half the bytes out here and half the bytes into both of them
    const enc_rsrc bgi = A2 ? enc_buf(reinterpret_cast<const __bf16*>(a.dgi) + (long)s * a.F * hid, (long)a.F * hid * 2)
                            : enc_buf(a.dgi + (long)s * a.F * hid, (long)a.F * hid * 4);
    const enc_rsrc bgh = A2 ? enc_buf(reinterpret_cast<const __bf16*>(a.dgh) + (long)s * a.F * G3, (long)a.F * G3 * 2)
                            : enc_buf(a.dgh + (long)s * a.F * G3, (long)a.F * G3 * 4);
    f32x4 danr[8];
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      f32x4 gr[4], gu[4], gn[4], gg[4], hp[4];
      unsigned wo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rl = rg * 32 + 4 * (4 * ih + u) + rsv;
        wo[u] = roww[rl];
        if (S16) {   // fp16 stash, [window][unit][r, z, n, W_hn h]: two 16-byte loads hold this lane's four units
          const unsigned go = 2u * (wo[u] + j4) + oob;
          const enc_u32x4 p0 = __builtin_amdgcn_raw_buffer_load_b128(bgs, go, 0, 2), p1 = __builtin_amdgcn_raw_buffer_load_b128(bgs, go, 16, 2);
          const f32x4 u0 = enc_unpack_gates(p0[0], p0[1]), u1 = enc_unpack_gates(p0[2], p0[3]);
          const f32x4 u2 = enc_unpack_gates(p1[0], p1[1]), u3 = enc_unpack_gates(p1[2], p1[3]);
          gr[u] = f32x4{u0[0], u1[0], u2[0], u3[0]}; gu[u] = f32x4{u0[1], u1[1], u2[1], u3[1]};
          gn[u] = f32x4{u0[2], u1[2], u2[2], u3[2]}; gg[u] = f32x4{u0[3], u1[3], u2[3], u3[3]};
        } else {
          const unsigned go = 4u * wo[u] + j4 + oob;
          gr[u] = enc_ld4s(bgs, go, 0); gu[u] = enc_ld4s(bgs, go, h4); gn[u] = enc_ld4s(bgs, go, 2 * h4); gg[u] = enc_ld4s(bgs, go, 3 * h4);
        }
        hp[u] = enc_ld4s(bhp, wo[u] + j4 + oob, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = 4 * ih + u;
        const int rl = rg * 32 + 4 * i + rsv;
        const float live = rlive[rl];
        f32x4 dan, dau, dar, dnr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float rr = gr[u][e], uu = gu[u][e], nn = gn[u][e], ghn = gg[u][e];
          const float dhn = dh[i][e];
          const float du = dhn * (hp[u][e] * hp_on - nn);
          const float dn = dhn * (1.0f - uu);
          dan[e] = dn * (1.0f - nn * nn);
          dau[e] = du * uu * (1.0f - uu);
          dar[e] = dan[e] * ghn * rr * (1.0f - rr);
          dnr[e] = dan[e] * rr;
          dhu[i][e] = dhn * uu;
        }
        if (A2) {
          const unsigned j2 = j4 >> 1, h2 = h4 >> 1;
          enc_st2h(dan, bgi, (wo[u] >> 1) + j2 + oob, 0);
          const unsigned o = 3u * (wo[u] >> 1) + j2 + oob;
          enc_st2h(dar, bgh, o, 0); enc_st2h(dau, bgh, o, h2); enc_st2h(dnr, bgh, o, 2 * h2);
        } else {
          enc_st4(dan, bgi, wo[u] + j4 + oob, 0);
          const unsigned o = 3u * wo[u] + j4 + oob;
          enc_st4(dar, bgh, o, 0); enc_st4(dau, bgh, o, h4); enc_st4(dnr, bgh, o, 2 * h4);
        }
        bsum[0] += live * dar; bsum[1] += live * dau; bsum[2] += live * dan; bsum[3] += live * dnr;
        danr[i] = dnr;
        if (s > 0 && jok) {   // gate images for the products: d r -> first pair, d z -> second pair
          uint2 h, l;
          split2(dar[0], dar[1], &h.x, &l.x); split2(dar[2], dar[3], &h.y, &l.y);
          *reinterpret_cast<uint2*>(Xhi + rl * ldx + jv) = h;
          if (!A2) *reinterpret_cast<uint2*>(Xlo + rl * ldx + jv) = l;
          split2(dau[0], dau[1], &h.x, &l.x); split2(dau[2], dau[3], &h.y, &l.y);
          *reinterpret_cast<uint2*>(Yhi + rl * ldx + jv) = h;
          if (!A2) *reinterpret_cast<uint2*>(Ylo + rl * ldx + jv) = l;
        }
      }
    }
    if (s == 0) break;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    auto product = [&](int g, const __bf16* Ih, const __bf16* Il) {   // acc += image_g (R x hid) W_hh[g] (hid x hid)
      const int nkt = q.Kp >> 4, nct = Jp >> 5;
      const __bf16* xh = Ih + (rg * 32 + l31) * ldx + 8 * half;
      const __bf16* xl = Il + (rg * 32 + l31) * ldx + 8 * half;
      ebf16x8 ah0, al0, ah1, al1;
      EncFrag f0[2][2], f1[2][2];  // [t][plane]
      auto load = [&](int kt, ebf16x8& ah, ebf16x8& al, EncFrag (&f)[2][2]) {
        ah = *reinterpret_cast<const ebf16x8*>(xh + kt * 16);
        if (!A2) al = *reinterpret_cast<const ebf16x8*>(xl + kt * 16);
        else al = ah;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint4* __restrict__ wf = q.wfrag + ((long)((g * nkt + ENC_WKT(kt)) * nct + cg * 2 + t) * 2) * 64;  // uniform
          f[t][0].u = wf[(unsigned)lane];
          f[t][1].u = (wf + 64)[(unsigned)lane];
        }
      };
      auto mma = [&](const ebf16x8& ah, const ebf16x8& al, const EncFrag (&f)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (!A2) acc[t] = ENC_MFMA(al, f[t][0].v, acc[t], 0, 0, 0);
          acc[t] = ENC_MFMA(ah, f[t][1].v, acc[t], 0, 0, 0);
          acc[t] = ENC_MFMA(ah, f[t][0].v, acc[t], 0, 0, 0);
        }
      };
      load(0, ah0, al0, f0);
      int kt = 0;
      for (; kt + 2 < nkt; kt += 2) {   // unconditional loads (see the forward kernel)
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        load(kt + 2, ah0, al0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah1, al1, f1);
      }
      if (kt + 1 < nkt) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        mma(ah1, al1, f1);
      } else {
        mma(ah0, al0, f0);
      }
    };
    __syncthreads();   // B1: both image pairs complete
    product(0, Xhi, Xlo);
    product(1, Yhi, Ylo);
    __syncthreads();   // B2: every wave has read the first pair
    if (jok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = rg * 32 + 4 * i + rsv;
        uint2 h, l;
        split2(danr[i][0], danr[i][1], &h.x, &l.x); split2(danr[i][2], danr[i][3], &h.y, &l.y);
        *reinterpret_cast<uint2*>(Xhi + rl * ldx + jv) = h;
        if (!A2) *reinterpret_cast<uint2*>(Xlo + rl * ldx + jv) = l;
      }
    }
    __syncthreads();   // B3
    product(2, Xhi, Xlo);
  }
  if (a.bias_part && jok) {
    float* bp = a.bias_part + ((long)blockIdx.x * (ENC_NW / q.ncg) + rg) * 4 * hid;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      f32x4 v = bsum[qq];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += __shfl_xor(v[e], 16, 64);
        v[e] += __shfl_xor(v[e], 32, 64);
      }
      if (rsub == 0) *reinterpret_cast<f32x4*>(bp + qq * hid + j0) = v;
    }
  }
}

// ---- BPTT with two 32-row tiles per wave (64 windows per row group), one workgroup per CU: the counterpart of
// enc_gru_fwd_r64_kernel (same reasoning: the d gates x W_hh products keep their MFMA work, every weight fragment feeds both row
// tiles). Same phases and barriers as enc_gru_bwd_wide_kernel, every row-layout phase run once per row tile. Measured (p2_face
// shape, two products, fp16 gate stash): 0.57 -> 0.52 ms per launch.
template <bool A2, bool S16>
__global__ __launch_bounds__(ENC_NT, 1) void enc_gru_bwd_r64_kernel(EncArgs a, EncFused q) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int cg = wave % q.ncg, rg = wave / q.ncg;
  const int hid = a.hid, G3 = 3 * hid, Jp = q.Jp;
  const int R2 = 2 * q.R;
  const int wbase = blockIdx.x * R2;
  const int ldx = q.Kp + 8;
  const int img = R2 * ldx;                                     // bf16 elements of one image
  __bf16* Xhi = reinterpret_cast<__bf16*>(enc_smem);
  __bf16* Xlo = Xhi + img;
  float* Treg = reinterpret_cast<float*>(Xlo + img);           // transpose tiles | second image pair (Yhi, Ylo)
  const int tfloats = max(ENC_NW * 2 * 32 * ENC_TP, img);      // (2 * img bf16 = img floats)
  float* Tw = Treg + wave * (2 * 32 * ENC_TP);
  __bf16* Yhi = reinterpret_cast<__bf16*>(Treg);
  __bf16* Ylo = Yhi + img;
  // A2 (two products: the hi pieces of the gate derivatives only): the first pair's lo image is never written - its space holds the
  // THIRD gate's image (d n * r), so the three products run back to back: no 64-register copy of d n * r across the first two, no
  // image rewrite between them, three barriers per step instead of four (round 5; same values in the same order)
  __bf16* Zhi = Xlo;
  unsigned* roww = reinterpret_cast<unsigned*>(Treg + tfloats);
  float* rlive = reinterpret_cast<float*>(roww + R2);
  for (int i = tid; i < img; i += ENC_NT) { Xhi[i] = (__bf16)0.0f; Xlo[i] = (__bf16)0.0f; }
  for (int i = tid; i < tfloats; i += ENC_NT) Treg[i] = 0.0f;   // (also the k padding of the second image pair)
  for (int i = tid; i < R2; i += ENC_NT) {
    roww[i] = (unsigned)min(wbase + i, a.F - 1) * (unsigned)(hid * 4);
    rlive[i] = wbase + i < a.F ? 1.0f : 0.0f;
  }
  const int rsub = lane >> 4, c4 = (lane & 15) * 4;
  const int j0 = cg * 64 + c4;
  const bool jok = j0 < hid;
  const unsigned h4 = (unsigned)hid * 4u;
  const enc_rsrc bdc = enc_buf(a.dcond, ((long)a.F * a.lddcond) * 4);
  f32x4 dhu[2][8];       // d h_s * z_s: the part of d h_{s-1} that does not go through W_hh (row layout)
  f32x4 bsum[4];         // column sums of d r, d z, d n, d n * r over this lane's rows and all steps
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int i = 0; i < 8; ++i) dhu[rt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) bsum[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc[2][2];      // [row tile][column tile]
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.0f;
  __syncthreads();
  // PIPE (two products + fp16 gate stash = the training default; round 5): the step's stash comes in four batches of 12 loads
  // (row tile x half, 4 rows each). Round 4 issued a batch and waited for it, four exposed HBM latencies per step; here batch b + 1 is
  // in flight while batch b is worked on. Same values, same order.
#ifdef LFI_ENC_BWD_NOPIPE   // (same-box A/B builds: tools/build_variant.sh; LFI_ENC_BWD_NOZ keeps round 4's two image phases)
  constexpr bool PIPE = false;
#else
  constexpr bool PIPE = A2 && S16;
#endif
#ifdef LFI_ENC_BWD_NOZ
  constexpr bool Z3 = false;
#else
  constexpr bool Z3 = A2;
#endif
  // Two buffers (batch b -> buffer b & 1). Batch 0 of the NEXT step is issued right after this step's last product has issued its
  // last weight load: it travels under the closing barrier, the transposes and B0. Measured (p2_face / p2_speech, ms per launch, same
  // box): round 4's form 0.517 / 0.353; this 0.487 / 0.331. Issued EARLIER - during this step's last batch, in front of the products -
  // it sits in the same in-order queue in front of their weight fragments and gives most of that back (0.516 / 0.358); with a third
  // buffer, one batch earlier still: 20 spilled VGPRs. Neither kept.
  enc_u32x4 rp0[2][4], rp1[2][4];   // [buffer][row of the batch]: the two 16-byte halves of a lane's four units' gates
  f32x4 rhp[2][4];                  // h_{s-1}
  unsigned rwo[2][4];
  constexpr int BUF[4] = {0, 1, 0, 1};
  auto issue = [&](int bi, int sq) {   // batch bi = 2 rt + ih of step sq -> buffer BUF[bi]
    int rsv = rsub, jv = j0;
    asm volatile("" : "+v"(rsv), "+v"(jv));
    const unsigned j4 = (unsigned)jv * 4u, oob = jv < hid ? 0u : 0x80000000u;
    const enc_rsrc gs = enc_buf(reinterpret_cast<const _Float16*>(a.gates) + (long)sq * a.F * 4 * hid, (long)a.F * hid * 8);
    const enc_rsrc hs = enc_buf(a.hseq + (long)(sq > 0 ? sq - 1 : 0) * a.F * hid, (long)a.F * hid * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rl = rg * 64 + (bi >> 1) * 32 + 4 * (4 * (bi & 1) + u) + rsv;
      const unsigned w = roww[rl];
      const unsigned go = 2u * (w + j4) + oob;
      rwo[BUF[bi]][u] = w;
      rp0[BUF[bi]][u] = __builtin_amdgcn_raw_buffer_load_b128(gs, go, 0, 2);
      rp1[BUF[bi]][u] = __builtin_amdgcn_raw_buffer_load_b128(gs, go, 16, 2);
      rhp[BUF[bi]][u] = enc_ld4s(hs, w + j4 + oob, 0);
    }
  };
  if (PIPE) issue(0, a.hist - 1);

  for (int s = a.hist - 1; s >= 0; --s) {
    // ---- d h_s in the row layout
    f32x4 dh[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      int rsv = rsub, cv = c4, jv = j0;
      asm volatile("" : "+v"(rsv), "+v"(cv), "+v"(jv));
      const unsigned j4 = (unsigned)jv * 4u, oob = jv < hid ? 0u : 0x80000000u;
      float* T = Tw + rt * (32 * ENC_TP);
      float* Trow = T + rsv * ENC_TP + cv;
      float* Tacc = T + (4 * half) * ENC_TP + l31;
      if (s == a.hist - 1) {
        const unsigned cb = (unsigned)a.col * 4u + j4 + oob;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned wrow = (roww[rg * 64 + rt * 32 + 4 * i + rsv] / h4) * (unsigned)(a.lddcond * 4);
          dh[rt][i] = enc_ld4(bdc, wrow + cb, 0);
          if (a.dup) dh[rt][i] += enc_ld4(bdc, wrow + cb, h4);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) Tacc[((r & 3) + 8 * (r >> 2)) * ENC_TP + 32 * t] = acc[rt][t][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) dh[rt][i] = *reinterpret_cast<const f32x4*>(Trow + 4 * i * ENC_TP) + dhu[rt][i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();   // B0: every wave is done with its transpose tiles (the second image pair overlays those) and with the
                       // previous step's n-gate product (which read the first image pair)
    if (q.Kp > hid) {   // the transposes left accumulator bits in the second pair's k padding: NaN patterns x 0 would poison the product
      const int pad = q.Kp - hid;
      for (int i = tid; i < R2 * pad; i += ENC_NT) {
        const int rl = i / pad, c = hid + (i - rl * pad);
        Yhi[rl * ldx + c] = (__bf16)0.0f; Ylo[rl * ldx + c] = (__bf16)0.0f;
      }
    }
    const int sp = s > 0 ? s - 1 : 0;
    const float hp_on = s > 0 ? 1.0f : 0.0f;
    const enc_rsrc bgs = S16 ? enc_buf(reinterpret_cast<const _Float16*>(a.gates) + (long)s * a.F * 4 * hid, (long)a.F * hid * 8)
                             : enc_buf(a.gates + (long)s * a.F * 4 * hid, (long)a.F * hid * 16);
    const enc_rsrc bhp = enc_buf(a.hseq + (long)sp * a.F * hid, (long)a.F * hid * 4);
    const enc_rsrc bgi = A2 ? enc_buf(reinterpret_cast<const __bf16*>(a.dgi) + (long)s * a.F * hid, (long)a.F * hid * 2)
                            : enc_buf(a.dgi + (long)s * a.F * hid, (long)a.F * hid * 4);
    const enc_rsrc bgh = A2 ? enc_buf(reinterpret_cast<const __bf16*>(a.dgh) + (long)s * a.F * G3, (long)a.F * G3 * 2)
                            : enc_buf(a.dgh + (long)s * a.F * G3, (long)a.F * G3 * 4);
    f32x4 danr[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      int rsv = rsub, jv = j0;
      asm volatile("" : "+v"(rsv), "+v"(jv));
      const unsigned j4 = (unsigned)jv * 4u, oob = jv < hid ? 0u : 0x80000000u;
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        f32x4 gr[4], gu[4], gn[4], gg[4], hp[4];
        unsigned wo[4];
        if (PIPE) {
          const int bi = 2 * rt + ih;
          if (bi < 3) issue(bi + 1, s);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const enc_u32x4 p0 = rp0[BUF[bi]][u], p1 = rp1[BUF[bi]][u];
            const f32x4 u0 = enc_unpack_gates(p0[0], p0[1]), u1 = enc_unpack_gates(p0[2], p0[3]);
            const f32x4 u2 = enc_unpack_gates(p1[0], p1[1]), u3 = enc_unpack_gates(p1[2], p1[3]);
            gr[u] = f32x4{u0[0], u1[0], u2[0], u3[0]}; gu[u] = f32x4{u0[1], u1[1], u2[1], u3[1]};
            gn[u] = f32x4{u0[2], u1[2], u2[2], u3[2]}; gg[u] = f32x4{u0[3], u1[3], u2[3], u3[3]};
            hp[u] = rhp[BUF[bi]][u];
            wo[u] = rwo[BUF[bi]][u];
          }
        }
#pragma unroll
        for (int u = 0; u < 4 && !PIPE; ++u) {
          const int rl = rg * 64 + rt * 32 + 4 * (4 * ih + u) + rsv;
          wo[u] = roww[rl];
          if (S16) {   // fp16 stash, [window][unit][r, z, n, W_hn h]: two 16-byte loads hold this lane's four units
            const unsigned go = 2u * (wo[u] + j4) + oob;
            const enc_u32x4 p0 = __builtin_amdgcn_raw_buffer_load_b128(bgs, go, 0, 2), p1 = __builtin_amdgcn_raw_buffer_load_b128(bgs, go, 16, 2);
            const f32x4 u0 = enc_unpack_gates(p0[0], p0[1]), u1 = enc_unpack_gates(p0[2], p0[3]);
            const f32x4 u2 = enc_unpack_gates(p1[0], p1[1]), u3 = enc_unpack_gates(p1[2], p1[3]);
            gr[u] = f32x4{u0[0], u1[0], u2[0], u3[0]}; gu[u] = f32x4{u0[1], u1[1], u2[1], u3[1]};
            gn[u] = f32x4{u0[2], u1[2], u2[2], u3[2]}; gg[u] = f32x4{u0[3], u1[3], u2[3], u3[3]};
          } else {
            const unsigned go = 4u * wo[u] + j4 + oob;
            gr[u] = enc_ld4s(bgs, go, 0); gu[u] = enc_ld4s(bgs, go, h4); gn[u] = enc_ld4s(bgs, go, 2 * h4); gg[u] = enc_ld4s(bgs, go, 3 * h4);
          }
          hp[u] = enc_ld4s(bhp, wo[u] + j4 + oob, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = 4 * ih + u;
          const int rl = rg * 64 + rt * 32 + 4 * i + rsv;
          const float live = rlive[rl];
          f32x4 dan, dau, dar, dnr;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float rr = gr[u][e], uu = gu[u][e], nn = gn[u][e], ghn = gg[u][e];
            const float dhn = dh[rt][i][e];
            const float du = dhn * (hp[u][e] * hp_on - nn);
            const float dn = dhn * (1.0f - uu);
            dan[e] = dn * (1.0f - nn * nn);
            dau[e] = du * uu * (1.0f - uu);
            dar[e] = dan[e] * ghn * rr * (1.0f - rr);
            dnr[e] = dan[e] * rr;
            dhu[rt][i][e] = dhn * uu;
          }
          if (A2) {
            const unsigned j2 = j4 >> 1, h2 = h4 >> 1;
            enc_st2h(dan, bgi, (wo[u] >> 1) + j2 + oob, 0);
            const unsigned o = 3u * (wo[u] >> 1) + j2 + oob;
            enc_st2h(dar, bgh, o, 0); enc_st2h(dau, bgh, o, h2); enc_st2h(dnr, bgh, o, 2 * h2);
          } else {
            enc_st4(dan, bgi, wo[u] + j4 + oob, 0);
            const unsigned o = 3u * wo[u] + j4 + oob;
            enc_st4(dar, bgh, o, 0); enc_st4(dau, bgh, o, h4); enc_st4(dnr, bgh, o, 2 * h4);
          }
          bsum[0] += live * dar; bsum[1] += live * dau; bsum[2] += live * dan; bsum[3] += live * dnr;
          if (!Z3) danr[rt][i] = dnr;
          if (s > 0 && jok) {   // gate images for the products: d r -> first pair, d z -> second pair (A2: d n * r -> the third image)
            uint2 h, l;
            split2(dar[0], dar[1], &h.x, &l.x); split2(dar[2], dar[3], &h.y, &l.y);
            *reinterpret_cast<uint2*>(Xhi + rl * ldx + jv) = h;
            if (!A2) *reinterpret_cast<uint2*>(Xlo + rl * ldx + jv) = l;
            split2(dau[0], dau[1], &h.x, &l.x); split2(dau[2], dau[3], &h.y, &l.y);
            *reinterpret_cast<uint2*>(Yhi + rl * ldx + jv) = h;
            if (!A2) *reinterpret_cast<uint2*>(Ylo + rl * ldx + jv) = l;
            if (Z3) {
              split2(dnr[0], dnr[1], &h.x, &l.x); split2(dnr[2], dnr[3], &h.y, &l.y);
              *reinterpret_cast<uint2*>(Zhi + rl * ldx + jv) = h;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s == 0) break;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.0f;
    auto product = [&](int g, const __bf16* Ih, const __bf16* Il) {   // acc += image_g (64 rows x hid) W_hh[g] (hid x hid)
      const int nkt = q.Kp >> 4, nct = Jp >> 5;
      const __bf16* xh = Ih + (rg * 64 + l31) * ldx + 8 * half;
      const __bf16* xl = Il + (rg * 64 + l31) * ldx + 8 * half;
      ebf16x8 ah0[2], al0[2], ah1[2], al1[2];
      EncFrag f0[2][2], f1[2][2];  // [t][plane]
      auto load = [&](int kt, ebf16x8 (&ah)[2], ebf16x8 (&al)[2], EncFrag (&f)[2][2]) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          ah[rt] = *reinterpret_cast<const ebf16x8*>(xh + rt * 32 * ldx + kt * 16);
          if (!A2) al[rt] = *reinterpret_cast<const ebf16x8*>(xl + rt * 32 * ldx + kt * 16);
          else al[rt] = ah[rt];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint4* __restrict__ wf = q.wfrag + ((long)((g * nkt + ENC_WKT(kt)) * nct + cg * 2 + t) * 2) * 64;  // uniform
          f[t][0].u = wf[(unsigned)lane];
          f[t][1].u = (wf + 64)[(unsigned)lane];
        }
      };
      auto mma = [&](const ebf16x8 (&ah)[2], const ebf16x8 (&al)[2], const EncFrag (&f)[2][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            if (!A2) acc[rt][t] = ENC_MFMA(al[rt], f[t][0].v, acc[rt][t], 0, 0, 0);
            acc[rt][t] = ENC_MFMA(ah[rt], f[t][1].v, acc[rt][t], 0, 0, 0);
            acc[rt][t] = ENC_MFMA(ah[rt], f[t][0].v, acc[rt][t], 0, 0, 0);
          }
      };
      load(0, ah0, al0, f0);
      int kt = 0;
      for (; kt + 2 < nkt; kt += 2) {   // unconditional loads (see the forward kernel)
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        load(kt + 2, ah0, al0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah1, al1, f1);
      }
      if (kt + 1 < nkt) {
        load(kt + 1, ah1, al1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ah0, al0, f0);
        mma(ah1, al1, f1);
      } else {
        mma(ah0, al0, f0);
      }
    };
    __syncthreads();   // B1: both image pairs complete
    product(0, Xhi, Xlo);
    product(1, Yhi, Ylo);
    if (Z3) {
      product(2, Zhi, Zhi);
      if (PIPE) issue(0, s - 1);   // (s >= 1 here: the step loop left at s == 0 above)
      __syncthreads();   // B2: every wave has read the second image (the next step's transpose tiles overlay it)
      continue;
    }
    __syncthreads();   // B2: every wave has read the first pair
    if (jok) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rl = rg * 64 + rt * 32 + 4 * i + rsub;
          uint2 h, l;
          split2(danr[rt][i][0], danr[rt][i][1], &h.x, &l.x); split2(danr[rt][i][2], danr[rt][i][3], &h.y, &l.y);
          *reinterpret_cast<uint2*>(Xhi + rl * ldx + j0) = h;
          if (!A2) *reinterpret_cast<uint2*>(Xlo + rl * ldx + j0) = l;
        }
    }
    __syncthreads();   // B3
    product(2, Xhi, Xlo);
    if (PIPE) issue(0, s - 1);
  }
  if (a.bias_part && jok) {
    float* bp = a.bias_part + ((long)blockIdx.x * (ENC_NW / q.ncg) + rg) * 4 * hid;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      f32x4 v = bsum[qq];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += __shfl_xor(v[e], 16, 64);
        v[e] += __shfl_xor(v[e], 32, 64);
      }
      if (rsub == 0) *reinterpret_cast<f32x4*>(bp + qq * hid + j0) = v;
    }
  }
}

template <typename Kf>
int enc_set_lds(Kf kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return LFI_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    lfi_set_error("window encoder: cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
    return LFI_ERR_LAUNCH;
  }
  return LFI_OK;
}

// shape of the fused path, or 0 when the recurrence is too wide for it
int enc_fused_shape(int hid, EncFused* q) {
  if (hid > 256) return 0;
  int tiles = lfi_cdiv(hid, 64), ncg = 1;
  while (ncg < tiles) ncg <<= 1;
  q->ncg = ncg;
  q->R = 32 * (ENC_NW / ncg);
  q->Kp = (hid + 15) & ~15;
  q->Jp = 64 * ncg;
  return 1;
}

int fill_args(const lfi_enc_desc* d, EncArgs* a, const char* who) {
  LFI_REQUIRE(d, "%s: null descriptor", who);
  LFI_REQUIRE(d->B > 0 && d->T > 0 && d->N > 0 && d->hist > 0 && d->hid > 0, "%s: bad dims", who);
  LFI_REQUIRE(d->start + d->N <= d->T, "%s: start + N > T", who);
  LFI_REQUIRE(d->hist <= d->start + 1, "%s: window longer than start+1", who);
  a->B = d->B; a->T = d->T; a->N = d->N; a->start = d->start; a->hist = d->hist; a->hid = d->hid;
  a->ldcond = d->ldcond; a->col = d->col; a->dup = d->dup;
  a->lstm = d->lstm ? 1 : 0; a->G = (d->lstm ? 4 : 3) * d->hid;
  a->F = d->N * d->B;
  return LFI_OK;
}

// part [rows][4 * hid] -> column sums; 32 columns per workgroup, 32 row groups (each a strided run of rows, added in order),
// then the 32 partial sums in a fixed order: bit-reproducible. (8 row groups of one 256-thread workgroup took 20 - 28 us per
// modality on 448 rows: a chain of 56 dependent loads per thread.)
// Blocks 0, 1 (d r, d z) go to both bias gradients, block 2 (d n) to db_ih, block 3 (d n * r) to db_hh's n block.
__global__ __launch_bounds__(1024) void enc_bias_fold_kernel(const float* __restrict__ part, long rows, int hid,
                                                            float* __restrict__ db_ih, float* __restrict__ db_hh) {
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c, W = 4 * hid;
  float acc = 0.0f;
  if (col < W)
    for (long r = rg; r < rows; r += 32) acc += part[r * W + col];
  red[rg][c] = acc;
  __syncthreads();
  if (rg == 0 && col < W) {
    float v = red[0][c];
#pragma unroll
    for (int i = 1; i < 32; ++i) v += red[i][c];
    const int blk = col / hid, j = col - blk * hid;
    if (blk < 2) { db_ih[col] = v; db_hh[col] = v; }
    else if (blk == 2) db_ih[col] = v;
    else db_hh[2 * hid + j] = v;
  }
}

int ew_blocks(long total) { return (int)(lfi_cdiv(total, 256) < 4096 ? lfi_cdiv(total, 256) : 4096); }

}  // namespace

static bool enc_wide_enabled() {
  static int wide = -1;
  if (wide < 0) {
    const char* e = getenv("LFI_ENC_WIDE");
    wide = (e && e[0] == '0') ? 0 : 1;
  }
  return wide != 0;
}
static bool enc_m16_enabled() {   // the forward 64-window kernel on v_mfma_f32_16x16x32_bf16 (read at every call: tests compare)
  const char* e = getenv("LFI_ENC_M16");
  return !(e && e[0] == '0');
}
// the forward kernel with the gate epilogue under the matrix phase (read at every call: tests compare): LFI_ENC_T16 = 0 never,
// 1 (default) where no stash is written (inference, validation, the sampler's static part: 0.46 against 0.51 ms on the p2_face
// shape), 2 with a stash too (training: measured SLOWER there, 0.545 against 0.532 ms alone and +0.15 ms on the whole step - the
// stash stores share the wave's in-order vmcnt queue with the weight stream)
static int enc_t16_mode() {
  const char* e = getenv("LFI_ENC_T16");
  return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1;
}
static bool enc_r64_enabled() {   // (read at every call: tests switch it inside one process)
  const char* e = getenv("LFI_ENC_R64");
  return !(e && e[0] == '0');
}
static size_t enc_bwd_r64_lds(const EncFused& q) {
  const int R2 = 2 * q.R;
  const long imgf = (long)R2 * (q.Kp + 8);
  return (size_t)2 * R2 * (q.Kp + 8) * sizeof(__bf16) +
         (size_t)(ENC_NW * 2 * 32 * ENC_TP > imgf ? ENC_NW * 2 * 32 * ENC_TP : imgf) * sizeof(float) + (size_t)2 * R2 * sizeof(unsigned);
}
static bool enc_wide_bwd_shape_ok(const lfi_enc_desc* d, int lddcond, EncFused* q);
// does lfi_encode_windows_bwd run the two-row-tile kernel (enc_gru_bwd_r64_kernel) for this shape? (shape-only: the pointer
// alignment the row-layout kernels need is REQUIREd there)
static bool enc_bwd_uses_r64(const lfi_enc_desc* d, EncFused* q) {
  if (!enc_r64_enabled() || !enc_wide_bwd_shape_ok(d, d->ldcond, q)) return false;
  return enc_bwd_r64_lds(*q) <= 160 * 1024 && lfi_cdiv((long)d->N * d->B, 2 * q->R) >= 128;
}

static size_t enc_fwd_r64_lds(const EncFused& q, int hist, bool masked) {
  const int R2 = 2 * q.R;
  return (size_t)2 * R2 * (q.Kp + 8) * sizeof(__bf16) + (size_t)ENC_NW * 2 * 32 * ENC_TP * sizeof(float) +
         (size_t)6 * q.Jp * sizeof(float) + (size_t)2 * R2 * sizeof(unsigned) + (masked ? (size_t)R2 * hist * sizeof(float) : 0);
}
static size_t enc_fwd_t16_lds(int hist, bool masked) {
  return (size_t)4 * 64 * 264 * sizeof(__bf16) + (size_t)6 * 256 * sizeof(float) + (size_t)2 * 64 * sizeof(unsigned) +
         (masked ? (size_t)64 * hist * sizeof(float) : 0);
}
// which forward kernel a descriptor takes (lfi_encode_windows_fwd_variant's numbering); fills q for the fused ones
static int enc_fwd_variant(const lfi_enc_desc* d, bool masked, bool stashed, bool aligned, EncFused* q) {
  if (d->lstm || !enc_fused_shape(d->hid, q)) return 0;
  const int hid = d->hid;
  const long F = (long)d->N * d->B;
  const bool vec_ok = d->precision == 1 && hid % 4 == 0 && d->ldcond % 4 == 0 && d->col % 4 == 0 && aligned;
  const int R2 = 2 * q->R;
  const bool take64 = enc_wide_enabled() && enc_r64_enabled() && vec_ok && enc_fwd_r64_lds(*q, d->hist, masked) <= 160 * 1024 &&
                      lfi_cdiv(F, R2) >= 128;
  const bool m16 = take64 && q->Kp == 256 && q->ncg * 64 == q->Jp && enc_m16_enabled();
  if (m16 && hid == 256 && R2 == 64 && enc_fwd_t16_lds(d->hist, masked) <= 160 * 1024 && enc_t16_mode() >= (stashed ? 2 : 1)) return 5;
  if (m16) return 4;
  if (take64) return 3;
  const size_t ldsw = (size_t)2 * q->R * (q->Kp + 8) * sizeof(__bf16) + (size_t)ENC_NW * 32 * ENC_TP * sizeof(float) +
                      (size_t)6 * q->Jp * sizeof(float) + (size_t)2 * q->R * sizeof(unsigned) +
                      (masked ? (size_t)q->R * d->hist * sizeof(float) : 0);
  if (enc_wide_enabled() && vec_ok && ldsw <= 80 * 1024) return 2;
  return 1;
}
extern "C" int lfi_encode_windows_fwd_variant(const lfi_enc_desc* d, int masked, int stashed) {
  EncFused q = {};
  return d ? enc_fwd_variant(d, masked != 0, stashed != 0, true, &q) : 0;
}

extern "C" long lfi_encode_windows_work_floats(const lfi_enc_desc* d) {
  if (!d) return 0;
  const long unfused = (long)d->N * d->B * (d->lstm ? 4 : 3) * d->hid;  // fwd: gh (F x G); bwd: two (LSTM: four) F x hid buffers
  const long fused = 3L * 256 * 256;                    // fused path: zero-padded weight image
  return unfused > fused ? unfused : fused;
}

extern "C" int lfi_encode_windows_fwd(const lfi_enc_desc* d, const float* Xp, const float* whh, const float* b_ih,
                                      const float* b_hh, const float* mask, float* cond, float* gates, float* hseq,
                                      float* work, void* stream) {
  EncArgs a = {};
  int rc = fill_args(d, &a, "lfi_encode_windows_fwd");
  if (rc) return rc;
  LFI_REQUIRE(Xp && whh && b_ih && b_hh && cond && work, "lfi_encode_windows_fwd: null pointer");
  a.Xp = Xp; a.b_ih = b_ih; a.b_hh = b_hh; a.mask = mask; a.cond = cond; a.gates = gates; a.hseq = hseq;
  a.stamps = g_lfi_stamps;
  hipStream_t st = (hipStream_t)stream;
  const int hid = d->hid, F = a.F;
  LFI_REQUIRE(!d->stash_f16 || lfi_encode_windows_stash_f16_ok(d), "lfi_encode_windows_fwd: lfi_enc_desc.stash_f16 is set for a shape "
              "whose gate stash is fp32 (ask lfi_encode_windows_stash_f16_ok)");
  if (d->lstm) {   // "enc: lstm": one GEMM + one gate kernel per history step (in no shipped hparams file: not fused)
    LFI_REQUIRE(gates && hseq, "lfi_encode_windows_fwd: the LSTM encoder keeps its cell state in the gate stash (5 * hid per row)");
    const int blocks = ew_blocks((long)F * hid);
    for (int s = 0; s < d->hist; ++s) {
      if (s > 0) {
        lfi_gemm_desc g = {};
        g.M = F; g.N = 4 * hid; g.K = hid; g.batch = 1;
        g.A = hseq + (long)(s - 1) * F * hid; g.lda = hid; g.a_kcontig = 1;
        g.B = whh; g.ldb = hid; g.b_kcontig = 1;
        g.C = work; g.ldc = 4 * hid; g.precision = d->precision;
        if ((rc = lfi_gemm_f32(&g, stream))) return rc;
      }
      hipLaunchKernelGGL(enc_lstm_gate_fwd_kernel, dim3(blocks), dim3(256), 0, st, a, s, s > 0 ? work : nullptr);
    }
    LFI_LAUNCH_CHECK("lfi_encode_windows_fwd (lstm)");
    return LFI_OK;
  }
  EncFused q = {};
  if (enc_fused_shape(hid, &q)) {
    LFI_REQUIRE(!gates || hseq, "lfi_encode_windows_fwd: the gate stash needs the state stash too");
    // (without a gate stash nothing is kept for a backward pass: hseq is not written either)
    const bool x3 = d->precision == 1;
    // (decided before the weights are converted: the 64-window kernels on the 16 x 16 x 32 shape want their own fragment order)
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool aligned = al16(Xp) && al16(cond) && (!gates || (al16(gates) && al16(hseq)));
    const int R2 = 2 * q.R;
    const int fv = enc_fwd_variant(d, mask != nullptr, gates != nullptr, aligned, &q);
    const bool take64 = fv >= 3, m16 = fv >= 4, t16 = fv == 5;
    const size_t lds64 = enc_fwd_r64_lds(q, d->hist, mask != nullptr), ldst = enc_fwd_t16_lds(d->hist, mask != nullptr);
    if (x3 && m16) hipLaunchKernelGGL(enc_frag_weights16_kernel, dim3(lfi_cdiv(6L * q.Kp * q.Jp, 256)), dim3(256), 0, st, whh, hid, q.Kp,
                                      q.Jp, reinterpret_cast<__bf16*>(work));
    else if (x3) hipLaunchKernelGGL(enc_frag_weights_kernel, dim3(lfi_cdiv(6L * q.Kp * q.Jp, 256)), dim3(256), 0, st, whh, hid, q.Kp,
                                    q.Jp, 1, reinterpret_cast<__bf16*>(work));
    else hipLaunchKernelGGL(enc_pad_weights_kernel, dim3(lfi_cdiv(3L * q.Kp * q.Jp, 256)), dim3(256), 0, st, whh, hid, q.Kp, q.Jp,
                            1, work);
    q.wpad = work;
    q.wfrag = reinterpret_cast<const uint4*>(work);
    const size_t lds = (size_t)q.Kp * (q.R + 1) * sizeof(float) + (x3 ? (size_t)2 * q.R * (q.Kp + 8) * sizeof(__bf16) : 0) +
                       (size_t)3 * q.R * sizeof(unsigned);   // state images + per-row offset tables
    const dim3 grid(lfi_cdiv(F, q.R));
    {
      // row-layout epilogue variant (16-byte accesses): needs 4-float granular rows everywhere it vectorises
      const size_t ldsw = (size_t)2 * q.R * (q.Kp + 8) * sizeof(__bf16) + (size_t)ENC_NW * 32 * ENC_TP * sizeof(float) +
                          (size_t)6 * q.Jp * sizeof(float) + (size_t)2 * q.R * sizeof(unsigned) +
                          (mask ? (size_t)q.R * d->hist * sizeof(float) : 0);
      // two row tiles per wave, one workgroup per CU (enc_gru_fwd_r64_kernel): half the L2 -> CU weight stream per window; taken
      // when its workgroups still cover the chip (LFI_ENC_R64=0 keeps the 32-window kernel)
      if (take64) {
        const dim3 grid64(lfi_cdiv(F, R2));
        rc = LFI_OK;
        switch ((gates && d->stash_f16 ? 4 : 0) | (gates ? 2 : 0) | (mask ? 1 : 0)) {
#define LFI_ENC_FWD64(ST, MK, H)                                                                                   \
  if (t16) {                                                                                                       \
    rc = enc_set_lds(enc_gru_fwd_t16_kernel<ST, MK, H>, ldst);                                                     \
    if (!rc) hipLaunchKernelGGL((enc_gru_fwd_t16_kernel<ST, MK, H>), grid64, dim3(ENC_NT), ldst, st, a, q);        \
    break;                                                                                                         \
  }                                                                                                                \
  if (m16) {                                                                                                       \
    rc = enc_set_lds(enc_gru_fwd_r64_kernel<ST, MK, H, true>, lds64);                                              \
    if (!rc) hipLaunchKernelGGL((enc_gru_fwd_r64_kernel<ST, MK, H, true>), grid64, dim3(ENC_NT), lds64, st, a, q); \
    break;                                                                                                         \
  }                                                                                                                \
  rc = enc_set_lds(enc_gru_fwd_r64_kernel<ST, MK, H>, lds64);                                                      \
  if (!rc) hipLaunchKernelGGL((enc_gru_fwd_r64_kernel<ST, MK, H>), grid64, dim3(ENC_NT), lds64, st, a, q);        \
  break
          case 7: LFI_ENC_FWD64(true, true, true);
          case 6: LFI_ENC_FWD64(true, false, true);
          case 3: LFI_ENC_FWD64(true, true, false);
          case 2: LFI_ENC_FWD64(true, false, false);
          case 1: LFI_ENC_FWD64(false, true, false);
          default: LFI_ENC_FWD64(false, false, false);
#undef LFI_ENC_FWD64
        }
        if (rc) return rc;
        LFI_LAUNCH_CHECK("lfi_encode_windows_fwd (fused, two row tiles per wave)");
        return LFI_OK;
      }
      if (fv == 2) {
        rc = LFI_OK;
        switch ((gates && d->stash_f16 ? 4 : 0) | (gates ? 2 : 0) | (mask ? 1 : 0)) {
#define LFI_ENC_FWDW(ST, MK, H)                                                                                  \
  rc = enc_set_lds(enc_gru_fwd_wide_kernel<ST, MK, H>, ldsw);                                                    \
  if (!rc) hipLaunchKernelGGL((enc_gru_fwd_wide_kernel<ST, MK, H>), grid, dim3(ENC_NT), ldsw, st, a, q);        \
  break
          case 7: LFI_ENC_FWDW(true, true, true);
          case 6: LFI_ENC_FWDW(true, false, true);
          case 3: LFI_ENC_FWDW(true, true, false);
          case 2: LFI_ENC_FWDW(true, false, false);
          case 1: LFI_ENC_FWDW(false, true, false);
          default: LFI_ENC_FWDW(false, false, false);
#undef LFI_ENC_FWDW
        }
        if (rc) return rc;
        LFI_LAUNCH_CHECK("lfi_encode_windows_fwd (fused, row-layout epilogue)");
        return LFI_OK;
      }
    }
    LFI_REQUIRE(!(gates && d->stash_f16), "lfi_encode_windows_fwd: an fp16 gate stash (lfi_enc_desc.stash_f16) needs the row-layout "
                "kernel (lfi_encode_windows_stash_f16_ok) and 16-byte aligned buffers");
    const int variant = (gates ? 4 : 0) | (mask ? 2 : 0) | (x3 ? 1 : 0);
    rc = LFI_OK;
    switch (variant) {
#define LFI_ENC_FWD(ST, MK, X)                                                                                              \
  rc = enc_set_lds(enc_gru_fwd_fused_kernel<ST, MK, X>, lds);                                                               \
  if (!rc) hipLaunchKernelGGL((enc_gru_fwd_fused_kernel<ST, MK, X>), grid, dim3(ENC_NT), lds, st, a, q);                    \
  break
      case 7: LFI_ENC_FWD(true, true, true);
      case 6: LFI_ENC_FWD(true, true, false);
      case 5: LFI_ENC_FWD(true, false, true);
      case 4: LFI_ENC_FWD(true, false, false);
      case 3: LFI_ENC_FWD(false, true, true);
      case 2: LFI_ENC_FWD(false, true, false);
      case 1: LFI_ENC_FWD(false, false, true);
      default: LFI_ENC_FWD(false, false, false);
#undef LFI_ENC_FWD
    }
    if (rc) return rc;
    LFI_LAUNCH_CHECK("lfi_encode_windows_fwd (fused)");
    return LFI_OK;
  }
  LFI_REQUIRE(hseq, "lfi_encode_windows_fwd: hid > 256 needs the state stash hseq");
  const int blocks = ew_blocks((long)F * hid);
  for (int s = 0; s < d->hist; ++s) {
    if (s > 0) {
      lfi_gemm_desc g = {};
      g.M = F; g.N = 3 * hid; g.K = hid; g.batch = 1;
      g.A = hseq + (long)(s - 1) * F * hid; g.lda = hid; g.a_kcontig = 1;
      g.B = whh; g.ldb = hid; g.b_kcontig = 1;  // (h W_hh^T)[w][n] = sum_k h[w][k] W_hh[n][k]
      g.C = work; g.ldc = 3 * hid; g.precision = d->precision;
      if ((rc = lfi_gemm_f32(&g, stream))) return rc;
    }
    hipLaunchKernelGGL(enc_gate_fwd_kernel, dim3(blocks), dim3(256), 0, st, a, s, s > 0 ? work : nullptr);
  }
  LFI_LAUNCH_CHECK("lfi_encode_windows_fwd");
  return LFI_OK;
}

extern "C" long lfi_encode_windows_bias_rows(const lfi_enc_desc* d) {
  EncFused q = {};
  if (!d || d->lstm || !enc_fused_shape(d->hid, &q)) return 0;
  const int rows_per_wg = enc_bwd_uses_r64(d, &q) ? 2 * q.R : q.R;
  return (long)lfi_cdiv((long)d->N * d->B, rows_per_wg) * (ENC_NW / q.ncg);
}

extern "C" int lfi_encode_windows_bias_grads(const float* bias_part, long rows, int hid, float* db_ih, float* db_hh,
                                             void* stream) {
  LFI_REQUIRE(bias_part && db_ih && db_hh && rows > 0 && hid > 0, "lfi_encode_windows_bias_grads: bad arguments");
  hipLaunchKernelGGL(enc_bias_fold_kernel, dim3(lfi_cdiv(4 * hid, 32)), dim3(1024), 0, (hipStream_t)stream, bias_part, rows, hid,
                     db_ih, db_hh);
  LFI_LAUNCH_CHECK("lfi_encode_windows_bias_grads");
  return LFI_OK;
}

// Does lfi_encode_windows_bwd leave dgi / dgh as bf16 arrays (same shapes, half the bytes)? Only the row-layout fused GRU kernel
// in two-product mode does: its consumers - the dW_hh product (A operand rounded to bf16) and the window scatter - then read
// bf16 (lfi_gemm_desc.a_bf16; lfi_encode_windows_scatter looks the same answer up itself).
static bool enc_wide_bwd_shape_ok(const lfi_enc_desc* d, int lddcond, EncFused* q) {
  if (!d || d->lstm || d->precision != 1 || !enc_fused_shape(d->hid, q)) return false;
  const char* e = getenv("LFI_ENC_WIDE_BWD");
  if (e && e[0] == '0') return false;
  const size_t tab = (size_t)2 * q->R * sizeof(unsigned);
  const long imgf = (long)q->R * (q->Kp + 8);
  const size_t ldsw = (size_t)2 * q->R * (q->Kp + 8) * sizeof(__bf16) +
                      (size_t)(ENC_NW * 32 * ENC_TP > imgf ? ENC_NW * 32 * ENC_TP : imgf) * sizeof(float) + tab;
  return d->hid % 4 == 0 && lddcond % 4 == 0 && d->col % 4 == 0 && ldsw <= 80 * 1024;
}
extern "C" int lfi_encode_windows_grad_stash_bf16(const lfi_enc_desc* d) {
  EncFused q = {};
  return (d && d->bwd_two_products && enc_wide_bwd_shape_ok(d, d->ldcond, &q)) ? 1 : 0;
}

// May the gate stash between lfi_encode_windows_fwd and _bwd be fp16 ([hist][F][hid][4] halves instead of [hist][F][4][hid] floats)?
// Only the two row-layout fused GRU kernels read / write that form.
static bool enc_wide_fwd_shape_ok(const lfi_enc_desc* d, EncFused* q, bool mask) {
  if (!d || d->lstm || d->precision != 1 || !enc_fused_shape(d->hid, q)) return false;
  const char* e = getenv("LFI_ENC_WIDE");
  if (e && e[0] == '0') return false;
  const size_t ldsw = (size_t)2 * q->R * (q->Kp + 8) * sizeof(__bf16) + (size_t)ENC_NW * 32 * ENC_TP * sizeof(float) +
                      (size_t)6 * q->Jp * sizeof(float) + (size_t)2 * q->R * sizeof(unsigned) +
                      (mask ? (size_t)q->R * d->hist * sizeof(float) : 0);
  return d->hid % 4 == 0 && d->ldcond % 4 == 0 && d->col % 4 == 0 && ldsw <= 80 * 1024;
}
extern "C" int lfi_encode_windows_stash_f16_ok(const lfi_enc_desc* d) {
  EncFused q = {};
  return (enc_wide_fwd_shape_ok(d, &q, true) && enc_wide_bwd_shape_ok(d, d->ldcond, &q)) ? 1 : 0;
}

extern "C" int lfi_encode_windows_bwd(const lfi_enc_desc* d, const float* dcond, int lddcond, const float* whh,
                                      const float* gates, const float* hseq, float* dgi, float* dgh, float* bias_part,
                                      float* work, void* stream) {
  EncArgs a = {};
  int rc = fill_args(d, &a, "lfi_encode_windows_bwd");
  if (rc) return rc;
  LFI_REQUIRE(dcond && whh && gates && hseq && dgi && dgh && work, "lfi_encode_windows_bwd: null pointer");
  a.dcond = dcond; a.lddcond = lddcond; a.gates = (float*)gates; a.hseq = (float*)hseq; a.dgi = dgi; a.dgh = dgh;
  a.bias_part = bias_part;
  hipStream_t st = (hipStream_t)stream;
  const int hid = d->hid, F = a.F;
  LFI_REQUIRE(!d->stash_f16 || lfi_encode_windows_stash_f16_ok(d), "lfi_encode_windows_bwd: lfi_enc_desc.stash_f16 is set for a shape "
              "whose gate stash is fp32 (ask lfi_encode_windows_stash_f16_ok)");
  if (d->lstm) {
    const int blocks = ew_blocks((long)F * hid);
    float* dhb[2] = {work, work + (long)F * hid};
    float* dcb[2] = {work + 2L * F * hid, work + 3L * F * hid};
    const float *dh_in = nullptr, *dc_in = nullptr;
    for (int s = d->hist - 1; s >= 0; --s) {
      float* dh_out = s > 0 ? dhb[s & 1] : nullptr;
      float* dc_out = s > 0 ? dcb[s & 1] : nullptr;
      hipLaunchKernelGGL(enc_lstm_gate_bwd_kernel, dim3(blocks), dim3(256), 0, st, a, s, dh_in, dc_in, dh_out, dc_out);
      if (s > 0) {
        lfi_gemm_desc g = {};
        g.M = F; g.N = hid; g.K = 4 * hid; g.batch = 1;
        g.A = dgh + (long)s * F * 4 * hid; g.lda = 4 * hid; g.a_kcontig = 1;
        g.B = whh; g.ldb = hid; g.b_kcontig = 0;
        g.C = dh_out; g.ldc = hid; g.accumulate = 1; g.precision = d->precision;
        if ((rc = lfi_gemm_f32(&g, stream))) return rc;
      }
      dh_in = dh_out; dc_in = dc_out;
    }
    LFI_LAUNCH_CHECK("lfi_encode_windows_bwd (lstm)");
    return LFI_OK;
  }
  EncFused q = {};
  if (enc_fused_shape(hid, &q)) {
    const size_t tab = (size_t)2 * q.R * sizeof(unsigned);   // per-row offset / liveness tables
    const size_t ldsf = (size_t)q.Kp * (q.R + 1) * sizeof(float) + tab;
    static int widebw = -1;
    if (widebw < 0) {
      const char* e = getenv("LFI_ENC_WIDE_BWD");
      widebw = (e && e[0] == '0') ? 0 : 1;
    }
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const long imgf = (long)q.R * (q.Kp + 8);   // floats of one image pair
    const size_t ldsw = (size_t)2 * q.R * (q.Kp + 8) * sizeof(__bf16) +
                        (size_t)(ENC_NW * 32 * ENC_TP > imgf ? ENC_NW * 32 * ENC_TP : imgf) * sizeof(float) + tab;
    const bool want16 = d->bwd_two_products && lddcond == d->ldcond && lfi_encode_windows_grad_stash_bf16(d);
    const bool wide_ok = d->precision == 1 && widebw && hid % 4 == 0 && lddcond % 4 == 0 && d->col % 4 == 0 && al16(dcond) && al16(gates) &&
                         al16(hseq) && al16(dgi) && al16(dgh) && (!bias_part || al16(bias_part)) && ldsw <= 80 * 1024;
    // bf16x3 recurrence: the row-layout kernels only. What they do not take (LFI_ENC_WIDE_BWD=0, hid not a multiple of 4, unaligned
    // buffers) runs the exact-f32 accumulator-layout kernel, whatever the engine's GEMM mode (see enc_gru_bwd_fused_kernel)
    const bool x3 = wide_ok;
    if (x3) hipLaunchKernelGGL(enc_frag_weights_kernel, dim3(lfi_cdiv(6L * q.Kp * q.Jp, 256)), dim3(256), 0, st, whh, hid, q.Kp,
                               q.Jp, 0, reinterpret_cast<__bf16*>(work));
    else hipLaunchKernelGGL(enc_pad_weights_kernel, dim3(lfi_cdiv(3L * q.Kp * q.Jp, 256)), dim3(256), 0, st, whh, hid, q.Kp, q.Jp,
                            0, work);
    q.wpad = work;
    q.wfrag = reinterpret_cast<const uint4*>(work);
    LFI_REQUIRE(!want16 || wide_ok, "lfi_encode_windows_bwd: lfi_encode_windows_grad_stash_bf16 promised a bf16 gradient stash but the "
                "buffers are not 16-byte aligned");
    LFI_REQUIRE(!d->stash_f16 || wide_ok, "lfi_encode_windows_bwd: an fp16 gate stash (lfi_enc_desc.stash_f16) needs the row-layout "
                "kernel (lfi_encode_windows_stash_f16_ok) and 16-byte aligned buffers");
    EncFused q64 = {};
    if (wide_ok && lddcond == d->ldcond && enc_bwd_uses_r64(d, &q64)) {   // two row tiles per wave, one workgroup per CU
      const size_t lds64 = enc_bwd_r64_lds(q);
      const dim3 grid64(lfi_cdiv(F, 2 * q.R));
      rc = LFI_OK;
      switch ((want16 ? 2 : 0) | (d->stash_f16 ? 1 : 0)) {
#define LFI_ENC_BWD64(A2, H)                                                                                   \
  rc = enc_set_lds(enc_gru_bwd_r64_kernel<A2, H>, lds64);                                                      \
  if (!rc) hipLaunchKernelGGL((enc_gru_bwd_r64_kernel<A2, H>), grid64, dim3(ENC_NT), lds64, st, a, q);        \
  break
        case 3: LFI_ENC_BWD64(true, true);
        case 2: LFI_ENC_BWD64(true, false);
        case 1: LFI_ENC_BWD64(false, true);
        default: LFI_ENC_BWD64(false, false);
#undef LFI_ENC_BWD64
      }
      if (rc) return rc;
    } else if (wide_ok) {
      rc = LFI_OK;
      switch ((want16 ? 2 : 0) | (d->stash_f16 ? 1 : 0)) {
#define LFI_ENC_BWDW(A2, H)                                                                                             \
  rc = enc_set_lds(enc_gru_bwd_wide_kernel<A2, H>, ldsw);                                                               \
  if (!rc) hipLaunchKernelGGL((enc_gru_bwd_wide_kernel<A2, H>), dim3(lfi_cdiv(F, q.R)), dim3(ENC_NT), ldsw, st, a, q);  \
  break
        case 3: LFI_ENC_BWDW(true, true);
        case 2: LFI_ENC_BWDW(true, false);
        case 1: LFI_ENC_BWDW(false, true);
        default: LFI_ENC_BWDW(false, false);
#undef LFI_ENC_BWDW
      }
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(enc_gru_bwd_fused_kernel, dim3(lfi_cdiv(F, q.R)), dim3(ENC_NT), ldsf, st, a, q);
    }
    LFI_LAUNCH_CHECK("lfi_encode_windows_bwd (fused)");
    return LFI_OK;
  }
  const int blocks = ew_blocks((long)F * hid);
  float* buf[2] = {work, work + (long)F * hid};
  const float* dh_in = nullptr;
  for (int s = d->hist - 1; s >= 0; --s) {
    float* dh_out = s > 0 ? buf[s & 1] : nullptr;
    hipLaunchKernelGGL(enc_gate_bwd_kernel, dim3(blocks), dim3(256), 0, st, a, s, dh_in, dh_out);
    if (s > 0) {
      lfi_gemm_desc g = {};
      g.M = F; g.N = hid; g.K = 3 * hid; g.batch = 1;
      g.A = dgh + (long)s * F * 3 * hid; g.lda = 3 * hid; g.a_kcontig = 1;
      g.B = whh; g.ldb = hid; g.b_kcontig = 0;  // (dgh W_hh)[w][j] = sum_k dgh[w][k] W_hh[k][j]
      g.C = dh_out; g.ldc = hid; g.accumulate = 1; g.precision = d->precision;
      if ((rc = lfi_gemm_f32(&g, stream))) return rc;
    }
    dh_in = dh_out;
  }
  LFI_LAUNCH_CHECK("lfi_encode_windows_bwd");
  return LFI_OK;
}

extern "C" int lfi_encode_windows_compact_dgi(const lfi_enc_desc* d) {
  EncFused q = {};
  return (d && !d->lstm && enc_fused_shape(d->hid, &q)) ? 1 : 0;
}

extern "C" int lfi_encode_windows_scatter(const lfi_enc_desc* d, const float* dgi, const float* dgh, const float* mask, float* dXp,
                                          void* stream) {
  EncArgs a = {};
  int rc = fill_args(d, &a, "lfi_encode_windows_scatter");
  if (rc) return rc;
  LFI_REQUIRE(dgi && dXp, "lfi_encode_windows_scatter: null pointer");
  a.compact = lfi_encode_windows_compact_dgi(d);
  LFI_REQUIRE(!a.compact || dgh, "lfi_encode_windows_scatter: the compact dgi of the fused backward needs dgh too");
  a.dgi = (float*)dgi; a.dgh = (float*)dgh; a.mask = mask;
  a.g16 = lfi_encode_windows_grad_stash_bf16(d);
  // bf16 compact stash (the training step's): 16-byte loads, several frame rows per workgroup (LFI_ENC_SCATTER16=0: the kernel above)
  const char* ev = getenv("LFI_ENC_SCATTER16");
  const int cpr = 3 * d->hid / 8;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!(ev && ev[0] == '0') && a.g16 && a.compact && d->hid % 8 == 0 && cpr <= 256 && al16(dgi) && al16(dgh) && al16(dXp)) {
    const int rows = d->B * d->T, rpw = 256 / cpr;
    hipLaunchKernelGGL(enc_scatter16_kernel, dim3(lfi_cdiv(rows, rpw)), dim3(256), 0, (hipStream_t)stream, a, dXp, rows, cpr, rpw);
  } else
    hipLaunchKernelGGL(enc_scatter_kernel, dim3(d->B * d->T), dim3(256), 0, (hipStream_t)stream, a, dXp);
  LFI_LAUNCH_CHECK("lfi_encode_windows_scatter");
  return LFI_OK;
}

extern "C" int lfi_gather_windows(const float* X, int B, int T, int dim, int N, int start, int hist, int incl,
                                  const float* mask, float* cond, int ldcond, int col, void* stream) {
  LFI_REQUIRE(X && cond, "lfi_gather_windows: null pointer");
  LFI_REQUIRE(B > 0 && T > 0 && dim > 0 && N > 0 && hist > 0 && (incl == 0 || incl == 1), "lfi_gather_windows: bad dims");
  LFI_REQUIRE(start - hist + incl >= 0 && start + N - 1 + incl <= T, "lfi_gather_windows: window out of range");
  hipLaunchKernelGGL(gather_windows_kernel, dim3(N * B), dim3(256), 0, (hipStream_t)stream, X, B, T, dim, N, start, hist,
                     incl, mask, cond, ldcond, col);
  LFI_LAUNCH_CHECK("lfi_gather_windows");
  return LFI_OK;
}

extern "C" int lfi_fill_frame_nb(const float* base, float offset, int B, int N, float* cond, int ldcond, int col, void* stream) {
  LFI_REQUIRE(cond && B > 0 && N > 0 && col >= 0 && col < ldcond, "lfi_fill_frame_nb: bad arguments");
  hipLaunchKernelGGL(fill_frame_nb_kernel, dim3(ew_blocks((long)N * B)), dim3(256), 0, (hipStream_t)stream, base, offset, B,
                     (long)N * B, cond, ldcond, col);
  LFI_LAUNCH_CHECK("lfi_fill_frame_nb");
  return LFI_OK;
}

extern "C" int lfi_leaky_grad(float* d, long ldd, const float* y, long ldy, int rows, int cols, float slope, void* stream) {
  LFI_REQUIRE(d && y && rows >= 0 && cols >= 0, "lfi_leaky_grad: bad arguments");
  if (rows == 0 || cols == 0) return LFI_OK;
  hipLaunchKernelGGL(leaky_grad_kernel, dim3(ew_blocks((long)rows * cols)), dim3(256), 0, (hipStream_t)stream, d, ldd, y, ldy,
                     rows, cols, slope);
  LFI_LAUNCH_CHECK("lfi_leaky_grad");
  return LFI_OK;
}
