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

namespace {

struct EncArgs {
  int B, T, N, start, hist, hid, ldcond, col, dup;
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
};

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
    const float nn = tanhf(mk * xp[2 * hid + j] + a.b_ih[2 * hid + j] + rr * ghn);
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

// dXp[b*T + p][c] = sum_{s} mask[w(n,b)][s] * dgi[s][w][c],  n = p - pos0 - s in [0, N)
__global__ __launch_bounds__(256) void enc_scatter_kernel(EncArgs a, float* __restrict__ dXp) {
  const int G3 = 3 * a.hid;
  const int row = blockIdx.x;  // b*T + p
  const int b = row / a.T, p = row - b * a.T;
  const int pos0 = a.start - a.hist + 1;
  for (int c = threadIdx.x; c < G3; c += 256) {
    float acc = 0.0f;
    for (int s = 0; s < a.hist; ++s) {
      const int n = p - pos0 - s;
      if (n < 0 || n >= a.N) continue;
      const long w = (long)n * a.B + b;
      const float mk = a.mask ? a.mask[w * a.hist + s] : 1.0f;
      acc += mk * a.dgi[((long)s * a.F + w) * G3 + c];
    }
    dXp[(long)row * G3 + c] = acc;
  }
}

__global__ __launch_bounds__(256) void gather_windows_kernel(const float* __restrict__ X, int B, int T, int dim, int N, int start,
                                                             int hist, int incl, float* __restrict__ cond, int ldcond, int col) {
  const int f = blockIdx.x;  // n*B + b
  const int n = f / B, b = f - n * B;
  const int t0 = start + n - hist + incl;
  const int tot = hist * dim;
  // window rows are consecutive frames of one sample: one contiguous run of hist*dim floats
  const float* src = X + ((long)b * T + t0) * dim;
  float* dst = cond + (long)f * ldcond + col;
  for (int i = threadIdx.x; i < tot; i += 256) dst[i] = src[i];
}

int fill_args(const lfi_enc_desc* d, EncArgs* a, const char* who) {
  LFI_REQUIRE(d, "%s: null descriptor", who);
  LFI_REQUIRE(d->B > 0 && d->T > 0 && d->N > 0 && d->hist > 0 && d->hid > 0, "%s: bad dims", who);
  LFI_REQUIRE(d->start + d->N <= d->T, "%s: start + N > T", who);
  LFI_REQUIRE(d->hist <= d->start + 1, "%s: window longer than start+1", who);
  a->B = d->B; a->T = d->T; a->N = d->N; a->start = d->start; a->hist = d->hist; a->hid = d->hid;
  a->ldcond = d->ldcond; a->col = d->col; a->dup = d->dup;
  a->F = d->N * d->B;
  return LFI_OK;
}

int ew_blocks(long total) { return (int)(lfi_cdiv(total, 256) < 4096 ? lfi_cdiv(total, 256) : 4096); }

}  // namespace

extern "C" long lfi_encode_windows_work_floats(const lfi_enc_desc* d) {
  if (!d) return 0;
  return (long)d->N * d->B * 3 * d->hid;  // fwd: gh (F x 3hid); bwd: two F x hid gradient buffers
}

extern "C" int lfi_encode_windows_fwd(const lfi_enc_desc* d, const float* Xp, const float* whh, const float* b_ih,
                                      const float* b_hh, const float* mask, float* cond, float* gates, float* hseq,
                                      float* work, void* stream) {
  EncArgs a = {};
  int rc = fill_args(d, &a, "lfi_encode_windows_fwd");
  if (rc) return rc;
  LFI_REQUIRE(Xp && whh && b_ih && b_hh && cond && hseq && work, "lfi_encode_windows_fwd: null pointer");
  a.Xp = Xp; a.b_ih = b_ih; a.b_hh = b_hh; a.mask = mask; a.cond = cond; a.gates = gates; a.hseq = hseq;
  hipStream_t st = (hipStream_t)stream;
  const int hid = d->hid, F = a.F;
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

extern "C" int lfi_encode_windows_bwd(const lfi_enc_desc* d, const float* dcond, int lddcond, const float* whh,
                                      const float* gates, const float* hseq, float* dgi, float* dgh, float* work,
                                      void* stream) {
  EncArgs a = {};
  int rc = fill_args(d, &a, "lfi_encode_windows_bwd");
  if (rc) return rc;
  LFI_REQUIRE(dcond && whh && gates && hseq && dgi && dgh && work, "lfi_encode_windows_bwd: null pointer");
  a.dcond = dcond; a.lddcond = lddcond; a.gates = (float*)gates; a.hseq = (float*)hseq; a.dgi = dgi; a.dgh = dgh;
  hipStream_t st = (hipStream_t)stream;
  const int hid = d->hid, F = a.F;
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

extern "C" int lfi_encode_windows_scatter(const lfi_enc_desc* d, const float* dgi, const float* mask, float* dXp,
                                          void* stream) {
  EncArgs a = {};
  int rc = fill_args(d, &a, "lfi_encode_windows_scatter");
  if (rc) return rc;
  LFI_REQUIRE(dgi && dXp, "lfi_encode_windows_scatter: null pointer");
  a.dgi = (float*)dgi; a.mask = mask;
  hipLaunchKernelGGL(enc_scatter_kernel, dim3(d->B * d->T), dim3(256), 0, (hipStream_t)stream, a, dXp);
  LFI_LAUNCH_CHECK("lfi_encode_windows_scatter");
  return LFI_OK;
}

extern "C" int lfi_gather_windows(const float* X, int B, int T, int dim, int N, int start, int hist, int incl, float* cond,
                                  int ldcond, int col, void* stream) {
  LFI_REQUIRE(X && cond, "lfi_gather_windows: null pointer");
  LFI_REQUIRE(B > 0 && T > 0 && dim > 0 && N > 0 && hist > 0 && (incl == 0 || incl == 1), "lfi_gather_windows: bad dims");
  LFI_REQUIRE(start - hist + incl >= 0 && start + N - 1 + incl <= T, "lfi_gather_windows: window out of range");
  hipLaunchKernelGGL(gather_windows_kernel, dim3(N * B), dim3(256), 0, (hipStream_t)stream, X, B, T, dim, N, start, hist,
                     incl, cond, ldcond, col);
  LFI_LAUNCH_CHECK("lfi_gather_windows");
  return LFI_OK;
}
