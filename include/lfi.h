/*
 * lfi.h — C ABI of liblfi_hip.so: the MI355X (gfx950) kernels behind the conditional-Glow hot path of
 * jonepatr/lets_face_it.
 *
 * The reference has no FFI of its own for this path (it is pure PyTorch: SURVEY.md §8b); the Python object
 * surface it exposes (SeqGlow.forward / inference / invert, LetsFaceItGlow.training_step /
 * configure_optimizers) is mirrored by lets_face_it_amd/glow/, and THIS header is what that host code binds
 * with ctypes. Each entry point names the reference code it replaces (paths relative to
 * /root/reference/code/glow_pytorch/).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer to fp32 unless noted.
 *  - The caller owns every buffer; the library allocates nothing and keeps no global mutable state in normal operation (per
 *    thread: the last error message; process-wide and null unless a diagnostic tool sets it: lfi_debug_set_stamps's pointer).
 *  - `stream` is a hipStream_t passed as void* (torch's current stream); all work is enqueued on it, no host
 *    synchronisation, so every call is hipGraph-capturable.
 *  - Return value: 0 = ok, < 0 = error; the message is in lfi_last_error() (thread-local).
 *  - Frames are time-major: frame f = n * B + b for timestep n (0 .. N-1) and sample b.
 */
#ifndef LFI_H
#define LFI_H

#ifdef __cplusplus
extern "C" {
#endif

#define LFI_OK 0
#define LFI_ERR_ARG (-1)
#define LFI_ERR_LAUNCH (-2)
#define LFI_ERR_UNSUPPORTED (-3)

const char* lfi_last_error(void);
int lfi_version(void);

/* ---------------------------------------------------------------- generic fp32 GEMM on MFMA
 * C[b] (+)= act( opA(A[b]) * opB(B[b]) + bias[b] ) ,  b = 0 .. batch-1
 *   a_kcontig = 1: A element (m,k) at A[m*lda + k];  0: at A[k*lda + m]
 *   b_kcontig = 1: B element (k,n) at B[n*ldb + k];  0: at B[k*ldb + n]
 * act: 0 none | 1 leaky_relu(slope) | 2 multiply by leaky_relu'(G) (G same shape/ld as C: G > 0 ? 1 : slope)
 * splitk > 1 needs `work` of lfi_gemm_work_floats() floats; partial sums are reduced deterministically.
 * Replaces the aten::addmm / aten::mm calls of nn.Linear / nn.GRU input projections and their autograd
 * (glow/models.py:63,187-190; glow/modules.py:93-95,186).
 */
typedef struct {
  int M, N, K;
  const float* A; long lda; int a_kcontig;
  const float* B; long ldb; int b_kcontig;
  float* C; long ldc;
  const float* bias;          /* per column n, or NULL */
  const float* G; long ldg;   /* act == 2 only */
  int batch; long strideA, strideB, strideC, strideBias, strideG;
  int accumulate;             /* 1: C = act(..) + C ; 2: C = act(.. + C) (pre-activation add) */
  int act; float slope;
  int splitk; float* work;    /* splitk 0 = let the library pick the split (needs work of lfi_gemm_work_floats floats; a NULL
                                 work then means no split) */
  int precision;              /* 0: exact fp32 (f32-input MFMA). 1: bf16x3 — operands split into bf16 hi + lo on the fly,
                                 three bf16 MFMAs per step into fp32 accumulators (~2^-16 relative); needs 16-byte aligned
                                 operands, otherwise the exact kernel runs. Tests may OR in 0x10 (force the 256 x 256 tile
                                 kernel) or 0x20 (force 128 x 128) instead of the library's own choice. Bit 2 (1 | 4 = 5): SIX bf16
                                 products of three-piece operands (x = p0 + p1 + p2, 24 mantissa bits; what is dropped is 2^-24 relative:
                                 fp32-grade results at 6/16 of the f32-input MFMA's cost; 128 x 128 tiles; the sampler's per-frame
                                 products). Bit 3 (1 | 8 = 9): three products of FP16 pieces (11 + 11 mantissa bits, 2^-22 relative: fp32-grade at the
                                 three-product cost) - for operands inside fp16's range only (a value beyond 65504 turns into NaN):
                                 activations and weights, never gradients; 128 x 128 tiles. Bits 8 / 9 (0x100 / 0x200) DROP the
                                 a_lo * b_hi / a_hi * b_lo product of the bf16x3 kernels (a measurement switch: what each GEMM
                                 class loses with one or two bf16 passes, tools/precision_sweep.py -> profiles/precision_sweep.md;
                                 never set by the engine's default configuration) */
  int a_bf16;                 /* 1: A points at bf16 values (same lda / strides, in elements) that stand for an operand already rounded
                                 to bf16 - the encoders' bf16 gradient stash. Needs precision bit 8 (two products: a_lo is never
                                 used), a_kcontig = b_kcontig = 0 and 8-byte aligned rows; runs the 256 x 256 bf16x3 kernel */
  float* colsum_part; long ld_part;  /* optional: the epilogue also leaves column sums of the STORED result over each block of
                                 rows it handles, row r of a (lfi_gemm_colpart_rows(d) x ld_part) matrix, columns as in C (batch
                                 entries side by side: strideC * batch <= ldc). Summing its rows (lfi_colsum_f32) gives the bias
                                 gradient of a Linear without a second pass over C (autograd of nn.Linear.bias,
                                 glow/models.py:187-190). NULL: off. lfi_gemm_colpart_rows returns 0 when the product would not
                                 run on a kernel that can do it (then leave colsum_part NULL) */
} lfi_gemm_desc;

long lfi_gemm_work_floats(const lfi_gemm_desc* d);
long lfi_gemm_colpart_rows(const lfi_gemm_desc* d);
int lfi_gemm_f32(const lfi_gemm_desc* d, void* stream);

/* ---- bf16x3 GEMM on PRE-SPLIT operands. lfi_gemm_f32's bf16x3 kernels split every fp32 operand element into bf16 hi + lo
 * inside every workgroup that touches it; for the big products of the step (cond_transform of all flow steps and its
 * autograd, glow/models.py:187-190; the hoisted W_ih[:, Ch:] c product of the coupling cell and its autograd, :176-179,
 * 206-208) the split is done ONCE instead - by lfi_planes_from_f32 for operands that exist as fp32, by the producing kernel's
 * epilogue (or the backward walk) for operands that are themselves results - into "planes" of the fp32 matrix X (rows x cols):
 * per 32-row tile rt and 16-column tile ct two 1-KB blocks (hi, lo), block ((rt * nct + ct) * 2 + plane) * 512 bf16, 16-byte
 * aligned, rows zero padded to whole 256-row panels and columns to whole tiles (lfi_planes_elems(rows, cols) bf16 elements),
 * which the kernel moves global -> LDS by LDS-DMA (block layout and both reads: lets_face_it_amd/csrc/lfi_pgemm.hip).
 * An operand takes the planes of X in one of two uses:
 *   use 0, row use: the operand is X (mn x k), k = X's columns: an nn.Linear input or weight in its forward product;
 *   use 1, transposed use: the operand is X^T, k = X's ROWS: the same matrices in the autograd products that sum over frames or
 *     over output units. The operand pointer must then address a whole 32-row tile (and a column tile that is a multiple of 2).
 * C[b] (+)= act(A[b] B[b]^T + bias[b]) with A: M x K, B: N x K. a_nkt / b_nkt = column tiles per row tile (nct) of the plane
 * buffers. Same products and accumulation order as lfi_gemm_f32's bf16x3 kernels (results bit-identical). A workgroup reads
 * whole 128- / 256-wide mn panels from a batch entry's first tile: when batch entries start inside one buffer, keep that
 * buffer one panel (256 rows) longer than lfi_planes_elems says (what is read there only feeds rows / columns past M / N,
 * which are never stored). */
long lfi_planes_elems(long rows, int cols);
int lfi_planes_from_f32(const float* X, long ldx, long rows, int cols, void* planes, void* stream);
typedef struct {
  int M, N, K;
  const void* Ap; int a_nkt; long a_stride;   /* planes of A; k-tiles per mn tile in that buffer; bf16 elements between batch
                                                 entries (a batch entry may be a k-tile range of one buffer: stride kt0 * 1024) */
  const void* Bp; int b_nkt; long b_stride;
  float* C; long ldc;
  const float* bias; const float* G; long ldg;   /* as lfi_gemm_desc */
  int batch; long strideC, strideBias, strideG;
  int accumulate, act; float slope;
  int skip;                                      /* bit 0 drops a_lo * b_hi, bit 1 a_hi * b_lo (as lfi_gemm_desc.precision bits 8 / 9) */
  int a_fmt, b_fmt;                              /* use of either operand's planes: 0 row use, 1 transposed use */
  int splitk; float* work;                       /* K split over grid.z with a deterministic reduce, as lfi_gemm_desc (0 / 1: none);
                                                    work: lfi_gemm_planes_work_floats floats */
  int store_f32;                                 /* 1: the result goes to C as fp32 rows; 0: plane outputs only (C may be NULL) */
  /* The result as operand planes of the products that consume it, in either use (splitk <= 1; batch entries side by side in C's
   * columns): planes of the matrix whose rows are C's rows and whose columns are cr_col0 + b * strideC + C's columns (cr_nkt column
   * tiles per row tile in that buffer). Rows >= M and columns >= N are written as zeros. NULL: off. */
  void* Cr; int cr_nkt; long cr_col0;
  /* act == 2 (multiply by leaky_relu'(G)): G may be given as the planes another lfi_gemm_planes call emitted (only the sign of
   * the hi plane is used), same indexing as Cr. NULL: the fp32 G above. */
  const void* Gr; int gr_nkt; long gr_col0;
  float* colsum_part; long ld_part;              /* as lfi_gemm_desc: lfi_gemm_planes_colpart_rows(d) rows of per-pass column sums */
  int out_hi_only;                               /* plane outputs: write the hi planes only - enough for a consumer that takes them
                                                    as its A operand with skip bit 0 (two products, A rounded to bf16), which never
                                                    fetches A's lo planes */
  int tile;                                      /* 0: the library picks 128 x 256 or 256 x 128 tiles by N; 1 / 2 pin them (tests) */
} lfi_pgemm_desc;
long lfi_gemm_planes_work_floats(const lfi_pgemm_desc* d);
long lfi_gemm_planes_colpart_rows(const lfi_pgemm_desc* d);
int lfi_gemm_planes(const lfi_pgemm_desc* d, void* stream);

/* out[c] (+)= scale * sum_r X[r*ldx + c]  for r < rows, c < cols; batched. Deterministic two-stage reduction.
 * (bias gradients: autograd of the nn.Linear / GRU biases.) work: lfi_colsum_work_floats floats. */
long lfi_colsum_work_floats(int rows, int cols, int batch);
int lfi_colsum_f32(const float* X, long ldx, long strideX, int rows, int cols, int batch,
                   float* out, long strideOut, float scale, int accumulate, float* work, void* stream);

/* Column folding. The reference's GRU encoders emit cat(seq[:, -1], h_n[0]) — the same vector twice (glow/models.py:63-64)
 * — so cond_transform multiplies two weight column blocks by identical inputs. The engine stores each block once and
 * folds the weights instead: dst[r][j] = src[r][a[j]] (+ src[r][b[j]] if b[j] >= 0), j < ncols. With b = NULL the same
 * call is a plain column gather, which is how the folded weight gradient is expanded back (both copies get it). */
int lfi_cols_fold(const float* src, long lds, long rows, const int* a, const int* b, int ncols, float* dst, long ldd,
                  void* stream);

/* ---------------------------------------------------------------- window encoders (ModalityEncoder, glow/models.py:55-80)
 * One modality: single-layer GRU from h0 = 0 over a `hist`-frame window ending at frame t (inclusive) for every
 * (sample, timestep) pair; output cat(seq[:, -1], h_n[0]) written into columns [col, col + 2*hid) of the
 * feature matrix `cond` (F x ldcond). Window w = n*B + b covers rows b*T + (start + n - hist + 1 + s), s < hist,
 * of the pre-projected input Xp = X @ W_ih^T (B*T x 3*hid, no bias; made with lfi_gemm_f32).
 * mask: optional (F x hist) dropout multipliers (glow/models.py:56-58); NULL in eval mode.
 * gates ([hist][F][4*hid]: r, z, n, W_hn h + b_hn; NULL to skip) and hseq ([hist][F][hid], required: it is the
 * recurrent state) are step-major stashes for the backward pass. work: lfi_encode_windows_work_floats floats.
 */
typedef struct {
  int B, T, N, start;   /* batch, sequence length, timesteps (T - start), first modelled frame */
  int hist, hid;        /* window length, hidden size */
  int ldcond, col;      /* leading dimension of cond and first output column */
  int precision;        /* lfi_gemm_desc.precision of the per-step recurrent GEMMs */
  int dup;              /* 1: write the state twice, columns [col, col+hid) and [col+hid, col+2*hid) as the reference's
                           cat(seq[:, -1], h_n[0]) does; 0: once (folded feature layout, see lfi_cols_fold) */
  int lstm;             /* 1: "enc: lstm" (nn.LSTM from zero (h, c), gate blocks i, f, g, o; glow/models.py:27-33,65-69): every
                           "3*hid" below reads 4*hid, the gate stash row is 5*hid (i, f, g, o, c) and is required, dgi = dgh */
  int bwd_two_products; /* lfi_encode_windows_bwd, bf16x3 fused GRU path: 1 = two bf16 products per k-step in the d gates x W_hh
                           recurrence (d gates rounded to bf16), as lfi_gemm_desc.precision bit 8 does for a GEMM; 0 = three */
  int stash_f16;        /* 1: `gates` is an fp16 array, [hist][F][hid][4] halves (r, z, n, W_hn h + b_hn of a hidden unit side by
                           side; half the bytes of the fp32 form) - written by lfi_encode_windows_fwd, read by _bwd; only where
                           lfi_encode_windows_stash_f16_ok(d) (the row-layout fused GRU kernels), both calls with the same value.
                           The state stash hseq stays fp32: it is a GEMM operand of dW_hh (glow/models.py:63, nn.GRU's BPTT) */
} lfi_enc_desc;

long lfi_encode_windows_work_floats(const lfi_enc_desc* d);
int lfi_encode_windows_fwd(const lfi_enc_desc* d, const float* Xp, const float* whh /* 3hid x hid */,
                           const float* b_ih, const float* b_hh, const float* mask,
                           float* cond, float* gates, float* hseq, float* work, void* stream);
/* BPTT of the above. dcond (F x lddcond): gradient of the feature matrix; the two output halves are summed.
 * Writes dgi ([hist][F][3*hid], or compact - see lfi_encode_windows_scatter: grads of the pre-activations on the input side,
 * before the dropout mask) and
 * dgh ([hist][F][3*hid]: on the hidden side). Weight gradients follow from those with lfi_gemm_f32/lfi_colsum:
 * dW_hh = dgh[1:]^T hseq[:-1], db_hh = colsum(dgh), db_ih = colsum(dgi), dW_ih = scatter(dgi)^T X. */
int lfi_encode_windows_bwd(const lfi_enc_desc* d, const float* dcond, int lddcond, const float* whh /* 3hid x hid */,
                           const float* gates, const float* hseq, float* dgi, float* dgh,
                           float* bias_part /* [lfi_encode_windows_bias_rows][4][hid] or NULL */, float* work, void* stream);
/* The fused backward (hid <= 256) also leaves per-workgroup partial sums over windows and steps of the four distinct
 * pre-activation gradients (d r, d z, d n, d n * r) in bias_part: db_ih = colsum of blocks {0, 1, 2}, db_hh = colsum of
 * blocks {0, 1, 3} - no second pass over dgi / dgh. Returns the number of partial rows, 0 when the unfused path runs
 * (then bias_part is not written and the biases follow from lfi_colsum_f32 over dgi / dgh). */
long lfi_encode_windows_bias_rows(const lfi_enc_desc* d);
/* Folds bias_part ([rows][4][hid]) into both bias gradients in one launch: db_ih ([3hid]) = column sums of blocks {0, 1, 2},
 * db_hh ([3hid]) = those of blocks {0, 1, 3}; rows are added in a fixed order (bit-reproducible). */
int lfi_encode_windows_bias_grads(const float* bias_part, long rows, int hid, float* db_ih, float* db_hh, void* stream);
/* dXp[b*T + p] = sum over the windows (n, s) that read row p of mask * dgi[(n*B+b)*hist + s]   (B*T x 3*hid)
 * Compact dgi: the input-side and hidden-side gate derivatives of a GRU differ only in the n block (d n vs d n * r), so the
 * fused backward (lfi_encode_windows_compact_dgi(d) == 1: hid <= 256, not lstm) writes dgi as that block alone,
 * [hist][F][hid], and the scatter takes the r and z blocks from dgh (then required; db_ih comes from bias_part). */
int lfi_encode_windows_compact_dgi(const lfi_enc_desc* d);
/* 1 when lfi_encode_windows_bwd (d->bwd_two_products set, fused GRU path, d->ldcond = lddcond) writes dgi / dgh as bf16 arrays of
 * the same shapes: the dW_hh product then takes dgh with lfi_gemm_desc.a_bf16 and lfi_encode_windows_scatter reads bf16. */
int lfi_encode_windows_grad_stash_bf16(const lfi_enc_desc* d);
/* 1 when the shape runs on the row-layout fused GRU kernels in both directions, i.e. lfi_enc_desc.stash_f16 may be set. */
int lfi_encode_windows_stash_f16_ok(const lfi_enc_desc* d);
/* Which forward kernel lfi_encode_windows_fwd takes for this descriptor with 16-byte aligned buffers (tests and the bench's kernel
 * name; shape and environment switches only): 0 = one GEMM + gate kernel per history step, 1 = fused accumulator-layout kernel,
 * 2 = 32-window row-layout kernel, 3 = 64-window kernel (32 x 32 x 16), 4 = 64-window kernel (16 x 16 x 32), 5 = 64-window
 * kernel with the gate epilogue under the matrix phase (transposed product, hid = 256). (glow/models.py:55-80) */
int lfi_encode_windows_fwd_variant(const lfi_enc_desc* d, int masked, int stashed);
int lfi_encode_windows_scatter(const lfi_enc_desc* d, const float* dgi, const float* dgh, const float* mask, float* dXp,
                               void* stream);
/* "enc: none" modality (glow/models.py:76-77), the flattened p1_face history (glow/models.py:601-603) and the input of an
 * "enc: mlp" modality (glow/models.py:70-71):
 * cond[f, col + s*dim + c] = mask[f, s] * X[b, start + n - hist + s + incl, c]   (incl = 0 for prev_p1_face, 1 otherwise;
 * mask (F x hist) = the dropout multipliers of glow/models.py:56-58, or NULL) */
int lfi_gather_windows(const float* X, int B, int T, int dim, int N, int start, int hist, int incl, const float* mask,
                       float* cond, int ldcond, int col, void* stream);
/* Zero-padded copy of a (rows x cols) matrix with row pitch lds into row pitch ldd >= cols (the padding columns are written as
 * zeros): the engine's 4-float granular copies of 50-d face / 27-d speech inputs and of W_ih, so that x W_ih^T (nn.GRU's input
 * projection, glow/models.py:63) and its weight gradient take the vector-load GEMM kernels. */
int lfi_pad_rows(const float* src, long rows, int cols, long lds, float* dst, long ldd, void* stream);
/* Dropout multipliers of ModalityEncoder.forward (glow/models.py:56-58: nn.Dropout(p) on ones(B, hist), one scalar per (sample,
 * history step) and timestep): out[i] = 1 / keep with probability keep, else 0, for up to 4 modalities in one launch. Philox4x32-10
 * keyed on (seed, offset): reproducible from those two numbers; same law as the reference's, another stream (tests inject
 * masks for parity). */
int lfi_dropout_masks(int count, float* const* out, const long* n, const float* keep, unsigned long long seed,
                      unsigned long long offset, void* stream);
/* Conditioning.use_frame_nb (glow/models.py:89,116-117,143-144): one extra feature column holding a frame counter,
 * cond[n*B + b, col] = base[b] + offset + 2n. SeqGlow.forward / invert pass base = batch["frame_nb"] (B floats) and
 * offset = 2 * start (glow/models.py:539-542,557-558,623-625); SeqGlow.inference starts from ones: base = NULL (:572-575). */
int lfi_fill_frame_nb(const float* base, float offset, int B, int N, float* cond, int ldcond, int col, void* stream);
/* d[r, c] *= (y[r, c] > 0 ? 1 : slope): backward of the LeakyReLU of an "enc: mlp" modality, in place on the feature
 * gradient block (its weight / bias gradients then follow from lfi_gemm_f32 / lfi_colsum_f32). */
int lfi_leaky_grad(float* d, long ldd, const float* y, long ldy, int rows, int cols, float slope, void* stream);

/* ---------------------------------------------------------------- flow (FlowStep / FlowNet / Glow, glow/models.py:217-521)
 * Parameters of the Ks = K*L flow steps are struct-of-arrays, step-major: e.g. whh is [Ks][3H][H].
 */
typedef struct {
  int B, N, C, H, D, Ks;     /* batch, timesteps, channels, hidden_channels, cond_dim, flow steps */
  int affine;                /* 1 affine coupling, 0 additive (glow/models.py:330-341) */
  int lstm;                  /* 0 GRUCell, 1 LSTMCell coupling net (glow/models.py:176-185); the LSTM starts from zero
                                (h, c) at the first modelled frame (the reference's own call, models.py:209-213, crashes) */
  float scale_eps;           /* Glow.scale_eps */
  int gemm_precision;        /* lfi_gemm_desc.precision of the GEMMs issued by lfi_flow_param_grads / lfi_flow_sample_seq (bits 0..15).
                                Bit 16 (round 5; the same value in lfi_flow_seq_bwd_planes and lfi_flow_param_grads, planes mode and
                                two-product thin products only - skip bit 0x100 set): the backward stash's dgi | dgh ROWS are bf16
                                arrays of the same shapes instead of fp32 - their only readers round that operand to bf16 anyway
                                (glow/models.py:204-214 backward: the GRUCell's weight gradients) */
} lfi_flow_dims;
/* derived: Ch = C/2, C2 = C - Ch, Cout = affine ? 2*C2 : C2, G = lstm ? 4H : 3H, I = Ch + D */

typedef struct {
  const float *an_bias, *an_logs;          /* [Ks][C]   ActNorm2d (glow/modules.py:10-80) */
  const float *inv_l, *inv_u, *inv_logs;   /* [Ks][C][C], [Ks][C][C], [Ks][C]  InvertibleConv1x1 LU params */
  const float *inv_p, *inv_sign;           /* [Ks][C][C], [Ks][C]  buffers (glow/modules.py:139-145) */
  const float *inv_w;                      /* [Ks][C][C] dense weight (LU_decomposed: false), else NULL */
  const float *w_ih, *w_hh, *b_ih, *b_hh;  /* [Ks][G][I], [Ks][G][H], [Ks][G], [Ks][G]  f.rnn */
  const float *w_fl, *b_fl, *l_fl;         /* [Ks][Cout][H], [Ks][Cout], [Ks][Cout]  f.final_linear (LinearZeros) */
} lfi_flow_params;

/* Derived weights, rebuilt once per optimiser step instead of once per (timestep, flow step) as
 * InvertibleConv1x1.get_weight does (glow/modules.py:147-178). Layout of `prep` (floats), step-major blocks:
 *   W [Ks][C][C], Wt [Ks][C][C], Winv [Ks][C][C] (reverse weight, fp64 inverse cast to fp32),
 *   wz_t [Ks][Ch][G], whh_t [Ks][H][G], wfl_t [Ks][H][Cout], wc [Ks][G][D] (= W_ih[:, Ch:], 16-byte aligned copy),
 *   logdet_const [1] = C * sum(an_logs + inv_logs)
 * followed by private images of the same weights in the layouts the cell kernels load (zero-padded f32 fragments; bf16 hi / lo
 * fragments of the backward recurrent weights; with_inverse: fp16 hi / lo fragments of the reverse cell's weights, which the
 * sampler's cells load instead of splitting f32 fragments in every workgroup of every generated frame). */
long lfi_flow_prep_floats(const lfi_flow_dims* d);
int lfi_flow_prep(const lfi_flow_dims* d, const lfi_flow_params* p, float* prep, int with_inverse, void* stream);

/* Teacher-forced pass over all N timesteps and Ks flow steps (SeqGlow.forward's loop, glow/models.py:546-559,
 * FlowNet.encode :444-451, FlowStep.normal_flow :311-342, f_seq.forward :204-214), walked as anti-diagonals of
 * the (n, k) grid. x0: p1_face (B x T x C, batch-first), first frame `start`. gic: [Ks][F][G] = hoisted
 * W_ih[:, Ch:] c + b_ih. stash: lfi_flow_stash_floats floats (activations for the backward pass; also holds z).
 * nll (F): per-frame NLL in bits (SeqGlow.loss :563-565); z (F x C). */
long lfi_flow_stash_floats(const lfi_flow_dims* d);
int lfi_flow_seq_fwd(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep,
                     const float* x0, int T, int start, const float* gic,
                     float* stash, float* z, float* nll, void* stream);
/* Backward of the above for loss = gscale * sum_f nll[f]. bstash: lfi_flow_bstash_floats floats; afterwards
 * it holds dlin [Ks][F][Cout], dgi [Ks][F][G], dgh [Ks][F][G], dy [Ks][F][C] and the per-tile partial sums that
 * lfi_flow_param_grads turns into parameter gradients. dgi doubles as the gradient of gic. */
long lfi_flow_bstash_floats(const lfi_flow_dims* d);
int lfi_flow_seq_bwd(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep,
                     const float* stash, float gscale, float* bstash, void* stream);

/* The same walk, also leaving dgi - the gradient of the hoisted W_ih[:, Ch:] c + b_ih product, (Ks F x G), flow step k in rows
 * [k F, (k + 1) F) - as operand planes (lfi_planes_elems(Ks F, G) bf16) for the two products that consume it (lfi_gemm_planes):
 * d pre-activation of cond_transform = dgi W_c sums over gate columns (row use), dW_c = dgi^T c sums over frames (transposed
 * use). Written by the walk from its bf16 hi / lo LDS images instead of by a conversion pass over the fp32 dgi (which is still
 * written: the thin dW_ih[:, :Ch] product reads it). lfi_flow_bwd_emits_planes(d) = 1 when the dims allow it (bf16x3 persistent
 * walk, H and B multiples of 32). */
int lfi_flow_bwd_emits_planes(const lfi_flow_dims* d);
int lfi_flow_seq_bwd_planes(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep,
                            const float* stash, float gscale, float* bstash, void* dgi_planes,
                            int hi_only /* 1: the hi planes only (lfi_pgemm_desc.out_hi_only) */, void* stream);

typedef struct {
  float *an_bias, *an_logs, *inv_l, *inv_u, *inv_logs, *inv_w;
  float *w_ih, *w_hh, *b_ih, *b_hh, *w_fl, *b_fl, *l_fl;
} lfi_flow_grads;
/* Parameter gradients of the flow from the two stashes. c: (F x ldc) LeakyReLU(cond_transform) outputs, step k in
 * columns [k*D, (k+1)*D), or NULL when the caller computes dW_ih[:, Ch:] = dgi^T c itself (on operand planes). Overwrites (accumulate = 0) or adds to (1) the gradient arrays. work: floats from
 * lfi_flow_param_grads_work_floats. */
long lfi_flow_param_grads_work_floats(const lfi_flow_dims* d);
int lfi_flow_param_grads(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep,
                         const float* stash, const float* bstash, const float* c, long ldc, float gscale,
                         const lfi_flow_grads* g, int accumulate, float* work, void* stream,
                         void* bias_stream /* NULL = stream. Otherwise the bias / ActNorm column sums (streams over the
                                              backward stash, independent of the weight-gradient products) are enqueued
                                              there: the caller orders it after lfi_flow_seq_bwd and joins it */);

/* Pointers into the stashes (host-side address arithmetic only). which: 0 a, 1 y, 2 x_out, 3 h, 4 gates, 5 o, 6 ldc,
 * 7 LSTM cell state (empty for GRU), 8 pipeline state for the forward stash; 0 dlin, 1 dgi, 2 dgh, 3 dy, 4 dx, 5 dh,
 * 6/7 per-tile partial sums, 8 carried d cell state (LSTM), 9 pipeline state for the backward stash.
 * Pipeline state = 32-bit words of the persistent walk (lfi_flow_seq_fwd / _bwd zero it before every launch): word 1 is
 * non-zero afterwards iff a bounded spin timed out and the walk was abandoned - the results are then invalid; callers
 * check it at their next synchronisation point. */
float* lfi_flow_stash_ptr(const lfi_flow_dims* d, float* stash, int which);
float* lfi_flow_bstash_ptr(const lfi_flow_dims* d, float* bstash, int which);

/* ActNorm data-dependent initialisation for flow step k from the first timestep (glow/modules.py:32-43):
 * stats: accumulates per-channel sum and sum of squares of the step's input over the local batch into
 * sums[2*C] (fp64 on device, so ranks can all-reduce them); apply: bias = -mean, logs = log(scale/(sqrt(var)+1e-6)). */
int lfi_actnorm_init_stats(const float* x, int rows, int C, double* sums, void* stream);
int lfi_actnorm_init_apply(const double* sums, double count, int C, float scale, float* bias, float* logs, void* stream);
/* One flow step on a (rows x C) batch with explicit state; used by the init walk, by module-level
 * FlowStep.forward and by the sampler. h_prev/h_out: (rows x H) (NULL h_prev = zeros); c_prev/c_out: the LSTM cell
 * state, same shape (lstm = 1 only; c_out required then, ignored for GRU). reverse = 1 runs
 * FlowStep.reverse_flow (glow/models.py:345-373). gic_k: (rows x G). ldc_acc (rows): log-det accumulator (+=). */
int lfi_flow_step(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, int k, int rows,
                  const float* x_in, long ldx, const float* h_prev, const float* c_prev, const float* gic_k,
                  float* x_out, long ldxo, float* h_out, float* c_out, float* ldc_acc, int reverse, void* stream);
/* SeqGlow.invert (glow/models.py:617-645), the teacher-forced reverse pass of a whole sequence in ONE persistent launch (the
 * reverse twin of lfi_flow_seq_fwd's walk; replaces N * Ks lfi_flow_step(reverse = 1) calls): z (N x B x C latents) ->
 * x_out (N x B x C), logdet (N x B) = the sum over the flow steps of the coupling log-dets, negated as FlowStep.reverse_flow
 * returns them (the constant ActNorm / invconv term is the caller's: lfi_flow_prep's constant). gic: [Ks][N*B][G] as for
 * lfi_flow_seq_fwd. h / cstate: [Ks][B][H] scratch for the recurrent state (zero state at the first timestep, as
 * init_rnn_hidden, glow/models.py:619). Only where lfi_flow_seq_rev_ok(d) (C <= 64, hidden_channels <= 128); prep must hold
 * the inverse weights (lfi_flow_prep with_inverse = 1). */
int lfi_flow_seq_rev_ok(const lfi_flow_dims* d);
long lfi_flow_seq_rev_work_floats(const lfi_flow_dims* d);
int lfi_flow_seq_rev(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep, const float* z, const float* gic,
                     float* x_out, float* logdet, float* h, float* cstate, float* work, void* stream);

/* Stand-alone module calls (what code/glow_pytorch/test_modules.py:9-29 exercises outside any flow).
 * lfi_actnorm_forward: ActNorm2d.forward (glow/modules.py:45-80) on a (rows x C) batch: out = (x + bias) exp(logs), or with
 * reverse = 1 x exp(-logs) - bias; dlogdet[0] (nullable) = +-C * sum(logs).
 * lfi_invconv_weights: InvertibleConv1x1.get_weight (glow/modules.py:147-178): W (C x C) from the LU parameters or the dense
 * weight inv_w, optionally the reverse weight Winv (fp64 inverses cast to fp32; required when with_inverse, and written
 * whenever inv_w is given), dlogdet[0] = C * sum(inv_logs) resp. C * log|det inv_w|. z = x W is then one lfi_gemm_f32.
 * work: lfi_invconv_work_floats(C) floats, 8-byte aligned. */
int lfi_actnorm_forward(const float* x, int rows, int C, const float* bias, const float* logs, int reverse, float* out,
                        float* dlogdet, void* stream);
long lfi_invconv_work_floats(int C);
int lfi_invconv_weights(int C, const float* inv_l, const float* inv_u, const float* inv_logs, const float* inv_p,
                        const float* inv_sign, const float* inv_w, int with_inverse, float* W, float* Winv, float* dlogdet,
                        float* work, void* stream);

/* ---------------------------------------------------------------- autoregressive sampling (SeqGlow.inference, glow/models.py:567-596)
 * Whole sequence in one call. faces (B x seq_len x C, batch-first) holds the `start` seed frames and receives the
 * generated ones. Per frame t: c = LeakyReLU(pre_static[n] + faces[:, t-hist1:t] Wct[:, :hist1*C]^T) for all Ks steps
 * (pre_static = everything of cond_transform that does not depend on generated frames, bias included:
 * (nframes*B) x (Ks*D), frame-major; CONSUMED: c of frame n is written over its rows), gic = c W_ih[:, Ch:]^T + b_ih, then the Ks reverse flow steps from the
 * injected prior noise (nframes x B x C, already scaled by eps_std). h: [Ks][B][H] recurrent state, zero on entry.
 * Only "enc: none" for p1_face (all shipped hparams). work: lfi_flow_sample_work_floats floats. */
/* The autoregressive prev_p1_face window may itself go through a ModalityEncoder (the reference's hparam search draws
 * p1_face "enc" from {rnn, mlp, none}, hparam_tuning_configs/large_hparam_search.py:45-62): then its features are
 * recomputed for every generated frame. kind 0: "none" (raw window); 1: "mlp" = LeakyReLU(window W1^T + b1);
 * 2: "rnn" = GRU from h0 = 0 over the window (output folded: the hidden state once); 3: "lstm" = nn.LSTM likewise
 * (glow/models.py:27-33,65-69; four gate blocks in w_ih / w_hh / b_ih / b_hh). `col` = first column of the
 * p1_face block in the (folded) feature layout / in wct; eval mode (no dropout), as SeqGlow.inference runs. */
typedef struct {
  int kind, hid;
  const float *w1, *b1;                        /* mlp: [hid][hist1*C], [hid] */
  const float *w_ih, *w_hh, *b_ih, *b_hh;      /* rnn: [3hid][C], [3hid][hid], [3hid], [3hid]; lstm: 4hid */
  int col;
} lfi_p1enc;

long lfi_flow_sample_work_floats(const lfi_flow_dims* d);
long lfi_flow_sample_p1_work_floats(const lfi_flow_dims* d, const lfi_p1enc* e, int hist1);
int lfi_flow_sample_seq(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep,
                        const float* wct /* [Ks*D][E] */, long E, int hist1,
                        float* pre_static, const float* noise,
                        float* faces, int seq_len, int start, int nframes,
                        float* h, float* cstate /* [Ks][B][H] LSTM cell state, zero on entry; NULL for GRU */,
                        const lfi_p1enc* p1 /* NULL = "none" */, float* p1work /* lfi_flow_sample_p1_work_floats */,
                        float* work, void* stream);
/* The same for a run of frames that continues a sequence: first_frame = number of frames earlier calls generated (the state in
 * h / cstate carries on when > 0); pre_static, noise and start are this run's own. Lets the caller compute the static part of
 * the next run on another stream while this run's chain of dependent cells executes. */
/* Range guard of the sampler's fp16-piece arithmetic: out_bits[0] = bit pattern of max |v| over `count` (<= 8) fp32 arrays, one
 * launch (finite |v| order like their bit patterns; an inf reads back as inf, a NaN as NaN). Replaces one blocking
 * `tensor.abs().max()` per input of SeqGlow.inference (glow/models.py:567-596 has no such guard: its arithmetic is plain fp32). */
int lfi_absmax_f32(int count, const float* const* ptrs, const long* n, unsigned* out_bits, void* stream);
/* A HIP stream that owns `cus_per_xcd` CUs of every XCD (1 .. 32 on MI355X) and leaves the rest of the chip to the streams beside
 * it: the sampler's static part (window encoders, static cond_transform columns of the NEXT run of frames) runs on one while the
 * per-frame chain keeps the other CUs - on an ordinary second stream the two only take turns. Not a reference
 * interface: SeqGlow.inference (glow/models.py:567-596) is one PyTorch stream. Destroy with lfi_stream_destroy. The CU-mask bit
 * layout is known (probed) for the 256-CU / 8-XCD part only: any other device gets LFI_ERR_ARG and the caller stays on an ordinary
 * second stream. */
int lfi_stream_create_partial(int cus_per_xcd, void** stream);
int lfi_stream_destroy(void* stream);
int lfi_flow_sample_seq_from(const lfi_flow_dims* d, const lfi_flow_params* p, const float* prep,
                             const float* wct, long E, int hist1, float* pre_static, const float* noise,
                             float* faces, int seq_len, int start, int nframes, int first_frame,
                             float* h, float* cstate, const lfi_p1enc* p1, float* p1work, float* work, void* stream);

/* ---------------------------------------------------------------- optimiser (configure_optimizers, glow/lets_face_it_glow.py:61-72)
 * Flat-buffer Adam with global-norm gradient clipping (Trainer gradient_clip_val, hparams/final_model.yaml:126):
 * norm: sumsq[0] = sum g^2 (fp64, deterministic). step: coef = min(1, clip/(sqrt(sumsq)+1e-6)) (clip <= 0: 1),
 * g *= coef * gmul; Adam (weight decay / amsgrad: lfi_adam_clip_step_ex below). step_count is 1-based. */
int lfi_grad_sumsq(const float* g, long n, double* sumsq, double* work /* 1024 doubles */, void* stream);
int lfi_adam_clip_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq,
                       float clip, float gmul, float lr, float beta1, float beta2, float eps, int step_count,
                       void* stream);

/* torch.optim.Adam's remaining constructor arguments, which the reference forwards verbatim from the YAML
 * (glow/lets_face_it_glow.py:61-70: `Adam(params, lr=lr, **Optim["args"]["adam"])`): weight_decay (L2 term `wd * p` added to the
 * clipped gradient before the moments) and amsgrad (vmax = running maximum of the second moment, used in the denominator; NULL =
 * off). hyper: NULL, or the device block of a captured step (as lfi_adam_clip_step_dev; lr / step_count are then not read). */
int lfi_adam_clip_step_ex(float* p, const float* g, float* m, float* v, float* vmax, long n, const double* sumsq, float clip,
                          float gmul, float lr, float beta1, float beta2, float eps, float weight_decay, int step_count,
                          const float* hyper, void* stream);

/* The other two optimisers configure_optimizers can build (glow/lets_face_it_glow.py:61-72: `{"adam", "sgd", "rmsprop"}[name](params,
 * lr=hparams.lr, **hparams.Optim["args"][name])`; hparam_tuning_configs/large_hparam_search.py:9-11 draws all three), on the same
 * flat buffers with the same global-norm clip in front: torch.optim.SGD (momentum, dampening, weight_decay, nesterov; final_model.yaml:
 * momentum 0.9; buf = the momentum buffer, may be NULL when momentum == 0; step_count 1-based: the first step sets buf = g) and
 * torch.optim.RMSprop (alpha, eps, weight_decay, momentum, centered: sq = the square average, buf = the momentum buffer or NULL,
 * gavg = the centered variant's gradient average or NULL; final_model.yaml: eps 1e-8, torch defaults otherwise). */
int lfi_sgd_clip_step(float* p, const float* g, float* buf, long n, const double* sumsq, float clip, float gmul, float lr,
                      float momentum, float dampening, float weight_decay, int nesterov, int step_count, void* stream);
int lfi_rmsprop_clip_step(float* p, const float* g, float* sq, float* buf, float* gavg, long n, const double* sumsq, float clip,
                          float gmul, float lr, float alpha, float eps, float weight_decay, float momentum, void* stream);

/* The training step as a captured hipGraph (lets_face_it_amd/glow/lets_face_it_glow.py, fused_training_step): what changes from
 * one optimiser step to the next - the dropout-mask key, Adam's bias-corrected step size - lives in a 32-byte device block
 * (u64 seed, u64 mask offset, f32 step_size = lr / (1 - beta1^t), f32 1 / sqrt(1 - beta2^t)) that lfi_set_step_params fills with
 * an ordinary launch before every replay; the _dev variants of the two kernels read it instead of taking the values as
 * arguments. The host computes the two floats exactly as lfi_adam_clip_step does (bit-identical parameters, tested). */
int lfi_set_step_params(void* params, unsigned long long seed, unsigned long long offset, float step_size, float inv_sqrt_bc2,
                        void* stream);
int lfi_dropout_masks_dev(int count, float* const* out, const long* n, const float* keep, const unsigned long long* seed_offset,
                          void* stream);
int lfi_adam_clip_step_dev(float* p, const float* g, float* m, float* v, long n, const double* sumsq, float clip, float gmul,
                           float beta1, float beta2, float eps, const float* hyper /* = params + 16 bytes */, void* stream);

/* ---------------------------------------------------------------- callers either side of the flow (SURVEY.md par. 8f)
 * Window sampler, replaces MimicryDataset.__getitem__ + DataLoader collation (code/glow_pytorch/mimicry_data_module.py:45-78):
 * one modality of a split lives in HBM as ONE (rows x dim) matrix, the recording bins back to back; a training window is T
 * consecutive rows of one bin. dst[b, t, :] = src[starts[b] + t, :] for b < B (starts: B row indices, int64, on the device;
 * the host-side index table guarantees starts[b] + T does not cross a bin boundary). dst: (B x T x dim), batch-first. */
int lfi_gather_sequences(const float* src, long rows, int dim, const long* starts, int B, int T, float* dst, void* stream);
/* calc_jerk (code/glow_pytorch/glow/utils.py:53-58; MimicryLogger, mimicry_logger.py:187-196): out[0] = mean over (B, T-3, C)
 * of |third difference along time| of x (B x T x C, batch-first, fp32 differences as the reference). work: 1024 doubles. */
int lfi_jerk_mean(const float* x, int B, int T, int C, float* out, double* work /* 1024 doubles */, void* stream);

/* ---------------------------------------------------------------- diagnostics */
/* Timing probe of the register-resident forward cells: a device buffer of >= 16 * Ks 64-bit slots stamped with s_memtime
 * (100 MHz) at the phase boundaries of the cells in workgroup column 0; NULL switches it off. Process-global - the ONE piece of
 * global mutable state in this library, null in normal operation (a per-thread pointer was tried in round 5: PyTorch runs the
 * backward pass on its autograd thread, so the walks' backward stamps never saw it). Set it only around a single-engine
 * diagnostic run (tools/pipe_stamps.py, tools/rev_stamps.py). */
int lfi_debug_set_stamps(void* device_buffer);
/* Checks the MFMA operand/accumulator lane maps the kernels rely on; out[0] = number of mismatches. */
int lfi_selftest_mfma(int* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LFI_H */
