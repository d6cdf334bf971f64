"""GPU: every C-ABI kernel family against a plain torch / oracle computation of the same op (call through the C ABI)."""
import ctypes as C
import os

import pytest
import torch

from helpers import rel_err, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(gpu_device):
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.engine import GlowEngine, ModelSpec
    return GlowEngine(ModelSpec(Namespace(**Fixture("tiny").hp)), gpu_device)


def test_mfma_lane_maps(gpu_device):
    from lets_face_it_amd import _lib
    out = torch.full((1,), -1, dtype=torch.int32, device=gpu_device)
    _lib.check(_lib.lib().lfi_selftest_mfma(out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert int(out.item()) == 0


GEMM_CASES = [
    # M, N, K, akc, bkc, batch, splitk, bias, act, accumulate
    (128, 128, 64, 1, 1, 1, 1, False, 0, 0),
    (200, 96, 50, 1, 1, 1, 1, True, 1, 0),
    (130, 70, 33, 0, 0, 1, 1, False, 0, 1),
    (64, 300, 17, 1, 0, 3, 1, True, 0, 0),
    (384, 28, 900, 0, 0, 4, 8, False, 0, 0),
    (300, 40, 1530, 1, 1, 1, 1, True, 1, 0),
    (56, 56, 2000, 0, 0, 2, 16, False, 0, 1),
    (1000, 512, 96, 1, 0, 2, 1, False, 2, 0),
    (33, 1, 7, 1, 1, 1, 1, True, 0, 2),
    (16, 520, 280, 1, 1, 1, 1, False, 1, 2),
    # the split-K reduce pass: four columns per thread with bias + activation + accumulate, the scalar form (N % 4 != 0),
    # the leaky-gradient gate G read by the reduce
    (260, 132, 1100, 1, 0, 2, 4, True, 1, 2),
    (130, 70, 1200, 0, 1, 1, 3, True, 0, 1),
    (256, 64, 2048, 1, 1, 2, 4, False, 2, 0),
]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_against_torch(eng, gpu_device, case):
    M, N, K, akc, bkc, batch, splitk, use_bias, act, accumulate = case
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((batch, M, K) if akc else (batch, K, M), generator=g).to(gpu_device)
    Bm = torch.randn((batch, N, K) if bkc else (batch, K, N), generator=g).to(gpu_device)
    C0 = torch.randn(batch, M, N, generator=g).to(gpu_device)
    bias = torch.randn(batch, N, generator=g).to(gpu_device) if use_bias else None
    G = torch.randn(batch, M, N, generator=g).to(gpu_device) if act == 2 else None
    Cm = C0.clone()
    eng.gemm(M, N, K, A, K if akc else M, akc, Bm, K if bkc else N, bkc, Cm, N, bias=bias, act=act, slope=0.01, G=G,
             ldg=N, batch=batch, sA=M * K, sB=N * K, sC=M * N, sBias=N, sG=M * N, accumulate=accumulate, splitk=splitk)
    torch.cuda.synchronize()
    Ad = (A if akc else A.transpose(1, 2)).double()
    Bd = (Bm.transpose(1, 2) if bkc else Bm).double()
    ref = Ad @ Bd
    if bias is not None:
        ref = ref + bias.double().unsqueeze(1)
    if accumulate == 2:
        ref = ref + C0.double()
    if act == 1:
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
    elif act == 2:
        ref = torch.where(G.double() > 0, ref, ref * 0.01)
    if accumulate == 1:
        ref = ref + C0.double()
    assert rel_err(Cm, ref) < 2e-6, case


def test_gemm_strided_views(eng, gpu_device):
    """The leading-dimension / offset forms the engine relies on (column blocks of wider matrices)."""
    g = torch.Generator().manual_seed(5)
    F, Ks, D, G3, I, Ch = 70, 3, 24, 36, 31, 7
    c = torch.randn(F, Ks * D, generator=g).to(gpu_device)
    w_ih = torch.randn(Ks, G3, I, generator=g).to(gpu_device)
    b_ih = torch.randn(Ks, G3, generator=g).to(gpu_device)
    gic = torch.zeros(Ks, F, G3, device=gpu_device)
    eng.gemm(F, G3, D, c, Ks * D, 1, w_ih, I, 1, gic, G3, bias=b_ih, batch=Ks, sA=D, sB=G3 * I, sC=F * G3, sBias=G3, b_off=Ch)
    torch.cuda.synchronize()
    ref = torch.stack([c[:, k * D:(k + 1) * D].double() @ w_ih[k, :, Ch:].double().t() + b_ih[k].double() for k in range(Ks)])
    assert rel_err(gic, ref) < 2e-6


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_padded_rows_take_the_16_byte_path(eng, gpu_device, akc, bkc):
    """Row strides that are multiples of 4 floats with K / M / N not filling them (folded feature matrix: 890 of 896)."""
    g = torch.Generator().manual_seed(17 + 2 * akc + bkc)
    M, N, K = 260, 136, 890
    lda = 896 if akc else 264
    ldb = 896 if bkc else 144
    A = torch.full((M, lda) if akc else (K, lda), float("nan"))
    Bm = torch.full((N, ldb) if bkc else (K, ldb), float("nan"))
    if akc:
        A[:, :K] = torch.randn(M, K, generator=g)
    else:
        A[:, :M] = torch.randn(K, M, generator=g)
    if bkc:
        Bm[:, :K] = torch.randn(N, K, generator=g)
    else:
        Bm[:, :N] = torch.randn(K, N, generator=g)
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    Cm = torch.zeros(M, N, device=gpu_device)
    eng.gemm(M, N, K, A, lda, akc, Bm, ldb, bkc, Cm, N)
    torch.cuda.synchronize()
    Ad = (A[:, :K] if akc else A[:, :M].t()).double()
    Bd = (Bm[:, :K].t() if bkc else Bm[:, :N]).double()
    assert torch.isfinite(Cm).all()
    assert rel_err(Cm, Ad @ Bd) < 2e-6


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("shape", [(260, 136, 890), (128, 512, 32), (1000, 96, 2500), (64, 64, 100)])
def test_gemm_bf16x3(eng, gpu_device, akc, bkc, shape):
    """Split-operand bf16 MFMA path: fp32-class accuracy (bound 3e-5 of the output scale; plain bf16 would be ~3e-3)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + 2 * akc + bkc)
    r4 = lambda v: (v + 3) // 4 * 4  # noqa: E731
    lda, ldb = (r4(K) + 4 if akc else r4(M) + 4), (r4(K) + 8 if bkc else r4(N))
    A = torch.full((M, lda) if akc else (K, lda), float("nan"))
    Bm = torch.full((N, ldb) if bkc else (K, ldb), float("nan"))
    if akc:
        A[:, :K] = torch.randn(M, K, generator=g)
    else:
        A[:, :M] = torch.randn(K, M, generator=g)
    if bkc:
        Bm[:, :K] = torch.randn(N, K, generator=g)
    else:
        Bm[:, :N] = torch.randn(K, N, generator=g)
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    Cm = torch.zeros(M, N, device=gpu_device)
    eng.precision = 1
    try:
        eng.gemm(M, N, K, A, lda, akc, Bm, ldb, bkc, Cm, N, bias=bias, act=1, slope=0.01,
                 splitk=4 if K >= 2000 else 1)
    finally:
        eng.precision = 0
    torch.cuda.synchronize()
    Ad = (A[:, :K] if akc else A[:, :M].t()).double()
    Bd = (Bm[:, :K].t() if bkc else Bm[:, :N]).double()
    ref = torch.nn.functional.leaky_relu(Ad @ Bd + bias.double(), 0.01)
    assert torch.isfinite(Cm).all()
    err = rel_err(Cm, ref)
    assert err < 3e-5, err


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("shape", [(260, 136, 890), (1024, 384, 512), (1000, 96, 2500), (64, 64, 100)])
def test_gemm_six_bf16_products_are_fp32_grade(eng, gpu_device, akc, bkc, shape):
    """lfi_gemm_desc.precision 5: operands in three bf16 pieces, six products (the sampler's per-frame GEMMs). What is dropped
    is 2^-24 relative - the fp32 rounding itself: its error against fp64 must be that of the exact f32-input MFMA kernel
    (within 1.5x; both are bounded by fp32 accumulation over K) and well below the three-product kernel's."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K + 2 * akc + bkc)
    r4 = lambda v: (v + 3) // 4 * 4  # noqa: E731
    lda, ldb = (r4(K) + 4 if akc else r4(M) + 4), (r4(K) + 8 if bkc else r4(N))
    A = torch.zeros((M, lda) if akc else (K, lda))
    Bm = torch.zeros((N, ldb) if bkc else (K, ldb))
    if akc:
        A[:, :K] = torch.randn(M, K, generator=g)
    else:
        A[:, :M] = torch.randn(K, M, generator=g)
    if bkc:
        Bm[:, :K] = torch.randn(N, K, generator=g)
    else:
        Bm[:, :N] = torch.randn(K, N, generator=g)
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    Ad = (A[:, :K] if akc else A[:, :M].t()).double()
    Bd = (Bm[:, :K].t() if bkc else Bm[:, :N]).double()
    ref = Ad @ Bd + bias.double()
    errs = {}
    for mode in (0, 1, 5, 9):
        Cm = torch.zeros(M, N, device=gpu_device)
        eng.precision = mode
        try:
            eng.gemm(M, N, K, A, lda, akc, Bm, ldb, bkc, Cm, N, bias=bias, splitk=4 if K >= 2000 else 1)
        finally:
            eng.precision = 0
        torch.cuda.synchronize()
        assert torch.isfinite(Cm).all()
        errs[mode] = rel_err(Cm, ref)
    report("GEMM %s (A %s, B %s): rel L2 err vs fp64: exact f32 %.2e, three bf16 products %.2e, six %.2e, three fp16 products %.2e"
           % (shape, "k-contig" if akc else "mn-contig", "k-contig" if bkc else "mn-contig", errs[0], errs[1], errs[5], errs[9]))
    assert errs[5] < max(1.5 * errs[0], 2e-7), errs
    assert errs[5] < 0.5 * errs[1], errs
    # precision 9: fp16 pieces (11 + 11 bits, 2^-22 relative per product; N(0, 1) operands sit inside fp16's range)
    assert errs[9] < max(1.5 * errs[0], 3e-7), errs


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("shape,splitk", [((700, 520, 330), 1), ((256, 300, 2100), 0), ((513, 257, 75), 1), ((300, 260, 128), 1)])
@pytest.mark.parametrize("pin", [0x11], ids=["k16"])
def test_gemm_bf16x3_256_tile(eng, gpu_device, akc, bkc, shape, splitk, pin):
    """The 256 x 256 kernel (k-tile 16: transposing LDS reads for mn-contiguous operands, register prefetch, wide epilogue),
    pinned with precision | 0x10: ragged edges, a partial last k-tile, short K (drain loop only), one steady iteration,
    library-chosen split."""
    M, N, K = shape
    g = torch.Generator().manual_seed(7 * M + N + K + 2 * akc + bkc)
    r4 = lambda v: (v + 3) // 4 * 4  # noqa: E731
    lda, ldb = (r4(K) + 4 if akc else r4(M) + 4), (r4(K) + 8 if bkc else r4(N))
    A = torch.full((M, lda) if akc else (K, lda), float("nan"))
    Bm = torch.full((N, ldb) if bkc else (K, ldb), float("nan"))
    if akc:
        A[:, :K] = torch.randn(M, K, generator=g)
    else:
        A[:, :M] = torch.randn(K, M, generator=g)
    if bkc:
        Bm[:, :K] = torch.randn(N, K, generator=g)
    else:
        Bm[:, :N] = torch.randn(K, N, generator=g)
    # the padding is NaN only where no load may touch it: rows are read up to the next multiple of 4 columns
    if not akc:
        A[:, M:r4(M)] = 0.0
    if not bkc:
        Bm[:, N:r4(N)] = 0.0
    if akc:
        A[:, K:r4(K)] = 0.0
    if bkc:
        Bm[:, K:r4(K)] = 0.0
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    ldc = r4(N) + 4
    Cm = torch.full((M, ldc), 7.0, device=gpu_device)
    eng.precision = pin
    try:
        eng.gemm(M, N, K, A, lda, akc, Bm, ldb, bkc, Cm, ldc, bias=bias, act=1, slope=0.01, splitk=splitk)
    finally:
        eng.precision = 0
    torch.cuda.synchronize()
    Ad = (A[:, :K] if akc else A[:, :M].t()).double()
    Bd = (Bm[:, :K].t() if bkc else Bm[:, :N]).double()
    ref = torch.nn.functional.leaky_relu(Ad @ Bd + bias.double(), 0.01)
    assert torch.isfinite(Cm).all() and bool((Cm[:, N:] == 7.0).all()), "wrote outside the N columns"
    err = rel_err(Cm[:, :N], ref)
    assert err < 3e-5, err


@pytest.mark.parametrize("shape", [(700, 520, 330), (256, 256, 16), (513, 257, 75), (300, 260, 128), (1024, 768, 896), (40, 33, 7)])
@pytest.mark.parametrize("epi", ["bias_leaky", "plain", "accumulate"])
def test_gemm_on_presplit_planes(eng, gpu_device, shape, epi, monkeypatch):
    """lfi_planes_from_f32 + lfi_gemm_planes (operands split to bf16 hi / lo ONCE, in MFMA fragment order, streamed to LDS by
    LDS-DMA through a three-slot ring) against the fp64 product and, bit for bit, against lfi_gemm_f32's 256 x 256 bf16x3
    kernel (same split, same products, same accumulation order): ragged M / N / K, fewer k-tiles than ring slots, many tiles.
    With an even number of k-tiles both operands by rows take the v_mfma_f32_16x16x32_bf16 kernel (k-tiles in pairs): the same
    products summed in another order - against fp64 and the 32 x 32 kernel to fp32 rounding, bit for bit with itself, and the
    32 x 32 kernel (LFI_PGEMM_16=0) bit for bit against lfi_gemm_f32 as before."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    r4 = lambda v: (v + 3) // 4 * 4  # noqa: E731
    lda, ldb, ldc = r4(K) + 4, r4(K), r4(N) + 4
    A = torch.full((M, lda), float("nan"))
    Bm = torch.full((N, ldb), float("nan"))
    A[:, :K] = torch.randn(M, K, generator=g)
    Bm[:, :K] = torch.randn(N, K, generator=g)
    A[:, K:r4(K)] = 0.0
    Bm[:, K:r4(K)] = 0.0
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device) if epi == "bias_leaky" else None
    act = 1 if epi == "bias_leaky" else 0
    acc = 1 if epi == "accumulate" else 0
    C0 = torch.randn(M, ldc, generator=g).to(gpu_device)
    Ap, nka = eng.planes("test.pa", A, lda, M, K)
    Bp, nkb = eng.planes("test.pb", Bm, ldb, N, K)
    C1 = C0.clone()
    eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C1, ldc, bias=bias, act=act, accumulate=acc)
    C2 = C0.clone()
    eng.precision = 0x11
    try:
        eng.gemm(M, N, K, A, lda, 1, Bm, ldb, 1, C2, ldc, bias=bias, act=act, slope=0.01, accumulate=acc)
    finally:
        eng.precision = 0
    torch.cuda.synchronize()
    ref = A[:, :K].double() @ Bm[:, :K].double().t()
    if bias is not None:
        ref = torch.nn.functional.leaky_relu(ref + bias.double(), 0.01)
    if acc:
        ref = ref + C0[:, :N].double()
    assert torch.equal(C1[:, N:], C0[:, N:]), "wrote outside the N columns"
    assert rel_err(C1[:, :N], ref) < 3e-5
    if ((K + 15) // 16) % 2 == 0 and os.environ.get("LFI_PGEMM_16", "1") != "0":   # (the suite also runs with the switches off)
        C3 = C0.clone()
        eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C3, ldc, bias=bias, act=act, accumulate=acc)
        monkeypatch.setenv("LFI_PGEMM_16", "0")
        C4 = C0.clone()
        eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C4, ldc, bias=bias, act=act, accumulate=acc)
        torch.cuda.synchronize()
        assert torch.equal(C1, C3)
        assert not torch.equal(C1, C4), "the 16 x 16 x 32 kernel did not run"
        assert rel_err(C1[:, :N], C4[:, :N]) < 2e-6
        C1 = C4
    assert torch.equal(C1, C2)


def test_gemm_on_presplit_planes_batched_k_ranges(eng, gpu_device):
    """Batch entries that are k-tile ranges of ONE plane buffer (the gic product: flow step k multiplies columns
    [k D, (k + 1) D) of c) against separate per-step weight planes."""
    g = torch.Generator().manual_seed(5)
    F, D, G, Ks = 300, 64, 96, 3
    c = torch.randn(F, Ks * D, generator=g).to(gpu_device)
    W = torch.randn(Ks, G, D, generator=g).to(gpu_device)
    bias = torch.randn(Ks, G, generator=g).to(gpu_device)
    out = torch.zeros(Ks, F, G, device=gpu_device)
    cp, nkc = eng.planes("test.pc", c, Ks * D, F, Ks * D)
    wp, nkw = eng.planes("test.pw", W.view(Ks * G, D), D, Ks * G, D)   # 288 rows -> padded to 512: per-step offset below
    # step k's weight rows start at row k * G = row tile 3 k: a whole number of 32-row tiles, so a batch stride works
    eng.gemm_planes(F, G, D, cp, nkc, wp, nkw, out, G, bias=bias, batch=Ks, a_stride=(D // 16) * 1024,
                    b_stride=(G // 32) * nkw * 1024, sC=F * G, sBias=G)
    torch.cuda.synchronize()
    for k in range(Ks):
        ref = c[:, k * D:(k + 1) * D].double() @ W[k].double().t() + bias[k].double()
        assert rel_err(out[k], ref) < 3e-5, k


@pytest.mark.parametrize("shape", [(700, 520, 330), (384, 512, 14336), (513, 257, 75), (130, 40, 16), (1024, 896, 2048)])
@pytest.mark.parametrize("fmt", ["RT", "TR", "TT"])
@pytest.mark.parametrize("splitk,tile", [(1, 1), (3, 1), (1, 2), (2, 2)])
def test_gemm_planes_kmajor_operands_and_split_k(eng, gpu_device, monkeypatch, shape, fmt, splitk, tile):
    """Planes in TRANSPOSED use (a_fmt / b_fmt = 1: the matrix' ROWS are the contraction index - every weight-gradient product sums
    over frames) in either operand slot, read with ds_read_b64_tr_b16, with and without a K split:
    against the fp64 product and, bit for bit, against lfi_gemm_f32's bf16x3 kernel on the same fp32 operands (same split, same
    products, same order - with a K split only when both split the same way, so that case is checked against fp64 alone).
    Both kernels: the 32 x 32 x 16 one (LFI_PGEMM_16=0: the bitwise comparison) and, where K has whole pairs of k-tiles, the
    16 x 16 x 32 one (default; equal to the other to fp32 rounding)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    r4 = lambda v: (v + 3) // 4 * 4  # noqa: E731
    at, bt = fmt[0] == "T", fmt[1] == "T"
    # a k-contiguous operand is (mn x K), an mn-contiguous one (K x mn)
    lda, ldb = (r4(M) + 4 if at else r4(K) + 4), (r4(N) if bt else r4(K))
    A = torch.zeros((K, lda) if at else (M, lda))
    Bm = torch.zeros((K, ldb) if bt else (N, ldb))
    if at:
        A[:, :M] = torch.randn(K, M, generator=g)
    else:
        A[:, :K] = torch.randn(M, K, generator=g)
    if bt:
        Bm[:, :N] = torch.randn(K, N, generator=g)
    else:
        Bm[:, :K] = torch.randn(N, K, generator=g)
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    ldc = r4(N) + 4
    bias = torch.randn(N, generator=g).to(gpu_device)
    C1 = torch.full((M, ldc), 7.0, device=gpu_device)
    # (the planes of the matrix as it is stored: K x M for a transposed-use operand, M x K for a row-use one)
    Ap, nka = eng.planes("test.pa", A, lda, K, M) if at else eng.planes("test.pa", A, lda, M, K)
    Bp, nkb = eng.planes("test.pb", Bm, ldb, K, N) if bt else eng.planes("test.pb", Bm, ldb, N, K)
    # tile 1: 128 x 256 tiles, 2: 256 x 128 (what the library picks for N = 384 or 896)
    C16 = torch.full((M, ldc), 7.0, device=gpu_device)
    eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C16, ldc, bias=bias, act=1, a_fmt=int(at), b_fmt=int(bt), splitk=splitk, tile=tile)
    monkeypatch.setenv("LFI_PGEMM_16", "0")
    eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C1, ldc, bias=bias, act=1, a_fmt=int(at), b_fmt=int(bt), splitk=splitk, tile=tile)
    C2 = torch.full((M, ldc), 7.0, device=gpu_device)
    eng.precision = 0x11
    try:
        eng.gemm(M, N, K, A, lda, 0 if at else 1, Bm, ldb, 0 if bt else 1, C2, ldc, bias=bias, act=1, slope=0.01)
    finally:
        eng.precision = 0
    torch.cuda.synchronize()
    Ad = (A[:, :M].t() if at else A[:, :K]).double()
    Bd = (Bm[:, :N] if bt else Bm[:, :K].t()).double()
    ref = torch.nn.functional.leaky_relu(Ad @ Bd + bias.double(), 0.01)
    assert bool((C1[:, N:] == 7.0).all()) and bool((C16[:, N:] == 7.0).all()), "wrote outside the N columns"
    assert rel_err(C1[:, :N], ref) < 3e-5 and rel_err(C16[:, :N], ref) < 3e-5
    assert rel_err(C16[:, :N], C1[:, :N]) < 2e-6
    if splitk == 1:
        assert torch.equal(C1, C2)


@pytest.mark.parametrize("fmt", ["RR", "TT", "RT"])
@pytest.mark.parametrize("k16", ["1", "0"])
def test_gemm_planes_two_products_never_touch_a_lo(eng, gpu_device, monkeypatch, fmt, k16):
    """Skip bit 0 (two products per k-step, A rounded to bf16): A's lo planes are neither fetched nor read - poisoned with NaN
    they leave the result unchanged - and the result equals lfi_gemm_f32's two-product kernel bit for bit (the 32 x 32 x 16 kernel;
    with B in transposed use the default is the 16 x 16 x 32 kernel, k-tiles in pairs: the same products in another summation order,
    equal to fp32 rounding). That is what lets the producers of A (the backward walk's dgi, the dpre product) write hi planes only
    (lfi_pgemm_desc.out_hi_only)."""
    monkeypatch.setenv("LFI_PGEMM_16T", k16)
    M, N, K = 384, 512, 2048
    g = torch.Generator().manual_seed(1)
    at, bt = fmt[0] == "T", fmt[1] == "T"
    A = torch.randn((K, M) if at else (M, K), generator=g).to(gpu_device)
    Bm = torch.randn((K, N) if bt else (N, K), generator=g).to(gpu_device)
    Ap, nka = eng.planes("test.pa", A, M, K, M) if at else eng.planes("test.pa", A, K, M, K)
    Bp, nkb = eng.planes("test.pb", Bm, N, K, N) if bt else eng.planes("test.pb", Bm, K, N, K)
    eng.pass_skip = {"t": 1}
    try:
        C1 = torch.zeros(M, N, device=gpu_device)
        eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C1, N, a_fmt=int(at), b_fmt=int(bt), cls="t")
        Ap.view(-1, 2, 512)[:, 1] = float("nan")          # every lo block of A
        C2 = torch.zeros(M, N, device=gpu_device)
        eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, C2, N, a_fmt=int(at), b_fmt=int(bt), cls="t")
        C3 = torch.zeros(M, N, device=gpu_device)
        eng.precision = 0x11
        eng.gemm(M, N, K, A, M if at else K, 0 if at else 1, Bm, N if bt else K, 0 if bt else 1, C3, N, cls="t")
    finally:
        eng.precision = 0
        eng.pass_skip = {}
    torch.cuda.synchronize()
    assert torch.isfinite(C2).all() and torch.equal(C1, C2)
    if bt and k16 == "1":
        assert not torch.equal(C1, C3), "the 16 x 16 x 32 kernel did not run"
        assert rel_err(C1, C3) < 2e-6
    else:
        assert torch.equal(C1, C3)
    ref = (A.t() if at else A).double() @ (Bm if bt else Bm.t()).double()
    assert 1e-5 < rel_err(C1, ref) < 3e-3       # a rounded operand: 2^-9 per element, averaged over K


@pytest.mark.parametrize("shape", [(700, 520, 352), (384, 512, 14336), (513, 257, 96), (130, 40, 32), (1024, 896, 2048)])
@pytest.mark.parametrize("fmt", ["RT", "TT"])
@pytest.mark.parametrize("splitk,tile", [(1, 1), (3, 1), (1, 2), (2, 2)])
def test_gemm_planes_two_products_on_16x16x32(eng, gpu_device, monkeypatch, shape, fmt, splitk, tile):
    """The backward products' kernel (two products, B in transposed use, A by rows or transposed, v_mfma_f32_16x16x32_bf16 on pairs
    of k-tiles, one plane per ring slot): ragged M / N, K down to one pair, split K (whole pairs per split or the 32 x 32 kernel),
    both tile shapes - against the 32 x 32 x 16 kernel (LFI_PGEMM_16T=0) to fp32 rounding, against the fp64 product of the ROUNDED A,
    bit for bit with itself, bias + LeakyReLU epilogue, nothing written outside the N columns."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    r4 = lambda v: (v + 3) // 4 * 4  # noqa: E731
    at = fmt[0] == "T"
    lda, ldb = (r4(M) + 4 if at else r4(K) + 4), r4(N)
    A = torch.zeros((K, lda) if at else (M, lda))
    Bm = torch.zeros((K, ldb))
    if at:
        A[:, :M] = torch.randn(K, M, generator=g)
    else:
        A[:, :K] = torch.randn(M, K, generator=g)
    Bm[:, :N] = torch.randn(K, N, generator=g)
    A, Bm = A.to(gpu_device), Bm.to(gpu_device)
    ldc = r4(N) + 4
    bias = torch.randn(N, generator=g).to(gpu_device)
    Ap, nka = eng.planes("test.pa", A, lda, K, M) if at else eng.planes("test.pa", A, lda, M, K)
    Bp, nkb = eng.planes("test.pb", Bm, ldb, K, N)
    outs = []
    eng.pass_skip = {"t": 1}
    try:
        for k16 in ("1", "1", "0"):
            monkeypatch.setenv("LFI_PGEMM_16T", k16)
            Cm = torch.full((M, ldc), 7.0, device=gpu_device)
            eng.gemm_planes(M, N, K, Ap, nka, Bp, nkb, Cm, ldc, bias=bias, act=1, a_fmt=int(at), b_fmt=1, splitk=splitk, tile=tile,
                            cls="t")
            outs.append(Cm)
    finally:
        eng.pass_skip = {}
    torch.cuda.synchronize()
    C1, C1b, C0 = outs
    Ad = (A[:, :M].t() if at else A[:, :K]).bfloat16().double()          # the A operand as the kernel sees it
    ref = torch.nn.functional.leaky_relu(Ad @ Bm[:, :N].double() + bias.double(), 0.01)
    assert bool((C1[:, N:] == 7.0).all()), "wrote outside the N columns"
    assert torch.equal(C1, C1b)
    assert rel_err(C1[:, :N], ref) < 3e-5
    assert rel_err(C1[:, :N], C0[:, :N]) < 2e-6


@pytest.mark.parametrize("M,N,K,batch", [(700, 512, 96, 1), (14336 // 8, 128, 384, 4), (333, 96, 64, 2)])
def test_gemm_planes_emits_its_result_as_planes(eng, gpu_device, M, N, K, batch):
    """Plane output of lfi_gemm_planes: the result leaves the epilogue as operand planes - bit for bit what lfi_planes_from_f32
    makes of the fp32 result (zero padding included) - with the fp32 store on or off, hi + lo or hi only, either tile shape, batch
    entries side by side in C's columns (the in-place dpre product, glow/models.py:187-190 backward), the act-2 operand read from
    the hi plane of another product's planes, and per-pass column sums (the bias gradient). And the emitted planes serve a
    following product in BOTH uses: by rows (sum over the columns) and transposed (sum over the rows)."""
    g = torch.Generator().manual_seed(M + N + K)
    Ncols = batch * N
    ldc = Ncols + 32
    A = torch.randn(M, batch * K, generator=g).to(gpu_device)        # batch entry b: columns [b K, (b + 1) K)
    W = torch.randn(batch * N, K, generator=g).to(gpu_device)        # batch entry b: rows [b N, (b + 1) N)
    Gm = torch.randn(M, ldc, generator=g).to(gpu_device)             # the LeakyReLU input whose sign gates the result
    Gp, nkg = eng.planes("test.pg", Gm, ldc, M, Ncols)
    Ap, nka = eng.planes("test.pa", A, batch * K, M, batch * K)
    Wp, nkw = eng.planes("test.pw", W, K, batch * N, K)
    ref_c = torch.empty(M, ldc, device=gpu_device).fill_(3.0)
    kw = dict(act=2, batch=batch, a_stride=(K // 16) * 1024, b_stride=(N // 32) * nkw * 1024, sC=N)
    eng.gemm_planes(M, N, K, Ap, nka, Wp, nkw, ref_c, ldc, G=Gm, ldg=ldc, sG=N, **kw)              # fp32 reference path
    want_r, nkr = eng.planes("test.wr", ref_c, ldc, M, Ncols)
    want_r = want_r.clone()
    n_r = eng.L.lfi_planes_elems(M, Ncols)
    X = torch.randn(Ncols, 96, generator=g).to(gpu_device)           # for the row-use consumer: result (M x Ncols) times X
    Y = torch.randn(M, 160, generator=g).to(gpu_device)              # for the transposed-use consumer: result^T (Ncols x M) times Y
    Xp, nkx = eng.planes("test.px", X.t().contiguous(), Ncols, 96, Ncols)
    Yp, nky = eng.planes("test.py", Y, 160, M, 160)
    for store, hi_only, tile in ((True, False, 1), (False, False, 1), (False, True, 2), (True, False, 2)):
        Cr = torch.full((want_r.numel(),), float("nan"), dtype=torch.bfloat16, device=gpu_device)
        out = torch.empty(M, ldc, device=gpu_device).fill_(3.0)
        sums = torch.zeros(Ncols, device=gpu_device)
        done = eng.gemm_planes(M, N, K, Ap, nka, Wp, nkw, out, ldc, Gr=Gp, gr_nkt=nkg, store=store, Cr=Cr, cr_nkt=nkr,
                               colsum_into=sums, hi_only=hi_only, tile=tile, **kw)
        torch.cuda.synchronize()
        assert done
        if store:
            assert torch.equal(out, ref_c)
        else:
            assert bool((out == 3.0).all())
        npl = 1 if hi_only else 2            # hi_only: the lo blocks stay as they were (NaN here)
        r_blocks = Cr[:n_r].view(-1, nkr, 2, 512)[:(M + 31) // 32]
        assert torch.equal(r_blocks[:, :, :npl].view(torch.int16),
                           want_r[:n_r].view(-1, nkr, 2, 512)[:(M + 31) // 32][:, :, :npl].view(torch.int16))
        if hi_only:
            assert bool(torch.isnan(r_blocks[:, :, 1].float()).all())
        assert rel_err(sums, ref_c[:, :Ncols].double().sum(0)) < 1e-5
        if not hi_only:
            o1 = torch.zeros(M, 96, device=gpu_device)
            eng.gemm_planes(M, 96, Ncols, Cr, nkr, Xp, nkx, o1, 96)                                   # by rows
            o2 = torch.zeros(Ncols, 160, device=gpu_device)
            eng.gemm_planes(Ncols, 160, M, Cr, nkr, Yp, nky, o2, 160, a_fmt=1, b_fmt=1)               # transposed
            torch.cuda.synchronize()
            rc = ref_c[:, :Ncols].double()
            assert rel_err(o1, rc @ X.double()) < 3e-5 and rel_err(o2, rc.t() @ Y.double()) < 3e-5


@pytest.mark.parametrize("M,N,K", [(700, 512, 96), (513, 544, 64), (1792, 8192, 896), (40, 48, 32)])
@pytest.mark.parametrize("hi_only", [False, True])
def test_gemm_planes_direct_plane_epilogue(eng, gpu_device, monkeypatch, M, N, K, hi_only):
    """The cond_transform forward product's epilogue (planes out, bias + LeakyReLU, no fp32 rows) without the LDS round trip:
    the 16 x 16 x 32 kernel computes the product transposed so that a lane holds four consecutive columns of a row and writes
    8 bytes per plane itself (gemm_epilogue_direct16). Same products, same sums: the emitted planes equal the through-LDS
    epilogue's (LFI_PGEMM_DIRECT=0) bit for bit - ragged M, N not a multiple of the tile, zero rows / columns inside the last
    blocks, hi planes only - and what lfi_planes_from_f32 makes of the fp32 result."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(gpu_device)
    W = torch.randn(N, K, generator=g).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    Ap, nka = eng.planes("test.pa", A, K, M, K)
    Wp, nkw = eng.planes("test.pw", W, K, N, K)
    ref = torch.empty(M, N, device=gpu_device)
    eng.gemm_planes(M, N, K, Ap, nka, Wp, nkw, ref, N, bias=bias, act=1)
    want, nkr = eng.planes("test.wr", ref, N, M, N)
    want = want.clone()
    n_el = eng.L.lfi_planes_elems(M, N)
    outs = []
    for direct in ("1", "0"):
        monkeypatch.setenv("LFI_PGEMM_DIRECT", direct)
        Cr = torch.full((want.numel(),), float("nan"), dtype=torch.bfloat16, device=gpu_device)
        eng.gemm_planes(M, N, K, Ap, nka, Wp, nkw, None, N, bias=bias, act=1, store=False, Cr=Cr, cr_nkt=nkr, hi_only=hi_only)
        torch.cuda.synchronize()
        outs.append(Cr)
    a, b = outs
    nblk = (M + 31) // 32 * nkr                       # blocks of the row tiles that exist
    av = a[:nblk * 1024].view(-1, 2, 512).view(torch.int16)
    bv = b[:nblk * 1024].view(-1, 2, 512).view(torch.int16)
    wv = want[:nblk * 1024].view(-1, 2, 512).view(torch.int16)
    npl = 1 if hi_only else 2
    assert torch.equal(av[:, :npl], bv[:, :npl])
    assert torch.equal(av[:, :npl], wv[:, :npl])
    assert n_el >= nblk * 1024


@pytest.mark.parametrize("M,N,K,batch", [(700, 512, 384, 1), (1792, 512, 384, 4), (333, 64, 96, 2)])
def test_gemm_planes_direct_epilogue_of_the_dpre_product(eng, gpu_device, monkeypatch, M, N, K, batch):
    """The in-place dpre product's epilogue without the LDS round trip (gemm_epilogue_direct16<4>; glow/models.py:187-190 backward):
    two products, A by rows, B transposed, LeakyReLU-gradient mask read from the hi plane of the planes it overwrites, hi planes
    out, per-wave column sums (the cond_transform bias gradient). Against the through-LDS epilogue (LFI_PGEMM_DIRECT=0): planes
    bit for bit (same products, same sums), column sums to fp32 rounding (another order over the rows)."""
    g = torch.Generator().manual_seed(M + N + K + batch)
    Ncols = batch * N
    A = torch.randn(M, batch * K, generator=g).to(gpu_device)          # dgi: batch entry b = columns [b K, (b + 1) K)
    W = torch.randn(batch * K, N, generator=g).to(gpu_device)          # W_c[b]: K x N, its ROWS are the contraction index
    Gm = torch.randn(M, Ncols, generator=g).to(gpu_device)             # c: its sign gates the result; overwritten by it
    Ap, nka = eng.planes("test.pa", A, batch * K, M, batch * K)
    Wp, nkw = eng.planes("test.pw", W, N, batch * K, N)
    outs = []
    eng.pass_skip = {"t": 1}
    try:
        for direct in ("1", "0"):
            monkeypatch.setenv("LFI_PGEMM_DIRECT", direct)
            Gp, nkg = eng.planes("test.pg", Gm, Ncols, M, Ncols)           # fresh planes of c every time: the product runs in place
            sums = torch.zeros(Ncols, device=gpu_device)
            done = eng.gemm_planes(M, N, K, Ap, nka, Wp, nkw, None, Ncols, act=2, slope=0.01, batch=batch, b_fmt=1,
                                   a_stride=(K // 16) * 1024, b_stride=(K // 32) * nkw * 1024, sC=N, store=False,
                                   Gr=Gp, gr_nkt=nkg, Cr=Gp, cr_nkt=nkg, hi_only=True, colsum_into=sums, cls="t")
            torch.cuda.synchronize()
            assert done
            outs.append((Gp.clone(), sums))
    finally:
        eng.pass_skip = {}
    (pa, sa), (pb, sb) = outs
    nblk = (M + 31) // 32 * nkg
    assert torch.equal(pa[:nblk * 1024].view(-1, 2, 512)[:, 0].view(torch.int16), pb[:nblk * 1024].view(-1, 2, 512)[:, 0].view(torch.int16))
    ref = torch.zeros(M, Ncols, dtype=torch.float64)
    for b in range(batch):
        ref[:, b * N:(b + 1) * N] = A[:, b * K:(b + 1) * K].bfloat16().double().cpu() @ W[b * K:(b + 1) * K].double().cpu()
    ref = torch.where(Gm.cpu().bfloat16().double() > 0, ref, 0.01 * ref)
    assert rel_err(sa, ref.sum(0)) < 2e-4 and rel_err(sb, ref.sum(0)) < 2e-4
    assert float((sa - sb).abs().max()) <= 1e-5 * max(1.0, float(sb.abs().max()))


@pytest.mark.parametrize("M,N,K,splitk", [(768, 256, 9000, 4), (384, 128, 2049, 1), (768, 52, 4100, 3)])
def test_gemm_with_a_bf16_operand(eng, gpu_device, M, N, K, splitk):
    """lfi_gemm_desc.a_bf16 (the window encoders' bf16 gradient stash as the A operand of dW_hh = dgh^T hseq, glow/models.py:60-64
    autograd): bit-identical to the three-product kernel with skip bit 0 on the fp32 form of the same bf16 values."""
    g = torch.Generator().manual_seed(M + K)
    A16 = torch.randn(K, M, generator=g).to(torch.bfloat16).to(gpu_device)       # mn-contiguous: (K x M)
    Bm = torch.randn(K, N + 4, generator=g).to(gpu_device)
    C1 = torch.zeros(M, N, device=gpu_device)
    C2 = torch.zeros(M, N, device=gpu_device)
    eng.precision = 0x11
    eng.pass_skip = {"t": 1}
    try:
        eng.gemm(M, N, K, A16, M, 0, Bm, N + 4, 0, C1, N, splitk=splitk, cls="t", a_bf16=True)
        eng.gemm(M, N, K, A16.float(), M, 0, Bm, N + 4, 0, C2, N, splitk=splitk, cls="t")
    finally:
        eng.precision = 0
        eng.pass_skip = {}
    torch.cuda.synchronize()
    assert torch.equal(C1, C2)
    assert rel_err(C1, A16.double().t() @ Bm[:, :N].double()) < 1e-4      # B's lo x A's hi is kept: only the dropped b_lo a_lo term


@pytest.mark.parametrize("pin,M", [(0x11, 700), (0x21, 300), (0x11, 256)])
def test_gemm_epilogue_column_sums(eng, gpu_device, pin, M):
    """lfi_gemm_desc.colsum_part: the wide epilogue leaves per-(row tile, pass) column sums of the stored result (the in-place
    dpre product, batched side by side in C, with its LeakyReLU-gradient gate); summed they are the bias gradient."""
    g = torch.Generator().manual_seed(M + pin)
    D, G, Ks = 128, 96, 3
    c = torch.randn(M, Ks * D, generator=g).to(gpu_device)
    dgi = torch.randn(Ks, M, G, generator=g).to(gpu_device)
    wc = torch.randn(Ks, G, D, generator=g).to(gpu_device)
    ref = torch.stack([(dgi[k].double() @ wc[k].double()) * torch.where(c[:, k * D:(k + 1) * D] > 0, 1.0, 0.01).double()
                       for k in range(Ks)], dim=1).reshape(M, Ks * D)
    sums = torch.full((Ks * D,), 7.0, device=gpu_device)
    eng.precision = pin
    try:
        done = eng.gemm(M, D, G, dgi, G, 1, wc, D, 0, c, Ks * D, act=2, slope=0.01, G=c, ldg=Ks * D, batch=Ks, sA=M * G, sB=G * D,
                        sC=D, sG=D, colsum_into=sums)
    finally:
        eng.precision = 0
    torch.cuda.synchronize()
    assert done
    assert rel_err(c, ref) < 3e-5
    assert rel_err(sums, c.double().sum(0)) < 1e-5
    eng.precision = 0   # exact-f32 kernels have no such epilogue: the caller is told to sum C itself
    assert eng.gemm(64, D, G, dgi, G, 1, wc, D, 0, c, Ks * D, colsum_into=sums) is False


def test_gemm_in_place_leaky_grad_epilogue(eng, gpu_device):
    """act = 2 with G == C (the in-place dpre product of the backward pass), batched, through the wide epilogue."""
    g = torch.Generator().manual_seed(11)
    F, D, G, Ks = 300, 128, 96, 3
    dgi = torch.randn(Ks, F, G, generator=g).to(gpu_device)
    wc = torch.randn(Ks, G, D, generator=g).to(gpu_device)
    c = torch.randn(F, Ks * D, generator=g).to(gpu_device)
    ref = c.double().clone()
    for k in range(Ks):
        prod = dgi[k].double() @ wc[k].double()
        blk = ref[:, k * D:(k + 1) * D]
        ref[:, k * D:(k + 1) * D] = torch.where(blk > 0, prod, 0.01 * prod)
    for prec in (0, 1, 0x11):
        cc = c.clone()
        eng.precision = prec
        try:
            eng.gemm(F, D, G, dgi, G, 1, wc, D, 0, cc, Ks * D, act=2, slope=0.01, G=cc, ldg=Ks * D, batch=Ks, sA=F * G,
                     sB=G * D, sC=D, sG=D)
        finally:
            eng.precision = 0
        torch.cuda.synchronize()
        assert rel_err(cc, ref) < (2e-6 if prec == 0 else 3e-5), prec


def test_masked_window_gather_and_leaky_grad(eng, gpu_device):
    import ctypes as C
    from lets_face_it_amd._lib import check
    g = torch.Generator().manual_seed(5)
    B, T, dim, start, hist = 6, 14, 7, 5, 3
    N = T - start
    x = torch.randn(B, T, dim, generator=g).to(gpu_device)
    mask = (torch.rand(N, B, hist, generator=g) < 0.5).float().mul(2.0).to(gpu_device)
    ld = 24
    out = torch.zeros(N * B, ld, device=gpu_device)
    st = torch.cuda.current_stream().cuda_stream
    check(eng.L.lfi_gather_windows(x.data_ptr(), B, T, dim, N, start, hist, 1, mask.data_ptr(), out.data_ptr(), ld, 0, st))
    torch.cuda.synchronize()
    for n in (0, 3, N - 1):
        for b in (0, B - 1):
            win = x[b, start + n - hist + 1:start + n + 1] * mask[n, b].unsqueeze(-1)
            assert torch.equal(out[n * B + b, :hist * dim], win.reshape(-1))
    y = torch.randn(9, 5, generator=g).to(gpu_device)
    d = torch.randn(9, 8, generator=g).to(gpu_device)
    want = d.clone()
    want[:, :5] = torch.where(y > 0, d[:, :5], 0.01 * d[:, :5])
    check(eng.L.lfi_leaky_grad(d.data_ptr(), 8, y.data_ptr(), 5, 9, 5, 0.01, st))
    torch.cuda.synchronize()
    assert torch.equal(d, want)


def test_colsum(eng, gpu_device):
    g = torch.Generator().manual_seed(3)
    X = torch.randn(3, 1000, 90, generator=g).to(gpu_device)
    out = torch.ones(3, 40, device=gpu_device)
    eng.colsum(X, 90, 1000 * 90, 1000, 40, 3, out, 40, scale=0.5, accumulate=1, x_off=10)
    torch.cuda.synchronize()
    ref = 1.0 + 0.5 * X[:, :, 10:50].double().sum(dim=1)
    assert rel_err(out, ref) < 2e-6


def test_partial_chip_stream(eng, gpu_device):
    """lfi_stream_create_partial: a stream that owns n CUs of every XCD. Argument errors; a product launched on it gives the bits it
    gives on the current stream; and it is confined - 896 one-CU-wide workgroups of the window encoder take longer on 4 CUs per
    XCD than on all 32 (the probe that decoded the mask's bit order: tools/probes/cu_mask_map_probe.hip)."""
    from lets_face_it_amd import _lib
    from lets_face_it_amd._lib import EncDesc, check
    L = _lib.lib()
    st = C.c_void_p()
    for bad in (0, -1, 33):
        assert L.lfi_stream_create_partial(bad, C.byref(st)) == -1 and b"CUs per XCD" in L.lfi_last_error()
    assert L.lfi_stream_create_partial(4, None) == -1
    check(L.lfi_stream_create_partial(4, C.byref(st)), "lfi_stream_create_partial")
    assert st.value
    ext = torch.cuda.ExternalStream(st.value, device=gpu_device)
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(gpu_device)   # noqa: E731
    hist, hid, B, T, start = 24, 256, 1024, 80, 24
    N = T - start
    F = N * B
    xp, whh, b_ih, b_hh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2, rnd(3 * hid), rnd(3 * hid)
    hseq = torch.zeros(hist * F * hid, device=gpu_device)
    d = EncDesc(B, T, N, start, hist, hid, 512, 0, 1, 0, 0, 1, 0)
    work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(d))), 1), device=gpu_device)
    outs, ms = [], []
    torch.cuda.synchronize()
    for stream in (torch.cuda.current_stream(gpu_device), ext):
        cond = torch.zeros(F, 512, device=gpu_device)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(3):
                if i == 1:
                    e0.record(stream)
                check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), None,
                                               cond.data_ptr(), None, hseq.data_ptr(), work.data_ptr(), stream.cuda_stream), "fwd")
            e1.record(stream)
        stream.synchronize()
        outs.append(cond)
        ms.append(e0.elapsed_time(e1) / 2)
    assert torch.equal(outs[0], outs[1])
    report("window encoder forward, 896 workgroups: %.2f ms on the whole chip, %.2f ms on a stream with 4 CUs of every XCD" % tuple(ms))
    assert ms[1] > 3.0 * ms[0], ms
    check(L.lfi_stream_destroy(st), "lfi_stream_destroy")
    check(L.lfi_stream_destroy(None), "lfi_stream_destroy(NULL)")


def test_adam_clip_step_matches_oracle(gpu_device):
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.engine import GlowEngine, ModelSpec
    from oracle import seqglow_oracle as oracle
    e = GlowEngine(ModelSpec(Namespace(**Fixture("tiny").hp)), gpu_device)
    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(e.n_params, generator=g)
    gr = torch.randn(e.n_params, generator=g) * 3
    e.params.copy_(p0)
    ps, ms, vs = [p0.double().clone()], [torch.zeros(e.n_params, dtype=torch.float64)], [torch.zeros(e.n_params, dtype=torch.float64)]
    for step in range(1, 4):
        e.grads.copy_(gr * step)
        e.optimizer_step(1e-3, 0.9, 0.9999, 1e-8, clip=20.0, gmul=0.5)
        oracle.adam_clip_step(ps, [gr.double() * step * 0.5], ms, vs, step, 1e-3, 0.9, 0.9999, 1e-8, 20.0)
    torch.cuda.synchronize()
    assert rel_err(e.params, ps[0]) < 1e-6
    e.grads.copy_(gr)
    assert abs(e.grad_norm() - float(gr.double().norm())) < 1e-6 * float(gr.double().norm())


def test_dropout_masks_and_padded_copies(gpu_device):
    """lfi_dropout_masks (nn.Dropout on ones(B, hist), glow/models.py:56-58; all modalities in one launch): values in {0, 1 / keep},
    keep-rate within 5 sigma, reproducible from (seed, call counter), different per call and per seed; lfi_pad_rows: exact
    copy + zero padding."""
    from argparse import Namespace
    from helpers import Fixture
    from lets_face_it_amd.engine import GlowEngine, ModelSpec
    hp = Fixture("mid").hp           # final_model.yaml histories and dropouts (0.6 / 0.5 / 0.3)
    e1, e2 = GlowEngine(ModelSpec(Namespace(**hp)), gpu_device), GlowEngine(ModelSpec(Namespace(**hp)), gpu_device)
    B, N = 64, 56
    a = {k: v.clone() for k, v in e1.draw_masks(B, N, 1234).items()}
    b = {k: v.clone() for k, v in e2.draw_masks(B, N, 1234).items()}
    c = {k: v.clone() for k, v in e1.draw_masks(B, N, 1234).items()}      # second call of e1: another counter
    d = {k: v.clone() for k, v in GlowEngine(ModelSpec(Namespace(**hp)), gpu_device).draw_masks(B, N, 99).items()}
    drops = {e.name: e.dropout for e in e1.spec.encoders if e.dropout > 0}
    assert set(a) == set(drops) and len(a) == 3
    for name, p in drops.items():
        keep = 1.0 - p
        m = a[name]
        assert m.shape == (N, B, hp["Conditioning"][name]["history"])
        on = m != 0
        assert torch.allclose(m[on], torch.full_like(m[on], 1.0 / keep))
        n = m.numel()
        assert abs(float(on.float().mean()) - keep) < 5.0 * (keep * (1 - keep) / n) ** 0.5
        assert torch.equal(m, b[name])                       # same seed, same call number
        assert not torch.equal(m, c[name]) and not torch.equal(m, d[name])
        assert abs(float((on & (c[name] != 0)).float().mean()) - keep * keep) < 0.02   # calls look independent
    from lets_face_it_amd import _lib
    L = _lib.lib()
    x = torch.randn(37, 50, device=gpu_device)
    out = torch.full((37, 52), float("nan"), device=gpu_device)
    _lib.check(L.lfi_pad_rows(x.data_ptr(), 37, 50, 50, out.data_ptr(), 52, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(out[:, :50], x) and bool((out[:, 50:] == 0).all())


@pytest.mark.parametrize("shape", [(256, 80, 24, 24, 256), (16, 30, 7, 5, 128), (5, 19, 3, 2, 64), (3, 12, 11, 12, 256)])
@pytest.mark.parametrize("masked", [True, False])
def test_window_scatter_kernels_match(gpu_device, monkeypatch, shape, masked):
    """lfi_encode_windows_scatter on the bf16 compact gradient stash (the training step's): the sum over the windows a frame
    belongs to - dXp[b T + p] = sum_s mask[w][s] (dgh[s][w][:2 hid], dgi[s][w]) with w = (p - pos0 - s) B + b - against that
    definition in fp64, and the 16-byte-load kernel (several frame rows per workgroup, default) bit for bit against the 8-byte one
    (LFI_ENC_SCATTER16=0): same products, same order over s. Ragged row counts, frames no window reaches (zero rows), more
    history steps than windows. (The input side of nn.GRU's weight gradient, glow/models.py:55-80.)"""
    import ctypes as C
    from lets_face_it_amd import _lib
    from lets_face_it_amd._lib import EncDesc, check
    L = _lib.lib()
    B, T, start, hist, hid = shape
    N = T - start
    F = N * B
    dev = gpu_device
    g = torch.Generator().manual_seed(B + T + hist)
    d = EncDesc(B, T, N, start, hist, hid, 896, 256, 1, 0, 0, 1, 1)
    assert L.lfi_encode_windows_grad_stash_bf16(C.byref(d)) and L.lfi_encode_windows_compact_dgi(C.byref(d))
    dgh = torch.randn(hist, N, B, 3 * hid, generator=g).to(torch.bfloat16).to(dev)
    dgi = torch.randn(hist, N, B, hid, generator=g).to(torch.bfloat16).to(dev)
    mask = ((torch.rand(N, B, hist, generator=g) < 0.6).float() * 1.7).to(dev) if masked else None
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("LFI_ENC_SCATTER16", sw)
        dxp = torch.full((B * T + 1, 3 * hid), 7.0, device=dev)
        check(L.lfi_encode_windows_scatter(C.byref(d), dgi.data_ptr(), dgh.data_ptr(), mask.data_ptr() if masked else None,
                                           dxp.data_ptr(), st), "scatter")
        torch.cuda.synchronize()
        assert bool((dxp[B * T] == 7.0).all()), "wrote past the last frame row"
        outs.append(dxp[:B * T])
    assert torch.equal(outs[0], outs[1])
    D = torch.cat([dgh[..., :2 * hid], dgi], dim=-1).double()       # (hist, N, B, 3 hid)
    ref = torch.zeros(B, T, 3 * hid, dtype=torch.float64, device=dev)
    pos0 = start - hist + 1
    for s_ in range(hist):
        term = D[s_] * (mask[..., s_].double().unsqueeze(-1) if masked else 1.0)
        ref[:, pos0 + s_:pos0 + s_ + N] += term.permute(1, 0, 2)
    err = rel_err(outs[0].view(B, T, 3 * hid), ref)
    report("window scatter %s masked=%s: %.2e vs fp64, two kernels bit-identical" % (shape, masked, err))
    assert err < 2e-6, err


@pytest.mark.parametrize("mod,two,s16,B", [("p2_face", 1, 1, 256), ("p2_face", 0, 0, 256), ("p2_speech", 1, 0, 256), ("p1_speech", 1, 1, 256),
                                          ("p2_face", 1, 1, 250)])   # (250: 14 000 windows - the last 64-window workgroup is ragged)
def test_window_encoder_tilings_match(gpu_device, monkeypatch, mod, two, s16, B):
    """The two tilings of the fused window-encoder recurrence - four waves of 64 windows x 64 hidden units, one workgroup per CU
    (default where the launch fills the chip: every weight fragment feeds two row tiles) and the 32-window row-layout kernels
    (LFI_ENC_R64=0) - at the benchmark's shapes, through the C ABI:
    same products in the same order per window, so features, gate stash, state stash and the gradient stashes are bit-identical;
    the bias gradients (sums over per-workgroup partials, another grouping) agree to rounding. (glow/models.py:55-80, nn.GRU over
    every window.)"""
    import ctypes as C
    from lets_face_it_amd import _lib
    from lets_face_it_amd._lib import EncDesc, check
    L = _lib.lib()
    hist, hid = {"p2_face": (24, 256), "p2_speech": (16, 256), "p1_speech": (2, 128)}[mod]
    T, start = 80, 24
    N = T - start
    F = N * B
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)   # noqa: E731
    xp, whh, b_ih, b_hh = rnd(B * T, 3 * hid), rnd(3 * hid, hid) * 0.2, rnd(3 * hid), rnd(3 * hid)
    mask = ((torch.rand(F, hist, generator=g) < 0.5).float() * 2).to(dev)
    ldc = 896
    dcond = rnd(F, ldc)
    st = torch.cuda.current_stream().cuda_stream
    d = EncDesc(B, T, N, start, hist, hid, ldc, 256, 1, 0, 0, two, s16)
    if s16:
        assert L.lfi_encode_windows_stash_f16_ok(C.byref(d))
    work = torch.zeros(max(int(L.lfi_encode_windows_work_floats(C.byref(d))), 1), device=dev)
    outs = {}
    rows_per_wg = {"r64": 2 * 32 * (4 // ((hid + 63) // 64)), "wide": 32 * (4 // ((hid + 63) // 64))}
    groups = {"r64": 4 // ((hid + 63) // 64), "wide": 4 // ((hid + 63) // 64)}
    rows_per_wg["r64m16"], groups["r64m16"] = rows_per_wg["r64"], groups["r64"]
    rows_per_wg["t16"], groups["t16"] = rows_per_wg["r64"], groups["r64"]
    variant = {}
    for name, env in (("r64", {"LFI_ENC_R64": "1", "LFI_ENC_M16": "0"}), ("r64m16", {"LFI_ENC_R64": "1", "LFI_ENC_M16": "1", "LFI_ENC_T16": "0"}),
                      ("t16", {"LFI_ENC_R64": "1", "LFI_ENC_M16": "1", "LFI_ENC_T16": "2"}), ("wide", {"LFI_ENC_R64": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        variant[name] = int(L.lfi_encode_windows_fwd_variant(C.byref(d), 1, 1))
        cond = torch.zeros(F, ldc, device=dev)
        gates = torch.zeros(hist * F * 4 * hid, device=dev)
        hseq = torch.zeros(hist * F * hid, device=dev)
        dgi = torch.zeros(hist * F * hid, device=dev)
        dgh = torch.zeros(hist * F * 3 * hid, device=dev)
        prow = int(L.lfi_encode_windows_bias_rows(C.byref(d)))
        wgs = -(-F // rows_per_wg[name])
        if wgs >= 128:      # the tiling is only taken where its workgroups still cover the chip
            assert prow == wgs * groups[name], (name, prow, wgs, groups[name])
        part = torch.zeros(prow * 4 * hid, device=dev)
        check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), mask.data_ptr(),
                                       cond.data_ptr(), gates.data_ptr(), hseq.data_ptr(), work.data_ptr(), st), "fwd")
        check(L.lfi_encode_windows_bwd(C.byref(d), dcond.data_ptr(), ldc, whh.data_ptr(), gates.data_ptr(), hseq.data_ptr(),
                                       dgi.data_ptr(), dgh.data_ptr(), part.data_ptr(), work.data_ptr(), st), "bwd")
        gbi, gbh = torch.zeros(3 * hid, device=dev), torch.zeros(3 * hid, device=dev)
        check(L.lfi_encode_windows_bias_grads(part.data_ptr(), prow, hid, gbi.data_ptr(), gbh.data_ptr(), st), "bias")
        torch.cuda.synchronize()
        outs[name] = (cond, gates, hseq, dgi, dgh, gbi, gbh)
    b = outs["wide"]
    assert torch.isfinite(b[0]).all() and float(b[0].abs().max()) > 0
    for name in ("r64",):
        a = outs[name]
        for i, what in enumerate(("features", "gate stash", "state stash", "dgi", "dgh")):
            assert torch.equal(a[i], b[i]), (name, what)
        for i in (5, 6):
            assert float((a[i] - b[i]).abs().max()) <= 2e-5 * max(1.0, float(b[i].abs().max())), name
    # the forward 64-window kernel on v_mfma_f32_16x16x32_bf16 (hid a multiple of 256): the same three products per k, summed in
    # another order - fp32 rounding on the state, one fp16 / bf16 rounding step on what is stashed in those types
    a = outs["r64m16"]
    if hid % 256 == 0 and -(-F // rows_per_wg["r64"]) >= 128:
        assert not torch.equal(a[0], b[0]), "the 16 x 16 x 32 kernel did not run"
        assert variant["r64m16"] == 4 and variant["t16"] == 5 and variant["r64"] == 3 and variant["wide"] == 2, variant
        # round 5: the same 16 x 16 x 32 products taken transposed (weights as the A operand) with the gate epilogue issued under the
        # next unit tile's MFMAs: same products, same order, same gate expressions - everything the forward pass writes is
        # bit-identical, and so is the BPTT that runs on those stashes
        for tn in ("t16",):
            t = outs[tn]
            for i, what in enumerate(("features", "gate stash", "state stash", "dgi", "dgh")):
                assert torch.equal(t[i], a[i]), (tn + " vs r64m16", what, float((t[i] - a[i]).abs().max()))
        # the default: taken where nothing is stashed (inference / validation forward), not in training
        monkeypatch.delenv("LFI_ENC_T16")
        monkeypatch.setenv("LFI_ENC_R64", "1")
        monkeypatch.setenv("LFI_ENC_M16", "1")
        assert int(L.lfi_encode_windows_fwd_variant(C.byref(d), 1, 0)) == 5 and int(L.lfi_encode_windows_fwd_variant(C.byref(d), 1, 1)) == 4
        plain = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("LFI_ENC_T16", mode)
            for masked in (True, False):
                cond = torch.zeros(F, ldc, device=dev)
                check(L.lfi_encode_windows_fwd(C.byref(d), xp.data_ptr(), whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                                               mask.data_ptr() if masked else None, cond.data_ptr(), None, None, work.data_ptr(), st), "fwd")
                torch.cuda.synchronize()
                plain[mode, masked] = cond
        assert torch.equal(plain["1", True], a[0]) and torch.equal(plain["0", True], a[0])
        assert torch.equal(plain["1", False], plain["0", False]) and not torch.equal(plain["1", False], a[0])
    else:
        assert variant["t16"] == variant["r64m16"], variant
    n_g, n_dgi, n_dgh = hist * F * 4 * hid, hist * F * hid, hist * F * 3 * hid
    views = [(a[0], b[0], 5e-6), (a[2], b[2], 5e-6)]
    views.append((a[1].view(torch.float16)[:n_g].float(), b[1].view(torch.float16)[:n_g].float(), 2e-3) if s16 else (a[1], b[1], 5e-6))
    if two:
        views += [(a[3].view(torch.bfloat16)[:n_dgi].float(), b[3].view(torch.bfloat16)[:n_dgi].float(), 1e-2),
                  (a[4].view(torch.bfloat16)[:n_dgh].float(), b[4].view(torch.bfloat16)[:n_dgh].float(), 1e-2)]
    else:
        views += [(a[3], b[3], 1e-4), (a[4], b[4], 1e-4)]
    for x, y, tol in views:
        assert float((x - y).abs().max()) <= tol * max(1.0, float(y.abs().max())), tol
    for i in (5, 6):
        assert float((a[i] - b[i]).abs().max()) <= 1e-3 * max(1.0, float(b[i].abs().max()))
