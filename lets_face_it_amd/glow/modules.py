"""Parameter containers with the names, shapes, initialisation and state-dict keys of
glow_pytorch/glow/modules.py (ActNorm2d :10-80, LinearZeros :83-95, InvertibleConv1x1 :122-194,
GaussianDiag :197-235).

Inside SeqGlow these modules only HOLD parameters: the fused engine reads them through one flat buffer and never
calls their forward(). The forward() methods below exist for the stand-alone module API (what the reference's
test_modules.py:9-29 exercises): they run on the GPU through the same library (lfi_actnorm_forward, lfi_actnorm_init_*,
lfi_invconv_weights + lfi_gemm_f32); a CPU tensor raises, as everywhere else in this package. Inference-only (no autograd).
"""
import math

import numpy as np
import scipy.linalg
import ctypes as C_

import torch
import torch.nn as nn

from .. import _lib


class ActNorm2d(nn.Module):
    """Per-channel affine with data-dependent init; log-det is multiplied by input.size(1) (modules.py:62)."""

    def __init__(self, num_features, scale=1.0):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, num_features))
        self.logs = nn.Parameter(torch.zeros(1, num_features))
        self.num_features = num_features
        self.scale = float(scale)
        self.inited = False  # plain attribute, not saved with the state dict (modules.py:27)

    def initialize_parameters(self, input):
        """Data-dependent init (modules.py:32-43): bias = -mean, logs = log(scale / (std + 1e-6)) over the batch."""
        if not self.training:
            return
        L, x, st = _gpu_call(input, "ActNorm2d")
        B, C = x.shape
        sums = torch.zeros(2 * C, dtype=torch.float64, device=x.device)
        _lib.check(L.lfi_actnorm_init_stats(x.data_ptr(), B, C, sums.data_ptr(), st), "lfi_actnorm_init_stats")
        bias, logs = self.bias.data.contiguous(), self.logs.data.contiguous()
        _lib.check(L.lfi_actnorm_init_apply(sums.data_ptr(), float(B), C, self.scale, bias.data_ptr(), logs.data_ptr(), st),
                   "lfi_actnorm_init_apply")
        self.bias.data.copy_(bias)
        self.logs.data.copy_(logs)
        self.inited = True

    def forward(self, input, logdet=None, reverse=False):
        if not self.inited:
            self.initialize_parameters(input)
        L, x, st = _gpu_call(input, "ActNorm2d")
        B, C = x.shape
        out = torch.empty_like(x)
        dlogdet = torch.empty(1, dtype=torch.float32, device=x.device)
        bias, logs = self.bias.data.contiguous(), self.logs.data.contiguous()
        _lib.check(L.lfi_actnorm_forward(x.data_ptr(), B, C, bias.data_ptr(), logs.data_ptr(), 1 if reverse else 0,
                                         out.data_ptr(), dlogdet.data_ptr(), st), "lfi_actnorm_forward")
        if logdet is not None:
            logdet = logdet + dlogdet[0]      # input.size(1) = C is already in it (modules.py:62)
        return out, logdet


def _gpu_call(input, who):
    """(library, contiguous fp32 (B, C) view of the input, stream) for a stand-alone module call; CPU tensors raise."""
    if not input.is_cuda:
        raise RuntimeError("%s (lets_face_it_amd) runs on the GPU only; got a %s tensor (there is no CPU fallback by design)"
                           % (who, input.device.type))
    if input.dim() != 2:
        raise ValueError("%s: expected a (batch, channels) tensor, got %s" % (who, tuple(input.shape)))
    return _lib.lib(), input.detach().float().contiguous(), torch.cuda.current_stream().cuda_stream


class LinearZeros(nn.Linear):
    """Zero-initialised linear layer with a learned output log-scale, factor 3 (modules.py:83-95)."""

    def __init__(self, in_channels, out_channels, logscale_factor=3):
        super().__init__(in_channels, out_channels)  # draws from the torch RNG like the reference, then zeroed
        self.logscale_factor = logscale_factor
        self.logs = nn.Parameter(torch.zeros(out_channels))
        self.weight.data.zero_()
        self.bias.data.zero_()

    def forward(self, input):
        return super().forward(input) * torch.exp(self.logs * self.logscale_factor)


class InvertibleConv1x1(nn.Module):
    """C x C mixing matrix, dense or LU-parameterised W = P (L*mask + I)(U*mask^T + diag(sign_s exp(log_s)))."""

    def __init__(self, num_channels, LU_decomposed=False):
        super().__init__()
        shape = [num_channels, num_channels]
        w0 = np.linalg.qr(np.random.randn(*shape))[0].astype(np.float32)  # numpy RNG, as modules.py:126
        self.w_shape = shape
        self.LU = LU_decomposed
        if not LU_decomposed:
            self.weight = nn.Parameter(torch.tensor(w0))
            return
        perm, lower, upper = scipy.linalg.lu(w0)
        diag = np.diag(upper)
        self.register_buffer("p", torch.tensor(perm.astype(np.float32)))
        self.register_buffer("sign_s", torch.tensor(np.sign(diag).astype(np.float32)))
        self.l = nn.Parameter(torch.tensor(lower.astype(np.float32)))
        self.log_s = nn.Parameter(torch.tensor(np.log(np.abs(diag)).astype(np.float32)))
        self.u = nn.Parameter(torch.tensor(np.triu(upper, k=1).astype(np.float32)))

    def get_weight(self, input, reverse):
        """-> (W or its fp64-inverted, fp32-cast reverse weight, dlogdet = log|det W| * input.size(1))   (modules.py:147-178)"""
        L, x, st = _gpu_call(input, "InvertibleConv1x1")
        C = self.w_shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        W, Winv, dlogdet = torch.empty(C, C, **f32), torch.empty(C, C, **f32), torch.empty(1, **f32)
        work = torch.empty(L.lfi_invconv_work_floats(C) + 2, **f32)

        def p(t):
            return t.detach().to(**f32).contiguous()

        if self.LU:
            args = [p(self.l), p(self.u), p(self.log_s), p(self.p), p(self.sign_s), None]
        else:
            args = [None, None, None, None, None, p(self.weight)]
        _lib.check(L.lfi_invconv_weights(C, *[_lib.ptr(a) for a in args], 1 if reverse else 0, W.data_ptr(), Winv.data_ptr(),
                                         dlogdet.data_ptr(), work.data_ptr(), st), "lfi_invconv_weights")
        # the kernel's log-det already carries the x C of modules.py:171 (input.size(1) == C for this repo's (B, C) tensors)
        return (Winv if reverse else W), dlogdet[0]

    def forward(self, input, logdet=None, reverse=False):
        w, dlogdet = self.get_weight(input, reverse)
        L, x, st = _gpu_call(input, "InvertibleConv1x1")
        B, C = x.shape
        z = torch.empty_like(x)
        g = _lib.GemmDesc()
        g.M, g.N, g.K = B, C, C
        g.A, g.lda, g.a_kcontig = x.data_ptr(), C, 1          # z = x @ w: A (B x C) k-contiguous
        g.B, g.ldb, g.b_kcontig = w.data_ptr(), C, 0          # B element (k, n) at w[k * C + n]
        g.C, g.ldc = z.data_ptr(), C
        g.batch, g.splitk = 1, 1
        _lib.check(L.lfi_gemm_f32(C_.byref(g), st), "lfi_gemm_f32")
        if logdet is not None:
            logdet = logdet - dlogdet if reverse else logdet + dlogdet
        return z, logdet


class GaussianDiag:
    """Standard-normal prior helpers (modules.py:197-235)."""

    Log2PI = float(math.log(2 * math.pi))

    @staticmethod
    def likelihood_simplified(x):
        return -0.5 * (x ** 2 + GaussianDiag.Log2PI)

    @staticmethod
    def logp_simplified(x):
        return GaussianDiag.likelihood_simplified(x).sum(dim=1)

    @staticmethod
    def likelihood(mean, logs, x):
        return -0.5 * (logs * 2.0 + (x - mean) ** 2 / torch.exp(logs * 2.0) + GaussianDiag.Log2PI)

    @staticmethod
    def logp(mean, logs, x):
        return GaussianDiag.likelihood(mean, logs, x).sum(dim=1)

    @staticmethod
    def sample(output_shape, eps_std=1):
        """output_shape is a TENSOR whose shape/dtype/device is used, as in the reference (modules.py:231-235)."""
        return torch.normal(mean=torch.zeros_like(output_shape), std=torch.ones_like(output_shape) * eps_std)
