"""Parameter containers with the names, shapes, initialisation and state-dict keys of
glow_pytorch/glow/modules.py (ActNorm2d :10-80, LinearZeros :83-95, InvertibleConv1x1 :122-194,
GaussianDiag :197-235).

Inside SeqGlow these modules only HOLD parameters: the fused engine reads them through one flat buffer and never
calls their forward(). The forward() methods below exist for the stand-alone module API (what the reference's
test_modules.py exercises) and are thin torch expressions of the same formulas — they are not the product path.
"""
import math

import numpy as np
import scipy.linalg
import torch
import torch.nn as nn


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
        if not self.training:
            return
        with torch.no_grad():
            bias = -input.mean(dim=0, keepdim=True)
            var = ((input + bias) ** 2).mean(dim=0, keepdim=True)
            self.bias.data.copy_(bias)
            self.logs.data.copy_(torch.log(self.scale / (var.sqrt() + 1e-6)))
            self.inited = True

    def forward(self, input, logdet=None, reverse=False):
        if not self.inited:
            self.initialize_parameters(input)
        dlogdet = self.logs.sum() * input.size(1)
        if not reverse:
            out = (input + self.bias) * torch.exp(self.logs)
        else:
            out = input * torch.exp(-self.logs) - self.bias
            dlogdet = -dlogdet
        if logdet is not None:
            logdet = logdet + dlogdet
        return out, logdet


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
        C = self.w_shape[0]
        if not self.LU:
            dlogdet = torch.slogdet(self.weight)[1] * input.size(1)
            w = self.weight if not reverse else torch.inverse(self.weight.double()).float()
            return w, dlogdet
        mask = torch.tril(torch.ones(C, C, device=self.l.device, dtype=self.l.dtype), -1)
        lower = self.l * mask + torch.eye(C, device=self.l.device, dtype=self.l.dtype)
        upper = self.u * mask.t() + torch.diag(self.sign_s * torch.exp(self.log_s))
        dlogdet = self.log_s.sum() * input.size(1)
        if not reverse:
            return self.p @ (lower @ upper), dlogdet
        li = torch.inverse(lower.double()).float()
        ui = torch.inverse(upper.double()).float()
        return ui @ (li @ self.p.t()), dlogdet

    def forward(self, input, logdet=None, reverse=False):
        w, dlogdet = self.get_weight(input, reverse)
        z = input @ w
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
