"""LBDRNModel / LBDRNLoss with the reference's constructor, state_dict keys and initial weights,
whose forward runs in liblbdrn_hip (ref LBDRNmodel.py:58-82, LBDRNloss.py:8-11).

Initialisation consumes the global torch CPU generator in the reference's order, layer by layer:
nn.Linear's own reset (weight then bias), then uniform_(weight), uniform_(bias) with bound
1/dim_in for the first layer and sqrt(6/dim_in)/w0 for the others (ref LBDRNmodel.py:32-36), so
a model built under torch.manual_seed(s) has bit-identical parameters and leaves the generator
where the reference leaves it (which fixes every later permutation draw).
"""
import math

import torch
from torch import nn

from . import ops


class Sine(nn.Module):
    """sin(w0 * x) (ref LBDRNmodel.py:7-13).  Inside LBDRNModel the HIP kernels apply it fused."""

    def __init__(self, w0=1.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


class SirenLayer(nn.Module):
    """Linear + activation holder; owns `linear` so that state_dict keys match the reference."""

    def __init__(self, dim_in, dim_out, w0=30.0, c=6.0, is_first=False, use_bias=True, activation=None):
        super().__init__()
        self.dim_in, self.is_first, self.w0 = dim_in, is_first, w0
        self.linear = nn.Linear(dim_in, dim_out, bias=use_bias)
        bound = 1.0 / dim_in if is_first else math.sqrt(c / dim_in) / w0
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)
            if use_bias:
                self.linear.bias.uniform_(-bound, bound)
        self.custom_activation = activation is not None
        self.activation = activation if activation is not None else Sine(w0)


class LBDRNModel(nn.Module):
    def __init__(self, dim_in, dim_hidden, dim_out=4, num_layers=1, w0=30.0, w0_initial=30.0,
                 use_bias=True, activation=None, final_activation=None):
        super().__init__()
        self.dim_in, self.dim_hidden, self.dim_out, self.num_layers = dim_in, dim_hidden, dim_out, num_layers
        self.use_bias = use_bias
        self.net = nn.Sequential(*[
            SirenLayer(dim_in if i == 0 else dim_hidden, dim_hidden,
                       w0=w0_initial if i == 0 else w0, is_first=(i == 0), use_bias=use_bias,
                       activation=activation)
            for i in range(num_layers)])
        self.last_layer = SirenLayer(dim_hidden, dim_out, w0=w0, use_bias=use_bias,
                                     activation=nn.Sigmoid() if final_activation is None else final_activation)
        # What liblbdrn_hip implements: the reference's default (Sine, w0 = 30) and the one alternative it names in its
        # sources (ref encode.py:75, decode.py:108: `activation=torch.nn.ReLU()`), Sigmoid head, biases on.
        relu = type(activation) is nn.ReLU
        self.hip_act = ops.ACT_RELU if relu else ops.ACT_SINE
        self._fused_ok = ((activation is None and w0 == 30.0 and w0_initial == 30.0 or relu)
                          and final_activation is None and use_bias and num_layers >= 1)

    def flat_parameters(self):
        """state_dict order, one float32 vector (ref encode.py:123-128)."""
        return torch.cat([v.detach().reshape(-1).float() for v in self.state_dict().values()])

    def hip_net(self):
        return ops.make_net(self.dim_in, self.dim_hidden, self.dim_out, self.num_layers, self.hip_act)

    def forward(self, x):
        if not self._fused_ok:
            raise NotImplementedError(
                "liblbdrn_hip implements the reference configuration (Sine w0=30 hidden layers, Sigmoid head, biases "
                "on) and activation=torch.nn.ReLU(); other activations have no HIP kernel")
        if not x.is_cuda:
            raise ops._lib.LbdrnError("LBDRNModel.forward needs a device tensor: this package has no CPU path")
        flat = self.flat_parameters().to(x.device)
        return ops.forward(self.hip_net(), flat, x)


class LBDRNLoss(nn.Module):
    """Mean squared error over batch x bands (ref LBDRNloss.py:8-11).  The fused training step
    computes the same quantity inside the kernel; this module exists for API parity."""

    def forward(self, y_pred, y):
        return nn.functional.mse_loss(y_pred, y)
