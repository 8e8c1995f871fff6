"""Activations and basic blocks of the layer API (counterpart of reference mxgraph/layers/common.py:32-57 plus
the Gluon `nn.Dense` / `nn.Dropout` blocks the reference layers are assembled from).

An activation object knows whether the native kernels can FUSE it into the producing GEMM / gather epilogue
(`fused` = 'leaky' | 'relu' | 'sigmoid' | 'tanh' | None for identity); the other reference activations
(elu, softrelu, softsign) run as a separate elementwise step.
"""
import math

import torch
from torch import nn

from .._native import functional as SF


class Activation(nn.Module):
    def __init__(self, name, fused=None, slope=0.1):
        super().__init__()
        self.name, self.fused, self.slope = name, fused, slope

    @property
    def fusable(self):
        return self.name in ("identity", "leaky", "relu", "sigmoid", "tanh")

    def forward(self, x, native=False):
        """native=True (opt-in; the layers pass it for the activation that follows the all-reduce of a node-partitioned
        aggregate): one native pass each way (sg_act_hip / sg_act_bwd_hip) whose derivative is evaluated from the OUTPUT --
        no double backward, not traceable by torch.func / torch.compile.  The default is plain torch ops, so the module
        behaves like any other nn.Module under gradgrad, functional transforms and tracing."""
        n = self.name
        if n == "identity":
            return x
        if native and self.fused is not None and x.is_cuda and x.dtype == torch.float32:
            return SF.activation(x, self.fused, self.slope)
        if n == "leaky":
            return torch.where(x > 0, x, self.slope * x)
        if n == "relu":
            return torch.relu(x)
        if n == "sigmoid":
            return torch.sigmoid(x)
        if n == "tanh":
            return torch.tanh(x)
        if n == "softrelu":
            return torch.nn.functional.softplus(x)
        if n == "softsign":
            return x / (1 + x.abs())
        if n == "elu":   # reference common.py:29: -alpha*relu(1-exp(x)) + relu(x), alpha = 1
            return -torch.relu(1.0 - torch.exp(x)) + torch.relu(x)
        raise NotImplementedError(n)

    def extra_repr(self):
        return self.name


_FUSED = {"leaky": "leaky", "relu": "relu", "sigmoid": "sigmoid", "tanh": "tanh", "identity": None}


def get_activation(act):
    """Same contract as reference common.py:32-57: None -> identity, str -> block, block -> itself."""
    if act is None:
        return Activation("identity")
    if isinstance(act, str):
        if act in _FUSED:
            return Activation(act, fused=_FUSED[act])
        if act in ("elu", "softrelu", "softsign"):
            return Activation(act)
        raise NotImplementedError(act)
    return act


def xavier_in_uniform_(w):
    """MXNet Xavier(rnd_type='uniform', factor_type='in', magnitude=3) used at reference STAR-GCN.py:548:
    U(-s, s) with s = sqrt(3 / fan_in)."""
    fan_in = w.shape[1] if w.dim() > 1 else w.shape[0]
    s = math.sqrt(3.0 / max(fan_in, 1))
    with torch.no_grad():
        return w.uniform_(-s, s)


class Dense(nn.Module):
    """Gluon `nn.Dense(units, flatten=False)` with deferred input size, running on the fp32 MFMA GEMM with the
    bias and (fusable) activation folded into the epilogue."""

    def __init__(self, units, activation=None, use_bias=True, in_units=0):
        super().__init__()
        self._units, self._use_bias = units, use_bias
        self.act = get_activation(activation)
        self.weight = nn.UninitializedParameter() if in_units == 0 else nn.Parameter(torch.empty(units, in_units))
        self.bias = nn.Parameter(torch.zeros(units)) if use_bias else None
        if in_units:
            xavier_in_uniform_(self.weight)

    def _materialize(self, in_units, device):
        if isinstance(self.weight, nn.UninitializedParameter):
            self.weight.materialize((self._units, in_units), device=device, dtype=torch.float32)
            xavier_in_uniform_(self.weight)
            if self.bias is not None and self.bias.device != device:
                self.bias.data = self.bias.data.to(device)

    def forward(self, x):
        self._materialize(x.shape[-1], x.device)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if isinstance(self.act, Activation) and self.act.fusable:
            y = SF.linear(x2, self.weight, self.bias, act=self.act.fused, slope=self.act.slope)
        else:
            y = self.act(SF.linear(x2, self.weight, self.bias))
        return y.reshape(*lead, self._units)


class LayerDictionary(nn.Module):
    """Mapping (any hashable key, e.g. ('user', 'movie')) -> block, registered so parameters are tracked
    (counterpart of reference layers.py:8-39)."""

    def __init__(self):
        super().__init__()
        self._key2idx = dict()
        self._layers = nn.ModuleList()

    def __len__(self):
        return len(self._layers)

    def __setitem__(self, key, layer):
        if key in self._key2idx:
            self._layers[self._key2idx[key]] = layer
        else:
            self._key2idx[key] = len(self._layers)
            self._layers.append(layer)

    def __getitem__(self, key):
        return self._layers[self._key2idx[key]]

    def __contains__(self, key):
        return key in self._key2idx

    def keys(self):
        return self._key2idx.keys()

    def items(self):
        return ((k, self._layers[i]) for k, i in self._key2idx.items())

    def values(self):
        return (self._layers[i] for i in self._key2idx.values())
