"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in stock torch functional ops, of the reference network forward:
  * MISO_1.forward            reference model.py:76-111
  * MISO_3.forward            reference model.py:350-395
  * init_Conv2d_/Conv2d_      model.py:401-416   (conv -> ELU -> InstanceNorm2d)
  * last_Deconv2d_/DeConv2d_  model.py:418-433
  * DenseBlock                model.py:437-482
  * TemporalConvNet/Block     model.py:486-550
  * DepthwiseSeparableConv    model.py:553-567
  * GlobalLayerNorm           model.py:609-632, InstanceNorm1d via chose_norm model.py:570-581

It works on a flat ``state_dict`` (the reference key names) instead of nn.Modules, so
it needs neither the reference sources nor the product package.  Parity pinning: the
goldens under tests/golden/ were produced by oracle/gen_golden.py importing the real
reference (with the shims listed in SURVEY.md 8(c)) in the build container;
tests/test_oracle_golden.py checks this restatement against them.

``taps`` (optional dict) receives intermediate tensors for per-stage parity checks.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

EPS_GLN = 1e-8          # model.py:6
# Arithmetic of the restatement.  float32 = the reference (model.py:77-80 `.float()`, float32 parameters).  Tests that
# need a ground truth BETTER than any float32 implementation (conditioning studies) switch to float64 through
# ``with precision(torch.float64): ...``; nothing else about the computation changes.
DTYPE = torch.float32


class precision:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global DTYPE
        self.old, DTYPE = DTYPE, self.dtype

    def __exit__(self, *exc):
        global DTYPE
        DTYPE = self.old


def _t(sd, key):
    v = sd[key]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.to(DTYPE)


def _conv_elu_in(x, w, b, stride, padding, transposed=False, act=True):
    # model.py:411-414 / 428-431 / 442-446: conv -> ELU(alpha=1) -> InstanceNorm2d(affine=False, eps=1e-5)
    if transposed:
        y = F.conv_transpose2d(x, w, b, stride=stride, padding=padding)
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=padding)
    if act:
        y = F.instance_norm(F.elu(y), eps=1e-5)
    return y


def _dense_block(x, sd, prefix):
    # model.py:467-482
    feats = [x]
    y = None
    for i in range(5):
        inp = torch.cat(feats, dim=1)
        y = _conv_elu_in(inp, _t(sd, f"{prefix}.conv{i + 1}.0.weight"), _t(sd, f"{prefix}.conv{i + 1}.0.bias"),
                         (1, 1), (1, 1))
        feats.append(y)
    return y


def _gln(y, gamma, beta):
    # model.py:629-632
    mean = y.mean(dim=1, keepdim=True).mean(dim=2, keepdim=True)
    var = (torch.pow(y - mean, 2)).mean(dim=1, keepdim=True).mean(dim=2, keepdim=True)
    return gamma * (y - mean) / torch.pow(var + EPS_GLN, 0.5) + beta


def _ds_conv(x, sd, p, dilation):
    # DepthwiseSeparableConv, model.py:553-567
    c = x.shape[1]
    y = F.conv1d(x, _t(sd, f"{p}.0.weight"), None, stride=1, padding=dilation, dilation=dilation, groups=c)
    y = F.prelu(y, _t(sd, f"{p}.1.weight"))
    y = _gln(y, _t(sd, f"{p}.2.gamma"), _t(sd, f"{p}.2.beta"))
    return F.conv1d(y, _t(sd, f"{p}.3.weight"), None)


def _cln(y, gamma, beta):
    # ChannelwiseLayerNorm, model.py:583-606: statistics over the channels of every frame
    mean = torch.mean(y, dim=1, keepdim=True)
    var = torch.var(y, dim=1, keepdim=True, unbiased=False)
    return gamma * (y - mean) / torch.pow(var + EPS_GLN, 0.5) + beta


def _outer_norm(y, sd, q, kind):
    """chose_norm(norm_type, C) of a TemporalBlock (model.py:530,535,570-581): "IN" = nn.InstanceNorm1d(affine=False),
    "gLN", "cLN", anything else = nn.BatchNorm1d -- in eval mode (run.py:79,106: the test path calls .eval())."""
    if kind == "IN":
        return F.instance_norm(y, eps=1e-5)
    if kind == "gLN":
        return _gln(y, _t(sd, f"{q}.gamma"), _t(sd, f"{q}.beta"))
    if kind == "cLN":
        return _cln(y, _t(sd, f"{q}.gamma"), _t(sd, f"{q}.beta"))
    return F.batch_norm(y, _t(sd, f"{q}.running_mean"), _t(sd, f"{q}.running_var"), _t(sd, f"{q}.weight"), _t(sd, f"{q}.bias"),
                        training=False, eps=1e-5)


def tcn_forward(x, sd, taps: Optional[Dict] = None, norm_type: str = "IN"):
    # TemporalConvNet(2,7,128,128,128,norm_type), model.py:31,486-550
    for r in range(2):
        for blk in range(7):
            d = 2 ** blk
            p = f"TCN.temporal_conv_net.{r}.{blk}.net"
            res = x
            y = F.elu(_outer_norm(x, sd, f"{p}.0", norm_type))
            y = _ds_conv(y, sd, f"{p}.2.net", d)
            y = F.elu(_outer_norm(y, sd, f"{p}.3", norm_type))
            y = _ds_conv(y, sd, f"{p}.5.net", d)
            x = y + res
            if taps is not None and r == 0 and blk == 0:
                taps["tcn_block0"] = x
    return x


def trunk_forward(x, sd, taps: Optional[Dict] = None, norm_type: str = "IN"):
    """x: float [B, Cin, T, 129] (real||imag channels) -> float [B, Cout, T, 129]."""
    xs = []
    for b in range(7):
        if b == 0:
            # init_Conv2d_: no activation / norm, model.py:401-406
            x = F.conv2d(x, _t(sd, "encoders.0.0.conv2d.weight"), _t(sd, "encoders.0.0.conv2d.bias"),
                         stride=(1, 1), padding=(1, 0))
            if taps is not None:
                taps["enc0_conv"] = x
            x = _dense_block(x, sd, "encoders.0.1")
        else:
            stride = (1, 1) if b == 6 else (1, 2)      # model.py:49-52
            x = _conv_elu_in(x, _t(sd, f"encoders.{b}.0.net.0.weight"), _t(sd, f"encoders.{b}.0.net.0.bias"),
                             stride, (1, 0))
            if b < 5:
                x = _dense_block(x, sd, f"encoders.{b}.1")
        xs.append(x)
        if taps is not None:
            taps[f"enc{b}"] = x
    # model.py:89: torch.squeeze -> [B,128,T]; TemporalBlock re-adds the batch dim when B == 1 (model.py:546-547)
    assert x.shape[-1] == 1, "encoder must reduce the frequency axis to one bin (n_freq = 129)"
    x = x[..., 0]
    x = tcn_forward(x, sd, taps, norm_type)
    if taps is not None:
        taps["tcn_out"] = x
    de = x.unsqueeze(-1)
    for b in range(7):
        de = torch.cat((de, xs[6 - b]), dim=1)          # model.py:99
        if b >= 2:
            de = _dense_block(de, sd, f"decoders.{b}.0")
            if b == 6:
                de = F.conv_transpose2d(de, _t(sd, "decoders.6.1.deconv2d.weight"), _t(sd, "decoders.6.1.deconv2d.bias"),
                                        stride=(1, 1), padding=(1, 0))   # last_Deconv2d_: no act / norm
            else:
                de = _conv_elu_in(de, _t(sd, f"decoders.{b}.1.net.0.weight"), _t(sd, f"decoders.{b}.1.net.0.bias"),
                                  (1, 2), (1, 0), transposed=True)
        else:
            stride = (1, 1) if b == 0 else (1, 2)        # model.py:68-71
            de = _conv_elu_in(de, _t(sd, f"decoders.{b}.0.net.0.weight"), _t(sd, f"decoders.{b}.0.net.0.bias"),
                              stride, (1, 0), transposed=True)
        if taps is not None:
            taps[f"dec{b}"] = de
    return de


def _split_complex(out):
    c = out.shape[1]
    return torch.complex(out[:, : c // 2].contiguous(), out[:, c // 2:].contiguous())   # model.py:103-111


@torch.no_grad()
def miso1_forward(mixture, sd, taps: Optional[Dict] = None, norm_type: str = "IN"):
    """mixture complex [B,M,T,129] -> complex64 [B,num_spks,T,129]  (model.py:76-111)."""
    x = torch.cat((mixture.real.to(DTYPE), mixture.imag.to(DTYPE)), dim=1)
    return _split_complex(trunk_forward(x, sd, taps, norm_type))


@torch.no_grad()
def miso3_forward(mixture, a, b, sd, taps: Optional[Dict] = None, norm_type: str = "IN"):
    """model.py:350-395.  Channel order: real(mix, a, b) then imag(mix, a, b); the reference's
    parameter names for a/b are swapped at the call site (tester.py:1242) -- order is what counts."""
    real = torch.cat((mixture.real.to(DTYPE), a.real.to(DTYPE), b.real.to(DTYPE)), dim=1)
    imag = torch.cat((mixture.imag.to(DTYPE), a.imag.to(DTYPE), b.imag.to(DTYPE)), dim=1)
    x = torch.cat((real, imag), dim=1)
    return _split_complex(trunk_forward(x, sd, taps, norm_type))
