"""Weight specification and deterministic synthetic weights for MISO_1 / MISO_3.

The tensor names and shapes are the ``state_dict`` keys of the reference modules
(reference model.py:8-73 MISO_1, model.py:282-347 MISO_3, building blocks
model.py:401-567, GlobalLayerNorm model.py:609-619).  They are derived here from
the constructor arguments, not by importing the reference, so that the GPU box
(which never sees /root/reference) can regenerate identical weights.

``make_state_dict(seed, ...)`` is the NumPy-seeded, key-name-ordered generator that
SURVEY.md 8(c) asks for: every tensor is drawn from its own
``default_rng([seed, crc32(key)])`` stream, so the values do not depend on module
construction order or on torch's RNG.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np

DEFAULT_EN_CH = (24, 32, 32, 32, 32, 64, 128)   # reference config/NN_BSS.yml:120-123
DEFAULT_DE_CH = (128, 64, 32, 32, 32, 32, 24)
N_FREQ = 129                                     # nperseg 256 (config/NN_BSS.yml:72-78); see SURVEY.md section 0
TCN_REPEATS, TCN_BLOCKS, TCN_CH = 2, 7, 128      # reference model.py:31


def norm_kind(norm_type: str) -> int:
    """``norm_type`` of the reference constructors -> the two OUTER norms of every TemporalBlock (model.py:530,535 through
    chose_norm, model.py:570-581): 0 "IN" (InstanceNorm1d, no parameters), 1 "gLN", 2 "cLN", 3 anything else (BatchNorm1d)."""
    return {"IN": 0, "gLN": 1, "cLN": 2}.get(norm_type, 3)


def tensor_spec(in_ch: int, out_ch: int,
                en_ch: Sequence[int] = DEFAULT_EN_CH,
                de_ch: Sequence[int] = DEFAULT_DE_CH, norm_type: str = "IN") -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {state_dict key: shape} for a MISO trunk.

    in_ch  = 2*num_ch (MISO_1, model.py:16) or 2*(num_ch+2) (MISO_3, model.py:290)
    out_ch = 2*num_spks (model.py:17 / 291)
    norm_type: the constructors' argument; adds the parameters of the TemporalBlocks' outer norms (gLN / cLN: gamma, beta
    [1, C, 1]; BatchNorm1d: weight, bias, running_mean, running_var [C] and the int64 scalar num_batches_tracked)
    """
    en = [in_ch] + list(en_ch)
    de = list(de_ch) + [out_ch]
    nb = len(en_ch)
    assert nb == 7 and len(de_ch) == 7, "the reference layer rules (model.py:40-73) are written for 7 blocks"
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def dense(prefix: str, init_ch: int, g1: int, g2: int) -> None:
        # DenseBlock, model.py:437-466
        for i in range(5):
            cout = g1 if i < 4 else g2
            spec[f"{prefix}.conv{i + 1}.0.weight"] = (cout, init_ch + i * g1, 3, 3)
            spec[f"{prefix}.conv{i + 1}.0.bias"] = (cout,)

    # encoders, model.py:40-54
    for b in range(nb):
        cin, cout = en[b], en[b + 1]
        if b == 0:
            spec[f"encoders.{b}.0.conv2d.weight"] = (cout, cin, 3, 3)
            spec[f"encoders.{b}.0.conv2d.bias"] = (cout,)
            dense(f"encoders.{b}.1", cout, cout, cout)
        elif b < 5:
            spec[f"encoders.{b}.0.net.0.weight"] = (cout, cin, 3, 3)
            spec[f"encoders.{b}.0.net.0.bias"] = (cout,)
            dense(f"encoders.{b}.1", cout, cout, cout)
        else:
            spec[f"encoders.{b}.0.net.0.weight"] = (cout, cin, 3, 3)
            spec[f"encoders.{b}.0.net.0.bias"] = (cout,)
    # decoders, model.py:56-73 (ConvTranspose2d weight is [Cin, Cout, 3, 3])
    for b in range(nb):
        cin, cout = 2 * de[b], de[b + 1]
        if b >= 2:
            dense(f"decoders.{b}.0", cin, cin // 2, cin)
            name = "deconv2d" if b == 6 else "net.0"
            spec[f"decoders.{b}.1.{name}.weight"] = (cin, cout, 3, 3)
            spec[f"decoders.{b}.1.{name}.bias"] = (cout,)
        else:
            spec[f"decoders.{b}.0.net.0.weight"] = (cin, cout, 3, 3)
            spec[f"decoders.{b}.0.net.0.bias"] = (cout,)
    # TCN, model.py:486-567
    c = TCN_CH
    nk = norm_kind(norm_type)
    for r in range(TCN_REPEATS):
        for x in range(TCN_BLOCKS):
            for half in (2, 5):
                q = f"TCN.temporal_conv_net.{r}.{x}.net.{half - 2}"      # norm_1 = net.0, norm_2 = net.3 (model.py:530-539)
                if nk in (1, 2):
                    spec[f"{q}.gamma"] = (1, c, 1)
                    spec[f"{q}.beta"] = (1, c, 1)
                elif nk == 3:
                    spec[f"{q}.weight"] = (c,)
                    spec[f"{q}.bias"] = (c,)
                    spec[f"{q}.running_mean"] = (c,)
                    spec[f"{q}.running_var"] = (c,)
                    spec[f"{q}.num_batches_tracked"] = ()       # int64 scalar: carried through state_dict, unused in eval
                p = f"TCN.temporal_conv_net.{r}.{x}.net.{half}.net"
                spec[f"{p}.0.weight"] = (c, 1, 3)       # depth-wise dilated conv
                spec[f"{p}.1.weight"] = (1,)            # PReLU slope
                spec[f"{p}.2.gamma"] = (1, c, 1)        # gLN
                spec[f"{p}.2.beta"] = (1, c, 1)
                spec[f"{p}.3.weight"] = (c, c, 1)       # point-wise conv
    return spec


def miso1_spec(num_spks: int = 2, num_ch: int = 6, en_ch=DEFAULT_EN_CH, de_ch=DEFAULT_DE_CH, norm_type: str = "IN"):
    return tensor_spec(2 * num_ch, 2 * num_spks, en_ch, de_ch, norm_type)


def miso3_spec(num_spks: int = 1, num_ch: int = 6, en_ch=DEFAULT_EN_CH, de_ch=DEFAULT_DE_CH, norm_type: str = "IN"):
    return tensor_spec(2 * (num_ch + 2), 2 * num_spks, en_ch, de_ch, norm_type)


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(key.encode("utf-8"))])


def make_state_dict(spec: "OrderedDict[str, Tuple[int, ...]]", seed: int) -> "OrderedDict[str, np.ndarray]":
    """Deterministic float32 weights for every key of ``spec``.

    Gains are chosen so that activations stay O(1) through the ~64 conv+norm layers
    (every conv is followed by an instance norm, so only the bias/weight ratio and
    the non-degeneracy of the channels matter).
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in spec.items():
        r = _rng(seed, key)
        if key.endswith(".num_batches_tracked"):
            sd[key] = np.asarray(7, dtype=np.int64)
            continue
        if key.endswith(".running_mean"):
            v = 0.2 * r.standard_normal(shape)
        elif key.endswith(".running_var"):
            v = 0.5 + r.uniform(0.0, 1.0, shape)
        elif len(shape) == 1 and key.endswith(".weight") and shape[0] > 1:      # BatchNorm1d weight
            v = 1.0 + 0.1 * r.standard_normal(shape)
        elif key.endswith(".bias"):
            v = 0.1 * r.standard_normal(shape)
        elif key.endswith(".gamma"):
            v = 1.0 + 0.1 * r.standard_normal(shape)
        elif key.endswith(".beta"):
            v = 0.1 * r.standard_normal(shape)
        elif key.endswith(".1.weight") and len(shape) == 1:       # PReLU slope
            v = 0.25 + 0.05 * r.standard_normal(shape)
        elif len(shape) == 4:                                     # Conv2d [Co,Ci,3,3] / ConvTranspose2d [Ci,Co,3,3]
            is_deconv = key.startswith("decoders") and (".net.0.weight" in key or ".deconv2d.weight" in key)
            fan_in = (shape[0] if is_deconv else shape[1]) * 9
            v = r.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif len(shape) == 3 and shape[1] == 1:                   # depth-wise [C,1,3]
            v = r.standard_normal(shape) * np.sqrt(1.0 / 3.0)
        elif len(shape) == 3:                                     # point-wise [C,C,1]
            v = r.standard_normal(shape) * np.sqrt(1.0 / shape[1])
        else:
            raise ValueError(f"unhandled tensor {key} {shape}")
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def synthetic_utterance(u: int, n_samples: int = 64000, n_mic: int = 6):
    """SURVEY.md 8(d) config-2 input: two sources 0.05*N(0,1) [n_samples, n_mic] float32; observation = s0+s1."""
    r = np.random.default_rng(1000 + int(u))
    s0 = (0.05 * r.standard_normal((n_samples, n_mic))).astype(np.float32)
    s1 = (0.05 * r.standard_normal((n_samples, n_mic))).astype(np.float32)
    return s0 + s1, s0, s1
