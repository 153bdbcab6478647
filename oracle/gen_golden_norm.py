#!/usr/bin/env python3
"""Golden G11: MISO_1 / MISO_3 of the REAL reference built with norm_type in {"gLN", "cLN", "BN"} -- the argument selects the
two outer norms of every TemporalBlock (reference model.py:530,535 through chose_norm, model.py:570-581; the committed config
uses "IN", config/NN_BSS.yml:123).  Run from the repo root:   python -m oracle.gen_golden_norm

Weights come from misonet_amd.weights.make_state_dict (seed 3 / 4) and are loaded into the reference modules with
load_state_dict (so the key names, order and shapes of weights.tensor_spec(..., norm_type) are checked against the
reference's own state_dict on the way); BatchNorm1d runs in eval mode as on the reference's test path (run.py:79,106)."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle.gen_golden import OUT, import_reference, synth_spec


def main():
    ref_model, _, _ = import_reference()
    from misonet_amd import weights as W
    torch.manual_seed(0)
    for nt in ("gLN", "cLN", "BN"):
        sd1 = W.make_state_dict(W.miso1_spec(norm_type=nt), seed=3)
        m1 = ref_model.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), nt).eval()
        assert list(m1.state_dict().keys()) == list(sd1.keys()), f"{nt}: key / ordering mismatch vs weights.miso1_spec"
        for k, v in m1.state_dict().items():
            assert tuple(v.shape) == tuple(sd1[k].shape), (nt, k, tuple(v.shape), sd1[k].shape)
        m1.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd1.items()})
        x = synth_spec(700 + len(nt), (2, 6, 40, 129))
        with torch.no_grad():
            y = m1(torch.from_numpy(x)).numpy()
        out = {"x": x, "y": y.astype(np.complex64)}
        if nt == "cLN":                                       # one MISO_3 case (16 input channels) as well
            sd3 = W.make_state_dict(W.miso3_spec(norm_type=nt), seed=4)
            m3 = ref_model.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), nt).eval()
            assert list(m3.state_dict().keys()) == list(sd3.keys())
            m3.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd3.items()})
            a = synth_spec(801, (2, 1, 40, 129))
            b = synth_spec(802, (2, 1, 40, 129))
            with torch.no_grad():
                y3 = m3(torch.from_numpy(x), torch.from_numpy(a), torch.from_numpy(b)).numpy()
            out.update(a=a, b=b, y3=y3.astype(np.complex64))
        path = os.path.join(OUT, f"g11_norm_{nt}_T40.npz")
        np.savez_compressed(path, **out)
        print(nt, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
