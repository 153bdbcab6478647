#!/usr/bin/env python3
"""bf16x6w debugging: stage taps of one forward against the oracle with NaN counts (experiment driver, GPU box only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, rel_l2   # noqa: E402
import misonet_amd as mz              # noqa: E402
from misonet_amd import weights as W  # noqa: E402
from oracle import miso_oracle        # noqa: E402

torch.set_num_threads(16)
sd1 = W.make_state_dict(W.miso1_spec(), 0)
mode = os.environ.get("MODE", "bf16x6w")
T = int(os.environ.get("T", "32"))
if T == 32:
    x = torch.from_numpy(golden("g1_miso1_T32.npz")["x"])
else:
    r = np.random.default_rng(T)
    x = torch.from_numpy((r.standard_normal((1, 6, T, 129)) + 1j * r.standard_normal((1, 6, T, 129))).astype(np.complex64))
taps = {}
y_ref = miso_oracle.miso1_forward(x, sd1, taps).numpy()
names = ["enc0_conv"] + [f"enc{b}" for b in range(7)] + ["tcn_out"] + [f"dec{b}" for b in range(7)]
m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
m.load_state_dict(sd1)
m = m.eval().set_precision(mode)
m.keep_activations(True)
y = m(x.cuda(), check_nan=False).cpu().numpy()
for nm in names:
    ref = taps[nm].numpy()
    if ref.ndim == 3:
        ref = ref[..., None]
    got = m.tap(nm, 1, T).cpu().numpy()
    bad = ~np.isfinite(got)
    msg = f"[{mode}] tap {nm:10s} shape {got.shape} nonfinite {int(bad.sum())}"
    if not bad.any():
        msg += f" rel_l2={rel_l2(got, ref):.3e}"
    else:
        idx = np.argwhere(bad)
        msg += f" first bad {idx[0].tolist()} last bad {idx[-1].tolist()} channels {sorted(set(idx[:, 1].tolist()))[:12]}"
    print(msg, flush=True)
    if nm in ("enc0", "enc1"):
        d = np.abs(got - ref)
        d[~np.isfinite(d)] = 1e9
        # error map per channel / row block / frame block
        print("  per-channel max err:", np.array2string(d.max(axis=(0, 2, 3)), precision=1, max_line_width=250, formatter={"float_kind": lambda v: "%.0e" % v}))
        print("  per-frame max err:", np.array2string(d.max(axis=(0, 1, 3)), precision=1, max_line_width=250, formatter={"float_kind": lambda v: "%.0e" % v}))
        print("  per-row(f) max err [first 24]:", np.array2string(d.max(axis=(0, 1, 2))[:24], precision=1, max_line_width=250, formatter={"float_kind": lambda v: "%.0e" % v}))

y2 = m(x.cuda(), check_nan=False).cpu().numpy()
print("second forward identical:", np.array_equal(y, y2, equal_nan=True), "max diff", float(np.nanmax(np.abs(y - y2))))
