#!/usr/bin/env python3
"""f32w (Winograd F(2x2,3x3) dense-block convs, conv_wino.hip) beside f32 (direct): stage taps of one forward against the
oracle, goldens G1/G3, ragged batched shapes, bit-exact batch invariance.  GPU box only (experiment driver, not a test)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, rel_l2, mag_parity   # noqa: E402
import misonet_amd as mz                           # noqa: E402
from misonet_amd import weights as W              # noqa: E402
from oracle import miso_oracle                     # noqa: E402

torch.set_num_threads(16)
sd1 = W.make_state_dict(W.miso1_spec(), 0)
sd3 = W.make_state_dict(W.miso3_spec(), 1)


def net1(mode):
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    return m.eval().set_precision(mode)


g = golden("g1_miso1_T32.npz")
x = torch.from_numpy(g["x"])
taps = {}
y_ref = miso_oracle.miso1_forward(x, sd1, taps).numpy()
names = ["enc0_conv"] + [f"enc{b}" for b in range(7)] + ["tcn_out"] + [f"dec{b}" for b in range(7)]
MODES = tuple(os.environ.get("WINO_CHECK_MODES", "f32,f32w").split(","))
for mode in MODES:
    m = net1(mode)
    m.keep_activations(True)
    y = m(x.cuda()).cpu().numpy()
    for nm in names:
        ref = taps[nm].numpy()
        if ref.ndim == 3:
            ref = ref[..., None]
        got = m.tap(nm, 1, 32).cpu().numpy()
        print(f"[{mode}] tap {nm:10s} rel_l2={rel_l2(got, ref):.3e}", flush=True)
    print(f"[{mode}] T=32 vs oracle {mag_parity(y, y_ref)}  vs golden {mag_parity(y, g['y'])}", flush=True)
    m.keep_activations(False)
    g96 = golden("g1_miso1_T96.npz")
    y96 = m(torch.from_numpy(g96["x"]).cuda()).cpu().numpy()
    print(f"[{mode}] T=96 vs golden {mag_parity(y96, g96['y'])}", flush=True)
    for B, T in [(3, 40), (2, 130), (1, 5), (2, 257), (2, 4)]:
        r = np.random.default_rng(B * 1000 + T)
        xx = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
        xx[1:] *= 3.0
        yy = m(torch.from_numpy(xx).cuda()).cpu().numpy()
        yr = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(xx[b:b + 1]), sd1).numpy() for b in range(B)])
        y1 = np.concatenate([m(torch.from_numpy(xx[b:b + 1]).cuda()).cpu().numpy() for b in range(B)])
        print(f"[{mode}] B={B} T={T} vs oracle {mag_parity(yy, yr)} batch-invariant bits: {np.array_equal(yy, y1)}", flush=True)
    # float64 truth at T = 96: which mode is closer
    with miso_oracle.precision(torch.float64):
        y64 = miso_oracle.miso1_forward(torch.from_numpy(g96["x"]), sd1).numpy()
    print(f"[{mode}] T=96 vs float64 oracle rel_l2(mag) = {mag_parity(y96, y64)[0]:.3e}", flush=True)
