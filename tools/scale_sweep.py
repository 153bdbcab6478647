"""f16x3 / bf16x6 / f32 error against the float64 oracle over input scales (documentation of the modes' input range)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import misonet_amd as mz
from misonet_amd import weights as W
from oracle import miso_oracle
from conftest import mag_parity
sd1 = W.make_state_dict(W.miso1_spec(), 0)
m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m.load_state_dict(sd1); m.eval()
r = np.random.default_rng(8)
x0 = (r.standard_normal((1, 6, 64, 129)) + 1j * r.standard_normal((1, 6, 64, 129))).astype(np.complex64)
for sc in (1e-6, 1e-4, 1e-2, 1.0, 1e2, 1e3, 1e4, 1e6):
    x = (x0 * sc).astype(np.complex64)
    with miso_oracle.precision(torch.float64):
        truth = miso_oracle.miso1_forward(torch.from_numpy(x).to(torch.complex128), sd1).numpy()
    row = []
    for mode in ("f32", "bf16x6", "f16x3"):
        try:
            y = m.set_precision(mode)(torch.from_numpy(x).cuda()).cpu().numpy()
            row.append(f"{mode} {mag_parity(y, truth)[0]:.2e}")
        except FloatingPointError:
            row.append(f"{mode} NaN-error")
    print(f"scale {sc:8.0e}: " + "  ".join(row), flush=True)
