import sys, numpy as np, torch
sys.path.insert(0, '.')
import misonet_amd as mz
from misonet_amd import weights as W
from oracle import miso_oracle
sd1 = W.make_state_dict(W.miso1_spec(), 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
r = np.random.default_rng(77 + T)
x = (r.standard_normal((2, 6, T, 129)) + 1j * r.standard_normal((2, 6, T, 129))).astype(np.complex64)[:1]
t32, t64 = {}, {}
y32 = miso_oracle.miso1_forward(torch.from_numpy(x), sd1, t32).numpy()
with miso_oracle.precision(torch.float64):
    y64 = miso_oracle.miso1_forward(torch.from_numpy(x).to(torch.complex128), sd1, t64).numpy()
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
for mode in ("f32", "bf16x6"):
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1); m1.eval().set_precision(mode); m1.keep_activations(True)
    y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
    print(mode, "out: hip-vs-64 %.3e  orc32-vs-64 %.3e" % (rel(np.abs(y), np.abs(y64)), rel(np.abs(y32), np.abs(y64))))
    for k in t64:
        try:
            v = m1.tap(k, 1, T).cpu().numpy()
        except Exception as e:
            continue
        a64 = t64[k].numpy(); a32 = t32[k].numpy()
        if a64.ndim == 3: a64 = a64[..., None]; a32 = a32[..., None]
        if a64.ndim == 2: a64 = a64[None, ..., None]; a32 = a32[None, ..., None]
        if v.shape != a64.shape:
            print(k, "shape", v.shape, a64.shape); continue
        print("  %-12s hip-vs-64 %.3e   orc32-vs-64 %.3e" % (k, rel(v, a64), rel(a32, a64)))
