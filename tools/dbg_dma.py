import sys, numpy as np, torch
sys.path.insert(0, '.')
import misonet_amd as mz
from misonet_amd import weights as W
B, T = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
sd1 = W.make_state_dict(W.miso1_spec(), 1)
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
m1.load_state_dict(sd1); m1.eval().set_precision(prec)
r = np.random.default_rng(5)
x = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
m1.set_precision("f32")
y0 = m1(torch.from_numpy(x).cuda()).cpu().numpy()
print(B, T, prec, "rel", np.linalg.norm(y - y0) / np.linalg.norm(y0))
