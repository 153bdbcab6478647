import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import misonet_amd as mz
from misonet_amd import weights as W
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
sd1 = W.make_state_dict(W.miso1_spec(), 1)
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
m1.load_state_dict(sd1); m1.eval().set_precision(prec)
r = np.random.default_rng(5)
x = torch.from_numpy((r.standard_normal((1, 6, T, 129)) + 1j * r.standard_normal((1, 6, T, 129))).astype(np.complex64)).cuda()
def rl(a, b): return float(np.linalg.norm(a - b) / np.linalg.norm(b))
y1 = m1(x).cpu().numpy(); y1b = m1(x).cpu().numpy()
xb = torch.cat([x] * 9, dim=0)
yb = m1(xb).cpu().numpy(); yb2 = m1(xb).cpu().numpy()
print(prec, T, "single repeat", rl(y1b, y1))
print("batch vs single:", [f"{rl(yb[i], y1[0]):.2e}" for i in range(9)])
print("batch repeat:", [f"{rl(yb2[i], yb[i]):.2e}" for i in range(9)])
