import sys, numpy as np, torch
sys.path.insert(0, '.')
import misonet_amd as mz
from misonet_amd import weights as W
T = int(sys.argv[1]) if len(sys.argv) > 1 else 130
N = 9
sd1 = W.make_state_dict(W.miso1_spec(), 1)
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
m1.load_state_dict(sd1); m1.eval().set_precision("bf16x3")
r = np.random.default_rng(5)
x = torch.from_numpy((r.standard_normal((1, 6, T, 129)) + 1j * r.standard_normal((1, 6, T, 129))).astype(np.complex64)).cuda()
yb = m1(torch.cat([x] * N, dim=0))
ws = m1._ws[(N, T)]
raw = ws[256:256 + (N * 12 * 2 + N * 120 * 2) * 8].cpu().numpy().view(np.float64)
e0 = raw[N * 12 * 2:].reshape(N, 120, 2)
for c in (24, 25, 47, 48, 72, 96):
    print("ch", c, "sum s0/s8:", repr(e0[0, c, 0]), repr(e0[8, c, 0]), " rel diff", (e0[0, c, 0] - e0[8, c, 0]) / abs(e0[8, c, 0]),
          "| sumsq rel diff", (e0[0, c, 1] - e0[8, c, 1]) / abs(e0[8, c, 1]))
d = np.abs(e0[0] - e0[8]) / (np.abs(e0[8]) + 1e-300)
print("max rel diff per slice:", [float(d[24 * k:24 * k + 24].max()) for k in range(5)])
print("s1 vs s8:", float((np.abs(e0[1] - e0[8]) / (np.abs(e0[8]) + 1e-300)).max()))
