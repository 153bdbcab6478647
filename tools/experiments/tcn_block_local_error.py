"""Round-4 experiment (GPU): error of every TCN block of the HIP path ON ITS OWN INPUT (weights of the later blocks zeroed, so
that the tcn_out tap shows the state after k blocks) against the float64 oracle block.  usage: python tools/experiments/tcn_block_local_error.py T sample mode"""
import sys, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import misonet_amd as mz
from misonet_amd import weights as W
from oracle import miso_oracle
sd0 = W.make_state_dict(W.miso1_spec(), 0)
T = int(sys.argv[1]); bsel = int(sys.argv[2]); mode = sys.argv[3]
r = np.random.default_rng(77 + T)
xall = (r.standard_normal((2, 6, T, 129)) + 1j * r.standard_normal((2, 6, T, 129))).astype(np.complex64)
x = xall[bsel:bsel + 1]
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
blocks = [(s, b) for s in range(2) for b in range(7)]
xs = []
for k in range(0, 15):
    sd = dict(sd0)
    for (s, b) in blocks[k:]:
        key = f"TCN.temporal_conv_net.{s}.{b}.net.5.net.3.weight"
        sd[key] = np.zeros_like(sd0[key])
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd); m1.eval().set_precision(mode); m1.keep_activations(True)
    m1(torch.from_numpy(x).cuda())
    xs.append(m1.tap("tcn_out", 1, T).cpu().numpy()[..., 0].astype(np.float64))     # [1,128,T]
    del m1
def block64(xin, s, b):
    with miso_oracle.precision(torch.float64):
        d = 2 ** b
        p = f"TCN.temporal_conv_net.{s}.{b}.net"
        xt = torch.from_numpy(xin)
        stages = {}
        y = F.instance_norm(xt, eps=1e-5); stages["in1"] = y
        y = F.elu(y)
        y = miso_oracle._ds_conv(y, sd0, f"{p}.2.net", d); stages["ds1"] = y
        y2 = F.instance_norm(y, eps=1e-5); stages["in2"] = y2
        y = F.elu(y2)
        y = miso_oracle._ds_conv(y, sd0, f"{p}.5.net", d)
        return (y + xt).numpy(), stages
for k, (s, b) in enumerate(blocks):
    want, st = block64(xs[k], s, b)
    e = np.abs(xs[k + 1] - want)
    ci = np.unravel_index(np.argmax(e), e.shape)
    # conditioning of the two instance norms: smallest |x0 - x1| / 2 over channels (T = 2)
    xin = xs[k][0]; d1 = np.abs(xin[:, 0] - xin[:, -1]) / 2
    ds1 = st["ds1"].numpy()[0]; d2 = np.abs(ds1[:, 0] - ds1[:, -1]) / 2
    print("block %2d (dil %2d): local err %.3e  worst ch %3d t %d |err| %.3e val %.3e | min delta: IN1 %.2e (ch %d)  IN2 %.2e (ch %d)" %
          (k, 2 ** b, rel(xs[k + 1], want), ci[1], ci[2], e[ci], want[ci], d1.min(), d1.argmin(), d2.min(), d2.argmin()))
