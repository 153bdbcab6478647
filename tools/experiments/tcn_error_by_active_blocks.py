"""Round-4 experiment (GPU): error at the TCN output against float64 as a function of the number of active TCN blocks
(the point-wise conv weights of the later blocks zeroed).  usage: python tools/experiments/tcn_error_by_active_blocks.py T sample mode"""
import sys, numpy as np, torch
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
for k in range(0, 15):
    sd = dict(sd0)
    for (s, b) in blocks[k:]:
        key = f"TCN.temporal_conv_net.{s}.{b}.net.5.net.3.weight"
        sd[key] = np.zeros_like(sd0[key])
    t64, t32 = {}, {}
    with miso_oracle.precision(torch.float64):
        miso_oracle.miso1_forward(torch.from_numpy(x).to(torch.complex128), sd, t64)
    miso_oracle.miso1_forward(torch.from_numpy(x), sd, t32)
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd); m1.eval().set_precision(mode); m1.keep_activations(True)
    m1(torch.from_numpy(x).cuda())
    v = m1.tap("tcn_out", 1, T).cpu().numpy()
    a64 = t64["tcn_out"].numpy(); a32 = t32["tcn_out"].numpy()
    a64 = a64.reshape(v.shape); a32 = a32.reshape(v.shape)
    e = np.abs(v.astype(np.float64) - a64)
    ci = np.unravel_index(np.argmax(e), e.shape)
    print("blocks active %2d: hip-vs-64 %.3e  orc32-vs-64 %.3e  worst ch %d t %d |err| %.3e val %.3e" % (k, rel(v, a64), rel(a32, a64), ci[1], ci[2], e[ci], a64[ci]))
    del m1
