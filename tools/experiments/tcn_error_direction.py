"""Round-4 experiment (GPU): what the float64 TCN does to the difference between the build's encoder output and the oracle's,
at T = 2 -- the amplification depends on the direction of the round-off (LAB, round 4)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import misonet_amd as mz
from misonet_amd import weights as W
from oracle import miso_oracle
sd = W.make_state_dict(W.miso1_spec(), 0)
T = 2
r = np.random.default_rng(77 + T)
xall = (r.standard_normal((2, 6, T, 129)) + 1j * r.standard_normal((2, 6, T, 129))).astype(np.complex64)
x = xall[1:2]
def rel(a, b): return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
t64, t32 = {}, {}
miso_oracle.miso1_forward(torch.from_numpy(x), sd, t32)
with miso_oracle.precision(torch.float64):
    miso_oracle.miso1_forward(torch.from_numpy(x).to(torch.complex128), sd, t64)
    e64 = t64["enc6"][..., 0].clone()
    out64 = miso_oracle.tcn_forward(e64, sd).numpy()
for mode in ("bf16x6", "f32"):
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd); m1.eval().set_precision(mode); m1.keep_activations(True)
    m1(torch.from_numpy(x).cuda())
    eh = m1.tap("enc6", 1, T).cpu().numpy()[..., 0].astype(np.float64)
    th = m1.tap("tcn_out", 1, T).cpu().numpy()[..., 0].astype(np.float64)
    with miso_oracle.precision(torch.float64):
        o = miso_oracle.tcn_forward(torch.from_numpy(eh), sd).numpy()
    d = eh - e64.numpy()
    print(mode, "enc6 err %.3e ; f64 TCN on HIP's enc6 vs truth: %.3e ; HIP tcn_out vs f64 TCN on HIP's enc6: %.3e ; HIP tcn_out vs truth %.3e" %
          (rel(eh, e64.numpy()), rel(o, out64), rel(th, o), rel(th, out64)))
    # structure of the enc6 error: common-mode (mean over the 2 frames) vs antisymmetric part per channel
    cm = d[0].mean(axis=1); an = (d[0][:, 0] - d[0][:, 1]) / 2
    print("   common-mode |.| max %.3e rms %.3e ; antisymmetric max %.3e rms %.3e ; f64 enc6 frame-sum max %.3e ; HIP enc6 frame-sum max %.3e" %
          (np.abs(cm).max(), np.sqrt((cm**2).mean()), np.abs(an).max(), np.sqrt((an**2).mean()),
           np.abs(e64.numpy()[0].sum(axis=1)).max(), np.abs(eh[0].sum(axis=1)).max()))
e32 = t32["enc6"][..., 0].numpy().astype(np.float64)
with miso_oracle.precision(torch.float64):
    o = miso_oracle.tcn_forward(torch.from_numpy(e32), sd).numpy()
d = e32 - e64.numpy()
cm = d[0].mean(axis=1); an = (d[0][:, 0] - d[0][:, 1]) / 2
print("oracle32 enc6 err %.3e ; f64 TCN on it vs truth %.3e ; common-mode max %.3e antisym max %.3e" % (rel(e32, e64.numpy()), rel(o, out64), np.abs(cm).max(), np.abs(an).max()))
