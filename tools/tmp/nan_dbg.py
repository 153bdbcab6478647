import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import misonet_amd as mz
from misonet_amd import weights as W, _lib
sd1 = W.make_state_dict(W.miso1_spec(), 0); sd3 = W.make_state_dict(W.miso3_spec(), 1)
for mode in ("f32", "bf16x6"):
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m1.load_state_dict(sd1); m1.eval().set_precision(mode)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m3.load_state_dict(sd3); m3.eval().set_precision(mode)
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    n = 20 * 64
    obs, s0, s1 = W.synthetic_utterance(1, n)
    good = torch.from_numpy(obs)[None].clone(); cg = torch.from_numpy(np.stack([s0[:, 0], s1[:, 0]], axis=1))[None].clone()
    bad = good.clone(); bad[0, 100, 2] = float("nan")
    for trial in range(6):
        res = []
        for w in (good, bad, good):
            out = enh.enhance_wav(w.cuda(), cg.cuda(), check_nan=False)
            torch.cuda.synchronize()
            ws = enh.workspace(1, 21)
            flag = int(ws[:4].view(torch.int32).cpu()[0])
            res.append((flag, bool(torch.isnan(torch.view_as_real(out)).any()), bool(torch.isfinite(torch.view_as_real(out)).all())))
        print(mode, trial, res)
