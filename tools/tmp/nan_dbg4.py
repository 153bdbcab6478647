import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import misonet_amd as mz
from misonet_amd import weights as W, _lib, stft as S
from misonet_amd.pipeline import Enhancer
sd1 = W.make_state_dict(W.miso1_spec(), 0); sd3 = W.make_state_dict(W.miso3_spec(), 1)
mode = "bf16x6"
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m1.load_state_dict(sd1); m1.eval().set_precision(mode)
m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m3.load_state_dict(sd3); m3.eval().set_precision(mode)
enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
n = 20 * 64
obs, s0, s1 = W.synthetic_utterance(1, n)
good = torch.from_numpy(obs)[None].clone(); cg = torch.from_numpy(np.stack([s0[:, 0], s1[:, 0]], axis=1))[None].clone()
bad = good.clone(); bad[0, 100, 2] = float("nan")
# reference results, synchronous
ref_good = enh.enhance_wav_int16(good.cuda(), cg.cuda(), check_nan=False).cpu().numpy()
ref_bad = enh.enhance_wav_int16(bad.cuda(), cg.cuda(), check_nan=False).cpu().numpy()
print("ref_good absmax", np.abs(ref_good.astype(np.int32)).max(), "ref_bad absmax", np.abs(ref_bad.astype(np.int32)).max(), "ref_bad unique", np.unique(ref_bad)[:5])
for trial in range(5):
    it = enh.stream_wav(iter([(good, cg), (bad, cg), (good, cg)]), depth=2, check_nan=False)
    outs = list(it)
    print(trial, [("==good" if np.array_equal(o, ref_good) else ("==bad" if np.array_equal(o, ref_bad) else "other absmax %d" % np.abs(o.astype(np.int32)).max())) for o in outs])
