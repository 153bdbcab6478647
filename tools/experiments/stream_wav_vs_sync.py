"""Round-4 experiment (GPU): batches through Enhancer.stream_wav against the synchronous result of each batch (the race of a
fresh slot buffer with queued work showed here as a second batch that depended on the FIRST batch's data)."""
import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, 'tests')
import misonet_amd as mz
from misonet_amd import weights as W, _lib, stft as S
sd1 = W.make_state_dict(W.miso1_spec(), 0); sd3 = W.make_state_dict(W.miso3_spec(), 1)
mode = "bf16x6"
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m1.load_state_dict(sd1); m1.eval().set_precision(mode)
m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m3.load_state_dict(sd3); m3.eval().set_precision(mode)
enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
n = 20 * 64
obs, s0, s1 = W.synthetic_utterance(1, n)
good = torch.from_numpy(obs)[None].clone(); cg = torch.from_numpy(np.stack([s0[:, 0], s1[:, 0]], axis=1))[None].clone()
obs2, _, _ = W.synthetic_utterance(2, n)
other = torch.from_numpy(obs2)[None].clone()
bad = good.clone(); bad[0, 100, 2] = float("nan")
def ref(w): return enh.enhance_wav_int16(w.cuda(), cg.cuda(), check_nan=False).cpu().numpy()
r_good, r_other, r_bad = ref(good), ref(other), ref(bad)
def name(o):
    for k, r in (("good", r_good), ("other", r_other), ("bad", r_bad)):
        if np.array_equal(o, r): return k
    return "??? absmax %d" % np.abs(o.astype(np.int32)).max()
for seq_name, seq in (("good,other,good", [good, other, good]), ("good,bad,good", [good, bad, good]), ("other,bad,other", [other, bad, other])):
    for trial in range(3):
        outs = list(enh.stream_wav(iter([(w, cg) for w in seq]), depth=2, check_nan=False))
        print(seq_name, trial, [name(o) for o in outs])
# is the result for `bad` deterministic when run synchronously many times?
for k in range(4):
    print("sync bad again:", name(ref(bad)), "| sync good:", name(ref(good)))
