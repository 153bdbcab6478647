import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import misonet_amd as mz
from misonet_amd import weights as W, _lib, stft as S
from misonet_amd.pipeline import Enhancer
sd1 = W.make_state_dict(W.miso1_spec(), 0); sd3 = W.make_state_dict(W.miso3_spec(), 1)
variant = sys.argv[1]
orig = Enhancer.enhance_wav_int16
def patched(self, wav, clean_wav=None, check_nan=True):
    if variant == "sync_before": torch.cuda.synchronize()
    spec = self.enhance_wav(wav, clean_wav, check_nan=check_nan)
    if variant == "sync_mid": torch.cuda.synchronize()
    if variant == "nan_probe":
        patched.log.append(bool(torch.isnan(torch.view_as_real(spec)).any()))
    r = S.istft_int16(spec)
    if variant == "sync_after": torch.cuda.synchronize()
    return r
patched.log = []
Enhancer.enhance_wav_int16 = patched
mode = "bf16x6"
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m1.load_state_dict(sd1); m1.eval().set_precision(mode)
m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m3.load_state_dict(sd3); m3.eval().set_precision(mode)
enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
n = 20 * 64
obs, s0, s1 = W.synthetic_utterance(1, n)
good = torch.from_numpy(obs)[None].clone(); cg = torch.from_numpy(np.stack([s0[:, 0], s1[:, 0]], axis=1))[None].clone()
bad = good.clone(); bad[0, 100, 2] = float("nan")
raised = 0
for trial in range(20):
    patched.log = []
    it = enh.stream_wav(iter([(good, cg), (bad, cg), (good, cg)]), depth=2)
    first = next(it)
    try:
        nxt = next(it)
    except FloatingPointError:
        raised += 1
    if variant == "nan_probe" and trial < 4: print("spec-has-nan per batch:", patched.log)
print(variant, "raised", raised, "of 20")
