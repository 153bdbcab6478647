import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import misonet_amd as mz
from misonet_amd import weights as W, _lib, stft as S
sd1 = W.make_state_dict(W.miso1_spec(), 0); sd3 = W.make_state_dict(W.miso3_spec(), 1)
use_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if use_torch:
    S._istft_hip_orig = S._istft_hip
    def istft_int16_torch(spec):
        lead = spec.shape[:-2]; T, F = spec.shape[-2:]
        z = spec.reshape(-1, T, F).transpose(1, 2).to(torch.complex64)
        x = torch.istft(z, n_fft=256, hop_length=64, win_length=256, window=S._window(z.device), center=True, normalized=False, onesided=True, length=(T - 1) * 64, return_complex=False)
        return (x.reshape(*lead, -1) * 32767).to(torch.int16)
    S.istft_int16 = istft_int16_torch
for mode in ("f32", "bf16x6"):
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m1.load_state_dict(sd1); m1.eval().set_precision(mode)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m3.load_state_dict(sd3); m3.eval().set_precision(mode)
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    n = 20 * 64
    obs, s0, s1 = W.synthetic_utterance(1, n)
    good = torch.from_numpy(obs)[None].clone(); cg = torch.from_numpy(np.stack([s0[:, 0], s1[:, 0]], axis=1))[None].clone()
    bad = good.clone(); bad[0, 100, 2] = float("nan")
    raised = 0
    for trial in range(20):
        it = enh.stream_wav(iter([(good, cg), (bad, cg), (good, cg)]), depth=2)
        first = next(it)
        try:
            nxt = next(it)
            print(mode, trial, "NO RAISE; second batch pcm absmax", np.abs(nxt.astype(np.int32)).max(), "first", np.abs(first.astype(np.int32)).max())
        except FloatingPointError:
            raised += 1
    print(mode, "raised", raised, "of 20", "(torch istft)" if use_torch else "(hip istft)")
