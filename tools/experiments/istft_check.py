"""Round-4 experiment (GPU): the HIP iSTFT against SciPy (float64) and torch, and its time against torch.istft."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from misonet_amd import stft as S
import scipy.signal
torch.manual_seed(0)
for (N, T) in ((3, 64), (2, 5), (1, 2), (32, 1001), (2, 501), (1, 200)):
    z = (torch.randn(N, T, 129) + 1j * torch.randn(N, T, 129)).to(torch.complex64) * 30
    zd = z.cuda()
    y_hip = S.istft(zd).cpu().numpy()
    # torch reference on CPU (float32) and scipy float64
    y_t = S.istft(z).numpy()
    ys = np.stack([scipy.signal.istft(z[i].numpy().T.astype(np.complex128) / 128.0, fs=16000, window="hann", nperseg=256, noverlap=192)[1][: (T - 1) * 64] for i in range(N)])
    e_hip = np.abs(y_hip - ys).max() / np.abs(ys).max(); e_t = np.abs(y_t - ys).max() / np.abs(ys).max()
    i_hip = S.istft_int16(zd / 300).cpu().numpy().astype(np.int32)
    i_ref = (ys / 300 * 32767).astype(np.int16).astype(np.int32)
    print(f"N={N} T={T}: HIP vs scipy64 max rel {e_hip:.2e}; torch-cpu vs scipy64 {e_t:.2e}; int16 max |diff| {np.abs(i_hip - i_ref).max()} LSB, differing {np.mean(i_hip != i_ref):.3%}")
z = (torch.randn(32, 1001, 129) + 1j * torch.randn(32, 1001, 129)).to(torch.complex64).cuda()
def torch_path(x):
    lead = x.shape[:-2]; T, F = x.shape[-2:]
    zz = x.reshape(-1, T, F).transpose(1, 2)
    y = torch.istft(zz, n_fft=256, hop_length=64, win_length=256, window=torch.hann_window(256, periodic=True, device=x.device), center=True, normalized=False, onesided=True, length=(T - 1) * 64, return_complex=False)
    return (y * 32767).to(torch.int16)
for name, fn in (("hip", S.istft_int16), ("torch", torch_path)):
    for _ in range(3): fn(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn(z)
    torch.cuda.synchronize(); print(name, "istft_int16 [32,1001,129]: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
