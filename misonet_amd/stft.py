"""STFT / iSTFT contract of the reference's test-time data path, on torch (device or CPU).

  * analysis  : dataloader/data.py:505-522,540-544 -- scipy.signal.stft(hann, 256, 192) / scale with
                scale = 1/sum(hann) = 1/128, i.e. the UN-normalised one-sided STFT, zero 'boundary' padding of
                nperseg/2 on both sides; chunking into 4 s pieces with a zero-padded tail and its ``gap``
                (data.py:555-595)
  * synthesis : tester.py:949-952, 979-990 -- istft(spec * scale) * 32767 -> int16 (truncation toward zero)

torch.stft/istft with center=True, zero padding and a periodic hann window compute the same transforms (checked
against SciPy in tests/test_stft.py).  On the device both directions are hand-written HIP (csrc/stft.hip: ``stft_hip``, and
since round 4 the iSTFT + int16 of :func:`istft` / :func:`istft_int16`); torch's versions remain for CPU tensors (tests, host
tools) -- the north_star allows PyTorch for "tensor containers and iSTFT".
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

NPERSEG, HOP = 256, 64
N_FREQ = NPERSEG // 2 + 1
MAX_INT16 = 32767


def _window(device, dtype=torch.float32):
    return torch.hann_window(NPERSEG, periodic=True, device=device, dtype=dtype)


def stft_hip(wav: torch.Tensor) -> torch.Tensor:
    """The hand-written HIP front-end (csrc/stft.hip, C ABI ``misonet_stft``): wav float32 [B, L, M] on the device
    (time-major, microphones interleaved -- the ``librosa.load(...).T`` array of dataloader/data.py:605-616) ->
    complex64 [B, M, T, 129], T = L // 64 + 1.  Same transform as :func:`stft` (DFT on the fp32 matrix cores)."""
    import ctypes as C
    from . import _lib
    if wav.dim() != 3 or not wav.is_cuda:
        raise ValueError("expected a device tensor [B, n_samples, n_mic]")
    w = wav.to(torch.float32).contiguous()
    B, L, M = w.shape
    lib = _lib.lib()
    T = lib.misonet_stft_frames(L)
    out = torch.empty((B, M, T, 129), dtype=torch.complex64, device=w.device)
    ws = torch.empty(lib.misonet_stft_workspace_bytes(B, M, L), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.misonet_stft(w.data_ptr(), B, L, M, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _lib.stream_ptr(w.device)))
    return out


def stft(wav: torch.Tensor) -> torch.Tensor:
    """wav float [..., L] -> complex64 [..., T, F] with T = L // 64 + 1, F = 129 (un-normalised, as fed to MISO_1).
    Host tool and test reference (torch.stft).  On a DEVICE tensor this is rocFFT with run-time compiled kernels: fine in one
    process, but one of eight processes started together on one GPU got a wrong spectrogram in a quarter of the runs (round 5,
    LAB section 0) -- device code paths use :func:`stft_hip` (or the wav entry points of the pipeline)."""
    lead = wav.shape[:-1]
    x = wav.reshape(-1, wav.shape[-1]).float()
    z = torch.stft(x, n_fft=NPERSEG, hop_length=HOP, win_length=NPERSEG, window=_window(x.device), center=True,
                   pad_mode="constant", normalized=False, onesided=True, return_complex=True)      # [N, F, T]
    return z.transpose(1, 2).reshape(*lead, z.shape[2], z.shape[1]).to(torch.complex64)


def _istft_hip(spec: torch.Tensor, want_i16: bool) -> torch.Tensor:
    """The hand-written HIP iSTFT (csrc/stft.hip ``istft_k``, C ABI ``misonet_istft``): device complex [..., T, 129] ->
    float32 or int16 [..., 64 (T - 1)] (windowed inverse DFT on the fp32 matrix cores, overlap-add, envelope division and --
    for int16 -- ``x 32767`` and the truncating cast in one launch)."""
    from . import _lib
    lead = spec.shape[:-2]
    T, F = spec.shape[-2:]
    z = spec.reshape(-1, T, F).to(torch.complex64).contiguous()
    out = torch.empty((z.shape[0], (T - 1) * HOP), dtype=torch.int16 if want_i16 else torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().misonet_istft(z.data_ptr(), z.shape[0], T, out.data_ptr() if want_i16 else None,
                                            None if want_i16 else out.data_ptr(), _lib.stream_ptr(z.device)))
    return out.reshape(*lead, (T - 1) * HOP)


def istft(spec: torch.Tensor, length: int = None) -> torch.Tensor:
    """complex [..., T, F] -> float32 [..., (T-1)*64]: inverse of :func:`stft` (== scipy istft(spec * scale)).  Device tensors
    of the network geometry run the HIP kernel; CPU tensors (tests, host tools) and other lengths go through torch."""
    lead = spec.shape[:-2]
    T, F = spec.shape[-2:]
    if spec.is_cuda and F == N_FREQ and T >= 2 and (length is None or length == (T - 1) * HOP) and spec.numel() > 0:
        return _istft_hip(spec, False)
    z = spec.reshape(-1, T, F).transpose(1, 2).to(torch.complex64)
    n = (T - 1) * HOP if length is None else length
    x = torch.istft(z, n_fft=NPERSEG, hop_length=HOP, win_length=NPERSEG, window=_window(z.device), center=True,
                    normalized=False, onesided=True, length=n, return_complex=False)
    return x.reshape(*lead, n)


def istft_int16(spec: torch.Tensor) -> torch.Tensor:
    """tester.py:950-952: time signal * 32767 -> int16 (C-style truncation, like ndarray.astype(np.int16))."""
    if spec.is_cuda and spec.shape[-1] == N_FREQ and spec.shape[-2] >= 2 and spec.numel() > 0:
        return _istft_hip(spec, True)
    return (istft(spec) * MAX_INT16).to(torch.int16)


def split_chunks(wav: np.ndarray, chunk: int) -> Tuple[List[np.ndarray], int]:
    """data.py:555-595: wav [L, M] -> list of [chunk, M] pieces (last one zero-padded) and the pad length ``gap``.

    The reference leaves L == chunk unhandled (neither branch, data.py:558-565); here it is one chunk, gap 0."""
    L = wav.shape[0]
    out, start = [], 0
    while True:
        piece = wav[start:start + chunk]
        gap = chunk - piece.shape[0]
        if gap:
            piece = np.pad(piece, ((0, gap), (0, 0)))
        out.append(piece)
        start += chunk
        if start >= L:
            return out, gap


def stitch_int16(chunks: List[np.ndarray], gap: int) -> np.ndarray:
    """tester.py:960-969: drop the zero-padded tail of the last chunk and concatenate."""
    chunks = list(chunks)
    if gap:
        chunks[-1] = chunks[-1][: len(chunks[-1]) - gap]
    return np.concatenate(chunks)


def write_wav_pcm24(path: str, samples_int16: np.ndarray, fs: int) -> None:
    """tester.py:972-974: ``sf.write(path, int16_array.T, fs, 'PCM_24')``.  libsndfile widens int16 to 24-bit PCM by a
    left shift of 8 bits, so each sample is written as the 3 little-endian bytes of ``int16 << 8``, behind the plain
    44-byte RIFF/WAVE header it writes for PCM (fmt chunk of 16 bytes, format tag 1); a data chunk of odd length is
    followed by one zero pad byte that the RIFF size counts and the data size does not -- byte for byte the layout of
    the reference's own outputs (sample/MISO3/*.wav: 64059 mono samples -> data 192177, RIFF 192214, file 192222).
    samples_int16: [n_samples] (mono) or [n_samples, n_channels]."""
    import struct
    x = np.asarray(samples_int16)
    if x.dtype != np.int16:
        raise TypeError("expected int16 samples (tester.py:952 casts before writing)")
    if x.ndim == 1:
        x = x[:, None]
    n, ch = x.shape
    v = (x.astype(np.int32) << 8)
    b = np.empty(x.shape + (3,), dtype=np.uint8)
    b[..., 0] = v & 0xFF
    b[..., 1] = (v >> 8) & 0xFF
    b[..., 2] = (v >> 16) & 0xFF
    data = b.tobytes()
    pad = len(data) & 1
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(data) + pad) + b"WAVE" +
           b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, int(fs), int(fs) * ch * 3, ch * 3, 24) +
           b"data" + struct.pack("<I", len(data)))
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(data)
        if pad:
            f.write(b"\x00")


def read_wav_pcm24(path: str):
    """Inverse of :func:`write_wav_pcm24` (tests): returns (int32 samples [n, ch] in 24-bit range, fs)."""
    import wave
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 3
        ch, fs, n = w.getnchannels(), w.getframerate(), w.getnframes()
        raw = np.frombuffer(w.readframes(n), dtype=np.uint8).reshape(n, ch, 3).astype(np.int32)
    v = raw[..., 0] | (raw[..., 1] << 8) | (raw[..., 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v)
    return v, fs
