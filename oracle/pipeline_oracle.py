"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's per-utterance inference orchestration
(Tester_Enhance.inference, reference tester.py:846-975) on top of miso_oracle /
mvdr_oracle, with the reference's B = 1 semantics (its batch > 1 path is buggy,
tester.py:1065 -- see SURVEY.md section 0), plus the STFT / iSTFT contract
(dataloader/data.py:505-522,540-544; tester.py:979-990,949-952).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import scipy.signal
import torch

from . import miso_oracle, mvdr_oracle

NPERSEG, NOVERLAP, WINDOW = 256, 192, "hann"          # config/NN_BSS.yml:72-88
SCALE = float(np.sqrt(1.0 / scipy.signal.get_window(WINDOW, NPERSEG).sum() ** 2))   # data.py:497-498 == 1/128


def stft_chunk(wav, fs=16000):
    """wav [L, M] float -> complex64 [M, T, F]: un-normalised one-sided STFT (data.py:505-522,542)."""
    chans = []
    for c in range(wav.shape[1]):
        _, _, z = scipy.signal.stft(wav[:, c], fs=fs, window=WINDOW, nperseg=NPERSEG, noverlap=NOVERLAP)
        chans.append(z)
    z = np.stack(chans, axis=0) / SCALE                     # [M,F,T]
    return np.ascontiguousarray(np.transpose(z, (0, 2, 1))).astype(np.complex64)


def istft_int16(spec_tf, fs=16000):
    """spec [T,F] complex -> int16 [ (T-1)*hop ]  (tester.py:950-952, 979-990): x*scale -> istft -> *32767 -> int16."""
    x = np.asarray(spec_tf).T * SCALE
    _, t_sig = scipy.signal.istft(x, fs=fs, window=WINDOW, nperseg=NPERSEG, noverlap=NOVERLAP)
    return (t_sig * np.iinfo(np.int16).max).astype(np.int16)


def miso1_inference(mix, sd1, ref_ch=0):
    """tester.py:1014-1068 for one utterance.  mix complex [M,T,F] -> complex64 [S,M,T,F] (speaker-aligned
    across the M circular shifts), plus the selected permutation per shift [M,S]."""
    mix_t = torch.as_tensor(mix)[None]
    M = mix_t.shape[1]
    order = list(np.roll(np.arange(M), -ref_ch))
    ref = miso_oracle.miso1_forward(torch.roll(mix_t, -ref_ch, dims=1), sd1)[0].numpy()     # [S,T,F]
    S = ref.shape[0]
    out = np.empty((S, M) + ref.shape[1:], dtype=np.complex64)
    sel_all = np.tile(np.arange(S), (M, 1))
    out[:, ref_ch] = ref
    for k in order[1:]:
        est = miso_oracle.miso1_forward(torch.roll(mix_t, -int(k), dims=1), sd1)[0].numpy()
        sel, _ = mvdr_oracle.pit_select(ref[None], est[None])
        sel_all[k] = sel[0]
        for i in range(S):
            out[i, k] = est[sel[0, i]]
    return out, sel_all


def enhance_utterance(mix, clean, sd1, sd3, ref_ch=0, epsi=1e-6, timings: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """tester.py:865-939 for one utterance / one 4 s split.

    mix   complex [M,T,F]; clean complex [S,T,F] (clean sources at ref_ch, tester.py:889-891)
    Returns miso1 [S,M,T,F] (after clean alignment), bf [S,T,F], out [S,T,F] (MISO3), sel_clean [S].
    ``timings`` (optional dict) receives the wall seconds of the stages: miso1_x6, mvdr_x2, miso3_x2 (bench.py's
    cpu_baseline leg).
    """
    import time
    t0 = time.perf_counter()
    est, sel_shift = miso1_inference(mix, sd1, ref_ch)
    sel, _ = mvdr_oracle.pit_select(np.asarray(clean)[None], est[None, :, ref_ch])       # tester.py:902-915
    est = est[sel[0]]
    t1 = time.perf_counter()
    S = est.shape[0]
    mix_bf = np.transpose(np.asarray(mix), (2, 0, 1))[None]                                # [1,F,M,T] tester.py:921
    bf, out = [], []
    mix_t = torch.as_tensor(np.asarray(mix))[None]
    t_bf = t_m3 = 0.0
    for s in range(S):
        ta = time.perf_counter()
        src = np.transpose(est[s], (2, 0, 1))[None]                                        # tester.py:923
        b = mvdr_oracle.apply_beamforming(src, mix_bf, epsi)                               # [1,T,F]
        bf.append(b[0])
        tb = time.perf_counter()
        o = miso_oracle.miso3_forward(mix_t, torch.from_numpy(b)[:, None],
                                      torch.from_numpy(est[s, ref_ch])[None, None], sd3)   # tester.py:937-939,1242
        out.append(o[0, 0].numpy())
        t_bf += tb - ta
        t_m3 += time.perf_counter() - tb
    if timings is not None:
        timings.update(miso1_x6=t1 - t0, mvdr_x2=t_bf, miso3_x2=t_m3)
    return dict(miso1=est, bf=np.stack(bf), out=np.stack(out), sel_clean=sel[0], sel_shift=sel_shift)


def beamform_utterance(obs_splits, clean_splits, gap, sd1, ref_ch=0, epsi=1e-6, fs=16000):
    """Utterance-wise MVDR of Tester_Beamforming (tester.py:340-449, utterance_flag): per split MISO1_Inference +
    clean alignment, iSTFT every (speaker, mic) estimate and the observation, stitch the splits (last one trimmed by
    ``gap``), re-STFT the whole recording, one MVDR per speaker over all frames, iSTFT -> int16.

    obs_splits: list of complex [M,T,F]; clean_splits: list of complex [S,T,F] (clean sources at ref_ch).
    Returns int16 [S, n] (n = recording length rounded up to a whole hop, as scipy's padded STFT/iSTFT pair gives)."""
    n_split = len(obs_splits)
    est_t, obs_t = None, None
    for k in range(n_split):
        est, _ = miso1_inference(obs_splits[k], sd1, ref_ch)                               # [S,M,T,F]
        sel, _ = mvdr_oracle.pit_select(np.asarray(clean_splits[k])[None], est[None, :, ref_ch])
        est = est[sel[0]]
        S, M = est.shape[:2]
        def to_time(x_mtf):                                                                # [M,T,F] -> [M, chunk]
            _, t_sig = scipy.signal.istft(np.transpose(x_mtf, (0, 2, 1)) * SCALE, fs=fs, window=WINDOW,
                                          nperseg=NPERSEG, noverlap=NOVERLAP)
            return t_sig
        e_t = [to_time(est[s]) for s in range(S)]
        o_t = to_time(np.asarray(obs_splits[k]))
        if k == n_split - 1 and gap:
            e_t = [e[:, : e.shape[1] - gap] for e in e_t]
            o_t = o_t[:, : o_t.shape[1] - gap]
        est_t = e_t if est_t is None else [np.append(a, b, axis=1) for a, b in zip(est_t, e_t)]
        obs_t = o_t if obs_t is None else np.append(obs_t, o_t, axis=1)

    def stft_utt(x_ml):                                                                    # [M, L] float64 -> [1,F,M,T]
        z = np.stack([scipy.signal.stft(x_ml[c], fs=fs, window=WINDOW, nperseg=NPERSEG, noverlap=NOVERLAP)[2]
                      for c in range(x_ml.shape[0])]) / SCALE                              # [M,F,T] complex128
        return np.transpose(z, (1, 0, 2))[None]
    mix = stft_utt(obs_t)
    out = []
    for s in range(len(est_t)):
        bf = mvdr_oracle.mvdr_parts(stft_utt(est_t[s]), mix, epsi, dtype=np.complex128)["out"][0]   # [T,F]
        _, t_sig = scipy.signal.istft(bf.T * SCALE, fs=fs, window=WINDOW, nperseg=NPERSEG, noverlap=NOVERLAP)
        out.append((t_sig * np.iinfo(np.int16).max).astype(np.int16))
    return np.stack(out)
