#!/usr/bin/env python3
"""G13 / G14: two more runs of the REAL reference's ``Tester_Enhance.inference`` (tester.py:846-975), same recipe as
oracle/gen_golden_full.py (build container only; inputs regenerated from the seed, weights from misonet_amd.weights):

  * G13 -- the 8 kHz geometry of the committed config (config/NN_BSS.yml: fs 8000, 4 s chunks => T = 501 frames):
    synthetic utterance 5 cut to 32000 samples, ref_ch = 0.  Kept: every 8th frame of the MVDR / MISO3 spectrograms, the
    magnitude sums of all frames, the int16 waves decimated by 8.
  * G14 -- ``ref_ch = 2`` (the alignment anchor, the clean references' microphone and the MISO3 input all move with it,
    tester.py:874, 889-890, 898-900, 937, 1030-1038): synthetic utterance 9, T = 64, 16 kHz; stored in full.

Run from the repo root:   python -m oracle.gen_golden_more
"""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle.gen_golden import OUT, build_models, import_reference


def run(ref_tester, sf_stub, m1, m3, obs, s0, s1, fs, ref_ch):
    n = (obs.shape[1] - 1) * 64
    tst = ref_tester.Tester_Enhance("SMS_WSJ", "MISO3", None, None, m1, m3, 6, "cpu", 2, n / fs, "/tmp/golden_out", ref_ch,
                                    False, fs=fs, window="hann", length=256, overlap=192)
    rec = {}
    o_bf, o_m3 = tst.Apply_Beamforming, tst.MISO3_inference

    def bfw(source, mixb, epsi=1e-6):
        r = o_bf(source, mixb, epsi)
        rec.setdefault("bf", []).append(r.numpy().copy())
        return r

    def m3w(mixt, bft, m1t):
        r = o_m3(mixt, bft, m1t)
        rec.setdefault("miso1_ref", []).append(m1t.numpy().copy())
        rec.setdefault("out", []).append(r.numpy().copy())
        return r
    tst.Apply_Beamforming, tst.MISO3_inference = bfw, m3w
    sf_stub.written.clear()
    loader = [({"0": torch.from_numpy(obs)[None]}, {"0": torch.from_numpy(s0)[None]}, {"0": torch.from_numpy(s1)[None]},
               [0], ["utt"])]
    tst.inference(loader, "/tmp/golden_out")
    wavs = np.stack([np.asarray(w[1]).reshape(-1) for w in sf_stub.written]).astype(np.int16)
    assert wavs.shape == (2, n), wavs.shape
    return (np.stack([b[0] for b in rec["bf"]]), np.stack([m[0, 0] for m in rec["miso1_ref"]]),
            np.stack([o[0, 0] for o in rec["out"]]), wavs)


def main():
    from misonet_amd.weights import synthetic_utterance
    from oracle.pipeline_oracle import stft_chunk
    ref_model, ref_tester, sf_stub = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m1, m3 = build_models(ref_model)

    # ---- G13: 8 kHz, T = 501 ----
    T, st = 501, 8
    w = synthetic_utterance(5, (T - 1) * 64)
    obs, s0, s1 = (stft_chunk(x, 8000) for x in w)
    bf, mr, out, wav = run(ref_tester, sf_stub, m1, m3, obs, s0, s1, 8000, 0)
    d = dict(utt=np.int64(5), frames=np.int64(T), frame_step=np.int64(st), fs=np.int64(8000))
    for name, z in (("bf", bf), ("miso1_ref", mr), ("out", out)):
        d[name + "_frames"] = z[:, ::st].astype(np.complex64)
        d[name + "_magsum"] = np.abs(z).astype(np.float64).sum(axis=-1)
    d["wav_dec8"] = wav[:, ::8]
    d["wav_abssum_1000"] = np.abs(wav.astype(np.int64)).reshape(2, -1, 1000).sum(axis=-1)
    p = os.path.join(OUT, "g13_pipeline_8k_T501.npz")
    np.savez_compressed(p, **d)
    print("G13 ->", p, os.path.getsize(p), "bytes")

    # ---- G14: ref_ch = 2, T = 64, 16 kHz ----
    T = 64
    w = synthetic_utterance(9, (T - 1) * 64)
    obs, s0, s1 = (stft_chunk(x, 16000) for x in w)
    bf, mr, out, wav = run(ref_tester, sf_stub, m1, m3, obs, s0, s1, 16000, 2)
    p = os.path.join(OUT, "g14_pipeline_refch2_T64.npz")
    np.savez_compressed(p, utt=np.int64(9), frames=np.int64(T), ref_ch=np.int64(2), bf=bf.astype(np.complex64),
                        miso1_ref=mr.astype(np.complex64), out=out.astype(np.complex64), wav=wav)
    print("G14 ->", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
