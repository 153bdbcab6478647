#!/usr/bin/env python3
"""G12: the REAL reference (/root/reference) at the FULL bench geometry -- synthetic utterance 0 of BASELINE configs[1..3]
(6 mics, 16 kHz, 4 s: T = 1001 frames, F = 129) through

  * ``MISO_1.forward`` (model.py:76-111): one forward of the un-shifted mixture, and
  * ``Tester_Enhance.inference`` (tester.py:846-975): 6 x MISO_1 over the circular shifts -> alignment -> 2 x MVDR ->
    2 x MISO_3 -> iSTFT -> int16,

so that full-size parity of the HIP path (and of the oracle) is pinned by the reference itself and not only through the oracle
(which the other goldens pin at T <= 96 and on a T = 501 slice).

Run from the repo root (build container only; the reference never travels):   python -m oracle.gen_golden_full
Inputs are NOT stored (they come from misonet_amd.weights.synthetic_utterance(0, 64000), as in bench.py); weights come from
misonet_amd.weights.make_state_dict.  To stay small the fixture keeps, per output: every 16th frame in complex64, the
per-frame magnitude sums of ALL frames (float64) and the int16 waves decimated by 16 plus their per-1000-sample |.| sums.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from oracle.gen_golden import OUT, build_models, import_reference

T_FULL = 1001
FRAME_STEP = 16


def main():
    from misonet_amd.weights import synthetic_utterance
    from oracle.pipeline_oracle import stft_chunk
    ref_model, ref_tester, sf_stub = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m1, m3 = build_models(ref_model)
    n = (T_FULL - 1) * 64
    obs_w, s0_w, s1_w = synthetic_utterance(0, n)
    obs, s0, s1 = stft_chunk(obs_w, 16000), stft_chunk(s0_w, 16000), stft_chunk(s1_w, 16000)
    assert obs.shape == (6, T_FULL, 129), obs.shape

    t0 = time.time()
    with torch.no_grad():
        y1 = m1(torch.from_numpy(obs)[None]).numpy()[0]                         # [2, T, F]
    print(f"MISO_1.forward at T = {T_FULL}: {time.time() - t0:.1f} s", y1.shape)

    tst = ref_tester.Tester_Enhance("SMS_WSJ", "MISO3", None, None, m1, m3, 6, "cpu", 2, n / 16000,
                                    "/tmp/golden_out", 0, False, fs=16000, window="hann", length=256, overlap=192)
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
    loader = [({"0": torch.from_numpy(obs)[None]}, {"0": torch.from_numpy(s0)[None]},
               {"0": torch.from_numpy(s1)[None]}, [0], ["utt"])]
    t0 = time.time()
    tst.inference(loader, "/tmp/golden_out")
    print(f"Tester_Enhance.inference at T = {T_FULL}: {time.time() - t0:.1f} s")
    wavs = [np.asarray(w[1]).reshape(-1) for w in sf_stub.written]
    assert len(wavs) == 2 and all(w.shape == (n,) for w in wavs), [w.shape for w in wavs]

    def pack(name, z, d):
        z = np.asarray(z)
        d[name + "_frames"] = z[..., ::FRAME_STEP, :].astype(np.complex64)
        d[name + "_magsum"] = np.abs(z).astype(np.float64).sum(axis=-1)

    d = dict(utt=np.int64(0), frames=np.int64(T_FULL), frame_step=np.int64(FRAME_STEP))
    pack("miso1_fwd", y1, d)                                                      # [2, T, F]
    pack("bf", np.stack([b[0] for b in rec["bf"]]), d)                            # [2, T, F]
    pack("miso1_ref", np.stack([m[0, 0] for m in rec["miso1_ref"]]), d)           # [2, T, F] aligned MISO1 estimate at ref_ch
    pack("out", np.stack([o[0, 0] for o in rec["out"]]), d)                       # [2, T, F]
    w = np.stack(wavs).astype(np.int16)
    d["wav_dec16"] = w[:, ::16]
    d["wav_abssum_1000"] = np.abs(w.astype(np.int64)).reshape(2, -1, 1000).sum(axis=-1)
    path = os.path.join(OUT, "g12_fullsize_T1001.npz")
    np.savez_compressed(path, **d)
    print("G12 ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
