#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) in the build container.

Run from the repo root:   python -m oracle.gen_golden
The reference never travels to the GPU box; only the small vectors written here do.

Shims needed to import / run reference tester.py under numpy 2 / without soundfile
(SURVEY.md 8(c)): (1) stub `soundfile` that records sf.write calls; (2) np.complex = complex;
(3) tester.solve with NumPy-1.x vector right-hand-side semantics; (4) inputs cast to complex64.
No reference source is copied: the reference modules are imported where they lie.

Weights are NOT stored: they come from misonet_amd.weights.make_state_dict (seed 0 -> MISO_1,
seed 1 -> MISO_3) and are loaded into the reference modules with load_state_dict.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference():
    sf_stub = types.ModuleType("soundfile")
    sf_stub.written = []
    sf_stub.write = lambda path, data, fs, subtype=None: sf_stub.written.append((path, np.array(data), fs, subtype))
    sys.modules["soundfile"] = sf_stub
    np.complex = complex                                   # shim (2)
    sys.path.insert(0, REF)
    import model as ref_model                              # noqa
    import tester as ref_tester                            # noqa
    ref_tester.solve = lambda a, b: np.linalg.solve(a, b[..., None])[..., 0]     # shim (3)
    return ref_model, ref_tester, sf_stub


def build_models(ref_model):
    from misonet_amd import weights as W
    sd1 = W.make_state_dict(W.miso1_spec(), seed=0)
    sd3 = W.make_state_dict(W.miso3_spec(), seed=1)
    m1 = ref_model.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").eval()
    m3 = ref_model.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").eval()
    assert list(m1.state_dict().keys()) == list(sd1.keys()), "MISO_1 key/ordering mismatch vs weights.miso1_spec"
    assert list(m3.state_dict().keys()) == list(sd3.keys()), "MISO_3 key/ordering mismatch vs weights.miso3_spec"
    for k, v in m1.state_dict().items():
        assert tuple(v.shape) == sd1[k].shape, k
    for k, v in m3.state_dict().items():
        assert tuple(v.shape) == sd3[k].shape, k
    m1.load_state_dict({k: torch.from_numpy(v) for k, v in sd1.items()})
    m3.load_state_dict({k: torch.from_numpy(v) for k, v in sd3.items()})
    return m1, m3


def synth_spec(seed, shape, scale=1.0):
    r = np.random.default_rng(seed)
    return (scale * (r.standard_normal(shape) + 1j * r.standard_normal(shape))).astype(np.complex64)


def synth_utt_stft(u, n_frames, fs):
    """Synthetic utterance of SURVEY 8(d) cut to (n_frames-1)*64 samples, STFT'd by the reference's own recipe."""
    from misonet_amd.weights import synthetic_utterance
    from oracle.pipeline_oracle import stft_chunk
    n = (n_frames - 1) * 64
    obs, s0, s1 = synthetic_utterance(u, n)
    return stft_chunk(obs, fs), stft_chunk(s0, fs), stft_chunk(s1, fs)


def hook_taps(m1):
    taps = {}

    def save(name):
        def fn(_m, _i, o):
            taps[name] = o.detach().clone()
        return fn
    hs = [m1.encoders[0][0].register_forward_hook(save("enc0_conv"))]
    for b in range(7):
        hs.append(m1.encoders[b].register_forward_hook(save(f"enc{b}")))
        hs.append(m1.decoders[b].register_forward_hook(save(f"dec{b}")))
    hs.append(m1.TCN.temporal_conv_net[0][0].register_forward_hook(save("tcn_block0")))
    hs.append(m1.TCN.register_forward_hook(save("tcn_out")))
    return taps, hs


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_model, ref_tester, sf_stub = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m1, m3 = build_models(ref_model)

    # ---- G1 / G2: MISO_1 forward + stage taps -----------------------------------------------------------------
    with torch.no_grad():
        for T in (32, 96):
            x = synth_spec(100 + T, (1, 6, T, 129))
            if T == 32:
                taps, hs = hook_taps(m1)
            y = m1(torch.from_numpy(x)).numpy()
            d = dict(x=x, y=y)
            if T == 32:
                for h in hs:
                    h.remove()
                for k, v in taps.items():
                    v = v.numpy()
                    if v.ndim == 3 and v.shape[0] != 1:
                        v = v[None]
                    if v.ndim == 2:                       # B == 1 squeeze quirk (model.py:89)
                        v = v[None]
                    if v.size > 60000:                    # keep fixtures small: every 4th channel
                        v = v[:, ::4]
                    d["tap_" + k] = v.astype(np.float32)
            np.savez_compressed(os.path.join(OUT, f"g1_miso1_T{T}.npz"), **d)
            print("G1", T, y.shape, np.abs(y).mean())

        # ---- G3: MISO_3 forward ------------------------------------------------------------------------------
        x = synth_spec(300, (1, 6, 32, 129)); a = synth_spec(301, (1, 1, 32, 129)); b = synth_spec(302, (1, 1, 32, 129))
        y = m3(torch.from_numpy(x), torch.from_numpy(a), torch.from_numpy(b)).numpy()
        np.savez_compressed(os.path.join(OUT, "g3_miso3_T32.npz"), x=x, a=a, b=b, y=y)
        print("G3", y.shape, np.abs(y).mean())

    # ---- tester object (fields per tester.py:799-825) -----------------------------------------------------------
    def make_tester(fs, chunk_time):
        return ref_tester.Tester_Enhance("SMS_WSJ", "MISO3", None, None, m1, m3, 6, "cpu", 2, chunk_time,
                                         "/tmp/golden_out", 0, False,
                                         fs=fs, window="hann", length=256, overlap=192)

    # ---- G4: MISO1_Inference (6 circular shifts + alignment), B = 1 ---------------------------------------------
    tst = make_tester(16000, 31 * 64 / 16000)
    x = synth_spec(400, (1, 6, 32, 129))
    est = tst.MISO1_Inference(torch.from_numpy(x), ref_ch=0)
    np.savez_compressed(os.path.join(OUT, "g4_miso1_inference_T32.npz"), x=x,
                        spk0=est[0].numpy(), spk1=est[1].numpy())
    print("G4", est[0].shape)

    # ---- G5: MVDR with all intermediates (monkey-patched taps) ---------------------------------------------------
    src = synth_spec(500, (1, 129, 6, 24)); mix = src + synth_spec(501, (1, 129, 6, 24), 0.7)
    rec = {}
    orig_pc, orig_bf = tst.PhaseCorrection, tst.get_mvdr_beamformer

    def pc(W):
        rec["steer0"] = np.array(W)
        out = orig_pc(W)
        rec["steer1"] = np.array(out)
        return out

    def bf(steer, rn, delta):
        rec["scm_n"] = np.array(rn)          # before += delta
        w = orig_bf(steer, rn, delta)
        rec["w"] = np.array(w)
        return w
    tst.PhaseCorrection, tst.get_mvdr_beamformer = pc, bf
    out = tst.Apply_Beamforming(src.copy(), mix.copy()).numpy()
    tst.PhaseCorrection, tst.get_mvdr_beamformer = orig_pc, orig_bf
    np.savez_compressed(os.path.join(OUT, "g5_mvdr.npz"), src=src, mix=mix, out=out.astype(np.complex64),
                        **{k: v.astype(np.complex64) for k, v in rec.items()})
    print("G5", out.shape, out.dtype)

    # ---- G6/G7: full Tester_Enhance.inference on one synthetic T=64 utterance (16 kHz) ---------------------------
    def run_inference(tst, obs, s0, s1):
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
        tst.inference(loader, "/tmp/golden_out")
        tst.Apply_Beamforming, tst.MISO3_inference = o_bf, o_m3
        wavs = [w[1] for w in sf_stub.written]
        return rec, wavs

    obs, s0, s1 = synth_utt_stft(7, 64, 16000)
    tst = make_tester(16000, 63 * 64 / 16000)
    rec, wavs = run_inference(tst, obs, s0, s1)
    np.savez_compressed(os.path.join(OUT, "g6_pipeline_T64.npz"), utt=np.int64(7),
                        bf=np.stack([b[0] for b in rec["bf"]]).astype(np.complex64),
                        miso1_ref=np.stack([m[0, 0] for m in rec["miso1_ref"]]).astype(np.complex64),
                        out=np.stack([o[0, 0] for o in rec["out"]]).astype(np.complex64),
                        wav0=wavs[0].astype(np.int16), wav1=wavs[1].astype(np.int16))
    print("G6", [w.shape for w in wavs], np.abs(rec["out"][0]).mean())

    # ---- G8 (BASELINE config 1): first 4 s of sample/Clean (8 kHz) -> MISO_1 forward -----------------------------
    import scipy.io.wavfile as wavfile
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, c0 = wavfile.read(os.path.join(REF, "sample/Clean/3_441c040w_445c040o_0.wav"))
        _, c1 = wavfile.read(os.path.join(REF, "sample/Clean/3_441c040w_445c040o_1.wav"))
    from oracle.pipeline_oracle import stft_chunk
    obs_w = (c0 + c1)[:32000].astype(np.float16).astype(np.float32)   # the fixture stores float16 samples
    x = stft_chunk(obs_w, 8000)[None]                                   # [1,6,501,129]
    with torch.no_grad():
        y = m1(torch.from_numpy(x)).numpy()
    mag = np.abs(y)
    np.savez_compressed(os.path.join(OUT, "g8_sample_clean_miso1.npz"),
                        obs_wav_f16=obs_w.astype(np.float16),        # data: input samples (float16 keeps it < 400 KB)
                        y_slice=y[:, :, 200:232].astype(np.complex64),
                        mag_sum_per_frame=mag.sum(axis=-1).astype(np.float32))
    print("G8", x.shape, y.shape)
    print("done ->", OUT)


if __name__ == "__main__":
    main()
