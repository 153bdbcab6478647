"""Pin the oracle (oracle/*.py) against vectors produced by the real reference (oracle/gen_golden.py).
CPU only.  Tolerances: the restatement calls the same ATen/LAPACK back-ends as the reference, so agreement
is at float32 round-off (1e-5 relative), far inside the 1e-3 parity budget of the GPU path."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2, mag_parity
from oracle import miso_oracle, mvdr_oracle, pipeline_oracle


def test_weight_spec_counts(sd1, sd3):
    # 268 tensors, 2,587,384 / 2,587,382 parameters
    assert len(sd1) == 268 and len(sd3) == 268
    assert sum(v.size for v in sd1.values()) == 2587384
    assert sum(v.size for v in sd3.values()) == 2587382
    assert sd3["encoders.0.0.conv2d.weight"].shape == (24, 16, 3, 3)
    assert sd3["decoders.6.1.deconv2d.weight"].shape == (48, 2, 3, 3)


def test_g1_miso1_forward_and_taps(sd1):
    for T in (32, 96):
        g = golden(f"g1_miso1_T{T}.npz")
        taps = {}
        y = miso_oracle.miso1_forward(torch.from_numpy(g["x"]), sd1, taps).numpy()
        assert y.dtype == np.complex64 and y.shape == (1, 2, T, 129)
        assert rel_l2(y, g["y"]) < 2e-5
        if T == 32:
            n = 0
            for k in g.files:
                if not k.startswith("tap_"):
                    continue
                v = taps[k[4:]].numpy()
                if v.size > 60000:
                    v = v[:, ::4]
                assert v.shape == g[k].shape, k
                assert rel_l2(v, g[k]) < 2e-5, k
                n += 1
            assert n >= 16


def test_g3_miso3_forward(sd3):
    g = golden("g3_miso3_T32.npz")
    y = miso_oracle.miso3_forward(torch.from_numpy(g["x"]), torch.from_numpy(g["a"]), torch.from_numpy(g["b"]), sd3)
    assert rel_l2(y.numpy(), g["y"]) < 2e-5


def test_g4_miso1_inference(sd1):
    g = golden("g4_miso1_inference_T32.npz")
    est, sel = pipeline_oracle.miso1_inference(g["x"][0], sd1, ref_ch=0)
    assert rel_l2(est[0], g["spk0"][0]) < 2e-5
    assert rel_l2(est[1], g["spk1"][0]) < 2e-5
    assert sel.shape == (6, 2)


def test_g5_mvdr_parts():
    g = golden("g5_mvdr.npz")
    p = mvdr_oracle.mvdr_parts(g["src"], g["mix"])
    assert rel_l2(p["scm_n"], g["scm_n"]) < 1e-5
    assert rel_l2(p["steer0"], g["steer0"]) < 1e-4
    assert rel_l2(p["steer1"], g["steer1"]) < 1e-4
    assert rel_l2(p["w"], g["w"]) < 1e-4
    assert rel_l2(p["out"], g["out"]) < 1e-4
    # double-precision evaluation of the same algebra stays within the complex64 noise of the reference
    p64 = mvdr_oracle.mvdr_parts(g["src"], g["mix"], dtype=np.complex128)
    assert rel_l2(p64["out"], g["out"]) < 1e-4


def test_g6_pipeline(sd1, sd3):
    g = golden("g6_pipeline_T64.npz")
    from misonet_amd.weights import synthetic_utterance
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), 63 * 64)
    mix = pipeline_oracle.stft_chunk(obs)
    clean = np.stack([pipeline_oracle.stft_chunk(s0)[0], pipeline_oracle.stft_chunk(s1)[0]])
    r = pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=0)
    assert rel_l2(r["miso1"][:, 0], g["miso1_ref"]) < 5e-5
    assert rel_l2(r["bf"], g["bf"]) < 5e-4
    rl2, bad = mag_parity(r["out"], g["out"])
    assert rl2 < 2e-4 and bad < 1e-3
    # G7: iSTFT -> int16 (tester.py:949-952).  Truncation makes +-1 LSB flips possible at round-off level.
    for s in range(2):
        w = pipeline_oracle.istft_int16(r["out"][s])
        assert w.shape == g[f"wav{s}"].shape == (63 * 64,)
        assert np.max(np.abs(w.astype(np.int32) - g[f"wav{s}"].astype(np.int32))) <= 2


def test_g12_full_size_reference(sd1, sd3):
    """G12: the real reference at the FULL bench geometry (utterance 0 of BASELINE configs[1..3], T = 1001): one
    MISO_1.forward and the whole Tester_Enhance.inference (oracle/gen_golden_full.py).  The fixture keeps every 16th frame,
    the magnitude sums of all frames and the decimated int16 waves."""
    from misonet_amd.weights import synthetic_utterance
    g = golden("g12_fullsize_T1001.npz")
    T, st = int(g["frames"]), int(g["frame_step"])
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), (T - 1) * 64)
    mix = pipeline_oracle.stft_chunk(obs)
    clean = np.stack([pipeline_oracle.stft_chunk(s0)[0], pipeline_oracle.stft_chunk(s1)[0]])
    y = miso_oracle.miso1_forward(torch.from_numpy(mix[None]), sd1).numpy()[0]
    assert rel_l2(y[:, ::st], g["miso1_fwd_frames"]) < 5e-5
    assert rel_l2(np.abs(y).astype(np.float64).sum(-1), g["miso1_fwd_magsum"]) < 2e-5
    r = pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=0)
    assert rel_l2(r["miso1"][:, 0, ::st], g["miso1_ref_frames"]) < 5e-5
    assert rel_l2(np.abs(r["miso1"][:, 0]).astype(np.float64).sum(-1), g["miso1_ref_magsum"]) < 2e-5
    assert rel_l2(r["bf"][:, ::st], g["bf_frames"]) < 5e-4
    rl2, bad = mag_parity(r["out"][:, ::st], g["out_frames"])
    assert rl2 < 2e-4 and bad < 1e-3
    assert rel_l2(np.abs(r["out"]).astype(np.float64).sum(-1), g["out_magsum"]) < 1e-4
    for s in range(2):
        w = pipeline_oracle.istft_int16(r["out"][s])
        assert w.shape == ((T - 1) * 64,)
        assert np.max(np.abs(w[::16].astype(np.int32) - g["wav_dec16"][s].astype(np.int32))) <= 2
        a = np.abs(w.astype(np.int64)).reshape(-1, 1000).sum(-1)
        assert np.max(np.abs(a - g["wav_abssum_1000"][s])) <= 2 * 1000


def _g13_inputs(g, fs):
    from misonet_amd.weights import synthetic_utterance
    T = int(g["frames"])
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), (T - 1) * 64)
    rc = int(g["ref_ch"]) if "ref_ch" in g.files else 0
    mix = pipeline_oracle.stft_chunk(obs, fs)
    clean = np.stack([pipeline_oracle.stft_chunk(s0, fs)[rc], pipeline_oracle.stft_chunk(s1, fs)[rc]])
    return mix, clean, rc


def test_g13_8khz_pipeline_reference(sd1, sd3):
    """G13: the real Tester_Enhance.inference at the committed config's 8 kHz geometry (T = 501), oracle/gen_golden_more.py."""
    g = golden("g13_pipeline_8k_T501.npz")
    st = int(g["frame_step"])
    mix, clean, _ = _g13_inputs(g, 8000)
    assert mix.shape == (6, 501, 129)
    r = pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=0)
    assert rel_l2(r["miso1"][:, 0, ::st], g["miso1_ref_frames"]) < 5e-5
    assert rel_l2(r["bf"][:, ::st], g["bf_frames"]) < 5e-4
    rl2, bad = mag_parity(r["out"][:, ::st], g["out_frames"])
    assert rl2 < 2e-4 and bad < 1e-3
    assert rel_l2(np.abs(r["out"]).astype(np.float64).sum(-1), g["out_magsum"]) < 1e-4
    for s in range(2):
        w = pipeline_oracle.istft_int16(r["out"][s], 8000)
        assert np.max(np.abs(w[::8].astype(np.int32) - g["wav_dec8"][s].astype(np.int32))) <= 2


def test_g14_ref_ch2_pipeline_reference(sd1, sd3):
    """G14: the real Tester_Enhance.inference with ref_ch = 2 (anchor of the shift alignment, microphone of the clean
    references and of the MISO3 input: tester.py:874, 889-890, 937, 1030-1038)."""
    g = golden("g14_pipeline_refch2_T64.npz")
    mix, clean, rc = _g13_inputs(g, 16000)
    assert rc == 2
    r = pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=rc)
    assert rel_l2(r["miso1"][:, rc], g["miso1_ref"]) < 5e-5
    assert rel_l2(r["bf"], g["bf"]) < 5e-4
    rl2, bad = mag_parity(r["out"], g["out"])
    assert rl2 < 2e-4 and bad < 1e-3
    for s in range(2):
        w = pipeline_oracle.istft_int16(r["out"][s])
        assert np.max(np.abs(w.astype(np.int32) - g["wav"][s].astype(np.int32))) <= 2


def test_g8_sample_clean_config1(sd1):
    """BASELINE.json configs[0]: first 4 s of sample/Clean (8 kHz, 6 mics) through one MISO_1 forward."""
    g = golden("g8_sample_clean_miso1.npz")
    x = pipeline_oracle.stft_chunk(g["obs_wav_f16"].astype(np.float32), 8000)[None]
    assert x.shape == (1, 6, 501, 129)
    y = miso_oracle.miso1_forward(torch.from_numpy(x), sd1).numpy()
    assert rel_l2(y[:, :, 200:232], g["y_slice"]) < 5e-5
    assert rel_l2(np.abs(y).sum(-1), g["mag_sum_per_frame"]) < 2e-5


def test_g9_utterance_wise_mvdr(sd1):
    """Tester_Beamforming's utterance_flag path (tester.py:340-449): golden from the real reference."""
    from misonet_amd.weights import synthetic_utterance
    from misonet_amd.stft import split_chunks
    g = golden("g9_utterance_mvdr.npz")
    frames, gap = int(g["frames"]), int(g["gap"])
    chunk = (frames - 1) * 64
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), 2 * chunk - gap)
    po, gap2 = split_chunks(obs, chunk)
    p0, _ = split_chunks(s0, chunk)
    p1, _ = split_chunks(s1, chunk)
    assert gap2 == gap
    obs_s = [pipeline_oracle.stft_chunk(p) for p in po]
    cl_s = [np.stack([pipeline_oracle.stft_chunk(a)[0], pipeline_oracle.stft_chunk(b)[0]]) for a, b in zip(p0, p1)]
    wav = pipeline_oracle.beamform_utterance(obs_s, cl_s, gap, sd1)
    assert wav.shape == (2, g["wav0"].shape[0])
    for s in range(2):
        d = np.abs(wav[s].astype(np.int32) - g[f"wav{s}"].astype(np.int32))
        assert d.max() <= 2, d.max()


def test_g10_three_speaker_pit():
    """MISO1_Inference with num_spks = 3: the PIT alignment over all 3! permutations (tester.py:1053-1064); golden from
    the real reference (oracle/gen_golden_pit3.py).  The shifts pick non-trivial permutations incl. 3-cycles."""
    from misonet_amd import weights as W
    g = golden("g10_miso1_inference_S3_T32.npz")
    sd = W.make_state_dict(W.miso1_spec(num_spks=3), seed=2)
    est, sel = pipeline_oracle.miso1_inference(g["x"][0], sd, ref_ch=0)
    assert est.shape == (3, 6, 32, 129)
    assert len({tuple(r) for r in sel.tolist()}) >= 3, sel
    assert rel_l2(est[:, :, ::2], g["est_even"]) < 2e-5
    assert rel_l2(np.abs(est).sum(-1), g["mag_sum"]) < 2e-5


@pytest.mark.parametrize("nt", ["gLN", "cLN", "BN"])
def test_g11_norm_type_variants(nt):
    """G11: the real reference built with norm_type = gLN / cLN / anything else (BatchNorm1d, eval) -- the outer norms of the
    TemporalBlocks (model.py:530,535,570-581).  The oracle restates chose_norm; the spec generator produced key names, order
    and shapes the reference's own load_state_dict accepted (oracle/gen_golden_norm.py)."""
    from misonet_amd import weights as W
    g = golden(f"g11_norm_{nt}_T40.npz")
    sd = W.make_state_dict(W.miso1_spec(norm_type=nt), seed=3)
    y = miso_oracle.miso1_forward(torch.from_numpy(g["x"]), sd, norm_type=nt).numpy()
    assert rel_l2(y, g["y"]) < 2e-5
    # the variants are really different networks: the IN oracle on the same weights is far from the golden
    y_in = miso_oracle.miso1_forward(torch.from_numpy(g["x"]), sd, norm_type="IN").numpy()
    assert rel_l2(y_in, g["y"]) > 1e-2
    if nt == "cLN":
        sd3 = W.make_state_dict(W.miso3_spec(norm_type=nt), seed=4)
        y3 = miso_oracle.miso3_forward(torch.from_numpy(g["x"]), torch.from_numpy(g["a"]), torch.from_numpy(g["b"]), sd3,
                                       norm_type=nt).numpy()
        assert rel_l2(y3, g["y3"]) < 2e-5
