"""GPU tests of the on-device STFT front-end (csrc/stft.hip) against the reference's SciPy contract
(dataloader/data.py:505-522,540-544, restated in oracle/pipeline_oracle.stft_chunk) and of the waveform entry of the
pipeline."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from test_gpu_parity import nets, _assert_parity, _need_gpu      # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,M,B", [(63 * 64, 6, 2), (64000, 6, 1), (1000, 3, 2), (64 * 5 + 17, 1, 1)])
def test_stft_hip_vs_scipy_contract(L, M, B):
    _need_gpu()
    from misonet_amd import stft as S
    from oracle import pipeline_oracle
    r = np.random.default_rng(L + M)
    wav = (0.05 * r.standard_normal((B, L, M))).astype(np.float32)
    got = S.stft_hip(torch.from_numpy(wav).cuda()).cpu().numpy()
    T = L // 64 + 1
    assert got.shape == (B, M, T, 129)
    for b in range(B):
        # scipy pads the tail to a whole hop ("padded=True"): same frames as the zero-extended signal
        ref = pipeline_oracle.stft_chunk(wav[b])
        assert ref.shape[1] >= T - 1
        n = min(T, ref.shape[1])
        e = rel_l2(got[b][:, :n], ref[:, :n])
        print(f"[stft] L={L} M={M} b={b}: rel_l2 {e:.3e}")
        assert e < 2e-6
    # and against the torch front-end used elsewhere
    tor = S.stft(torch.from_numpy(wav).cuda().permute(0, 2, 1)).cpu().numpy()
    assert rel_l2(got, tor) < 2e-6


def test_enhance_wav_equals_enhance_on_stft(nets):
    """Waveform entry == spectrogram entry fed with the reference-contract STFT (both precisions of the same chunk)."""
    import misonet_amd as mz
    from misonet_amd.weights import synthetic_utterance
    from oracle import pipeline_oracle
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    n = 63 * 64
    utts = [synthetic_utterance(u, n) for u in (7, 11)]
    wav = torch.from_numpy(np.stack([u[0] for u in utts])).cuda()                         # [B, L, 6]
    cwav = torch.from_numpy(np.stack([np.stack([u[1][:, 0], u[2][:, 0]], axis=1) for u in utts])).cuda()   # [B, L, 2]
    out_w, ex = enh.enhance_wav(wav, cwav, want_bf=True)
    mix = torch.from_numpy(np.stack([pipeline_oracle.stft_chunk(u[0]) for u in utts])).cuda()
    clean = torch.from_numpy(np.stack([np.stack([pipeline_oracle.stft_chunk(u[1])[0], pipeline_oracle.stft_chunk(u[2])[0]])
                                       for u in utts])).cuda()
    out_s, ex_s = enh.enhance(mix, clean, want_bf=True)
    _assert_parity(out_w.cpu().numpy(), out_s.cpu().numpy(), "enhance_wav vs enhance(stft)")
    _assert_parity(ex["bf"].cpu().numpy(), ex_s["bf"].cpu().numpy(), "enhance_wav bf vs enhance(stft) bf")
    with pytest.raises(ValueError):
        enh.enhance_wav(wav[:, :, :5])


def test_utterance_wise_mvdr_vs_reference_golden(nets, sd1):
    """Tester_Beamforming's utterance_flag mode (tester.py:340-449) -- SCMs over the whole recording.
    Against the golden produced by the real reference (G9) and the oracle."""
    import misonet_amd as mz
    from conftest import golden
    from misonet_amd.weights import synthetic_utterance
    from misonet_amd.stft import split_chunks
    from oracle import pipeline_oracle
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    g = golden("g9_utterance_mvdr.npz")
    frames, gap = int(g["frames"]), int(g["gap"])
    chunk = (frames - 1) * 64
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), 2 * chunk - gap)
    po, _ = split_chunks(obs, chunk)
    p0, _ = split_chunks(s0, chunk)
    p1, _ = split_chunks(s1, chunk)
    obs_s = [pipeline_oracle.stft_chunk(p) for p in po]
    cl_s = [np.stack([pipeline_oracle.stft_chunk(a)[0], pipeline_oracle.stft_chunk(b)[0]]) for a, b in zip(p0, p1)]
    wav = enh.beamform_utterance([torch.from_numpy(o) for o in obs_s], [torch.from_numpy(c) for c in cl_s], gap)
    assert wav.shape == (2, g["wav0"].shape[0]) and wav.dtype == np.int16
    for s in range(2):
        d = np.abs(wav[s].astype(np.int32) - g[f"wav{s}"].astype(np.int32))
        scale = np.abs(g[f"wav{s}"].astype(np.int32)).max()
        print(f"[utt-mvdr] spk{s}: max |diff| {d.max()} LSB of peak {scale}")
        assert d.max() <= 1


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
def test_tester_beamforming_class_drop_in(sd1, sd3, tmp_path, mode):
    """misonet_amd.tester.Tester_Beamforming: the reference's harness class (tester.py:259-449) with the arguments run.py
    passes, built from MISO_1 ALONE (a separation-only pipeline: misonet_pipeline_create with miso3 = NULL).
    utterance_flag = True against the golden of the real class (G9: two splits, last one trimmed by gap);
    utterance_flag = False (MISO1 -> MVDR per chunk = BASELINE configs[2]) against the oracle's per-chunk beamformer."""
    import misonet_amd as mz
    from conftest import golden
    from misonet_amd import weights as W, stft as S
    from misonet_amd.tester import Tester_Beamforming
    from misonet_amd.weights import synthetic_utterance
    from misonet_amd.stft import split_chunks
    from oracle import pipeline_oracle
    from test_gpu_parity import _need_gpu
    _need_gpu()
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1)
    m1.eval().set_precision(mode)
    g = golden("g9_utterance_mvdr.npz")
    frames, gap = int(g["frames"]), int(g["gap"])
    chunk = (frames - 1) * 64
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), 2 * chunk - gap)
    po, _ = split_chunks(obs, chunk)
    p0, _ = split_chunks(s0, chunk)
    p1, _ = split_chunks(s1, chunk)
    obs_d = {str(k): torch.from_numpy(pipeline_oracle.stft_chunk(p))[None] for k, p in enumerate(po)}      # [1,6,T,F]
    s0_d = {str(k): torch.from_numpy(pipeline_oracle.stft_chunk(p))[None] for k, p in enumerate(p0)}
    s1_d = {str(k): torch.from_numpy(pipeline_oracle.stft_chunk(p))[None] for k, p in enumerate(p1)}
    item = (obs_d, s0_d, s1_d, [gap], ["rec"])
    args = dict(fs=16000, window="hann", length=256, overlap=192)
    tst = Tester_Beamforming("SMS_WSJ", [item], [item], [item], m1, 6, 0, 2, chunk / 16000, str(tmp_path / "u"), 0, True,
                             False, True, **args)
    res = tst.test()
    assert sorted(res) == ["cv_dev93", "test_eval92"]
    for sub in res:
        wav = res[sub]["rec"]
        assert wav.shape == (2, g["wav0"].shape[0]) and wav.dtype == np.int16
        for s in range(2):
            d = np.abs(wav[s].astype(np.int32) - g[f"wav{s}"].astype(np.int32))
            print(f"[Tester_Beamforming {mode}] {sub} spk{s}: max |diff| {d.max()} LSB vs the real class (G9)")
            assert d.max() <= 1
            v, fs = S.read_wav_pcm24(str(tmp_path / "u" / sub / f"rec_{s}.wav"))
            assert fs == 16000 and np.array_equal(v[:, 0], wav[s].astype(np.int32) << 8)
    # chunk-wise branch + the training-set directory
    tst = Tester_Beamforming("SMS_WSJ", [item], [], [], m1, 6, 0, 2, chunk / 16000, str(tmp_path / "c"), 0, True,
                             True, False, **args)
    res = tst.test()
    assert sorted(res) == ["train_si284"]
    wav = res["train_si284"]["rec"]
    pcs = [[], []]
    for k in range(2):
        mx = obs_d[str(k)][0].numpy()
        cl = np.stack([s0_d[str(k)][0, 0].numpy(), s1_d[str(k)][0, 0].numpy()])
        bf = pipeline_oracle.enhance_utterance(mx, cl, sd1, sd3, ref_ch=0)["bf"]          # [S,T,F] per-chunk MVDR (tester.py:917-924)
        for s in range(2):
            pcs[s].append(pipeline_oracle.istft_int16(bf[s]))
    for s in range(2):
        ref = S.stitch_int16(pcs[s], gap)
        d = np.abs(wav[s].astype(np.int32) - ref.astype(np.int32))
        print(f"[Tester_Beamforming {mode}] chunk-wise spk{s}: max |diff| {d.max()} LSB vs the oracle")
        assert wav[s].shape == ref.shape and d.max() <= 1
    # the separation-only pipeline refuses the MISO3 stages loudly
    with pytest.raises(RuntimeError):
        tst._enh.enhance(obs_d["0"].cuda())
    with pytest.raises(TypeError):
        Tester_Beamforming("SMS_WSJ", [], [], [], object(), 6, 0, 2, 4.0, str(tmp_path), 0, True, False, True, **args)


@pytest.mark.parametrize("N,T", [(1, 2), (2, 5), (3, 64), (2, 200), (2, 501)])
def test_hip_istft_vs_scipy(N, T):
    """misonet_istft (csrc/stft.hip istft_k) against the reference's own synthesis, scipy.signal.istft(hann, 256, 192) of
    ``spec * scale`` in float64 (tester.py:949-952, 979-990): float32 output to 2e-6 of full scale, int16 within 1 LSB (the
    truncating cast flips where the float64 value sits within round-off of an integer), shortest input T = 2 included."""
    import scipy.signal
    from misonet_amd import stft as S
    from test_gpu_parity import _need_gpu
    _need_gpu()
    r = np.random.default_rng(N * 1000 + T)
    z = (r.standard_normal((N, T, 129)) + 1j * r.standard_normal((N, T, 129))).astype(np.complex64) * 30
    want = np.stack([scipy.signal.istft(z[i].T.astype(np.complex128) / 128.0, fs=16000, window="hann", nperseg=256,
                                        noverlap=192)[1][: (T - 1) * 64] for i in range(N)])
    zd = torch.from_numpy(z).cuda()
    y = S.istft(zd)
    assert y.is_cuda and y.dtype == torch.float32 and tuple(y.shape) == (N, (T - 1) * 64)
    e = np.abs(y.cpu().numpy() - want).max() / np.abs(want).max()
    assert e <= 2e-6, e
    q = S.istft_int16(zd / 300).cpu().numpy().astype(np.int32)
    ref = (want / 300 * 32767).astype(np.int16).astype(np.int32)
    assert q.shape == ref.shape and np.abs(q - ref).max() <= 1 and np.mean(q != ref) < 1e-3
    # leading dimensions are kept, and the CPU path (torch.istft) agrees
    y4 = S.istft(zd.reshape(1, N, T, 129))
    assert tuple(y4.shape) == (1, N, (T - 1) * 64) and torch.equal(y4[0], y)
    assert np.abs(S.istft(torch.from_numpy(z)).numpy() - want).max() / np.abs(want).max() <= 2e-6


def test_enhance_recording_loader_side_drop_in(nets, sd1, sd3, tmp_path):
    """recording in -> ``<wav>_{0,1}.wav`` out with no host STFT dicts (AudioDataset_Test.__getitem__,
    dataloader/data.py:524-597, + Tester_Enhance.inference, tester.py:846-975, as ONE device-side call).

    (a) golden G13 = the REAL reference harness on one 4 s recording at 8 kHz: the waves within 1 LSB;
    (b) a 2.4-chunk recording of 12 microphones with ``num_ch_utilize = 6`` (every second microphone, data.py:544) against the
        oracle run chunk by chunk on the sub-sampled, zero-padded pieces: <= 1 LSB, padded tail trimmed, files byte-exact."""
    import misonet_amd as mz
    from misonet_amd import stft as S
    from misonet_amd.weights import synthetic_utterance
    from oracle import pipeline_oracle
    from conftest import golden
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    # ---- (a) ----
    g = golden("g13_pipeline_8k_T501.npz")
    T = int(g["frames"])
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), (T - 1) * 64)
    wav = enh.enhance_recording(obs, (s0, s1), chunk_size=(T - 1) * 64, fs=8000)
    assert wav.shape == (2, (T - 1) * 64) and wav.dtype == np.int16
    for s in range(2):
        d = np.abs(wav[s][::8].astype(np.int32) - g["wav_dec8"][s].astype(np.int32))
        assert d.max() <= 1, d.max()
    # ---- (b) ----
    chunk = 47 * 64
    L = 2 * chunk + 1200
    r = np.random.default_rng(21)
    src = [(0.05 * r.standard_normal((L, 12))).astype(np.float32) for _ in range(2)]
    rec = src[0] + src[1]
    got = enh.enhance_recording(rec, src, num_ch_utilize=6, chunk_size=chunk, max_batch=2, save_path=str(tmp_path / "rec7"),
                                fs=16000)
    assert got.shape == (2, L)
    mics = list(range(0, 12, 2))
    want = []
    for k in range(3):
        def piece(x):
            p = x[k * chunk:(k + 1) * chunk][:, mics]
            return np.pad(p, ((0, chunk - p.shape[0]), (0, 0)))
        mix = pipeline_oracle.stft_chunk(piece(rec))
        clean = np.stack([pipeline_oracle.stft_chunk(piece(src[0]))[0], pipeline_oracle.stft_chunk(piece(src[1]))[0]])
        o = pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=0)["out"]
        want.append(np.stack([pipeline_oracle.istft_int16(o[s]) for s in range(2)]))
    want = np.concatenate(want, axis=1)[:, :L]
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    print(f"[recording] 3 chunks, 12 -> 6 mics: max |diff| = {d.max()} LSB, {float((d > 0).mean()):.2e} of the samples differ")
    assert d.max() <= 1
    for s in range(2):
        v, fs = S.read_wav_pcm24(str(tmp_path / f"rec7_{s}.wav"))
        assert fs == 16000 and np.array_equal(v[:, 0], got[s].astype(np.int32) << 8)
    with pytest.raises(ValueError):
        enh.enhance_recording(rec, src, num_ch_utilize=4, chunk_size=chunk)      # [0:12:3] = 4 microphones, the networks take 6
