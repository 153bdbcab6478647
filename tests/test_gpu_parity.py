"""GPU parity tests (run on the MI355X box with ``-m gpu``): the HIP path, called through the C ABI via the
host-side mirror classes, against (a) the committed goldens produced by the real reference and (b) the CPU
oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star / SURVEY.md 8(d)): relative L2 error of the complex-spectrogram magnitudes
<= 1e-3, and element-wise | |x^|-|x| | <= 1e-3*|x| + 1e-3*median|x| (violations allowed on < 0.1 % of bins).
"""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2, mag_parity

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


# Every test that takes ``nets`` (here and in test_gpu_frontend.py) runs three times: in the exact-f32 MFMA mode, in "f32w" (the
# same matrix cores with the DenseBlock convs in Winograd F(2x2, 3x3) form, conv_wino.hip) and in the bench's headline
# arithmetic "bf16x6" (fp32-faithful split-bf16; also the library default) -- same goldens, same tolerances, int16 waves
# still within 1 LSB.
HEADLINE_MODES = ("f32", "f32w", "bf16x6")


@pytest.fixture(scope="module", params=HEADLINE_MODES)
def nets(request, sd1, sd3):
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN")
    m1.cuda(0)
    m1.load_state_dict(sd1)
    m1.eval().set_precision(request.param)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN")
    m3.cuda(0)
    m3.load_state_dict(sd3)
    m3.eval().set_precision(request.param)
    return m1, m3


def _assert_parity(got, want, what, tol=TOL):
    r, bad = mag_parity(got, want, tol)
    rc = rel_l2(got, want)
    print(f"[parity] {what}: rel_l2(mag)={r:.3e} rel_l2(complex)={rc:.3e} bad_frac={bad:.2e}")
    assert np.isfinite(r) and r <= tol, f"{what}: magnitude rel-L2 {r:.3e} > {tol}"
    assert bad <= 1e-3, f"{what}: {bad:.2e} of bins outside the element-wise bound"


def test_miso1_stage_taps_vs_oracle(nets, sd1, request):
    """Every stage of one forward (T=32) against the oracle's taps: localises a wrong kernel variant."""
    from oracle import miso_oracle
    m1, _ = nets
    g = golden("g1_miso1_T32.npz")
    x = torch.from_numpy(g["x"])
    taps = {}
    y_ref = miso_oracle.miso1_forward(x, sd1, taps).numpy()
    y_shared = m1(x.cuda()).cpu().numpy()              # default workspace: buffers with disjoint lifetimes share memory
    with pytest.raises(Exception):
        m1.tap("enc0_conv", 1, 32)                     # ... so intermediate taps are refused (dec6 = the output is not)
    m1.keep_activations(True)
    request.addfinalizer(lambda: m1.keep_activations(False))
    y = m1(x.cuda()).cpu().numpy()
    assert np.array_equal(y, y_shared)                 # the memory plan does not change a bit of the result
    names = ["enc0_conv"] + [f"enc{b}" for b in range(7)] + ["tcn_out"] + [f"dec{b}" for b in range(7)]
    worst = 0.0
    for nm in names:
        ref = taps[nm].numpy()
        if ref.ndim == 3:
            ref = ref[..., None]
        got = m1.tap(nm, 1, 32).cpu().numpy()
        assert got.shape == ref.shape, (nm, got.shape, ref.shape)
        e = rel_l2(got, ref)
        print(f"[tap] {nm:10s} shape={got.shape} rel_l2={e:.3e}")
        worst = max(worst, e)
        assert e < 1e-3, f"stage {nm}: rel_l2 {e:.3e}"
    _assert_parity(y, y_ref, "miso1 T=32 vs oracle")
    _assert_parity(y, g["y"], "miso1 T=32 vs reference golden")


@pytest.mark.parametrize("T", [32, 96])
def test_miso1_vs_golden(nets, T):
    m1, _ = nets
    g = golden(f"g1_miso1_T{T}.npz")
    y = m1(torch.from_numpy(g["x"]).cuda())
    assert y.dtype == torch.complex64 and tuple(y.shape) == (1, 2, T, 129)
    _assert_parity(y.cpu().numpy(), g["y"], f"miso1 T={T} vs reference golden")


def test_miso3_vs_golden(nets):
    _, m3 = nets
    g = golden("g3_miso3_T32.npz")
    y = m3(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["a"]).cuda(), torch.from_numpy(g["b"]).cuda())
    assert tuple(y.shape) == (1, 1, 32, 129)
    _assert_parity(y.cpu().numpy(), g["y"], "miso3 T=32 vs reference golden")


@pytest.mark.parametrize("B,T", [(3, 40), (2, 130), (1, 5), (2, 257)])
def test_miso1_ragged_batched_vs_oracle(nets, sd1, B, T):
    """Frame counts that are not multiples of the 32/128-frame tiles, batch > 1 (per-sample norms)."""
    from oracle import miso_oracle
    m1, _ = nets
    r = np.random.default_rng(B * 1000 + T)
    x = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
    x[1:] *= 3.0                                             # different scales per sample
    y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
    y_ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd1).numpy() for b in range(B)])
    _assert_parity(y, y_ref, f"miso1 B={B} T={T} vs oracle")


@pytest.mark.parametrize("T", [2, 3, 4])
def test_miso1_shortest_inputs_vs_oracle(nets, sd1, T, request):
    """The shortest inputs the reference accepts (T = 2: its instance norms need more than one element at the F = 1
    bottleneck, model.py:89): every tile is ragged in T and the dilated TCN taps (dilation up to 64) fall outside the signal.
    From T = 4 on the usual bound against the float32 oracle holds.  At T = 2, 3 the FUNCTION ITSELF is ill-conditioned: the
    TCN is 28 instance norms over 2-3 values, and a channel whose frames nearly coincide is re-amplified by 1 / sqrt(eps) =
    316 at every norm until it saturates -- the float64 oracle turns the 5e-5 difference between this build's and its own
    encoder output into 2e-2 at the TCN output (and the float32 oracle's 8e-5 difference into 2e-5: the direction of the
    round-off decides), the float32 oracle moves by 9e-4 at T = 3 when only its thread count changes (all measured,
    tools/experiments/tcn_*.py, round 4).  A bound on the END result is therefore meaningless there; what is checked instead
    is every part on ITS OWN input, against the oracle in float64: the encoder (well-conditioned) end to end, the TCN on the
    encoder output this build produced, and that the result is finite and of the right size.  (This case made the TCN subtract
    its means before scaling and accumulate its statistics about a pivot -- tcn.hip -- which took the TCN's own error at
    T = 2 from 4e-4 to 2e-5 ... 1e-4.)"""
    from oracle import miso_oracle
    m1, _ = nets
    r = np.random.default_rng(77 + T)
    x = (r.standard_normal((2, 6, T, 129)) + 1j * r.standard_normal((2, 6, T, 129))).astype(np.complex64)
    if T >= 4:
        y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
        y_ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd1).numpy() for b in range(2)])
        _assert_parity(y, y_ref, f"miso1 T={T} vs oracle")
        return
    m1.keep_activations(True)
    request.addfinalizer(lambda: m1.keep_activations(False))
    y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
    assert y.shape == (2, 2, T, 129) and np.isfinite(y.view(np.float32)).all()
    enc5 = m1.tap("enc5", 2, T).cpu().numpy().astype(np.float64)
    enc6 = m1.tap("enc6", 2, T).cpu().numpy().astype(np.float64)[..., 0]               # [B,128,T]
    tcn = m1.tap("tcn_out", 2, T).cpu().numpy().astype(np.float64)[..., 0]
    for b in range(2):
        t64 = {}
        with miso_oracle.precision(torch.float64):
            truth = miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]).to(torch.complex128), sd1, t64).numpy()
            tcn_local = miso_oracle.tcn_forward(torch.from_numpy(enc6[b:b + 1]), sd1).numpy()
        # yardstick for the TCN's conditioning ON THIS INPUT: the float32 oracle (stock torch CPU) against the float64 one.  Whether
        # a channel's two or three frames nearly coincide somewhere in the 28 instance norms depends on the last bits of the
        # encoder output, i.e. on the arithmetic mode that produced it (round 5: the f32w encoder output hit such a channel in
        # sample 1 at T = 2 -- float32 evaluations of the TCN then differ by 0.1-0.3 from float64 whoever computes them)
        tcn_f32 = miso_oracle.tcn_forward(torch.from_numpy(enc6[b:b + 1].astype(np.float32)), sd1).numpy()
        e32 = rel_l2(tcn_f32, tcn_local)
        e5 = rel_l2(enc5[b:b + 1], t64["enc5"].numpy())
        e6 = rel_l2(enc6[b:b + 1], t64["enc6"].numpy()[..., 0])
        et = rel_l2(tcn[b:b + 1], tcn_local)
        eo = mag_parity(y[b:b + 1], truth)[0]
        print(f"[parity] miso1 T={T} sample {b}: encoder 5 {e5:.2e}, encoder 6 {e6:.2e} vs float64 truth; TCN on its own input "
              f"{et:.2e} (float32 oracle on the same input: {e32:.2e}); end result {eo:.2e} (ill-conditioned, not bounded at 1e-3)")
        assert e5 <= 2e-5 and e6 <= 1e-3, (T, b, e5, e6)      # encoder 6 is one instance norm over T values per channel
        # f32 and bf16x6 keep the strict bound; only the mode whose encoder output lands on the ill-conditioned channel (f32w, T = 2)
        # is measured against the float32 oracle's own distance on that input (a relaxed bound everywhere would let a
        # real TCN regression through)
        bound = max(5e-3, 10.0 * e32) if m1.precision == "f32w" else 5e-3
        assert et <= bound, (m1.precision, T, b, et, e32)
        assert eo <= 0.2, (T, b, eo)


def test_forward_errors(nets):
    m1, m3 = nets
    with pytest.raises(ValueError):                                                 # T = 1: the reference raises the same
        m1(torch.zeros((1, 6, 1, 129), dtype=torch.complex64, device="cuda"))       # ValueError (InstanceNorm, one element)
    with pytest.raises(Exception):
        m1(torch.zeros((0, 6, 8, 129), dtype=torch.complex64, device="cuda"))       # empty batch
    with pytest.raises(ValueError):
        m1(torch.zeros((1, 6, 8, 257), dtype=torch.complex64, device="cuda"))       # F != 129
    with pytest.raises(ValueError):
        m1(torch.zeros((1, 5, 8, 129), dtype=torch.complex64, device="cuda"))       # wrong mic count
    with pytest.raises(TypeError):
        m1(torch.zeros((1, 6, 8, 129), dtype=torch.float32, device="cuda"))
    with pytest.raises(RuntimeError):
        m1(torch.zeros((1, 6, 8, 129), dtype=torch.complex64))                      # CPU tensor to a device model
    x = torch.zeros((1, 6, 8, 129), dtype=torch.complex64, device="cuda")
    x[0, 0, 0, 0] = float("nan")
    with pytest.raises(FloatingPointError):
        m1(x)                                                                        # model.py:109-110 -> error, not pdb
    with pytest.raises(RuntimeError):
        m1.load_state_dict({"bogus": torch.zeros(1)})


def test_mvdr_vs_golden_and_oracle():
    _need_gpu()
    from misonet_amd import Apply_Beamforming
    from oracle import mvdr_oracle
    g = golden("g5_mvdr.npz")
    out, dbg = Apply_Beamforming(g["src"], g["mix"], return_debug=True)
    assert out.dtype == torch.complex64 and tuple(out.shape) == (1, 24, 129) and out.device.type == "cpu"
    e_s = rel_l2(dbg["steer1"].cpu().numpy(), g["steer1"])
    e_w = rel_l2(dbg["w"].cpu().numpy(), g["w"])
    e_o = rel_l2(out.numpy(), g["out"])
    print(f"[mvdr] golden: steer {e_s:.3e} w {e_w:.3e} out {e_o:.3e}")
    assert e_s < 1e-4 and e_w < 1e-4 and e_o < 1e-4
    # other shapes: M = 4 mics, T = 50 frames (ragged), B = 3, F = 17; torch device inputs, non-contiguous views
    r = np.random.default_rng(5)
    for (B, F, M, T) in [(3, 17, 4, 50), (2, 129, 6, 300), (1, 9, 2, 7), (1, 5, 8, 33)]:
        src = (r.standard_normal((B, F, M, T)) + 1j * r.standard_normal((B, F, M, T))).astype(np.complex64)
        mix = src + 0.5 * (r.standard_normal((B, F, M, T)) + 1j * r.standard_normal((B, F, M, T))).astype(np.complex64)
        ref = mvdr_oracle.mvdr_parts(src, mix, dtype=np.complex128)["out"]
        s_dev = torch.from_numpy(np.ascontiguousarray(src.transpose(0, 2, 3, 1))).cuda().permute(0, 3, 1, 2)  # view
        o = Apply_Beamforming(s_dev, torch.from_numpy(mix).cuda())
        assert o.device.type == "cuda"
        e = rel_l2(o.cpu().numpy(), ref)
        print(f"[mvdr] B={B} F={F} M={M} T={T}: out {e:.3e}")
        assert e < 1e-4


def test_pit_select_vs_oracle():
    _need_gpu()
    from misonet_amd.beamform import pit_select
    from oracle import mvdr_oracle
    r = np.random.default_rng(9)
    B, S, T, F = 5, 2, 37, 129
    a = (r.standard_normal((B, S, T, F)) + 1j * r.standard_normal((B, S, T, F))).astype(np.complex64)
    c = a[:, ::-1].copy() + 0.1 * (r.standard_normal((B, S, T, F)) + 1j * r.standard_normal((B, S, T, F))).astype(np.complex64)
    c[2] = a[2] + 0.1                                         # identity permutation for one item
    sel, dist = pit_select(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda(), return_dist=True)
    sel_ref, dist_ref = mvdr_oracle.pit_select(a, c)
    assert np.array_equal(sel.cpu().numpy(), sel_ref)
    assert rel_l2(dist.cpu().numpy(), dist_ref) < 1e-5
    # exact tie -> first permutation (torch.argmin / np.argmin convention)
    sel_t = pit_select(torch.from_numpy(a).cuda(), torch.from_numpy(np.stack([a[:, 0], a[:, 0]], 1)).cuda())
    assert np.array_equal(sel_t.cpu().numpy(), np.tile([0, 1], (B, 1)))


def _utt_inputs(u, n_frames):
    from misonet_amd.weights import synthetic_utterance
    from oracle import pipeline_oracle
    obs, s0, s1 = synthetic_utterance(u, (n_frames - 1) * 64)
    mix = pipeline_oracle.stft_chunk(obs)
    clean = np.stack([pipeline_oracle.stft_chunk(s0)[0], pipeline_oracle.stft_chunk(s1)[0]])
    return mix, clean


def test_pipeline_vs_golden_and_oracle(nets, sd1, sd3):
    """MISO1 x6 -> alignment -> MVDR x2 -> MISO3 x2 on device, B = 2 different utterances, against the reference
    golden (utterance 7) and the oracle run utterance by utterance (B = 1 semantics)."""
    import misonet_amd as mz
    from oracle import pipeline_oracle
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    g = golden("g6_pipeline_T64.npz")
    ins = [_utt_inputs(7, 64), _utt_inputs(11, 64)]
    mix = torch.from_numpy(np.stack([i[0] for i in ins])).cuda()
    clean = torch.from_numpy(np.stack([i[1] for i in ins])).cuda()
    out, extra = enh.enhance(mix, clean, want_bf=True, want_miso1=True)
    out, bf, m1o = out.cpu().numpy(), extra["bf"].cpu().numpy(), extra["miso1"].cpu().numpy()
    _assert_parity(m1o[0][:, 0], g["miso1_ref"], "pipeline miso1@ref vs golden")
    _assert_parity(bf[0], g["bf"], "pipeline bf vs golden")
    _assert_parity(out[0], g["out"], "pipeline miso3 vs golden")
    for b, (mx, cl) in enumerate(ins):
        r = pipeline_oracle.enhance_utterance(mx, cl, sd1, sd3, ref_ch=0)
        _assert_parity(m1o[b], r["miso1"], f"pipeline[{b}] miso1 (all mics) vs oracle")
        _assert_parity(bf[b], r["bf"], f"pipeline[{b}] bf vs oracle")
        _assert_parity(out[b], r["out"], f"pipeline[{b}] miso3 vs oracle")
    # G7: int16 wave of the golden utterance (tester.py:949-952); +-1 LSB: round-off across the int16 truncation
    wav = enh.to_wav_int16([torch.from_numpy(out[0]).cuda()], gap=0)
    for s in range(2):
        d = np.abs(wav[s].astype(np.int32) - g[f"wav{s}"].astype(np.int32))
        print(f"[wav] spk{s}: max |diff| = {d.max()} LSB, mean {d.mean():.3f}")
        assert d.max() <= 1


def test_pipeline_8khz_and_ref_ch_vs_reference_goldens(nets):
    """G13 / G14 (oracle/gen_golden_more.py): the REAL ``Tester_Enhance.inference`` at the committed config's 8 kHz geometry
    (T = 501; the STFT scaling does not depend on fs, only the frame count does) and with ``ref_ch = 2`` (alignment anchor,
    clean-reference microphone and MISO3 input all move with it, tester.py:874, 889-890, 937, 1030-1038)."""
    import misonet_amd as mz
    from misonet_amd.weights import synthetic_utterance
    from oracle import pipeline_oracle
    m1, m3 = nets
    # ---- G13 ----
    g = golden("g13_pipeline_8k_T501.npz")
    T, st = int(g["frames"]), int(g["frame_step"])
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), (T - 1) * 64)
    mix = pipeline_oracle.stft_chunk(obs, 8000)
    clean = np.stack([pipeline_oracle.stft_chunk(s0, 8000)[0], pipeline_oracle.stft_chunk(s1, 8000)[0]])
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    out, ex = enh.enhance(torch.from_numpy(mix[None]).cuda(), torch.from_numpy(clean[None]).cuda(), want_bf=True, want_miso1=True)
    _assert_parity(ex["miso1"][0].cpu().numpy()[:, 0, ::st], g["miso1_ref_frames"], "8 kHz T=501 miso1@ref vs golden G13")
    _assert_parity(ex["bf"][0].cpu().numpy()[:, ::st], g["bf_frames"], "8 kHz T=501 bf vs golden G13")
    _assert_parity(out[0].cpu().numpy()[:, ::st], g["out_frames"], "8 kHz T=501 miso3 vs golden G13")
    assert rel_l2(np.abs(out[0].cpu().numpy()).astype(np.float64).sum(-1), g["out_magsum"]) <= 1e-4
    wav = enh.to_wav_int16([out[0]], gap=0)
    for s in range(2):
        d = np.abs(wav[s][::8].astype(np.int32) - g["wav_dec8"][s].astype(np.int32))
        assert d.max() <= 1, d.max()
        a = np.abs(wav[s].astype(np.int64)).reshape(-1, 1000).sum(-1)
        assert np.max(np.abs(a - g["wav_abssum_1000"][s])) <= 1000
    # ---- G14 ----
    g = golden("g14_pipeline_refch2_T64.npz")
    T, rc = int(g["frames"]), int(g["ref_ch"])
    obs, s0, s1 = synthetic_utterance(int(g["utt"]), (T - 1) * 64)
    mix = pipeline_oracle.stft_chunk(obs)
    clean = np.stack([pipeline_oracle.stft_chunk(s0)[rc], pipeline_oracle.stft_chunk(s1)[rc]])
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=rc)
    out, ex = enh.enhance(torch.from_numpy(mix[None]).cuda(), torch.from_numpy(clean[None]).cuda(), want_bf=True, want_miso1=True)
    _assert_parity(ex["miso1"][0].cpu().numpy()[:, rc], g["miso1_ref"], "ref_ch=2 miso1@ref vs golden G14")
    _assert_parity(ex["bf"][0].cpu().numpy(), g["bf"], "ref_ch=2 bf vs golden G14")
    _assert_parity(out[0].cpu().numpy(), g["out"], "ref_ch=2 miso3 vs golden G14")
    wav = enh.to_wav_int16([out[0]], gap=0)
    for s in range(2):
        assert np.abs(wav[s].astype(np.int32) - g["wav"][s].astype(np.int32)).max() <= 1


def test_pipeline_without_clean_and_ref_ch(nets, sd1, sd3):
    """clean=None skips the clean re-ordering; ref_ch != 0 changes the alignment anchor and the MISO3 input."""
    import misonet_amd as mz
    from oracle import pipeline_oracle, mvdr_oracle, miso_oracle
    m1, m3 = nets
    mx, cl = _utt_inputs(3, 48)
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=2)
    out, extra = enh.enhance(torch.from_numpy(mx[None]).cuda(), None, want_bf=True, want_miso1=True)
    est, _ = pipeline_oracle.miso1_inference(mx, sd1, ref_ch=2)
    _assert_parity(extra["miso1"][0].cpu().numpy(), est, "ref_ch=2 miso1 vs oracle")
    mix_bf = np.transpose(mx, (2, 0, 1))[None]
    for s in range(2):
        b = mvdr_oracle.apply_beamforming(np.transpose(est[s], (2, 0, 1))[None], mix_bf)
        _assert_parity(extra["bf"][0, s].cpu().numpy(), b[0], f"ref_ch=2 bf spk{s} vs oracle")
        o = miso_oracle.miso3_forward(torch.from_numpy(mx[None]), torch.from_numpy(b)[:, None],
                                      torch.from_numpy(est[s, 2])[None, None], sd3)
        _assert_parity(out[0, s].cpu().numpy(), o[0, 0].numpy(), f"ref_ch=2 miso3 spk{s} vs oracle")


def test_full_size_properties(nets, sd1):
    """BASELINE geometry T = 1001 (16 kHz, 4 s): one forward against the oracle, and size-independent properties:
    a batch equals its samples run alone (per-sample norms), and the result does not depend on batch position."""
    from oracle import miso_oracle
    m1, _ = nets
    mx, _ = _utt_inputs(1, 1001)
    assert mx.shape == (6, 1001, 129)
    x = torch.from_numpy(mx[None]).cuda()
    y1 = m1(x)
    y_ref = miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1).numpy()
    _assert_parity(y1.cpu().numpy(), y_ref, "miso1 T=1001 vs oracle")
    xb = torch.cat([x, 2 * x, torch.roll(x, 1, dims=1)], dim=0)
    yb = m1(xb)
    # bit for bit: the statistics are accumulated exactly (csrc/det_stats.hpp), nothing depends on the batch
    assert np.array_equal(yb[0].cpu().numpy(), y1[0].cpu().numpy())
    y3 = m1(torch.roll(x, 1, dims=1))
    assert np.array_equal(yb[2].cpu().numpy(), y3[0].cpu().numpy())
    assert np.array_equal(m1(x).cpu().numpy(), y1.cpu().numpy())            # and from run to run


def test_config1_sample_clean_8khz(nets):
    """BASELINE.json configs[0]: first 4 s of the reference's sample/Clean recording (8 kHz, 6 mics, T = 501) through
    one MISO_1 forward, against the golden produced by the real reference (G8)."""
    from oracle import pipeline_oracle
    m1, _ = nets
    g = golden("g8_sample_clean_miso1.npz")
    x = pipeline_oracle.stft_chunk(g["obs_wav_f16"].astype(np.float32), 8000)[None]
    y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
    assert y.shape == (1, 2, 501, 129)
    _assert_parity(y[:, :, 200:232], g["y_slice"], "config-1 sample/Clean slice vs reference golden")
    assert rel_l2(np.abs(y).sum(-1), g["mag_sum_per_frame"]) < 1e-4


def test_inference_loader_two_splits(nets, sd1, sd3, tmp_path):
    """Enhancer.inference as a drop-in for Tester_Enhance.inference (tester.py:846-975): a recording of two 4 s-style
    splits (here 48 frames each) with a zero-padded tail; waves vs the oracle run split by split."""
    import misonet_amd as mz
    from misonet_amd import stft as S
    from oracle import pipeline_oracle
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    chunk = 47 * 64
    from misonet_amd.weights import synthetic_utterance
    obs, s0, s1 = synthetic_utterance(21, 2 * chunk - 500)
    parts_o, gap = S.split_chunks(obs, chunk)
    parts_0, _ = S.split_chunks(s0, chunk)
    parts_1, _ = S.split_chunks(s1, chunk)
    assert gap == 500 and len(parts_o) == 2
    od, d0, d1 = {}, {}, {}
    for k in range(2):
        od[str(k)] = torch.from_numpy(pipeline_oracle.stft_chunk(parts_o[k]))[None]
        d0[str(k)] = torch.from_numpy(pipeline_oracle.stft_chunk(parts_0[k]))[None]
        d1[str(k)] = torch.from_numpy(pipeline_oracle.stft_chunk(parts_1[k]))[None]
    res = enh.inference([(od, d0, d1, [gap], ["rec"])], str(tmp_path), fs=16000)
    wav = res["rec"]
    assert wav.shape == (2, 2 * chunk - 500) and wav.dtype == np.int16
    ref = []
    for k in range(2):
        r = pipeline_oracle.enhance_utterance(od[str(k)][0].numpy(), np.stack([d0[str(k)][0, 0].numpy(), d1[str(k)][0, 0].numpy()]),
                                              sd1, sd3, ref_ch=0)
        ref.append([pipeline_oracle.istft_int16(r["out"][s]) for s in range(2)])
    for s in range(2):
        full = np.concatenate([ref[0][s], ref[1][s][: chunk - gap]])
        d = np.abs(wav[s].astype(np.int32) - full.astype(np.int32))
        print(f"[inference] spk{s}: max |diff| {d.max()} LSB")
        assert d.max() <= 1
        v, fs = S.read_wav_pcm24(str(tmp_path / f"rec_{s}.wav"))
        assert fs == 16000 and np.array_equal(v[:, 0], wav[s].astype(np.int32) << 8)


def test_tester_enhance_class_drop_in(nets, sd1, sd3, tmp_path):
    """misonet_amd.tester.Tester_Enhance: the reference's harness class (tester.py:798-975) with the arguments run.py:272-274
    passes; test() writes cv_dev93/ and test_eval92/ like the reference (tester.py:833,841)."""
    from misonet_amd.tester import Tester_Enhance
    from misonet_amd import stft as S
    from oracle import pipeline_oracle
    m1, m3 = nets
    frames = 40
    mx, cl = _utt_inputs(5, frames)
    s0 = np.broadcast_to(cl[0][None], (6,) + cl[0].shape)        # loaders carry all mics; only ref_ch is read (tester.py:889)
    s1 = np.broadcast_to(cl[1][None], (6,) + cl[1].shape)
    item = ({"0": torch.from_numpy(mx)[None]}, {"0": torch.from_numpy(s0.copy())[None]},
            {"0": torch.from_numpy(s1.copy())[None]}, [0], ["utt5"])
    tst = Tester_Enhance("SMS_WSJ", "MISO3", [item], [item], m1, m3, 6, 0, 2, (frames - 1) * 64 / 16000, str(tmp_path), 0, True,
                         fs=16000, window="hann", length=256, overlap=192)
    res = tst.test()
    ref = pipeline_oracle.enhance_utterance(mx, cl, sd1, sd3, ref_ch=0)
    for sub in ("cv_dev93", "test_eval92"):
        wav = res[sub]["utt5"]
        for s in range(2):
            want = pipeline_oracle.istft_int16(ref["out"][s])
            assert np.abs(wav[s].astype(np.int32) - want.astype(np.int32)).max() <= 1
            v, fs = S.read_wav_pcm24(str(tmp_path / sub / f"utt5_{s}.wav"))
            assert fs == 16000 and np.array_equal(v[:, 0], wav[s].astype(np.int32) << 8)
    with pytest.raises(ValueError):
        Tester_Enhance("SMS_WSJ", "MISO3", [], [], m1, m3, 6, 0, 2, 4.0, str(tmp_path), 0, True,
                       fs=16000, window="hann", length=512, overlap=384)
