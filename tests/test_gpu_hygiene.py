"""GPU tests of the round-2 boundary work: PIT over S! permutations (S = 3 golden from the real reference), no limit on
the number of frames in the TCN, device handling of the host-side mirror, Enhancer re-commit."""
import numpy as np
import os
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch

from conftest import golden, rel_l2, modes, needs_alt_modes
from test_gpu_parity import _assert_parity, _utt_inputs, _need_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["f32", "f32w", "bf16x6"])
def test_three_speaker_separation_vs_reference_golden(sd3, mode):
    """Enhancer.separate with num_spks = 3 against G10 (Tester_Enhance.MISO1_Inference of the real reference)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    g = golden("g10_miso1_inference_S3_T32.npz")
    sd = W.make_state_dict(W.miso1_spec(num_spks=3), seed=2)
    m1 = mz.MISO_1(3, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m3.load_state_dict(sd3)
    enh = mz.Enhancer(m1.eval().set_precision(mode), m3.eval().set_precision(mode), num_spks=3, ref_ch=0)
    est = enh.separate(torch.from_numpy(g["x"]).cuda(), None)[0].cpu().numpy()          # [S,M,T,F]
    assert est.shape == (3, 6, 32, 129)
    _assert_parity(est[:, :, ::2], g["est_even"], "3-speaker MISO1_Inference vs reference golden")
    assert rel_l2(np.abs(est).sum(-1), g["mag_sum"]) < 1e-4


@pytest.mark.parametrize("S", [1, 3, 4])
def test_pit_select_general_s_vs_oracle(S):
    _need_gpu()
    from misonet_amd.beamform import pit_select
    from oracle import mvdr_oracle
    r = np.random.default_rng(90 + S)
    B, T, F = 7, 29, 129
    a = (r.standard_normal((B, S, T, F)) + 1j * r.standard_normal((B, S, T, F))).astype(np.complex64)
    c = np.empty_like(a)
    for b in range(B):                                     # a random permutation of the anchors + noise per item
        c[b] = a[b, r.permutation(S)]
    c += 0.1 * (r.standard_normal(c.shape) + 1j * r.standard_normal(c.shape)).astype(np.complex64)
    sel, dist = pit_select(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda(), return_dist=True)
    sel_ref, dist_ref = mvdr_oracle.pit_select(a, c)
    assert np.array_equal(sel.cpu().numpy(), sel_ref)
    assert rel_l2(dist.cpu().numpy(), dist_ref) < 1e-5
    # all candidates equal: every permutation ties, the first (identity) wins (argmin convention)
    same = np.repeat(a[:, :1], S, axis=1)
    sel_t = pit_select(torch.from_numpy(a).cuda(), torch.from_numpy(same).cuda())
    assert np.array_equal(sel_t.cpu().numpy(), np.tile(np.arange(S), (B, 1)))


def test_long_utterance_no_frame_limit(sd1):
    """T = 2500 frames (10 s): the depth-wise TCN kernel walks rows longer than its LDS row in segments; the reference
    has no length limit (model.py:553-567)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    from oracle import miso_oracle
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1)
    mx, _ = _utt_inputs(2, 2500)
    m1.keep_activations(True)                          # for the tcn_out tap below
    y = m1.eval()(torch.from_numpy(mx[None]).cuda())
    y_ref = miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1).numpy()
    _assert_parity(y.cpu().numpy(), y_ref, "miso1 T=2500 vs oracle")
    m1.keep_activations(False)
    for mode in modes("f32w", "bf16x6w"):            # the persistent Winograd kernels: 40 column tiles per row tile, the last one ragged
        m1.set_precision(mode)
        _assert_parity(m1(torch.from_numpy(mx[None]).cuda()).cpu().numpy(), y_ref, f"miso1 T=2500 vs oracle [{mode}]")
    m1.set_precision("bf16x6")
    m1.keep_activations(True)
    y = m1(torch.from_numpy(mx[None]).cuda())
    tcn = m1.tap("tcn_out", 1, 2500).cpu().numpy()
    taps = {}
    miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1, taps)
    ref = taps["tcn_out"].numpy()
    ref = ref[..., None] if ref.ndim == 3 else ref
    assert rel_l2(tcn, ref) < 1e-4


def test_device_handling(sd1, sd3):
    """.to('cuda') (index-less) must not reject cuda:0 inputs; an Enhancer follows a later load_state_dict;
    wrong shapes / devices raise Python errors instead of faulting on the GPU."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").to("cuda")
    assert m1._device == torch.device("cuda", torch.cuda.current_device())
    m1.load_state_dict(sd1)
    x = torch.zeros((1, 6, 8, 129), dtype=torch.complex64, device="cuda:0")
    x.real.normal_()
    y0 = m1.eval()(x)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda("cuda")
    m3.load_state_dict(sd3)
    enh = mz.Enhancer(m1, m3.eval(), num_spks=2, ref_ch=0)
    o0 = enh.enhance(x)
    # a checkpoint loaded AFTER the Enhancer was built clears the committed state: enhance must re-commit
    sd1b = {k: (v * 1.5 if k.endswith("conv2d.weight") else v) for k, v in sd1.items()}
    m1.load_state_dict(sd1b)
    o1 = enh.enhance(x)
    assert torch.isfinite(torch.view_as_real(o1)).all() and tuple(o1.shape) == tuple(o0.shape)
    m1.load_state_dict(sd1)
    assert rel_l2(enh.enhance(x).cpu().numpy(), o0.cpu().numpy()) < 1e-5
    with pytest.raises(RuntimeError):
        enh.enhance(x.cpu())                                                       # CPU tensor
    with pytest.raises(ValueError):
        enh.enhance(x[:, :5])                                                      # wrong mic count
    with pytest.raises(ValueError):
        enh.enhance(x, torch.zeros((1, 3, 8, 129), dtype=torch.complex64, device="cuda"))   # clean shape
    with pytest.raises(ValueError):
        enh.enhance(x, out=torch.zeros((1, 2, 8, 128), dtype=torch.complex64, device="cuda"))
    with pytest.raises(ValueError):
        enh.separate(x[:, :4])
    with pytest.raises(RuntimeError):
        enh.separate(x.cpu())
    assert y0.shape == (1, 2, 8, 129)


@pytest.mark.parametrize("mode", modes("f32", "f32w", "bf16x6", "bf16x6w", "f16x3", "bf16x3"))
def test_one_chunk_layers(mode):
    """8-channel dense-block growth: layers of ONE and TWO 8-channel K-chunks (the bf16x6 producers fold the set-up of the
    coming tile into fewer iterations there) and output groups of 8 channels."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    from oracle import miso_oracle
    en, de = (8, 16, 40, 32, 48, 64, 128), (128, 64, 48, 32, 40, 16, 8)
    sd = W.make_state_dict(W.tensor_spec(12, 4, en, de), seed=9)
    m = mz.MISO_1(2, 6, 7, list(en), list(de), "IN").cuda(0)
    m.load_state_dict(sd)
    m.eval().set_precision(mode)
    r = np.random.default_rng(43)
    x = (r.standard_normal((2, 6, 150, 129)) + 1j * r.standard_normal((2, 6, 150, 129))).astype(np.complex64)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd).numpy() for b in range(2)])
    _assert_parity(y, ref, f"[{mode}] 8-channel growth (en={en})")


@pytest.mark.parametrize("mode", modes("f32", "f32w", "bf16x6", "bf16x6w", "f16x3", "bf16x3"))
def test_non_default_geometry(mode):
    """A geometry other than config/NN_BSS.yml's: 4 microphones, 3 speakers, bottleneck channels
    (16,24,40,32,48,64,128) -- output groups of 16/24/40/48 channels, 2-chunk layers, a dense block that grows to 200
    channels -- against the oracle (which derives every shape from the state_dict).  The constructor arguments are the
    reference's (model.py:9); en/de lists must mirror each other (model.py:35,99)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    from oracle import miso_oracle
    en, de = (16, 24, 40, 32, 48, 64, 128), (128, 64, 48, 32, 40, 24, 16)
    sd = W.make_state_dict(W.tensor_spec(8, 6, en, de), seed=5)
    m = mz.MISO_1(3, 4, 7, list(en), list(de), "IN").cuda(0)
    m.load_state_dict(sd)
    m.eval().set_precision(mode)
    r = np.random.default_rng(41)
    x = (r.standard_normal((2, 4, 70, 129)) + 1j * r.standard_normal((2, 4, 70, 129))).astype(np.complex64)
    y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd).numpy() for b in range(2)])
    assert y.shape == (2, 3, 70, 129)
    _assert_parity(y, ref, f"[{mode}] non-default geometry (4 mics, 3 speakers, en={en})")


@needs_alt_modes
def test_f16x3_overflow_fails_loudly(sd1):
    """f16x3 keeps activations as fp16 pieces: values beyond 65504 overflow.  That must surface as the library's NaN
    error (FloatingPointError), never as a silently wrong spectrogram; the float32-range modes take the same input."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    m.eval()
    r = np.random.default_rng(8)
    x = (1e6 * (r.standard_normal((1, 6, 40, 129)) + 1j * r.standard_normal((1, 6, 40, 129)))).astype(np.complex64)
    xd = torch.from_numpy(x).cuda()
    y6 = m.set_precision("bf16x6")(xd).cpu().numpy()
    y32 = m.set_precision("f32")(xd).cpu().numpy()
    assert np.isfinite(y6).all() and rel_l2(y6, y32) < 1e-4
    with pytest.raises(FloatingPointError):
        m.set_precision("f16x3")(xd)


def test_hip_graph_capture_replays_identical_bits(sd1, sd3):
    """The whole MISO1 x6 -> PIT -> MVDR -> MISO3 x2 pass (about 450 launches) captured into a HIP graph: a replay returns
    the bits of the eager pass, also after new inputs were copied into the captured tensors."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m3.load_state_dict(sd3)
    enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=2, ref_ch=0)
    a, b = _utt_inputs(7, 64), _utt_inputs(11, 64)
    mix = torch.from_numpy(a[0][None]).cuda()
    clean = torch.from_numpy(a[1][None]).cuda()
    eager_a = enh.enhance(mix, clean).clone()
    eager_b = enh.enhance(torch.from_numpy(b[0][None]).cuda(), torch.from_numpy(b[1][None]).cuda()).clone()
    cap = enh.capture_graph(mix, clean)
    g, out = cap
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager_a)
    mix.copy_(torch.from_numpy(b[0][None]))
    clean.copy_(torch.from_numpy(b[1][None]))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager_b)
    enh.check(1, 64, captured=cap)          # the graph owns its workspace (tests/test_gpu_wavpath.py: lifetime)
    cap.check()                             # ... the same through the captured pass itself
    # `g.replay(); enh.check(B, T)` used to allocate a fresh eager workspace and report "clean" whatever the replay
    # did; while a captured pass of that shape is alive and no eager pass has run, the ambiguous call refuses to answer
    with pytest.raises(RuntimeError, match="captured"):
        enh.check(1, 64)
    # a NaN fed through the graph is seen by the captured pass's check, not lost
    bad = torch.from_numpy(b[0][None]).clone()
    bad[0, 0, 3, 5] = complex(float("nan"), 0.0)
    mix.copy_(bad)
    g.replay()
    with pytest.raises(FloatingPointError):
        cap.check()
    # after an eager pass of that shape its own workspace exists again and check(B, T) means it
    mix.copy_(torch.from_numpy(b[0][None]))
    enh.enhance(mix, clean)
    enh.check(1, 64)


def test_wav_entry_guards_and_cold_graph_capture(sd1, sd3):
    """enhance_wav has the guards of enhance (one frame raises the reference's ValueError, a separation-only
    Enhancer raises instead of failing inside the library); and -- VERDICT r4 item 3 -- a HIP graph can capture
    enhance_wav_int16 as the FIRST call of a fresh process: the STFT / iSTFT tables are built by misonet_net_commit /
    misonet_pipeline_create, never inside an asynchronous call."""
    _need_gpu()
    import subprocess
    import sys
    import misonet_amd as mz
    from misonet_amd import weights as W
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m3.load_state_dict(sd3)
    enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=2, ref_ch=0)
    with pytest.raises(ValueError, match="more than 1 spatial element"):
        enh.enhance_wav(torch.zeros((1, 40, 6), device="cuda"))
    sep = mz.Enhancer(m1, None, num_spks=2, ref_ch=0)
    with pytest.raises(RuntimeError, match="without MISO_3"):
        sep.enhance_wav(torch.zeros((1, 64 * 20, 6), device="cuda"))
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
import misonet_amd as mz
from misonet_amd import weights as W
m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m1.load_state_dict(W.make_state_dict(W.miso1_spec(), 0))
m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0); m3.load_state_dict(W.make_state_dict(W.miso3_spec(), 1))
enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=2, ref_ch=0)
obs, s0, s1 = W.synthetic_utterance(5, 64 * 47)
wav = torch.from_numpy(obs[None]).cuda()
ws = enh.workspace(1, 48)                       # allocation only: no kernel of the wav path has run in this process
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):         # the FIRST wav-path call of the process, inside a capture
        pcm = enh.enhance_wav_int16(wav, None, check_nan=False)
g.replay(); torch.cuda.synchronize()
a = pcm.cpu().numpy().copy()
b = enh.enhance_wav_int16(wav, None).cpu().numpy()
assert np.array_equal(a, b) and np.abs(a).max() > 0, (np.abs(a).max(), np.abs(a.astype(int) - b).max())
print("COLD_CAPTURE_OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COLD_CAPTURE_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@needs_alt_modes
@pytest.mark.parametrize("scale", [1e-4, 1e-2, 1e3])
def test_f16x3_input_scale_range(sd1, scale):
    """f16x3 keeps the RAW first-layer output (whose scale follows the input) in the exact three-bf16 layout and runs the
    dense block that reads it in the bf16x6 arithmetic, so low-level and loud inputs are as accurate as in the exact modes;
    the ground truth is the float64 oracle (at small scales every float32 implementation, the reference included, loses
    accuracy to the cancellation against the first conv's bias).  Beyond ~1e4 x unit scale the un-normalised output of
    that block leaves fp16's range and the forward fails loudly (test_f16x3_overflow_fails_loudly)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    from oracle import miso_oracle
    from conftest import mag_parity
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    m.eval()
    r = np.random.default_rng(8)
    x = (scale * (r.standard_normal((1, 6, 64, 129)) + 1j * r.standard_normal((1, 6, 64, 129)))).astype(np.complex64)
    with miso_oracle.precision(torch.float64):
        truth = miso_oracle.miso1_forward(torch.from_numpy(x).to(torch.complex128), sd1).numpy()
    err = {}
    for mode in ("f32", "bf16x6", "f16x3"):
        err[mode] = mag_parity(m.set_precision(mode)(torch.from_numpy(x).cuda()).cpu().numpy(), truth)[0]
    print(f"[scale {scale:g}] " + "  ".join(f"{k} {v:.2e}" for k, v in err.items()))
    assert err["f16x3"] <= 2.0 * err["f32"] + 2e-6, err
    assert err["bf16x6"] <= 2.0 * err["f32"] + 2e-6, err


def test_tap_refuses_a_workspace_written_with_another_plan(sd1):
    """forward with the shared arena, then misonet_net_keep_activations(net, 1), then a tap on the SAME workspace.
    The host-side keep flag is now true, but the buffers in that workspace were laid out by the shared plan (0.26 instead of
    0.55 GB per sample): the tap must fail with MISONET_ESTATE instead of reading past the workspace.  The forward stamps
    its layout into the workspace header; the tap compares."""
    _need_gpu()
    import ctypes as C
    import misonet_amd as mz
    from misonet_amd import _lib, weights as W
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    mx, _ = _utt_inputs(4, 40)
    y = m.eval()(torch.from_numpy(mx[None]).cuda())
    ws = m._ws[(1, 40)]                                   # the workspace of that forward (shared-arena plan)
    L = _lib.lib()
    dst = torch.empty((1, 24, 40, 127), dtype=torch.float32, device="cuda")
    # the output tap is always readable ...
    c, f = C.c_int(), C.c_int()
    _lib.check(L.misonet_net_tap_shape(m._net, b"enc0", C.byref(c), C.byref(f)))
    assert (c.value, f.value) == (24, 127)
    # ... an inner one is refused while buffers share memory
    rc = L.misonet_net_tap(m._net, b"enc0", ws.data_ptr(), 1, 40, dst.data_ptr(), _lib.stream_ptr(ws.device))
    assert rc == _lib.ESTATE
    # switch the plan WITHOUT a new forward: still refused (this sequence used to read out of bounds)
    _lib.check(L.misonet_net_keep_activations(m._net, 1))
    rc = L.misonet_net_tap(m._net, b"enc0", ws.data_ptr(), 1, 40, dst.data_ptr(), _lib.stream_ptr(ws.device))
    assert rc == _lib.ESTATE and b"another buffer plan" in L.misonet_last_error()
    _lib.check(L.misonet_net_keep_activations(m._net, 0))
    # the supported sequence: keep first, forward, tap -- and a different T on that workspace is refused again
    m.keep_activations(True)
    y2 = m(torch.from_numpy(mx[None]).cuda())
    assert torch.equal(y, y2)
    t = m.tap("enc0", 1, 40)
    assert tuple(t.shape) == (1, 24, 40, 127) and torch.isfinite(t).all()
    ws2 = m._ws[(1, 40)]
    rc = L.misonet_net_tap(m._net, b"enc0", ws2.data_ptr(), 1, 39, dst.data_ptr(), _lib.stream_ptr(ws2.device))
    assert rc == _lib.ESTATE


def test_pit_select_checks_its_scratch_size():
    """the scratch of misonet_pit_select grew from B*S*S to B*S*S*(F+1) doubles between ABI 200 and 300 with
    no size in the signature; ABI 400 takes dist_bytes and answers MISONET_ENOMEM."""
    _need_gpu()
    from misonet_amd import _lib
    L = _lib.lib()
    assert L.misonet_version() >= 400
    B, S, T, F = 2, 2, 16, 129
    need = L.misonet_pit_scratch_bytes(B, S, F)
    assert need == B * S * S * (F + 1) * 8
    a = torch.randn((B, S, T, F), dtype=torch.complex64, device="cuda")
    sel = torch.empty((B, S), dtype=torch.int32, device="cuda")
    dist = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    st = _lib.stream_ptr(a.device)
    assert L.misonet_pit_select(a.data_ptr(), a.data_ptr(), B, S, T, F, sel.data_ptr(), dist.data_ptr(), B * S * S * 8, st) == _lib.ENOMEM
    assert L.misonet_pit_select(a.data_ptr(), a.data_ptr(), B, S, T, F, sel.data_ptr(), dist.data_ptr(), need - 1, st) == _lib.ENOMEM
    _lib.check(L.misonet_pit_select(a.data_ptr(), a.data_ptr(), B, S, T, F, sel.data_ptr(), dist.data_ptr(), need, st))
    assert sel.cpu().tolist() == [[0, 1], [0, 1]]


def test_default_precision_is_bf16x6_and_golden_green(sd1, monkeypatch):
    """the arithmetic of a freshly constructed network changed from f32 to bf16x6 in ABI 300.  This pins the
    UNTOUCHED default (no set_precision call, no MISONET_PRECISION) to the reference golden G1 and states which mode it is."""
    _need_gpu()
    monkeypatch.delenv("MISONET_PRECISION", raising=False)
    import misonet_amd as mz
    from misonet_amd import _lib, weights as W
    from conftest import golden
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    assert m.precision == "bf16x6" and _lib.lib().misonet_net_get_precision(m._net) == 3
    g = golden("g1_miso1_T32.npz")
    y = m.eval()(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    _assert_parity(y, g["y"], "default-constructed MISO_1 vs G1")


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
@pytest.mark.parametrize("nt", ["gLN", "cLN", "BN"])
def test_norm_type_variants_vs_reference_golden(nt, mode):
    """norm_type = gLN / cLN / BatchNorm1d for the outer norms of the TemporalBlocks (reference model.py:530,535,570-581;
    round-3 review: "rejected with a message"): golden G11 from the REAL reference built with that argument, both headline
    arithmetic modes, plus bit-exact batch invariance (the gLN sum over a sample's partials is a fixed-order tree)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    g = golden(f"g11_norm_{nt}_T40.npz")
    sd = W.make_state_dict(W.miso1_spec(norm_type=nt), seed=3)
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), nt).cuda(0)
    m.load_state_dict(sd)
    m.eval().set_precision(mode)
    x = torch.from_numpy(g["x"]).cuda()
    y = m(x)
    _assert_parity(y.cpu().numpy(), g["y"], f"MISO_1(norm_type={nt}) [{mode}] vs G11")
    y0 = m(x[:1])
    y1 = m(x[1:])
    assert torch.equal(y0[0], y[0]) and torch.equal(y1[0], y[1])
    assert torch.equal(m(x), y)
    if nt == "cLN":
        sd3 = W.make_state_dict(W.miso3_spec(norm_type=nt), seed=4)
        m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), nt).cuda(0)
        m3.load_state_dict(sd3)
        m3.eval().set_precision(mode)
        y3 = m3(x, torch.from_numpy(g["a"]).cuda(), torch.from_numpy(g["b"]).cuda())
        _assert_parity(y3.cpu().numpy(), g["y3"], f"MISO_3(norm_type=cLN) [{mode}] vs G11")


def test_norm_type_pipeline_mixed(sd3):
    """The pipeline with a gLN MISO_1 and the default (IN) MISO_3: the two networks share one workspace whatever their norm
    types (the cLN frame statistics are part of a network's own layout)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    from oracle import pipeline_oracle, miso_oracle
    sd1 = W.make_state_dict(W.miso1_spec(norm_type="cLN"), seed=3)
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "cLN").cuda(0)
    m1.load_state_dict(sd1)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m3.load_state_dict(sd3)
    enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=2, ref_ch=0)
    mx, cl = _utt_inputs(5, 48)
    out, ex = enh.enhance(torch.from_numpy(mx[None]).cuda(), torch.from_numpy(cl[None]).cuda(), want_miso1=True)
    # the separation stage against the oracle run with the same norm type (mic 0 = the un-shifted forward)
    y_ref = miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1, norm_type="cLN").numpy()[0]
    got = ex["miso1"][0, :, 0].cpu().numpy()
    e = min(rel_l2(got, y_ref), rel_l2(got[::-1], y_ref))          # (the clean alignment may swap the speakers)
    assert e < 1e-3, e
    assert torch.isfinite(torch.view_as_real(out)).all()


@needs_alt_modes
def test_bf16x6w_mode_goldens_ragged_shapes_batch_invariance(sd1, sd3):
    """bf16x6w (conv_wino6.hip: the DenseBlock convs in Winograd F(2x2,3x3) form, bf16x6 arithmetic): reference goldens G1 / G3,
    ragged batched shapes against the oracle (row tiles of 7 / 15 / 31 / 63 / 127 rows, column tiles that end inside a tile, T
    below one tile), bit-exact batch invariance, and the float32-faithful accuracy class (within 3 x of the exact-f32 mode's
    distance to the float64 oracle)."""
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    from oracle import miso_oracle
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    m.eval().set_precision("bf16x6w")
    for name in ("g1_miso1_T32.npz", "g1_miso1_T96.npz"):
        g = golden(name)
        _assert_parity(m(torch.from_numpy(g["x"]).cuda()).cpu().numpy(), g["y"], f"[bf16x6w] {name}")
    for B, T in [(3, 40), (2, 130), (1, 5), (2, 257)]:
        r = np.random.default_rng(B * 1000 + T)
        x = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
        x[1:] *= 3.0
        y = m(torch.from_numpy(x).cuda()).cpu().numpy()
        ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd1).numpy() for b in range(B)])
        _assert_parity(y, ref, f"[bf16x6w] B={B} T={T}")
        y1 = np.concatenate([m(torch.from_numpy(x[b:b + 1]).cuda()).cpu().numpy() for b in range(B)])
        assert np.array_equal(y, y1), f"bf16x6w: batch of {B} differs from its utterances one by one (T={T})"
    g96 = golden("g1_miso1_T96.npz")
    with miso_oracle.precision(torch.float64):
        y64 = miso_oracle.miso1_forward(torch.from_numpy(g96["x"]), sd1).numpy()
    err = {}
    for mode in ("f32", "bf16x6w"):
        m.set_precision(mode)
        err[mode] = rel_l2(np.abs(m(torch.from_numpy(g96["x"]).cuda()).cpu().numpy()), np.abs(y64))
    assert err["bf16x6w"] <= 3.0 * err["f32"] + 2e-6, err
