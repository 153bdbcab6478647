"""The bf16x6 mode is the bench headline because it is FP32-FAITHFUL (reference arithmetic: float32, model.py:77-80,
401-482): both operands of every product are represented exactly (three bf16 pieces = 24 bits) and the six leading
partial products are accumulated in float32.  These tests hold it to the error level of the exact-f32 MFMA mode itself:
for every case the error against the oracle is measured in BOTH modes, bf16x6 must stay within 5e-6 (single forwards;
or within 1.25x the f32 mode's error where that itself exceeds 5e-6)
and within 3x the f32 mode's own error + 3e-6 everywhere (pipeline outputs, where f32 itself sits at ~1e-5)."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2, mag_parity, modes, needs_alt_modes
from test_gpu_parity import _utt_inputs, _need_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair(sd1, sd3):
    _need_gpu()
    import misonet_amd as mz
    from misonet_amd import weights as W
    nets = {}
    for mode in modes("f32", "bf16x6", "f32w", "f16x3"):
        m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
        m1.load_state_dict(sd1)
        m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
        m3.load_state_dict(sd3)
        nets[mode] = (m1.eval().set_precision(mode), m3.eval().set_precision(mode))
    return nets


# the two fast modes that are held to the exact-f32 mode's own error level: "bf16x6" (exact three-piece operands, six terms)
# and "f32w" (literal float32 in Winograd form; the bench headline candidate of round 6) -- every test below runs once per mode
FAITHFUL = ("bf16x6", "f32w")


def _cmp(what, got6, got32, ref, single_forward, mode="bf16x6"):
    e6, e32 = mag_parity(got6, ref)[0], mag_parity(got32, ref)[0]
    d = rel_l2(got6, got32)
    print(f"[{mode}] {what}: vs oracle  {mode} {e6:.3e}  f32 {e32:.3e}   {mode} vs f32 {d:.3e}")
    assert np.isfinite(e6)
    # (a 5-frame utterance has 5-sample statistics in the bottleneck: the f32 mode itself sits at ~1e-5 there, so only the
    # looser bound below is asserted for it)
    if single_forward and got6.shape[2] >= 16:
        assert e6 <= max(5e-6, 1.25 * e32), f"{what}: {mode} {e6:.3e} > 5e-6 and > 1.25 x f32's {e32:.3e}"
    assert e6 <= 3.0 * e32 + 3e-6, f"{what}: {mode} {e6:.3e} vs f32 {e32:.3e}"
    return e6, e32


@pytest.mark.parametrize("fmode", FAITHFUL)
@pytest.mark.parametrize("T", [32, 96])
def test_bf16x6_forward_vs_reference_golden(pair, T, fmode):
    g = golden(f"g1_miso1_T{T}.npz")
    x = torch.from_numpy(g["x"]).cuda()
    _cmp(f"miso1 T={T} vs reference golden", pair[fmode][0](x).cpu().numpy(), pair["f32"][0](x).cpu().numpy(), g["y"], True, fmode)


@pytest.mark.parametrize("fmode", FAITHFUL)
def test_bf16x6_miso3_vs_reference_golden(pair, fmode):
    g = golden("g3_miso3_T32.npz")
    args = [torch.from_numpy(g[k]).cuda() for k in ("x", "a", "b")]
    _cmp("miso3 T=32 vs reference golden", pair[fmode][1](*args).cpu().numpy(), pair["f32"][1](*args).cpu().numpy(),
         g["y"], True, fmode)


@pytest.mark.parametrize("fmode", FAITHFUL)
def test_full_size_forward_vs_reference_golden_g12(pair, fmode):
    """the T = 1001 forward of the REAL reference (G12: every 16th frame of MISO_1.forward on the bench's utterance 0, model.py:76-111)
    -- the per-forward bound of a headline mode at the bench geometry"""
    g = golden("g12_fullsize_T1001.npz")
    st = int(g["frame_step"])
    mx, _ = _utt_inputs(0, 1001)
    xd = torch.from_numpy(mx[None]).cuda()
    yf = pair[fmode][0](xd).cpu().numpy()[0][:, ::st]
    y32 = pair["f32"][0](xd).cpu().numpy()[0][:, ::st]
    _cmp("miso1 T=1001 vs reference golden G12", yf[None], y32[None], g["miso1_fwd_frames"][None], True, fmode)


@pytest.mark.parametrize("fmode", FAITHFUL)
def test_bf16x6_stage_taps(pair, sd1, fmode):
    """every stage of a forward: the error does not grow faster through the layers than in the f32 mode"""
    from oracle import miso_oracle
    g = golden("g1_miso1_T32.npz")
    x = torch.from_numpy(g["x"])
    taps = {}
    miso_oracle.miso1_forward(x, sd1, taps)
    got = {}
    for mode in ("f32", fmode):
        m1 = pair[mode][0]
        m1.keep_activations(True)                      # taps need un-shared activation buffers
        try:
            m1(x.cuda())
            got[mode] = {nm: m1.tap(nm, 1, 32).cpu().numpy() for nm in
                         ["enc0_conv"] + [f"enc{b}" for b in range(7)] + ["tcn_out"] + [f"dec{b}" for b in range(7)]}
        finally:
            m1.keep_activations(False)
    for nm in got["f32"]:
        ref = taps[nm].numpy()
        ref = ref[..., None] if ref.ndim == 3 else ref
        e6, e32 = rel_l2(got[fmode][nm], ref), rel_l2(got["f32"][nm], ref)
        print(f"[tap {fmode}] {nm:10s} {fmode} {e6:.3e}  f32 {e32:.3e}")
        assert e6 <= 2.0 * e32 + 2e-6, nm


@pytest.mark.parametrize("fmode", FAITHFUL)
@pytest.mark.parametrize("B,T", [(1, 5), (3, 40), (1, 128), (2, 130), (2, 257)])
def test_bf16x6_ragged_shapes(pair, sd1, B, T, fmode):
    from oracle import miso_oracle
    r = np.random.default_rng(606 + T)
    x = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
    x[1:] *= 3.0
    ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd1).numpy() for b in range(B)])
    xd = torch.from_numpy(x).cuda()
    _cmp(f"miso1 B={B} T={T} vs oracle", pair[fmode][0](xd).cpu().numpy(), pair["f32"][0](xd).cpu().numpy(), ref, True, fmode)


@pytest.mark.parametrize("fmode", FAITHFUL)
def test_bf16x6_full_size_forward(pair, sd1, fmode):
    from oracle import miso_oracle
    mx, _ = _utt_inputs(1, 1001)
    ref = miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1).numpy()
    xd = torch.from_numpy(mx[None]).cuda()
    _cmp("miso1 T=1001 vs oracle", pair[fmode][0](xd).cpu().numpy(), pair["f32"][0](xd).cpu().numpy(), ref, True, fmode)


@pytest.mark.parametrize("fmode", FAITHFUL)
def test_bf16x6_pipeline_vs_reference_golden(pair, sd1, sd3, fmode):
    """MISO1 x6 -> PIT -> MVDR x2 -> MISO3 x2 against the golden of the real reference (G6) and the int16 wave (G7)"""
    import misonet_amd as mz
    g = golden("g6_pipeline_T64.npz")
    mx, cl = _utt_inputs(7, 64)
    res = {}
    for mode in ("f32", fmode):
        enh = mz.Enhancer(pair[mode][0], pair[mode][1], num_spks=2, ref_ch=0)
        out, ex = enh.enhance(torch.from_numpy(mx[None]).cuda(), torch.from_numpy(cl[None]).cuda(), want_bf=True)
        res[mode] = (out[0].cpu().numpy(), ex["bf"][0].cpu().numpy(), enh.to_wav_int16([out[0]], gap=0))
    _cmp("pipeline bf vs golden", res[fmode][1], res["f32"][1], g["bf"], False, fmode)
    _cmp("pipeline miso3 vs golden", res[fmode][0], res["f32"][0], g["out"], False, fmode)
    for s in range(2):
        d = np.abs(res[fmode][2][s].astype(np.int32) - g[f"wav{s}"].astype(np.int32))
        assert d.max() <= 1, d.max()


@pytest.mark.parametrize("fmode", FAITHFUL)
def test_bf16x6_batch_invariance(pair, fmode):
    """a sample's result does not depend on the batch it runs in nor on its position (T = 1001, 9 samples)"""
    m1 = pair[fmode][0]
    mx, _ = _utt_inputs(1, 1001)
    x = torch.from_numpy(mx[None]).cuda()
    y1 = m1(x).cpu().numpy()
    xb = torch.cat([x * (1.0 + 0.25 * i) for i in range(8)] + [x], dim=0)
    yb = m1(xb).cpu().numpy()
    assert np.array_equal(yb[8], y1[0])        # bit for bit (exact statistics accumulation, csrc/det_stats.hpp)
    assert np.array_equal(yb[0], y1[0])


@needs_alt_modes
def test_f16x3_is_at_the_f32_error_level(pair, sd1):
    """"f16x3" rounds the operands to two fp16 pieces (22 bits, the "3xTF32" scheme) -- so it is NOT labelled fp32-faithful
    -- but its error against the oracle sits at the f32 mode's own level: operand rounding at 2^-22 is below the noise of
    float32 accumulation over K = 216 ... 1728 terms.  Measured here for every stage of a forward, ragged shapes and the
    full-size forward, next to f32 and bf16x6."""
    from oracle import miso_oracle
    m32, m6, mh = pair["f32"][0], pair["bf16x6"][0], pair["f16x3"][0]
    cases = [("T=32 golden", torch.from_numpy(golden("g1_miso1_T32.npz")["x"]), golden("g1_miso1_T32.npz")["y"])]
    r = np.random.default_rng(1603)
    for B, T in ((2, 130), (1, 257)):
        x = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
        ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd1).numpy() for b in range(B)])
        cases.append((f"B={B} T={T}", torch.from_numpy(x), ref))
    mx, _ = _utt_inputs(1, 1001)
    cases.append(("T=1001", torch.from_numpy(mx[None]), miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1).numpy()))
    for what, x, ref in cases:
        xd = x.cuda()
        e32 = mag_parity(m32(xd).cpu().numpy(), ref)[0]
        e6 = mag_parity(m6(xd).cpu().numpy(), ref)[0]
        eh = mag_parity(mh(xd).cpu().numpy(), ref)[0]
        print(f"[f16x3] {what}: vs oracle  f32 {e32:.3e}  bf16x6 {e6:.3e}  f16x3 {eh:.3e}")
        assert np.isfinite(eh) and eh <= 2.0 * e32 + 2e-6, what
