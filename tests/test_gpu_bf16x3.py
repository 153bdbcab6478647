"""GPU parity of the bf16x3 precision mode (three-term bf16 split on the bf16 matrix cores, conv_bf16.hip) against the
same goldens / oracle and the same 1e-3 tolerance as the exact-f32 mode."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2, mag_parity
from test_gpu_parity import _assert_parity, _utt_inputs, _need_gpu

pytestmark = pytest.mark.gpu


# "bf16x3": oct-layout activations, instance norm folded into per-sample weights, LDS-DMA staging (conv_bf16_dma.hip);
# "bf16x3p": planar float32 activations, normalise-on-load staging (conv_bf16.hip)
@pytest.fixture(scope="module", params=["bf16x3", "bf16x3p"])
def nets_bf(request, sd1, sd3):
    _need_gpu()
    PREC = request.param
    import misonet_amd as mz
    from misonet_amd import weights as W
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1)
    m1.eval().set_precision(PREC)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m3.load_state_dict(sd3)
    m3.eval().set_precision(PREC)
    return m1, m3


def test_bf16x3_stage_taps(nets_bf, sd1):
    from oracle import miso_oracle
    m1, _ = nets_bf
    g = golden("g1_miso1_T32.npz")
    x = torch.from_numpy(g["x"])
    taps = {}
    y_ref = miso_oracle.miso1_forward(x, sd1, taps).numpy()
    y = m1(x.cuda()).cpu().numpy()
    for nm in ["enc0_conv"] + [f"enc{b}" for b in range(7)] + ["tcn_out"] + [f"dec{b}" for b in range(7)]:
        ref = taps[nm].numpy()
        if ref.ndim == 3:
            ref = ref[..., None]
        got = m1.tap(nm, 1, 32).cpu().numpy()
        e = rel_l2(got, ref)
        print(f"[tap bf16x3] {nm:10s} rel_l2={e:.3e}")
        assert e < 1e-3, nm
    _assert_parity(y, y_ref, "bf16x3 miso1 T=32 vs oracle")
    _assert_parity(y, g["y"], "bf16x3 miso1 T=32 vs reference golden")


@pytest.mark.parametrize("B,T", [(1, 96), (2, 130), (3, 40), (1, 5)])
def test_bf16x3_forward_shapes(nets_bf, sd1, B, T):
    from oracle import miso_oracle
    m1, _ = nets_bf
    r = np.random.default_rng(77 + T)
    x = (r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)
    y = m1(torch.from_numpy(x).cuda()).cpu().numpy()
    y_ref = np.concatenate([miso_oracle.miso1_forward(torch.from_numpy(x[b:b + 1]), sd1).numpy() for b in range(B)])
    _assert_parity(y, y_ref, f"bf16x3 miso1 B={B} T={T} vs oracle")


def test_bf16x3_pipeline_vs_golden(nets_bf):
    import misonet_amd as mz
    m1, m3 = nets_bf
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    g = golden("g6_pipeline_T64.npz")
    mx, cl = _utt_inputs(7, 64)
    out, extra = enh.enhance(torch.from_numpy(mx[None]).cuda(), torch.from_numpy(cl[None]).cuda(), want_bf=True)
    _assert_parity(extra["bf"][0].cpu().numpy(), g["bf"], "bf16x3 pipeline bf vs golden")
    _assert_parity(out[0].cpu().numpy(), g["out"], "bf16x3 pipeline miso3 vs golden")


def test_bf16x3_full_size(nets_bf, sd1):
    from oracle import miso_oracle
    m1, _ = nets_bf
    mx, _ = _utt_inputs(1, 1001)
    y = m1(torch.from_numpy(mx[None]).cuda()).cpu().numpy()
    y_ref = miso_oracle.miso1_forward(torch.from_numpy(mx[None]), sd1).numpy()
    _assert_parity(y, y_ref, "bf16x3 miso1 T=1001 vs oracle")
