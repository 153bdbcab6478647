"""The UNFUSED drop-in flow of INTEGRATION.md sections 2-3, run as a flow: what a maintainer of the reference gets after
the two-import change (``from misonet_amd import MISO_1, MISO_3`` in run.py, ``Apply_Beamforming`` in tester.py) and
nothing else.  The harness below is the reference's own call sequence restated at B = 1 (the only batch size the
reference is correct for):

    tester.py:1014-1068  MISO1_Inference: torch.roll(mix, -k, dims=1) -> model_sep(...) per microphone shift, speakers of
                         every shift aligned to the reference-microphone forward with torch PIT on the device
    tester.py:875-878    .detach().cpu()
    tester.py:889-915    speakers re-ordered against the clean references (torch, CPU)
    tester.py:917-924    permute to [B,F,Ch,T] ndarrays -> Apply_Beamforming -> CPU torch tensor [B,T,F]
    tester.py:928-939    .cuda() -> model(observe, bf, MISO1@ref_ch)   (MISO3_inference, tester.py:1231-1244)

It is compared with the fused on-device ``Enhancer.enhance`` of the same inputs and with the golden of the real reference
(G6).  Runs in the exact-f32 mode and in the headline mode."""
from itertools import permutations

import numpy as np
import pytest
import torch

from conftest import golden, rel_l2
from test_gpu_parity import nets, _assert_parity, _utt_inputs      # noqa: F401

pytestmark = pytest.mark.gpu


def _pit(anchor_mag, cand_mag):
    """first-minimum permutation of sum | |anchor_i| - |cand_p(i)| | (tester.py:1047-1064 / 906-915), one utterance"""
    S = anchor_mag.shape[0]
    dist = torch.stack([torch.stack([(anchor_mag[i] - cand_mag[j]).abs().sum() for j in range(S)]) for i in range(S)])
    perms = list(permutations(range(S)))
    cost = torch.stack([sum(dist[i, p[i]] for i in range(S)) for p in perms])
    return perms[int(torch.argmin(cost))]


def _unfused_reference_flow(model_sep, model, mix, clean, ref_ch, num_spks=2):
    """mix complex [1,M,T,F] on the device, clean complex [1,S,T,F] (CPU).  Returns (miso1 [S,M,T,F], bf [S,T,F], out [S,T,F])."""
    from misonet_amd import Apply_Beamforming
    B, M, T, F = mix.shape
    assert B == 1
    order = np.roll(np.arange(M), -ref_ch)
    est = [torch.empty((B, M, T, F), dtype=torch.complex64) for _ in range(num_spks)]
    with torch.no_grad():
        ref_out = model_sep(torch.roll(mix, -ref_ch, dims=1))                        # [1,S,T,F] device
        for s in range(num_spks):
            est[s][:, ref_ch] = ref_out[:, s].cpu()
        ref_mag = ref_out[0].abs()
        for k in order[1:]:
            out_k = model_sep(torch.roll(mix, -int(k), dims=1))
            perm = _pit(ref_mag, out_k[0].abs())
            for s in range(num_spks):
                est[s][:, int(k)] = out_k[:, perm[s]].cpu()                           # device -> host (tester.py:875-878)
    mix_cpu = mix.detach().cpu()
    # clean-reference alignment at ref_ch (tester.py:889-915): anchors = clean sources
    perm = _pit(clean[0].abs(), torch.stack([est[s][0, ref_ch] for s in range(num_spks)]).abs())
    est = [est[perm[s]] for s in range(num_spks)]
    # MVDR on ndarrays, as the reference calls it (tester.py:917-924): returns a CPU torch tensor [B,T,F]
    observe_bf = mix_cpu.permute(0, 3, 1, 2).numpy()
    bf = []
    for s in range(num_spks):
        source = est[s].permute(0, 3, 1, 2).numpy()
        b = Apply_Beamforming(source, observe_bf)
        assert isinstance(b, torch.Tensor) and b.device.type == "cpu" and tuple(b.shape) == (B, T, F)
        bf.append(b.unsqueeze(1))
    # back to the device, MISO3 per speaker (tester.py:928-939, 1231-1244)
    outs = []
    with torch.no_grad():
        for s in range(num_spks):
            o = model(mix, bf[s].cuda(), est[s][:, ref_ch].unsqueeze(1).cuda())
            outs.append(o.squeeze(1).cpu())
    return (torch.stack([e[0] for e in est]).numpy(), torch.stack([b[0, 0] for b in bf]).numpy(),
            torch.stack([o[0] for o in outs]).numpy())


@pytest.mark.parametrize("ref_ch", [0, 2])
def test_unfused_dropin_flow_equals_fused_pipeline(nets, ref_ch):
    import misonet_amd as mz
    m1, m3 = nets
    mx, cl = _utt_inputs(7, 64)
    mix = torch.from_numpy(mx[None]).cuda()
    clean = torch.from_numpy(cl[None])
    miso1_u, bf_u, out_u = _unfused_reference_flow(m1, m3, mix, clean, ref_ch)
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=ref_ch)
    out_f, ex = enh.enhance(mix, clean.cuda(), want_bf=True, want_miso1=True)
    out_f, bf_f, miso1_f = out_f[0].cpu().numpy(), ex["bf"][0].cpu().numpy(), ex["miso1"][0].cpu().numpy()
    # the same kernels see the same numbers in both flows (the hand-over through complex64 / the host is exact): the
    # separated estimates are identical, what follows differs at most by the MVDR kernels' two input layouts
    assert np.array_equal(miso1_u, miso1_f), "MISO1_Inference (all microphones, aligned)"
    e_bf, e_out = rel_l2(bf_u, bf_f), rel_l2(out_u, out_f)
    print(f"[unfused vs fused] ref_ch={ref_ch} precision={m1.precision}: bf {e_bf:.3e} out {e_out:.3e}")
    assert e_bf < 1e-5 and e_out < 1e-4
    if ref_ch == 0:
        g = golden("g6_pipeline_T64.npz")                      # the real reference's Tester_Enhance on this utterance
        _assert_parity(miso1_u[:, 0], g["miso1_ref"], "unfused flow miso1@ref vs reference golden")
        _assert_parity(bf_u, g["bf"], "unfused flow bf vs reference golden")
        _assert_parity(out_u, g["out"], "unfused flow miso3 vs reference golden")


def test_tester_enhance_rejects_other_enhance_modes(nets, tmp_path):
    """tester.py:935-945 branches on enhance_mode; only the MISO3 branch exists here -- anything else must fail at
    construction instead of silently running MISO3."""
    from misonet_amd.tester import Tester_Enhance
    m1, m3 = nets
    with pytest.raises(ValueError):
        Tester_Enhance("SMS_WSJ", "MISO2", [], [], m1, m3, 6, 0, 2, 4.0, str(tmp_path), 0, True,
                       fs=16000, window="hann", length=256, overlap=192)
