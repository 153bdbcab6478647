"""GPU tests of the wav-in -> int16-wav-out path (reference tester.py:865-867 H2D, 949-974 iSTFT -> int16 -> file): the
device-resident entry, the overlapped host stream, and the lifetime of a captured HIP graph."""
import numpy as np
import pytest
import torch

from test_gpu_parity import nets, _need_gpu, _utt_inputs      # noqa: F401

pytestmark = pytest.mark.gpu


def _wav_batches(n, us):
    from misonet_amd.weights import synthetic_utterance
    utts = [synthetic_utterance(u, n) for u in us]
    wav = torch.from_numpy(np.stack([u[0] for u in utts]))                                   # [B, L, 6]
    cwav = torch.from_numpy(np.stack([np.stack([u[1][:, 0], u[2][:, 0]], axis=1) for u in utts]))   # [B, L, 2]
    return wav, cwav


def test_enhance_wav_int16_is_the_stitched_per_speaker_path(nets):
    """One batched iSTFT + cast == the per-(chunk, speaker) iSTFT of to_wav_int16, bit for bit, and within 1 LSB of the
    oracle's SciPy iSTFT of the same spectrograms (tester.py:949-952)."""
    import misonet_amd as mz
    from oracle import pipeline_oracle
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    wav, cwav = _wav_batches(63 * 64, (3, 4))
    pcm = enh.enhance_wav_int16(wav.cuda(), cwav.cuda())
    assert pcm.dtype == torch.int16 and tuple(pcm.shape) == (2, 2, 63 * 64) and pcm.is_cuda
    spec = enh.enhance_wav(wav.cuda(), cwav.cuda())
    for b in range(2):
        one = enh.to_wav_int16([spec[b]], gap=0)
        assert np.array_equal(one, pcm[b].cpu().numpy())
        for s in range(2):
            want = pipeline_oracle.istft_int16(spec[b, s].cpu().numpy())
            assert np.abs(pcm[b, s].cpu().numpy().astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_stream_wav_equals_batch_by_batch(nets, depth):
    """Host-resident batches through the overlapped three-stream pipeline return, in order, exactly the bits of the
    device-resident call on each batch -- pinned and un-pinned inputs, with and without clean references, and a batch of a
    different size in the middle (slots are re-allocated)."""
    import misonet_amd as mz
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    n = 40 * 64
    items = []
    for k, us in enumerate([(1, 2), (3, 4), (5,), (6, 7), (8, 9)]):
        wav, cwav = _wav_batches(n, us)
        if k % 2:
            wav, cwav = wav.pin_memory(), cwav.pin_memory()
        items.append((wav, cwav))
    want = [enh.enhance_wav_int16(w.cuda(), c.cuda()).cpu().numpy() for w, c in items]
    got = list(enh.stream_wav(iter(items), depth=depth))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.dtype == np.int16 and np.array_equal(g, w)
    # without clean references (no clean alignment): plain tensors instead of tuples
    want2 = [enh.enhance_wav_int16(w.cuda(), None).cpu().numpy() for w, _ in items[:3]]
    got2 = list(enh.stream_wav((w for w, _ in items[:3]), depth=depth))
    for g, w in zip(got2, want2):
        assert np.array_equal(g, w)


def test_stream_wav_fresh_slot_buffers_wait_for_queued_work(nets):
    """Regression (round 4): a slot's device buffer is allocated from the current stream's pool while earlier batches are
    still queued -- its block can be the previous batch's spectrogram, which that batch's pipeline has yet to write.  The
    copy-in stream must wait for the queued work before its first copy into a fresh buffer; without that the SECOND batch ran
    on bytes of the first one's spectrogram (hidden for three rounds by torch.istft's host synchronisation, exposed when the
    iSTFT became a HIP kernel).  Short utterances, three different batches, repeated generators."""
    import misonet_amd as mz
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    n = 20 * 64
    items = [_wav_batches(n, (u,)) for u in (1, 2, 1)]
    want = [enh.enhance_wav_int16(w.cuda(), c.cuda()).cpu().numpy() for w, c in items]
    for _ in range(5):
        got = list(enh.stream_wav(iter(items), depth=2, check_nan=False))
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), f"batch {k} of the stream differs from the synchronous result"


def test_stream_wav_reports_nan_of_the_right_batch(nets):
    import misonet_amd as mz
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    n = 20 * 64
    good, cg = _wav_batches(n, (1,))
    bad = good.clone()
    bad[0, 100, 2] = float("nan")
    it = enh.stream_wav(iter([(good, cg), (bad, cg), (good, cg)]), depth=2)
    first = next(it)
    assert first.shape == (1, 2, n)
    with pytest.raises(FloatingPointError):
        next(it)


def test_captured_graph_survives_other_workspaces(nets):
    """the graph records the workspace's address.  After capture the Enhancer is used with another
    batch size, another length and another arithmetic mode (each of which used to FREE the captured workspace), then the
    graph is replayed: it must still return the eager result's bits, and an eager pass of the captured shape between two
    replays must not disturb it."""
    import misonet_amd as mz
    m1, m3 = nets
    enh = mz.Enhancer(m1, m3, num_spks=2, ref_ch=0)
    a, b = _utt_inputs(7, 48), _utt_inputs(11, 48)
    mix = torch.from_numpy(a[0][None]).cuda()
    clean = torch.from_numpy(a[1][None]).cuda()
    eager_a = enh.enhance(mix, clean).clone()
    cap = enh.capture_graph(mix, clean)
    g, out = cap                                           # unpacks like the old (graph, out) pair
    assert out is cap.out
    # churn: other shapes and another mode allocate (and drop) other workspaces
    c = _utt_inputs(3, 80)
    enh.enhance(torch.from_numpy(np.stack([c[0], c[0]])).cuda(), torch.from_numpy(np.stack([c[1], c[1]])).cuda())
    prec = m1.precision
    other = "f32" if prec != "f32" else "bf16x6"
    m1.set_precision(other); m3.set_precision(other)
    enh.enhance(torch.from_numpy(b[0][None]).cuda(), torch.from_numpy(b[1][None]).cuda())
    m1.set_precision(prec); m3.set_precision(prec)
    junk = [torch.full((1 << 22,), 7.0, device="cuda") for _ in range(8)]     # re-use whatever the allocator got back
    eager_b = enh.enhance(torch.from_numpy(b[0][None]).cuda(), torch.from_numpy(b[1][None]).cuda()).clone()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager_a)
    cap.load(torch.from_numpy(b[0][None]).cuda(), torch.from_numpy(b[1][None]).cuda())
    assert torch.equal(cap.replay(), eager_b)
    enh.check(1, 48, captured=cap)
    del junk
    # the graph records addresses: inputs that would be converted (copied) are refused
    with pytest.raises(ValueError):
        enh.capture_graph(mix.to(torch.complex128), clean)
    with pytest.raises(ValueError):
        enh.capture_graph(mix[:, :, ::2], None)            # non-contiguous view
