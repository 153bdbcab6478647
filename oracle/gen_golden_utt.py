#!/usr/bin/env python3
"""Golden G9: the reference's utterance-wise MVDR mode (Tester_Beamforming.inference with utterance_flag=True,
reference tester.py:340-449) on one synthetic recording of two splits.  Run: python -m oracle.gen_golden_utt
Same shims / weights as oracle/gen_golden.py."""
import os

import numpy as np
import torch

from oracle.gen_golden import import_reference, build_models, OUT

FRAMES, HOP = 32, 64


def main():
    ref_model, ref_tester, sf_stub = import_reference()
    m1, _ = build_models(ref_model)
    from misonet_amd.weights import synthetic_utterance
    from misonet_amd.stft import split_chunks
    from oracle.pipeline_oracle import stft_chunk
    chunk = (FRAMES - 1) * HOP
    obs, s0, s1 = synthetic_utterance(31, 2 * chunk - 300)
    po, gap = split_chunks(obs, chunk)
    p0, _ = split_chunks(s0, chunk)
    p1, _ = split_chunks(s1, chunk)
    od = {str(k): torch.from_numpy(stft_chunk(po[k]))[None] for k in range(2)}
    d0 = {str(k): torch.from_numpy(stft_chunk(p0[k]))[None] for k in range(2)}
    d1 = {str(k): torch.from_numpy(stft_chunk(p1[k]))[None] for k in range(2)}
    tst = ref_tester.Tester_Beamforming("SMS_WSJ", None, None, None, m1, 6, "cpu", 2, chunk / 16000.0, "/tmp/golden_out",
                                        0, False, False, True, fs=16000, window="hann", length=256, overlap=192)
    # shim (5): the utterance path hands torch tensors to the NumPy MVDR (tester.py:442); with torch >= 1.10 the lazy
    # conjugate of a torch tensor cannot be consumed by np.einsum, so convert to ndarray at the call (dtype unchanged).
    orig_bf = tst.Apply_Beamforming
    tst.Apply_Beamforming = lambda s_, m_, epsi=1e-6: orig_bf(
        s_.resolve_conj().numpy() if isinstance(s_, torch.Tensor) else s_,
        m_.resolve_conj().numpy() if isinstance(m_, torch.Tensor) else m_, epsi)
    sf_stub.written.clear()
    os.makedirs("/tmp/golden_out", exist_ok=True)
    tst.inference([(od, d0, d1, torch.tensor([gap]), ["utt"])], "/tmp/golden_out")
    wavs = [w[1] for w in sf_stub.written]
    assert len(wavs) == 2, len(wavs)
    np.savez_compressed(os.path.join(OUT, "g9_utterance_mvdr.npz"), utt=np.int64(31), frames=np.int64(FRAMES),
                        gap=np.int64(gap), wav0=wavs[0].astype(np.int16).ravel(), wav1=wavs[1].astype(np.int16).ravel())
    print("G9", [w.shape for w in wavs], gap)


if __name__ == "__main__":
    main()
