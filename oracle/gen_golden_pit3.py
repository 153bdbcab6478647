#!/usr/bin/env python3
"""Golden G10: the REAL reference's MISO1_Inference (tester.py:1014-1068) with num_spks = 3 -- its PIT alignment
enumerates all 3! permutations (tester.py:1053-1064).  Run from the repo root in the build container:

    python -m oracle.gen_golden_pit3

Writes tests/golden/g10_miso1_inference_S3_T32.npz (input, the three aligned speakers at all six microphones, about
0.6 MB as float16 pairs would lose the parity margin, so complex64 of every second frame is stored + full checksums).
Weights are not stored: misonet_amd.weights.make_state_dict(miso1_spec(num_spks=3), seed=2).
"""
import os

import numpy as np
import torch

from oracle.gen_golden import import_reference, synth_spec, OUT


def main():
    from misonet_amd import weights as W
    ref_model, ref_tester, _ = import_reference()
    torch.set_num_threads(8)
    sd = W.make_state_dict(W.miso1_spec(num_spks=3), seed=2)
    m1 = ref_model.MISO_1(3, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").eval()
    assert list(m1.state_dict().keys()) == list(sd.keys())
    m1.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tst = ref_tester.Tester_Enhance("SMS_WSJ", "MISO3", None, None, m1, None, 6, "cpu", 3, 31 * 64 / 16000,
                                    "/tmp/golden_out", 0, False, fs=16000, window="hann", length=256, overlap=192)
    x = synth_spec(1000, (1, 6, 32, 129))
    est = tst.MISO1_Inference(torch.from_numpy(x), ref_ch=0)          # list of 3 x [1,6,32,129]
    est = np.stack([e.numpy()[0] for e in est]).astype(np.complex64)  # [3,6,32,129]
    np.savez_compressed(os.path.join(OUT, "g10_miso1_inference_S3_T32.npz"), x=x, est_even=est[:, :, ::2],
                        mag_sum=np.abs(est).sum(axis=-1).astype(np.float32))
    print("G10", est.shape, np.abs(est).mean())


if __name__ == "__main__":
    main()
