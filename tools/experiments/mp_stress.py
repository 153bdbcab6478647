#!/usr/bin/env python3
"""Several processes on ONE GPU, each repeating the same forward: are the results still bit-identical run to run?  (Kernels of
different processes share CUs; a latent race between the waves of a workgroup shows up as a result that changes.)
usage: mp_stress.py [nproc] [mode] [reps] [T] [B]        (experiment driver, GPU box only)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(idx, mode, reps, T, B):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import misonet_amd as mz
    from misonet_amd import weights as W
    sd1 = W.make_state_dict(W.miso1_spec(), 0)
    m = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m.load_state_dict(sd1)
    m.eval().set_precision(mode)
    m.keep_activations(True)
    r = np.random.default_rng(5)
    x = torch.from_numpy((r.standard_normal((B, 6, T, 129)) + 1j * r.standard_normal((B, 6, T, 129))).astype(np.complex64)).cuda()
    names = ["enc0_conv"] + [f"enc{b}" for b in range(7)] + ["tcn_out"] + [f"dec{b}" for b in range(7)]
    ref = None
    bad = 0
    for it in range(reps):
        y = m(x, check_nan=False).clone()
        taps = {nm: m.tap(nm, B, T).clone() for nm in names}
        torch.cuda.synchronize()
        if ref is None:
            ref = (y, taps)
            continue
        if not torch.equal(y, ref[0]):
            bad += 1
            first = next((nm for nm in names if not torch.equal(taps[nm], ref[1][nm])), "?")
            d = (taps[first] - ref[1][first]).abs() if first != "?" else None
            where = ""
            if d is not None:
                idx_ = torch.nonzero(d > 0)
                where = f" first tap {first}: {idx_.shape[0]} elements differ, max {float(d.max()):.3e}, channels {sorted(set(idx_[:, 1].tolist()))[:8]} samples {sorted(set(idx_[:, 0].tolist()))}"
            print(f"[proc {idx}] rep {it}: output differs from rep 0;{where}", flush=True)
    print(f"[proc {idx}] {mode}: {bad} of {reps - 1} repetitions differ", flush=True)


def worker_pipe(idx, mode, reps, T, B):
    """the whole MISO1 x 6 -> PIT -> MVDR -> MISO3 pass of Enhancer.enhance"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import misonet_amd as mz
    from misonet_amd import weights as W, stft
    sd1 = W.make_state_dict(W.miso1_spec(), 0)
    sd3 = W.make_state_dict(W.miso3_spec(), 1)
    m1 = mz.MISO_1(2, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m1.load_state_dict(sd1)
    m3 = mz.MISO_3(1, 6, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(0)
    m3.load_state_dict(sd3)
    m1.set_precision(mode); m3.set_precision(mode)
    enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=2, ref_ch=0)
    n = (T - 1) * 64
    mixes, cleans = [], []
    for u in range(14, 14 + B):
        obs, s0, s1 = W.synthetic_utterance(u, n)
        mixes.append(stft.stft(torch.from_numpy(obs.T.copy()).cuda()))
        cleans.append(torch.stack([stft.stft(torch.from_numpy(s[:, 0].copy()).cuda()) for s in (s0, s1)]))
    mix, clean = torch.stack(mixes).contiguous(), torch.stack(cleans).contiguous()
    ref = None
    bad = 0
    for it in range(reps):
        out, extra = enh.enhance(mix, clean, check_nan=False, want_bf=True, want_miso1=True)
        cur = {"miso1": extra["miso1"].clone(), "bf": extra["bf"].clone(), "out": out.clone()}
        torch.cuda.synchronize()
        if ref is None:
            ref = cur
            continue
        diff = [k for k in ("miso1", "bf", "out") if not torch.equal(cur[k], ref[k])]
        if diff:
            bad += 1
            k = diff[0]
            d = (cur[k] - ref[k]).abs()
            print(f"[proc {idx}] rep {it}: differs in {diff}; {k}: {int((d > 0).sum())} elements, max {float(d.max()):.3e} (|ref| max {float(ref[k].abs().max()):.3e})", flush=True)
    print(f"[proc {idx}] pipeline {mode}: {bad} of {reps - 1} repetitions differ", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        if os.environ.get("MP_STRESS_PIPE"):
            worker_pipe(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
            sys.exit(0)
        worker(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
        sys.exit(0)
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x6"
    reps = sys.argv[3] if len(sys.argv) > 3 else "20"
    T = sys.argv[4] if len(sys.argv) > 4 else "1001"
    B = sys.argv[5] if len(sys.argv) > 5 else "2"
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(i), mode, reps, T, B]) for i in range(nproc)]
    for p in ps:
        p.wait()
