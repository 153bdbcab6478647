#!/usr/bin/env python3
"""bench.py's 8-ranks-on-one-device shape as a determinism test: every rank repeats the SAME Enhancer.enhance pass, all ranks
released together by a gloo barrier (so that eight processes launch the same kernels at the same moment), and compares each result
with its first one bit for bit.  torchrun --nproc-per-node 8 tools/experiments/mp_stress2.py [mode] [reps] [extras]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import misonet_amd as mz                         # noqa: E402
from misonet_amd import weights as W, stft      # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
extras = len(sys.argv) > 3 and sys.argv[3] == "1"
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
B, T = 2, 1001
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
for u in range(rank * B, rank * B + B):
    obs, s0, s1 = W.synthetic_utterance(u, n)
    mixes.append(stft.stft(torch.from_numpy(obs.T.copy()).cuda()))
    cleans.append(torch.stack([stft.stft(torch.from_numpy(s[:, 0].copy()).cuda()) for s in (s0, s1)]))
mix, clean = torch.stack(mixes).contiguous(), torch.stack(cleans).contiguous()
out = torch.empty((B, 2, T, 129), dtype=torch.complex64, device="cuda:0")
ref = None
bad = 0
for it in range(reps):
    dist.barrier()
    torch.cuda.synchronize()
    if extras:
        o, ex = enh.enhance(mix, clean, check_nan=False, out=out, want_bf=True, want_miso1=True)
        cur = {"miso1": ex["miso1"].clone(), "bf": ex["bf"].clone(), "out": out.clone()}
    else:
        enh.enhance(mix, clean, check_nan=False, out=out)
        cur = {"out": out.clone()}
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    diff = [k for k in cur if not torch.equal(cur[k], ref[k])]
    if diff:
        bad += 1
        k = diff[0]
        d = (cur[k] - ref[k]).abs()
        nz = torch.nonzero(d > 0)
        print(f"[rank {rank}] rep {it}: differs in {diff}; {k}: {nz.shape[0]} of {d.numel()} elements, max {float(d.max()):.3e} (|ref| max {float(ref[k].abs().max()):.3e}); "
              f"index ranges {[(int(nz[:, j].min()), int(nz[:, j].max())) for j in range(nz.shape[1])]}", flush=True)
dist.barrier()
print(f"[rank {rank}] {mode}: {bad} of {reps - 1} repetitions differ from the first", flush=True)
