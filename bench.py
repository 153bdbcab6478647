#!/usr/bin/env python3
"""Throughput bench of the MISO1 -> MVDR -> MISO3 hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the complete reference semantics (6 x MISO_1 forward over the circular mic shifts +
shift alignment + clean alignment + 2 x MVDR + 2 x MISO_3 forward; reference tester.py:865-939) over one batch of
synthetic 6-mic / 16 kHz / 4 s utterances (T = 1001 frames, F = 129) already resident in HBM.  Workload =
BASELINE.json configs[3] (batch 16 per GPU, full pipeline); utterances are sharded over ranks with no data-path
collective (weak scaling).  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline     -- dominant kernel conv3x3_mfma (fp32 MFMA bound): algorithmic FLOPs / its summed launch time,
                  timed live with HIP events on the launch stream during the timed steps (misonet_profile_*).
  cpu_baseline -- the CPU oracle (oracle/: stock torch-CPU + NumPy restatement of the reference) timed on this
                  host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_MIC, N_SPK, N_SAMPLES = 6, 2, 64000
PEAK_F32_MFMA_TF = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, fp32 matrix peak (spec)
PEAK_BF16_MFMA_TF = 2500.0      # dense bf16 matrix peak (spec); the bf16x3 mode issues 3 bf16 MFMA FLOPs per algorithmic FLOP
PEAK_HBM_TBS = 8.0
ALGO_BYTES_PER_UTT = 6.23e9     # 8 forwards x 0.776 GB + 2 x 13.43 MB


def conv_flops_per_forward(in_ch, out_ch, T, en=(24, 32, 32, 32, 32, 64, 128), de=(128, 64, 32, 32, 32, 32, 24)):
    """Algorithmic FLOPs (2 x MACs) of the 3x3 conv / transposed-conv layers of one trunk forward for one sample,
    counted the way SURVEY.md 2.2 does (Conv2d: out positions x Cin x Cout x 9; ConvTranspose2d: in positions x ...)."""
    Fe = [127, 63, 31, 15, 7, 3, 1]
    mac = 0

    def dense(c0, g1, g2, F):
        return sum((c0 + i * g1) * (g1 if i < 4 else g2) for i in range(5)) * 9 * F * T
    ench = [in_ch] + list(en)
    for b in range(7):
        mac += ench[b] * ench[b + 1] * 9 * Fe[b] * T
        if b < 5:
            mac += dense(en[b], en[b], en[b], Fe[b])
    dech = list(de) + [out_ch]
    for i in range(7):
        Fi = Fe[6 - i]
        if i >= 2:
            mac += dense(2 * de[i], de[i], 2 * de[i], Fi)
        mac += 2 * de[i] * dech[i + 1] * 9 * Fi * T
    return 2.0 * mac


def conv_bytes_per_forward(in_ch, out_ch, T, en=(24, 32, 32, 32, 32, 64, 128), de=(128, 64, 32, 32, 32, 32, 24)):
    """Algorithmic HBM bytes of the conv layers of one forward-sample with layer-level fusion only (every conv reads its
    whole (concatenated) input once and writes its output once, float32): the 1.46 GB figure of SURVEY.md 8(d)."""
    Fe = [127, 63, 31, 15, 7, 3, 1]
    el = 0

    def dense(c0, g1, g2, F):
        return sum((c0 + i * g1) + (g1 if i < 4 else g2) for i in range(5)) * F
    ench = [in_ch] + list(en)
    Fin = [129] + Fe
    for b in range(7):
        el += ench[b] * Fin[b] + ench[b + 1] * Fe[b]
        if b < 5:
            el += dense(en[b], en[b], en[b], Fe[b])
    dech = list(de) + [out_ch]
    Fo = [3, 7, 15, 31, 63, 127, 129]
    for i in range(7):
        Fi = Fe[6 - i]
        if i >= 2:
            el += dense(2 * de[i], de[i], 2 * de[i], Fi)
        el += 2 * de[i] * Fi + dech[i + 1] * Fo[i]
    return 4.0 * el * T


def run_steps(enh, mix, clean, out, steps, warmup, dist, L, _lib, profile):
    """warm-up, then time exactly `steps` passes between barrier + synchronize; returns (seconds, ms_by_kind, counts)."""
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        enh.enhance(mix, clean, check_nan=False, out=out)
    barrier()
    if profile:
        _lib.check(L.misonet_profile_begin(steps * 400))
    t0 = time.perf_counter()
    for _ in range(steps):
        enh.enhance(mix, clean, check_nan=False, out=out)
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_double * 4)()
    cnt = (C.c_longlong * 4)()
    if profile:
        _lib.check(L.misonet_profile_end(ms, cnt))
    return dt, list(ms), list(cnt)


def roofline_objects(precision, B, T, steps, dt_conv_ms, n_launch, value_per_gpu):
    """roofline of the dominant kernel (the 3x3 conv launches) for one precision mode."""
    fl1 = conv_flops_per_forward(2 * N_MIC, 2 * N_SPK, T)
    fl3 = conv_flops_per_forward(2 * (N_MIC + 2), 2, T)
    by1 = conv_bytes_per_forward(2 * N_MIC, 2 * N_SPK, T)
    by3 = conv_bytes_per_forward(2 * (N_MIC + 2), 2, T)
    flops_step = B * (N_MIC * fl1 + N_SPK * fl3)
    bytes_step = B * (N_MIC * by1 + N_SPK * by3)
    conv_s = dt_conv_ms / 1e3
    ach_tf = flops_step * steps / conv_s / 1e12
    ach_tb = bytes_step * steps / conv_s / 1e12
    traffic, busy, clk = None, None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))[precision]
        traffic, busy, clk = tj["bytes_per_launch"], tj.get("mfma_busy_frac"), tj.get("clock_ghz_observed")
    except Exception:
        pass
    common = {"kernel": {"f32": "conv3x3_mfma", "bf16x3": "conv3x3_bf16x3_dma2", "bf16x3p": "conv3x3_bf16x3"}[precision],
              "launches_per_step": int(n_launch // steps), "avg_launch_ms": round(dt_conv_ms / max(n_launch, 1), 4),
              "algorithmic_gflop_per_launch": round(flops_step * steps / max(n_launch, 1) / 1e9, 2),
              "algorithmic_gbyte_per_launch": round(bytes_step * steps / max(n_launch, 1) / 1e9, 3),
              "traffic": traffic, "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, x2 fetch correction)",
              "mfma_busy_frac_pmc": busy,
              # engine clock seen in the PMC pass (the peaks below are the guide's 2.4 GHz figures; under the bf16x3 load
              # the part is power-limited, profiles/r01_clocks_power_*.txt)
              "clock_ghz_observed_pmc": clk}
    mfma_peak = PEAK_F32_MFMA_TF if precision == "f32" else PEAK_BF16_MFMA_TF
    r_mfma = dict(common, bound="mfma", achieved=round(ach_tf, 3), peak=mfma_peak, unit="TFLOP/s",
                  frac=round(ach_tf / mfma_peak, 4))
    if precision != "f32":
        r_mfma["issued_tflops"] = round(3 * ach_tf, 1)          # 3 bf16 MFMAs per algorithmic product
        r_mfma["frac_issued"] = round(3 * ach_tf / mfma_peak, 4)
    r_hbm = dict(common, bound="hbm", achieved=round(ach_tb * 1e3, 1), peak=PEAK_HBM_TBS * 1e3, unit="GB/s",
                 frac=round(ach_tb / PEAK_HBM_TBS, 4))
    # binding roofline: arithmetic intensity of the layer vs the ridge of the mode's effective matrix peak
    eff_peak = mfma_peak if precision == "f32" else mfma_peak / 3.0
    ai = flops_step / bytes_step
    binding = r_mfma if ai >= eff_peak * 1e12 / (PEAK_HBM_TBS * 1e12) else r_hbm
    return binding, (r_hbm if binding is r_mfma else r_mfma)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU per step (BASELINE configs[3]: 16)")
    ap.add_argument("--frames", type=int, default=1001)
    ap.add_argument("--precision", choices=["f32", "bf16x3", "bf16x3p"], default="bf16x3",
                    help="arithmetic of the 3x3 convs: exact f32 MFMA, or 3-term bf16 split on the bf16 MFMA (default)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short run of the other precision mode (N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket launches with HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # testing hooks for a 1-GPU box (the driver's multi-GPU runs use neither): all ranks on device 0 over gloo exercises
    # the rendezvous / barrier / MAX-reduce path of this script without a second GPU
    one_dev = bool(os.environ.get("MISONET_BENCH_ONE_DEVICE"))
    backend = os.environ.get("MISONET_BENCH_BACKEND", "nccl")
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import misonet_amd as mz
    from misonet_amd import _lib, stft, weights as W

    sd1 = W.make_state_dict(W.miso1_spec(), 0)
    sd3 = W.make_state_dict(W.miso3_spec(), 1)
    m1 = mz.MISO_1(N_SPK, N_MIC, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(local_rank)
    m1.load_state_dict(sd1)
    m3 = mz.MISO_3(1, N_MIC, 7, list(W.DEFAULT_EN_CH), list(W.DEFAULT_DE_CH), "IN").cuda(local_rank)
    m3.load_state_dict(sd3)
    m1.set_precision(args.precision)
    m3.set_precision(args.precision)
    enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=N_SPK, ref_ch=0)

    # ---- synthetic inputs (SURVEY.md 8(d) config 2-5 generator), global utterance index = rank*B + i ----
    B, T = args.batch, args.frames
    n = (T - 1) * 64
    mixes, cleans = [], []
    for i in range(B):
        obs, s0, s1 = W.synthetic_utterance(rank * B + i, n)
        mixes.append(stft.stft(torch.from_numpy(obs.T.copy()).to(dev)))                       # [M,T,F]
        cleans.append(torch.stack([stft.stft(torch.from_numpy(s[:, 0].copy()).to(dev)) for s in (s0, s1)]))
    mix = torch.stack(mixes).contiguous()
    clean = torch.stack(cleans).contiguous()
    out = torch.empty((B, N_SPK, T, 129), dtype=torch.complex64, device=dev)

    L = _lib.lib()
    profile = not args.no_profile
    dt, ms, cnt = run_steps(enh, mix, clean, out, args.steps, args.warmup, dist, L, _lib, profile)
    if not os.environ.get("MISONET_BENCH_NOCHECK"):      # (timing experiments with deliberately wrong results)
        _lib.check(L.misonet_pipeline_check(enh._pipe, enh.workspace(B, T).data_ptr(), _lib.stream_ptr(dev)))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        utt = world * B * args.steps
        value = utt / dt
        roof, roof2 = None, None
        if profile and cnt[0] > 0:
            roof, roof2 = roofline_objects(args.precision, B, T, args.steps, ms[0], cnt[0], value / world)
            roof["time_share"] = {"conv_ms_per_step": round(ms[0] / args.steps, 2),
                                  "tcn_ms_per_step": round(ms[1] / args.steps, 2),
                                  "mvdr_ms_per_step": round(ms[2] / args.steps, 2)}
            roof["hbm_frac_pipeline"] = round(ALGO_BYTES_PER_UTT * (value / world) / (PEAK_HBM_TBS * 1e12), 4)
        alt = None
        if world == 1 and not args.no_alt and profile:
            other = "f32" if args.precision != "f32" else "bf16x3"
            m1.set_precision(other)
            m3.set_precision(other)
            k = max(2, min(3, args.steps))
            dt2, ms2, cnt2 = run_steps(enh, mix, clean, out, k, 1, None, L, _lib, True)
            r2, _ = roofline_objects(other, B, T, k, ms2[0], cnt2[0], B * k / dt2)
            alt = {"dtype": other, "value": round(B * k / dt2, 3), "unit": "utt/s", "steps": k, "roofline": r2}
            m1.set_precision(args.precision)
            m3.set_precision(args.precision)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sd1, sd3, T)
        line = {
            "metric": "utterances/sec MISO1->MVDR->MISO3, 6-mic 16kHz 4s",
            "value": round(value, 3), "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: synthetic 6-mic 16 kHz 4 s, full MISO1x6 -> align -> MVDRx2 -> MISO3x2",
                       "batch_per_gpu": B, "frames": T, "freq_bins": 129, "parallelism": f"utterance-shard x{world}"},
            "realtime_factor": round(value * (n / 16000.0), 2),
            "roofline": roof, "roofline_other": roof2, "alt_precision": alt, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sd1, sd3, T):
    """The oracle (kind "port": our stock-torch-CPU/NumPy restatement of the reference path, B = 1 per call as in
    tester.py) on this host's cores.  Bounded sample: 1 forward warm-up, then whole utterances until >= 12 s."""
    from misonet_amd import weights as W
    from oracle import pipeline_oracle, miso_oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    n = (T - 1) * 64

    def utt(u):
        obs, s0, s1 = W.synthetic_utterance(u, n)
        mix = pipeline_oracle.stft_chunk(obs)
        clean = np.stack([pipeline_oracle.stft_chunk(s0)[0], pipeline_oracle.stft_chunk(s1)[0]])
        return mix, clean
    mix, clean = utt(0)
    miso_oracle.miso1_forward(torch.from_numpy(mix[None]), sd1)          # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=0)
        done += 1
        el = time.perf_counter() - t0
        if el >= 12.0 or done >= 4:
            break
    return {"value": round(done / el, 4), "unit": "utt/s", "cores": threads, "kind": "port",
            "sample": f"{done} utterance(s) of the same synthetic workload (T={T}), B=1 per call, {el:.1f} s"}


if __name__ == "__main__":
    main()
