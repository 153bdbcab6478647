#!/usr/bin/env python3
"""Throughput bench of the MISO1 -> MVDR -> MISO3 hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: either launched by the driver as ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) or started plainly -- then this script re-executes itself under
``torch.distributed.run`` with N ranks on 127.0.0.1, one rank per GPU over RCCL.

A "step" = one pass of the complete reference semantics (6 x MISO_1 forward over the circular mic shifts +
shift alignment + clean alignment + 2 x MVDR + 2 x MISO_3 forward; reference tester.py:865-939) over one batch of
synthetic 6-mic / 16 kHz / 4 s utterances (T = 1001 frames, F = 129) already resident in HBM.  Workload at N = 1 =
BASELINE.json configs[3] (batch 16, full pipeline); at N = 8 = configs[4] (batch 128 = 8 x 16): utterances are sharded
over ranks with no data-path collective (weak scaling; ``config.workload`` names what ran).  Every N > 1 line verifies itself
(after the timed loop: each rank repeats its pass bit for bit, the results are all_gathered over RCCL -- the path's one
collective -- checksummed per shard, and rank 0 checks an utterance of the LAST rank against the CPU oracle: ``parity``,
``shard_checksums_match``, ``second_pass_bit_identical``; ``--no-verify-gather`` skips it).  Prints ONE JSON line on rank 0.

The timed loop is un-instrumented.  The headline arithmetic is ``HEADLINE_PRECISION`` -- "auto": a short un-timed calibration
picks between the two fp32-faithful fast modes on this box (``headline_selection``) -- ; the other two product modes are timed
beside it with the same steps / warm-up and reported under ``alt_precision``.

Extra objects on the line:
  roofline     -- dominant kernel (the 3x3 conv launches, MFMA bound): algorithmic FLOPs / summed launch time, from a
                  SEPARATE pass of the same workload with HIP events around every launch on the launch stream
                  (misonet_profile_*); ``instrumented_ms_per_step`` shows what the events cost.
  cpu_baseline -- the CPU oracle (oracle/: stock torch-CPU + NumPy restatement of the reference) timed on this
                  host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_MIC, N_SPK, N_SAMPLES = 6, 2, 64000
PEAK_F32_MFMA_TF = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, fp32 matrix peak (spec)
PEAK_BF16_MFMA_TF = 2500.0      # dense bf16 matrix peak (spec)
PEAK_HBM_TBS = 8.0
ALGO_BYTES_PER_UTT = 6.23e9     # 8 forwards x 0.776 GB + 2 x 13.43 MB

# arithmetic of the 3x3 convs: name -> (dominant kernel, 16-bit MFMA products issued per algorithmic product, or 0 for
# the f32 MFMA, operands exact float32?, significant bits per operand)
MODES = {
    # ---- the product library's three modes ----
    "f32":     ("conv3x3_mfma", 0, True, 24),            # v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain
    "f32w":    ("conv3x3_wino_f32", 0, True, 24),        # f32 MFMA, DenseBlock convs in Winograd F(2x2, 3x3) form (conv_wino.hip)
    "bf16x6":  ("conv3x3_bf16x6", 6, True, 24),          # operands split EXACTLY into 3 bf16 pieces, 6 leading terms
    # ---- experiment build only (csrc: make exp; MISONET_LIB_PATH): measured alternatives that are not product modes ----
    "bf16x6w": ("conv3x3_wino_x6", 6, True, 24),         # bf16x6 arithmetic, DenseBlock convs in Winograd F(2x2, 3x3) form (conv_wino6.hip)
    "f16x3":   ("conv3x3_bf16x3_dma2<F16>", 3, False, 22),   # 2 fp16 pieces (22 bits, "3xTF32"), 3 terms: measured at the
                                                         # f32 mode's error level, but the operands are rounded
    "bf16x3":  ("conv3x3_bf16x3_dma2", 3, False, 16),    # 2 bf16 pieces (16 bits), 3 terms: inside 1e-3, not fp32-faithful
    "bf16x3p": ("conv3x3_bf16x3", 3, False, 16),
}
# The headline arithmetic.  "auto" (round 6): the two fp32-faithful fast modes are within 1 % of each other and which one leads
# depends on the box (`bf16x6` runs against the socket power limit at 1.75 GHz, `f32w` at 2.29 GHz: five boxes gave f32w / bf16x6 =
# 0.984 ... 1.014) -- a short un-timed calibration on THIS box picks the literal-float32 mode `f32w` when it is at least as fast
# as `bf16x6` (within HEADLINE_TIE of it: the calibration's own noise), `bf16x6` otherwise; the line says which and why
# (`headline_selection`).  `--precision MODE` fixes the mode.
HEADLINE_PRECISION = "auto"
HEADLINE_CANDIDATES = ("bf16x6", "f32w")
HEADLINE_TIE = 0.005


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


def dense_flops_per_forward(in_ch, out_ch, T, en=(24, 32, 32, 32, 32, 64, 128), de=(128, 64, 32, 32, 32, 32, 24)):
    """the share of conv_flops_per_forward in the 50 stride-1 same-padded DenseBlock convs (model.py:437-482): the layers the
    f32w mode runs in Winograd F(2x2, 3x3) form, 16 instead of 36 products per 2 x 2 outputs"""
    Fe = [127, 63, 31, 15, 7, 3, 1]

    def dense(c0, g1, g2, F):
        return sum((c0 + i * g1) * (g1 if i < 4 else g2) for i in range(5)) * 9 * F * T
    mac = sum(dense(en[b], en[b], en[b], Fe[b]) for b in range(5))
    mac += sum(dense(2 * de[i], de[i], 2 * de[i], Fe[6 - i]) for i in range(2, 7))
    return 2.0 * mac


def conv_bytes_per_forward(in_ch, out_ch, T, en=(24, 32, 32, 32, 32, 64, 128), de=(128, 64, 32, 32, 32, 32, 24),
                           inner_elem_bytes=4.0):
    """Algorithmic HBM bytes of the conv layers of one forward-sample with layer-level fusion only (every conv reads its
    whole (concatenated) input once and writes its output once, float32): the 1.46 GB figure of SURVEY.md 8(d).
    inner_elem_bytes: bytes per element of every tensor except the network input / output (6 in the oct3 layout of the
    bf16x6 mode: three bf16 pieces)."""
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
    el_io = ench[0] * Fin[0] + dech[7] * Fo[6]
    return (4.0 * el_io + inner_elem_bytes * (el - el_io)) * T


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


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))


def _git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


def _source_sha16():
    """sha256 (first 16 hex digits) over the kernel sources and this script: identifies the tree a counter measurement
    belongs to on boxes that have no .git (the GPU boxes get a snapshot of the working tree)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "misonet_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "misonet_amd", "csrc", "*.hpp"))) + [os.path.abspath(__file__)]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _pmc_read_db(db_path):
    """{counter: {kernel: [dispatches, sum]}} and {kernel: [dispatches, total_ns]} of one rocprofv3 --pmc --kernel-trace run
    (rocpd sqlite output; tools/rocpd_pmc.py / rocpd_stats.py print the same tables)."""
    import re
    import sqlite3
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'")]
    src = "counters_collection" if "counters_collection" in views else "pmc_events"
    cols = [r[1] for r in cur.execute(f"pragma table_info({src})")]
    ncol = "kernel_name" if "kernel_name" in cols else "name"
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else "id"
    ctr = {}
    for name, c, _d, val in cur.execute(f"select {ncol}, {ccol}, {dcol}, sum({vcol}) from {src} group by {ncol}, {ccol}, {dcol}"):
        a = ctr.setdefault(c, {}).setdefault(re.sub(r"\(.*$", "", name), [0, 0.0])
        a[0] += 1
        a[1] += val
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    kn = "name" if "name" in kcols else "kernel_name"
    dur = {}
    for name, st, en in cur.execute(f"select {kn}, start, end from kernels"):
        a = dur.setdefault(re.sub(r"\(.*$", "", name), [0, 0])
        a[0] += 1
        a[1] += en - st
    return ctr, dur


def pmc_live(precision, B, T, timeout_s=150):
    """Hardware counters of THIS box at THIS tree: re-executes this script (1 warm-up + 1 timed step, nothing else) under
    ``rocprofv3 --pmc ... --kernel-trace`` once per counter group -- FETCH_SIZE, WRITE_SIZE (they do not fit one pass),
    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -- and derives, over the conv3x3_* launches, as the guide's HBM / rocprofv3
    sections prescribe:
        bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches   (FETCH_SIZE counts 1/2 of a wide stream on gfx950)
        mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
        clock_ghz        = GRBM_GUI_ACTIVE / 8 / summed kernel duration of the SAME pass
        useful_over_issued = 6 * algorithmic FLOPs / (SQ_VALU_MFMA_BUSY_CYCLES / 32 * 32768)   (bf16x6: 32-cycle 32x32x16 MFMAs)
    Returns None when rocprofv3 is missing or a pass fails (the caller falls back to the committed file and says so)."""
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-profile",
           "--no-alt", "--no-pmc", "--precision", precision, "--batch", str(B), "--frames", str(T)]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    ctr_all, dur_of = {}, {}
    t_all = time.perf_counter()
    for grp in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="misonet_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", *grp, "--kernel-trace", "-d", d, "-o", "pmc", "--"] + cmd, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "*.db")) + glob.glob(os.path.join(d, "*", "*.db"))
            if r.returncode != 0 or not dbs:
                return None
            ctr, dur = _pmc_read_db(dbs[0])
            for c in grp:
                if c not in ctr:
                    return None
                ctr_all[c] = ctr[c]
                dur_of[c] = dur
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def conv_sum(c):
        ks = {k: v for k, v in ctr_all[c].items() if "conv3x3_" in k}
        return sum(v[1] for v in ks.values()), sum(v[0] for v in ks.values())
    fetch_kb, n_f = conv_sum("FETCH_SIZE")
    write_kb, n_w = conv_sum("WRITE_SIZE")
    busy, n_b = conv_sum("SQ_VALU_MFMA_BUSY_CYCLES")
    gui, _ = conv_sum("GRBM_GUI_ACTIVE")
    if not (n_f and n_f == n_w == n_b and gui > 0):
        return None
    conv_ns = sum(v[1] for k, v in dur_of["GRBM_GUI_ACTIVE"].items() if "conv3x3_" in k)
    terms = MODES[precision][1]
    passes = 2                                                         # warm-up + the timed step
    flops = passes * B * (N_MIC * conv_flops_per_forward(2 * N_MIC, 2 * N_SPK, T) + N_SPK * conv_flops_per_forward(2 * (N_MIC + 2), 2, T))
    out = {"conv_launches": int(n_f), "fetch_kb": fetch_kb, "write_kb": write_kb,
           "bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024.0 / n_f),
           "mfma_busy_frac": round(busy / (1024.0 * gui / 8.0), 4),
           "clock_ghz_observed": round(gui / 8.0 / conv_ns, 3) if conv_ns else None,
           "pipeline_passes_profiled": passes, "collect_seconds": round(time.perf_counter() - t_all, 1),
           "git_head": _git_head(), "source_sha16": _source_sha16()}
    if terms:                                                          # 32 pipe cycles and 32768 FLOP per 32x32x16 16-bit MFMA
        out["useful_over_issued_mfma"] = round(terms * flops / (busy / 32.0 * 32768.0), 4)
    return out


def _traffic_entry(precision):
    """measured HBM bytes per conv launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/)"""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04z_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if precision in tj:
                return tj[precision], f"profiles/{name}"
        except Exception:
            pass
    return None, None


def roofline_objects(precision, B, T, steps, dt_conv_ms, n_launch, live=None):
    """roofline of the dominant kernel (the 3x3 conv launches) for one precision mode.  live: pmc_live()'s dict (counters of
    this box, this tree) or None -> the committed measurement under profiles/, labelled as such."""
    kernel, terms = MODES[precision][:2]
    fl1 = conv_flops_per_forward(2 * N_MIC, 2 * N_SPK, T)
    fl3 = conv_flops_per_forward(2 * (N_MIC + 2), 2, T)
    by1 = conv_bytes_per_forward(2 * N_MIC, 2 * N_SPK, T)
    by3 = conv_bytes_per_forward(2 * (N_MIC + 2), 2, T)
    flops_step = B * (N_MIC * fl1 + N_SPK * fl3)
    bytes_step = B * (N_MIC * by1 + N_SPK * by3)
    ieb = 6.0 if precision == "bf16x6" else 4.0          # the oct3 layout stores three bf16 pieces per element
    lay_step = B * (N_MIC * conv_bytes_per_forward(2 * N_MIC, 2 * N_SPK, T, inner_elem_bytes=ieb) +
                    N_SPK * conv_bytes_per_forward(2 * (N_MIC + 2), 2, T, inner_elem_bytes=ieb))
    conv_s = dt_conv_ms / 1e3
    ach_tf = flops_step * steps / conv_s / 1e12
    ach_tb = bytes_step * steps / conv_s / 1e12
    tj, tsrc = _traffic_entry(precision)
    if live:
        tj, tsrc = live, None
    traffic = tj["bytes_per_launch"] if tj else None
    common = {"kernel": kernel,
              "launches_per_step": int(n_launch // steps), "avg_launch_ms": round(dt_conv_ms / max(n_launch, 1), 4),
              "algorithmic_gflop_per_launch": round(flops_step * steps / max(n_launch, 1) / 1e9, 2),
              "algorithmic_gbyte_per_launch": round(bytes_step * steps / max(n_launch, 1) / 1e9, 3),
              # the same read-once / write-once count at the bytes per element of the mode's activation layout
              "layout_gbyte_per_launch": round(lay_step * steps / max(n_launch, 1) / 1e9, 3),
              "traffic": traffic,
              "traffic_over_layout_bytes": round(traffic / (lay_step * steps / max(n_launch, 1)), 3) if traffic else None,
              "traffic_source": ("live: rocprofv3 --pmc passes of this script on this box (FETCH_SIZE x2 + WRITE_SIZE; "
                                 f"{live.get('conv_launches')} conv launches, {live.get('collect_seconds')} s)") if live
              else ((tsrc + " (committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of the N = 1 command on the builder's box, x2 "
                             "fetch correction: NOT this run's counters)") if tsrc else None),
              # the three PMC-derived fields (traffic, mfma_busy_frac_pmc, clock_ghz_observed_pmc): live = counters of THIS
              # box at THIS tree (bench.py re-executes itself under rocprofv3 --pmc, one pass per counter group); otherwise
              # the committed measurement of the same command (another box / day), stamped with the commit it was taken at
              "pmc_fields_measured_live": bool(live),
              "pmc_git_head": (live.get("git_head") if live else (tj.get("git_head") if tj else None)),
              # the tree the counters belong to (kernel sources + this script); equals source_sha16 of the line when live
              "pmc_source_sha16": (live.get("source_sha16") if live else (tj.get("source_sha16") if tj else None)),
              "useful_over_issued_mfma_pmc": tj.get("useful_over_issued_mfma") if tj else None,
              "mfma_busy_frac_pmc": tj.get("mfma_busy_frac") if tj else None,
              # engine clock seen in the PMC pass (the peaks below are the guide's 2.4 GHz figures; under the bf16 MFMA
              # load the part is power-limited)
              "clock_ghz_observed_pmc": tj.get("clock_ghz_observed") if tj else None}
    # peak of the arithmetic: the guide's dense MFMA peak of the instruction the mode issues, divided by the number of
    # MFMA products the mode spends per algorithmic product (1 for the f32 MFMA; 6 / 3 for the split modes on the
    # 2.5 PF bf16 / fp16 MFMA) -- i.e. the rate the mode would reach with the matrix pipe 100 % busy on useful work
    raw_peak = PEAK_BF16_MFMA_TF if terms else PEAK_F32_MFMA_TF
    mfma_peak = round(raw_peak / terms, 1) if terms else raw_peak
    wino = None
    if precision in ("f32w", "bf16x6w"):
        # the DenseBlock layers issue 16 / 36 of their algorithmic products (Winograd F(2x2, 3x3)), the other layers all of them:
        # peak = the algorithmic rate at which the matrix pipe would be 100 % busy with exactly those products (x 6 MFMAs each in bf16x6w)
        dn = B * (N_MIC * dense_flops_per_forward(2 * N_MIC, 2 * N_SPK, T) + N_SPK * dense_flops_per_forward(2 * (N_MIC + 2), 2, T))
        issued = dn * 16.0 / 36.0 + (flops_step - dn)
        base_peak = mfma_peak
        mfma_peak = round(base_peak * flops_step / issued, 1)
        wino = {"dense_block_share_of_flops": round(dn / flops_step, 4), "issued_over_algorithmic_products": round(issued / flops_step, 4),
                "issued_tflops": round(ach_tf * issued / flops_step, 1),
                "frac_of_direct_peak": round(ach_tf / base_peak, 4)}
    r_mfma = dict(common, bound="mfma", achieved=round(ach_tf, 3), peak=mfma_peak, unit="TFLOP/s",
                  frac=round(ach_tf / mfma_peak, 4))
    if wino:
        r_mfma["peak_basis"] = (f"{base_peak} TF/s direct-form peak of the arithmetic x algorithmic / issued products (Winograd "
                                "F(2x2,3x3) on the DenseBlock convs: 16 of 36)")
        r_mfma.update(wino)
        if terms:
            r_mfma["mfma_products_per_product"] = terms
    if terms and not wino:
        r_mfma["peak_basis"] = f"{raw_peak:.0f} TF/s dense 16-bit MFMA peak / {terms} MFMA products per algorithmic product"
        r_mfma["mfma_products_per_product"] = terms
        r_mfma["issued_tflops"] = round(terms * ach_tf, 1)
        r_mfma["frac_of_raw_16bit_peak"] = round(ach_tf / raw_peak, 4)
    r_hbm = dict(common, bound="hbm", achieved=round(ach_tb * 1e3, 1), peak=PEAK_HBM_TBS * 1e3, unit="GB/s",
                 frac=round(ach_tb / PEAK_HBM_TBS, 4))
    # binding roofline: arithmetic intensity of the layer vs the ridge of the mode's effective matrix peak (f32w: 107 FLOP/B against a
    # ridge of 333 TF/s / 8 TB/s = 42); the other object is on the line as roofline_other
    eff_peak = mfma_peak
    ai = flops_step / bytes_step
    binding = r_mfma if ai >= eff_peak * 1e12 / (PEAK_HBM_TBS * 1e12) else r_hbm
    return binding, (r_hbm if binding is r_mfma else r_mfma)


def wav_leg(enh, W, rank, B, n, steps, warmup, dev, headline_value):
    """wav in -> int16 wav out (reference tester.py:865-867 H2D, 949-974 iSTFT -> int16): the same batch as the headline,
    entering as float32 waveforms [B, n, 6] (+ the clean references [B, n, 2]) and leaving as int16 [B, 2, n]."""
    wav_h = torch.empty((B, n, N_MIC), dtype=torch.float32).pin_memory()
    clean_h = torch.empty((B, n, N_SPK), dtype=torch.float32).pin_memory()
    for i in range(B):
        obs, s0, s1 = W.synthetic_utterance(rank * B + i, n)
        wav_h[i] = torch.from_numpy(obs)
        clean_h[i, :, 0] = torch.from_numpy(s0[:, 0].copy())
        clean_h[i, :, 1] = torch.from_numpy(s1[:, 0].copy())
    wav_d, clean_d = wav_h.to(dev), clean_h.to(dev)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    # (a) device-resident: inputs in HBM, int16 result left in HBM
    def dev_loop(k):
        pcm = None
        for _ in range(k):
            pcm = enh.enhance_wav_int16(wav_d, clean_d, check_nan=False)
        return pcm
    dev_loop(warmup)
    dt_dev, pcm_d = timed(lambda: dev_loop(steps))

    # (b) host-resident, copies overlapped with compute on two extra streams (Enhancer.stream_wav)
    def gen(k):
        for _ in range(k):
            yield (wav_h, clean_h)

    def host_loop(k):
        last = None
        for pcm in enh.stream_wav(gen(k), check_nan=False):
            last = pcm
        return last
    host_loop(max(warmup, 2))
    dt_host, pcm_h = timed(lambda: host_loop(steps))

    # (c) host-resident, serial (what the per-split .to(device) / .cpu() of the reference harness does)
    def serial_loop(k):
        last = None
        for _ in range(k):
            last = enh.enhance_wav_int16(wav_h.to(dev), clean_h.to(dev), check_nan=False).cpu()
        return last
    serial_loop(1)
    dt_ser, _ = timed(lambda: serial_loop(steps))
    same = bool(np.array_equal(pcm_h, pcm_d.cpu().numpy()))
    v_dev, v_host, v_ser = B * steps / dt_dev, B * steps / dt_host, B * steps / dt_ser
    return {"unit": "utt/s", "steps": steps,
            "device_resident": round(v_dev, 3), "host_resident_overlapped": round(v_host, 3),
            "host_resident_serial": round(v_ser, 3),
            "vs_headline": {"device_resident": round(v_dev / headline_value, 4),
                            "host_resident_overlapped": round(v_host / headline_value, 4),
                            "host_resident_serial": round(v_ser / headline_value, 4)},
            "pcie_bytes_per_utterance": {"h2d": n * (N_MIC + N_SPK) * 4, "d2h": N_SPK * n * 2},
            "host_equals_device_result": same,
            "what": "float32 wav [B, 64000, 6] (+ clean [B, 64000, 2]) -> HIP STFT (stft_pack_k) -> MISO1x6/align/MVDRx2/"
                    "MISO3x2 -> ONE istft_k launch (inverse DFT, overlap-add, x 32767, truncating cast) -> int16 [B, 2, 64000]; "
                    "the headline starts and ends at spectrograms"}, pcm_d


def workload_name(world, B):
    """BASELINE.json's name of what this launch runs: configs[3] = batch 16 on one GPU, configs[4] = batch 128 sharded over
    8 GPUs (8 x 16); other rank counts / batch sizes run configs[4]'s sharding at their own global batch."""
    pipe = "synthetic 6-mic 16 kHz 4 s, full MISO1x6 -> align -> MVDRx2 -> MISO3x2"
    if world == 1:
        return f"BASELINE configs[3]: {pipe}" + ("" if B == 16 else f" (batch {B} instead of 16)")
    if world == 8 and B == 16:
        return f"BASELINE configs[4]: {pipe}, batch 128 sharded over 8 GPUs (8 x 16), no data-path collective"
    return (f"BASELINE configs[4] sharding at {world} ranks: {pipe}, global batch {world * B} = {world} x {B} "
            f"(configs[4] itself is 8 x 16), no data-path collective")


def rank_utterances(rank, world, B):
    """Global utterance indices of this rank's shard: the contiguous block split of the global batch world * B
    (misonet_amd.pipeline.shard_range; the reference's utterances are independent, dataloader/data.py:558-595).
    At 8 x 16 rank r owns utterances [16 r, 16 r + 16) of BASELINE configs[4]'s batch of 128."""
    from misonet_amd.pipeline import shard_range
    lo, hi = shard_range(world * B, rank, world)
    return list(range(lo, hi))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(n):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script on this node."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU per step (BASELINE configs[3]: 16)")
    ap.add_argument("--frames", type=int, default=1001)
    ap.add_argument("--precision", choices=sorted(MODES) + ["auto"], default=HEADLINE_PRECISION,
                    help="arithmetic of the 3x3 convs (default: the fp32-faithful headline mode)")
    ap.add_argument("--no-alt", action="store_true", help="skip the runs of the other precision modes (N = 1 only)")
    ap.add_argument("--alt", default="f32,f32w,bf16x6",
                    help="comma-separated precision modes timed beside the headline (alt_precision on the line); the default is "
                         "the product library's other two modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wav", action="store_true",
                    help="time the reference's real unit of work, wav in -> int16 wav out (HIP STFT front-end, pipeline, batched "
                         "iSTFT, x 32767 -> int16), device-resident and host-resident with overlapped copies; runs by default "
                         "at N = 1 unless --no-alt, this flag forces it")
    ap.add_argument("--no-profile", action="store_true", help="skip the event-instrumented pass (no roofline object)")
    ap.add_argument("--pmc", action="store_true",
                    help="force the live rocprofv3 --pmc passes (they run by default at N = 1 unless --no-alt / --no-pmc)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not re-execute under rocprofv3 --pmc for the live counters of the roofline object (traffic, matrix-"
                         "pipe busy fraction, observed clock): ~2 min; the fields then come from profiles/*_traffic.json")
    ap.add_argument("--verify-gather", action="store_true",
                    help="after the timed loop: every rank repeats its pass (bit-identical?), all_gather of the enhanced "
                         "spectrograms over the process group (RCCL on GPUs; 33 MB per rank at batch 16, SURVEY.md 8(e)), shard "
                         "checksums, and rank 0 checks one utterance that came from the LAST rank against the CPU oracle "
                         "(~3 s at 16 threads): parity / gather_parity / shard_checksums_match / second_pass_bit_identical on "
                         "the line.  ON BY DEFAULT when N > 1 (a multi-GPU line without correctness evidence is not a result)")
    ap.add_argument("--no-verify-gather", action="store_true", help="N > 1: skip the verification leg")
    args = ap.parse_args()
    if args.gpus > 1 and not args.no_verify_gather:
        args.verify_gather = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch_under_torchrun(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
    auto = args.precision == "auto"
    if auto:
        args.precision = HEADLINE_CANDIDATES[0]      # (until the calibration below has picked)
    m1.set_precision(args.precision)
    m3.set_precision(args.precision)
    enh = mz.Enhancer(m1.eval(), m3.eval(), num_spks=N_SPK, ref_ch=0)

    # ---- synthetic inputs (SURVEY.md 8(d) config 2-5 generator), global utterance index = rank*B + i ----
    B, T = args.batch, args.frames
    n = (T - 1) * 64
    mixes, cleans = [], []
    my_utts = rank_utterances(rank, world, B)
    for u in my_utts:
        obs, s0, s1 = W.synthetic_utterance(u, n)
        # the product's own front end (csrc/stft.hip), not torch.stft: rocFFT's run-time compiled kernels gave ONE of eight
        # processes that started together on one GPU a wrong spectrogram (1e-2) in a quarter of the runs (round 5: the 8-rank test)
        mixes.append(stft.stft_hip(torch.from_numpy(obs[None].copy()).to(dev))[0])           # [M,T,F]
        cleans.append(stft.stft_hip(torch.from_numpy(np.stack([s0[:, 0], s1[:, 0]], axis=1)[None].copy()).to(dev))[0])
    mix = torch.stack(mixes).contiguous()
    clean = torch.stack(cleans).contiguous()
    out = torch.empty((B, N_SPK, T, 129), dtype=torch.complex64, device=dev)

    L = _lib.lib()
    # ---- headline mode by calibration on this box (un-timed; see HEADLINE_PRECISION) ----
    selection = None
    if auto:
        cal = {m: [] for m in HEADLINE_CANDIDATES}
        for rnd in range(2):                             # alternating, the best of two short runs per mode
            for m in HEADLINE_CANDIDATES:
                m1.set_precision(m)
                m3.set_precision(m)
                dtc, _, _ = run_steps(enh, mix, clean, out, 3, 1, dist, L, _lib, False)
                cal[m].append(dtc / 3)
        best = torch.tensor([min(cal[m]) for m in HEADLINE_CANDIDATES], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(best, op=dist.ReduceOp.MAX)  # every rank picks from the same numbers
        t_x6, t_w = float(best[0].item()), float(best[1].item())
        args.precision = "f32w" if t_w <= t_x6 * (1.0 + HEADLINE_TIE) else "bf16x6"
        selection = {"rule": "f32w (literal float32) when its calibration step time is <= (1 + %g) x bf16x6's on this box, else bf16x6" % HEADLINE_TIE,
                     "calibration_ms_per_step": {"bf16x6": round(t_x6 * 1e3, 2), "f32w": round(t_w * 1e3, 2)},
                     "calibration": "best of 2 alternating runs of 3 steps (1 warm-up each), max over ranks", "picked": args.precision}
        m1.set_precision(args.precision)
        m3.set_precision(args.precision)
    # ---- the headline: un-instrumented timed loop ----
    dt, _, _ = run_steps(enh, mix, clean, out, args.steps, args.warmup, dist, L, _lib, False)
    if not os.environ.get("MISONET_BENCH_NOCHECK"):      # (timing experiments with deliberately wrong results)
        _lib.check(L.misonet_pipeline_check(enh._pipe, enh.workspace(B, T).data_ptr(), _lib.stream_ptr(dev)))
    dt_rank = dt
    per_rank = [B * args.steps / dt]
    rccl_ranks = 1
    rank_ranges = [[my_utts[0], my_utts[-1] + 1]]
    if dist is not None:
        rr = torch.tensor([my_utts[0], my_utts[-1] + 1], dtype=torch.int64, device=dev)
        allr = [torch.zeros_like(rr) for _ in range(world)]
        dist.all_gather(allr, rr)
        rank_ranges = [[int(x[0]), int(x[1])] for x in allr]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [B * args.steps / float(x.item()) for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rccl_ranks = dist.get_world_size()
    # ---- optional: the one collective of the path (result gather, outside the timed loop), verified by the oracle ----
    gather = None
    if args.verify_gather:
        from misonet_amd.pipeline import gather_outputs
        enh.enhance(mix, clean, check_nan=False, out=out)
        torch.cuda.synchronize()
        # run-to-run repeatability on every rank (the ranks of a one-device run share the CUs: a latent race would show here)
        first = out.clone()
        enh.enhance(mix, clean, check_nan=False, out=out)
        torch.cuda.synchronize()
        rep_equal = bool(torch.equal(first, out))
        if dist is not None:
            dist.barrier()
        tg = time.perf_counter()
        allout = gather_outputs(out, world * B) if dist is not None else out
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        # the gather moved what every rank computed: a float64 checksum of each rank's own result, gathered as scalars, against the
        # checksum of its shard of the gathered tensor
        sums_ok = None
        if dist is not None:
            mine = torch.view_as_real(out).double().abs().sum().reshape(1)
            alls = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(alls, mine)
            got_s = [float(torch.view_as_real(allout[r * B:(r + 1) * B]).double().abs().sum()) for r in range(world)]
            sums_ok = [bool(float(a.item()) == g) for a, g in zip(alls, got_s)]
            if os.environ.get("MISONET_BENCH_DEBUG_INPUTS"):      # (debugging aid: is every rank's device STFT of its first utterance right?)
                from oracle import pipeline_oracle as _po
                obs_, _s0, _s1 = W.synthetic_utterance(my_utts[0], n)
                mref = _po.stft_chunk(obs_)
                e_in = float(np.linalg.norm(mix[0].cpu().numpy() - mref) / np.linalg.norm(mref))
                rep_equal = rep_equal and e_in < 1e-4
                print(f"[rank {rank}] input STFT rel err {e_in:.3e}", file=sys.stderr, flush=True)
            flag = torch.tensor([1.0 if rep_equal else 0.0], dtype=torch.float64, device=dev)
            allf = [torch.zeros_like(flag) for _ in range(world)]
            dist.all_gather(allf, flag)
            rep_equal = [bool(f.item() == 1.0) for f in allf]
        if rank == 0:
            from oracle import pipeline_oracle
            u = (world - 1) * B                              # first utterance of the LAST rank's shard
            obs, s0, s1 = W.synthetic_utterance(u, n)
            mx = pipeline_oracle.stft_chunk(obs)
            cl = np.stack([pipeline_oracle.stft_chunk(s0)[0], pipeline_oracle.stft_chunk(s1)[0]])
            ref = pipeline_oracle.enhance_utterance(mx, cl, sd1, sd3, ref_ch=0)["out"]
            got = allout[u].cpu().numpy()
            err = float(np.linalg.norm(np.abs(got) - np.abs(ref)) / np.linalg.norm(np.abs(ref)))
            gather = {"gather_ms": round(gather_ms, 3), "gathered_shape": list(allout.shape),
                      "bytes_per_rank": int(out.numel() * 8), "shard_checksums_match": sums_ok, "second_pass_bit_identical": rep_equal,
                      "gather_parity": {"utterance": u, "from_rank": world - 1,
                                        "rel_l2_magnitudes_vs_oracle": float(f"{err:.3e}"), "tolerance": 1e-3,
                                        "ok": bool(np.isfinite(err) and err < 1e-3)}}
    # ---- roofline leg: the same steps once more with HIP events around every launch ----
    prof = None
    if not args.no_profile:
        kp = max(1, min(args.steps, 5))
        dtp, ms, cnt = run_steps(enh, mix, clean, out, kp, 1, dist, L, _lib, True)
        prof = (kp, dtp, ms, cnt)

    if rank == 0:
        utt = world * B * args.steps
        value = utt / dt
        roof, roof2 = None, None
        if prof and prof[3][0] > 0:
            kp, dtp, ms, cnt = prof
            live = None
            if world == 1 and not args.no_pmc and (args.pmc or not args.no_alt):   # (--no-alt = the quick A/B form of this script)
                live = pmc_live(args.precision, B, T)
            roof, roof2 = roofline_objects(args.precision, B, T, kp, ms[0], cnt[0], live)
            roof["time_share"] = {"conv_ms_per_step": round(ms[0] / kp, 2), "tcn_ms_per_step": round(ms[1] / kp, 2),
                                  "mvdr_ms_per_step": round(ms[2] / kp, 2), "other_ms_per_step": round(ms[3] / kp, 2)}
            roof["instrumented_steps"] = kp
            roof["instrumented_ms_per_step"] = round(dtp / kp * 1e3, 2)
            roof["hbm_frac_pipeline"] = round(ALGO_BYTES_PER_UTT * (value / world) / (PEAK_HBM_TBS * 1e12), 4)
        alts = []
        if world == 1 and not args.no_alt:
            for other in [m for m in args.alt.split(",") if m]:
                if other == args.precision or other not in MODES:
                    continue
                try:
                    m1.set_precision(other)
                    m3.set_precision(other)
                except ValueError:
                    continue
                dt2, _, _ = run_steps(enh, mix, clean, out, args.steps, args.warmup, None, L, _lib, False)
                a = {"dtype": other, "fp32_faithful": MODES[other][2], "operand_bits": MODES[other][3],
                     "value": round(B * args.steps / dt2, 3),
                     "unit": "utt/s", "steps": args.steps, "warmup": args.warmup}
                if not args.no_profile:
                    k = max(1, min(args.steps, 3))
                    _, ms2, cnt2 = run_steps(enh, mix, clean, out, k, 0, None, L, _lib, True)
                    # the two modes whose arithmetic is literally float32 get their own LIVE counters
                    # live counters for the OTHER fast mode (f32w or bf16x6: whichever is not the headline); the exact-f32 mode's
                    # come from the committed profile (its kernels did not change; three PMC triples made the default run 90 s)
                    live2 = pmc_live(other, B, T) if (other in ("f32w", "bf16x6") and not args.no_pmc) else None
                    a["roofline"] = roofline_objects(other, B, T, k, ms2[0], cnt2[0], live2)[0]
                    a["roofline"]["time_share"] = {"conv_ms_per_step": round(ms2[0] / k, 2), "tcn_ms_per_step": round(ms2[1] / k, 2),
                                                   "mvdr_ms_per_step": round(ms2[2] / k, 2), "other_ms_per_step": round(ms2[3] / k, 2)}
                alts.append(a)
            m1.set_precision(args.precision)
            m3.set_precision(args.precision)
            # The literal-float32 figures once more inside the roofline object (they are also alt_precision entries; the driver's
            # parsed record keeps neither whole -- the stdout tail does): "exact_f32" = every
            # conv an fmaf chain on v_mfma_f32_32x32x2_f32 (the reference's own arithmetic, model.py:77-80, 401-416);
            # "winograd_f32" = the same matrix cores with the DenseBlock convs in Winograd F(2x2, 3x3) form.
            if roof is not None:
                for key, mode in (("exact_f32", "f32"), ("winograd_f32", "f32w")):
                    src = [a for a in alts if a["dtype"] == mode and "roofline" in a]
                    if mode == args.precision:
                        src = [{"value": round(value, 3), "steps": args.steps, "roofline": roof}]
                    if src:
                        r2 = src[0]["roofline"]
                        roof[key] = {"dtype": mode, "value": src[0]["value"], "unit": "utt/s",
                                     "ms_per_step": round(B / src[0]["value"] * 1e3, 2),
                                     **{kk: r2.get(kk) for kk in ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "traffic",
                                                                  "traffic_over_layout_bytes", "mfma_busy_frac_pmc",
                                                                  "clock_ghz_observed_pmc", "pmc_fields_measured_live",
                                                                  "pmc_source_sha16", "time_share", "peak_basis",
                                                                  "issued_tflops", "frac_of_direct_peak")
                                        if r2.get(kk) is not None}}
        # latency of ONE utterance (B = 1: the reference harness' own batch size, tester.py:846-975) in the headline mode
        b1 = None
        if world == 1 and not args.no_alt:
            o1 = torch.empty((1, N_SPK, T, 129), dtype=torch.complex64, device=dev)
            for _ in range(3):
                enh.enhance(mix[:1], clean[:1], check_nan=False, out=o1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                enh.enhance(mix[:1], clean[:1], check_nan=False, out=o1)
            torch.cuda.synchronize()
            b1 = {"batch": 1, "ms_per_utterance": round((time.perf_counter() - t1) / 10 * 1e3, 3),
                  "note": "MISO1 runs 6 samples, MISO3 2: frame-tile columns instead of samples are dealt to the 8 XCDs"}
            try:                                                        # the same pass replayed from a HIP graph
                m1c, c1c = mix[:1].clone(), clean[:1].clone()
                gr, og = enh.capture_graph(m1c, c1c)
                for _ in range(3):
                    gr.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    gr.replay()
                torch.cuda.synchronize()
                b1["ms_per_utterance_hip_graph"] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
                b1["graph_equals_eager"] = bool(torch.equal(og, o1))
                del gr
            except Exception as e:                                      # a graph is an optimisation, never a requirement
                b1["hip_graph_error"] = str(e)[:200]
            # a RECORDING at the reference harness' operating point (B = 1 loader): 10 s = three 4 s splits
            # (dataloader/data.py:558-595).  The reference runs them one pass per split (tester.py:860); Enhancer.inference
            # runs the splits of a loader item as one batch (bit-identical by batch invariance).
            try:
                o3 = torch.empty((3, N_SPK, T, 129), dtype=torch.complex64, device=dev)
                def rec_batched():
                    enh.enhance(mix[:3], clean[:3], check_nan=False, out=o3)
                def rec_serial():
                    for k in range(3):
                        enh.enhance(mix[k:k + 1], clean[k:k + 1], check_nan=False, out=o1)
                res = {}
                for name, fn in (("split_by_split", rec_serial), ("splits_as_one_batch", rec_batched)):
                    for _ in range(2):
                        fn()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(8):
                        fn()
                    torch.cuda.synchronize()
                    res[name] = round((time.perf_counter() - t1) / 8 * 1e3, 3)
                b1["recording_3_splits_ms"] = res
            except Exception as e:
                b1["recording_3_splits_error"] = str(e)[:200]
            enh.enhance(mix, clean, check_nan=False, out=out)            # restore the batch workspace / result
            torch.cuda.synchronize()
        wavp, wav_pcm = None, None
        if world == 1 and (args.wav or not args.no_alt) and T == 1001:
            try:
                wavp, wav_pcm = wav_leg(enh, W, rank, B, n, args.steps, args.warmup, dev, value)
            except Exception as e:                                       # the extra leg never costs the headline its line
                wavp = {"error": repr(e)[:300]}
            enh.enhance(mix, clean, check_nan=False, out=out)
            torch.cuda.synchronize()
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            enh.enhance(mix, clean, check_nan=False, out=out)            # the headline mode's result for the checker
            torch.cuda.synchronize()
            cpu = cpu_baseline(sd1, sd3, T, out[:4].cpu().numpy(),
                               wav_pcm[:4].cpu().numpy() if wav_pcm is not None else None)
        line = {
            "metric": "utterances/sec MISO1->MVDR->MISO3, 6-mic 16kHz 4s",
            "value": round(value, 3), "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "headline_selection": selection, "fp32_faithful": MODES[args.precision][2],
            "operand_bits": MODES[args.precision][3], "data": "synthetic",
            "config": {"workload": workload_name(world, B),
                       "batch_per_gpu": B, "global_batch": world * B, "frames": T, "freq_bins": 129,
                       "parallelism": f"utterance-shard x{world}"},
            "realtime_factor": round(value * (n / 16000.0), 2), "source_sha16": _source_sha16(),
            # every MISONET_* variable of this process: {} in a clean run.  The product library reads none of them (the kernel
            # experiment switches exist only in `make exp`'s libmisonet_hip_exp.so, selected through MISONET_LIB_PATH -- which
            # would show here, as would the harness hooks MISONET_BENCH_* and MISONET_PRECISION)
            "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MISONET_")},
            "library": os.path.basename(_lib.LIB_PATH),
            "rccl_ranks": rccl_ranks, "per_rank_utt_per_s": [round(v, 3) for v in per_rank],
            "per_rank_utterances": rank_ranges, "backend": (backend if world > 1 else None),
            # N = 1: the oracle's utterances 0-3 against the timed batch (cpu_baseline's by-product); N > 1: the verification leg's
            # oracle check of an utterance computed by the LAST rank
            "parity": ((cpu or {}).pop("parity_of_headline", None) if cpu else
                       ({"rel_l2_magnitudes_vs_oracle_by_utterance": {str(gather["gather_parity"]["utterance"]): gather["gather_parity"]["rel_l2_magnitudes_vs_oracle"]},
                         "worst": gather["gather_parity"]["rel_l2_magnitudes_vs_oracle"], "tolerance": 1e-3,
                         "checked_rank": gather["gather_parity"]["from_rank"]} if gather else None)),
            "roofline": roof, "roofline_other": roof2, "alt_precision": alts, "single_utterance": b1, "wav_path": wavp,
            "cpu_baseline": cpu,
        }
        if gather:
            line.update(gather)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _physical_cores():
    """physical cores of this host (unique (package, core) pairs), logical CPUs if /proc/cpuinfo does not say"""
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(sd1, sd3, T, gpu_out=None, gpu_pcm=None):
    """The oracle (kind "port": our stock-torch-CPU/NumPy restatement of the reference path, B = 1 per call as in
    tester.py:846-975) on this host's cores.  Bounded sample: one forward warm-up, one
    utterance at each of a ladder of thread counts (8 ... physical cores), then 3 different utterances end to end at
    the fastest count (median reported, stages timed separately).  ``cores`` = the host's physical cores, ``threads`` = the
    thread count of the quoted value (B = 1 convolutions get SLOWER beyond 16-32 threads on the 2-socket boxes)."""
    from misonet_amd import weights as W
    from oracle import pipeline_oracle, miso_oracle
    logical = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    physical = _physical_cores()
    threads = max(1, min(usable, physical))
    n = (T - 1) * 64

    def utt(u):
        obs, s0, s1 = W.synthetic_utterance(u, n)
        mix = pipeline_oracle.stft_chunk(obs)
        clean = np.stack([pipeline_oracle.stft_chunk(s0)[0], pipeline_oracle.stft_chunk(s1)[0]])
        return mix, clean

    ref_out = {}                                     # utterance -> the oracle's enhanced spectrograms (the checker)

    def timed(u, stages=None):
        mix, clean = utt(u)
        t0 = time.perf_counter()
        r = pipeline_oracle.enhance_utterance(mix, clean, sd1, sd3, ref_ch=0, timings=stages)
        dt = time.perf_counter() - t0
        ref_out[u] = r["out"]
        return dt

    # B = 1 convolutions do not scale to a whole 2-socket host (128 threads were 5x SLOWER than 8 on the EPYC 9575F box):
    # probe a ladder of thread counts on one utterance each and quote the best one, stating every probe
    mix0, _ = utt(0)
    torch.set_num_threads(min(8, threads))
    miso_oracle.miso1_forward(torch.from_numpy(mix0[None]), sd1)          # warm-up
    ladder = sorted({c for c in (8, 16, 32) if c <= threads})          # (64 / 128 threads were 2-7 x slower on every box seen: not probed)
    probes, t_all = {}, time.perf_counter()
    for c in ladder:
        torch.set_num_threads(c)
        probes[c] = timed(0)
        if time.perf_counter() - t_all > 40.0:           # bounded: a probe costs 3-10 s; "best" = best of the probes listed
            break
    best = min(probes, key=probes.get)
    torch.set_num_threads(best)
    times, stages = [], []
    for u in (1, 2, 3):
        st = {}
        times.append(timed(u, st))
        stages.append(st)
        if time.perf_counter() - t_all > 50.0:
            break
    med = float(np.median(times))
    stage_med = {k: round(float(np.median([s[k] for s in stages if k in s])), 3) for k in stages[0]}
    parity = None
    if gpu_out is not None:
        # the utterances the oracle just processed are utterances 0..3 of the GPU batch: the oracle as the CHECKER of the
        # number on this line (rel-L2 of the complex-spectrogram magnitudes, the north_star's parity metric)
        errs = {}
        for u, ref in ref_out.items():
            if u < gpu_out.shape[0]:
                g = np.abs(gpu_out[u])
                errs[str(u)] = float(np.linalg.norm(g - np.abs(ref)) / np.linalg.norm(np.abs(ref)))
        parity = {"rel_l2_magnitudes_vs_oracle_by_utterance": {k: float(f"{v:.3e}") for k, v in errs.items()},
                  "worst": float(f"{max(errs.values()):.3e}") if errs else None, "tolerance": 1e-3}
        if gpu_pcm is not None:
            # the wav leg's int16 output against the oracle's own iSTFT -> x 32767 -> int16 of ITS spectrograms (tester.py:
            # 949-952): the two paths differ by the STFT front-end (HIP vs SciPy) and everything after it
            lsb = {}
            for u, ref in ref_out.items():
                if u < gpu_pcm.shape[0]:
                    want = np.stack([pipeline_oracle.istft_int16(ref[s]) for s in range(ref.shape[0])])
                    lsb[str(u)] = int(np.abs(gpu_pcm[u].astype(np.int32) - want.astype(np.int32)).max())
            parity["wav_int16_max_abs_diff_lsb_by_utterance"] = lsb
    return {"parity_of_headline": parity, "value": round(1.0 / med, 4), "unit": "utt/s", "cores": physical, "threads": best, "kind": "port",
            "host_logical_cpus": logical, "host_physical_cores": physical, "usable_cpus": usable,
            "utt_per_s_by_threads": {str(c): round(1.0 / t, 4) for c, t in probes.items()},
            "stage_seconds_median": stage_med,
            "sample": f"median of {len(times)} utterance(s) of the same synthetic workload (T={T}), B=1 per call "
                      f"(reference semantics), at the best of the probed thread counts ({best}); "
                      f"{sum(times) + sum(probes.values()):.1f} s of CPU work in all"}


if __name__ == "__main__":
    main()
