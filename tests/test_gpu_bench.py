"""bench.py end to end on the GPU box: the single-process line, and the world_size-2 launch path (rendezvous, barrier, MAX
reduce of the elapsed time, rank-0 JSON line) with both ranks on device 0 over gloo -- a second GPU is not needed to
exercise the script's distributed code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_process_line():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 12.5          # north_star: >= 50x real time
    assert d["config"]["workload"].startswith("BASELINE configs[3]")
    assert d["fp32_faithful"] is True, "the headline must run fp32-faithful arithmetic (reference model.py:77-80)"
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # a clean run carries no MISONET_* variable and runs the product library
    assert d["env_overrides"] == {} and d["library"] == "libmisonet_hip.so", (d["env_overrides"], d["library"])
    # the headline mode was picked on this box by the recorded rule (round 6): f32w when at least as fast as bf16x6, within 0.5 %
    hs = d["headline_selection"]
    assert hs["picked"] == d["dtype"] and d["dtype"] in ("bf16x6", "f32w"), hs
    cal = hs["calibration_ms_per_step"]
    ratio = cal["f32w"] / cal["bf16x6"]
    if abs(ratio - 1.005) > 5e-4:                 # (the line carries the times rounded to 10 us: no verdict at the threshold itself)
        assert (ratio <= 1.005) == (hs["picked"] == "f32w"), hs


def test_bench_fixed_precision_has_no_selection():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt", "--no-profile",
                        "--precision", "bf16x6"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["dtype"] == "bf16x6" and d["headline_selection"] is None


def test_bench_line_records_environment():
    env = dict(os.environ, MISONET_BENCH_NOCHECK="1", MISONET_X6_SLOTS="8")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-alt", "--no-profile"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["env_overrides"] == {"MISONET_BENCH_NOCHECK": "1", "MISONET_X6_SLOTS": "8"}
    assert d["value"] > 100.0            # ... and the product library ignored the kernel switch (it would run 64 of 256 CUs)


def test_bench_mode_floors():
    """Per-mode floor on the measured roofline fraction of the conv kernels (the bench's own instrumented pass): round 3
    shipped an f32 mode that had silently lost 36 % (88 -> 56 utt/s, frac 0.68 -> 0.43) to register spills.  Boxes differ
    by a few per cent in sustained clocks (round 4: bf16x6 0.429 ... 0.450 on seven boxes, f32 0.675 ... 0.684); the floors sit
    5 % under the slowest box seen (a false alarm on a slow box costs more than a missed 5 %) -- a spilled kernel loses 30 %."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-pmc"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    # the gate is the roofline FRACTION (what a spilled kernel loses); absolute utt/s floors only where a box's class is known
    # (MISONET_TEST_ABS_FLOORS=1: the pool's MI355X boxes; ADVICE r4: a throttled or shared box must not fail the suite)
    absf = bool(os.environ.get("MISONET_TEST_ABS_FLOORS"))
    # the line carries all three product modes: the headline + the other two under alt_precision
    got = {d["dtype"]: (d["value"], d["roofline"])}
    got.update({a["dtype"]: (a["value"], a["roofline"]) for a in d["alt_precision"]})
    assert sorted(got) == ["bf16x6", "f32", "f32w"], sorted(got)
    assert got["bf16x6"][1]["frac"] >= 0.405, got["bf16x6"][1]
    assert not absf or got["bf16x6"][0] >= 133.0, got["bf16x6"][0]
    assert got["f32"][1]["frac"] >= 0.62, got["f32"][1]
    assert not absf or got["f32"][0] >= 80.0, got["f32"][0]
    # f32w: the same matrix cores with 16 / 36 of the products on the DenseBlock layers; round 6: 142-144 utt/s, 1.6 x f32
    assert got["f32w"][1]["frac"] >= 0.47 and got["f32w"][0] >= 1.45 * got["f32"][0], got["f32w"]
    assert d["roofline"]["exact_f32"]["frac"] == got["f32"][1]["frac"]
    assert d["roofline"]["winograd_f32"]["value"] == got["f32w"][0]


def test_bench_live_pmc_fields():
    """The roofline object's counter fields come from THIS box and THIS tree: bench.py re-executes itself under
    ``rocprofv3 --pmc`` (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in separate passes)."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--alt", ""],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    rf = d["roofline"]
    assert rf["pmc_fields_measured_live"] is True, rf.get("traffic_source")
    assert rf["traffic_source"].startswith("live")
    assert 1.2e9 < rf["traffic"] < 3.5e9                                   # 1.39 GB algorithmic (fp32), 2.07 GB in the oct3 layout
    assert 0.5 < rf["mfma_busy_frac_pmc"] < 0.98
    assert 1.0 < rf["clock_ghz_observed_pmc"] <= 2.45
    if d["dtype"] == "bf16x6":                                           # (counted in 16-bit MFMA products)
        assert 0.7 < rf["useful_over_issued_mfma_pmc"] <= 1.0
    # and the wav-in -> int16-out leg is on the same line
    wp = d["wav_path"]
    assert wp["host_equals_device_result"] and wp["vs_headline"]["host_resident_overlapped"] > 0.9


def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, MISONET_BENCH_ONE_DEVICE="1", MISONET_BENCH_BACKEND="gloo")
    # plain ``python bench.py --gpus 2``: the script starts its own ranks (torch.distributed.run on 127.0.0.1)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt",
           "--batch", "4"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == 4
    assert d["rccl_ranks"] == 2 and len(d["per_rank_utt_per_s"]) == 2
    assert d["per_rank_utterances"] == [[0, 4], [4, 8]] and d["backend"] == "gloo"
    assert "configs[4]" in d["config"]["workload"] and d["config"]["global_batch"] == 8
    # every N > 1 line verifies itself WITHOUT a flag: the all_gather of the results (the path's only
    # collective), one utterance of the LAST rank's shard checked by the oracle on rank 0 -> `parity` on the line
    assert d["parity"] is not None and d["parity"]["worst"] < 1e-3 and d["parity"]["checked_rank"] == 1, d["parity"]
    assert d["gathered_shape"] == [8, 2, 1001, 129] and d["gather_ms"] > 0
    gp = d["gather_parity"]
    assert gp["from_rank"] == 1 and gp["utterance"] == 4 and gp["ok"] and gp["rel_l2_magnitudes_vs_oracle"] < 1e-3
    assert d["shard_checksums_match"] == [True, True] and d["second_pass_bit_identical"] == [True, True]
    # whole-job aggregate: 2 ranks x 4 utterances x 2 steps over the max-over-ranks time
    assert abs(d["value"] - 2 * 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-2


def test_bench_eight_ranks_on_one_device():
    """BASELINE configs[4]'s launch shape without its hardware: 8 real ranks, rendezvous on 127.0.0.1,
    barrier, MAX-reduce of the elapsed time, the result gather verified from rank 7 -- all ranks on device 0 over gloo, 2
    utterances each (the block split of dataloader/data.py:558-595's independent utterances: rank r owns [2 r, 2 r + 2))."""
    env = dict(os.environ, MISONET_BENCH_ONE_DEVICE="1", MISONET_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt",
           "--batch", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "weak" and d["backend"] == "gloo"
    assert d["per_rank_utterances"] == [[2 * r_, 2 * r_ + 2] for r_ in range(8)]
    assert "configs[4]" in d["config"]["workload"] and d["config"]["global_batch"] == 16
    assert len(d["per_rank_utt_per_s"]) == 8 and min(d["per_rank_utt_per_s"]) > 0
    gp = d["gather_parity"]
    assert d["parity"] is not None and d["parity"]["worst"] < 1e-3 and d["parity"]["checked_rank"] == 7, d["parity"]   # no --verify-gather on the command line
    assert d["roofline"] is None or "NOT this run's counters" in (d["roofline"].get("traffic_source") or "")
    assert d["gathered_shape"] == [16, 2, 1001, 129]
    assert gp["from_rank"] == 7 and gp["utterance"] == 14 and gp["ok"] and gp["rel_l2_magnitudes_vs_oracle"] < 1e-3
    # every rank's shard arrived as computed, and every rank reproduces its result bit for bit with seven other processes on the
    # same CUs (round 5: the inputs come from the product's STFT -- torch.stft's run-time compiled rocFFT kernels gave one of
    # eight simultaneously started processes a wrong spectrogram in a quarter of the runs)
    assert d["shard_checksums_match"] == [True] * 8 and d["second_pass_bit_identical"] == [True] * 8
    assert abs(d["value"] - 8 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-2


def test_bench_two_ranks_rccl():
    """The real thing when the box has two GPUs: one rank per device over RCCL (backend "nccl"), the gather verified by
    the oracle, rank 1's inputs = utterances 16-31 of the global batch.  Skipped on the 1-GPU boxes of the pool (the
    one-device gloo test above covers the script's distributed code there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MISONET_BENCH_ONE_DEVICE", "MISONET_BENCH_BACKEND"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alt"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "nccl"
    assert d["per_rank_utterances"] == [[0, 16], [16, 32]]
    assert d["gather_parity"]["ok"] and d["gather_parity"]["from_rank"] == 1 and d["gather_parity"]["utterance"] == 16
    assert d["gathered_shape"] == [32, 2, 1001, 129]
