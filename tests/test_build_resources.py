"""Register budget of the hot kernels, checked at BUILD time (no GPU).

Round 3 lost 36 % of the f32 mode's speed to a commit that made `conv3x3_mfma<1|2,0,0>` spill (0 -> 74 / 241 VGPRs, measured
only on another mode).  The Makefile now compiles every object with -Rpass-analysis=kernel-resource-usage and leaves the
remarks in misonet_amd/csrc/build/*.res; this test reads them (tools/kernel_resources.py) and fails when a kernel that
carries the bench's time starts to use scratch memory.  tests/test_gpu_bench.py holds the matching per-mode floors on the
measured roofline fraction."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# every instantiation that runs in the bench's timed region in the `bf16x6` (headline) and `f32` modes, plus the TCN / MVDR /
# layout kernels of both: no spills, no scratch
HOT = [
    "mn::conv3x3_mfma<1, 0, 0, false, false>", "mn::conv3x3_mfma<2, 0, 0, false, false>",        # f32: stride-1 convs, 32 / 64 channels
    "mn::conv3x3_mfma<1, 1, 0, false, false>", "mn::conv3x3_mfma<2, 1, 0, false, false>",        # f32: stride-2 convs
    "mn::conv3x3_mfma<1, 2, 0, false, false>",                                            # f32: stride-2 transposed convs
    "mn::conv3x3_mfma<1, 0, 0, true, false>", "mn::conv3x3_mfma<1, 0, 3, true, false>",          # first layer (12 input channels)
    "mn::conv3x3_mfma<1, 0, 3, false, false>",                                            # (first layer on the f32 kernel: MISONET_X6_FIRST=0)
    "mn::conv3x3_wino_f32<0, false>", "mn::conv3x3_wino_f32<0, true>",                                                     # f32w: the DenseBlock convs (Winograd F(2x2, 3x3))
    "mn::conv3x3_few<4>", "mn::conv3x3_few<2>",                                    # the 4- / 2-channel last layer on the vector ALU (f32, f32w)
    "mn::conv3x3_mfma<1, 3, 0, false, false>", "mn::conv3x3_mfma<2, 3, 0, false, false>",        # decoder 0 on the single bottleneck row
    "mn::conv3x3_mfma<1, 1, 0, false, true>", "mn::conv3x3_mfma<1, 2, 0, false, true>",          # f32w: stride-2 convs / transposed convs, 1-D Winograd along T
    "mn::conv3x3_mfma<1, 0, 0, true, true>", "mn::conv3x3_mfma<1, 0, 0, false, true>",           # f32w: the first layers (12 / 16 input channels) in that form
    "mn::conv3x3_mfma<1, 4, 0, false, true>",                                             # f32w: encoder 6 (one output row: a channel group per wave)
    "mn::conv3x3_mfma<1, 3, 0, false, true>",                                             # f32w: decoder 0 (one input row, one tap per output row) in the 1-D form
    "mn::conv3x3_x6_first<3>", "mn::conv3x3_x6_first<4>",                          # first layer in bf16x6 (round 4)
    "mn::conv3x3_bf16x6<0, 8, false, 4, false, false, 0>", "mn::conv3x3_bf16x6<0, 8, false, 3, false, false, 0>",    # 38 % + 28 % of the bf16x6 step
    "mn::conv3x3_bf16x6<0, 8, false, 4, false, true, 0>",                             # F <= 31 layers (two statistic units)
    "mn::conv3x3_bf16x6<0, 8, false, 4, true, false, 0>",                             # the 48-channel layer (10 %)
    "mn::conv3x3_bf16x6<0, 4, false, 4, false, false, 0>", "mn::conv3x3_bf16x6<1, 4, false, 4, false, false, 0>",
    "mn::conv3x3_bf16x6<2, 8, false, 4, false, false, 0>", "mn::conv3x3_bf16x6<2, 4, false, 4, false, false, 0>",
    "mn::conv3x3_bf16x6<3, 8, false, 4, false, false, 0>", "mn::conv_wprep6_k",
    "mn::conv3x3_bf16x6<0, 4, false, 4, false, false, 1>", "mn::conv3x3_bf16x6<0, 4, false, 4, false, false, 2>",   # F = 1 pair
    "mn::tcn_pw_k<false, true>", "mn::tcn_pw_k<true, true>", "mn::tcn_pw_k<false, false>", "mn::tcn_pw_k<true, false>",
    "mn::tcn_dw_k<0>", "mn::tcn_dw_k<1>", "mn::tcn_dw_k<2>", "mn::tcn_dw_k<3>", "mn::tcn_cln_stats_k", "mn::tcn_prepare_k",
    "mn::mvdr_scm_eig<6>", "mn::mvdr_solve<6>", "mn::mvdr_apply<6>",
    "mn::pack_k", "mn::unpack_k", "mn::assemble3_k", "mn::pit_dist_k<2>", "mn::pit_pick_k<2>", "mn::stft_pack_k",
]

# known spillers that are NOT on the bench's path, with the scratch they are allowed (bytes per lane): instantiations of
# the opt-in modes / of shapes this network does not have.  A number going UP here is a regression too.
TOLERATED = {
    "mn::conv3x3_mfma<2, 0, 3, false, false>": 64, "mn::conv3x3_mfma<2, 0, 4, false, false>": 32,      # 64-channel planar-in / oct-out: unused
    "mn::conv3x3_mfma<1, 2, 3, false, false>": 28,                                              # row-pair transposed tile writing oct3: MISONET_X6_FIRST=0 runs only
    "mn::conv3x3_bf16x6<0, 8, true, 4, false, false, 0>": 224,                                     # fp16-piece hand-over layer: experiment-build mode f16x3 only
    "mn::mvdr_scm_eig<8>": 600,                                                          # M = 8 microphones (tests only)
}


@pytest.fixture(scope="module")
def table():
    subprocess.run(["make", "-C", os.path.join(ROOT, "misonet_amd", "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    import kernel_resources
    t = kernel_resources.parse()
    assert len(t) >= 60, f"only {len(t)} kernels in build/*.res: was the library built by this Makefile?"
    return t


# the one hot instantiation that is allowed scratch: the 8-row stride-2-transposed tile holds 1 spilled VGPR (8 bytes) since
# round 2 (3 % of a step); every attempt to remove it moved the allocator to 38+ spills (compiling the timeline stamps out:
# 256 VGPRs / 38 spilled).  It may not grow.
# The G16 instantiation (two chunk bodies and two epilogues in one persistent kernel) keeps 26 loop-invariant values of its
# tile set-up in scratch: they are written once per workgroup and re-loaded a few times per TILE (~150 k cycles); its two
# MFMA loops contain no scratch access (checked on the ISA, LAB.md round 4).
# conv3x3_x6_first (round 6): the weight image in flight under the patch staging costs 4-6 spilled staging temporaries at the 168
# registers of three workgroups per CU; measured faster with them (1.18 -> 1.09 ms per step) than without at two per CU (1.12).
HOT_SCRATCH_ALLOWED = {"mn::conv3x3_bf16x6<2, 8, false, 4, false, false, 0>": 8, "mn::conv3x3_bf16x6<0, 8, false, 4, true, false, 0>": 120,
                       "mn::conv3x3_x6_first<3>": 16, "mn::conv3x3_x6_first<4>": 24}


def test_hot_kernels_do_not_spill(table):
    missing = [k for k in HOT if k not in table]
    assert not missing, f"hot kernels not found in the build remarks (renamed instantiation?): {missing}"
    bad = {k: (table[k]["vgpr_spill"], table[k]["scratch"]) for k in HOT
           if table[k]["scratch"] > HOT_SCRATCH_ALLOWED.get(k, 0) or table[k]["vgpr_spill"] > HOT_SCRATCH_ALLOWED.get(k, 0) // 4}
    assert not bad, f"(VGPR spills, scratch bytes) of hot kernels: {bad}"
    # SGPR spills go to VGPR lanes (v_writelane, no memory): the persistent kernels' producer bookkeeping has 13-41 of them.
    # They are bounded here so that a jump is seen.
    sg = {k: table[k]["sgpr_spill"] for k in HOT if table[k]["sgpr_spill"] > (80 if k.endswith("4, true, false, 0>") else 56)}
    assert not sg, f"SGPR spills: {sg}"


def test_no_other_kernel_spills_unnoticed(table):
    bad = {}
    for k, r in table.items():
        if k in HOT:
            continue
        if r["scratch"] > TOLERATED.get(k, 0):
            bad[k] = (r["scratch"], TOLERATED.get(k, 0))
    assert not bad, f"scratch bytes per lane (found, allowed): {bad}"


def test_headline_kernel_occupancy(table):
    # the persistent bf16x6 kernels are written for ONE 512-thread workgroup per CU = 2 waves per SIMD: <= 256 VGPRs
    for k in ("mn::conv3x3_bf16x6<0, 8, false, 4, false, false, 0>", "mn::conv3x3_bf16x6<0, 8, false, 3, false, false, 0>",
              "mn::conv3x3_bf16x6<0, 8, false, 4, false, true, 0>"):
        assert table[k]["vgprs"] <= 256 and table[k]["occupancy"] >= 2, table[k]
    # the f32 32-channel kernel is tuned for three workgroups per CU (<= 168 VGPRs)
    assert table["mn::conv3x3_mfma<1, 0, 0, false, false>"]["vgprs"] <= 168
    assert table["mn::conv3x3_mfma<1, 0, 0, false, false>"]["occupancy"] >= 3
    # the first-layer kernel is latency-bound: three workgroups per CU (<= 168 VGPRs, 52 KB of LDS each)
    assert table["mn::conv3x3_x6_first<3>"]["vgprs"] <= 168 and table["mn::conv3x3_x6_first<3>"]["occupancy"] >= 3


def test_wino_kernel_register_files(table):
    """conv3x3_wino_f32 keeps its 256 accumulators in FIXED AGPRs behind inline asm (conv_wino.hip): one wave per SIMD, all 256
    AGPRs declared, nothing spilled -- a spilled VGPR could be parked in an AGPR between two asm statements -- and, in the
    ISA, no v_accvgpr_* instruction that the compiler generated itself (every one sits inside an asm block)."""
    for k in ("mn::conv3x3_wino_f32<0, false>", "mn::conv3x3_wino_f32<0, true>"):      # the 32-row body and the 16-row body (G16)
        r = table[k]
        assert r["agprs"] == 256 and r["vgprs"] <= 256 and r["occupancy"] == 1, r
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r
    csrc = os.path.join(ROOT, "misonet_amd", "csrc")
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S",
                          "--cuda-device-only", "conv_wino.hip", "-o", "-"], cwd=csrc, check=True, capture_output=True,
                         text=True).stdout.splitlines()
    inside, own, total, dma = False, 0, 0, 0
    for ln in asm:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            inside = True
        elif t.startswith(";;#ASMEND"):
            inside = False
        elif t.startswith("v_accvgpr"):
            total += 1
            own += inside
        elif " lds" in t and t.startswith("buffer_load"):
            dma += 1
    assert total == own and total >= 896, (total, own)          # per body: epilogue reads (256 / 128) + 256 prologue zeroing writes
    assert dma >= 2 * 6 * 10, dma                               # two instantiations x six chunk bodies x (4 U-image + 6 raw-input pieces)
    assert not any("flat_load" in ln or "flat_store" in ln or "scratch_" in ln for ln in asm)
