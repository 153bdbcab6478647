#!/usr/bin/env python3
"""Build-time guard of conv3x3_wino_f32's register discipline (csrc/Makefile runs it whenever conv_wino.o is rebuilt).

The kernel keeps 256 live accumulators in FIXED physical AGPRs a0..a255 that only inline asm touches (csrc/wino_regs.hpp);
the compiler knows them as clobbers, not as live state.  That is sound only while the compiler never uses an AGPR itself:
no spill (a spilled VGPR may be parked in an AGPR between two asm statements), no scratch, and not one v_accvgpr_*
instruction outside an asm block.  Another hipcc version or other flags may break any of these silently -- so the BUILD
fails, not a test that needs a GPU.  usage: check_wino_build.py <conv_wino.res> <conv_wino.hip> [hipcc flags...]"""
import re
import subprocess
import sys


def fail(msg):
    sys.stderr.write(f"check_wino_build: {msg}\n")
    sys.exit(1)


def main():
    res, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    txt = open(res).read()
    blocks = re.split(r"remark: Function Name: ", txt)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "conv3x3_wino_f32" not in name:
            continue
        seen += 1
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
        if get(r"VGPRs Spill") or get(r"ScratchSize \[bytes/lane\]"):
            fail(f"{name}: VGPR spills / scratch ({get(r'VGPRs Spill')}, {get(r'ScratchSize .bytes/lane.')} bytes): the fixed-AGPR "
                 "accumulators are no longer safe")
        if get(r"AGPRs") != 256 or get(r"Occupancy \[waves/SIMD\]") != 1:
            fail(f"{name}: expected 256 AGPRs at one wave per SIMD")
    if not seen:
        fail("no conv3x3_wino_f32 kernel in the resource remarks")
    asm = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", src, "-o", "-"], check=True,
                         capture_output=True, text=True).stdout.splitlines()
    inside, own, total = False, 0, 0
    for ln in asm:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            inside = True
        elif t.startswith(";;#ASMEND"):
            inside = False
        elif t.startswith("v_accvgpr"):
            total += 1
            own += inside
        elif "scratch_" in t or t.startswith("flat_load") or t.startswith("flat_store"):
            fail(f"scratch / flat access in the kernel: {t}")
    if total != own:
        fail(f"{total - own} compiler-generated v_accvgpr_* instruction(s): the compiler is using AGPRs next to the fixed accumulators")
    if total < 512:
        fail(f"only {total} v_accvgpr_* in asm blocks (expected 256 epilogue reads + 256 prologue writes)")
    print(f"check_wino_build: ok ({seen} kernel(s), {total} v_accvgpr_* all inside asm blocks, 0 spills)")


if __name__ == "__main__":
    main()
