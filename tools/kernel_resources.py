#!/usr/bin/env python3
"""Parse the -Rpass-analysis=kernel-resource-usage remarks the Makefile leaves in misonet_amd/csrc/build/*.res.

    python tools/kernel_resources.py            # table of every kernel: VGPRs, spills, scratch, LDS, occupancy
Used by tests/test_build_resources.py (the hot instantiations must not spill)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "misonet_amd", "csrc", "build")
_FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
           "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
           "LDS Size [bytes/block]": "lds"}


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return [re.sub(r"\(.*$", "", l.replace("void ", "")).strip() for l in out.splitlines()]


def parse(build_dir=BUILD):
    """{demangled kernel name without arguments: {vgprs, vgpr_spill, scratch, lds, occupancy, file, ...}}"""
    recs = []
    for path in sorted(glob.glob(os.path.join(build_dir, "*.res"))):
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
            if not m:
                continue
            body = m.group(1)
            if body.startswith("Function Name:"):
                cur = {"mangled": body.split(":", 1)[1].strip(), "file": os.path.basename(path)[:-4] + ".hip"}
                recs.append(cur)
            elif cur is not None and ":" in body:
                k, v = body.rsplit(":", 1)
                if k.strip() in _FIELDS:
                    try:
                        cur[_FIELDS[k.strip()]] = int(v)
                    except ValueError:
                        pass
    names = _demangle([r["mangled"] for r in recs]) if recs else []
    return {n: r for n, r in zip(names, recs)}


if __name__ == "__main__":
    tab = parse(sys.argv[1] if len(sys.argv) > 1 else BUILD)
    print(f"{'kernel':64s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
    for n, r in sorted(tab.items(), key=lambda kv: (kv[1]["file"], kv[0])):
        print(f"{n[:64]:64s} {r.get('vgprs', -1):5d} {r.get('vgpr_spill', -1):5d} {r.get('scratch', -1):7d} "
              f"{r.get('lds', -1):7d} {r.get('occupancy', -1):3d}")
