#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of libmpc_amd as the compiler reports them for the code object
that is shipped (hipcc -Rpass-analysis=kernel-resource-usage with the flags of mpc_amd/_build.py; no GPU needed):
  tools/kernel_resources.py [out.csv]
rocprofv3's own register columns count allocation granules differently (they came out at half the code object's
figures), so the profile summaries take these columns from here."""
import csv
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mpc_amd import _build

out = sys.argv[1] if len(sys.argv) > 1 else "profiles/kernel_resources.csv"
jobs = [(s, []) for s in _build.SOURCES if s != "mpc_abi.hip"] + [(s, f) for s, _, f in _build.VARIANTS]


def one(job):
    src, extra = job
    cmd = [_build._hipcc()] + _build.FLAGS + _build.EXTRA_FLAGS.get(src, []) + extra + ["-Rpass-analysis=kernel-resource-usage", "-c",
                                                                                       os.path.join(_build.CSRC, src), "-o", "/dev/null"]
    return src + (" " + " ".join(extra) if extra else ""), subprocess.run(cmd, capture_output=True, text=True).stderr


rows = []
with ThreadPoolExecutor(max_workers=4) as ex:
    for src, text in ex.map(one, jobs):
        cur = None
        for line in text.splitlines():
            m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "Function Name":
                cur = {"source": src, "mangled": v}
                rows.append(cur)
            elif cur is not None:
                cur[k.split(" [")[0]] = v
names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.splitlines()
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "source", "vgpr", "agpr", "sgpr", "scratch_bytes_per_lane", "static_lds_bytes", "waves_per_simd"])
    for r, n in zip(rows, names):
        w.writerow([n, r["source"], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")])
print(out, len(rows), "kernels")
