#!/usr/bin/env python3
"""Turn the rocprofv3 databases of tools/profile_round.sh (gpurun_out/prof/*) into the small per-kernel summaries
kept under profiles/: calls, total, average, min, max duration and register/LDS/scratch figures per kernel, plus the
HBM counter totals per dispatch of the two --pmc passes."""
import csv
import glob
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
tag = sys.argv[3] if len(sys.argv) > 3 else "r01b"
os.makedirs(dst, exist_ok=True)
traffic = {}
# Registers and scratch come from the compiler's report on the shipped code object (tools/kernel_resources.py), matched
# by kernel name: rocprofv3's own register columns count allocation granules and came out at half those figures.
res = {}
res_csv = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"{tag}_kernel_resources.csv")
if os.path.exists(res_csv):
    for r in csv.DictReader(open(res_csv)):
        res["".join(r["kernel"].split())] = r


def resources(name):
    key = "".join(name.split())
    for k, r in res.items():
        if k == key or k.startswith(key) or key.startswith(k.split("(")[0].replace("void", "")):
            return [r["vgpr"], r["agpr"], r["sgpr"], r["scratch_bytes_per_lane"]]
    return ["", "", "", ""]


for db in sorted(glob.glob(os.path.join(src, "*", "*_results.db"))):
    name = os.path.basename(os.path.dirname(db))
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "vgpr", "agpr", "sgpr", "scratch_bytes_per_lane",
                    "lds_bytes", "grid_x", "workgroup_x"])
        for r in rows[:12]:
            iv = [int(v) if v is not None else "" for v in r[1:]]
            w.writerow([r[0][:120]] + iv[:5] + resources(r[0]) + [iv[8], iv[10], iv[11]])
    pmc = con.execute("select count(*) from pmc_events").fetchone()[0]
    if pmc:
        cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
        q = con.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                        "group by kernel_name, counter_name order by sum(value) desc").fetchall() if "kernel_name" in cols else []
        with open(os.path.join(dst, f"{tag}_{name}_counters.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "counter", "dispatches", "sum", "avg_per_dispatch"])
            for r in q[:12]:
                w.writerow([r[0][:120], r[1], int(r[2]), r[3], r[4]])
    if pmc and name.startswith("pmc"):
        # HBM bytes of the timed launch: the MPC kernel dispatches in order are [settling launch (c2 only)], warm-up,
        # timed region, ...; FETCH_SIZE / WRITE_SIZE are in KiB, and gfx950 tallies reads at half their size
        kind, wl = name[3], name[5:]
        if kind not in "fw":
            continue
        disp = con.execute("select dispatch_id, sum(value), max(duration) from counters_collection where kernel_name like '%k_mpc%' "
                           "group by dispatch_id order by dispatch_id").fetchall()
        idx = 2 if wl == "c2_driver" else 1
        if len(disp) > idx:
            traffic.setdefault(wl, {})["fetch_bytes" if kind == "f" else "write_bytes"] = disp[idx][1] * 1024 * (2 if kind == "f" else 1)
            traffic[wl]["launch_ns_under_counters"] = disp[idx][2]
    print(name, "kernels:", [(r[0][:40], r[1], round(r[3])) for r in rows[:3]], "pmc rows", pmc)

if traffic:
    import json
    keys = {"c2_driver": "c2:B=4096:steps=20:warmup=5", "c2_default": "c2:B=4096:steps=1000:warmup=50", "c3": "c3:B=8192:steps=100:warmup=10",
            "c4": "c4:B=8192:steps=300:warmup=20", "c5": "c5:B=131072:steps=50:warmup=5"}
    out = {}
    for wl, t in traffic.items():
        if "fetch_bytes" in t and "write_bytes" in t:
            out[keys.get(wl, wl)] = dict(t, hbm_bytes_per_launch=t["fetch_bytes"] + t["write_bytes"])
    with open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out)
