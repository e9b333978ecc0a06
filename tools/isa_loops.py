"""Per-loop instruction statistics of one kernel's ISA: tools/isa_loops.py <source.hip> <kernel-name-substring>"""
import re, subprocess, sys, os
src, key = sys.argv[1], sys.argv[2]
out = "/tmp/isa_loops.s"
extra = ["-mllvm", "-disable-machine-licm"] if os.path.basename(src) in ("mpc_solver.hip", "mpc_nl.hip") else []
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only"] + extra + ["-o", out, src],
               check=True, stderr=subprocess.DEVNULL)
text = open(out).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
lines = text[start:end]
print("kernel lines", len(lines), "scratch ld/st", sum("scratch_load" in l for l in lines), sum("scratch_store" in l for l in lines))
for h in [i for i, l in enumerate(lines) if "Inner Loop Header" in l]:
    j = h
    while j > 0 and not re.match(r"^\.LBB\d+_\d+:", lines[j]): j -= 1
    label = lines[j].split(":")[0]
    e = None
    for k in range(h, min(len(lines), h + 4000)):
        if re.search(r"s_c?branch\w* " + re.escape(label) + r"\b", lines[k]): e = k
    if e is None: continue
    body = lines[h:e]
    cnt = lambda pat: sum(bool(re.search(pat, l)) for l in body)
    print(f"{label:12s} len {e - h:4d} scratch {cnt('scratch_load'):3d}/{cnt('scratch_store'):3d} global {cnt('global_load|buffer_load'):3d}/{cnt('global_store|buffer_store'):3d} "
          f"lds {cnt(r'ds_read|ds_write'):3d} fp64 {cnt(r'v_fma|v_mul_f64|v_add_f64|v_fmac'):4d} mov {cnt(r'v_mov|v_accvgpr'):3d} waits {cnt('s_waitcnt vmcnt'):2d}")
