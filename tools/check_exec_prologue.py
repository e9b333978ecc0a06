"""Scan kernel ISA for register saves (copies to AGPRs, scratch stores) placed ahead of a join block's EXEC restore
(s_or_b64 exec, exec, sN): such an instruction runs with the lanes that skipped the region still masked off, so it saves
nothing for those lanes and the matching reload hands them stale data (seen with hipcc 7.2 under heavy register
pressure: the interior-point counters of the lane kernel came back as garbage on cold-started lanes).
  tools/check_exec_prologue.py file.s        -> lists offending blocks, exit status 1 when any"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
bad = 0
kernel = None
i = 0
while i < len(lines):
    l = lines[i]
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kernel = m.group(1)
    if re.match(r"^\.LBB\d+_\d+:", l):
        label = l.split(":")[0]
        j = i + 1
        pre = []
        saved_here = set()          # EXEC masks saved inside this block: their restores pair up here, they are no join prologue
        while j < len(lines):
            t = lines[j].split(";")[0].strip()
            if re.match(r"^\.LBB\d+_\d+:", lines[j]) or re.match(r"s_c?branch|s_endpgm", t):
                break
            ms = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\])", t)
            if ms:
                saved_here.add(ms.group(1))
            mr = re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\])", t)
            if mr and mr.group(1) in saved_here:
                saved_here.discard(mr.group(1))
                j += 1
                continue
            if re.match(r"s_or_b64 exec, exec,", t):
                # register saves only: other vector code ahead of the restore is the tail of the region itself
                vec = [p for p in pre if re.match(r"v_accvgpr_write_b32|scratch_store", p)]
                if vec:
                    bad += 1
                    print(f"{kernel} {label} (asm line {i + 1}): {len(vec)} register saves before the EXEC restore, e.g. {vec[0]}")
                break
            if t:
                pre.append(t)
            j += 1
    i += 1
print("offending blocks:", bad)
sys.exit(1 if bad else 0)
