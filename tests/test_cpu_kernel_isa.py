"""Build-time guard on the generated code of the solver kernels (no GPU needed: hipcc cross-compiles).  Under heavy
register pressure hipcc 7.2 has placed register saves (copies to AGPRs) at the top of a join block, ahead of the EXEC
restore: lanes that skipped the region got stale values back (the lane kernel's iteration counters on cold-started
lanes).  tools/check_exec_prologue.py finds that pattern in the ISA; every kernel file must be free of it."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_register_saves_ahead_of_an_exec_restore(tmp_path):
    from mpc_amd import _build
    cc = _build._hipcc()

    def check(job):
        src, objname, extra = job
        out = str(tmp_path / objname.replace(".o", ".s"))
        subprocess.check_call([cc] + [f for f in _build.FLAGS if f != "-Wall"] + _build.EXTRA_FLAGS.get(src, []) + extra +
                              ["-S", "--cuda-device-only", "-o", out, os.path.join(_build.CSRC, src)], stderr=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_exec_prologue.py"), out], capture_output=True, text=True)
        return objname, r.returncode, r.stdout

    jobs = [(s, s.replace(".hip", ".o"), []) for s in ("mpc_lanes.hip", "mpc_solver.hip", "mpc_generic.hip", "mpc_nl.hip")] + _build.VARIANTS
    with ThreadPoolExecutor(max_workers=5) as ex:
        results = list(ex.map(check, jobs))
    bad = [(s, o) for s, rc, o in results if rc != 0]
    assert not bad, bad
