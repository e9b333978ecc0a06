"""Builds libmpc_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmpc_amd.so")
SOURCES = ["mpc_solver.hip", "mpc_generic.hip", "mpc_lanes.hip", "mpc_nl.hip", "mpc_aux.hip", "mpc_abi.hip"]
# second builds of a source with other macros: (source, object name, flags).  mpc_lanes.hip without the longitudinal
# position in its recursions (see LANE_X0 there).
VARIANTS = [("mpc_lanes.hip", "mpc_lanes_nopx.o", ["-DLANE_X0=1"])]
HEADERS = ["mpc_device.hpp", "mpc_wave.hpp", "mpc_ipm.hpp", "mpc_aset.hpp", "mpc_kernels.h", os.path.join("..", "..", "include", "mpc_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# Machine LICM hoists the literal constants of the inlined math routines out of the in-kernel step loop and then
# spills them to scratch; rematerialising them in place keeps the per-step hot path free of scratch traffic.
EXTRA_FLAGS = {"mpc_solver.hip": ["-mllvm", "-disable-machine-licm"], "mpc_nl.hip": ["-mllvm", "-disable-machine-licm"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmpc_amd.so cannot be built")
    return exe


DEV_MARK = os.path.join(LIB_DIR, "DEVELOPER_BUILD")      # left by a build with MPC_DEV_* set: the next plain build redoes it


def _dev_env() -> bool:
    return bool(os.environ.get("MPC_DEV_HORIZONS") or os.environ.get("MPC_DEV_DEFINES"))


def developer_build_left_behind() -> bool:
    """The in-tree library was built with MPC_DEV_* set and this process runs without them."""
    return os.path.exists(LIB_PATH) and os.path.exists(DEV_MARK) and not _dev_env()


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if os.path.exists(DEV_MARK) and not _dev_env():
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cc = _hipcc()

    # developer builds: MPC_DEV_HORIZONS="50" instantiates the register-resident kernels for that horizon only (a full
    # build takes minutes); MPC_DEV_DEFINES="-DNAME=VALUE ..." adds macros.  Never set for a build that is shipped.
    dev = []
    if os.environ.get("MPC_DEV_HORIZONS"):
        hs = " ".join(f"X({int(n)})" for n in os.environ["MPC_DEV_HORIZONS"].split(","))
        dev.append(f"-DMPC_HORIZONS(X)={hs}")
    dev += os.environ.get("MPC_DEV_DEFINES", "").split()

    def compile_one(job):
        src, objname, extra = job
        obj = os.path.join(objdir, objname)
        cmd = [cc] + FLAGS + EXTRA_FLAGS.get(src, []) + extra + dev + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    jobs = [(s, s.replace(".hip", ".o"), []) for s in SOURCES] + VARIANTS
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(compile_one, jobs))
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    if _dev_env():
        open(DEV_MARK, "w").write("built with MPC_DEV_* set: not every horizon is instantiated\n")
    elif os.path.exists(DEV_MARK):
        os.remove(DEV_MARK)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
