"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every symbol include/mpc_amd.h declares
(no compute call is made without a GPU); the host set-up arithmetic agrees with the checker; the product fails
loudly without a GPU and never touches the checker."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from mpc_amd import _build
    return _build.build_library()


def _declared():
    text = open(os.path.join(ROOT, "include", "mpc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib_path):
    names = _declared()
    assert len(names) >= 17
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/mpc_amd.h but not exported"


def test_ctypes_binding_covers_the_header(lib_path):
    from mpc_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()
    _lib.load()                                   # sets argtypes on every symbol; raises if one is missing


def test_ctypes_signatures_match_the_header():
    """Every declaration of include/mpc_amd.h against the argument list its ctypes binding sets: number of arguments and
    the kind of each (pointer, double, int), and the return type."""
    from mpc_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mpc_amd.h")).read(), flags=re.S)
    decls = re.findall(r"\b(int|void|const char \*)\s*(mpc_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    assert len(decls) == len(_lib._SIGS)

    def kind_c(arg):
        return "ptr" if "*" in arg else "double" if re.search(r"\bdouble\b", arg) else "int"

    def kind_py(t):
        if t is ctypes.c_double:
            return "double"
        if t is ctypes.c_int:
            return "int"
        return "ptr"

    for ret, name, args in decls:
        args = [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]
        argtypes, restype = _lib._SIGS[name]
        assert [kind_c(a) for a in args] == [kind_py(t) for t in argtypes], name
        assert {"int": ctypes.c_int, "void": None, "const char *": ctypes.c_char_p}[ret] is restype, name


def test_config_struct_matches_header_layout(tmp_path):
    """The ctypes mirror against the C compiler's layout of struct mpc_config in include/mpc_amd.h, field by field."""
    import subprocess
    from mpc_amd import _lib
    c = _lib.MpcConfig
    names = [f[0] for f in c._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mpc_amd.h"', 'int main(void) {',
             '  printf("%zu\\n", sizeof(mpc_config));']
    lines += [f'  printf("%zu\\n", offsetof(mpc_config, {n}));' for n in names]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert ctypes.sizeof(c) == out[0]
    assert [getattr(c, n).offset for n in names] == out[1:]


@pytest.mark.parametrize("kind,N,Tf", [("L", 50, 0.5), ("OFL", 50, 1.0), ("LPV", 80, 0.8), ("L", 10, 0.5)])
def test_host_setup_matches_checker(orc64, kind, N, Tf):
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp, LPVOcp, OFLOcp
    cls = {"L": LOcp, "OFL": OFLOcp, "LPV": LPVOcp}[kind]
    ctl = cls(N, Tf, _host_only=True)
    cfg = ctl._make_config()
    ref = O.make_cfg(kind, N, Tf)
    assert (cfg.kind, cfg.N) == (int(ref[0]), int(ref[1]))
    np.testing.assert_allclose(cfg.dt, ref[2], rtol=0, atol=0)
    np.testing.assert_array_equal(np.array(cfg.mp[:]), ref[3:9])           # tyre stiffness bit-exact with np.interp
    assert (cfg.max_steering, cfg.max_rate) == (ref[9], ref[10])
    np.testing.assert_array_equal(np.array(cfg.w[:]), ref[11:18])
    np.testing.assert_allclose(np.array(cfg.We[:]), ref[18:54], rtol=1e-12, atol=1e-9)
    if kind != "LPV":
        assert ctl.P.shape == (5, 5) and ctl.K.shape == (1, 5)
        assert ctl.A_stability.dtype == np.float32 and ctl.A_stability.shape == (5, 5)


def test_terminal_set_known_answer_through_host_model(golden):
    """terminal_set.m:4-10 of the reference: float32 forward-Euler (A, B) at vx = 15, dt = 0.02."""
    from mpc_amd.controllers import LOcp
    ctl = LOcp(50, 1.0, _host_only=True)          # dt = 0.02
    g = golden["kat_terminal_set"]
    np.testing.assert_allclose(ctl.A_stability, g["A"], rtol=2e-7, atol=5e-9)
    np.testing.assert_allclose(ctl.B_stability[:, 0], g["B"], rtol=1e-7, atol=0)   # float32 0.02 vs the printed 0.02
    assert (ctl.Cf, ctl.Cr) == (18879.96006089543, 24292.066639693276)


def test_product_needs_a_gpu_and_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mpc_amd.controllers import LOcp
    from mpc_amd.continuous_dynamics import Dynamics
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LOcp(50, 0.5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Dynamics(0.01)


def test_product_never_references_the_checker():
    """The CPU checker is test infrastructure: the package, the public header and the measurement tools never name it
    (the studies that use it live under tests/)."""
    bad = []
    for top in ("mpc_amd", "include", "tools"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".sh")):
                    if re.search(r"\boracle\b", open(os.path.join(d, f), errors="ignore").read()):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_parameter_files_read_like_the_reference(tmp_path, monkeypatch):
    """parameters_*.yaml is looked up in the working directory first (LMPC2.py:31), then in the package."""
    import yaml
    from mpc_amd.controllers import LOcp, read_parameters
    base = read_parameters("parameters_L.yaml")
    assert base["controller"]["r"] == 10.0 and base["model"]["m"] == 180
    mod = yaml.safe_load(yaml.safe_dump(base)); mod["controller"]["r"] = 3.0; mod["controller"]["beta"] = 2
    (tmp_path / "parameters_L.yaml").write_text(yaml.safe_dump(mod))
    monkeypatch.chdir(tmp_path)
    ctl = LOcp(50, 0.5, _host_only=True)
    assert ctl.W[6, 6] == 3.0
    ref = LOcp(50, 0.5, _host_only=True, params=base)
    np.testing.assert_allclose(ctl.W_e[1:6, 1:6], 2 * 0.5 * ctl.P)
    np.testing.assert_allclose(ctl.P, ref.P)      # the reference's DARE takes R' = W[4,4] = 0, so r does not enter P


def test_measurement_model_matches_reference_recording(golden):
    """Dynamics.measure_state_noiseless / measure_state_noise against values recorded from the reference
    (oracle/make_golden.py): the noisy form adds sigma times ONE scalar drawn from N(len(x), 1), from default_rng(1)
    (continuous_dynamics.py:68-72,216-220)."""
    from mpc_amd.continuous_dynamics import Dynamics
    g = golden["measurement"]
    for nx, dist in ((8, False), (10, True)):
        dyn = Dynamics(0.01, disturbance=dist, _host_only=True)
        np.testing.assert_array_equal(dyn.measurement_jacobian(), g[f"H{nx}"])
        np.testing.assert_array_equal(dyn.measurement_noises, g["sigma"])
        for i, x in enumerate(g[f"x{nx}"]):
            np.testing.assert_array_equal(dyn.measure_state_noiseless(x), g[f"y_noiseless{nx}"][i])
            np.testing.assert_allclose(dyn.measure_state_noise(x), g[f"y_noisy{nx}"][i], rtol=0, atol=1e-15)
        # the quirk itself: every channel carries the same draw, centred on the state length
        d = (g[f"y_noisy{nx}"] - g[f"y_noiseless{nx}"]) / g["sigma"]
        assert np.allclose(d, d[:, :1], rtol=0, atol=1e-12) and abs(d.mean() - nx) < 2.0
    # the tensor form used by the batched loop draws the same numbers in the same order
    a = Dynamics(0.01, True, _host_only=True); b = Dynamics(0.01, True, _host_only=True)
    lit = a.measurement_noise(3, 1, literal=True)
    ref = np.array([b.measure_state_noise(np.zeros(10)) for _ in range(3)])
    np.testing.assert_allclose(lit[:, 0], ref, rtol=0, atol=1e-15)
    assert a.measurement_noise(4, 5).shape == (4, 5, 7)


def test_discrete_flag_behaves_like_the_reference_constructors():
    """LMPC2.py:370-411 / OFLMPC2.py:389-429: with discrete=False the stability analysis pairs the 6x6 continuous A with
    the 5x5 Q and control.ctrb raises ValueError, so no L / OFL controller can be built that way; LPVMPC.py:33 stores
    the flag and never reads it."""
    from mpc_amd.controllers import LOcp, OFLOcp, LPVOcp
    for cls in (LOcp, OFLOcp):
        with pytest.raises(ValueError):
            cls(10, 0.1, discrete=False, _host_only=True)
    a = LPVOcp(10, 0.1, discrete=False, _host_only=True)
    b = LPVOcp(10, 0.1, _host_only=True)
    assert a.discrete is False and np.array_equal(a.W, b.W) and np.array_equal(a.W_e, b.W_e)


def test_create_rejects_bad_definitions_before_touching_a_device(lib_path):
    """mpc_create checks the controller definition first (no HIP call has happened when it refuses): unknown kind, layout
    or QP method, non-positive bounds and weights, horizons outside 2..128."""
    from mpc_amd import _lib
    from mpc_amd.controllers import LOcp
    lib = _lib.load()
    good = LOcp(50, 0.5, _host_only=True)._make_config()

    def refused(**changes):
        cfg = _lib.MpcConfig.from_buffer_copy(good)
        for k, v in changes.items():
            setattr(cfg, k, v)
        h = ctypes.c_void_p()
        rc = lib.mpc_create(ctypes.byref(cfg), ctypes.byref(h))
        assert not h.value
        return rc, lib.mpc_last_error(None).decode()

    for changes, word in (({"kind": 9}, "kind"), ({"N": 1}, "horizon"), ({"N": 129}, "horizon"), ({"B": 0}, "B >= 1"),
                          ({"max_rate": 0.0}, "max_rate"), ({"max_steering": -0.3}, "max_rate"), ({"solver_layout": 5}, "solver_layout"),
                          ({"qp_method": 3}, "qp_method"), ({"qp_method": -1}, "qp_method"), ({"step_ramp_length": -1.0}, "step_ramp_length")):
        rc, msg = refused(**changes)
        assert rc == -1 and word in msg, (changes, rc, msg)
