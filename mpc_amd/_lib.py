"""ctypes binding of libmpc_amd.so (include/mpc_amd.h).  There is no CPU fallback: a missing library or
a failing call raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

KIND = {"L": 0, "OFL": 1, "LPV": 2, "NL": 3}
PLANNER_SKIDPAD, PLANNER_STEP = 0, 1
MAX_HORIZON = 128
LAYOUT = {"auto": 0, "wave": 1, "lane": 2}
QP_METHOD = {"auto": 0, "interior_point": 1, "active_set": 2}


class MpcConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("N", C.c_int), ("B", C.c_int), ("device", C.c_int),
        ("dt", C.c_double), ("mp", C.c_double * 6), ("max_steering", C.c_double), ("max_rate", C.c_double),
        ("w", C.c_double * 7), ("We", C.c_double * 36), ("eps", C.c_double), ("tol_mu", C.c_double),
        ("tol_res", C.c_double), ("max_iter", C.c_int), ("lpv_lin_prev", C.c_int),
        ("w_nl", C.c_double * 8), ("we_nl", C.c_double * 7), ("eps_nl", C.c_double), ("sqp_tol", C.c_double),
        ("sqp_max_iter", C.c_int), ("ipm_cold_start", C.c_int), ("step_ramp_length", C.c_double),
        ("solver_layout", C.c_int), ("qp_method", C.c_int),
        ("emulate_acados", C.c_int), ("reg_epsilon", C.c_double), ("sqp_kkt_tol", C.c_double),
    ]


# slots of mpc_get_stats (include/mpc_amd.h); the *_max slots combine by maximum over GPUs, the others add
STATS_LEN = 32
STATS_SLOTS = {0: "scenarios", 1: "sum_iterations", 2: "n_status_nonzero", 3: "n_constrained", 4: "sum_abs_u0", 5: "max_abs_u0",
               6: "max_complementarity", 7: "max_residual", 8: "sum_sqp_iterations", 9: "n_active_set_solves",
               10: "sum_active_set_iterations", 11: "n_active_set_too_large", 12: "n_active_set_unsettled",
               13: "sum_active_set_flops", 14: "max_table_residual", 15: "n_active_set_switched_off", 16: "sum_stage_cost", 17: "sum_lateral_error_sq", 18: "sum_heading_pair_violation",
               19: "sum_input_sq", 20: "max_bound_violation", 21: "max_heading_pair_violation",
               22: "max_predicted_heading_pair_violation", 23: "closed_loop_steps"}
STATS_MAX_SLOTS = (5, 6, 7, 14, 20, 21, 22)

_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_SIGS = {
    "mpc_create": ([C.POINTER(MpcConfig), C.POINTER(_vp)], C.c_int),
    "mpc_destroy": ([_vp], None),
    "mpc_last_error": ([_vp], C.c_char_p),
    "mpc_set_model_params": ([_vp, _vp], C.c_int),
    "mpc_set_cost_weights": ([_vp, _vp], C.c_int),
    "mpc_condense_fixed": ([_vp, _vp, C.c_double, C.c_int], C.c_int),
    "mpc_solve_only": ([_vp, _vp, C.c_int] + [_vp] * 10, C.c_int),
    "mpc_optimize_host": ([_vp, _vp, C.c_int] + [_vp] * 10, C.c_int),
    "mpc_set_lpv_lin_mode": ([_vp, C.c_int], C.c_int),
    "mpc_solve_only_noalloc": ([C.POINTER(MpcConfig), _vp, C.c_int] + [_vp] * 8, C.c_int),
    "mpc_closed_loop_step": ([_vp, _vp, C.c_int, C.c_int] + [_vp] * 12, C.c_int),
    "mpc_closed_loop_run": ([_vp, _vp, C.c_int, C.c_int, C.c_int] + [_vp] * 14, C.c_int),
    "mpc_ofl_closed_loop_run": ([_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _dp, _dp] + [_vp] * 7, C.c_int),
    "mpc_plant_rk4": ([_vp, C.c_int, C.c_int, C.c_double, _dp, _vp, _vp], C.c_int),
    "mpc_plant_f": ([_vp, C.c_int, C.c_int, _dp, _vp, _vp, _vp], C.c_int),
    "mpc_ekf_time_update": ([_vp, C.c_int, C.c_double, _dp, _dp, _vp, _vp, _vp], C.c_int),
    "mpc_ekf_measurement_update": ([_vp, C.c_int, _dp, _vp, _vp, _vp, _vp], C.c_int),
    "mpc_ekf_kalman_gain": ([_vp, C.c_int, _dp, _vp, _vp], C.c_int),
    "mpc_ekf_jacobian": ([_vp, C.c_int, C.c_double, _dp, _vp, _vp], C.c_int),
    "mpc_dlqr_run": ([_vp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _vp, C.c_double, C.c_double, _vp, _vp, _vp], C.c_int),
    "mpc_lsim_run": ([_vp, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "mpc_plan_skidpad": ([_vp, C.c_int, C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "mpc_plan_skidpad_profile": ([_vp, C.c_int, C.c_int, C.c_double] + [_vp] * 8, C.c_int),
    "mpc_plan_step": ([_vp, C.c_int, C.c_int, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "mpc_control_buffer_interp": ([_vp, C.c_int, C.c_int, C.c_double, _vp, C.c_int, C.c_int, _vp, C.c_double, _vp], C.c_int),
    "mpc_invariance_check": ([_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, _vp, _vp], C.c_int),
    "mpc_lp_maximize": ([_vp, C.c_int, C.c_int, C.c_int] + [_vp] * 6, C.c_int),
    "mpc_get_stats": ([_vp, _vp, _dp], C.c_int),
    "mpc_reset_warm_start": ([_vp, _vp], C.c_int),
    "mpc_set_terminal_set": ([_vp, C.c_int, _vp], C.c_int),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib_path() -> str:
    """The in-tree library, or the one MPC_AMD_LIBRARY names (a prebuilt libmpc_amd.so kept elsewhere, e.g. an earlier
    build for an A/B measurement)."""
    return os.environ.get("MPC_AMD_LIBRARY") or _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Loads the shared library (building it with hipcc first if it is absent and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if os.environ.get("MPC_AMD_LIBRARY"):
        if not os.path.exists(path):
            raise RuntimeError(f"MPC_AMD_LIBRARY names {path}, which does not exist")
    elif not os.path.exists(path) or _build.developer_build_left_behind():
        # absent, or a developer build (a subset of the horizons, experiment macros) that a plain run must not load
        if not build_if_missing:
            raise RuntimeError(f"{path} is missing or a developer build: run __graft_entry__.build() (needs hipcc)")
        _build.build_library(force=True)
    lib = C.CDLL(path)
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = the build is out of date
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def hptr(a: np.ndarray):
    """Host double pointer of a C-contiguous float64 array."""
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def check(rc: int, handle=None, what: str = "libmpc_amd") -> None:
    if rc != 0:
        msg = load().mpc_last_error(handle).decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
