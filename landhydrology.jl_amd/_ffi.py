"""ctypes binding of liblandhydro_hip.so (include/landhydro.h).

This is the only place the product touches native code.  There is no CPU
fallback: if the library cannot be loaded, or no HIP device is present,
construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblandhydro_hip.so")

# enums of include/landhydro.h
LH_F32, LH_F64 = 0, 1
LH_MODEL_RICHARDS, LH_MODEL_HEAT, LH_MODEL_COUPLED = 0, 1, 2
LH_BC_NONE, LH_BC_FLUX, LH_BC_DIRICHLET, LH_BC_FREE_DRAINAGE = 0, 1, 2, 3
LH_FACE_BOTTOM, LH_FACE_TOP = 0, 1
LH_COMP_ENERGY, LH_COMP_HYDROLOGY = 0, 1
LH_FACTOR_NONE, LH_FACTOR_ON = 0, 1
LH_VAR_VARTHETA_L, LH_VAR_THETA_I, LH_VAR_RHOE_INT, LH_VAR_T = 0, 1, 2, 3
LH_DIAG_K, LH_DIAG_PSI, LH_DIAG_KAPPA, LH_DIAG_T = 0, 1, 2, 3
LH_MATH_FAST, LH_MATH_LIBM = 0, 1
LH_PLACE_MOVE_INPUT = 1
LH_ENGINE_FUSED_STAGES, LH_ENGINE_COLUMN_STEPPER = 0, 1
LH_COMM_ID_BYTES = 128
LH_PC = dict(vg_n=0, vg_alpha=1, vg_theta_r=2, vg_Ksat=3, nu=4, S_s=5)
LH_OK, LH_EINVAL, LH_ENODEVICE, LH_ENOMEM, LH_EMODEL, LH_ESTATE = 0, -1, -2, -3, -4, -5


class lh_config(C.Structure):
    _fields_ = [("ncols", C.c_int64), ("nlev", C.c_int32), ("dtype", C.c_int32),
                ("zmin", C.c_double), ("zmax", C.c_double), ("model", C.c_int32),
                ("device", C.c_int32), ("stream", C.c_void_p)]


class lh_earth_params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("rho_liq", "rho_ice", "cp_l", "cp_i", "T_0", "LH_f0", "K_therm")]


class lh_soil_params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("nu", "S_s", "nu_ss_gravel", "nu_ss_om", "nu_ss_quartz", "rho_c_ds",
                 "kappa_solid", "rho_p", "kappa_sat_unfrozen", "kappa_sat_frozen", "a", "b",
                 "kappa_dry_parameter")]


class lh_vg_params(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("n", "alpha", "theta_r", "Ksat")]


class lh_atmos_forcing(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("u_atm", "theta_atm", "z_atm", "theta_scale", "rho_a_sfc", "q_atm", "z_0m", "z_0s",
                 "R_v", "R_d", "grav", "cp_d", "cp_v", "LH_v0", "T_triple", "press_triple",
                 "von_karman")]


# every symbol include/landhydro.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_DP = C.POINTER(C.c_double)
SIGNATURES = {
    "lh_create": (C.c_int, [C.POINTER(_P), C.POINTER(lh_config)]),
    "lh_destroy": (C.c_int, [_P]),
    "lh_last_error": (C.c_char_p, [_P]),
    "lh_version": (C.c_int, []),
    "lh_set_earth_params": (C.c_int, [_P, C.POINTER(lh_earth_params)]),
    "lh_set_soil_params": (C.c_int, [_P, C.POINTER(lh_soil_params)]),
    "lh_set_vg_params": (C.c_int, [_P, C.POINTER(lh_vg_params)]),
    "lh_set_percol_param": (C.c_int, [_P, C.c_int32, _DP]),
    "lh_set_conductivity_factors": (C.c_int, [_P, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                              C.c_double]),
    "lh_set_bc": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_double, _DP]),
    "lh_set_atmos_forcing": (C.c_int, [_P, C.POINTER(lh_atmos_forcing), _DP]),
    "lh_atmos_surface_fluxes": (C.c_int, [_P, C.c_int64, _DP, _DP, _DP, _DP, _DP]),
    "lh_set_bottom_sign_consistent": (C.c_int, [_P, C.c_int32]),
    "lh_set_math_mode": (C.c_int, [_P, C.c_int32]),
    "lh_set_tuning": (C.c_int, [_P, C.c_char_p]),
    "lh_state_create": (C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    "lh_state_destroy": (C.c_int, [_P, _P]),
    "lh_upload": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, C.c_int64]),
    "lh_download": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, C.c_int64]),
    "lh_download_level": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "lh_upload_profile": (C.c_int, [_P, _P, C.c_int32, _P]),
    "lh_state_fill": (C.c_int, [_P, _P, C.c_int32, C.c_double]),
    "lh_state_copy": (C.c_int, [_P, _P, _P]),
    "lh_state_device_ptr": (C.c_int, [_P, _P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64)]),
    "lh_state_release_ptr": (C.c_int, [_P, _P, C.c_int32]),
    "lh_coordinates": (C.c_int, [_P, _DP]),
    "lh_rhs": (C.c_int, [_P, C.c_double, _P, _P, _P]),
    "lh_rhs_stable_dt": (C.c_int, [_P, C.c_double, _P, _P, _P, C.c_double, _P]),
    "lh_boundary_fluxes": (C.c_int, [_P, _P, _P, C.c_double, C.c_int32, _DP, _DP]),
    "lh_diagnostics": (C.c_int, [_P, _P, _P, _P]),
    "lh_step_ssprk33": (C.c_int, [_P, _P, _P, C.c_double, C.c_double, C.c_int64, _DP]),
    "lh_step_engine": (C.c_int, [_P, C.c_int64, C.c_int32]),
    "lh_ssprk33_stage": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_double, _DP]),
    "lh_step_ssprk33_device_dt": (C.c_int, [_P, _P, _P, C.c_double, _P, _DP]),
    "lh_step_ssprk33_adaptive": (C.c_int, [_P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_int64, _P, _P]),
    "lh_tune_placement": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_uint32, C.POINTER(C.c_float),
                                    C.POINTER(C.c_float)]),
    "lh_stable_dt": (C.c_int, [_P, _P, _P, C.c_double, _DP]),
    "lh_stable_dt_device": (C.c_int, [_P, _P, _P, C.c_double, _P]),
    "lh_block_range": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64)]),
    "lh_comm_unique_id": (C.c_int, [_P]),
    "lh_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "lh_comm_destroy": (C.c_int, [_P]),
    "lh_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lh_allreduce_min": (C.c_int, [_P, _P]),
    "lh_get_status": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "lh_synchronize": (C.c_int, [_P]),
    "lh_stream_probe": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.c_int, C.POINTER(C.c_float)]),
    "lh_timer_start": (C.c_int, [_P]),
    "lh_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
}

_lib = None


class LandHydroError(RuntimeError):
    """error(msg) of the Julia shim: a failing C-ABI call."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class ModelError(LandHydroError, ValueError):
    """A model/BC combination the reference has no method for (LH_EMODEL), or
    an ArgumentError-class failure (LH_EINVAL)."""


def lib() -> C.CDLL:
    """Load the HIP library.  Raises if it has not been built: the product has
    no other compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                f = getattr(L, name)
            except AttributeError:
                # tools/ab_libs.py times OLDER builds of the library next to the product one: those may
                # lack the newest entry points (the product library never does: tests/test_abi_cpu.py)
                if os.environ.get("LH_ALLOW_MISSING_SYMBOLS") == "1":
                    continue
                raise
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int, ctx=None):
    if rc == LH_OK:
        return
    msg = lib().lh_last_error(ctx)
    msg = msg.decode() if msg else "unknown error"
    if rc in (LH_EMODEL, LH_EINVAL):
        raise ModelError(rc, msg)
    raise LandHydroError(rc, msg)


def dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return LH_F64
    if dtype == np.float32:
        return LH_F32
    raise TypeError(f"FT must be Float32 or Float64, got {dtype}")
