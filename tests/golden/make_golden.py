#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

Provenance: the reference (Julia) cannot run in the build container, so these
vectors are produced by the CPU oracle (oracle/, a restatement pinned by the
reference's known-answer tests -- tests/test_oracle_pins.py) AFTER it passes
those pins.  Each file holds the inputs, the tendencies of one rhs! evaluation
and the state after `nsteps` SSPRK33 steps, for a tiny batch of one parity case
(tests/parity_cases.py).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import case_model as M  # noqa: E402
import parity_cases as pc  # noqa: E402

CASES = {  # name -> (ncols, nsteps)
    "c1_dirichlet_f64": (1, 20),
    "c2_richards_f64": (4, 20),
    "c3_coupled_f32": (4, 20),
    "c3_coupled_f64": (4, 20),
    "c4_richards_f64_128": (4, 10),
    "c5_percol_f64": (4, 10),
    "heat_dirichlet_f64": (4, 20),
    "mixed_smooth_f64": (5, 20),
    "mixed_factors_f32": (5, 0),
    "richards_viscosity_f64": (3, 10),
}


def build(name):
    ncols, nsteps = CASES[name]
    case = pc.make_case(name, ncols=ncols)
    O = pc.O
    out = {"ncols": ncols, "nsteps": nsteps}
    for k in ("vl", "ti", "rhoe", "T_aux"):
        a = getattr(case, k)
        if a is not None:
            out["in_" + k] = a
    for k, v in pc.run_oracle_rhs(case).items():
        out["d_" + k] = v
    if nsteps:
        dt = O.stable_dt(case.om, case.vl, case.ti, case.rhoe, 0.2, case.T_aux)
        cp = lambda a: None if a is None else a.copy()
        vl, ti, re = cp(case.vl), cp(case.ti), cp(case.rhoe)
        O.ssprk33(case.om, dt, nsteps, vl=vl, ti=ti, rhoe=re, T_aux=case.T_aux)
        out["dt"] = dt
        for k, v in (("vl", vl), ("ti", ti), ("rhoe", re)):
            if v is not None and not (case.om.model == M.MODEL_HEAT and k != "rhoe") \
                    and not (case.om.model == M.MODEL_RICHARDS and k == "rhoe"):
                out["end_" + k] = v
    return out


if __name__ == "__main__":
    for name in CASES:
        data = build(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print(name, {k: getattr(v, "shape", v) for k, v in data.items()})
