#!/usr/bin/env python3
"""Share of cells of one tendency evaluation that agree with the oracle to a PLAIN relative bound
(1e-13 Float64, 1e-6 Float32, of the field's largest |tendency|) for every parity case:
the statistic tests/parity_cases.assert_tendencies_close asserts next to its tolerance model."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch  # noqa: F401
import parity_cases as pc

CASES = ["c1_dirichlet_f64", "c2_richards_f64", "c2_richards_f32", "c3_coupled_f32", "c3_coupled_f64",
         "c4_richards_f64_128", "c5_percol_f64", "heat_dirichlet_f64", "heat_dirichlet_f32", "mixed_factors_f64",
         "mixed_factors_f32", "mixed_smooth_f64", "mixed_smooth_f32", "richards_viscosity_f64", "single_cell_f64"]
print(f"{'case':26s} {'field':5s} share within {pc.PLAIN_REL[np.dtype(np.float64)]:g} (f64) / {pc.PLAIN_REL[np.dtype(np.float32)]:g} (f32)   tolerance-model use (max err/allowed)")
for name in CASES:
    case = pc.make_case(name)
    got, want = pc.run_gpu_rhs(case), pc.run_oracle_rhs(case)
    Cw = 4.0 if case.dtype == np.float64 else 16.0
    use = pc.error_summary(case, got, want, Cw)
    for k, share in pc.plain_statistic(case, got, want).items():
        print(f"{name:26s} {k:5s} {share:.5f}   {use.get(k, float('nan')):.3f}")
