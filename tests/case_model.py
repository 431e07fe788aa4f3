"""The plain-number model descriptions live in the package (landhydrology.jl_amd/case_model.py, so
that bench.py depends on nothing under tests/); the tests and tools keep importing this name."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as _g  # noqa: E402

_m = _g.load_package().case_model
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
