"""CPU-side checks of the C-ABI boundary (no GPU compute): the library builds,
loads, exports every symbol include/landhydro.h declares, and fails loudly --
never silently -- when there is no HIP device."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    g.build() if not os.path.exists(os.path.join(g.PKG_DIR, "lib", "liblandhydro_hip.so")) else None
    return g.load_package()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "landhydro.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lh_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(pkg):
    L = pkg._ffi.lib()
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(L, n), f"{n} declared in landhydro.h but not exported"
    # and the binding covers exactly the declared set
    assert sorted(pkg._ffi.SIGNATURES) == names


def test_version(pkg):
    assert pkg._ffi.lib().lh_version() == 1


def test_create_rejects_bad_arguments(pkg):
    F = pkg._ffi
    L = F.lib()
    ctx = C.c_void_p()
    bad = [
        F.lh_config(0, 64, F.LH_F64, -1.0, 0.0, F.LH_MODEL_RICHARDS, -1, None),   # ncols
        F.lh_config(10, 0, F.LH_F64, -1.0, 0.0, F.LH_MODEL_RICHARDS, -1, None),   # nlev
        F.lh_config(10, 64, F.LH_F64, 0.0, -1.0, F.LH_MODEL_RICHARDS, -1, None),  # zlim order
        F.lh_config(10, 64, 7, -1.0, 0.0, F.LH_MODEL_RICHARDS, -1, None),         # dtype
        F.lh_config(10, 64, F.LH_F64, -1.0, 0.0, 9, -1, None),                    # model
    ]
    for cfg in bad:
        rc = L.lh_create(C.byref(ctx), C.byref(cfg))
        assert rc == F.LH_EINVAL and not ctx.value
        assert L.lh_last_error(None)
    # every entry point refuses a NULL context before it touches a device
    assert L.lh_step_engine(None, 10, 0) == F.LH_EINVAL
    assert L.lh_upload_profile(None, None, F.LH_VAR_VARTHETA_L, None) == F.LH_EINVAL
    assert L.lh_state_release_ptr(None, None, F.LH_VAR_VARTHETA_L) == F.LH_EINVAL
    assert L.lh_boundary_fluxes(None, None, None, 0.0, 0, None, None) == F.LH_EINVAL
    assert L.lh_rhs(None, 0.0, None, None, None) == F.LH_EINVAL
    assert L.lh_step_ssprk33(None, None, None, 0.0, 1.0, 1, None) == F.LH_EINVAL


def test_no_device_is_a_loud_error_not_a_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    F = pkg._ffi
    ctx = C.c_void_p()
    cfg = F.lh_config(10, 64, F.LH_F64, -1.0, 0.0, F.LH_MODEL_RICHARDS, -1, None)
    rc = F.lib().lh_create(C.byref(ctx), C.byref(cfg))
    assert rc == F.LH_ENODEVICE and not ctx.value
    assert b"no CPU path" in F.lib().lh_last_error(None)
    with pytest.raises(F.LandHydroError):
        F.check(rc, None)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under landhydrology.jl_amd/ may
    reference it (SURVEY/DESIGN: a product path through the oracle voids parity)."""
    for dirpath, _, files in os.walk(g.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".jl")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0]
                    assert "oracle_py" not in code and "lh_oracle" not in code and "liblh_oracle" not in code, (
                        f"{f}: product code references the oracle: {line.strip()}")


def test_host_mirror_domain_and_errors(pkg):
    """test/test_domains.jl:13-32 through the host mirror (no device needed)."""
    import numpy as np
    for FT in (np.float32, np.float64):
        d = pkg.Column(FT, zlim=(0.0, 1.0), nelements=2)
        assert d.zlim == (0.0, 1.0) and d.nelements == 2
        assert d.ndims() == 1
        assert pkg.Column(FT, zlim=(1.0, 2.0), nelements=2).length() == 1.0
        assert pkg.Column(FT, zlim=(1.0, 4.0), nelements=2).size() == 3.0
        assert repr(d) == "[0.0, 1.0]"
        assert d.FT is FT
    with pytest.raises(AssertionError):
        pkg.Column(np.float64, zlim=(1.0, 0.0), nelements=2)     # domain.jl:30
    zc, zf = pkg.make_function_space(pkg.Column(np.float64, zlim=(-2.0, 0.0), nelements=20))
    want = np.array([(-195 + 10 * i) / 100 for i in range(20)])
    assert np.allclose(zc, want, rtol=0, atol=4.5e-16)           # coupled.jl:198
    with pytest.raises(TypeError):                                # Base.@kwdef: every field is required
        pkg.PrescribedAtmosForcing(u_atm=1.0)
    atm = pkg.PrescribedAtmosForcing(u_atm=0.34, θ_atm=299.0, z_atm=0.05, θ_scale=299.0, ρ_a_sfc=1.17,
                                     q_atm=0.015)
    assert atm.θ_atm == 299.0 and atm.rho_a_sfc == 1.17
    pkg.SoilColumnBC(top=atm, bottom=pkg.SoilComponentBC())
    with pytest.raises(TypeError):                                # BBC <: SoilComponentBC
        pkg.SoilColumnBC(top=pkg.SoilComponentBC(), bottom=atm)


def test_bench_input_generation_executes_nothing_of_the_oracle():
    """bench.py may use the oracle in its cpu_baseline leg only: importing bench.py and building
    the synthetic inputs of every workload must not even IMPORT anything from oracle/ -- nor from
    tests/ (the case generator and the C-ABI driver are package modules: workloads.py, case_model.py)."""
    import subprocess
    import sys
    code = (
        "import sys, os\n"
        "sys.argv = ['bench.py']\n"
        "import bench\n"
        "for w in bench.WORKLOADS:\n"
        "    c = bench.build_case(w, 300, 7)\n"
        "    assert c.vl is not None and c.vl.shape[0] == 300, w\n"
        "assert 'oracle_py' not in sys.modules, 'input generation imported the oracle binding'\n"
        "assert not any('oracle' in (getattr(m, '__file__', '') or '').split(os.sep) for m in list(sys.modules.values())), 'a module from oracle/ was imported'\n"
        "assert not any('tests' in (getattr(m, '__file__', '') or '').split(os.sep) for m in list(sys.modules.values())), 'bench.py imported a module from tests/'\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def _julia_ccalls(text):
    """Every `ccall((:name, lib), Ret, (ArgTypes...), args...)` of the Julia shim ->
    (name, ret, [argtypes]); the type tuple may span lines."""
    import re
    out = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*lib\),\s*(\w+),\s*\(", text):
        i, depth = m.end(), 1
        while depth:                      # matching parenthesis of the type tuple
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        types = [t.strip() for t in re.split(r",(?![^{]*\})", text[m.end():i - 1]) if t.strip()]
        out.append((m.group(1), m.group(2), types))
    return out


def test_julia_shim_ccalls_match_the_abi(pkg):
    """Julia cannot run in this image, so the shim is checked as text: every ccall names an
    exported symbol and passes the number, order and width of arguments the C ABI declares
    (the header is checked against the same table in test_header_symbols_binding_agree)."""
    F = pkg._ffi
    path = os.path.join(g.PKG_DIR, "julia", "LandHydrologyHIP.jl")
    text = open(path, encoding="utf-8").read()

    def jl_kind(t):
        if t.startswith("Ptr{") or t == "Cstring":
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Int64": "i64", "Float64": "f64",
                "Cfloat": "f32"}[t]

    def c_kind(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or issubclass(t, C._Pointer):
            return "ptr"
        return {C.c_int: "i32", C.c_int32: "i32", C.c_uint32: "u32", C.c_int64: "i64",
                C.c_double: "f64", C.c_float: "f32"}[t]

    calls = _julia_ccalls(text)
    assert len(calls) >= 25
    seen = set()
    for name, ret, types in calls:
        assert name in F.SIGNATURES, f"the shim calls {name}, which the ABI does not declare"
        res, args = F.SIGNATURES[name]
        assert jl_kind(ret) == c_kind(res), (name, ret)
        assert [jl_kind(t) for t in types] == [c_kind(a) for a in args], (name, types)
        seen.add(name)
    # the entry points a drop-in needs are all bound
    for must in ("lh_create", "lh_destroy", "lh_last_error", "lh_set_earth_params", "lh_set_soil_params",
                 "lh_set_vg_params", "lh_set_conductivity_factors", "lh_set_bc", "lh_state_create",
                 "lh_state_destroy", "lh_upload", "lh_download", "lh_state_fill", "lh_rhs",
                 "lh_rhs_stable_dt", "lh_step_ssprk33", "lh_ssprk33_stage", "lh_step_ssprk33_device_dt",
                 "lh_stable_dt", "lh_coordinates", "lh_block_range", "lh_comm_unique_id", "lh_comm_init",
                 "lh_comm_destroy"):
        assert must in seen, f"the Julia shim never calls {must}"
    # the closure of make_rhs does not shadow the device method, and uploads the aux fields
    body = text[text.index("function make_rhs(model::SoilModel{FT}, backend::HIPBackend)"):]
    assert "function rhs!(" not in body and "device_rhs!(ens, dYd, Yd, Yad, t)" in body
    assert "upload!(Yad, :T" in body and "upload!(Yad, :ϑ_l" in body and "aux_mask(model)" in body
    # finalizers: a state never calls into a destroyed context
    assert "s.ens.ctx != C_NULL" in text and "e.ctx = C_NULL" in text
