"""CPU: the C-ABI library builds, loads and exports every symbol include/dynhor_hip.h declares (no compute calls:
there is no GPU here); the host-only entry points behave; the product path has no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dynhor_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(hiplib):
    from dynhor_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/dynhor_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in dynhor_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header must list the same entry points"


def test_host_only_entry_points(hiplib):
    from dynhor_amd import _lib
    assert hiplib.dh_version() >= 1
    assert hiplib.dh_num_params() == 802491
    assert hiplib.dh_packed_floats() > 1_000_000
    assert b"bad argument" in hiplib.dh_strerror(-1) and hiplib.dh_strerror(0) == b"ok"
    # flat layout is dense and in state_dict order: sdf lin0..8 (bias, g, v), variance, colour lin0..4
    end = 0
    for net, n in ((0, 9), (1, 1), (2, 5)):
        for l in range(n):
            b, g, v, o, i = _lib.param_layout(net, l)
            if net == 1:
                assert v == end
                end += 1
                continue
            assert (b, g, v) == (end, end + o, end + 2 * o)
            end = v + o * i
    assert end == 802491
    assert _lib.param_layout(0, 3)[3:] == (217, 256) and _lib.param_layout(0, 8)[3:] == (257, 256)
    assert _lib.param_layout(2, 0)[3:] == (256, 289)
    with pytest.raises(_lib.DynhorHipError):
        _lib.param_layout(0, 9)
    inf, fwd, tot = _lib.workspace_floats(128 * 2048)
    assert 0 < inf < fwd < tot
    # an empty launch still owns the fixed-size table of per-class maxima (workspace.h absmax: 64 x 64 words)
    assert _lib.workspace_floats(0)[:2] == (4096, 4096)
    # the packed buffer's sections (dh_packed_section) lie inside it, in order, without overlap
    secs = [_lib.packed_section(i) for i in range(5)]
    assert all(o >= 0 and n > 0 and o + n <= hiplib.dh_packed_floats() for o, n in secs)
    by_off = sorted(secs)
    assert all(a[0] + a[1] <= b[0] for a, b in zip(by_off, by_off[1:]))
    assert secs[0][1] == 132 * 8 * 3 * 64 * 4 and secs[2][1] == 132 * 8 * 2 * 64 * 4
    with pytest.raises(_lib.DynhorHipError):
        _lib.packed_section(5)


def test_argument_validation_without_gpu(hiplib):
    null = ctypes.c_void_p(0)
    assert hiplib.dh_sdf_nograd(null, null, 0, null, null) == 0          # empty input is a no-op
    assert hiplib.dh_sdf_nograd(null, null, 5, null, null) == -1         # null pointers
    assert hiplib.dh_sdf_nograd(null, null, -1, null, null) == -1
    assert hiplib.dh_upsample_step(null, null, null, null, 4, 200, 16, 64.0, null, null, null) == -2   # unsupported n


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dynhor_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DynhorHipError):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "dynhor_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_example_configs_parse_into_upstream_constructor_kwargs():
    """configs/*.yaml use the reference's flat-YAML style; their model blocks must be valid upstream ctor kwargs."""
    import yaml
    from dynhor_amd.fields import RenderingNetwork, SDFNetwork, SingleVarianceNetwork
    from dynhor_amd.runner import DEFAULT_CONF, _merge
    for name in ("custom_shoes.yaml", "synthetic.yaml"):
        conf = _merge(DEFAULT_CONF, yaml.safe_load(open(os.path.join(ROOT, "configs", name))))
        assert {"seq_name", "exp_name", "data_info", "train", "model"} <= set(conf)
        sdf = SDFNetwork(**conf["model"]["sdf_network"])
        col = RenderingNetwork(**conf["model"]["rendering_network"])
        var = SingleVarianceNetwork(**conf["model"]["variance_network"])
        assert sum(p.numel() for m in (sdf, col, var) for p in m.parameters()) == 802491
        r = conf["model"]["neus_renderer"]
        assert r["n_samples"] + r["n_importance"] <= 128 and r["n_outside"] == 0


def test_stage_kernel_table_names_kernels_that_exist_in_the_library():
    """dynhor_amd/_lib.py:STAGE_KERNELS feeds bench.py's roofline block and scripts/make_traffic_json.py: every name in it must
    be a kernel of the built code object (the mangled names are registered as strings in the host part of the .so)."""
    from dynhor_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    names = {k for table in _lib.STAGE_KERNELS.values() for k in table.values()} | set(_lib.HASH_STAGE_KERNELS.values())
    for k in sorted(names):
        assert k.encode() in blob, f"{k} is not a kernel of libdynhor_hip.so (stale STAGE_KERNELS entry)"
    assert set(_lib.STAGE_KERNELS[0]) == set(_lib.STAGE_KERNELS[1]) == set(_lib.STAGE_KERNELS[2]), "all arithmetic modes list the same stages"


def test_traffic_tool_rejects_a_profile_that_lacks_a_shipping_kernel(tmp_path):
    import json
    import subprocess
    import sys
    from dynhor_amd import _lib
    entry = {"FETCH_SIZE": {"mean_per_dispatch": 1000.0, "dispatches": 3}, "WRITE_SIZE": {"mean_per_dispatch": 500.0, "dispatches": 3}}
    ship = _lib.STAGE_KERNELS[_lib.ARITH_DEFAULT]
    full = {"prof_pmc2": {k: entry for k in ship.values()}, "prof_pmc3": {k: entry for k in ship.values()}}
    src, dst = tmp_path / "pmc.json", tmp_path / "traffic.json"
    json.dump(full, open(src, "w"))
    tool = os.path.join(ROOT, "scripts", "make_traffic_json.py")
    p = subprocess.run([sys.executable, tool, str(src), str(dst)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = json.load(open(dst))
    assert out["weight_grads_gemm"]["kernel"] == ship["weight_grads_gemm"] == "dw_f16x2_kernel"
    assert out["weight_grads_gemm"]["hbm_bytes_per_launch"] == 1000.0 * 1024 * 2 + 500.0 * 1024      # FETCH_SIZE x2 on gfx950
    stale = {"prof_pmc2": dict(full["prof_pmc2"]), "prof_pmc3": dict(full["prof_pmc3"])}
    del stale["prof_pmc2"]["sdf_tangent_h_kernel"]                                                  # e.g. a renamed kernel
    json.dump(stale, open(src, "w"))
    p = subprocess.run([sys.executable, tool, str(src), str(dst)], capture_output=True, text=True)
    assert p.returncode != 0 and "sdf_tangent_h_kernel" in (p.stderr + p.stdout)


def test_committed_profiles_name_the_shipping_kernels():
    """profiles/pmc_traffic.json, the rocprofv3 kernel stats and the headline bench line must be OF the kernels that ship
    (round 1 committed a traffic table of kernels that no longer existed)."""
    import csv
    import json
    import re
    from dynhor_amd import _lib
    ship = _lib.STAGE_KERNELS[_lib.ARITH_DEFAULT]
    traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for stage, e in traffic.items():
        assert ship.get(stage) == e["kernel"], (stage, e["kernel"], ship.get(stage))
    names = set()
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_kernel_stats.csv"))):
        names.add(re.sub(r"[<(].*", "", re.sub(r"^dh::", "", re.sub(r"^void ", "", r["Name"]))))
    missing = sorted(set(ship.values()) - names)
    assert not missing, f"profiles/r06_kernel_stats.csv lacks shipping kernels {missing}"
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_n1.json")).read().strip().split("\n")[-1])
    assert line["roofline"]["kernel"] == ship[line["roofline"]["stage"]]
    for stage, v in line["kernels"].items():
        if "kernel" in v:
            assert v["kernel"] == ship[stage], (stage, v["kernel"])
    table = os.path.join(ROOT, "scripts", "kernel_table.py")
    import subprocess
    import sys
    p = subprocess.run([sys.executable, table], capture_output=True, text=True)
    assert p.returncode == 0 and "dw_f16x2_kernel" in p.stdout, p.stderr
