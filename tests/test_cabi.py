"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/mocha_hip.h declares (no compute calls here: there is no GPU)."""
import os
import re

import pytest

from mocha_sigasia2023_amd import _C, weights

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    if not os.path.exists(_C.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _C.load_library()


def header_symbols():
    txt = open(os.path.join(REPO, "include", "mocha_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mocha_[a-z_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    syms = header_symbols()
    assert syms, "no symbols parsed from the header"
    assert sorted(_C.SIGNATURES) == syms


def test_library_exports_every_symbol():
    lib = _built()
    for s in header_symbols():
        assert hasattr(lib, s), f"libmocha_hip.so does not export {s}"
    assert lib.mocha_abi_version() == _C.ABI_VERSION


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_C, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _C.load_library(str(tmp_path / "libmocha_hip.so"))


def test_product_path_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the product package may import or call it."""
    pkg = os.path.join(REPO, "mocha_sigasia2023_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "mocha_oracle" not in src, f


def test_weight_schema_matches_reference_count():
    # 6 116 559 learnable parameters
    n = sum(int(__import__("numpy").prod(s)) for s in weights.param_shapes().values())
    assert n == 6116559
    sd = weights.synthetic_state_dict(1, 1.0)
    assert set(weights.param_shapes()) | set(weights.buffer_arrays()) == set(sd)


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` spawns its own ranks; on a node with fewer GPUs (here: none) it must refuse before any
    rank starts, without touching a GPU."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "exposes" in (r.stderr + r.stdout)


def test_device_code_has_no_packed_fp32_op_sel_pattern():
    """tools/isa_lint.py over the built library: no kernel outside its allow-list contains ``v_pk_*_f32`` with ``op_sel`` on a high
    register - the pattern that, fed from LDS, made mocha_body_front depend on what shared its CU (DESIGN_HISTORY.md §8, Concurrency)."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    _built()
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(REPO, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec); spec.loader.exec_module(lint)
    found = lint.scan(_C.LIB_PATH)
    bad = {k: v for k, v in found.items() if not any(a in k for a in lint.ALLOW)}
    assert not bad, bad


def test_bcast_plan_covers_every_float_exactly_once():
    """mocha_bank_broadcast's split (csrc/mocha_api.cpp: bcast_plan): `world` equal chunks, one per rank, contiguous from 0,
    plus a count % world tail that is broadcast whole.  Pure host arithmetic, checked here for ragged counts."""
    import ctypes as C
    lib = _built()
    out = (C.c_int64 * 4)()
    for world in (1, 2, 3, 5, 8):
        for count in (0, 1, 7, 23040, 23040 * 5, 23040 * 585, 23040 * 4096 + 3, (1 << 31) + 11):
            cover = 0
            for rank in range(world):
                assert lib.mocha_bcast_plan(count, world, rank, out) == 0
                off, chunk, toff, tail = list(out)
                assert chunk == count // world and off == rank * chunk
                assert toff == chunk * world and tail == count - toff and 0 <= tail < world
                cover += chunk
            assert cover + tail == count
    assert lib.mocha_bcast_plan(10, 0, 0, out) != 0 and lib.mocha_bcast_plan(10, 2, 2, out) != 0 and lib.mocha_bcast_plan(-1, 2, 0, out) != 0


def test_build_info_names_the_toolchain():
    lib = _built()
    assert lib.mocha_build_info().decode().startswith("hipcc HIP ")


def test_native_host_example_builds_against_the_header_and_library():
    """examples/cabi_demo.cpp (C++ + the HIP runtime, no torch) compiles against include/mocha_hip.h and links against the in-tree
    libmocha_hip.so - hipcc cross-compiles without a GPU; tests/test_native_host.py runs it on the GPU box."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples")], check=True, capture_output=True)
    exe = os.path.join(root, "examples", "cabi_demo")
    assert os.path.exists(exe)
    needed = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libmocha_hip.so" in needed and "libtorch" not in needed and "libpython" not in needed
