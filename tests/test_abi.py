"""The C-ABI library loads and exports exactly what include/pwaf.h declares (no GPU, no compute)."""
import ctypes as C
import os
import re
import random

import pytest

from oracle import pyoracle
from pingoo_amd import _abi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "pwaf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pwaf_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = engine.lib()
    names = header_functions()
    assert len(names) >= 20, names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/pwaf.h but not exported by libpwaf.so: {missing}"
    assert L.pwaf_abi_version() == _abi.ABI_VERSION


def test_struct_sizes_match_header_layout():
    # the numbers are what a C compiler gives for include/pwaf.h on x86-64 (checked by tests/abi_sizes.c at build time)
    assert C.sizeof(_abi.RuleDesc) == 32 and C.sizeof(_abi.ListDesc) == 24 and C.sizeof(_abi.GeoipEntry) == 24
    assert C.sizeof(_abi.Options) == 32 and C.sizeof(_abi.CompileError) == 256 and C.sizeof(_abi.Verdict) == 8
    assert C.sizeof(_abi.Batch) == 16 + 5 * 16 + 6 * 8 + 5 * 4 + 4 + 2 * 8 and C.sizeof(_abi.Counts) == 32 and C.sizeof(_abi.Stats) == 64
    assert C.sizeof(_abi.KernelTime) == 64 and C.sizeof(_abi.Request) == 104


def test_struct_sizes_against_c_compiler(tmp_path):
    import subprocess
    prog = tmp_path / "sizes.c"
    prog.write_text('#include <stdio.h>\n#include "pwaf.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",sizeof(pwaf_rule_desc),sizeof(pwaf_list_desc),'
                    'sizeof(pwaf_geoip_entry),sizeof(pwaf_options),sizeof(pwaf_compile_error),sizeof(pwaf_verdict),sizeof(pwaf_batch),sizeof(pwaf_counts),sizeof(pwaf_stats),'
                    'sizeof(pwaf_kernel_time),sizeof(pwaf_request));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(t) for t in (_abi.RuleDesc, _abi.ListDesc, _abi.GeoipEntry, _abi.Options, _abi.CompileError, _abi.Verdict, _abi.Batch, _abi.Counts, _abi.Stats,
                                  _abi.KernelTime, _abi.Request)]
    assert got == want


def test_error_paths_do_not_need_a_gpu():
    L = engine.lib()
    assert L.pwaf_node_evaluate_device(None, None, None, None, None) == _abi.E_INVALID_ARG
    assert L.pwaf_node_synchronize(None) == _abi.E_INVALID_ARG and L.pwaf_node_allreduce_counts(None, None, None, None) == _abi.E_INVALID_ARG
    assert L.pwaf_evaluate_batch(None, None, None, None) == _abi.E_INVALID_ARG
    # round 4: the residual-program hooks and the page-locked memory entry points
    assert L.pwaf_engine_residual_mode(None) == 0 and L.pwaf_program_residual_source(None, 1, None, 0) == 0
    assert L.pwaf_program_residual_compile(None, b"gfx950", None, 0) == _abi.E_INVALID_ARG
    assert L.pwaf_host_alloc(16, None) == _abi.E_INVALID_ARG and L.pwaf_host_register(None, 16) == _abi.E_INVALID_ARG and L.pwaf_host_unregister(None) == _abi.E_INVALID_ARG
    assert b"NULL" in L.pwaf_last_error()
    with pytest.raises(engine.ExpressionIsNotValid):
        engine.compile_expression("a ==")
    engine.compile_expression("x in [1]")
    with pytest.raises(engine.ExpressionIsNotValid, match="unknown operator: in"):
        engine.validate_expression("x in [1]")
    with pytest.raises(engine.ExpressionIsNotValid, match="expression is empty"):
        engine.validate_expression("")


def test_field_derivation_matches_golden_and_oracle(kat):
    d = kat["derive"]
    for raw, want in d["user_agent"]:
        h = None if raw is None else raw.encode("latin-1")
        assert engine.get_user_agent(h) == want.encode("latin-1") == pyoracle.derive_user_agent(h)
    for uri, hdr, want in d["host"]:
        a = None if uri is None else uri.encode("latin-1")
        b = None if hdr is None else hdr.encode("latin-1")
        assert engine.get_host(a, b) == want.encode("latin-1") == pyoracle.derive_host(a, b)
    for raw, want in d["path"]:
        assert engine.get_path(raw.encode()) == want.encode() == pyoracle.derive_path(raw.encode())
    rng = random.Random(5)
    alpha = [b" ", b"\t", b"a", b"/", b"\x7f", b"\x80", b"\n", b"~", b"\x1f"]
    for _ in range(3000):
        s = b"".join(rng.choice(alpha) for _ in range(rng.randint(0, 12))) * rng.choice([1, 1, 1, 30])
        present = rng.random() < 0.9
        h = s if present else None
        assert engine.get_user_agent(h) == pyoracle.derive_user_agent(h)
        h2 = s[::-1] if rng.random() < 0.5 else None
        assert engine.get_host(h, h2) == pyoracle.derive_host(h, h2)
        assert engine.get_path(s) == pyoracle.derive_path(s)
