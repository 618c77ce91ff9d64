"""The SPECIALIZED form of the residual rules (pingoo_amd/csrc/residual_jit.cpp) on the CPU.

An engine does not interpret its residual rules (tests/test_residual.py) unless it has to: at creation their stack programs are
translated to straight-line code — every stack slot a local, every constant a literal, the same op_* functions of residual.h the
interpreter calls — and compiled for the device by hiprtc (csrc/rtc.cpp). Here, without a GPU:
  * the translation of fuzzed rule sets is compiled with g++ (TEST-ONLY host build) and must give, for every rule and request, the
    interpreter's three-way result (match / execution error / neither) and the oracle's verdict (pingoo/rules.rs:37-51);
  * the whole device program of such a rule set must compile for gfx950 through hiprtc (no device needed), through the C ABI's
    inspection hooks (pwaf_program_residual_source / _compile).
The device path itself is tests/test_gpu_residual.py."""
import ctypes as C
import os
import random
import subprocess

import pytest

import helpers as H
from pingoo_amd import _abi
from pingoo_amd.engine import RequestBatch
from oracle import pyoracle
import test_residual as TR

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "pingoo_amd", "csrc")
BUILD = os.path.join(HERE, "_build")

HARNESS = r"""
#define PWAF_RVM_RULE static inline
#include "residual.h"
%s
extern "C" int spec_eval(const uint8_t *blob, uint32_t rule, const uint8_t *const *data, const uint32_t *const *off, uint32_t r, const uint8_t *ip16, uint32_t v6,
                         uint32_t port, uint32_t asn, uint32_t country) {
    pwaf::rvm::Machine m;
    m.blob = blob;
    m.h = reinterpret_cast<const pwaf::rvm::Header *>(blob);
    m.q.data = data;
    m.q.off = off;
    m.q.r = r;
    m.q.ip = ip16;
    m.q.v6 = v6;
    m.q.port = port;
    m.q.asn = asn;
    m.q.country = country;
    m.heap_n = 0;
    return (int)pwaf::rvm::rvm_rule_dispatch(m, rule);
}
"""


def heap_items_of(m) -> int:
    """Header::heap_items of the program image (residual.h): the compiler's bound on what any rule puts on its lane's heap."""
    n = C.c_size_t(0)
    TR.vm().rvmh_blob.restype = C.c_void_p
    TR.vm().rvmh_blob.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    p = TR.vm().rvmh_blob(m._h, C.byref(n))
    words = (C.c_uint32 * 16).from_address(p)
    assert words[0] == 0x314D5652 and words[12] == n.value, "Header layout (residual.h) changed"
    return int(words[13])


def build_specialized(m: "TR.HostVM", tag: str):
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(BUILD, f"spec_{tag}.cpp")
    lib = os.path.join(BUILD, f"libspec_{tag}.so")
    with open(src, "w") as f:
        f.write(HARNESS % m.specialized_source())
    # the heap exactly as large as the device program's (residual_jit.cpp defines PWAF_RVM_HEAP the same way): a bound that is too
    # small shows up here as an execution error the interpreter (64 items) does not report
    heap = max(1, min(heap_items_of(m), 64))
    subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", f"-DPWAF_RVM_HEAP={heap}", "-I", CSRC, "-I", os.path.join(HERE, "..", "include"), src, "-o", lib], check=True)
    L = C.CDLL(lib)
    L.spec_eval.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    return L


def blob_of(m):
    n = C.c_size_t(0)
    TR.vm().rvmh_blob.restype = C.c_void_p
    TR.vm().rvmh_blob.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    p = TR.vm().rvmh_blob(m._h, C.byref(n))
    return p


def accepted(exprs, lists):
    out = []
    for e in exprs:
        try:
            pyoracle.compile_expression(e)
            TR.HostVM([e], lists)
        except (pyoracle.OracleError, ValueError):
            continue
        out.append(e)
    return out


def check_rule_set(exprs, lists, batch, tag):
    m = TR.HostVM(exprs, lists)
    L = build_specialized(m, tag)
    blob = blob_of(m)
    m.bind(batch)
    # ONE oracle over the whole rule set: the headers map holds the names ALL its rules mention with a literal key, and a computed key
    # (`http_request.headers["x-" + "a"]`, round 5) sees that map — a single-rule oracle would see only the rule's own names
    orc = pyoracle.Oracle([(f"r{k}", e, [H.B]) for k, e in enumerate(exprs)], lists, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    assert orc.header_names == m.header_names
    for k, e in enumerate(exprs):
        for i in range(batch.n):
            b = batch
            asn = int(b.asn[i]) if b.asn is not None else 0
            country = int(b.country[i]) if b.country is not None else int.from_bytes(b"XX", "little")
            got = int(L.spec_eval(blob, k, m._data, m._off, i, b.ip[i].ctypes.data, int(b.ip_is_v6[i]), int(b.port[i]), asn, country))
            assert got == m.eval3(k, i), (tag, e, i, "specialized form and interpreter disagree")
            want3 = orc.execute_rule(k, batch, i)  # 1 true, 0 false, 2 non-Bool, 3 error
            assert got == {1: 1, 0: 0, 2: 0, 3: 2}[want3], (tag, e, i, got, want3)


@pytest.mark.parametrize("seed", range(10))
def test_specialized_rule_sets_agree_with_interpreter_and_oracle(seed):
    rng = random.Random(424200 + seed)  # (the seeds of test_residual.py: the same expressions)
    exprs = accepted([TR.dbool(rng) for _ in range(14)], TR.LISTS)
    assert len(exprs) >= 4
    batch = RequestBatch.from_requests(TR.requests(rng, 40))
    check_rule_set(exprs, TR.LISTS, batch, f"r{seed}")


@pytest.mark.parametrize("seed", range(4))
def test_specialized_general_fuzzer_expressions(seed):
    rng = random.Random(515100 + seed)
    lists = H.fuzz_lists(rng)
    exprs = accepted([H.rexpr(rng, lists) for _ in range(15)], lists)
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, 40, with_geo=True))
    check_rule_set(exprs, lists, batch, f"g{seed}")


def test_known_shapes():
    """Short-circuit operators, the three exits of ?:, list / map construction, errors that must survive to the result."""
    exprs = [
        'http_request.path + "x" == "/ax"',
        '(http_request.method == "GET" ? http_request.host : http_request.path).length() > 3',
        '(client.remote_port ? true : false)',  # non-Bool condition: an execution error
        '(client.remote_port > 1000 ? 1 : "a") == 1',
        '1 / (client.remote_port - client.remote_port) == 1 || true',  # the error is on the left: the result is an error
        'true || 1 / (client.remote_port - client.remote_port) == 1',  # decided before the error is reached
        'client.remote_port in [80, 443, client.asn]',
        '{"a": http_request.host, "b": http_request.path}.b.starts_with("/")',
        '[http_request.host, http_request.path][client.remote_port % 2].length() >= 1',
        '!(http_request.url + http_request.host).contains("..") && client.country + "x" != "USx"',
        '(http_request.path + "?" + http_request.host).matches("^/[a-z]+\\\\?")',
        'http_request.path.length() * 2 + 1 > http_request.url.length()',
    ]
    rng = random.Random(7)
    batch = RequestBatch.from_requests(TR.requests(rng, 60))
    ok = accepted(exprs, TR.LISTS)
    assert len(ok) >= len(exprs) - 2, [e for e in exprs if e not in ok]
    check_rule_set(ok, TR.LISTS, batch, "known")


def test_device_program_compiles_for_gfx950():
    """pwaf_program_compile -> pwaf_program_residual_source / _compile: the text hiprtc gets at engine creation, compiled here for
    gfx950 (hiprtc needs no device). Also: a rule set without residual rules has no such program."""
    from pingoo_amd.engine import CompiledProgram
    rules = [
        ("col", 'http_request.path.starts_with("/admin")', [H.B]),
        ("r1", 'http_request.path + "x" == "/ax"', [H.B]),
        ("r2", '(client.remote_port > 1000 ? http_request.host : http_request.path).length() > 3 && client.asn + 1 > 5', [H.B]),
        ("r3", 'client.remote_port in [80, 443, client.asn]', [H.B]),
        ("r4", '(http_request.path + "?" + http_request.host).matches("^/[a-z]+\\\\?")', [H.B]),
    ]
    prog = CompiledProgram(rules, TR.LISTS)
    fns, text = prog.residual_source(0), prog.residual_source(1)
    assert len(fns) > 200 and len(text) > len(fns) + 10000  # (kind 1 carries residual.h)
    assert "rvm_jit_kernel" in text and "rvm_rule_3(" in text and "rvm_rule_4(" not in text  # (four residual rules; the first rule has a column form)
    assert prog.residual_compile("gfx950") > 1000
    prog = CompiledProgram(rules[:1], TR.LISTS)
    assert prog.residual_source(1) == "" and prog.residual_compile("gfx950") == 0
