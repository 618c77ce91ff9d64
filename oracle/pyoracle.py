"""ctypes wrapper of the CPU oracle (oracle/libpwaf_oracle.so).

ORACLE — TEST INFRASTRUCTURE ONLY. Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package `pingoo_amd`. The oracle is a CPU restatement of the
reference's per-request rule evaluation (see oracle_expr.h / oracle_engine.cpp for the file:line
map). PARITY UNPINNED by the reference: it ships no tests and its interpreter is an un-vendored
dependency that cannot be built here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from pingoo_amd import _abi
from pingoo_amd.batch import VERDICT_DTYPE, RequestBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpwaf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("oracle_expr.cpp", "oracle_regex.cpp", "oracle_engine.cpp", "oracle_expr.h", "oracle_regex.h", "unicode_data.inc", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "pwaf.h"))
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libpwaf_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.pwaf_oracle_create.argtypes = [C.POINTER(_abi.RuleDesc), C.c_size_t, C.POINTER(_abi.ListDesc), C.c_size_t, C.POINTER(_abi.GeoipTable), C.c_uint32,
                                         C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.pwaf_oracle_create.restype = C.c_int
        L.pwaf_oracle_destroy.argtypes = [C.c_void_p]
        L.pwaf_oracle_header_count.argtypes = [C.c_void_p]
        L.pwaf_oracle_header_count.restype = C.c_uint32
        L.pwaf_oracle_header_name.argtypes = [C.c_void_p, C.c_uint32]
        L.pwaf_oracle_header_name.restype = C.c_char_p
        L.pwaf_oracle_destroy.restype = None
        L.pwaf_oracle_evaluate.argtypes = [C.c_void_p, C.POINTER(_abi.Batch), C.c_void_p, C.c_int]
        L.pwaf_oracle_evaluate.restype = C.c_int
        L.pwaf_oracle_execute_rule.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_abi.Batch), C.c_uint32]
        L.pwaf_oracle_execute_rule.restype = C.c_int
        L.pwaf_oracle_compile_expression.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.pwaf_oracle_validate_expression.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.pwaf_oracle_regex_is_match.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.pwaf_oracle_ipnet_contains.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.pwaf_oracle_parse_ip.argtypes = [C.c_char_p, C.c_char_p]
        L.pwaf_oracle_geoip_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint32), C.c_char_p]
        L.pwaf_oracle_derive_path.argtypes = [C.c_char_p, C.c_size_t]
        L.pwaf_oracle_derive_path.restype = C.c_size_t
        L.pwaf_oracle_derive_user_agent.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.pwaf_oracle_derive_user_agent.restype = None
        L.pwaf_oracle_derive_host.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.pwaf_oracle_derive_host.restype = None
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def compile_expression(expr: str) -> None:
    buf = C.create_string_buffer(512)
    rc = lib().pwaf_oracle_compile_expression(expr.encode(), buf, 512)
    if rc != 0:
        raise OracleError(rc, buf.value.decode(errors="replace"))


def validate_expression(expr: str) -> None:
    buf = C.create_string_buffer(512)
    rc = lib().pwaf_oracle_validate_expression(expr.encode(), buf, 512)
    if rc != 0:
        raise OracleError(rc, buf.value.decode(errors="replace"))


def regex_is_match(pattern: str | bytes, hay: bytes) -> bool:
    buf = C.create_string_buffer(512)
    p = pattern.encode() if isinstance(pattern, str) else pattern
    rc = lib().pwaf_oracle_regex_is_match(p, hay, len(hay), buf, 512)
    if rc < 0:
        raise OracleError(rc, buf.value.decode(errors="replace"))
    return rc == 1


def ipnet_contains(net: str, ip16: bytes, v6: bool) -> bool:
    rc = lib().pwaf_oracle_ipnet_contains(net.encode(), ip16, int(v6))
    if rc < 0:
        raise OracleError(rc, "invalid network " + net)
    return rc == 1


def parse_ip(s: str):
    out = C.create_string_buffer(16)
    fam = lib().pwaf_oracle_parse_ip(s.encode(), out)
    return fam, out.raw


def derive_path(p: bytes) -> bytes:
    return p[: lib().pwaf_oracle_derive_path(p, len(p))]


def derive_user_agent(hdr: bytes | None) -> bytes:
    s, l = C.c_size_t(), C.c_size_t()
    h = hdr or b""
    lib().pwaf_oracle_derive_user_agent(h, len(h), int(hdr is not None), C.byref(s), C.byref(l))
    return h[s.value:s.value + l.value]


def derive_host(uri_host: bytes | None, host_hdr: bytes | None) -> bytes:
    s, l, fh = C.c_size_t(), C.c_size_t(), C.c_int()
    a, b = uri_host or b"", host_hdr or b""
    lib().pwaf_oracle_derive_host(a, len(a), int(uri_host is not None), b, len(b), int(host_hdr is not None), C.byref(fh), C.byref(s), C.byref(l))
    src = b if fh.value else a
    return src[s.value:s.value + l.value]


class Oracle:
    """rules: [(name, expr|None, [action codes])]; lists: {name: (type, [items])}; geoip: GEOIP_DTYPE array."""

    def __init__(self, rules, lists=None, geoip=None, flags: int = 0):
        self._m = _abi.Marshalled()
        r, nr = _abi.marshal_rules(rules, self._m)
        l, nl = _abi.marshal_lists(lists, self._m)
        g = _abi.marshal_geoip(geoip, self._m)
        h = C.c_void_p()
        buf = C.create_string_buffer(512)
        rc = lib().pwaf_oracle_create(r, nr, l, nl, g, flags, C.byref(h), buf, 512)
        if rc != 0:
            raise OracleError(rc, buf.value.decode(errors="replace"))
        self._h = h
        self.n_rules = nr
        # EXTENSION: the header names the rule set mentions = the header columns batches are handed over with, in this order
        self.header_names = [lib().pwaf_oracle_header_name(h, i).decode() for i in range(lib().pwaf_oracle_header_count(h))]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pwaf_oracle_destroy(self._h)
            self._h = None

    def evaluate(self, batch: RequestBatch, threads: int = 1) -> np.ndarray:
        out = np.zeros(batch.n, dtype=VERDICT_DTYPE)
        st = batch.as_struct(self.header_names)
        rc = lib().pwaf_oracle_evaluate(self._h, C.byref(st), out.ctypes.data, threads)
        if rc != 0:
            raise OracleError(rc, "oracle evaluate failed (malformed batch)")
        return out

    def execute_rule(self, rule: int, batch: RequestBatch, i: int) -> int:
        """1 Bool(true), 0 Bool(false), 2 non-Bool, 3 execution error."""
        st = batch.as_struct(self.header_names)
        return lib().pwaf_oracle_execute_rule(self._h, rule, C.byref(st), i)

    def geoip_lookup(self, ip16: bytes, v6: bool):
        asn = C.c_uint32()
        cc = C.create_string_buffer(2)
        lib().pwaf_oracle_geoip_lookup(self._h, ip16, int(v6), C.byref(asn), cc)
        return asn.value, cc.raw
