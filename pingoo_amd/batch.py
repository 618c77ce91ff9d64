"""Request / RequestBatch: the host-side mirror of the reference's per-request inputs.

`Request` carries exactly what the reference puts into `RequestData` / `ClientData`
(pingoo/rules.rs:16-34) plus the `captcha_verified` bit the listener computes from the cookie
(http_listener.rs:222-236). `RequestBatch` is the struct-of-arrays wire layout of include/pwaf.h:
per string field one byte arena + n+1 offsets (field-major, so a field's bytes are contiguous in
request order — what the scan kernels stream), plus fixed-width numeric columns.
"""
from __future__ import annotations

import ctypes as C
import ipaddress
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _abi

GEOIP_DTYPE = np.dtype(
    [("addr", np.uint8, (16,)), ("prefix_len", np.uint8), ("is_v6", np.uint8), ("country", np.uint8, (2,)), ("asn", np.uint32)]
)
VERDICT_DTYPE = np.dtype([("action", np.uint8), ("pad", np.uint8, (3,)), ("rule_idx", np.uint32)])
assert GEOIP_DTYPE.itemsize == 24 and VERDICT_DTYPE.itemsize == 8


def _b(x) -> bytes:
    return x.encode("utf-8") if isinstance(x, str) else bytes(x)


@dataclass
class Request:
    """One request as the rule context sees it (after the reference's field derivation)."""

    host: bytes | str = b""
    url: bytes | str = b"/"
    path: bytes | str = b""
    method: bytes | str = b"GET"
    user_agent: bytes | str = b"Mozilla/5.0"
    ip: str = "192.0.2.1"
    remote_port: int = 40000
    asn: Optional[int] = None  # None => the engine looks the ip up in its GeoIP table
    country: Optional[str] = None
    captcha_verified: bool = False
    headers: Optional[dict] = None  # EXTENSION: exact header name -> value; an absent header reads as ""


def ip_to_bytes16(ip: str) -> tuple[bytes, bool]:
    a = ipaddress.ip_address(ip)
    if a.version == 4:
        return a.packed + b"\0" * 12, False
    return a.packed, True


class RequestBatch:
    """Host-resident SoA batch (numpy). `as_struct()` yields a pwaf_batch pointing at the arrays."""

    def __init__(self, data: Sequence[np.ndarray], offsets: Sequence[np.ndarray], ip: np.ndarray, ip_is_v6: np.ndarray,
                 port: np.ndarray, flags: np.ndarray, asn: Optional[np.ndarray] = None, country: Optional[np.ndarray] = None,
                 headers: Optional[dict] = None):
        assert len(data) == _abi.N_FIELDS and len(offsets) == _abi.N_FIELDS
        self.n = int(len(port))
        # header columns by name: (arena, offsets) like a field; a consumer orders them by ITS list of names (as_struct)
        self.headers = {}
        for name, (hd, ho) in (headers or {}).items():
            hd, ho = np.ascontiguousarray(hd, dtype=np.uint8), np.ascontiguousarray(ho, dtype=np.uint32)
            assert len(ho) == self.n + 1 and len(hd) >= int(ho[-1]) + _abi.ARENA_PAD
            self.headers[name] = (hd, ho)
        self.data = [np.ascontiguousarray(d, dtype=np.uint8) for d in data]
        self.offsets = [np.ascontiguousarray(o, dtype=np.uint32) for o in offsets]
        for d, o in zip(self.data, self.offsets):
            assert len(o) == self.n + 1
            assert len(d) >= int(o[-1]) + _abi.ARENA_PAD, "arena must carry PWAF_ARENA_PAD slack bytes"
        self.ip = np.ascontiguousarray(ip, dtype=np.uint8).reshape(self.n, 16)
        self.ip_is_v6 = np.ascontiguousarray(ip_is_v6, dtype=np.uint8)
        self.port = np.ascontiguousarray(port, dtype=np.uint16)
        self.flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert (asn is None) == (country is None)
        self.asn = None if asn is None else np.ascontiguousarray(asn, dtype=np.uint32)
        self.country = None if country is None else np.ascontiguousarray(country, dtype=np.uint16)

    @staticmethod
    def from_requests(reqs: Iterable[Request], with_geoip: Optional[bool] = None) -> "RequestBatch":
        reqs = list(reqs)
        n = len(reqs)
        if with_geoip is None:
            with_geoip = n > 0 and all(r.asn is not None and r.country is not None for r in reqs)
        datas, offs = [], []
        for f in _abi.FIELD_NAMES:
            vals = [_b(getattr(r, f)) for r in reqs]
            o = np.zeros(n + 1, dtype=np.uint32)
            if n:
                o[1:] = np.cumsum([len(v) for v in vals], dtype=np.uint64).astype(np.uint32)
            blob = b"".join(vals) + b"\0" * _abi.ARENA_PAD
            datas.append(np.frombuffer(blob, dtype=np.uint8).copy())
            offs.append(o)
        ip = np.zeros((n, 16), dtype=np.uint8)
        v6 = np.zeros(n, dtype=np.uint8)
        for i, r in enumerate(reqs):
            b16, is6 = ip_to_bytes16(r.ip)
            ip[i] = np.frombuffer(b16, dtype=np.uint8)
            v6[i] = is6
        port = np.array([r.remote_port for r in reqs], dtype=np.uint16)
        flags = np.array([_abi.FLAG_CAPTCHA_VERIFIED if r.captcha_verified else 0 for r in reqs], dtype=np.uint8)
        asn = country = None
        if with_geoip:
            asn = np.array([r.asn or 0 for r in reqs], dtype=np.uint32)
            country = np.array([int.from_bytes(_b(r.country or "XX")[:2], "little") for r in reqs], dtype=np.uint16)
        names = []
        for r in reqs:
            for k in (r.headers or {}):
                if k not in names:
                    names.append(k)
        headers = {}
        for name in names:
            vals = [_b((r.headers or {}).get(name, b"")) for r in reqs]
            o = np.zeros(n + 1, dtype=np.uint32)
            if n:
                o[1:] = np.cumsum([len(v) for v in vals], dtype=np.uint64).astype(np.uint32)
            headers[name] = (np.frombuffer(b"".join(vals) + b"\0" * _abi.ARENA_PAD, dtype=np.uint8).copy(), o)
        return RequestBatch(datas, offs, ip, v6, port, flags, asn, country, headers)

    def header_bytes(self, name: str, i: int) -> bytes:
        if name not in self.headers:
            return b""
        d, o = self.headers[name]
        return d[int(o[i]):int(o[i + 1])].tobytes()

    def field_bytes(self, field: int, i: int, header_names: Sequence[str] = ()) -> bytes:
        """Value of field id `field` of request i: 0..4 the fixed fields, 5 + k the k-th name of `header_names`."""
        if field >= _abi.N_FIELDS:
            return self.header_bytes(header_names[field - _abi.N_FIELDS], i)
        o = self.offsets[field]
        return self.data[field][int(o[i]):int(o[i + 1])].tobytes()

    def slice(self, lo: int, hi: int) -> "RequestBatch":
        """Requests [lo, hi) as an independent batch (arenas re-based)."""
        datas, offs = [], []
        for d, o in zip(self.data, self.offsets):
            b0, b1 = int(o[lo]), int(o[hi])
            nd = np.zeros(b1 - b0 + _abi.ARENA_PAD, dtype=np.uint8)
            nd[: b1 - b0] = d[b0:b1]
            datas.append(nd)
            offs.append((o[lo:hi + 1] - o[lo]).astype(np.uint32))
        headers = {}
        for name, (d, o) in self.headers.items():
            b0, b1 = int(o[lo]), int(o[hi])
            nd = np.zeros(b1 - b0 + _abi.ARENA_PAD, dtype=np.uint8)
            nd[: b1 - b0] = d[b0:b1]
            headers[name] = (nd, (o[lo:hi + 1] - o[lo]).astype(np.uint32))
        return RequestBatch(datas, offs, self.ip[lo:hi], self.ip_is_v6[lo:hi], self.port[lo:hi], self.flags[lo:hi],
                            None if self.asn is None else self.asn[lo:hi], None if self.country is None else self.country[lo:hi], headers)

    def take(self, idx) -> "RequestBatch":
        """The requests at the (ascending or not) indices `idx` as an independent batch (test helper: a random sample of a large batch)."""
        idx = np.asarray(idx, dtype=np.int64)

        def gather(d, o):
            o64 = o.astype(np.int64)
            lens = o64[idx + 1] - o64[idx]
            no = np.zeros(len(idx) + 1, dtype=np.uint32)
            no[1:] = np.cumsum(lens).astype(np.uint32)
            src = np.repeat(o64[idx] - no[:-1].astype(np.int64), lens) + np.arange(int(no[-1]), dtype=np.int64)
            nd = np.zeros(int(no[-1]) + _abi.ARENA_PAD, dtype=np.uint8)
            nd[: int(no[-1])] = d[src]
            return nd, no

        cols = [gather(d, o) for d, o in zip(self.data, self.offsets)]
        headers = {name: gather(d, o) for name, (d, o) in self.headers.items()}
        return RequestBatch([c[0] for c in cols], [c[1] for c in cols], self.ip[idx], self.ip_is_v6[idx], self.port[idx], self.flags[idx],
                            None if self.asn is None else self.asn[idx], None if self.country is None else self.country[idx], headers)

    def tile(self, times: int) -> "RequestBatch":
        """The batch repeated `times` times (test / bench helper for very large uniform batches)."""
        datas, offs = [], []
        for d, o in zip(self.data, self.offsets):
            body = d[int(o[0]):int(o[-1])]
            nd = np.concatenate([np.tile(body, times), np.zeros(_abi.ARENA_PAD, dtype=np.uint8)])
            lens = np.tile(np.diff(o.astype(np.int64)), times)
            no = np.zeros(self.n * times + 1, dtype=np.uint32)
            no[1:] = np.cumsum(lens).astype(np.uint32)
            datas.append(nd)
            offs.append(no)
        rep = lambda a: None if a is None else np.tile(a, (times,) + (1,) * (a.ndim - 1))
        assert not self.headers, "tile(): header columns are not supported"
        return RequestBatch(datas, offs, rep(self.ip), rep(self.ip_is_v6), rep(self.port), rep(self.flags), rep(self.asn), rep(self.country))

    def _arrays(self):
        arrs = list(self.data) + list(self.offsets) + [self.ip, self.ip_is_v6, self.port, self.flags]
        if self.asn is not None:
            arrs += [self.asn, self.country]
        for hd, ho in self.headers.values():
            arrs += [hd, ho]
        return [a for a in arrs if a.nbytes]  # (pwaf_host_register refuses zero bytes: an all-empty header column has nothing to lock)

    def pin(self) -> "RequestBatch":
        """Page-locks every column (pwaf_host_register): the copy engine then reads the caller's bytes directly — no staging copy inside
        the runtime, and pwaf_evaluate_batch validates the offsets while the copies are in flight. What a listener that parses requests
        into pwaf_host_alloc'd arenas gets without this step. Undo with unpin() before the arrays are freed."""
        from .engine import _raise, lib  # (engine imports this module)

        if getattr(self, "_pinned", None):
            return self
        done = []
        for a in self._arrays():
            rc = lib().pwaf_host_register(a.ctypes.data, a.nbytes)
            if rc != 0:
                for q in done:
                    lib().pwaf_host_unregister(q.ctypes.data)
                _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
            done.append(a)
        self._pinned = done
        return self

    def unpin(self) -> None:
        from .engine import lib

        for a in getattr(self, "_pinned", None) or []:
            lib().pwaf_host_unregister(a.ctypes.data)
        self._pinned = None

    def view(self, lo: int, hi: int) -> "RequestBatch":
        """Requests [lo, hi) as a SLAB VIEW: the same arenas, offsets[lo : hi + 1] unchanged (absolute positions, offsets[0] != 0 in
        general) — what pwaf_node_evaluate_batch hands each device's engine (csrc/node.cpp) and what a caller that splits one parsed
        buffer passes. `slice()` is the re-based copy."""
        headers = {name: (d, o[lo:hi + 1]) for name, (d, o) in self.headers.items()}
        v = RequestBatch.__new__(RequestBatch)
        v.n = hi - lo
        v.headers = {name: (d, np.ascontiguousarray(o)) for name, (d, o) in headers.items()}
        v.data = list(self.data)
        v.offsets = [np.ascontiguousarray(o[lo:hi + 1]) for o in self.offsets]
        v.ip, v.ip_is_v6, v.port, v.flags = self.ip[lo:hi], self.ip_is_v6[lo:hi], self.port[lo:hi], self.flags[lo:hi]
        v.asn = None if self.asn is None else self.asn[lo:hi]
        v.country = None if self.country is None else self.country[lo:hi]
        return v

    def algorithmic_bytes(self) -> int:
        """sum(field bytes) + 4*(5+1) offset bytes + 22 B numerics + 8 B verdict per request
        (+6 B when GeoIP is precomputed on the host)."""
        strings = sum(int(o[-1]) - int(o[0]) for o in self.offsets) + sum(int(o[-1]) - int(o[0]) for _, o in self.headers.values())
        per_req = 24 + 22 + 8 + (6 if self.asn is not None else 0) + 4 * len(self.headers)
        return strings + per_req * self.n

    def as_struct(self, header_names: Sequence[str] = ()) -> _abi.Batch:
        """pwaf_batch over the arrays; header columns in the order of `header_names` (a name the batch does not carry = empty)."""
        b = _abi.Batch()
        b.struct_size = C.sizeof(_abi.Batch)
        b.n = self.n
        b.memory = _abi.MEM_HOST
        if header_names:
            cols = (_abi.StrCol * len(header_names))()
            hb = (C.c_uint32 * len(header_names))()
            empty = (np.zeros(_abi.ARENA_PAD, dtype=np.uint8), np.zeros(self.n + 1, dtype=np.uint32))
            keep = [cols, hb, empty]
            for k, name in enumerate(header_names):
                d, o = self.headers.get(name, empty)
                cols[k].data = d.ctypes.data
                cols[k].offsets = o.ctypes.data
                hb[k] = int(o[-1])
            b.n_headers = len(header_names)
            b.headers = cols
            b.header_bytes = hb
            b._keep = keep  # (ctypes does not keep the arrays alive by itself)
        for f in range(_abi.N_FIELDS):
            b.field[f].data = self.data[f].ctypes.data
            b.field[f].offsets = self.offsets[f].ctypes.data
        b.ip = self.ip.ctypes.data
        b.ip_is_v6 = self.ip_is_v6.ctypes.data
        b.port = self.port.ctypes.data
        b.flags = self.flags.ctypes.data
        b.asn = None if self.asn is None else self.asn.ctypes.data
        b.country = None if self.country is None else self.country.ctypes.data
        return b


def geoip_entries(rows: Iterable[tuple[str, int, str]]) -> np.ndarray:
    """rows of (cidr, asn, country) -> GEOIP_DTYPE array."""
    rows = list(rows)
    out = np.zeros(len(rows), dtype=GEOIP_DTYPE)
    for i, (cidr, asn, country) in enumerate(rows):
        net = ipaddress.ip_network(cidr, strict=False)
        packed = net.network_address.packed
        out[i]["addr"][: len(packed)] = np.frombuffer(packed, dtype=np.uint8)
        out[i]["prefix_len"] = net.prefixlen
        out[i]["is_v6"] = net.version == 6
        out[i]["country"] = np.frombuffer(_b(country)[:2].ljust(2, b"?"), dtype=np.uint8)
        out[i]["asn"] = asn
    return out
