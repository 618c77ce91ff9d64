"""TEST-ONLY writer of MaxMind DB files (format specification v2.0), used to feed `pwaf_geoip_from_mmdb`.

Independent of the reader in pingoo_amd/csrc/loaders.cpp (different language, tree built top-down here and walked there); it can
emit the three record sizes, IPv4 and IPv6 trees, extended size encodings, and pointers for repeated keys / values (the two places
real databases use them)."""
from __future__ import annotations

import ipaddress
import struct

MARKER = b"\xab\xcd\xefMaxMind.com"


def _ctrl(type_: int, size: int) -> bytes:
    """control byte(s) for a non-pointer field of `type_` with payload size / entry count `size`."""
    if size < 29:
        s, extra = size, b""
    elif size < 285:
        s, extra = 29, bytes([size - 29])
    elif size < 65821:
        s, extra = 30, struct.pack(">H", size - 285)
    else:
        s, extra = 31, struct.pack(">I", size - 65821)[1:]
    if type_ <= 7:
        return bytes([(type_ << 5) | s]) + extra
    return bytes([s, type_ - 7]) + extra  # extended type: type bits 0, next byte = type - 7


def _pointer(off: int) -> bytes:
    if off < 2048:
        return bytes([(1 << 5) | (0 << 3) | (off >> 8), off & 0xFF])
    if off < 526336:
        v = off - 2048
        return bytes([(1 << 5) | (1 << 3) | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if off < 134744064:
        v = off - 526336
        return bytes([(1 << 5) | (2 << 3) | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([(1 << 5) | (3 << 3)]) + struct.pack(">I", off)


class DataSection:
    def __init__(self, use_pointers: bool):
        self.buf = bytearray()
        self.use_pointers = use_pointers
        self.seen = {}  # encoded scalar -> offset

    def _scalar(self, v) -> bytes:
        if isinstance(v, bool):
            return _ctrl(14, int(v))
        if isinstance(v, str):
            b = v.encode()
            return _ctrl(2, len(b)) + b
        if isinstance(v, bytes):
            return _ctrl(4, len(v)) + v
        if isinstance(v, int):
            if v < 0:
                return _ctrl(8, 4) + struct.pack(">i", v)
            b = v.to_bytes((v.bit_length() + 7) // 8, "big")
            return _ctrl(5 if len(b) <= 2 else 6 if len(b) <= 4 else 9, len(b)) + b
        if isinstance(v, float):
            return _ctrl(3, 8) + struct.pack(">d", v)
        raise TypeError(type(v))

    def encode(self, v) -> bytes:
        """encoding of v as it appears INLINE; repeated strings become pointers to their first standalone copy."""
        if isinstance(v, dict):
            out = _ctrl(7, len(v))
            for k, x in v.items():
                out += self.encode(k) + self.encode(x)
            return out
        if isinstance(v, list):
            out = _ctrl(11, len(v))
            for x in v:
                out += self.encode(x)
            return out
        enc = self._scalar(v)
        if self.use_pointers and isinstance(v, str):
            if enc not in self.seen:
                self.seen[enc] = len(self.buf)
                self.buf += enc  # standalone copy other records can point to
            return _pointer(self.seen[enc])
        return enc

    def add_record(self, rec) -> int:
        enc = self.encode(rec)  # (may append standalone strings first)
        off = len(self.buf)
        self.buf += enc
        return off


def write_mmdb(networks, ip_version=4, record_size=24, use_pointers=False, extra_metadata=None) -> bytes:
    """networks: list of (cidr string, record) — record is any encodable value (normally {"asn": "AS1", "country": "FR"}).
    IPv4 networks in an ip_version 6 tree are stored below ::/96, like real databases do."""
    bits_total = 32 if ip_version == 4 else 128
    data = DataSection(use_pointers)
    root = {}  # node: {0: child|("data", off), 1: ...}
    for cidr, rec in networks:
        net = ipaddress.ip_network(cidr, strict=False)
        if net.version == 4 and ip_version == 6:
            value, plen = int(net.network_address), net.prefixlen + 96
        elif net.version == ip_version:
            value, plen = int(net.network_address), net.prefixlen
        else:
            raise ValueError("IPv6 network in an IPv4 tree")
        off = data.add_record(rec)
        node = root
        if plen == 0:
            raise ValueError("/0 cannot be represented as a single record")
        for d in range(plen):
            bit = (value >> (bits_total - 1 - d)) & 1
            if d == plen - 1:
                node[bit] = ("data", off)
            else:
                nxt = node.get(bit)
                if not isinstance(nxt, dict):
                    nxt = {}
                    node[bit] = nxt
                node = nxt
    # number the nodes breadth-first
    nodes, queue = [], [root]
    index = {id(root): 0}
    while queue:
        n = queue.pop(0)
        nodes.append(n)
        for b in (0, 1):
            c = n.get(b)
            if isinstance(c, dict):
                index[id(c)] = len(nodes) + len(queue)
                queue.append(c)
    node_count = len(nodes)

    def rec_value(c):
        if c is None:
            return node_count
        if isinstance(c, dict):
            return index[id(c)]
        return node_count + 16 + c[1]
    tree = bytearray()
    for n in nodes:
        l, r = rec_value(n.get(0)), rec_value(n.get(1))
        if record_size == 24:
            tree += l.to_bytes(3, "big") + r.to_bytes(3, "big")
        elif record_size == 28:
            tree += (l & 0xFFFFFF).to_bytes(3, "big") + bytes([((l >> 24) << 4) | (r >> 24)]) + (r & 0xFFFFFF).to_bytes(3, "big")
        else:
            tree += l.to_bytes(4, "big") + r.to_bytes(4, "big")
    meta = {"binary_format_major_version": 2, "binary_format_minor_version": 0, "build_epoch": 1700000000, "database_type": "pingoo-geoip-test",
            "description": {"en": "test database"}, "ip_version": ip_version, "languages": ["en"], "node_count": node_count, "record_size": record_size}
    if extra_metadata:
        meta.update(extra_metadata)
    return bytes(tree) + b"\0" * 16 + bytes(data.buf) + MARKER + DataSection(False).encode(meta)
