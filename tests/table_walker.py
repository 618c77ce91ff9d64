"""TEST-ONLY interpreter of the compiled tables that `pwaf_program_dump` exports.

It re-implements, in plain Python and one request at a time, what the HIP kernels do with the
tables (DFA walk with emit/end lists, numeric atoms, trie lookups, DNF rule evaluation, first match
wins). It exists so that the HOST COMPILER (parser, typing, DFA construction, trie construction,
DNF) can be checked against the oracle on CPU-only machines. It is test infrastructure: the product
never imports it, and it is far too slow to be anyone's fallback.
"""
from __future__ import annotations

import struct

import numpy as np

from pingoo_amd import _abi

LIT_NEG = 1 << 30
LIT_TERM_END = 1 << 31
LIT_ATOM_MASK = (1 << 24) - 1
TRIE_LEAF = 0x80000000
ATOM_LEN, ATOM_INT, ATOM_INTSET, ATOM_IPSET, ATOM_COUNTRY = 2, 3, 4, 5, 6
OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE = range(6)

NUMA_DTYPE = np.dtype([("col", "<u4"), ("kind", "u1"), ("var", "u1"), ("op", "u1"), ("pad", "u1"), ("ref", "<u4"), ("ref2", "<u4"), ("c", "<i8")])
RULE_DTYPE = np.dtype([("lit_off", "<u4"), ("lit_cnt", "<u4"), ("public_idx", "<u4"), ("eff_u", "u1"), ("eff_v", "u1"), ("pad", "u1", (2,))])
GREC_DTYPE = np.dtype([("asn", "<u4"), ("country", "<u2"), ("pad", "<u2")])
assert NUMA_DTYPE.itemsize == 24 and RULE_DTYPE.itemsize == 16 and GREC_DTYPE.itemsize == 8


def parse_dump(blob: bytes):
    assert blob[:8] == b"PWAFPRG1", "bad dump magic"
    pos = 8
    sections = []
    while pos < len(blob):
        tag = blob[pos:pos + 4].decode()
        count, = struct.unpack_from("<I", blob, pos + 4)
        length, = struct.unpack_from("<Q", blob, pos + 8)
        payload = blob[pos + 16:pos + 16 + length]
        pos += 16 + length
        pos = (pos + 7) // 8 * 8
        sections.append((tag, count, payload))
    return sections


def scalar_class(um: bytes, data: bytes, j: int) -> int:
    """csrc/utf8.h restated: the class of the scalar value whose lead byte is data[j] (the ill-formed class when the bytes from j on
    are not one well-formed sequence inside the field)."""
    ill, img = um[0], um[8:]
    b0 = data[j]
    ln = 4 if b0 >= 0xF0 else 3 if b0 >= 0xE0 else 2
    if b0 < 0xC2 or b0 > 0xF4 or len(data) - j < ln:
        return ill
    try:
        cp = ord(bytes(data[j:j + ln]).decode("utf-8"))
    except (UnicodeDecodeError, TypeError):
        return ill
    block = img[2 * (cp >> 7)] | (img[2 * (cp >> 7) + 1] << 8)
    return img[2 * (0x110000 >> 7) + block * 128 + (cp & 127)]


def cont_covered(data: bytes, j: int) -> bool:
    """csrc/utf8.h utf8_cont_covered restated: the continuation byte data[j] belongs to a well-formed sequence of the field (then the walker
    stays: the sequence was read at its lead byte); otherwise it is an ill-formed unit like any other (round 6)."""
    for d in (1, 2, 3):
        if d > j:
            return False
        lead = data[j - d]
        if lead < 0x80:
            return False
        if lead < 0xC0:
            continue
        ln = 4 if lead >= 0xF0 else 3 if lead >= 0xE0 else 2
        if ln <= d or lead < 0xC2 or lead > 0xF4 or j - d + ln > len(data):
            return False
        try:
            bytes(data[j - d:j - d + ln]).decode("utf-8")
            return True
        except UnicodeDecodeError:
            return False
    return False


class Tables:
    def __init__(self, blob):
        """`blob`: a program dump, or the CompiledProgram itself (then the confirm tier of its filtered passes is evaluated too,
        through the program's test hook; from a bare dump a candidate is walked through the pass's full DFA, as PWAF_OPT_NO_CONFIRM does)."""
        if hasattr(blob, "dump"):
            self.program = blob
            blob = blob.dump()
        self.groups = []
        cur = None
        for tag, count, pl in parse_dump(blob):
            if tag == "HEAD":
                (self.n_cols, self.n_scan_cols, self.n_groups, self.n_rules, self.n_ip_lists, self.set_words, self.has_geo, self.flags) = struct.unpack("<8I", pl)
            elif tag == "GHDR":
                f, ns, nc, _, _, ab, nl, na = struct.unpack("<8I", pl)
                cur = dict(field=f, n_states=ns, n_classes=nc, atom_base=ab, n_local=nl, n_atoms=na)
                self.groups.append(cur)
            elif tag == "GCLS":
                cur["classmap"] = np.frombuffer(pl, dtype=np.uint8)
            elif tag == "GUMP":
                cur["umap"] = pl  # byte 0: the class of an ill-formed byte; from byte 8: [stage1 u16 x 8704][stage2]
            elif tag == "RUMP":
                cur["rtier"]["umap"] = pl
            elif tag == "GTRN":
                cur["trans"] = np.frombuffer(pl, dtype="<u2").reshape(cur["n_states"], cur["n_classes"])
            elif tag == "GEMO":
                cur["emit_off"] = np.frombuffer(pl, dtype="<u4")
            elif tag == "GEML":
                cur["emit_list"] = np.frombuffer(pl, dtype="<u2")
            elif tag == "GENO":
                cur["end_off"] = np.frombuffer(pl, dtype="<u4")
            elif tag == "GENL":
                cur["end_list"] = np.frombuffer(pl, dtype="<u2")
            elif tag == "GFHD":
                init, n_heads = struct.unpack_from("<II", pl, 0)
                cur["f_init"] = init
                cur["f_mul"], cur["f_stride"], n_heads = n_heads >> 16, (n_heads >> 8) & 0xFF, n_heads & 0xFF
                cur["f_heads"] = [(pl[8 + 20 * k:8 + 20 * k + 16][:pl[8 + 20 * k + 16]], pl[8 + 20 * k + 17], struct.unpack_from("<H", pl, 8 + 20 * k + 18)[0])
                                  for k in range(n_heads)]  # (literal, exact, local atom)
            elif tag == "GFTB":
                cur["f_table"] = np.frombuffer(pl, dtype="<u4")
            elif tag == "GFLT":
                cur["filter_cols"] = [int(x) for x in np.frombuffer(pl, dtype="<u4")]
            elif tag == "GCNF":
                cur["confirm"], cur["confirm_walk"], cur["confirm_entries"], cur["confirm_literals"] = struct.unpack("<4I", pl)
            elif tag == "RHDR":
                ns, nc = struct.unpack("<2I", pl)
                cur["rtier"] = dict(n_states=ns, n_classes=nc, atom_base=cur["atom_base"])
            elif tag == "RCLS":
                cur["rtier"]["classmap"] = np.frombuffer(pl, dtype=np.uint8)
            elif tag == "RTRN":
                cur["rtier"]["trans"] = np.frombuffer(pl, dtype="<u2").reshape(cur["rtier"]["n_states"], cur["rtier"]["n_classes"])
            elif tag in ("REMO", "RENO"):
                cur["rtier"]["emit_off" if tag == "REMO" else "end_off"] = np.frombuffer(pl, dtype="<u4")
            elif tag in ("REML", "RENL"):
                cur["rtier"]["emit_list" if tag == "REML" else "end_list"] = np.frombuffer(pl, dtype="<u2")
            elif tag == "FCMP":
                self.fcmp = np.frombuffer(pl, dtype=np.dtype([("col", "<u4"), ("op", "u1"), ("a", "u1"), ("b", "u1"), ("pad", "u1")]))
            elif tag == "HDRS":
                self.header_names = [x.decode() for x in pl.split(b"\0")[:count]]
            elif tag == "NUMA":
                self.num_atoms = np.frombuffer(pl, dtype=NUMA_DTYPE)
            elif tag == "INTP":
                self.int_pool = np.frombuffer(pl, dtype="<i8")
            elif tag == "CLUT":
                self.country_luts = np.frombuffer(pl, dtype="<u4")
            elif tag == "RULE":
                self.rules = np.frombuffer(pl, dtype=RULE_DTYPE)
            elif tag == "LITS":
                self.lits = np.frombuffer(pl, dtype="<u4")
            elif tag == "SETM":
                self.set_masks = np.frombuffer(pl, dtype="<u4")
            elif tag in ("IR4 ", "IR6 ", "INOD", "GR4 ", "GR6 ", "GNOD"):
                setattr(self, tag.strip().lower(), np.frombuffer(pl, dtype="<u4"))
            elif tag == "GREC":
                self.geo_recs = np.frombuffer(pl, dtype=GREC_DTYPE)
            elif tag == "RSDL":
                self.n_residual, self.residual_base = struct.unpack("<2I", pl)
            elif tag == "RVMB":
                self.residual_blob = bytes(pl)

    # --- pieces ---
    FILTER_MUL = 0x9E37

    @staticmethod
    def filter_bin(b0: int, b1: int, mul: int) -> int:
        f0, f1 = b0 & ~((b0 >> 1) & 0x20), b1 & ~((b1 >> 1) & 0x20)  # program.h: filter_fold (bit 5 cleared where bit 6 is set)
        return (((f0 | (f1 << 8)) * mul) & 0xFFFF) >> 4

    def filter_flags(self, g: dict, data: bytes):
        """The bigram prefilter of a pass exactly as filter_kernel applies it to ONE field value inside an arena (csrc/filter.cpp:
        filter_field_positions): the field starts A = arena_offset + filter_phase + 16 arena_chunks bytes into an arena whose other
        bytes are '~' (what the test hook pwaf_program_confirm_field builds around it). The state at the field's first sampled
        position is what the (up to four) sampled bigrams in front of it left — the init state at the arena's start — and the last
        position pairs the field's last byte with the byte behind it: windows of short factors reach one bigram beyond the factor.
        Returns the 16-byte arena chunks holding a position of the field that completed a window."""
        tab, flags, s = g["f_table"], set(), g["f_stride"]
        A = self.arena_offset + self.filter_phase + 16 * self.arena_chunks
        arena = b"~" * A + bytes(data) + b"~" * 32
        fs, fe = A, A + len(data)
        i0 = (fs + 1) & ~1 if s == 2 else fs
        st = g["f_init"]
        for i in range(i0 - 4 * s if i0 >= 4 * s else i0 % s, fe, s):
            st = ((st << 8) | int(tab[self.filter_bin(arena[i], arena[i + 1], g["f_mul"])])) & 0xFFFFFFFF
            if i >= i0 and (~st) & 0xFF000000:
                flags.add(i // 16)
        return flags

    def filter_candidate(self, g: dict, data: bytes) -> bool:
        return bool(self.filter_flags(g, data))

    def scan_pass(self, g: dict, data: bytes, cols: set, walked: set = None):
        """One pass as the device runs it: behind its prefilter (heads + DFA for candidates only) when it has one; a GATED gap pass
        only visits a request when one of its prefilter factors was found by a DFA WALK of the owning pass — that walk is what
        enqueues the request (a column set by a filter head enqueues nothing). `walked` collects the columns set by walks."""
        walked = walked if walked is not None else set()
        if g.get("filter_cols") and self.use_gates:
            if not (set(g["filter_cols"]) & walked):
                return
            self.n_gated_walks += 1
        if "f_table" not in g or not self.use_filter:
            mine = set()
            self.scan_field(g, data, mine)
            cols |= mine
            walked |= mine
            return
        # heads are compared by the filter kernel for every request
        for lit, exact, local in g["f_heads"]:
            if data[:len(lit)] == lit and (not exact or len(data) == len(lit)):
                cols.add(g["atom_base"] + local)
        flags = self.filter_flags(g, data)
        if flags and g.get("confirm") and self.program is not None and self.use_confirm:
            # the confirm tier (run by csrc/confirm.h through the test hook: the code the device compiles): literal atoms decided at the
            # flagged chunks; the request is walked — through the DFA of the pass's NON-literal atoms — only when a regex factor was confirmed
            self.n_candidates += 1
            atoms, flagged, walk = self.program.confirm_field(self.groups.index(g), data, self.arena_offset + self.filter_phase + 16 * self.arena_chunks)  # (filter_phase: the tests' way of saying "one byte further into the arena")
            assert flagged
            mine = {g["atom_base"] + a for a in atoms}
            self.n_confirm_hits += len(mine)
            if walk:
                self.n_confirm_walks += 1
                self.scan_field(g.get("rtier", g), data, mine)
            cols |= mine
            walked |= mine
        elif flags:
            self.n_candidates += 1
            mine = set()
            self.scan_field(g, data, mine)
            cols |= mine
            walked |= mine

    use_filter = True
    use_confirm = True
    program = None        # the CompiledProgram the dump came from (the confirm tier runs through its test hook)
    arena_chunks = 0      # whole 16-byte chunks in front of the field in the hook's arena (tests vary it: chunk-bitmap word boundaries)
    n_confirm_hits = 0
    n_confirm_walks = 0
    use_gates = True
    n_gated_walks = 0
    filter_phase = 0  # offset of the first sampled byte of a field (the device: parity of the field's arena offset, stride-2 passes)
    arena_offset = 0  # where the field starts in its arena, modulo 16 (chunk boundaries) — tests vary it
    n_candidates = 0
    n_steps = 0       # DFA steps taken by filtered passes

    def scan_field(self, g: dict, data: bytes, cols: set):
        """Walks one field through group g's DFA (the pass's, or its R tier's), adding the device column ids that hold."""
        st = 0  # states are in BFS order from the start state

        def emit(s):
            for a in g["emit_list"][g["emit_off"][s]:g["emit_off"][s + 1]]:
                cols.add(g["atom_base"] + int(a))
        emit(st)
        cm, tr, um = g["classmap"], g["trans"], g.get("umap")
        for j in range(len(data)):
            c = int(cm[data[j]])
            if um is not None and data[j] >= 0xC0:
                c = scalar_class(um, data, j)  # scalar mode: the class of the scalar value that begins here (csrc/utf8.h)
            stays = um is not None and 0x80 <= data[j] < 0xC0 and cont_covered(data, j)  # (a continuation byte of a well-formed sequence stays and emits nothing)
            if um is not None and 0x80 <= data[j] < 0xC0 and not stays:
                c = um[0]  # a stray continuation byte: the ill-formed class
            nxt = int(tr[st, c])
            if not stays:
                st = nxt
                emit(st)
            self.n_steps += 1
        for a in g["end_list"][g["end_off"][st]:g["end_off"][st + 1]]:
            cols.add(g["atom_base"] + int(a))

    @staticmethod
    def trie_lookup(root, nodes, ip: bytes) -> int:
        if root is None or len(root) == 0:
            return 0
        e = int(root[(ip[0] << 8) | ip[1]])
        k = 2
        while not (e & TRIE_LEAF):
            e = int(nodes[e * 256 + ip[k]])
            k += 1
        return e & ~TRIE_LEAF

    def geo(self, ip: bytes, v6: bool):
        asn, country = 0, b"XX"
        if not self.has_geo:
            return asn, country
        if not v6:
            if ip[0] == 127 or (ip[0] & 0xF0) == 0xE0:
                return asn, country
        else:
            if ip[:15] == b"\0" * 15 and ip[15] == 1:
                return asn, country
            if ip[0] == 0xFF:
                return asn, country
        rec = self.trie_lookup(self.gr6 if v6 else self.gr4, self.gnod, ip)
        r = self.geo_recs[rec]
        return int(r["asn"]), int(r["country"]).to_bytes(2, "little")

    def evaluate(self, batch, i: int):
        """Returns (action, rule_idx) for request i of a RequestBatch, exactly as the device pipeline would."""
        cols = {0}
        fields = [batch.field_bytes(f, i) for f in range(5)] + [batch.header_bytes(h, i) for h in getattr(self, "header_names", [])]
        walked = set()
        for g in self.groups:
            self.scan_pass(g, fields[g["field"]], cols, walked)
        for d in getattr(self, "fcmp", []):  # one field against another
            x, y, op = fields[int(d["a"])], fields[int(d["b"])], int(d["op"])
            if [x == y, y in x, x.startswith(y), x.endswith(y), len(x) == len(y), len(x) < len(y), len(x) <= len(y)][op]:
                cols.add(int(d["col"]))
        ip = batch.ip[i].tobytes()
        v6 = bool(batch.ip_is_v6[i])
        port = int(batch.port[i])
        verified = bool(batch.flags[i] & _abi.FLAG_CAPTCHA_VERIFIED)
        if batch.asn is not None:
            asn, country = int(batch.asn[i]), int(batch.country[i]).to_bytes(2, "little")
        else:
            asn, country = self.geo(ip, v6)
        c0, c1 = country[0] - 65, country[1] - 65
        cidx = c0 * 26 + c1 if 0 <= c0 < 26 and 0 <= c1 < 26 else 23 * 26 + 23
        set_id = self.trie_lookup(self.ir6 if v6 else self.ir4, self.inod, ip) if self.n_ip_lists else 0
        for d in self.num_atoms:
            kind, var, op, c = int(d["kind"]), int(d["var"]), int(d["op"]), int(d["c"])
            t = False
            if kind in (ATOM_LEN, ATOM_INT):
                v = len(fields[var]) if kind == ATOM_LEN else (port if var == 0 else asn)
                t = {OP_EQ: v == c, OP_LT: v < c, OP_LE: v <= c}[op]
            elif kind == ATOM_INTSET:
                v = port if var == 0 else asn
                t = v in set(int(x) for x in self.int_pool[int(d["ref"]):int(d["ref2"])])
            elif kind == ATOM_IPSET:
                r = int(d["ref"])
                t = bool((int(self.set_masks[set_id * self.set_words + (r >> 5)]) >> (r & 31)) & 1)
            elif kind == ATOM_COUNTRY:
                t = bool((int(self.country_luts[int(d["ref"]) * 22 + (cidx >> 5)]) >> (cidx & 31)) & 1)
            if t:
                cols.add(int(d["col"]))
        if getattr(self, "n_residual", 0):
            # residual rules: the program image compile.cpp produced, run by the TEST-ONLY host build of the interpreter
            # (tests/rvm_host.cpp — the same residual.h the device kernel compiles)
            import test_residual as TR

            if getattr(self, "_rvm", None) is None:
                self._rvm = TR.HostVM.from_blob(self.residual_blob, getattr(self, "header_names", []))
            if getattr(self, "_rvm_batch", None) is not batch:
                self._rvm.bind(batch)
                self._rvm_batch = batch
            geo_asn, geo_cc = (asn, country)
            for k in range(self.n_residual):
                if self._rvm.eval(k, i, asn=geo_asn, country=int.from_bytes(geo_cc, "little")):
                    cols.add(self.residual_base + k)
        for r in self.rules:
            acc_or, acc_and = False, True
            for k in range(int(r["lit_off"]), int(r["lit_off"]) + int(r["lit_cnt"])):
                lit = int(self.lits[k])
                v = (lit & LIT_ATOM_MASK) in cols
                if lit & LIT_NEG:
                    v = not v
                acc_and = acc_and and v
                if lit & LIT_TERM_END:
                    acc_or = acc_or or acc_and
                    acc_and = True
            if acc_or:
                eff = int(r["eff_v"] if verified else r["eff_u"])
                if eff:
                    return eff, int(r["public_idx"])
        return _abi.ACTION_ALLOW, _abi.RULE_NONE
