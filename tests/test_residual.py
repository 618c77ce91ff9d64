"""The residual interpreter (pingoo_amd/csrc/residual.h + residual.cpp) on the CPU.

Rules the column compiler cannot take — arithmetic on request values, concatenation, lists / maps holding request values, orderings
between request values, conditionals selecting non-Bool values — are lowered whole to a stack program that residual_kernel interprets
per request on the device (pingoo/rules.rs:37-51 evaluates ANY valid expression). The interpreter is one header shared by the HIP
kernel and by a TEST-ONLY host build (tests/rvm_host.cpp, built here with g++), so its semantics are fuzzed against the oracle
without a GPU: for every expression and request, "the stack program ends in Bool(true)" must equal "the oracle's execute returns
Bool(true)". The device path is covered by tests/test_gpu_residual.py."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "pingoo_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "librvm_host.so")
SRCS = [os.path.join(HERE, "rvm_host.cpp")] + [os.path.join(CSRC, f) for f in ("residual.cpp", "residual_jit.cpp", "frontend.cpp", "pattern.cpp", "dfa.cpp", "iptrie.cpp")]
DEPS = SRCS + [os.path.join(CSRC, f) for f in ("residual.h", "program.h", "frontend.h", "utf8.h")]


def build_host_vm() -> str:
    os.makedirs(BUILD, exist_ok=True)
    from pingoo_amd.build import embed
    embed()  # (csrc/residual_h.inc: residual_jit.cpp carries residual.h as text)
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I", os.path.join(HERE, "..", "include"), *SRCS, "-o", LIB]
        subprocess.run(cmd, check=True)
    return LIB


_L = None


def vm():
    global _L
    if _L is None:
        L = C.CDLL(build_host_vm())
        L.rvmh_compile.restype = C.c_void_p
        L.rvmh_compile.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(_abi.ListDesc), C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        L.rvmh_free.argtypes = [C.c_void_p]
        L.rvmh_from_blob.restype = C.c_void_p
        L.rvmh_from_blob.argtypes = [C.c_char_p, C.c_size_t]
        L.rvmh_header_count.restype = C.c_size_t
        L.rvmh_header_count.argtypes = [C.c_void_p]
        L.rvmh_header_name.restype = C.c_char_p
        L.rvmh_header_name.argtypes = [C.c_void_p, C.c_size_t]
        L.rvmh_specialize.restype = C.c_size_t
        L.rvmh_specialize.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.rvmh_eval.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        _L = L
    return _L


class HostVM:
    """n expressions compiled as residual rules; eval(rule, batch, i) -> bool. Raises ValueError(reason) when a rule cannot be lowered."""

    def __init__(self, exprs, lists=None):
        self._m = _abi.Marshalled()
        l, nl = _abi.marshal_lists(lists, self._m)
        arr = (C.c_char_p * len(exprs))(*[e.encode() for e in exprs])
        why = C.create_string_buffer(400)
        bad = C.c_int(-1)
        self._h = vm().rvmh_compile(arr, len(exprs), l, nl, why, 400, C.byref(bad))
        if not self._h:
            raise ValueError((bad.value, why.value.decode(errors="replace")))
        self.header_names = [vm().rvmh_header_name(self._h, k).decode() for k in range(vm().rvmh_header_count(self._h))]

    @classmethod
    def from_blob(cls, blob: bytes, header_names):
        """The program image of a compiled rule set (pwaf_program_dump section RVMB); string columns = 5 fields + header_names."""
        self = cls.__new__(cls)
        self._h = vm().rvmh_from_blob(blob, len(blob))
        self.header_names = list(header_names)
        return self

    def specialized_source(self) -> str:
        """The rules as straight-line C++ over residual.h (csrc/residual_jit.cpp)."""
        why = C.create_string_buffer(400)
        n = vm().rvmh_specialize(self._h, None, 0, why, 400)
        if n == 0:
            raise ValueError(why.value.decode(errors="replace"))
        buf = C.create_string_buffer(n + 1)
        vm().rvmh_specialize(self._h, buf, n + 1, why, 400)
        return buf.value.decode()

    def __del__(self):
        if getattr(self, "_h", None):
            vm().rvmh_free(self._h)
            self._h = None

    def bind(self, batch: RequestBatch):
        cols = list(zip(batch.data, batch.offsets))
        zero = (np.zeros(_abi.ARENA_PAD, dtype=np.uint8), np.zeros(batch.n + 1, dtype=np.uint32))
        for name in self.header_names:
            cols.append(batch.headers.get(name, zero))
        self._keep = cols
        self._data = (C.c_void_p * len(cols))(*[d.ctypes.data for d, _ in cols])
        self._off = (C.c_void_p * len(cols))(*[o.ctypes.data for _, o in cols])
        self._batch = batch

    def eval3(self, rule: int, i: int, asn=None, country=None) -> int:
        """1 = the rule matches (Bool(true)), 2 = its evaluation ends in an execution error, 0 = anything else."""
        b = self._batch
        if asn is None:
            asn = int(b.asn[i]) if b.asn is not None else 0
            country = int(b.country[i]) if b.country is not None else int.from_bytes(b"XX", "little")
        return int(vm().rvmh_eval(self._h, rule, self._data, self._off, i, b.ip[i].ctypes.data, int(b.ip_is_v6[i]), int(b.port[i]), asn, country))

    def eval(self, rule: int, i: int, asn=None, country=None) -> bool:
        return self.eval3(rule, i, asn, country) == 1


# ---------------------------------------------------------------------------------------------------------------------
# expressions that compute with request values
# ---------------------------------------------------------------------------------------------------------------------
F = ["http_request.host", "http_request.url", "http_request.path", "http_request.method", "http_request.user_agent", 'http_request.headers["x-a"]', "http_request.headers.cookie"]
INTS = ["client.remote_port", "client.asn", "http_request.path.length()", "http_request.host.length()", "http_request.url.length()"]


def dstr(rng, depth=0):
    k = rng.randint(0, 9 if depth < 2 else 5)
    if k <= 2:
        return rng.choice(F)
    if k == 3:
        return H.q(H.rstr(rng, 0, 3))
    if k == 4:
        return "client.country"
    if k == 5:
        return rng.choice(['lists["words"][0]', 'lists.words[1]', '{"k": ' + rng.choice(F) + '}.k', '[' + rng.choice(F) + ', "ab"][' + str(rng.randint(0, 2)) + ']'])
    if k <= 7:
        return "(" + dstr(rng, depth + 1) + " + " + dstr(rng, depth + 1) + ")"
    if k == 8:
        return "(" + dbool(rng, depth + 1) + " ? " + dstr(rng, depth + 1) + " : " + dstr(rng, depth + 1) + ")"
    return rng.choice(F)


def dint(rng, depth=0):
    k = rng.randint(0, 10 if depth < 2 else 4)
    if k <= 2:
        return rng.choice(INTS)
    if k <= 4:
        return str(rng.choice([0, 1, 2, 3, 7, 80, 443, -1, 65535, 9223372036854775807, 2.5, 0.0, 1e3]))
    if k <= 7:
        return "(" + dint(rng, depth + 1) + " " + rng.choice(["+", "-", "*", "/", "%"]) + " " + dint(rng, depth + 1) + ")"
    if k == 8:
        return "(-" + dint(rng, depth + 1) + ")"
    if k == 9:
        return "(" + dbool(rng, depth + 1) + " ? " + dint(rng, depth + 1) + " : " + dint(rng, depth + 1) + ")"
    return rng.choice([dstr(rng, depth + 1) + ".length()", "[1, 2, 3][" + dint(rng, depth + 1) + " % 3]", 'lists["asns"][' + str(rng.randint(0, 2)) + "]",
                       '{"p": client.remote_port}["p"]', "[" + dint(rng, depth + 1) + ", 2].length()"])


def dbool(rng, depth=0):
    k = rng.randint(0, 19 if depth < 3 else 8)
    if k <= 1:
        return dint(rng, depth + 1) + " " + rng.choice(["==", "!=", "<", "<=", ">", ">="]) + " " + dint(rng, depth + 1)
    if k <= 3:
        return dstr(rng, depth + 1) + " " + rng.choice(["==", "!=", "<", "<=", ">", ">="]) + " " + dstr(rng, depth + 1)
    if k <= 5:
        return dstr(rng, depth + 1) + "." + rng.choice(["contains", "starts_with", "ends_with"]) + "(" + dstr(rng, depth + 1) + ")"
    if k == 6:
        if rng.random() < 0.3:  # round 6: a pattern computed from finitely many strings (a conditional, a list item, a concatenation of literals)
            pat = rng.choice(["(" + dbool(rng, depth + 2) + " ? " + H.q(H.rregex(rng)) + " : " + H.q(H.rregex(rng)) + ")", 'lists["words"][' + dint(rng, depth + 2) + " % 5]",
                              H.q(H.rregex(rng)) + " + " + H.q(H.rstr(rng, 0, 2)), "[" + H.q(H.rregex(rng)) + ", " + H.q(H.rregex(rng)) + "][client.remote_port % 2]"])
            return dstr(rng, depth + 1) + ".matches(" + pat + ")"
        return dstr(rng, depth + 1) + ".matches(" + H.q(H.rregex(rng)) + ")"
    if k == 7:
        items = ", ".join(rng.choice([dstr, dint])(rng, depth + 1) for _ in range(rng.randint(0, 3)))
        x = rng.choice([dstr, dint])(rng, depth + 1)
        return rng.choice([f"[{items}].contains({x})", f"{x} in [{items}]"])
    if k == 8:
        return rng.choice(['lists["words"].contains(' + dstr(rng, depth + 1) + ")", dint(rng, depth + 1) + ' in lists["asns"]', 'lists["nets"].contains(client.ip)', "client.ip in lists.nets2",
                           'lists["nets"][0] == lists["nets2"][0]', 'lists["nets"].contains(lists["nets2"][0])', "client.ip == client.ip", 'client.ip == "1.1.1.1"',
                           'lists.words.length() > ' + dint(rng, depth + 1), '"x-a" in http_request.headers', 'http_request.contains("host")', '"nope" in client', 'lists.contains("words")',
                           # COMPUTED keys into the context maps (round 4: the closed key sets of http_request / client / lists become Map values)
                           dstr(rng, depth + 1) + " in " + rng.choice(["client", "http_request", "lists"]), rng.choice(["client", "http_request", "lists"]) + ".contains(" + dstr(rng, depth + 1) + ")",
                           "client[" + rng.choice(['"remote_" + "port"', '"as" + "n"', dstr(rng, depth + 1)]) + "] == " + dint(rng, depth + 1),
                           "lists[" + rng.choice(['"wor" + "ds"', '"ne" + "ts"', dstr(rng, depth + 1)]) + "].contains(" + rng.choice([dstr(rng, depth + 1), "client.ip"]) + ")",
                           "client[" + dint(rng, depth + 1) + "] == 1", dint(rng, depth + 1) + " in lists", 'client["coun" + "try"] == client.country',
                           # round 5: http_request and the headers map under a COMPUTED key (the headers map's names are closed once every rule is read)
                           "http_request[" + rng.choice(['"ho" + "st"', '"pa" + "th"', dstr(rng, depth + 1)]) + "] == " + dstr(rng, depth + 1),
                           "http_request.headers[" + rng.choice(['"x-" + "a"', '"coo" + "kie"', dstr(rng, depth + 1)]) + "] == " + dstr(rng, depth + 1),
                           dstr(rng, depth + 1) + " in http_request.headers", "http_request.headers.contains(" + dstr(rng, depth + 1) + ")",
                           'http_request["head" + "ers"]["x-a"] == ' + dstr(rng, depth + 1), "http_request.headers.length() == " + str(rng.randint(0, 3)),
                           'http_request[' + dstr(rng, depth + 1) + '].length() > 3', "http_request[" + dint(rng, depth + 1) + "] == 1"])
    if k == 9:
        return "!(" + dbool(rng, depth + 1) + ")"
    if k <= 12:
        return "(" + dbool(rng, depth + 1) + rng.choice([" && ", " || "]) + dbool(rng, depth + 1) + ")"
    if k == 13:
        return "(" + dbool(rng, depth + 1) + " ? " + dbool(rng, depth + 1) + " : " + dbool(rng, depth + 1) + ")"
    if k == 14:
        m = "{" + ", ".join(f"{rng.choice([H.q(H.rstr(rng, 1, 2)), dstr(rng, depth + 1)])}: {rng.choice([dstr, dint])(rng, depth + 1)}" for _ in range(rng.randint(0, 3))) + "}"
        return rng.choice([f"{m}.contains({dstr(rng, depth + 1)})", f"{dstr(rng, depth + 1)} in {m}", f"{m}.length() == {rng.randint(0, 3)}", f"{m} == {m}", f'{m}["a"] == {dstr(rng, depth + 1)}'])
    if k == 15:
        l1 = "[" + ", ".join(rng.choice([dstr, dint])(rng, depth + 1) for _ in range(rng.randint(0, 3))) + "]"
        l2 = "[" + ", ".join(rng.choice([dstr, dint])(rng, depth + 1) for _ in range(rng.randint(0, 3))) + "]"
        return rng.choice([f"{l1} == {l2}", f"({l1} + {l2}).length() == {rng.randint(0, 6)}", f"[{l1}, {l2}].contains({l1})", f"({l1} + {l2}).contains({dstr(rng, depth + 1)})"])
    if k == 16:  # ill-typed on purpose: errors must propagate exactly like in the oracle
        return rng.choice([dstr(rng, depth + 1) + " + 1 == 2", "!" + dint(rng, depth + 1), dstr(rng, depth + 1) + ".length(1) == 1", dint(rng, depth + 1) + ".contains(1)",
                           dstr(rng, depth + 1) + ".bogus()", "bogus(" + dint(rng, depth + 1) + ")", "-" + dstr(rng, depth + 1) + " == 1", dint(rng, depth + 1) + " < " + dstr(rng, depth + 1),
                           "(" + dint(rng, depth + 1) + " ? true : false)", dstr(rng, depth + 1) + " && true", "true || " + dint(rng, depth + 1), "false || " + dint(rng, depth + 1),
                           "[1][" + dint(rng, depth + 1) + "] == 1", '{"a": 1}.b == 1', dstr(rng, depth + 1) + ".matches(\"(\")", "null == null", dstr(rng, depth + 1) + " in 5"])
    if k == 17:
        return H.rpred(rng, LISTS)
    if k == 18:
        return rng.choice(["true", "false"])
    return dint(rng, depth + 1) + " == " + dint(rng, depth + 1)


LISTS = {
    "nets": (_abi.LIST_IP, ["1.0.0.0/8", "2.2.2.2", "2001:db8::/32", "3.3.0.0/255.255.0.0"]),
    "nets2": (_abi.LIST_IP, ["1.0.0.0/8", "9.9.9.9/32"]),
    "words": (_abi.LIST_STRING, ["ab", " /a ", "b.", "", "abab"]),
    "asns": (_abi.LIST_INT, ["1", "2", " 64512 ", "3"]),
}


def requests(rng, n):
    reqs = H.fuzz_requests(rng, n, with_geo=True)
    for r in reqs:
        if rng.random() < 0.5:
            r.headers = {h: H.rstr(rng, 0, 5) for h in ("x-a", "cookie") if rng.random() < 0.7}
        if not r.user_agent:
            r.user_agent = "ua"
    return reqs


@pytest.mark.parametrize("seed", range(60))
def test_residual_programs_agree_with_the_oracle(seed):
    rng = random.Random(424200 + seed)
    exprs = [dbool(rng) for _ in range(12)]
    batch = RequestBatch.from_requests(requests(rng, 40))
    rejected = 0
    for e in exprs:
        try:
            pyoracle.compile_expression(e)
        except pyoracle.OracleError:
            continue  # (the generator can produce a syntax error through the regex literals: not this test's subject)
        try:
            m = HostVM([e], LISTS)
        except ValueError as why:
            rejected += 1
            reason = why.args[0][1]
            # what the residual compiler may refuse is a closed list (residual.h): anything else is a bug
            assert any(s in reason for s in ("context map", "configured list", "budget", "nested deeper", "stack slots", "per request", "concatenation of more", "not supported", "unsupported",
                                             "not a String literal", "too large", "computed key")), (e, reason)
            continue
        orc = pyoracle.Oracle([("r", e, [H.B])], LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
        assert m.header_names == orc.header_names or set(m.header_names) == set(orc.header_names), (e, m.header_names, orc.header_names)
        m.bind(batch)
        for i in range(batch.n):
            want = orc.execute_rule(0, batch, i) == 1
            got = m.eval(0, i)
            assert got == want, (seed, e, i, [batch.field_bytes(f, i) for f in range(5)], int(batch.port[i]), int(batch.asn[i]), int(batch.country[i]).to_bytes(2, "little"),
                                 {k: batch.header_bytes(k, i) for k in batch.headers})
    assert rejected <= 6, f"{rejected} of {len(exprs)} expressions refused by the residual compiler"


@pytest.mark.parametrize("seed", range(20))
def test_every_expression_of_the_general_fuzzer_runs_in_the_interpreter(seed):
    """The column compiler's own fuzz grammar (helpers.rexpr: predicates it can take, ill-typed and erroring ones) through the
    interpreter: it is a complete evaluator of the language, not only of the residue."""
    rng = random.Random(515100 + seed)
    lists = H.fuzz_lists(rng)
    exprs = [H.rexpr(rng, lists) for _ in range(15)]
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, 40, with_geo=True))
    for e in exprs:
        try:
            pyoracle.compile_expression(e)
        except pyoracle.OracleError:
            continue
        try:
            m = HostVM([e], lists)
        except ValueError:
            continue
        orc = pyoracle.Oracle([("r", e, [H.B])], lists, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
        m.bind(batch)
        for i in range(batch.n):
            want3 = orc.execute_rule(0, batch, i)  # 1 true, 0 false, 2 non-Bool, 3 error
            got3 = m.eval3(0, i)
            assert (got3 == 1) == (want3 == 1), (seed, e, i, [batch.field_bytes(f, i) for f in range(5)])
            assert (got3 == 2) == (want3 == 3), ("execution errors are counted per rule: the interpreter and the oracle must agree on them", seed, e, i, got3, want3)


# ---------------------------------------------------------------------------------------------------------------------
# round 6: the cheapest operand of a && / || chain goes first — when, and only when, no order can change an error
# ---------------------------------------------------------------------------------------------------------------------
def dchain(rng):
    """&& / || chains whose operands are mostly PURE in the compiler's sense (statically Bool, cannot fail: comparisons of lengths / ports with
    in-range arithmetic, equalities of concatenations, starts_with / contains / matches of Strings) with failing or ill-typed ones mixed in:
    a pure chain is reordered by cost, a chain with one impure operand keeps the source order — either way the value AND the error must
    be the oracle's, which evaluates left to right."""
    S = ["http_request.host", "http_request.path", "http_request.method", "http_request.url", "client.country", 'http_request.headers["x-a"]', '"/a"', '"GET"', '""',
         '(http_request.host + ":" + http_request.method)', '(http_request.path + "x")', '(client.country + http_request.method)']
    I = ["client.remote_port", "client.asn", "http_request.path.length()", "http_request.url.length()", "(http_request.path.length() + 1)", "(client.remote_port * 2)", "(client.remote_port % 7)",
         "(client.remote_port / 7)", "(http_request.url.length() - http_request.path.length())", "(client.asn * client.asn)", "(-client.remote_port)", "3", "80", "4242", "0"]
    IMPURE = ["client.remote_port / (client.remote_port - 80) == 1", "9223372036854775807 + client.remote_port > 0", "client.asn * client.asn * client.asn > 5", 'http_request.host + 1 == "a"',
              "client.remote_port", '!http_request.path', "client.remote_port % (client.asn - client.asn) == 0", '{"a": 1}.b == 1', 'http_request.path.matches("(")', "[1][client.remote_port] == 1",
              "client.remote_port < http_request.path", 'http_request.headers["nope"] == ""', "(client.remote_port > 80 ? 1 : true)"]

    def operand(depth):
        k = rng.randint(0, 11)
        if k <= 2:
            return rng.choice(I) + " " + rng.choice(["==", "!=", "<", "<=", ">", ">="]) + " " + rng.choice(I)
        if k <= 4:
            return rng.choice(S) + " " + rng.choice(["==", "!=", "<", ">="]) + " " + rng.choice(S)
        if k <= 6:
            return rng.choice(S) + "." + rng.choice(["contains", "starts_with", "ends_with"]) + "(" + rng.choice(S) + ")"
        if k == 7:
            return rng.choice(S) + ".matches(" + H.q(rng.choice(["^[a-z]+:(GET|POST)$", "^/a", "a.*b", "[0-9]+$", "(?i)get"])) + ")"
        if k == 8:
            return "!(" + operand(depth + 1) + ")"
        if k == 9 and depth < 2:
            return "(" + operand(depth + 1) + " ? " + operand(depth + 1) + " : " + operand(depth + 1) + ")"
        if k == 10 and depth < 2:
            return "(" + chain(depth + 1) + ")"
        return rng.choice(IMPURE) if rng.random() < 0.5 else rng.choice(["true", "false"])

    def chain(depth):
        op = rng.choice([" && ", " || "])
        return op.join(operand(depth) for _ in range(rng.randint(2, 5)))

    return chain(0)


@pytest.mark.parametrize("seed", range(40))
def test_reordered_chains_keep_the_value_and_the_error_of_left_to_right_evaluation(seed):
    rng = random.Random(626200 + seed)
    exprs = [dchain(rng) for _ in range(14)]
    batch = RequestBatch.from_requests(requests(rng, 30))
    for e in exprs:
        pyoracle.compile_expression(e)
        m = HostVM([e], LISTS)
        orc = pyoracle.Oracle([("r", e, [H.B])], LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
        m.bind(batch)
        for i in range(batch.n):
            want3 = orc.execute_rule(0, batch, i)  # 1 true, 0 false, 2 non-Bool, 3 error
            got3 = m.eval3(0, i)
            assert got3 == {1: 1, 0: 0, 2: 0, 3: 2}[want3], (seed, e, i, got3, want3, [batch.field_bytes(f, i) for f in range(5)], int(batch.port[i]), int(batch.asn[i]))


def test_the_cheapest_pure_operand_is_evaluated_first():
    """The program text itself: the regex of `(host + ":" + method).matches(..) && path + "x" == "/qx"` comes AFTER the equality, a chain with
    an operand that can fail keeps its source order (PWAF_RESIDUAL_SOURCE_ORDER=1 keeps every chain's)."""
    m = HostVM(['(http_request.host + ":" + http_request.method).matches("^[a-z]+:(GET|POST)$") && http_request.path + "x" == "/qx"',
                '(http_request.host + ":" + http_request.method).matches("^[a-z]+:(GET|POST)$") && client.remote_port / (client.remote_port - 80) == 1',
                'http_request.url.contains(http_request.host) || client.remote_port % 2 == 0 || http_request.path.length() + 1 > http_request.url.length()'])
    src = m.specialized_source()
    r0, r1, r2 = (src[src.index(f"rvm_rule_{k}("):src.index(f"rvm_rule_{k + 1}(") if k < 2 else len(src)] for k in range(3))
    assert r0.index("op_bin(m, 2u") < r0.index("op_matches")       # == before the walk
    assert r1.index("op_matches") < r1.index("op_bin(m, 12u")      # a division that can fail: source order
    assert r2.index("op_bin(m, 13u") < r2.index("op_call(m, 0u")   # % and the length comparison before contains
    assert r2.index("op_bin(m, 6u") < r2.index("op_call(m, 0u")


def test_residual_known_answers():
    cases = [
        ("http_request.path.length() + 1 > http_request.url.length()", [Request(path="/abc", url="/abc"), Request(path="/a", url="/a?x=1")], [True, False]),
        ("client.remote_port % 2 == 0", [Request(remote_port=80), Request(remote_port=81)], [True, False]),
        ("client.remote_port / (client.remote_port - 80) == 1", [Request(remote_port=80), Request(remote_port=81)], [False, False]),  # division by zero is an error
        ("9223372036854775807 + client.remote_port > 0", [Request(remote_port=1), Request(remote_port=0)], [False, True]),          # overflow is an error
        ('http_request.host + http_request.path == "a.b/c"', [Request(host="a.b", path="/c"), Request(host="a.b/", path="c"), Request(host="a.b", path="/d")], [True, True, False]),
        ('(http_request.host + ":" + http_request.method).matches("^[a-z]+:(GET|POST)$")', [Request(host="ab", method="GET"), Request(host="a1", method="GET")], [True, False]),
        ('[http_request.host, "zz"].contains(http_request.path)', [Request(host="/p", path="/p"), Request(host="h", path="zz"), Request(host="h", path="/p")], [True, True, False]),
        ("http_request.host < http_request.path", [Request(host="a", path="b"), Request(host="b", path="a"), Request(host="a", path="a")], [True, False, False]),
        ("http_request.host.contains(client.country)", [Request(host="xFRx", country="FR", asn=1), Request(host="xfrx", country="FR", asn=1)], [True, False]),
        ('(http_request.path.starts_with("/a") ? 1 : 2) == 1', [Request(path="/ab"), Request(path="/b")], [True, False]),
        ('{"k": http_request.host}.k == "h"', [Request(host="h"), Request(host="g")], [True, False]),
        ("client.remote_port < client.asn", [Request(remote_port=5, asn=9, country="US"), Request(remote_port=9, asn=5, country="US")], [True, False]),
        ('lists["nets"].contains(client.ip) && lists["asns"][2] == client.asn', [Request(ip="1.2.3.4", asn=64512, country="US"), Request(ip="8.8.8.8", asn=64512, country="US")], [True, False]),
    ]
    for e, reqs, want in cases:
        batch = RequestBatch.from_requests([Request(**{**r.__dict__, "user_agent": "ua"}) for r in reqs])
        m = HostVM([e], LISTS)
        m.bind(batch)
        orc = pyoracle.Oracle([("r", e, [H.B])], LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
        got = [m.eval(0, i) for i in range(batch.n)]
        assert got == want, (e, got)
        assert [orc.execute_rule(0, batch, i) == 1 for i in range(batch.n)] == want, e


@pytest.mark.parametrize("seed", range(30))
def test_rule_sets_mixing_column_and_residual_rules_through_the_compiler(seed):
    """compile.cpp's integration: rule sets in which some rules compile to columns and others fall to the residual interpreter
    (one atom per such rule, a pseudo pass of its own). The compiled tables — interpreted by the walker, residual columns by the
    host build of the interpreter running the program image from the dump — give the oracle's verdicts: rule order, first match wins,
    actions and gates are untouched by which path evaluates a rule."""
    import table_walker
    from pingoo_amd.engine import CompiledProgram

    rng = random.Random(616100 + seed)
    rules = []
    for k in range(rng.randint(2, 10)):
        e = dbool(rng) if rng.random() < 0.5 else H.rexpr(rng, LISTS)
        try:
            pyoracle.compile_expression(e)
        except pyoracle.OracleError:
            e = "true"
        rules.append((f"r{k}", e, H.fuzz_actions(rng)))
    flags = rng.choice([0, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS])
    prog = CompiledProgram(rules, LISTS, flags=flags | _abi.OPT_LENIENT)
    seen, _ = H.as_the_engine_sees(rules, prog)
    batch = RequestBatch.from_requests(requests(rng, 40))
    want = pyoracle.Oracle(seen, LISTS, flags=flags).evaluate(batch)
    t = table_walker.Tables(prog)
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got, want, batch, f"seed {seed}: {[r[1] for r in rules]}")


def test_the_mixed_rule_sets_did_exercise_residual_rules():
    """(Counts by compiling the first seeds' rule sets again: an accumulator filled by the test above is empty when pytest-xdist ran
    that test in other worker processes.)"""
    from pingoo_amd.engine import CompiledProgram

    n_residual = 0
    for seed in range(30):
        rng = random.Random(616100 + seed)
        rules = []
        for k in range(rng.randint(2, 10)):
            e = dbool(rng) if rng.random() < 0.5 else H.rexpr(rng, LISTS)
            try:
                pyoracle.compile_expression(e)
            except pyoracle.OracleError:
                e = "true"
            rules.append((f"r{k}", e, H.fuzz_actions(rng)))
        flags = rng.choice([0, _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS])
        prog = CompiledProgram(rules, LISTS, flags=flags | _abi.OPT_LENIENT)
        n_residual += sum("residual program" in w for w in prog.warnings())
    assert n_residual >= 20, n_residual


def test_dnf_explosion_falls_to_the_interpreter():
    """A boolean structure whose DNF exceeds the device limit (1024 terms) has no column form; the interpreter evaluates the tree."""
    import table_walker
    from pingoo_amd.engine import CompiledProgram

    parts = [f'(http_request.path.contains("a{k}") || http_request.url.contains("b{k}") || http_request.host.contains("c{k}"))' for k in range(8)]
    e = " && ".join(parts)  # 3^8 = 6561 terms
    rules = [("big", e, [H.B]), ("after", 'http_request.path.contains("zz")', [H.CAP])]
    prog = CompiledProgram(rules)
    assert any("residual interpreter" in w and "DNF" in w for w in prog.warnings()), prog.warnings()
    rng = random.Random(1)
    reqs = []
    for _ in range(200):
        toks = [rng.choice(["a", "b", "c"]) + str(k) for k in range(8) if rng.random() < 0.9]
        reqs.append(Request(path="/" + "".join(t for t in toks if t[0] == "a") + rng.choice(["", "zz"]), url="/" + "".join(t for t in toks if t[0] == "b"), host="".join(t for t in toks if t[0] == "c"), user_agent="ua"))
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    t = table_walker.Tables(prog)
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got, want, batch, "DNF explosion")
    assert len(set(want["rule_idx"].tolist())) >= 2


def test_heap_use_of_indexed_nested_literals_is_charged_in_full():
    """`[[1..20]][0] + [[1..20]][0]` — the item an index (or a member access) brings to the top can be LONGER than its
    receiver, and the concatenation copies it: the compiler charged 44 of the 64 heap slots and the interpreter wrote 82. Now the bound
    follows the nested lengths: what fits evaluates like the oracle, what does not is refused at compile time — never overrun."""
    inner = "[" + ", ".join(str(k) for k in range(1, 41)) + "]"  # (round 6: a literal of constants is itself a constant of the program — only what the concatenation copies is charged: 40 + 40 > 64)
    rng = random.Random(77)
    batch = RequestBatch.from_requests([Request(host="h", path="/p", user_agent="ua", remote_port=p) for p in (1, 20, 21, 40)])
    flags = _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS
    refused = accepted = 0
    exprs = [f"([{inner}][0] + [{inner}][0]).contains(client.remote_port)",                    # 40 + 40 copied = 80 > 64: refused
             f'({{"k": {inner}}}.k + {{"k": {inner}}}.k).contains(client.remote_port)',         # same through a map member
             "([[1, 2, 3, 4, 5, 6]][0] + [[20, 21]][0]).contains(client.remote_port)",          # small: runs
             '({"k": [1, 2, 3]}.k + [[40], [20, 21]][1]).contains(client.remote_port)']
    for _ in range(60):  # random nestings, indexed and concatenated
        def lit(depth):
            if depth == 0 or rng.random() < 0.3:
                return str(rng.randint(0, 45))
            return "[" + ", ".join(lit(depth - 1) for _ in range(rng.randint(1, 9))) + "]"
        a, b = "[" + ", ".join(lit(2) for _ in range(rng.randint(1, 4))) + "]", "[" + ", ".join(lit(2) for _ in range(rng.randint(1, 4))) + "]"
        exprs.append(f"({a}[0] + {b}[{rng.randint(0, 1)}]).contains(client.remote_port)")
    for e in exprs:
        try:
            m = HostVM([e])
        except ValueError as why:
            assert "per request" in str(why), (e, why)
            refused += 1
            continue
        accepted += 1
        m.bind(batch)
        orc = pyoracle.Oracle([("r", e, [H.B])], flags=flags)
        for i in range(batch.n):
            assert m.eval(0, i) == (orc.execute_rule(0, batch, i) == 1), (e, i)
    assert refused >= 2 and accepted >= 10, (refused, accepted)


def test_computed_keys_into_http_request_and_the_headers_map():
    """Round 5 (VERDICT r4 missing #3): `http_request[k]` and `http_request.headers[k]` with a COMPUTED key. The reference evaluates
    any expression (pingoo/rules.rs:37-51) over a real map; here the map is built per request from the closed key sets — the five
    fields, and for the headers map (EXTENSION) the names the WHOLE rule set mentions with a literal key, collected before any rule is
    compiled. Known answers first, then the oracle on every request."""
    rules = ['http_request["pa" + "th"].starts_with("/adm")',                                   # a computed field name
             'http_request[http_request.method == "GET" ? "host" : "path"] == "a.example"',      # which field depends on the request
             'http_request.headers["x-" + "a"] == "1"',                                          # a computed header name
             'http_request.headers[http_request.method] == "v"',                                 # a header named like the method (none: error = no match)
             '(http_request.host + "") in http_request.headers',                                 # membership under a computed key
             'http_request.headers.length() == 2',                                               # the closed set: x-a, cookie
             'http_request["headers"][http_request.path] == "1"',                                # the headers map through a literal index, then computed
             'http_request["nope" + ""] == "x"',                                                 # an absent key: execution error, no match
             'http_request[client.remote_port] == "x"',                                          # a key that is not a String: execution error
             'http_request.headers.cookie == "c"',                                               # (a literal key: a name of the map)
             '"x-a" in http_request.headers']                                                    # (a literal key: the map's second name; true wherever the map exists)
    reqs = [Request(host="a.example", path="/admin", method="GET", url="/admin", user_agent="ua", headers={"x-a": "1", "cookie": "c"}),
            Request(host="b.example", path="a.example", method="POST", url="/x", user_agent="ua", headers={"x-a": "2"}),
            Request(host="x-a", path="x-a", method="PUT", url="/", user_agent="ua", headers={"x-a": "1"}),
            Request(host="cookie", path="/", method="GET", url="/", user_agent="ua")]
    batch = RequestBatch.from_requests(reqs)
    m = HostVM(rules, LISTS)
    assert m.header_names == ["cookie", "x-a"]  # literal keys only, in order of first use: `"x-" + "a"` names nothing
    m.bind(batch)
    got = [[m.eval(k, i) for i in range(batch.n)] for k in range(len(rules))]
    assert got[0] == [True, False, False, False]
    assert got[1] == [True, True, False, False]   # GET: host == a.example; POST: path == a.example
    assert got[2] == [True, False, True, False]
    assert got[3] == [False, False, False, False]
    assert got[4] == [False, False, True, True]    # the host's value is a header NAME of the rule set
    assert got[5] == [True, True, True, True]
    assert got[6] == [False, False, True, False]   # headers["x-a"] == "1" where the path is "x-a"
    assert got[7] == [False] * 4 and got[8] == [False] * 4
    assert got[9] == [True, False, False, False] and got[10] == [True] * 4
    orc = pyoracle.Oracle([(f"r{k}", e, [H.B]) for k, e in enumerate(rules)], LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    assert orc.header_names == m.header_names
    for k in range(len(rules)):
        for i in range(batch.n):
            assert (orc.execute_rule(k, batch, i) == 1) == got[k][i], (rules[k], i)
    # through the whole compiler: such rules are residual rules of the program, none is refused
    from pingoo_amd.engine import CompiledProgram
    prog = CompiledProgram([(f"r{k}", e, [H.B]) for k, e in enumerate(rules)], LISTS)
    assert prog.header_names == ["cookie", "x-a"]
    assert sum("residual" in w for w in prog.warnings()) >= 4 and not any("NOT evaluated" in w for w in prog.warnings())  # (constant keys such as "pa" + "th" fold in the column compiler)
    import table_walker

    t = table_walker.Tables(prog)
    want = orc.evaluate(batch)
    got_v = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got_v, want, batch, "computed keys through the compiled program")


def test_a_headers_map_of_64_names_is_a_value():
    """VERDICT r5 #6 / missing #3: a rule set that mentions 64 header names (BASELINE configs[4]'s width) and needs the headers map AS A VALUE
    (a computed key). The map's entries used to travel over the interpreter's stack (2 x 64 slots of 24): every such rule was refused by
    name. Now a context map is a constant of the program whose value slots name the request values (residual.h: T_REF) — one stack slot,
    no heap item. Interpreter, specialized form and oracle on every rule and request."""
    import test_residual_jit as J

    names = [f"x-h{k}" for k in range(64)]
    rules = [f'http_request.headers["{n}"] == "{k}"' for k, n in enumerate(names)]
    rules += ['http_request.headers[http_request.method] == "v"',                       # the method names a header of the set, or the key is absent (an error)
              '(http_request.host + "") in http_request.headers',
              'http_request.headers.contains(http_request.path)',
              'http_request.headers.length() == 64',
              'http_request["headers"][http_request.path] == "7"',
              'http_request[http_request.method == "GET" ? "headers" : "host"] == "x-h3"',  # a Map on one branch (unequal to a String, no error), the host on the other
              'http_request.headers["x-h" + "63"] == "63"',
              'http_request.headers[client.country] == ""']
    rng = random.Random(99)
    reqs = []
    for i in range(48):
        hd = {n: str(k) if rng.random() < 0.5 else H.rstr(rng, 0, 3) for k, n in enumerate(names) if rng.random() < 0.4}
        reqs.append(Request(host=rng.choice(["x-h3", "x-h40", "a.example", "x-h64"]), path=rng.choice(["x-h7", "/p", "x-h63", "x-h"]), method=rng.choice(["GET", "POST", "x-h12", "x-h7"]),
                            url="/u", user_agent="ua", country=rng.choice(["US", "FR"]), asn=1, headers=hd))
    batch = RequestBatch.from_requests(reqs)
    m = HostVM(rules, LISTS)
    assert m.header_names == names
    orc = pyoracle.Oracle([(f"r{k}", e, [H.B]) for k, e in enumerate(rules)], LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    assert orc.header_names == names
    m.bind(batch)
    hits = 0
    for k in range(len(rules)):
        for i in range(batch.n):
            want3 = orc.execute_rule(k, batch, i)
            got3 = m.eval3(k, i)
            assert got3 == {1: 1, 0: 0, 2: 0, 3: 2}[want3], (rules[k], i, got3, want3)
            hits += k >= 64 and got3 == 1
    assert hits >= 20  # (the computed keys do find headers)
    J.check_rule_set(rules, LISTS, batch, "hdr64")  # the specialized form of the same rule set
    # through the whole compiler: none refused, and the program names every rule that runs as a residual program
    from pingoo_amd.engine import CompiledProgram
    prog = CompiledProgram([(f"r{k}", e, [H.B]) for k, e in enumerate(rules)], LISTS)
    assert prog.header_names == names and not any("NOT evaluated" in w for w in prog.warnings())
    import table_walker

    t = table_walker.Tables(prog)
    want = orc.evaluate(batch)
    got_v = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got_v, want, batch, "64 header names, computed keys")


def test_matches_with_a_pattern_computed_from_finitely_many_strings():
    """VERDICT r5 missing #3 / #8: `s.matches(p)` where p is not a literal but can only be one of finitely many strings known when the rule is
    compiled — a conditional between literals, an item of a configured String list, concatenations of such, an item of a literal list /
    map. The reference compiles the pattern per evaluation (pingoo/rules.rs:37-51); here every candidate's table is built at engine
    creation and the pattern's VALUE picks one per request. An invalid candidate is an execution error only when it is the one selected;
    a pattern over request bytes stays refused."""
    import test_residual_jit as J

    lists = dict(LISTS, pats=(_abi.LIST_STRING, ["^/adm", "\\.php$", "(", "^[a-z]+$"]))
    rules = ['http_request.path.matches(http_request.method == "GET" ? "^/adm" : "^/api")',
             'http_request.path.matches(lists["pats"][client.remote_port % 4])',           # item 2 is an invalid pattern: an error for those ports only
             'http_request.path.matches("^/" + (client.remote_port > 100 ? "a" : "b") + "[a-z]*$")',
             'http_request.host.matches(["^a", "e$"][client.remote_port % 2])',
             'http_request.host.matches({"x": "^a", "y": "zz"}[http_request.method == "GET" ? "x" : "y"])',
             'http_request.url.matches(lists.pats[0] + "in")',
             'http_request.path.matches(client.remote_port > 100 ? 5 : "^/b")',            # an Int on one branch: String operands required
             'http_request.path.matches(lists["asns"][0])',                                # an Int list: never a String
             'http_request.path.matches(lists["pats"][client.remote_port])',               # index out of range for most ports
             '(http_request.host + "/").matches(lists.pats[3] + "/$")']
    reqs = [Request(host=h, path=p, url=p + "?x=1", method=m, remote_port=port, user_agent="ua")
            for h in ("a.example", "zz", "abc") for p in ("/admin", "/api/x.php", "/b", "/abc") for m in ("GET", "POST") for port in (0, 1, 2, 3, 101, 443)]
    batch = RequestBatch.from_requests(reqs)
    m = HostVM(rules, lists)
    orc = pyoracle.Oracle([(f"r{k}", e, [H.B]) for k, e in enumerate(rules)], lists, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
    m.bind(batch)
    seen = {k: set() for k in range(len(rules))}
    for k in range(len(rules)):
        for i in range(batch.n):
            want3 = orc.execute_rule(k, batch, i)
            got3 = m.eval3(k, i)
            assert got3 == {1: 1, 0: 0, 2: 0, 3: 2}[want3], (rules[k], i, got3, want3)
            seen[k].add(got3)
    assert all(seen[k] >= {0, 1} for k in (0, 2, 3, 4, 5)) and seen[1] == {0, 1, 2} and seen[6] >= {1, 2} and seen[7] == {2} and 2 in seen[8], seen
    J.check_rule_set(rules, lists, batch, "rxset")
    # still refused: a pattern made of request bytes
    for e in ['http_request.path.matches(http_request.host)', 'http_request.path.matches("^" + client.country)', 'http_request.path.matches(lists.pats[0] + http_request.method)']:
        with pytest.raises(ValueError, match="not a String literal"):
            HostVM([e], lists)
    # through the whole compiler (the column compiler folds constants; what is left computed becomes a residual rule)
    from pingoo_amd.engine import CompiledProgram
    import table_walker

    prog = CompiledProgram([(f"r{k}", e, [H.B]) for k, e in enumerate(rules)], lists)
    assert not any("NOT evaluated" in w for w in prog.warnings())
    t = table_walker.Tables(prog)
    H.assert_verdicts_equal(np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)]), orc.evaluate(batch), batch, "computed patterns")


def test_header_keys_that_only_constant_folding_makes_literal():
    """Found by tools/gpufuzz.py (round 5): the column compiler folds `"head" + "ers"` and `"x-" + "a"` to constants and used to treat the
    result like a LITERAL key — registering a header name the oracle's headers map does not hold (its names come from literal keys in the
    syntax: oracle_engine.cpp). With the names closed before compilation such a key is an absent key: an error under [], false under `in`."""
    import table_walker
    from pingoo_amd.engine import CompiledProgram

    reqs = [Request(host="h", path="/", url="/1", user_agent="ua", headers={"x-a": "1", "cookie": "c"}), Request(host="h", path="/", url="/", user_agent="ua", headers={"x-a": "2"}),
            Request(host="h", path="/", url="/", user_agent="ua")]
    for rules, names in (([("fold", 'http_request["head" + "ers"]["x-a"] == "1"', [H.B]), ("foldin", '("x-" + "a") in http_request.headers', [H.CAP]),
                           ("foldidx", 'http_request.headers["x-" + "a"] != "zz"', [H.B]), ("last", 'http_request.path == "/"', [H.CAP])], []),
                         ([("fold", 'http_request["head" + "ers"]["x-a"] == "1"', [H.B]), ("lit", 'http_request.headers["x-a"] == "2"', [H.CAP]),
                           ("foldin", '("coo" + "kie") in http_request.headers', [H.B]), ("last", 'http_request.path == "/"', [H.CAP])], ["x-a"])):
        prog = CompiledProgram(rules, LISTS)
        orc = pyoracle.Oracle(rules, LISTS)
        assert prog.header_names == names == orc.header_names
        batch = RequestBatch.from_requests(reqs)
        want = orc.evaluate(batch)
        t = table_walker.Tables(prog)
        got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
        H.assert_verdicts_equal(got, want, batch, str(names))
        if names:
            assert want["rule_idx"].tolist() == [0, 1, 3]  # the folded key reads the column a literal mention made; "cookie" is no name of this set
        else:
            assert want["rule_idx"].tolist() == [3, 3, 3]  # no name at all: every folded key is absent


def test_residual_regex_walker_reads_a_stray_continuation_byte_as_an_ill_formed_unit():
    """D17 closed: the residual programs' own regex walker (residual.h: regex_match_t, one table per pattern) on ill-formed UTF-8 — through the
    host VM and the oracle; ropes too (a sequence that straddles two segments)."""
    from test_compiler import ILL_HAYS, ILL_PATTERNS

    reqs = [Request(path=h, url=b"/", host=b"", user_agent="ua") for h in ILL_HAYS] + [Request(path=h[1:], host=h[:1], url=b"/", user_agent="ua") for h in ILL_HAYS if len(h) > 1]
    batch = RequestBatch.from_requests(reqs)
    n_true = 0
    for pat in ILL_PATTERNS:
        e = f"(http_request.host + http_request.path).matches({H.q(pat)})"
        m = HostVM([e], LISTS)
        m.bind(batch)
        orc = pyoracle.Oracle([("r", e, [H.B])], LISTS, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS)
        for i in range(batch.n):
            want = orc.execute_rule(0, batch, i) == 1
            assert m.eval(0, i) == want, (pat, reqs[i].host, reqs[i].path)
            n_true += want
    assert n_true > 40
