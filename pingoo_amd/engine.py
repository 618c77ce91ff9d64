"""Host-side mirror of the reference's rule interface, over the C ABI of libpwaf.so.

Names follow the reference so call sites read the same:

    rules::Action {Block, Captcha}        rules/rules.rs:30-35      -> Action
    rules::compile_expression             rules/rules.rs:45-53      -> compile_expression
    rules::validate_expression            rules/rules.rs:55-77      -> validate_expression
    pingoo::rules::Rule {name, expression: Option<_>, actions}      -> Rule
    Rule::match_request / the rule loop   http_listener.rs:251-264  -> RuleEngine.evaluate(...)
    Error::ExpressionIsNotValid           rules/rules.rs:41-42      -> ExpressionIsNotValid

Every evaluation runs on the GPU through libpwaf.so; there is no Python or CPU evaluation path.
If the library (or a HIP device) is missing the calls raise — loudly.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .batch import VERDICT_DTYPE, Request, RequestBatch

_LIB = None
# PWAF_LIB_VARIANT=prof loads libpwaf_prof.so, the -DPWAF_PROFILING build with the timing-experiment switches (tools/ only: results may
# be wrong when a switch is set). Nothing else is ever loaded: there is no fallback of any kind.
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libpwaf_{os.environ['PWAF_LIB_VARIANT']}.so" if os.environ.get("PWAF_LIB_VARIANT") else "libpwaf.so")


class PwafError(RuntimeError):
    def __init__(self, code: int, message: str, rule_index: Optional[int] = None):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message
        self.rule_index = rule_index


class ExpressionIsNotValid(PwafError):
    """rules::Error::ExpressionIsNotValid (rules/rules.rs:41-42)."""


class UnsupportedExpression(PwafError):
    """A valid expression outside the device-compilable subset."""


class DeviceError(PwafError):
    """No usable HIP device / a HIP call failed. The caller decides whether to fail open."""


def lib():
    """Loads libpwaf.so. Raises if it has not been built — the product path never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: build it with `python -m pingoo_amd.build` (hipcc, gfx950). "
                          "pingoo_amd has no CPU fallback.")
    # ONE HIP runtime per process. libpwaf.so needs "libamdhip64.so.7"; PyTorch-ROCm bundles its own copy with that
    # very soname (plus its own libhsa-runtime64). Whichever is mapped first serves both users — two HSA runtimes in
    # one process do not work (the second sees no GPU). bench.py and the tests use torch for device buffers, streams
    # and torch.distributed, so when torch is installed its runtime is mapped first and libpwaf.so binds to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.pwaf_abi_version.restype = C.c_uint32
    L.pwaf_last_error.restype = C.c_char_p
    L.pwaf_compile_expression.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.pwaf_validate_expression.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    create_args = [C.POINTER(_abi.RuleDesc), C.c_size_t, C.POINTER(_abi.ListDesc), C.c_size_t, C.POINTER(_abi.GeoipTable), C.POINTER(_abi.Options)]
    L.pwaf_program_compile.argtypes = create_args + [C.POINTER(vp), C.POINTER(_abi.CompileError)]
    L.pwaf_program_destroy.argtypes = [vp]
    L.pwaf_program_destroy.restype = None
    L.pwaf_program_dump.argtypes = [vp, vp, C.c_size_t]
    L.pwaf_program_dump.restype = C.c_size_t
    L.pwaf_program_warning_count.argtypes = [vp]
    L.pwaf_program_warning_count.restype = C.c_size_t
    L.pwaf_program_warning.argtypes = [vp, C.c_size_t]
    L.pwaf_program_warning.restype = C.c_char_p
    L.pwaf_program_stats.argtypes = [vp, C.POINTER(_abi.Stats)]
    L.pwaf_program_rule_status.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t]
    L.pwaf_program_confirm_field.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint16), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pwaf_engine_rule_errors.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t]
    L.pwaf_engine_residual_mode.argtypes = [vp]
    L.pwaf_engine_residual_fallback.argtypes = [vp]
    L.pwaf_engine_residual_fallback.restype = C.c_char_p
    L.pwaf_program_residual_source.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    L.pwaf_program_residual_source.restype = C.c_size_t
    L.pwaf_program_residual_compile.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    L.pwaf_program_residual_compile.restype = C.c_long
    L.pwaf_engine_create.argtypes = create_args + [C.POINTER(vp), C.POINTER(_abi.CompileError)]
    L.pwaf_engine_destroy.argtypes = [vp]
    L.pwaf_engine_destroy.restype = None
    L.pwaf_engine_program.argtypes = [vp]
    L.pwaf_engine_program.restype = vp
    L.pwaf_engine_stats.argtypes = [vp, C.POINTER(_abi.Stats)]
    for fn_name in ("pwaf_engine_header_count", "pwaf_program_header_count"):
        getattr(L, fn_name).argtypes = [vp]
        getattr(L, fn_name).restype = C.c_uint32
    for fn_name in ("pwaf_engine_header_name", "pwaf_program_header_name"):
        getattr(L, fn_name).argtypes = [vp, C.c_uint32]
        getattr(L, fn_name).restype = C.c_char_p
    L.pwaf_engine_stream.argtypes = [vp]
    L.pwaf_engine_stream.restype = vp
    L.pwaf_evaluate_batch.argtypes = [vp, C.POINTER(_abi.Batch), vp, vp]
    L.pwaf_evaluate_device.argtypes = [vp, C.POINTER(_abi.Batch), vp, vp, vp, vp, vp]
    L.pwaf_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.pwaf_host_free.argtypes = [vp]
    L.pwaf_host_free.restype = None
    L.pwaf_host_register.argtypes = [vp, C.c_size_t]
    L.pwaf_host_unregister.argtypes = [vp]
    L.pwaf_evaluate_one.argtypes = [vp, C.POINTER(_abi.Request), C.POINTER(_abi.Verdict)]
    L.pwaf_engine_device_status.argtypes = [vp]
    L.pwaf_engine_set_profiling.argtypes = [vp, C.c_int]
    L.pwaf_engine_tune.argtypes = [vp, C.POINTER(_abi.Batch)]
    L.pwaf_program_tune.argtypes = [vp, C.POINTER(_abi.Batch)]
    L.pwaf_node_create.argtypes = create_args + [C.POINTER(C.c_int), C.c_size_t, C.POINTER(vp), C.POINTER(_abi.CompileError)]
    L.pwaf_node_destroy.argtypes = [vp]
    L.pwaf_node_destroy.restype = None
    L.pwaf_node_device_count.argtypes = [vp]
    L.pwaf_node_device_count.restype = C.c_size_t
    L.pwaf_node_engine.argtypes = [vp, C.c_size_t]
    L.pwaf_node_engine.restype = vp
    L.pwaf_node_tune.argtypes = [vp, C.POINTER(_abi.Batch)]
    L.pwaf_node_evaluate_batch.argtypes = [vp, C.POINTER(_abi.Batch), vp, vp]
    L.pwaf_node_evaluate_device.argtypes = [vp, C.POINTER(C.POINTER(_abi.Batch)), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pwaf_node_synchronize.argtypes = [vp]
    L.pwaf_node_allreduce_counts.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pwaf_node_shard_bounds.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pwaf_node_shard_bounds.restype = None
    L.pwaf_batcher_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.pwaf_batcher_evaluate.argtypes = [vp, C.POINTER(_abi.Request), C.POINTER(_abi.Verdict)]
    L.pwaf_batcher_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pwaf_batcher_destroy.argtypes = [vp]
    L.pwaf_batcher_destroy.restype = None
    L.pwaf_geoip_from_mmdb.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(_abi.GeoipEntry)), C.POINTER(C.c_size_t)]
    L.pwaf_geoip_from_file_image.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(_abi.GeoipEntry)), C.POINTER(C.c_size_t)]
    L.pwaf_zstd_decompress.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.pwaf_buffer_free.argtypes = [C.c_void_p]
    L.pwaf_buffer_free.restype = None
    L.pwaf_geoip_free.argtypes = [C.POINTER(_abi.GeoipEntry)]
    L.pwaf_geoip_free.restype = None
    L.pwaf_list_parse_csv.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.c_size_t)]
    L.pwaf_list_free.argtypes = [C.POINTER(C.c_char_p), C.c_size_t]
    L.pwaf_list_free.restype = None
    L.pwaf_engine_kernel_times.argtypes = [vp, C.POINTER(_abi.KernelTime), C.c_int]
    L.pwaf_derive_path.argtypes = [C.c_char_p, C.c_size_t]
    L.pwaf_derive_path.restype = C.c_size_t
    L.pwaf_derive_user_agent.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.pwaf_derive_user_agent.restype = None
    L.pwaf_derive_host.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.pwaf_derive_host.restype = None
    if L.pwaf_abi_version() != _abi.ABI_VERSION:
        raise ImportError("libpwaf.so ABI version mismatch: rebuild with `python -m pingoo_amd.build --force`")
    _LIB = L
    return L


def _raise(code: int, message: str, rule_index: Optional[int] = None):
    cls = {_abi.E_SYNTAX: ExpressionIsNotValid, _abi.E_UNSUPPORTED: UnsupportedExpression, _abi.E_DEVICE: DeviceError}.get(code, PwafError)
    raise cls(code, message, rule_index)


class Action(enum.IntEnum):
    """rules::Action (rules/rules.rs:30-35). Values are the PWAF_RULE_ACTION_* codes."""

    Block = _abi.RULE_ACTION_BLOCK
    Captcha = _abi.RULE_ACTION_CAPTCHA


class Decision(enum.IntEnum):
    """What the listener does with the request (http_listener.rs:196-264)."""

    Allow = _abi.ACTION_ALLOW
    Block = _abi.ACTION_BLOCK
    Captcha = _abi.ACTION_CAPTCHA
    Bypass = _abi.ACTION_BYPASS  # /__pingoo/captcha endpoints: rules are skipped


@dataclass
class Rule:
    """pingoo::rules::Rule (pingoo/rules.rs:9-14). expression None == match all."""

    name: str
    expression: Optional[str]
    actions: Sequence[Action] = field(default_factory=list)

    def as_tuple(self) -> Tuple[str, Optional[str], List[int]]:
        return (self.name, self.expression, [int(a) for a in self.actions])


@dataclass
class Verdict:
    decision: Decision
    rule_idx: Optional[int]  # index of the deciding rule; None for Allow / the built-in gates
    gate: Optional[str] = None  # "user_agent" | "captcha_endpoint" when a built-in gate decided


def compile_expression(expression: str) -> None:
    """rules::compile_expression: raises ExpressionIsNotValid on a syntax error."""
    buf = C.create_string_buffer(512)
    rc = lib().pwaf_compile_expression(expression.encode(), buf, 512)
    if rc != 0:
        _raise(rc, buf.value.decode(errors="replace"))


def validate_expression(expression: str) -> None:
    """rules::validate_expression: additionally rejects "" and the `in` operator."""
    buf = C.create_string_buffer(512)
    rc = lib().pwaf_validate_expression(expression.encode(), buf, 512)
    if rc != 0:
        _raise(rc, buf.value.decode(errors="replace"))


def get_path(uri_path: bytes) -> bytes:
    """get_path (http_utils.rs:114-116)."""
    return uri_path[: lib().pwaf_derive_path(uri_path, len(uri_path))]


def get_user_agent(header: Optional[bytes]) -> bytes:
    """The User-Agent derivation of http_listener.rs:159-165."""
    s, l = C.c_size_t(), C.c_size_t()
    h = header or b""
    lib().pwaf_derive_user_agent(h, len(h), int(header is not None), C.byref(s), C.byref(l))
    return h[s.value:s.value + l.value]


def get_host(uri_host: Optional[bytes], host_header: Optional[bytes]) -> bytes:
    """get_host (http_listener.rs:284-296)."""
    s, l, fh = C.c_size_t(), C.c_size_t(), C.c_int()
    a, b = uri_host or b"", host_header or b""
    lib().pwaf_derive_host(a, len(a), int(uri_host is not None), b, len(b), int(host_header is not None), C.byref(fh), C.byref(s), C.byref(l))
    src = b if fh.value else a
    return src[s.value:s.value + l.value]


def _options(flags=0, device=-1, lds_table_budget=0, max_dfa_states=0, max_table_bytes=0) -> _abi.Options:
    o = _abi.Options()
    o.struct_size = C.sizeof(_abi.Options)
    o.flags = flags
    o.device = device
    o.lds_table_budget = lds_table_budget
    o.max_dfa_states = max_dfa_states
    o.max_table_bytes = max_table_bytes
    return o


def _norm_rules(rules) -> List[Tuple[str, Optional[str], List[int]]]:
    return [r.as_tuple() if isinstance(r, Rule) else (r[0], r[1], [int(a) for a in r[2]]) for r in rules]


class CompiledProgram:
    """Host-side compilation only (no GPU): stats, warnings and the table dump used by the tests."""

    def __init__(self, rules, lists: Optional[Dict[str, Tuple[int, Sequence[str]]]] = None, geoip: Optional[np.ndarray] = None, **opts):
        L = lib()
        m = _abi.Marshalled()
        r, nr = _abi.marshal_rules(_norm_rules(rules), m)
        l, nl = _abi.marshal_lists(lists, m)
        g = _abi.marshal_geoip(geoip, m)
        o = _options(**opts)
        h = C.c_void_p()
        err = _abi.CompileError()
        rc = L.pwaf_program_compile(r, nr, l, nl, g, C.byref(o), C.byref(h), C.byref(err))
        if rc < 0:
            _raise(rc, err.message.decode(errors="replace"), None if err.rule_index == 0xFFFFFFFF else err.rule_index)
        self.partial = rc == _abi.W_PARTIAL  # PWAF_OPT_LENIENT: some rule is not evaluated (rule_status / warnings say which)
        self._h = h
        self._owned = True

    @property
    def header_names(self) -> List[str]:
        """EXTENSION: the header names the rule set mentions = the header columns a batch must carry, in this order."""
        return [lib().pwaf_program_header_name(self._h, i).decode() for i in range(lib().pwaf_program_header_count(self._h))]

    @classmethod
    def _borrow(cls, handle) -> "CompiledProgram":
        self = cls.__new__(cls)
        self._h = C.c_void_p(handle)
        self._owned = False
        return self

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib().pwaf_program_destroy(self._h)
            self._h = None

    def dump(self) -> bytes:
        n = lib().pwaf_program_dump(self._h, None, 0)
        buf = C.create_string_buffer(n)
        lib().pwaf_program_dump(self._h, buf, n)
        return buf.raw

    def confirm_field(self, group: int, data: bytes, arena_offset: int = 0):
        """TEST HOOK (pwaf_program_confirm_field): prefilter + confirm tier of pass `group` over one field value, run by the very code the
        device compiles (csrc/confirm.h). Returns (local atoms of the confirmed literal predicates, flagged, walk)."""
        atoms = (C.c_uint16 * 4096)()
        n, fl, wk = C.c_size_t(), C.c_int(), C.c_int()
        rc = lib().pwaf_program_confirm_field(self._h, group, data, len(data), arena_offset, atoms, 4096, C.byref(n), C.byref(fl), C.byref(wk))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        return sorted(set(atoms[k] for k in range(n.value))), bool(fl.value), bool(wk.value)

    def residual_source(self, kind: int = 1) -> str:
        """Inspection hook: the specialized form of the residual rules (csrc/residual_jit.cpp). kind 0 = the rule functions alone,
        1 = the whole device program as handed to hiprtc at engine creation. "" when the rule set has no residual rules."""
        n = lib().pwaf_program_residual_source(self._h, kind, None, 0)
        if n == 0:
            return ""
        buf = C.create_string_buffer(n + 1)
        lib().pwaf_program_residual_source(self._h, kind, buf, n + 1)
        return buf.value.decode()

    def residual_compile(self, arch: str = "gfx950") -> int:
        """Inspection hook: compiles that program with hiprtc for `arch` (no device needed); the code object's size (0: no residual rules)."""
        err = C.create_string_buffer(4000)
        rc = lib().pwaf_program_residual_compile(self._h, arch.encode(), err, 4000)
        if rc < 0:
            _raise(int(rc), err.value.decode(errors="replace"))
        return int(rc)

    def rule_status(self, i: int) -> Tuple[int, str]:
        """(PWAF_OK, "") or (PWAF_E_UNSUPPORTED, reason): a rule the device compiler cannot take never matches and says so here."""
        buf = C.create_string_buffer(300)
        rc = lib().pwaf_program_rule_status(self._h, i, buf, 300)
        return rc, buf.value.decode(errors="replace")

    def unsupported_rules(self, n_rules: int) -> List[int]:
        return [i for i in range(n_rules) if self.rule_status(i)[0] != 0]

    def tune(self, sample: RequestBatch) -> None:
        """The host half of RuleEngine.tune on this program alone (no device): the prefilters in the dump become the tuned ones."""
        st = sample.as_struct(self.header_names)
        rc = lib().pwaf_program_tune(self._h, C.byref(st))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def warnings(self) -> List[str]:
        return [lib().pwaf_program_warning(self._h, i).decode(errors="replace") for i in range(lib().pwaf_program_warning_count(self._h))]

    def stats(self) -> dict:
        s = _abi.Stats()
        lib().pwaf_program_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in _abi.Stats._fields_ if k != "reserved"}


class RuleEngine:
    """The engine handle that replaces `(Arc<Vec<Rule>>, lists, geoip)` (pingoo/server.rs:40-47,76).

    rules: ordered [Rule | (name, expression|None, [actions])]
    lists: {name: (LIST_STRING|LIST_INT|LIST_IP, [csv column-0 strings])}
    geoip: GEOIP_DTYPE array (see pingoo_amd.batch.geoip_entries) or None
    """

    def __init__(self, rules, lists: Optional[Dict[str, Tuple[int, Sequence[str]]]] = None, geoip: Optional[np.ndarray] = None, **opts):
        L = lib()
        m = _abi.Marshalled()
        self.rules = _norm_rules(rules)
        r, nr = _abi.marshal_rules(self.rules, m)
        l, nl = _abi.marshal_lists(lists, m)
        g = _abi.marshal_geoip(geoip, m)
        o = _options(**opts)
        h = C.c_void_p()
        err = _abi.CompileError()
        rc = L.pwaf_engine_create(r, nr, l, nl, g, C.byref(o), C.byref(h), C.byref(err))
        if rc < 0:
            _raise(rc, err.message.decode(errors="replace"), None if err.rule_index == 0xFFFFFFFF else err.rule_index)
        self.partial = rc == _abi.W_PARTIAL  # PWAF_OPT_LENIENT: some rule is not evaluated (program.rule_status / warnings say which)
        self._h = h
        self.header_names = [L.pwaf_engine_header_name(h, i).decode() for i in range(L.pwaf_engine_header_count(h))]

    def close(self):
        if getattr(self, "_h", None):
            lib().pwaf_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def program(self) -> CompiledProgram:
        return CompiledProgram._borrow(lib().pwaf_engine_program(self._h))

    @property
    def residual_mode(self) -> int:
        """0: no residual rules; 1: interpreted per request (residual_kernel); 2: specialized — compiled for this device by hiprtc when
        the engine was created (csrc/residual_jit.cpp). With 1 and no OPT_NO_RESIDUAL_JIT, `residual_fallback` says why."""
        return int(lib().pwaf_engine_residual_mode(self._h))

    @property
    def residual_fallback(self) -> str:
        return (lib().pwaf_engine_residual_fallback(self._h) or b"").decode(errors="replace")

    def rule_errors(self, n_rules: int) -> List[int]:
        """Per caller rule: requests (over every batch so far) for which the rule's evaluation ended in an execution error — what the
        reference logs per occurrence (pingoo/rules.rs:41-45). Static errors are creation-time warnings instead."""
        arr = (C.c_uint64 * n_rules)()
        rc = lib().pwaf_engine_rule_errors(self._h, arr, n_rules)
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        return list(arr)

    def stats(self) -> dict:
        s = _abi.Stats()
        lib().pwaf_engine_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in _abi.Stats._fields_ if k != "reserved"}

    # ---- evaluation -------------------------------------------------------------------------------
    def evaluate_batch(self, batch: RequestBatch, with_counts: bool = False, out: Optional[np.ndarray] = None):
        """Host batch in, numpy VERDICT_DTYPE array out (and the 4 action counters when asked). `out`: a caller-owned result array
        (e.g. page-locked: PinnedVerdicts(n).array) instead of a fresh one per call."""
        if out is None:
            out = np.zeros(batch.n, dtype=VERDICT_DTYPE)
        assert out.dtype == VERDICT_DTYPE and len(out) >= batch.n and out.flags["C_CONTIGUOUS"]
        counts = _abi.Counts()
        st = batch.as_struct(self.header_names)
        rc = lib().pwaf_evaluate_batch(self._h, C.byref(st), out.ctypes.data, C.addressof(counts))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        if with_counts:
            return out, np.array(list(counts.by_action), dtype=np.uint64)
        return out

    def evaluate(self, request: Request) -> Verdict:
        """RuleEngine::evaluate(Request) -> Action (pwaf_evaluate_one): a batch of one through the same device path."""
        st, _keep = _request_struct(request, self.header_names)
        out = _abi.Verdict()
        rc = lib().pwaf_evaluate_one(self._h, C.byref(st), C.byref(out))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        return verdict_from_record({"action": out.action, "rule_idx": out.rule_idx})

    def evaluate_device(self, dbatch: "DeviceBatch", out=None, counts=None, match_idx=None, n_matches=None, stream=None):
        """Device-resident evaluation on torch's current stream (or `stream`). Tensors stay on the GPU."""
        import torch

        if out is None:
            out = torch.empty((dbatch.n, 2), dtype=torch.int32, device=dbatch.device)
        if stream is None:
            stream = torch.cuda.current_stream(dbatch.device).cuda_stream
        st = dbatch.as_struct(self.header_names)
        rc = lib().pwaf_evaluate_device(self._h, C.byref(st), out.data_ptr(), counts.data_ptr() if counts is not None else None,
                                        match_idx.data_ptr() if match_idx is not None else None, n_matches.data_ptr() if n_matches is not None else None,
                                        C.c_void_p(stream))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        return out

    def device_status(self) -> None:
        """Synchronises and raises if the last device-resident batch ran out of scan scratch."""
        rc = lib().pwaf_engine_device_status(self._h)
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def tune(self, sample: RequestBatch) -> None:
        """Re-selects the LDS-resident DFA rows from a host traffic sample (speed only; verdicts never change)."""
        st = sample.as_struct(self.header_names)
        rc = lib().pwaf_engine_tune(self._h, C.byref(st))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def set_profiling(self, on):
        """False / 0: off; True / 1: HIP events around every kernel; 2: only around the launches that stream the request bytes."""
        lib().pwaf_engine_set_profiling(self._h, int(on))

    def kernel_times(self) -> List[Tuple[str, float, int]]:
        arr = (_abi.KernelTime * 8192)()
        n = lib().pwaf_engine_kernel_times(self._h, arr, 8192)
        if n < 0:
            _raise(n, lib().pwaf_last_error().decode(errors="replace"))
        return [(arr[i].name.decode(), float(arr[i].ms), int(arr[i].alg_bytes)) for i in range(n)]


class NodeEngine:
    """One process driving several GPUs (pwaf_node_*): an engine replica per device, host batches cut into 64-aligned slabs, one host
    thread and stream per device, action counters summed on the host. The single-process counterpart of bench.py's rank-per-GPU mode."""

    def __init__(self, rules, lists=None, geoip=None, devices: Sequence[int] = (0,), **opts):
        L = lib()
        m = _abi.Marshalled()
        r, nr = _abi.marshal_rules(_norm_rules(rules), m)
        l, nl = _abi.marshal_lists(lists, m)
        g = _abi.marshal_geoip(geoip, m)
        o = _options(**opts)
        h = C.c_void_p()
        err = _abi.CompileError()
        devs = (C.c_int * len(devices))(*devices)
        rc = L.pwaf_node_create(r, nr, l, nl, g, C.byref(o), devs, len(devices), C.byref(h), C.byref(err))
        if rc < 0:
            _raise(rc, err.message.decode(errors="replace") or L.pwaf_last_error().decode(errors="replace"), None if err.rule_index == 0xFFFFFFFF else err.rule_index)
        self.partial = rc == _abi.W_PARTIAL  # PWAF_OPT_LENIENT dropped a rule (see the program's rule status / warnings)
        self._h = h
        e0 = L.pwaf_node_engine(h, 0)
        self.header_names = [L.pwaf_engine_header_name(e0, i).decode() for i in range(L.pwaf_engine_header_count(e0))]
        self.n_devices = L.pwaf_node_device_count(h)

    def evaluate_batch(self, batch: RequestBatch, with_counts: bool = False):
        out = np.zeros(batch.n, dtype=VERDICT_DTYPE)
        counts = _abi.Counts()
        st = batch.as_struct(self.header_names)
        rc = lib().pwaf_node_evaluate_batch(self._h, C.byref(st), out.ctypes.data, C.addressof(counts))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        return (out, np.array(list(counts.by_action), dtype=np.uint64)) if with_counts else out

    def evaluate_device(self, dbatches, outs, counts=None, streams=None) -> None:
        """pwaf_node_evaluate_device: one DEVICE-resident slab per device of the node (DeviceBatch list), verdict tensors `outs`
        ((n_r, 2) int32 on device r), optional per-device int64[4] counter tensors (accumulated into). Asynchronous: call
        synchronize() before reading."""
        k = len(dbatches)
        structs = [b.as_struct(self.header_names) for b in dbatches]
        arr = (C.POINTER(_abi.Batch) * k)(*[C.pointer(s) for s in structs])
        o = (C.c_void_p * k)(*[t.data_ptr() for t in outs])
        c = (C.c_void_p * k)(*[t.data_ptr() for t in counts]) if counts is not None else None
        st = (C.c_void_p * k)(*[int(x) for x in streams]) if streams is not None else None
        rc = lib().pwaf_node_evaluate_device(self._h, arr, o, c, st)
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def allreduce_counts(self, comms, counts, streams=None) -> None:
        """pwaf_node_allreduce_counts: the per-device int64[4] counter tensors summed in place over RCCL; `comms` = one ncclComm_t
        (integer handle) per device, e.g. from ncclCommInitAll."""
        k = len(counts)
        cm = (C.c_void_p * k)(*[int(x) for x in comms])
        c = (C.c_void_p * k)(*[t.data_ptr() for t in counts])
        st = (C.c_void_p * k)(*[int(x) for x in streams]) if streams is not None else None
        rc = lib().pwaf_node_allreduce_counts(self._h, cm, c, st)
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def synchronize(self) -> None:
        rc = lib().pwaf_node_synchronize(self._h)
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def tune(self, sample: RequestBatch) -> None:
        st = sample.as_struct(self.header_names)
        rc = lib().pwaf_node_tune(self._h, C.byref(st))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            lib().pwaf_node_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class PinnedVerdicts:
    """A page-locked verdict array (pwaf_host_alloc) for RuleEngine.evaluate_batch(out=...): the device-to-host copy of the results
    lands in it directly. `.array` is the numpy view; free() releases the memory (the view must not be used afterwards)."""

    def __init__(self, n: int):
        p = C.c_void_p()
        nbytes = max(1, n) * VERDICT_DTYPE.itemsize
        rc = lib().pwaf_host_alloc(nbytes, C.byref(p))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        self._p = p
        self.array = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=VERDICT_DTYPE, count=n)

    def free(self) -> None:
        if self._p:
            self.array = None
            lib().pwaf_host_free(self._p)
            self._p = None


def node_shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    lo, hi = C.c_uint32(), C.c_uint32()
    lib().pwaf_node_shard_bounds(n, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class ServiceRouter:
    """Service selection on the same engine: the reference walks `services` in order and hands the request to the
    first one whose `route:` expression is true — or has none (http_listener.rs:266-271, http_proxy_service.rs:84-95: same language,
    same context, error / non-bool = no match). That is first-match-wins over one expression per service, i.e. a rule set whose rule
    k 'blocks' with rule index k: no new device code, just an engine without the two request gates.
    `route_batch` -> int32 array: index of the selected service, -1 = none (the reference answers 404)."""

    def __init__(self, routes: Sequence[Tuple[str, Optional[str]]], lists=None, geoip=None, **opts):
        rules = [(name, expr, [_abi.RULE_ACTION_BLOCK]) for name, expr in routes]
        self._engine = RuleEngine(rules, lists or {}, geoip, flags=_abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS, **opts)

    def route_batch(self, batch: RequestBatch) -> np.ndarray:
        v = self._engine.evaluate_batch(batch)
        return np.where(v["action"] == _abi.ACTION_BLOCK, v["rule_idx"].astype(np.int64), -1).astype(np.int32)

    def close(self):
        self._engine.close()


def _request_struct(r: Request, header_names: Sequence[str] = ()):
    """pwaf_request for one Request; returns (struct, keepalive) — the byte strings must outlive the call. header_names = the engine's
    header columns: the request's values are handed over in that order (an absent header reads as "")."""
    from .batch import ip_to_bytes16

    fields = [x.encode() if isinstance(x, str) else bytes(x) for x in (r.host, r.url, r.path, r.method, r.user_agent)]
    st = _abi.Request()
    for name, b in zip(("host", "url", "path", "method", "user_agent"), fields):
        setattr(st, name, b)
        setattr(st, name + "_len", len(b))
    ip, v6 = ip_to_bytes16(r.ip)
    C.memmove(st.ip, ip, 16)
    st.ip_is_v6 = int(v6)
    st.flags = _abi.FLAG_CAPTCHA_VERIFIED if r.captcha_verified else 0
    st.port = r.remote_port
    if r.asn is not None and r.country is not None:
        st.has_geoip = 1
        cc = r.country.encode() if isinstance(r.country, str) else bytes(r.country)
        st.country[0], st.country[1] = cc[0], cc[1]
        st.asn = r.asn
    keep = [fields]
    if header_names:
        spans = (_abi.Span * len(header_names))()
        vals = []
        for k, name in enumerate(header_names):
            v = (r.headers or {}).get(name)
            if v is None:
                continue
            v = v.encode() if isinstance(v, str) else bytes(v)
            vals.append(v)
            spans[k].data = v
            spans[k].len = len(v)
        st.n_headers = len(header_names)
        st.headers = C.cast(spans, C.c_void_p)
        keep += [spans, vals]
    return st, keep


class MicroBatcher:
    """Deadline micro-batcher over a RuleEngine (pwaf_batcher_*): `evaluate(Request)` blocks until the request's batch — closed at
    `max_batch` requests or after `max_delay_us` — has been evaluated on the GPU. Safe to call from many threads (ctypes drops the GIL)."""

    def __init__(self, engine: "RuleEngine", max_batch: int = 4096, max_delay_us: int = 200):
        self._engine = engine  # keeps the engine alive
        h = C.c_void_p()
        rc = lib().pwaf_batcher_create(engine._h, max_batch, max_delay_us, C.byref(h))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        self._h = h

    def evaluate(self, request: Request) -> Verdict:
        st, _keep = _request_struct(request, self._engine.header_names)
        out = _abi.Verdict()
        rc = lib().pwaf_batcher_evaluate(self._h, C.byref(st), C.byref(out))
        if rc != 0:
            _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
        return verdict_from_record({"action": out.action, "rule_idx": out.rule_idx})

    def stats(self) -> Tuple[int, int]:
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        lib().pwaf_batcher_stats(self._h, C.byref(nb), C.byref(nr))
        return nb.value, nr.value

    def close(self):
        if self._h:
            lib().pwaf_batcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def native_batcher_latency(engine: "RuleEngine", batch: RequestBatch, threads: int = 64, per_thread: int = 150, max_batch: int = 4096, max_delay_us: int = 200, pool: int = 256) -> dict:
    """Per-request latency through the deadline micro-batcher as a NATIVE host sees it: tools/libbatcher_bench.so (C++, `threads`
    std::threads calling pwaf_batcher_evaluate back to back over the first `pool` requests of `batch`). Measurement only."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "libbatcher_bench.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing (built by __graft_entry__.build())")
    bb = C.CDLL(path)
    bb.bb_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_abi.Request), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_double)]
    pool = min(pool, batch.n)
    reqs = (_abi.Request * pool)()
    keep = []
    names = engine.header_names
    for i in range(pool):
        hdrs = {h: batch.header_bytes(h, i) for h in names} if names else None
        r = Request(host=batch.field_bytes(0, i), url=batch.field_bytes(1, i), path=batch.field_bytes(2, i), method=batch.field_bytes(3, i), user_agent=batch.field_bytes(4, i),
                    ip="203.0.113.%d" % (i % 250 + 1), remote_port=int(batch.port[i]), headers=hdrs)
        st, k = _request_struct(r, names)
        reqs[i] = st
        keep.append(k)
    mb = MicroBatcher(engine, max_batch=max_batch, max_delay_us=max_delay_us)
    n = threads * per_thread
    lat = (C.c_double * n)()
    secs = C.c_double(0)
    fn = C.cast(lib().pwaf_batcher_evaluate, C.c_void_p)
    failed = bb.bb_run(fn, mb._h, reqs, pool, threads, per_thread, lat, None, C.byref(secs))
    nb, nr = mb.stats()
    mb.close()
    xs = sorted(lat)
    pct = lambda p: xs[min(n - 1, int(round(p / 100.0 * (n - 1))))]  # noqa: E731
    return {"caller_threads": threads, "callers": "native (std::thread, tools/batcher_bench.cpp)", "requests": n, "failed": failed, "max_batch": max_batch, "deadline_us": max_delay_us,
            "batches": nb, "requests_per_s": n / secs.value if secs.value > 0 else 0.0, "latency_ms": {"p50": pct(50), "p99": pct(99), "max": xs[-1]}}


def geoip_from_mmdb(content: bytes, path: str = "geoip.mmdb") -> np.ndarray:
    """MaxMind DB file content -> GEOIP_DTYPE prefix table for RuleEngine(..., geoip=...) (pingoo/geoip.rs:43-91). A `path` ending in
    ".zst" means the content is ZSTD-compressed (geoip.rs:49-55)."""
    from .batch import GEOIP_DTYPE

    ptr, n = C.POINTER(_abi.GeoipEntry)(), C.c_size_t(0)
    rc = lib().pwaf_geoip_from_file_image(path.encode(), content, len(content), C.byref(ptr), C.byref(n))
    if rc != 0:
        _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
    try:
        out = np.zeros(n.value, dtype=GEOIP_DTYPE)
        if n.value:
            C.memmove(out.ctypes.data, ptr, n.value * GEOIP_DTYPE.itemsize)
    finally:
        lib().pwaf_geoip_free(ptr)
    return out


def zstd_decompress(data: bytes) -> bytes:
    out, n = C.c_void_p(), C.c_size_t(0)
    rc = lib().pwaf_zstd_decompress(data, len(data), C.byref(out), C.byref(n))
    if rc != 0:
        _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
    try:
        return C.string_at(out, n.value)
    finally:
        lib().pwaf_buffer_free(out)


def parse_list_csv(text) -> List[str]:
    """List file content -> items (pingoo/lists.rs:62-117: CSV, no header, 1-2 columns, first column trimmed)."""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    items, n = C.POINTER(C.c_char_p)(), C.c_size_t(0)
    rc = lib().pwaf_list_parse_csv(raw, len(raw), C.byref(items), C.byref(n))
    if rc != 0:
        _raise(rc, lib().pwaf_last_error().decode(errors="replace"))
    try:
        return [items[k].decode(errors="surrogateescape") for k in range(n.value)]
    finally:
        lib().pwaf_list_free(items, n.value)


def verdict_from_record(v) -> Verdict:
    ridx = int(v["rule_idx"])
    gate = {_abi.RULE_UA_GATE: "user_agent", _abi.RULE_CAPTCHA_ENDPOINT: "captcha_endpoint"}.get(ridx)
    return Verdict(Decision(int(v["action"])), None if ridx >= _abi.RULE_CAPTCHA_ENDPOINT else ridx, gate)


class DeviceBatch:
    """A RequestBatch uploaded to HBM as torch tensors (PyTorch is only the allocator/stream owner here)."""

    def __init__(self, batch: RequestBatch, device="cuda:0"):
        import torch

        self.device = torch.device(device)
        self.n = batch.n
        self.algorithmic_bytes = batch.algorithmic_bytes()
        self.field_bytes = [int(o[-1]) - int(o[0]) for o in batch.offsets]
        self.arena_bytes = [int(o[-1]) for o in batch.offsets]  # offsets[n]: what pwaf_batch.field_bytes carries
        t = lambda a: torch.from_numpy(a).to(self.device, non_blocking=False)  # noqa: E731
        self.data = [t(d) for d in batch.data]
        self.offsets = [t(o.view(np.int32)) for o in batch.offsets]
        self.ip = t(batch.ip)
        self.ip_is_v6 = t(batch.ip_is_v6)
        self.port = t(batch.port.view(np.int16))
        self.flags = t(batch.flags)
        self.asn = None if batch.asn is None else t(batch.asn.view(np.int32))
        self.country = None if batch.country is None else t(batch.country.view(np.int16))
        self.headers = {name: (t(d), t(o.view(np.int32)), int(o[-1])) for name, (d, o) in batch.headers.items()}
        self._empty = None

    def as_struct(self, header_names: Sequence[str] = ()) -> _abi.Batch:
        import torch

        b = _abi.Batch()
        b.struct_size = C.sizeof(_abi.Batch)
        b.n = self.n
        b.memory = _abi.MEM_DEVICE
        if header_names:
            cols = (_abi.StrCol * len(header_names))()
            hb = (C.c_uint32 * len(header_names))()
            for k, name in enumerate(header_names):
                if name in self.headers:
                    d, o, nbytes = self.headers[name]
                    cols[k].data, cols[k].offsets, hb[k] = d.data_ptr(), o.data_ptr(), nbytes
                # (a missing name stays NULL: the engine reads it as the empty string for every request)
            b.n_headers = len(header_names)
            b.headers = cols
            b.header_bytes = hb
            b._keep = [cols, hb]
        for f in range(_abi.N_FIELDS):
            b.field[f].data = self.data[f].data_ptr()
            b.field[f].offsets = self.offsets[f].data_ptr()
            b.field_bytes[f] = self.arena_bytes[f]
        b.ip = self.ip.data_ptr()
        b.ip_is_v6 = self.ip_is_v6.data_ptr()
        b.port = self.port.data_ptr()
        b.flags = self.flags.data_ptr()
        b.asn = None if self.asn is None else self.asn.data_ptr()
        b.country = None if self.country is None else self.country.data_ptr()
        return b
