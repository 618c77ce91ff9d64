"""ctypes mirror of include/pwaf.h (struct layouts and constants only — no behaviour).

Kept in one place so the product wrapper (pingoo_amd.engine) and the test-side oracle wrapper
(oracle/pyoracle.py) marshal the *same* bytes across their respective C ABIs.
"""
import ctypes as C

ABI_VERSION = 3

OK = 0
E_INVALID_ARG = -1
E_SYNTAX = -2
E_UNSUPPORTED = -3
E_LIST = -4
E_DEVICE = -5
E_BATCH = -6
E_NOMEM = -7

ACTION_ALLOW, ACTION_BLOCK, ACTION_CAPTCHA, ACTION_BYPASS = 0, 1, 2, 3
RULE_NONE = 0xFFFFFFFF
RULE_UA_GATE = 0xFFFFFFFE
RULE_CAPTCHA_ENDPOINT = 0xFFFFFFFD
RULE_ACTION_BLOCK, RULE_ACTION_CAPTCHA = 1, 2

LIST_STRING, LIST_INT, LIST_IP = 0, 1, 2
OPT_NO_UA_GATE, OPT_NO_CAPTCHA_BYPASS, OPT_NO_PREFILTER, OPT_STRICT, OPT_FILTER_STRIDE2, OPT_LENIENT, OPT_NO_RESIDUAL, OPT_GLOBAL_VERDICT_TABLES, OPT_NO_CONFIRM, OPT_NO_RESIDUAL_JIT = 1, 2, 4, 8, 16, 32, 64, 128, 512, 1024
OPT_DENSE_VERDICT, OPT_TINY_VERDICT_SLOTS, OPT_NO_DIR_SUMMARY, OPT_SPARSE_VERDICT, OPT_EAGER_CMP, OPT_NO_DENSE_SWITCH = 2048, 4096, 8192, 16384, 32768, 65536
W_PARTIAL = 1
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_CAPTCHA_VERIFIED = 1
N_FIELDS = 5
FIELD_NAMES = ("host", "url", "path", "method", "user_agent")
ARENA_PAD = 16


class RuleDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("expression", C.c_char_p),
        ("actions", C.POINTER(C.c_uint8)),
        ("n_actions", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class ListDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("type", C.c_uint32),
        ("n_items", C.c_uint32),
        ("items", C.POINTER(C.c_char_p)),
    ]


class GeoipEntry(C.Structure):
    _fields_ = [
        ("addr", C.c_uint8 * 16),
        ("prefix_len", C.c_uint8),
        ("is_v6", C.c_uint8),
        ("country", C.c_uint8 * 2),
        ("asn", C.c_uint32),
    ]


class GeoipTable(C.Structure):
    _fields_ = [("entries", C.POINTER(GeoipEntry)), ("n_entries", C.c_size_t)]


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("device", C.c_int32),
        ("lds_table_budget", C.c_uint32),
        ("max_dfa_states", C.c_uint32),
        ("max_table_bytes", C.c_uint32),
        ("reserved", C.c_uint32 * 2),
    ]


class CompileError(C.Structure):
    _fields_ = [("code", C.c_int32), ("rule_index", C.c_uint32), ("message", C.c_char * 248)]


class StrCol(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n", C.c_uint32),
        ("memory", C.c_uint32),
        ("n_headers", C.c_uint32),
        ("field", StrCol * N_FIELDS),
        ("ip", C.c_void_p),
        ("ip_is_v6", C.c_void_p),
        ("port", C.c_void_p),
        ("flags", C.c_void_p),
        ("asn", C.c_void_p),
        ("country", C.c_void_p),
        ("field_bytes", C.c_uint32 * N_FIELDS),
        ("reserved", C.c_uint32),
        ("headers", C.POINTER(StrCol)),
        ("header_bytes", C.POINTER(C.c_uint32)),
    ]


class Verdict(C.Structure):
    _fields_ = [("action", C.c_uint8), ("pad", C.c_uint8 * 3), ("rule_idx", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [("by_action", C.c_uint64 * 4)]


class Request(C.Structure):
    _fields_ = [
        ("host", C.c_char_p), ("url", C.c_char_p), ("path", C.c_char_p), ("method", C.c_char_p), ("user_agent", C.c_char_p),
        ("host_len", C.c_uint32), ("url_len", C.c_uint32), ("path_len", C.c_uint32), ("method_len", C.c_uint32), ("user_agent_len", C.c_uint32),
        ("ip", C.c_uint8 * 16),
        ("ip_is_v6", C.c_uint8),
        ("flags", C.c_uint8),
        ("port", C.c_uint16),
        ("has_geoip", C.c_uint8),
        ("country", C.c_uint8 * 2),
        ("pad", C.c_uint8),
        ("asn", C.c_uint32),
        ("n_headers", C.c_uint32),
        ("headers", C.c_void_p),  # pwaf_span[n_headers]
    ]


class Span(C.Structure):
    _fields_ = [("data", C.c_char_p), ("len", C.c_uint32), ("reserved", C.c_uint32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("alg_bytes", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [
        ("n_rules", C.c_uint32), ("n_atoms", C.c_uint32), ("n_scan_atoms", C.c_uint32), ("n_numeric_atoms", C.c_uint32),
        ("n_dfa_groups", C.c_uint32), ("n_dfa_states_total", C.c_uint32), ("max_dfa_states", C.c_uint32), ("dfa_table_bytes_total", C.c_uint32),
        ("n_ip_lists", C.c_uint32), ("ipset_trie_nodes", C.c_uint32), ("geo_trie_nodes", C.c_uint32), ("n_dnf_literals", C.c_uint32),
        ("n_warnings", C.c_uint32), ("n_filtered_groups", C.c_uint32), ("n_gated_groups", C.c_uint32), ("n_confirm_literals", C.c_uint32),
    ]


class Marshalled:
    """Keeps the Python objects backing a set of C descriptors alive."""

    def __init__(self):
        self.keep = []


def marshal_rules(rules, m):
    """rules: iterable of (name, expression-or-None, [action codes]) -> (RuleDesc array, n)."""
    rules = list(rules)
    arr = (RuleDesc * max(1, len(rules)))()
    for i, (name, expr, actions) in enumerate(rules):
        acts = (C.c_uint8 * max(1, len(actions)))(*actions)
        m.keep.append(acts)
        arr[i].name = name.encode() if isinstance(name, str) else name
        arr[i].expression = None if expr is None else (expr.encode() if isinstance(expr, str) else expr)
        arr[i].actions = C.cast(acts, C.POINTER(C.c_uint8))
        arr[i].n_actions = len(actions)
    m.keep.append(arr)
    return arr, len(rules)


def marshal_lists(lists, m):
    """lists: dict name -> (type code, [item strings]) -> (ListDesc array, n)."""
    lists = lists or {}
    arr = (ListDesc * max(1, len(lists)))()
    for i, (name, (ltype, items)) in enumerate(lists.items()):
        enc = [(s.encode() if isinstance(s, str) else s) for s in items]
        carr = (C.c_char_p * max(1, len(enc)))(*enc)
        m.keep.append((enc, carr))
        arr[i].name = name.encode()
        arr[i].type = ltype
        arr[i].n_items = len(enc)
        arr[i].items = C.cast(carr, C.POINTER(C.c_char_p))
    m.keep.append(arr)
    return arr, len(lists)


def marshal_geoip(entries, m):
    """entries: numpy structured array (see pingoo_amd.batch.GEOIP_DTYPE) or None -> pointer or None."""
    if entries is None:
        return None
    import numpy as np

    entries = np.ascontiguousarray(entries)
    assert entries.dtype.itemsize == C.sizeof(GeoipEntry), entries.dtype
    t = GeoipTable()
    t.entries = C.cast(entries.ctypes.data, C.POINTER(GeoipEntry))
    t.n_entries = len(entries)
    m.keep.append((entries, t))
    return C.pointer(t)
