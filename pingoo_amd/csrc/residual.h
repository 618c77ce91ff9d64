// residual.h — the per-request bytecode interpreter for rules the column compiler cannot take.
//
// The reference evaluates ANY valid expression per request (Program::execute, pingoo/rules.rs:37-51). The column compiler
// (compile.cpp) turns the predicates WAF rules are made of into atoms, DFAs and tries; expressions that compute with request values —
// arithmetic on lengths / ports / ASNs, string concatenation, lists and maps holding request values, orderings and equalities between
// two request values, conditionals that select non-Bool values, boolean structures whose DNF explodes — have no column form. Such a
// rule is compiled WHOLE into a small stack program (residual.cpp) and evaluated by residual_kernel with one lane per request:
// dynamically typed values, the interpreter's left-to-right short circuit and error propagation, result
// `Bool(true)` = the rule matches. The result lands in the hit record of a pseudo pass, one column per residual rule; the verdict
// kernel treats it like any other atom, so rule order / first-match-wins / actions are untouched.
//
// This file is shared by the device kernel (kernels.hip) and by a TEST-ONLY host build (tests/rvm_host.cpp): the same interpreter
// source runs under g++ so that the CPU suite can fuzz it against the oracle without a GPU. The product library exports no host
// evaluation entry point.
//
// Not covered (such a rule stays PWAF_E_UNSUPPORTED): `matches` with a pattern that is not a String literal (the regex would
// have to be compiled per request), a literal pattern whose DFA exceeds the residual budget, http_request[...] and the headers map
// with a computed key (client[k], lists[k] and membership of a computed key in http_request / client / lists are taken: the compiler
// builds the closed key set as a Map value, residual.cpp: gen_ctx_map), nesting / stack / heap beyond the limits below.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstddef>
#include <cstdint>
#else  // hiprtc (the specialized form, residual_jit.cpp): no host include path to rely on — the compiler's own type macros
typedef __UINT8_TYPE__ uint8_t;
typedef __UINT16_TYPE__ uint16_t;
typedef __UINT32_TYPE__ uint32_t;
typedef __UINT64_TYPE__ uint64_t;
typedef __INT8_TYPE__ int8_t;
typedef __INT16_TYPE__ int16_t;
typedef __INT32_TYPE__ int32_t;
typedef __INT64_TYPE__ int64_t;
typedef __SIZE_TYPE__ size_t;
#define INT64_MIN (-__INT64_MAX__ - 1)
#define INT64_MAX __INT64_MAX__
#endif

#if defined(__HIPCC__)
#define PWAF_HD __host__ __device__ __forceinline__
#define PWAF_HD_NOINLINE __host__ __device__ __noinline__
#else
#define PWAF_HD inline
#define PWAF_HD_NOINLINE inline
#endif
// The string helpers: out of line in the interpreter (one copy, called from a dozen places of a program that is decoded at run time);
// in line in the SPECIALIZED form (residual_jit.cpp, compiled by hiprtc), where the source of nearly every string — which field, a
// constant, a rope — is a literal at the call site and the dispatch inside them folds away.
#if defined(__HIPCC_RTC__)
#define PWAF_HD_STR PWAF_HD
// Loops over the segments of a rope: unrolled in the specialized form, where the segment count is a literal once the concatenation
// that built the rope is inlined — the lane's heap is then indexed by constants only and lives in registers, not in scratch memory
// (measured: 0.68 ms per 10M requests for ONE rule matching a regex against host + ":" + method with the heap in scratch).
#define PWAF_UNROLL _Pragma("unroll")
#else
#define PWAF_HD_STR PWAF_HD_NOINLINE
#define PWAF_UNROLL
#endif

namespace pwaf {
namespace rvm {

// ---- value model -----------------------------------------------------------------------------------------------------------------
enum Type : uint32_t { T_ERR = 0, T_NULL, T_BOOL, T_INT, T_FLT, T_STR, T_IP, T_NET, T_LIST, T_CLIST, T_MAP, T_REF };
// T_STR   a = length; p = src << 48 | offset. src: < S_CONST = string column (field id: the request's bytes), S_CONST = the program's
//         string pool, S_INLINE = up to 6 bytes held in p itself (client.country), S_ROPE = a concatenation: offset = index of its
//         first segment in the lane's heap | segments << 24 (segments are plain T_STR values)
// T_LIST  a = item count, p = index of the first item: bit 63 clear = the program's constant values, set = the lane's heap
// T_CLIST a = configured list id (lists["name"]): typed items in the program's list tables
// T_NET   p = index into the program's network table (an item of an Ip list; there is no literal syntax)
// T_MAP   like T_LIST with 2 * a values: key (T_STR), value, key, value ...; keys are distinct
// T_REF   only as a VALUE slot of a CONSTANT map (a context map as a value: residual.cpp gen_ctx_map): a = the instruction that reads the
//         request (R_FIELD, R_COUNTRY, R_PORT, R_ASN, R_IP), p = its operand; replaced by the value itself when the slot is read (deref)
struct Val {
    uint32_t t, a;
    uint64_t p;
};
static constexpr uint32_t S_CONST = 0xF0, S_INLINE = 0xF1, S_ROPE = 0xF2;
static constexpr uint64_t kHeapBit = 1ull << 63;

// ---- program -----------------------------------------------------------------------------------------------------------------------
enum Op : uint8_t {
    R_END = 0,
    R_CONST,    // b = index into consts
    R_FIELD,    // b = string column (0-4 fixed fields, 5+ header columns)
    R_COUNTRY, R_PORT, R_ASN, R_IP,
    R_CLIST,    // b = configured list id
    R_INDEX,    // [obj, idx] -> item
    R_SELECT,   // a map value [obj] . const key b (string const index)
    R_CALL,     // a = function (Fn), b = argc << 12 | aux (matches: regex table id, 0xFFF = invalid pattern, 0x800 | k = pattern SET k: the
                // pattern is computed per request but ranges over a finite set of strings enumerated at compile time — RegexSetDesc)
    R_NOT, R_NEG,
    R_BIN,      // a = BinOp (frontend.h numbering from B_EQ on)
    R_AND_L,    // b = target: pops the left operand of &&; undecided -> falls through to the right operand
    R_OR_L,
    R_BOOL_CHK, // the right operand of && / || must be a Bool (an error stays an error)
    R_COND,     // b = else target: pops the condition
    R_JMP,      // b = target
    R_MKLIST,   // b = n: pops n values into a heap list
    R_MKMAP,    // b = n pairs: pops 2n values (key, value ...) into a heap map (later duplicates win)
    R_FAIL,     // b = n: pops n values, pushes an execution error (an undeclared method, a wrong argument count: operands are evaluated, then the call fails)
};
enum Fn : uint8_t { FN_CONTAINS = 0, FN_STARTS, FN_ENDS, FN_LENGTH, FN_MATCHES };
struct Ins {
    uint8_t op, a;
    uint16_t b;
};
// PWAF_RVM_HEAP: the specialized form of a rule set (residual_jit.cpp) is compiled with the heap its rules can actually fill
// (Header::heap_items) — a lane's heap that does not fold into registers then costs that much scratch memory, not 1 KiB.
#ifndef PWAF_RVM_HEAP
#define PWAF_RVM_HEAP 64
#endif
static constexpr uint32_t kStack = 24, kHeap = PWAF_RVM_HEAP, kMaxNest = 4, kMaxRope = 16;

struct NetItem {
    uint8_t addr[16];
    uint8_t prefix, v6, pad[2];
};
struct ListDesc {     // one configured list
    uint32_t type;    // PWAF_LIST_STRING / _INT / _IP
    uint32_t first, n;  // STRING: entries [first, first + n) of lstr (offset, length pairs into strpool); INT: of lints; IP: of nets
    uint32_t pad;
};
struct RegexDesc {    // one literal `matches` pattern compiled to a DFA (a single-pattern DfaGroup, see program.h)
    uint32_t trans;   // byte offset in the blob of uint16 trans[n_states][n_classes]
    uint32_t classmap;  // ... of 256 class bytes
    uint32_t flags;   // ... of n_states bytes: bit 0 = entering the state is a match, bit 1 = ending the haystack in it is a match, bit 2 = dead (neither can be reached any more)
    uint32_t n_classes;
    uint32_t umap;    // SCALAR MODE (dfa.cpp): byte offset of the scalar-value -> class map ([stage1 u16 x 8704][stage2]: csrc/utf8.h), 0 = the table reads bytes
    uint32_t ill_class;  // ... and the class of a byte that begins no well-formed sequence
};
// `s.matches(p)` with a pattern that is not a literal but can only be one of finitely many strings (a conditional between literals, an item of
// a configured String list, concatenations of such): every candidate is compiled when the engine is created; per request the pattern's
// VALUE selects the table by string equality (round 6; the reference compiles the pattern per evaluation, pingoo/rules.rs:37-51).
struct RegexSetDesc {
    uint32_t first, n;  // entries [first, first + n) of rxitems: {offset into strpool, length, regex id (0xFFF: an invalid pattern), 0}
};
// The whole residual program of a rule set is ONE blob (uploaded as is): header, then sections at the header's byte offsets.
struct Header {
    uint32_t magic;          // 'RVM1'
    uint32_t n_rules;        // residual rules = columns of the pseudo pass
    uint32_t rules;          // -> uint32 entry[n_rules]: first instruction of each rule
    uint32_t code;           // -> Ins[]
    uint32_t consts;         // -> Val[]
    uint32_t strpool;        // -> bytes
    uint32_t lists;          // -> ListDesc[]
    uint32_t lstr;           // -> uint32 pairs (offset into strpool, length)
    uint32_t lints;          // -> int64[]
    uint32_t nets;           // -> NetItem[]
    uint32_t regexes;        // -> RegexDesc[]
    uint32_t needs_geo;      // some rule reads client.asn / client.country
    uint32_t total_bytes;
    uint32_t heap_items;     // no rule puts more values than this on its lane's heap (the compiler's static bound; <= the interpreter's kHeap)
    uint32_t rxsets;         // -> RegexSetDesc[]
    uint32_t rxitems;        // -> uint32 x 4 per candidate pattern
};

// ---- request view ------------------------------------------------------------------------------------------------------------------
struct Req {
    const uint8_t *const *data;   // per string column: arena
    const uint32_t *const *off;   // ... and offsets (n + 1)
    uint32_t r;                   // request index
    const uint8_t *ip;            // 16 bytes
    uint32_t v6, port, asn;
    uint32_t country;             // two bytes, memory order
};

struct Machine {
    const uint8_t *blob;
    const Header *h;
    Req q;
    Val heap[kHeap];
    uint32_t heap_n;
};

// ---- helpers -------------------------------------------------------------------------------------------------------------------------
PWAF_HD Val mk(uint32_t t, uint32_t a = 0, uint64_t p = 0) { Val v; v.t = t; v.a = a; v.p = p; return v; }
PWAF_HD Val mk_bool(bool b) { return mk(T_BOOL, 0, b ? 1u : 0u); }
PWAF_HD Val mk_int(int64_t i) { return mk(T_INT, 0, (uint64_t)i); }
PWAF_HD Val mk_flt(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return mk(T_FLT, 0, u); }
PWAF_HD double as_flt(const Val &v) { double d; __builtin_memcpy(&d, &v.p, 8); return d; }
PWAF_HD uint32_t str_src(const Val &v) { return (uint32_t)(v.p >> 48); }

template <class T>
PWAF_HD const T *section(const Machine &m, uint32_t off) { return reinterpret_cast<const T *>(m.blob + off); }
PWAF_HD const Val *items_of(const Machine &m, const Val &l) {
    return (l.p & kHeapBit) ? &m.heap[(uint32_t)(l.p & 0xFFFFFFFFu)] : section<Val>(m, m.h->consts) + (uint32_t)(l.p & 0xFFFFFFFFu);
}

// The batch's columns are device (global) memory: said so where their pointers come out of the pointer table, the loads through them
// are global loads instead of flat ones.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
PWAF_HD const T *in_global(const T *p) { return (const T *)(const __attribute__((address_space(1))) T *)p; }
#else
template <class T>
PWAF_HD const T *in_global(const T *p) { return p; }
#endif
// Eight bytes at any address (strings are read a word at a time: a lane's string lives in its own cache lines, so every load
// instruction of a wave touches up to 64 of them — the cost is per instruction, not per byte). Reads up to 7 bytes past the string:
// field arenas carry PWAF_ARENA_PAD, the program image ends in padding, the inline buffer (≤ 6 value bytes) has 16.
typedef uint64_t __attribute__((aligned(1), may_alias)) u64_unaligned;
PWAF_HD uint64_t load8(const uint8_t *p) { return *reinterpret_cast<const u64_unaligned *>(p); }
// Bytes of a non-rope string (nullptr for a rope).
PWAF_HD const uint8_t *flat_ptr(const Machine &m, const Val &s, uint8_t (&inl)[16]) {
    const uint32_t src = str_src(s);
    const uint32_t off = (uint32_t)(s.p & 0xFFFFFFFFu);
    if (src < S_CONST) return in_global(m.q.data[src]) + in_global(m.q.off[src])[m.q.r] + off;
    if (src == S_CONST) return m.blob + m.h->strpool + off;
    if (src == S_INLINE) {
        for (int k = 0; k < 16; k++) inl[k] = k < 6 ? (uint8_t)(s.p >> (8 * k)) : (uint8_t)0;  // (16: load8 at any offset of the ≤ 6 value bytes stays inside)
        return inl;
    }
    return nullptr;
}
// Byte i of any string (ropes: a walk over the segments — residual rules are rare, clarity over speed).
PWAF_HD_STR uint8_t str_byte(const Machine &m, const Val &s, uint32_t i) {
    uint8_t inl[16];
    if (str_src(s) != S_ROPE) return flat_ptr(m, s, inl)[i];
    const uint32_t first = (uint32_t)(s.p & 0xFFFFFFu), nseg = (uint32_t)((s.p >> 24) & 0xFFu);
    PWAF_UNROLL
    for (uint32_t k = 0; k < nseg; k++) {
        const Val &sg = m.heap[first + k];
        if (i < sg.a) return flat_ptr(m, sg, inl)[i];
        i -= sg.a;
    }
    return 0;
}
PWAF_HD_STR bool str_eq_at(const Machine &m, const Val &h, uint32_t at, const Val &n) {  // h[at, at + n.a) == n (caller checks the bounds)
    if (str_src(h) != S_ROPE && str_src(n) != S_ROPE) {  // (decided by the KIND of the sources, which the specialized form usually knows at compile time)
        uint8_t i1[16], i2[16];
        const uint8_t *ph = flat_ptr(m, h, i1), *pn = flat_ptr(m, n, i2);
        for (uint32_t k = 0; k < n.a; k += 8) {
            uint64_t x = load8(ph + at + k) ^ load8(pn + k);
            const uint32_t left = n.a - k;
            if (left < 8) x &= (1ull << (8 * left)) - 1ull;
            if (x) return false;
        }
        return true;
    }
    for (uint32_t k = 0; k < n.a; k++)
        if (str_byte(m, h, at + k) != str_byte(m, n, k)) return false;
    return true;
}
PWAF_HD int str_cmp(const Machine &m, const Val &a, const Val &b) {  // bytewise, like std::string_view::compare
    const uint32_t n = a.a < b.a ? a.a : b.a;
    if (str_src(a) != S_ROPE && str_src(b) != S_ROPE) {
        uint8_t i1[16], i2[16];
        const uint8_t *pa = flat_ptr(m, a, i1), *pb = flat_ptr(m, b, i2);
        for (uint32_t k = 0; k < n; k += 8) {
            const uint64_t wa = load8(pa + k), wb = load8(pb + k);
            uint64_t x = wa ^ wb;
            const uint32_t left = n - k;
            if (left < 8) x &= (1ull << (8 * left)) - 1ull;
            if (x) {
                const uint32_t sh = (uint32_t)__builtin_ctzll(x) & ~7u;  // the first differing byte (little endian: lowest address = lowest byte)
                return ((wa >> sh) & 0xFFu) < ((wb >> sh) & 0xFFu) ? -1 : 1;
            }
        }
        return a.a < b.a ? -1 : a.a > b.a ? 1 : 0;
    }
    for (uint32_t k = 0; k < n; k++) {
        const uint8_t x = str_byte(m, a, k), y = str_byte(m, b, k);
        if (x != y) return x < y ? -1 : 1;
    }
    return a.a < b.a ? -1 : a.a > b.a ? 1 : 0;
}
PWAF_HD bool str_find(const Machine &m, const Val &h, const Val &n) {
    if (n.a > h.a) return false;
    for (uint32_t s = 0; s + n.a <= h.a; s++)
        if (str_eq_at(m, h, s, n)) return true;
    return false;
}
PWAF_HD bool net_contains(const NetItem &nt, const uint8_t *ip, uint32_t v6) {  // ipnetwork semantics: family must match, mask compare
    if ((nt.v6 != 0) != (v6 != 0)) return false;
    const uint32_t bytes = v6 ? 16u : 4u;
    uint32_t left = nt.prefix;
    for (uint32_t k = 0; k < bytes && left; k++) {
        const uint32_t take = left >= 8 ? 8u : left;
        const uint8_t mask = (uint8_t)(0xFFu << (8u - take));
        if ((nt.addr[k] & mask) != (ip[k] & mask)) return false;
        left -= take;
    }
    return true;
}

// A value slot of a map, read: a constant map's T_REF slots name a request value (the map itself costs no stack slot and no heap item,
// whatever its size — a headers map of 64 names as a value; until round 6 its entries were pushed and popped: 2 x 64 stack slots).
PWAF_HD Val op_field(const Machine &m, uint32_t f);
PWAF_HD Val op_country(const Machine &m);
PWAF_HD Val op_port(const Machine &m);
PWAF_HD Val op_asn(const Machine &m);
PWAF_HD Val deref(const Machine &m, const Val &v) {
    if (v.t != T_REF) return v;
    switch (v.a) {
        case R_FIELD: return op_field(m, (uint32_t)v.p);
        case R_COUNTRY: return op_country(m);
        case R_PORT: return op_port(m);
        case R_ASN: return op_asn(m);
        default: return mk(T_IP);
    }
}

// val_eq (D4: cross-type equality is false). Lists and maps nest at most kMaxNest deep (the compiler rejects deeper literals), so the
// recursion is unrolled at compile time through the depth parameter instead of recursing on the device.
template <int D>
struct Eq {
    static PWAF_HD_NOINLINE bool eq(const Machine &m, const Val &a, const Val &b) {
        if (a.t == T_INT && b.t == T_FLT) return (double)(int64_t)a.p == as_flt(b);
        if (a.t == T_FLT && b.t == T_INT) return as_flt(a) == (double)(int64_t)b.p;
        if (a.t != b.t) return false;
        switch (a.t) {
            case T_NULL: return true;
            case T_BOOL: case T_INT: return a.p == b.p;
            case T_FLT: return as_flt(a) == as_flt(b);
            case T_STR: return a.a == b.a && str_eq_at(m, a, 0, b);
            case T_IP: return true;  // the only Ip value is the request's client.ip
            case T_NET: {
                const NetItem *nets = section<NetItem>(m, m.h->nets);
                const NetItem &p = nets[a.p], &q = nets[b.p];
                if (p.prefix != q.prefix || p.v6 != q.v6) return false;
                for (int k = 0; k < 16; k++) if (p.addr[k] != q.addr[k]) return false;
                return true;
            }
            case T_LIST: {
                if (a.a != b.a) return false;
                const Val *ia = items_of(m, a), *ib = items_of(m, b);
                for (uint32_t k = 0; k < a.a; k++)
                    if (!Eq<D - 1>::eq(m, ia[k], ib[k])) return false;
                return true;
            }
            case T_MAP: {  // same key set, equal values
                if (a.a != b.a) return false;
                const Val *ia = items_of(m, a), *ib = items_of(m, b);
                for (uint32_t k = 0; k < a.a; k++) {
                    bool found = false;
                    for (uint32_t j = 0; j < b.a && !found; j++)
                        if (ia[2 * k].a == ib[2 * j].a && str_eq_at(m, ia[2 * k], 0, ib[2 * j])) {
                            found = true;
                            if (!Eq<D - 1>::eq(m, deref(m, ia[2 * k + 1]), deref(m, ib[2 * j + 1]))) return false;
                        }
                    if (!found) return false;
                }
                return true;
            }
            default: return false;  // (T_CLIST: the compiler rejects comparisons of configured lists)
        }
    }
};
template <>
struct Eq<-1> {
    static PWAF_HD bool eq(const Machine &, const Val &, const Val &) { return false; }  // unreachable: nesting is bounded at compile time
};
PWAF_HD bool val_eq(const Machine &m, const Val &a, const Val &b) {
    // scalars in line (the same answers as the general comparison above): in the specialized form one operand's type is usually known
    // at compile time, and the out-of-line general comparison — with the registers it needs — is then never referenced
    if (a.t == T_INT && b.t == T_FLT) return (double)(int64_t)a.p == as_flt(b);
    if (a.t == T_FLT && b.t == T_INT) return as_flt(a) == (double)(int64_t)b.p;
    if (a.t != b.t) return false;
    switch (b.t) {
        case T_NULL: case T_IP: return true;
        case T_BOOL: case T_INT: return a.p == b.p;
        case T_FLT: return as_flt(a) == as_flt(b);
        case T_STR: return a.a == b.a && str_eq_at(m, a, 0, b);
        default: return Eq<(int)kMaxNest>::eq(m, a, b);  // networks, lists, maps
    }
}

// -1 / 0 / 1, or 2 when not comparable (D5)
PWAF_HD int val_cmp(const Machine &m, const Val &a, const Val &b) {
    const bool an = a.t == T_INT || a.t == T_FLT, bn = b.t == T_INT || b.t == T_FLT;
    if (a.t == T_INT && b.t == T_INT) return (int64_t)a.p < (int64_t)b.p ? -1 : (int64_t)a.p > (int64_t)b.p ? 1 : 0;
    if (an && bn) {
        const double x = a.t == T_INT ? (double)(int64_t)a.p : as_flt(a), y = b.t == T_INT ? (double)(int64_t)b.p : as_flt(b);
        if (x != x || y != y) return 2;
        return x < y ? -1 : x > y ? 1 : 0;
    }
    if (a.t == T_STR && b.t == T_STR) return str_cmp(m, a, b);
    return 2;
}

// list.contains(x) / x in list (D10, D12: a network item contains an Ip by CIDR containment)
PWAF_HD_STR bool list_contains(const Machine &m, const Val &l, const Val &x) {
    if (l.t == T_LIST) {
        const Val *it = items_of(m, l);
        const NetItem *nets = section<NetItem>(m, m.h->nets);
        for (uint32_t k = 0; k < l.a; k++) {
            if (it[k].t == T_NET && x.t == T_IP) {
                if (net_contains(nets[it[k].p], m.q.ip, m.q.v6)) return true;
            } else if (val_eq(m, it[k], x)) {
                return true;
            }
        }
        return false;
    }
    const ListDesc &d = section<ListDesc>(m, m.h->lists)[l.a];
    if (d.type == 0 /* PWAF_LIST_STRING */) {
        if (x.t != T_STR) return false;
        const uint32_t *sp = section<uint32_t>(m, m.h->lstr) + 2 * (size_t)d.first;
        for (uint32_t k = 0; k < d.n; k++) {
            if (sp[2 * k + 1] != x.a) continue;
            const Val item = mk(T_STR, sp[2 * k + 1], ((uint64_t)S_CONST << 48) | sp[2 * k]);
            if (str_eq_at(m, x, 0, item)) return true;
        }
        return false;
    }
    if (d.type == 1 /* PWAF_LIST_INT */) {
        const int64_t *li = section<int64_t>(m, m.h->lints) + d.first;
        if (x.t == T_INT) { for (uint32_t k = 0; k < d.n; k++) if (li[k] == (int64_t)x.p) return true; }
        else if (x.t == T_FLT) { const double f = as_flt(x); for (uint32_t k = 0; k < d.n; k++) if ((double)li[k] == f) return true; }
        return false;
    }
    const NetItem *nets = section<NetItem>(m, m.h->nets) + d.first;
    if (x.t == T_IP) { for (uint32_t k = 0; k < d.n; k++) if (net_contains(nets[k], m.q.ip, m.q.v6)) return true; }
    else if (x.t == T_NET) {
        const NetItem &q = section<NetItem>(m, m.h->nets)[x.p];
        for (uint32_t k = 0; k < d.n; k++) {
            bool same = nets[k].prefix == q.prefix && nets[k].v6 == q.v6;
            for (int b = 0; b < 16 && same; b++) same = nets[k].addr[b] == q.addr[b];
            if (same) return true;
        }
    }
    return false;
}
PWAF_HD bool map_has(const Machine &m, const Val &mp, const Val &key, Val *out) {
    const Val *it = items_of(m, mp);
    for (uint32_t k = 0; k < mp.a; k++)
        if (it[2 * k].a == key.a && str_eq_at(m, it[2 * k], 0, key)) {
            if (out) *out = deref(m, it[2 * k + 1]);
            return true;
        }
    return false;
}

// (SCALAR is a template argument so that the walk of a table that reads BYTES keeps the loop — and the registers — it had: the
// specialized program names the mode of each pattern at translation time, residual_jit.cpp)
template <bool SCALAR>
PWAF_HD_STR bool regex_match_t(const Machine &m, uint32_t id, const Val &s) {
    // Table entries (residual.cpp): next state | 0x8000 when entering it decides the match | 0x4000 when it is dead (nothing the rest
    // of the string holds can make the pattern match: an anchored pattern that has failed) — one table load per byte.
    // SCALAR MODE (a class beyond ASCII in the pattern: the regex crate matches scalar values, Cargo.lock:1694-1700): a byte below 0x80
    // is its own symbol, a well-formed sequence is ONE symbol — its scalar's class from the two-stage map — taken when its last byte
    // arrives (the string may be a rope: the sequence may straddle two segments), a lead byte that the next byte does not continue is
    // the ill-formed class and that byte is then read on its own; a continuation byte that continues nothing is the ill-formed class too, as for the
    // table walkers of kernels.hip (utf8.h decodes at the lead byte there and asks of a continuation byte whether a sequence holds it).
    const RegexDesc &d = section<RegexDesc>(m, m.h->regexes)[id];
    const uint16_t *trans = section<uint16_t>(m, d.trans);
    const uint8_t *cm = m.blob + d.classmap, *fl = m.blob + d.flags;
    const uint8_t *um = SCALAR ? m.blob + d.umap : nullptr;
    const uint32_t nc = d.n_classes;
    uint32_t st = 0;
    if (fl[0] & 1u) return true;
    if (fl[0] & 4u) return false;
    uint32_t pend = 0, need = 0, cp = 0;  // scalar mode: bytes of the open sequence still to come, its length, the bits so far
    // segment by segment (a flat string is its own only segment), eight bytes per load
    const bool rope = str_src(s) == S_ROPE;
    const uint32_t first = (uint32_t)(s.p & 0xFFFFFFu), nseg = rope ? (uint32_t)((s.p >> 24) & 0xFFu) : 1u;
    PWAF_UNROLL
    for (uint32_t k = 0; k < nseg; k++) {
        const Val sg = rope ? m.heap[first + k] : s;
        uint8_t inl[16];
        const uint8_t *p = flat_ptr(m, sg, inl);
        for (uint32_t i = 0; i < sg.a; i += 8) {
            uint64_t w = load8(p + i);
            const uint32_t take = sg.a - i < 8u ? sg.a - i : 8u;
            for (uint32_t j = 0; j < take; j++, w >>= 8) {
                const uint32_t b = (uint32_t)w & 0xFFu;
                uint32_t cls = cm[b];
                if (SCALAR && (b >= 0x80u || pend != 0u)) {
                    bool closed = false;  // this byte ended an open sequence: cls is that sequence's symbol
                    if (pend != 0u) {
                        if ((b & 0xC0u) == 0x80u) {  // the open sequence goes on
                            cp = (cp << 6) | (b & 0x3Fu);
                            if (--pend != 0u) continue;
                            const bool bad = (need == 2u && cp < 0x800u) || (need == 3u && (cp < 0x10000u || cp > 0x10FFFFu)) || (cp >= 0xD800u && cp <= 0xDFFFu);
                            cls = bad ? d.ill_class : um[(0x110000u >> 7) * 2u + (uint32_t)reinterpret_cast<const uint16_t *>(um)[cp >> 7] * 128u + (cp & 127u)];
                            closed = true;
                        } else {  // broken off: the lead byte was ill-formed; this byte is read on its own below
                            pend = 0;
                            const uint32_t e0 = trans[st * nc + d.ill_class];
                            if (e0 & 0x8000u) return true;
                            if (e0 & 0x4000u) return false;
                            st = e0;
                        }
                    }
                    if (!closed && pend == 0u && b >= 0xC0u) {  // a lead byte
                        if (b < 0xC2u || b > 0xF4u) cls = d.ill_class;
                        else {
                            need = pend = b >= 0xF0u ? 3u : b >= 0xE0u ? 2u : 1u;
                            cp = b & (0x3Fu >> need);
                            continue;
                        }
                    } else if (!closed && pend == 0u && b >= 0x80u) {
                        cls = d.ill_class;  // a continuation byte that continues nothing: an ill-formed unit (round 6: it used to be skipped)
                    }
                }
                const uint32_t e = trans[st * nc + cls];
                if (e & 0x8000u) return true;
                if (e & 0x4000u) return false;
                st = e;
            }
        }
    }
    if (SCALAR && pend != 0u) {  // the string ends inside a sequence: its lead byte was ill-formed
        const uint32_t e0 = trans[st * nc + d.ill_class];
        if (e0 & 0x8000u) return true;
        st = e0 & 0x3FFFu;
    }
    return (fl[st] & 2u) != 0;
}

PWAF_HD_STR bool regex_match(const Machine &m, uint32_t id, const Val &s) {
    return section<RegexDesc>(m, m.h->regexes)[id].umap ? regex_match_t<true>(m, id, s) : regex_match_t<false>(m, id, s);
}
// recv.matches(<literal pattern>) with the table's mode known (what op_call's FN_MATCHES case does, for the specialized program)
template <bool SCALAR>
PWAF_HD Val op_matches(const Machine &m, uint32_t aux, const Val &recv, const Val &arg) {
    if (recv.t == T_ERR || arg.t == T_ERR || recv.t != T_STR || arg.t != T_STR || aux == 0xFFFu) return mk(T_ERR);  // (an invalid pattern is an execution error)
    return mk_bool(regex_match_t<SCALAR>(m, aux, recv));
}

PWAF_HD Val arith(uint32_t op /* B_ADD.. */, const Val &l, const Val &r, Machine &m);

// ---- the operations of the stack program ------------------------------------------------------------------------------------------------
// One function per instruction, shared by the interpreter below (run_rule: decodes the program per request) and by the SPECIALIZED form
// of a rule set (residual_jit.cpp: the same program translated instruction by instruction into straight-line calls of these functions
// with the stack slots as local variables and the constants as literals, compiled for the device when the engine is created).
// BinOp numbering of frontend.h: B_OR 0, B_AND 1, B_EQ 2, B_NE 3, B_LT 4, B_LE 5, B_GT 6, B_GE 7, B_IN 8, B_ADD 9, B_SUB 10, B_MUL 11, B_DIV 12, B_MOD 13
PWAF_HD Val op_field(const Machine &m, uint32_t f) {
    const uint32_t *off = in_global(m.q.off[f]) + m.q.r;
    return mk(T_STR, off[1] - off[0], (uint64_t)f << 48);
}
PWAF_HD Val op_country(const Machine &m) { return mk(T_STR, 2, ((uint64_t)S_INLINE << 48) | (m.q.country & 0xFFFFu)); }
PWAF_HD Val op_port(const Machine &m) { return mk_int((int64_t)m.q.port); }
PWAF_HD Val op_asn(const Machine &m) { return mk_int((int64_t)m.q.asn); }
PWAF_HD Val op_not(const Val &x) {
    if (x.t == T_ERR) return x;
    return x.t == T_BOOL ? mk_bool(x.p == 0) : mk(T_ERR);
}
PWAF_HD Val op_neg(const Val &x) {
    if (x.t == T_INT) return (int64_t)x.p == INT64_MIN ? mk(T_ERR) : mk_int(-(int64_t)x.p);
    if (x.t == T_FLT) return mk_flt(-as_flt(x));
    return mk(T_ERR);
}
// The left operand of && (is_or false) / || (true), popped: true = it decides (`out` = the result: the operand itself, or an error when
// it is not a Bool) and the right operand is skipped; false = the right operand is evaluated and IS the result (after op_bool_chk).
PWAF_HD bool op_logic_left(bool is_or, const Val &l, Val &out) {
    if (l.t != T_BOOL) { out = mk(T_ERR); return true; }
    if (is_or == (l.p != 0)) { out = l; return true; }
    return false;
}
PWAF_HD Val op_bool_chk(const Val &x) { return x.t == T_BOOL ? x : mk(T_ERR); }  // the right operand of && / || must be a Bool (an error stays an error)
PWAF_HD uint32_t op_cond(const Val &c) { return c.t != T_BOOL ? 2u : c.p == 0 ? 1u : 0u; }  // 0: then branch, 1: else branch, 2: not a Bool (the conditional is an error)
PWAF_HD Val op_mklist(Machine &m, const Val *items, uint32_t n) {
    bool err = false;
    for (uint32_t k = 0; k < n; k++) err = err || items[k].t == T_ERR;
    if (err || m.heap_n + n > kHeap) return mk(T_ERR);  // (the compiler bounds heap use statically; the check here is the safety net behind it)
    const Val v = mk(T_LIST, n, kHeapBit | m.heap_n);
    for (uint32_t k = 0; k < n; k++) m.heap[m.heap_n++] = items[k];
    return v;
}
PWAF_HD Val op_mkmap(Machine &m, const Val *kv /* key, value, key, value ... */, uint32_t n) {
    bool err = false;
    for (uint32_t k = 0; k < 2 * n; k++) err = err || kv[k].t == T_ERR || ((k & 1) == 0 && kv[k].t != T_STR);
    if (err || m.heap_n + 2 * n > kHeap) return mk(T_ERR);
    const uint32_t base = m.heap_n;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < n; k++) {
        const Val &key = kv[2 * k], &val = kv[2 * k + 1];
        bool dup = false;
        for (uint32_t j = 0; j < cnt && !dup; j++)
            if (m.heap[base + 2 * j].a == key.a && str_eq_at(m, m.heap[base + 2 * j], 0, key)) { m.heap[base + 2 * j + 1] = val; dup = true; }  // later duplicates win
        if (!dup) { m.heap[base + 2 * cnt] = key; m.heap[base + 2 * cnt + 1] = val; cnt++; }
    }
    m.heap_n = base + 2 * cnt;
    return mk(T_MAP, cnt, kHeapBit | base);
}
PWAF_HD Val op_index(const Machine &m, const Val &o, const Val &i) {
    Val v = mk(T_ERR);
    if (o.t == T_ERR || i.t == T_ERR) {}
    else if (o.t == T_MAP) { if (i.t == T_STR) { Val out; if (map_has(m, o, i, &out)) v = out; } }
    else if (o.t == T_LIST) { if (i.t == T_INT && (int64_t)i.p >= 0 && i.p < o.a) v = items_of(m, o)[i.p]; }
    else if (o.t == T_CLIST) {
        const ListDesc &d = section<ListDesc>(m, m.h->lists)[o.a];
        if (i.t == T_INT && (int64_t)i.p >= 0 && i.p < d.n) {
            if (d.type == 0) { const uint32_t *sp2 = section<uint32_t>(m, m.h->lstr) + 2 * (size_t)(d.first + i.p); v = mk(T_STR, sp2[1], ((uint64_t)S_CONST << 48) | sp2[0]); }
            else if (d.type == 1) v = mk_int(section<int64_t>(m, m.h->lints)[d.first + i.p]);
            else v = mk(T_NET, 0, d.first + i.p);
        }
    }
    return v;
}
PWAF_HD Val op_select(const Machine &m, const Val &o, const Val &key) {
    if (o.t == T_ERR) return o;
    Val got;
    if (o.t == T_MAP && map_has(m, o, key, &got)) return got;
    return mk(T_ERR);
}
// recv.fn(arg): argc is 0 (length) or 1 (the compiler turns every other arity into R_FAIL); `arg` is ignored when argc == 0
PWAF_HD Val op_call(const Machine &m, uint32_t fn, uint32_t argc, uint32_t aux, const Val &recv, const Val &arg) {
    Val v = mk(T_ERR);
    if (recv.t == T_ERR || (argc && arg.t == T_ERR)) return v;
    switch (fn) {
        case FN_CONTAINS:
            if (argc != 1) break;
            if (recv.t == T_STR) { if (arg.t == T_STR) v = mk_bool(str_find(m, recv, arg)); }
            else if (recv.t == T_LIST || recv.t == T_CLIST) v = mk_bool(list_contains(m, recv, arg));
            else if (recv.t == T_MAP) { if (arg.t == T_STR) v = mk_bool(map_has(m, recv, arg, nullptr)); }
            break;
        case FN_STARTS: case FN_ENDS:
            if (argc != 1 || recv.t != T_STR || arg.t != T_STR) break;
            if (arg.a > recv.a) v = mk_bool(false);
            else v = mk_bool(str_eq_at(m, recv, fn == FN_STARTS ? 0u : recv.a - arg.a, arg));
            break;
        case FN_LENGTH:
            if (argc != 0) break;
            if (recv.t == T_STR || recv.t == T_LIST || recv.t == T_MAP) v = mk_int((int64_t)recv.a);
            else if (recv.t == T_CLIST) v = mk_int((int64_t)section<ListDesc>(m, m.h->lists)[recv.a].n);
            break;
        case FN_MATCHES:
            if (argc != 1 || recv.t != T_STR || arg.t != T_STR || aux == 0xFFFu) break;  // (an invalid pattern is an execution error)
            if (aux & 0x800u) {  // a computed pattern: which of the set's candidates is it?
                const RegexSetDesc &sd = section<RegexSetDesc>(m, m.h->rxsets)[aux & 0x7FFu];
                const uint32_t *it = section<uint32_t>(m, m.h->rxitems) + 4 * (size_t)sd.first;
                uint32_t id = 0xFFFu;  // (a value outside the enumeration cannot occur: the compiler's set is a superset of the expression's values)
                for (uint32_t k = 0; k < sd.n; k++)
                    if (it[4 * k + 1] == arg.a && str_eq_at(m, arg, 0, mk(T_STR, it[4 * k + 1], ((uint64_t)S_CONST << 48) | it[4 * k]))) { id = it[4 * k + 2]; break; }
                if (id == 0xFFFu) break;
                v = mk_bool(regex_match(m, id, recv));
                break;
            }
            v = mk_bool(regex_match(m, aux, recv));
            break;
        default: break;
    }
    return v;
}
PWAF_HD Val op_bin(Machine &m, uint32_t op, const Val &l, const Val &r) {
    Val v = mk(T_ERR);
    if (l.t == T_ERR || r.t == T_ERR) return v;
    switch (op) {
        case 2: v = mk_bool(val_eq(m, l, r)); break;
        case 3: v = mk_bool(!val_eq(m, l, r)); break;
        case 4: case 5: case 6: case 7: {
            const int c = val_cmp(m, l, r);
            if (c != 2) v = mk_bool(op == 4 ? c < 0 : op == 5 ? c <= 0 : op == 6 ? c > 0 : c >= 0);
            break;
        }
        case 8:
            if (r.t == T_LIST || r.t == T_CLIST) v = mk_bool(list_contains(m, r, l));
            else if (r.t == T_MAP && l.t == T_STR) v = mk_bool(map_has(m, r, l, nullptr));
            break;
        default: v = arith(op, l, r, m); break;
    }
    return v;
}
// 1 = the rule's program ends in Bool(true) (the rule matches), 2 = it ends in an execution ERROR (what the reference logs with warn!
// and treats as "no match": pingoo/rules.rs:41-45), 0 = anything else (false, or a non-Bool value: no match, no error).
PWAF_HD uint32_t rule_result(const Val &top) { return top.t == T_ERR ? 2u : (top.t == T_BOOL && top.p == 1) ? 1u : 0u; }

// ---- the interpreter: one rule, one request ---------------------------------------------------------------------------------------------
PWAF_HD_NOINLINE uint32_t run_rule(Machine &m, uint32_t rule) {
    const Ins *code = section<Ins>(m, m.h->code);
    const Val *consts = section<Val>(m, m.h->consts);
    uint32_t pc = section<uint32_t>(m, m.h->rules)[rule];
    Val st[kStack];
    uint32_t sp = 0;
    m.heap_n = 0;
    for (;;) {
        const Ins in = code[pc++];
        switch (in.op) {
            case R_END: return sp != 1 ? 0u : rule_result(st[0]);
            case R_CONST: st[sp++] = consts[in.b]; break;
            case R_FIELD: st[sp++] = op_field(m, in.b); break;
            case R_COUNTRY: st[sp++] = op_country(m); break;
            case R_PORT: st[sp++] = op_port(m); break;
            case R_ASN: st[sp++] = op_asn(m); break;
            case R_IP: st[sp++] = mk(T_IP); break;
            case R_CLIST: st[sp++] = mk(T_CLIST, in.b); break;
            case R_NOT: st[sp - 1] = op_not(st[sp - 1]); break;
            case R_NEG: st[sp - 1] = op_neg(st[sp - 1]); break;
            case R_AND_L: case R_OR_L: {
                const Val l = st[--sp];
                Val out;
                if (op_logic_left(in.op == R_OR_L, l, out)) { st[sp++] = out; pc = in.b; }  // decided by the left operand
                break;
            }
            case R_BOOL_CHK: st[sp - 1] = op_bool_chk(st[sp - 1]); break;
            case R_COND: {
                const uint32_t c = op_cond(st[--sp]);
                if (c == 2u) { st[sp++] = mk(T_ERR); pc = code[in.b - 1].b; }  // (the instruction before the else branch is the then-branch's R_JMP to the end)
                else if (c == 1u) pc = in.b;
                break;
            }
            case R_JMP: pc = in.b; break;
            case R_FAIL: sp -= in.b; st[sp++] = mk(T_ERR); break;
            case R_MKLIST: {
                const uint32_t n = in.b;
                const Val v = op_mklist(m, &st[sp - n], n);
                sp -= n;
                st[sp++] = v;
                break;
            }
            case R_MKMAP: {
                const uint32_t n = in.b;
                const Val v = op_mkmap(m, &st[sp - 2 * n], n);
                sp -= 2 * n;
                st[sp++] = v;
                break;
            }
            case R_INDEX: {
                const Val i = st[--sp], o = st[--sp];
                st[sp++] = op_index(m, o, i);
                break;
            }
            case R_SELECT: st[sp - 1] = op_select(m, st[sp - 1], consts[in.b]); break;
            case R_CALL: {
                const uint32_t argc = in.b >> 12, aux = in.b & 0xFFFu;
                const Val recv = st[sp - 1 - argc], arg = argc ? st[sp - argc] : mk(T_ERR);
                sp -= argc + 1;
                st[sp++] = op_call(m, in.a, argc, aux, recv, arg);
                break;
            }
            case R_BIN: {
                const Val r = st[--sp], l = st[--sp];
                st[sp++] = op_bin(m, in.a, l, r);
                break;
            }
            default: return false;
        }
    }
}

PWAF_HD Val arith(uint32_t op, const Val &l, const Val &r, Machine &m) {
    const Val ERR = mk(T_ERR);
    if (l.t == T_INT && r.t == T_INT) {
        const int64_t a = (int64_t)l.p, b = (int64_t)r.p;
        int64_t o;
        switch (op) {
            case 9: return __builtin_add_overflow(a, b, &o) ? ERR : mk_int(o);
            case 10: return __builtin_sub_overflow(a, b, &o) ? ERR : mk_int(o);
            case 11: return __builtin_mul_overflow(a, b, &o) ? ERR : mk_int(o);
            case 12: if (b == 0 || (a == INT64_MIN && b == -1)) return ERR; return mk_int(a / b);
            case 13: if (b == 0) return ERR; if (a == INT64_MIN && b == -1) return mk_int(0); return mk_int(a % b);
            default: return ERR;
        }
    }
    const bool ln = l.t == T_INT || l.t == T_FLT, rn = r.t == T_INT || r.t == T_FLT;
    if (ln && rn) {  // at least one Float
        const double a = l.t == T_FLT ? as_flt(l) : (double)(int64_t)l.p, b = r.t == T_FLT ? as_flt(r) : (double)(int64_t)r.p;
        switch (op) {
            case 9: return mk_flt(a + b);
            case 10: return mk_flt(a - b);
            case 11: return mk_flt(a * b);
            case 12: return mk_flt(a / b);
            default: return ERR;
        }
    }
    if (op == 9 && l.t == T_STR && r.t == T_STR) {
        // concatenation: a rope over the operands' segments (no bytes are copied)
        // (an empty operand still becomes a segment: the SHAPE of the result — how many segments, where they come from — then depends
        // on the program alone, not on the request, which is what lets the specialized form keep a rope in registers)
        const uint32_t first = m.heap_n;
        uint32_t n = 0;
        {
            const uint32_t nl = str_src(l) == S_ROPE ? (uint32_t)((l.p >> 24) & 0xFFu) : 1u, nr = str_src(r) == S_ROPE ? (uint32_t)((r.p >> 24) & 0xFFu) : 1u;
            if (first + nl + nr > kHeap) return ERR;
        }
        PWAF_UNROLL
        for (int side = 0; side < 2; side++) {
            const Val &s = side == 0 ? l : r;
            if (str_src(s) == S_ROPE) {
                const uint32_t f0 = (uint32_t)(s.p & 0xFFFFFFu), ns = (uint32_t)((s.p >> 24) & 0xFFu);
                PWAF_UNROLL
                for (uint32_t k = 0; k < ns; k++) m.heap[m.heap_n++] = m.heap[f0 + k], n++;
            } else {
                m.heap[m.heap_n++] = s;
                n++;
            }
        }
        return mk(T_STR, l.a + r.a, ((uint64_t)S_ROPE << 48) | ((uint64_t)n << 24) | first);
    }
    if (op == 9 && l.t == T_LIST && r.t == T_LIST) {
        const uint32_t first = m.heap_n;
        if (first + l.a + r.a > kHeap) return ERR;
        const Val *a = items_of(m, l), *b = items_of(m, r);
        for (uint32_t k = 0; k < l.a; k++) m.heap[m.heap_n++] = a[k];
        for (uint32_t k = 0; k < r.a; k++) m.heap[m.heap_n++] = b[k];
        return mk(T_LIST, l.a + r.a, kHeapBit | first);
    }
    return ERR;
}

}  // namespace rvm
}  // namespace pwaf
