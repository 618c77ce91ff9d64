// compile.cpp — rule set -> Program: static typing of each expression, lowering to ATOMS, a
// three-valued boolean DAG (true / false / execution-error), DNF per rule, DFA grouping, column layout.
//
// What this replaces in the reference, and the semantics it must reproduce:
//   - Rule::match_request (pingoo/rules.rs:37-51): the rule matches iff the expression evaluates to
//     Bool(true); an execution error or a non-Bool result means "no match". Because the variable
//     surface is fixed (pingoo/rules.rs:16-34) every expression can be typed statically, so
//     "execution error" becomes a compile-time third truth value carried through && || ! ?: with the
//     interpreter's left-to-right short-circuit order (decisions D5-D7).
//   - the rule loop (http_listener.rs:251-264): first matching rule whose action list produces an
//     effect decides. Per rule we precompute the effect for verified / unverified clients, so the
//     device resolves "first match wins" as a minimum over firing rule indices.
//   - gates A and B (http_listener.rs:196-204) become two pseudo rules in front of the user's.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <set>
#include <tuple>
#include <unordered_map>

#include "frontend.h"
#include "program.h"

namespace pwaf {

namespace {

// ---------------------------------------------------------------------------------------------------
// three-valued boolean DAG with hash-consing
// ---------------------------------------------------------------------------------------------------
enum NodeOp : uint8_t { G_FALSE, G_TRUE, G_ATOM, G_NOT, G_AND, G_OR };
struct GNode {
    NodeOp op;
    int a, b;
};
struct Dag {
    std::vector<GNode> n;
    std::map<std::tuple<int, int, int>, int> memo;
    Dag() {
        n.push_back({G_FALSE, 0, 0});
        n.push_back({G_TRUE, 0, 0});
    }
    int mk(NodeOp op, int a, int b) {
        auto key = std::make_tuple((int)op, a, b);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        n.push_back({op, a, b});
        memo.emplace(key, (int)n.size() - 1);
        return (int)n.size() - 1;
    }
    int atom(int id) { return mk(G_ATOM, id, 0); }
    int Not(int x) {
        if (x == 0) return 1;
        if (x == 1) return 0;
        if (n[x].op == G_NOT) return n[x].a;
        return mk(G_NOT, x, 0);
    }
    int And(int x, int y) {
        if (x == 0 || y == 0) return 0;
        if (x == 1) return y;
        if (y == 1) return x;
        if (x == y) return x;
        if (Not(x) == y) return 0;
        if (x > y) std::swap(x, y);
        return mk(G_AND, x, y);
    }
    int Or(int x, int y) {
        if (x == 1 || y == 1) return 1;
        if (x == 0) return y;
        if (y == 0) return x;
        if (x == y) return x;
        if (Not(x) == y) return 1;
        if (x > y) std::swap(x, y);
        return mk(G_OR, x, y);
    }
};

// a three-valued boolean: T = "evaluates to true", F = "evaluates to false"; neither = execution error
struct TF {
    int t, f;
};

// ---------------------------------------------------------------------------------------------------
// static values
// ---------------------------------------------------------------------------------------------------
struct CVal {
    enum K : uint8_t { Null, Bool, Int, Float, Str, List, Map } k = Null;
    bool b = false;
    int64_t i = 0;
    double f = 0;
    std::string s;
    std::vector<CVal> items;                          // List
    std::vector<std::pair<std::string, CVal>> pairs;  // Map
};

struct HostList {
    std::string name;
    uint32_t type;
    std::vector<std::string> strs;
    std::vector<int64_t> ints;
    std::vector<PrefixEntry> nets;
    int ip_list_index = -1;  // position among IP lists (bit in the membership set)
    size_t size() const { return type == PWAF_LIST_STRING ? strs.size() : type == PWAF_LIST_INT ? ints.size() : nets.size(); }
};

struct SVal {
    enum K : uint8_t {
        ERR,       // evaluating this always raises an execution error
        CONST,     // compile-time constant
        FIELD,     // http_request.<field>              (String)
        COUNTRY,   // client.country                    (String, 2 bytes)
        INTVAR,    // client.remote_port / client.asn   (Int)
        IPVAR,     // client.ip                         (Ip)
        LEN,       // <FIELD>.length()                  (Int)
        BOOLX,     // boolean over atoms
        LISTREF,   // lists["name"]
        NETCONST,  // an item of an Ip list (no literal syntax exists; only reachable through indexing)
        MAP_HTTP, MAP_CLIENT, MAP_LISTS, MAP_HEADERS
    } k = ERR;
    CVal c;
    int field = 0;  // FIELD/LEN: PWAF_FIELD_*; INTVAR: IntVar
    TF tf{0, 0};
    int list = -1;
    std::string emsg;
};

struct Unsupported {
    std::string msg;
};

static const char *const kFieldNames[5] = {"host", "url", "path", "method", "user_agent"};

class RuleCompiler {
public:
    RuleCompiler(Program &p, Dag &d, std::vector<HostList> &l) : prog(p), dag(d), lists(l) {}

    Program &prog;
    Dag &dag;
    std::vector<HostList> &lists;
    std::map<std::string, int> atom_index;
    std::map<std::vector<int64_t>, uint32_t> intset_index;
    std::map<std::string, uint32_t> lut_index;
    const Syntax *syn = nullptr;
    uint32_t max_dfa_states = 0, max_table_bytes = 0;

    // ---- atoms ----
    // Once the rule set's header names are CLOSED (compile_program collected them from every rule's literal keys before compiling any),
    // a name that is not among them is an ABSENT key of the headers map — reachable through a key that only constant folding makes a
    // literal (`http_request["head" + "ers"]["x"]`, `http_request.headers["x-" + "a"]`): the oracle's headers map does not hold it
    // (oracle_engine.cpp: the names come from literal keys in the SYNTAX), so neither does this one. -1 = absent.
    bool headers_closed = false;
    int header_lookup(const std::string &name) {
        if (!headers_closed) return header_field(name);
        for (size_t k = 0; k < prog.header_names.size(); k++)
            if (prog.header_names[k] == name) return PWAF_N_FIELDS + (int)k;
        return -1;
    }
    int header_field(const std::string &name) {
        for (size_t k = 0; k < prog.header_names.size(); k++)
            if (prog.header_names[k] == name) return PWAF_N_FIELDS + (int)k;
        if (prog.header_names.size() >= kMaxHeaders) throw Unsupported{"more than " + std::to_string(kMaxHeaders) + " distinct header names"};
        prog.header_names.push_back(name);
        return PWAF_N_FIELDS + (int)prog.header_names.size() - 1;
    }
    int intern_atom(Atom &&a) {
        auto it = atom_index.find(a.key);
        if (it != atom_index.end()) return it->second;
        int id = (int)prog.atoms.size();
        atom_index.emplace(a.key, id);
        prog.atoms.push_back(std::move(a));
        return id;
    }
    TF atom_tf(int id) {
        int n = dag.atom(id);
        return {n, dag.Not(n)};
    }
    static TF tf_const(bool v) { return v ? TF{1, 0} : TF{0, 1}; }
    static TF tf_err() { return TF{0, 0}; }
    TF tf_not(TF x) { return {x.f, x.t}; }

    // string predicate over FIELD f or COUNTRY, given as a pattern
    TF string_atom(const SVal &target, RNodeP rx) {
        if (target.k == SVal::FIELD) {
            Atom a;
            a.kind = ATOM_SCAN;
            a.field = (uint8_t)target.field;
            a.pattern = rx;
            a.min_len = rx_min_len(*rx);
            a.key = "S" + std::to_string(target.field) + ":" + rx_key(*rx);
            return atom_tf(intern_atom(std::move(a)));
        }
        // COUNTRY: the domain is the 676 two-letter codes (geoip.rs:128-142), so fold the predicate into a table
        std::vector<ScanPattern> one{{rx, 0}};
        DfaGroup g;
        std::string err;
        if (!build_dfa(one, 4096, 1u << 20, g, err)) throw Unsupported{"client.country predicate too complex: " + err};
        std::bitset<704> lut;
        std::vector<uint16_t> hit;
        for (int c0 = 0; c0 < 26; c0++)
            for (int c1 = 0; c1 < 26; c1++) {
                uint8_t s[2] = {(uint8_t)('A' + c0), (uint8_t)('A' + c1)};
                hit.clear();
                dfa_run_host(g, s, 2, hit);
                if (!hit.empty()) lut.set((size_t)c0 * 26 + c1);
            }
        if (lut.none()) return tf_const(false);
        if (lut.count() == 676) return tf_const(true);
        std::string k = lut.to_string();
        auto it = lut_index.find(k);
        uint32_t idx;
        if (it == lut_index.end()) {
            if (prog.country_luts.size() >= kDevMaxCountryLuts) throw Unsupported{"more than " + std::to_string(kDevMaxCountryLuts) + " distinct client.country predicates"};
            idx = (uint32_t)prog.country_luts.size();
            prog.country_luts.push_back(lut);
            lut_index.emplace(k, idx);
        } else idx = it->second;
        Atom a;
        a.kind = ATOM_COUNTRY;
        a.ref = idx;
        a.key = "C" + std::to_string(idx);
        return atom_tf(intern_atom(std::move(a)));
    }

    // one request string field against another (pingoo/rules.rs:37-51 evaluates any expression: `url.contains(host)`,
    // `host == headers["x-forwarded-host"]`, `path.length() < url.length()` are all legal): a device atom of its own kind
    // The device evaluates them in one pseudo pass that holds kMaxFcmpAtoms predicates over kMaxFcmpFields distinct fields (fcmp_kernel:
    // the field pointers travel as kernel arguments). A rule that would need one more is lowered to a residual program instead (the caller
    // catches Unsupported) — it used to fail the WHOLE creation, lenient or not, without naming a rule. (Counted over every atom made, used
    // or not: conservative.)
    std::set<std::string> fcmp_keys;
    std::vector<int> fcmp_fields;
    // (mirrors of the device tables' widths: kernels.h kMaxHeaderLens, engine.cpp's 128 asn comparisons / 256 country tables / 128 integer
    // sets per client variable — the class-row and membership-row words of attr_kernel)
    static constexpr uint32_t kDevMaxHeaderLens = 8, kDevMaxAsnCmp = 128, kDevMaxCountryLuts = 256, kDevMaxIntSets = 128;
    std::vector<int> hlen_fields_seen;
    uint32_t n_asn_cmp = 0, n_int_sets[2] = {0, 0};
    TF fcmp_atom(int a, int b, FcmpOp op) {
        Atom at;
        at.kind = ATOM_FCMP;
        at.field = (uint8_t)a;
        at.ref = (uint32_t)b;
        at.c = (int64_t)op;
        at.key = "F" + std::to_string(a) + ":" + std::to_string((int)op) + ":" + std::to_string(b);
        if (!fcmp_keys.count(at.key)) {
            auto known = [&](int f) { return std::find(fcmp_fields.begin(), fcmp_fields.end(), f) != fcmp_fields.end(); };
            const size_t nf = fcmp_fields.size() + (known(a) ? 0 : 1) + ((b != a && !known(b)) ? 1 : 0);
            if (fcmp_keys.size() >= kMaxFcmpAtoms || nf > kMaxFcmpFields)
                throw Unsupported{"more field-against-field predicates than the device's table holds (" + std::to_string(kMaxFcmpAtoms) + " predicates over " +
                                  std::to_string(kMaxFcmpFields) + " distinct fields)"};
            fcmp_keys.insert(at.key);
            for (int f : {a, b})
                if (std::find(fcmp_fields.begin(), fcmp_fields.end(), f) == fcmp_fields.end()) fcmp_fields.push_back(f);
        }
        return atom_tf(intern_atom(std::move(at)));
    }

    static RNodeP anchored(RNodeP body, bool start, bool end) {
        std::vector<RNodeP> k;
        if (start) k.push_back(rx_assert(A_TEXT_START));
        k.push_back(body);
        if (end) k.push_back(rx_assert(A_TEXT_END));
        return rx_cat(std::move(k));
    }
    static RNodeP any_of(const std::vector<std::string> &strs) {
        std::set<std::string> uniq(strs.begin(), strs.end());
        std::vector<RNodeP> alts;
        for (auto &s : uniq) alts.push_back(rx_literal(s));
        return rx_alt(std::move(alts));
    }
    TF string_in_set(const SVal &target, const std::vector<std::string> &strs) {
        if (strs.empty()) return tf_const(false);
        return string_atom(target, anchored(any_of(strs), true, true));
    }

    // integer comparisons: LEN(field) or INTVAR, normalised to EQ / LT / LE (+ negation)
    TF int_cmp_atom(const SVal &v, CmpOp op, int64_t c) {
        bool neg = false;
        if (op == OP_NE) { op = OP_EQ; neg = true; }
        if (op == OP_GT) { op = OP_LE; neg = true; }
        if (op == OP_GE) { op = OP_LT; neg = true; }
        // domain knowledge: lengths, ports and ASNs are never negative
        TF r;
        if ((op == OP_EQ && c < 0) || (op == OP_LT && c <= 0) || (op == OP_LE && c < 0)) r = tf_const(false);
        else {
            Atom a;
            a.kind = v.k == SVal::LEN ? ATOM_LEN : ATOM_INT;
            a.field = (uint8_t)v.field;
            a.op = op;
            a.c = c;
            a.key = std::string(v.k == SVal::LEN ? "L" : "I") + std::to_string(v.field) + "o" + std::to_string((int)op) + ":" + std::to_string(c);
            if (!atom_index.count(a.key)) {
                // Device-table limits that used to fail engine creation AS A WHOLE (engine.cpp checks them again): a rule that would
                // exceed one is lowered to a residual program instead — the caller catches Unsupported. Counted over every atom made.
                if (v.k == SVal::LEN && v.field >= PWAF_N_FIELDS) {
                    if (std::find(hlen_fields_seen.begin(), hlen_fields_seen.end(), v.field) == hlen_fields_seen.end()) {
                        if (hlen_fields_seen.size() >= kDevMaxHeaderLens) throw Unsupported{"length() of more than " + std::to_string(kDevMaxHeaderLens) + " distinct headers is compared"};
                        hlen_fields_seen.push_back(v.field);
                    }
                }
                if (v.k != SVal::LEN && v.field == VAR_ASN) {
                    if (n_asn_cmp >= kDevMaxAsnCmp) throw Unsupported{"more than " + std::to_string(kDevMaxAsnCmp) + " distinct client.asn comparisons"};
                    n_asn_cmp++;
                }
            }
            r = atom_tf(intern_atom(std::move(a)));
        }
        return neg ? tf_not(r) : r;
    }
    TF int_cmp_double(const SVal &v, CmpOp op, double c) {
        if (std::isnan(c)) {
            if (op == OP_EQ) return tf_const(false);
            if (op == OP_NE) return tf_const(true);
            return tf_err();  // "values are not comparable"
        }
        const double lim = 9.2e18;
        switch (op) {
            case OP_EQ: case OP_NE: {
                bool integral = std::floor(c) == c && std::fabs(c) < lim;
                TF r = integral ? int_cmp_atom(v, OP_EQ, (int64_t)c) : tf_const(false);
                return op == OP_NE ? tf_not(r) : r;
            }
            case OP_LT: case OP_GE: {
                double t = std::ceil(c);
                TF r = t >= lim ? tf_const(true) : t <= -lim ? tf_const(false) : int_cmp_atom(v, OP_LT, (int64_t)t);
                return op == OP_GE ? tf_not(r) : r;
            }
            default: {
                double t = std::floor(c);
                TF r = t >= lim ? tf_const(true) : t <= -lim ? tf_const(false) : int_cmp_atom(v, OP_LE, (int64_t)t);
                return op == OP_GT ? tf_not(r) : r;
            }
        }
    }
    TF int_in_set(const SVal &v, std::vector<int64_t> set) {
        std::sort(set.begin(), set.end());
        set.erase(std::unique(set.begin(), set.end()), set.end());
        while (!set.empty() && set.front() < 0) set.erase(set.begin());
        if (set.empty()) return tf_const(false);
        if (set.size() == 1) return int_cmp_atom(v, OP_EQ, set[0]);
        if (v.k == SVal::LEN) {
            // lengths: a handful of equality atoms
            if (set.size() > 16) throw Unsupported{"length() membership in a set of more than 16 values"};
            TF r = tf_const(false);
            for (int64_t c : set) r = tf_or(r, int_cmp_atom(v, OP_EQ, c));
            return r;
        }
        auto it = intset_index.find(set);
        uint32_t idx;
        if (it == intset_index.end()) {
            idx = (uint32_t)prog.int_sets.size();
            prog.int_sets.push_back(set);
            intset_index.emplace(set, idx);
        } else idx = it->second;
        Atom a;
        a.kind = ATOM_INTSET;
        a.field = (uint8_t)v.field;
        a.ref = idx;
        a.key = "J" + std::to_string(v.field) + ":" + std::to_string(idx);
        if (!atom_index.count(a.key)) {
            uint32_t &cnt = n_int_sets[v.field == VAR_ASN ? 1 : 0];
            if (cnt >= kDevMaxIntSets) throw Unsupported{"more than " + std::to_string(kDevMaxIntSets) + " integer-set predicates on one client variable"};
            cnt++;
        }
        return atom_tf(intern_atom(std::move(a)));
    }

    // ---- three-valued connectives (left-to-right short circuit) ----
    TF tf_or(TF x, TF y) { return {dag.Or(x.t, dag.And(x.f, y.t)), dag.And(x.f, y.f)}; }
    TF tf_and(TF x, TF y) { return {dag.And(x.t, y.t), dag.Or(x.f, dag.And(x.t, y.f))}; }
    TF tf_eq(TF x, TF y) {
        return {dag.Or(dag.And(x.t, y.t), dag.And(x.f, y.f)), dag.Or(dag.And(x.t, y.f), dag.And(x.f, y.t))};
    }
    int tf_defined(TF x) { return dag.Or(x.t, x.f); }

    // ---- SVal helpers ----
    static SVal sv_err(const std::string &m) { SVal v; v.k = SVal::ERR; v.emsg = m; return v; }
    static SVal sv_bool(bool b) { SVal v; v.k = SVal::CONST; v.c.k = CVal::Bool; v.c.b = b; return v; }
    static SVal sv_int(int64_t i) { SVal v; v.k = SVal::CONST; v.c.k = CVal::Int; v.c.i = i; return v; }
    static SVal sv_float(double f) { SVal v; v.k = SVal::CONST; v.c.k = CVal::Float; v.c.f = f; return v; }
    static SVal sv_str(const std::string &s) { SVal v; v.k = SVal::CONST; v.c.k = CVal::Str; v.c.s = s; return v; }
    static SVal sv_tf(TF t) {
        SVal v;
        if (t.t == 1 && t.f == 0) return sv_bool(true);
        if (t.t == 0 && t.f == 1) return sv_bool(false);
        if (t.t == 0 && t.f == 0) return sv_err("execution error");
        v.k = SVal::BOOLX;
        v.tf = t;
        return v;
    }
    static bool is_const(const SVal &v, CVal::K k) { return v.k == SVal::CONST && v.c.k == k; }
    static bool is_dyn_string(const SVal &v) { return v.k == SVal::FIELD || v.k == SVal::COUNTRY; }
    static bool is_dyn_int(const SVal &v) { return v.k == SVal::INTVAR || v.k == SVal::LEN; }
    static bool is_boolish(const SVal &v) { return v.k == SVal::BOOLX || is_const(v, CVal::Bool); }
    static bool is_numeric_const(const SVal &v) { return is_const(v, CVal::Int) || is_const(v, CVal::Float); }
    TF as_tf(const SVal &v) {
        if (v.k == SVal::BOOLX) return v.tf;
        if (is_const(v, CVal::Bool)) return tf_const(v.c.b);
        return tf_err();  // ERR, or a non-Bool where a Bool is required
    }

    // ---- constant folding (same rules as the dynamic semantics) ----
    static bool c_eq(const CVal &a, const CVal &b) {
        if (a.k == CVal::Int && b.k == CVal::Float) return (double)a.i == b.f;
        if (a.k == CVal::Float && b.k == CVal::Int) return a.f == (double)b.i;
        if (a.k != b.k) return false;
        switch (a.k) {
            case CVal::Null: return true;
            case CVal::Bool: return a.b == b.b;
            case CVal::Int: return a.i == b.i;
            case CVal::Float: return a.f == b.f;
            case CVal::Str: return a.s == b.s;
            case CVal::List:
                if (a.items.size() != b.items.size()) return false;
                for (size_t k = 0; k < a.items.size(); k++) if (!c_eq(a.items[k], b.items[k])) return false;
                return true;
            case CVal::Map: {
                if (a.pairs.size() != b.pairs.size()) return false;
                for (auto &p : a.pairs) {
                    bool found = false;
                    for (auto &q : b.pairs) if (p.first == q.first) { found = c_eq(p.second, q.second); break; }
                    if (!found) return false;
                }
                return true;
            }
        }
        return false;
    }
    static int c_cmp(const CVal &a, const CVal &b) {  // 2 = not comparable
        auto sg = [](auto x, auto y) { return x < y ? -1 : x > y ? 1 : 0; };
        if (a.k == CVal::Int && b.k == CVal::Int) return sg(a.i, b.i);
        bool an = a.k == CVal::Int || a.k == CVal::Float, bn = b.k == CVal::Int || b.k == CVal::Float;
        if (an && bn) {
            double x = a.k == CVal::Int ? (double)a.i : a.f, y = b.k == CVal::Int ? (double)b.i : b.f;
            if (std::isnan(x) || std::isnan(y)) return 2;
            return sg(x, y);
        }
        if (a.k == CVal::Str && b.k == CVal::Str) return sg(a.s.compare(b.s), 0);
        return 2;
    }
    static bool c_list_contains(const CVal &l, const CVal &x) {
        for (auto &it : l.items) if (c_eq(it, x)) return true;
        return false;
    }
    static void map_set(CVal &m, const std::string &k, const CVal &v) {
        for (auto &p : m.pairs) if (p.first == k) { p.second = v; return; }
        m.pairs.emplace_back(k, v);
    }

    SVal fold_arith(BinOp op, const CVal &a, const CVal &b) {
        if (a.k == CVal::Int && b.k == CVal::Int) {
            int64_t r;
            switch (op) {
                case B_ADD: if (__builtin_add_overflow(a.i, b.i, &r)) return sv_err("integer overflow"); return sv_int(r);
                case B_SUB: if (__builtin_sub_overflow(a.i, b.i, &r)) return sv_err("integer overflow"); return sv_int(r);
                case B_MUL: if (__builtin_mul_overflow(a.i, b.i, &r)) return sv_err("integer overflow"); return sv_int(r);
                case B_DIV:
                    if (b.i == 0) return sv_err("division by zero");
                    if (a.i == std::numeric_limits<int64_t>::min() && b.i == -1) return sv_err("integer overflow");
                    return sv_int(a.i / b.i);
                case B_MOD:
                    if (b.i == 0) return sv_err("modulo by zero");
                    if (a.i == std::numeric_limits<int64_t>::min() && b.i == -1) return sv_int(0);
                    return sv_int(a.i % b.i);
                default: break;
            }
        }
        bool an = a.k == CVal::Int || a.k == CVal::Float, bn = b.k == CVal::Int || b.k == CVal::Float;
        if (an && bn && (a.k == CVal::Float || b.k == CVal::Float)) {
            double x = a.k == CVal::Int ? (double)a.i : a.f, y = b.k == CVal::Int ? (double)b.i : b.f;
            switch (op) {
                case B_ADD: return sv_float(x + y);
                case B_SUB: return sv_float(x - y);
                case B_MUL: return sv_float(x * y);
                case B_DIV: return sv_float(x / y);
                default: return sv_err("unsupported operand types");
            }
        }
        if (op == B_ADD && a.k == CVal::Str && b.k == CVal::Str) return sv_str(a.s + b.s);
        if (op == B_ADD && a.k == CVal::List && b.k == CVal::List) {
            SVal v;
            v.k = SVal::CONST;
            v.c.k = CVal::List;
            v.c.items = a.items;
            v.c.items.insert(v.c.items.end(), b.items.begin(), b.items.end());
            return v;
        }
        return sv_err("unsupported operand types");
    }

    // ---- membership: x in <collection>  /  <collection>.contains(x) ----
    SVal membership(const SVal &coll, const SVal &x) {
        if (coll.k == SVal::LISTREF) {
            const HostList &hl = lists[(size_t)coll.list];
            if (x.k == SVal::CONST) {
                if (hl.type == PWAF_LIST_STRING && x.c.k == CVal::Str) return sv_bool(std::find(hl.strs.begin(), hl.strs.end(), x.c.s) != hl.strs.end());
                if (hl.type == PWAF_LIST_INT && (x.c.k == CVal::Int || x.c.k == CVal::Float)) {
                    for (int64_t v : hl.ints) if (x.c.k == CVal::Int ? v == x.c.i : (double)v == x.c.f) return sv_bool(true);
                    return sv_bool(false);
                }
                return sv_bool(false);  // no element can equal a value of another type
            }
            if (is_dyn_string(x)) return hl.type == PWAF_LIST_STRING ? sv_tf(string_in_set(x, hl.strs)) : sv_bool(false);
            if (is_dyn_int(x)) return hl.type == PWAF_LIST_INT ? sv_tf(int_in_set(x, hl.ints)) : sv_bool(false);
            if (x.k == SVal::IPVAR) {
                if (hl.type != PWAF_LIST_IP) return sv_bool(false);
                if (hl.nets.empty()) return sv_bool(false);
                Atom a;
                a.kind = ATOM_IPSET;
                a.ref = (uint32_t)hl.ip_list_index;
                a.key = "P" + std::to_string(hl.ip_list_index);
                return sv_tf(atom_tf(intern_atom(std::move(a))));
            }
            if (x.k == SVal::BOOLX) return sv_tf({0, tf_defined(x.tf)});
            throw Unsupported{"membership test of this value in a list"};
        }
        if (is_const(coll, CVal::List)) {
            if (x.k == SVal::CONST) return sv_bool(c_list_contains(coll.c, x.c));
            if (is_dyn_string(x)) {
                std::vector<std::string> strs;
                for (auto &it : coll.c.items) if (it.k == CVal::Str) strs.push_back(it.s);
                return sv_tf(string_in_set(x, strs));
            }
            if (is_dyn_int(x)) {
                std::vector<int64_t> ints;
                for (auto &it : coll.c.items) {
                    if (it.k == CVal::Int) ints.push_back(it.i);
                    else if (it.k == CVal::Float && std::floor(it.f) == it.f && std::fabs(it.f) < 9.2e18) ints.push_back((int64_t)it.f);
                }
                return sv_tf(int_in_set(x, ints));
            }
            if (x.k == SVal::IPVAR) return sv_bool(false);  // there is no Ip literal syntax: no element can equal it
            if (x.k == SVal::BOOLX) {
                bool has_t = false, has_f = false;
                for (auto &it : coll.c.items) if (it.k == CVal::Bool) (it.b ? has_t : has_f) = true;
                int t = dag.Or(has_t ? x.tf.t : 0, has_f ? x.tf.f : 0);
                return sv_tf({t, dag.And(tf_defined(x.tf), dag.Not(t))});
            }
            throw Unsupported{"membership test of this value in a list"};
        }
        if (is_const(coll, CVal::Map) || coll.k == SVal::MAP_HTTP || coll.k == SVal::MAP_CLIENT || coll.k == SVal::MAP_LISTS || coll.k == SVal::MAP_HEADERS) {
            if (!is_const(x, CVal::Str)) {
                if (x.k == SVal::CONST) return sv_err("map keys are Strings");
                if (is_dyn_string(x)) throw Unsupported{"map key lookup with a request field"};
                return sv_err("map keys are Strings");
            }
            if (coll.k == SVal::MAP_HTTP) { for (auto f : kFieldNames) if (x.c.s == f) return sv_bool(true); return sv_bool(x.c.s == "headers"); }
            // (the headers map holds exactly the names the rule set mentions with a literal key: a literal `"x" in ..` names x — the
            // pre-scan has seen it —, a key that only folding made constant is one of them or absent)
            if (coll.k == SVal::MAP_HEADERS) return sv_bool(header_lookup(x.c.s) >= 0);
            if (coll.k == SVal::MAP_CLIENT) return sv_bool(x.c.s == "ip" || x.c.s == "remote_port" || x.c.s == "asn" || x.c.s == "country");
            if (coll.k == SVal::MAP_LISTS) { for (auto &l : lists) if (l.name == x.c.s) return sv_bool(true); return sv_bool(false); }
            for (auto &p : coll.c.pairs) if (p.first == x.c.s) return sv_bool(true);
            return sv_bool(false);
        }
        return sv_err("membership test on a non-collection");
    }

    // ---- string functions ----
    enum StrFn { F_CONTAINS, F_STARTS, F_ENDS };
    SVal string_fn(StrFn fn, const SVal &recv, const SVal &arg) {
        bool rs = is_dyn_string(recv) || is_const(recv, CVal::Str), as = is_dyn_string(arg) || is_const(arg, CVal::Str);
        if (!rs || !as) {
            if (recv.k == SVal::ERR) return recv;
            if (arg.k == SVal::ERR) return arg;
            return sv_err("String operands required");
        }
        if (recv.k == SVal::CONST && arg.k == SVal::CONST) {
            const std::string &h = recv.c.s, &p = arg.c.s;
            if (fn == F_CONTAINS) return sv_bool(h.find(p) != std::string::npos);
            if (p.size() > h.size()) return sv_bool(false);
            if (fn == F_STARTS) return sv_bool(h.compare(0, p.size(), p) == 0);
            return sv_bool(h.compare(h.size() - p.size(), p.size(), p) == 0);
        }
        if (is_dyn_string(recv) && arg.k == SVal::CONST) {
            RNodeP lit = rx_literal(arg.c.s);
            if (arg.c.s.empty()) return sv_bool(true);
            return sv_tf(string_atom(recv, anchored(lit, fn == F_STARTS, fn == F_ENDS)));
        }
        if (recv.k == SVal::CONST && is_dyn_string(arg)) {
            // "literal".f(field): the field must be one of the literal's substrings / prefixes / suffixes
            const std::string &h = recv.c.s;
            if (h.size() > 48) throw Unsupported{"String literal longer than 48 bytes as the receiver of a field argument"};
            std::vector<std::string> cands{""};
            if (fn == F_CONTAINS) for (size_t a = 0; a < h.size(); a++) for (size_t l = 1; a + l <= h.size(); l++) cands.push_back(h.substr(a, l));
            if (fn == F_STARTS) for (size_t l = 1; l <= h.size(); l++) cands.push_back(h.substr(0, l));
            if (fn == F_ENDS) for (size_t l = 1; l <= h.size(); l++) cands.push_back(h.substr(h.size() - l));
            return sv_tf(string_in_set(arg, cands));
        }
        if (recv.k == SVal::FIELD && arg.k == SVal::FIELD) {
            if (recv.field == arg.field) return sv_bool(true);  // every string contains / starts with / ends with itself
            return sv_tf(fcmp_atom(recv.field, arg.field, fn == F_CONTAINS ? FC_CONTAINS : fn == F_STARTS ? FC_STARTS : FC_ENDS));
        }
        throw Unsupported{"string function between client.country and a request field"};
    }

    SVal regex_fn(const SVal &recv, const SVal &arg) {
        bool rs = is_dyn_string(recv) || is_const(recv, CVal::Str), as = is_dyn_string(arg) || is_const(arg, CVal::Str);
        if (!rs || !as) {
            if (recv.k == SVal::ERR) return recv;
            if (arg.k == SVal::ERR) return arg;
            return sv_err("matches: String operands required");
        }
        if (arg.k != SVal::CONST) throw Unsupported{"matches() with a pattern that is not a String literal"};
        int status;
        std::string err;
        RNodeP rx = regex_parse(arg.c.s, status, err);
        if (status == 1) return sv_err("matches: invalid regex: " + err);  // run-time error in the reference => never matches
        if (status == 2) throw Unsupported{err};
        if (recv.k == SVal::CONST) {
            std::vector<ScanPattern> one{{rx, 0}};
            DfaGroup g;
            std::string e2;
            if (!build_dfa(one, 65535, 1u << 24, g, e2)) throw Unsupported{"regex too complex: " + e2};
            std::vector<uint16_t> hit;
            dfa_run_host(g, (const uint8_t *)recv.c.s.data(), recv.c.s.size(), hit);
            return sv_bool(!hit.empty());
        }
        return sv_tf(string_atom(recv, rx));
    }

    // ---- the lowering proper ----
    SVal lower(int ni) {
        const Ex &e = syn->nodes[(size_t)ni];
        switch (e.kind) {
            case EX_INT: return sv_int(e.ival);
            case EX_FLOAT: return sv_float(e.fval);
            case EX_STR: return sv_str(e.text);
            case EX_BOOL: return sv_bool(e.bval);
            case EX_NULL: { SVal v; v.k = SVal::CONST; v.c.k = CVal::Null; return v; }
            case EX_IDENT: {
                SVal v;
                if (e.text == "http_request") { v.k = SVal::MAP_HTTP; return v; }
                if (e.text == "client") { v.k = SVal::MAP_CLIENT; return v; }
                if (e.text == "lists") { v.k = SVal::MAP_LISTS; return v; }
                return sv_err("undeclared reference to '" + e.text + "'");
            }
            case EX_MEMBER: {
                SVal o = lower(e.kids[0]);
                return select(o, e.text);
            }
            case EX_INDEX: {
                SVal o = lower(e.kids[0]);
                if (o.k == SVal::ERR) return o;
                SVal i = lower(e.kids[1]);
                if (i.k == SVal::ERR) return i;
                if (o.k == SVal::MAP_HTTP || o.k == SVal::MAP_CLIENT || o.k == SVal::MAP_LISTS || o.k == SVal::MAP_HEADERS || is_const(o, CVal::Map)) {
                    if (is_const(i, CVal::Str)) return select(o, i.c.s);
                    if (is_dyn_string(i)) throw Unsupported{"map index computed from a request field"};
                    return sv_err("map keys are Strings");
                }
                if (is_const(o, CVal::List)) {
                    if (is_const(i, CVal::Int)) {
                        if (i.c.i < 0 || (uint64_t)i.c.i >= o.c.items.size()) return sv_err("index out of range");
                        SVal v;
                        v.k = SVal::CONST;
                        v.c = o.c.items[(size_t)i.c.i];
                        return v;
                    }
                    if (is_dyn_int(i)) throw Unsupported{"list index computed from a request value"};
                    return sv_err("list index must be an Int");
                }
                if (o.k == SVal::LISTREF) {
                    const HostList &hl = lists[(size_t)o.list];
                    if (is_const(i, CVal::Int)) {
                        if (i.c.i < 0 || (uint64_t)i.c.i >= hl.size()) return sv_err("index out of range");
                        if (hl.type == PWAF_LIST_STRING) return sv_str(hl.strs[(size_t)i.c.i]);
                        if (hl.type == PWAF_LIST_INT) return sv_int(hl.ints[(size_t)i.c.i]);
                        SVal v;
                        v.k = SVal::NETCONST;
                        return v;
                    }
                    if (is_dyn_int(i)) throw Unsupported{"list index computed from a request value"};
                    return sv_err("list index must be an Int");
                }
                return sv_err("index on a non-indexable value");
            }
            case EX_GCALL: {
                // arguments are evaluated first by a call, but an unknown function is an error either way
                return sv_err("undeclared function '" + e.text + "'");
            }
            case EX_MCALL: return method(e);
            case EX_LIST: {
                SVal v;
                v.k = SVal::CONST;
                v.c.k = CVal::List;
                for (int k : e.kids) {
                    SVal it = lower(k);
                    if (it.k == SVal::ERR) return it;
                    if (it.k != SVal::CONST) throw Unsupported{"list literal with an element computed from the request"};
                    v.c.items.push_back(it.c);
                }
                return v;
            }
            case EX_MAP: {
                SVal v;
                v.k = SVal::CONST;
                v.c.k = CVal::Map;
                for (size_t k = 0; k + 1 < e.kids.size(); k += 2) {
                    SVal key = lower(e.kids[k]);
                    if (key.k == SVal::ERR) return key;
                    if (key.k != SVal::CONST) throw Unsupported{"map literal with a key computed from the request"};
                    if (key.c.k != CVal::Str) return sv_err("map keys are Strings");
                    SVal val = lower(e.kids[k + 1]);
                    if (val.k == SVal::ERR) return val;
                    if (val.k != SVal::CONST) throw Unsupported{"map literal with a value computed from the request"};
                    map_set(v.c, key.c.s, val.c);
                }
                return v;
            }
            case EX_NOT: {
                SVal x = lower(e.kids[0]);
                if (x.k == SVal::ERR) return x;
                if (!is_boolish(x)) return sv_err("'!' requires a Bool");
                return sv_tf(tf_not(as_tf(x)));
            }
            case EX_NEG: {
                SVal x = lower(e.kids[0]);
                if (x.k == SVal::ERR) return x;
                if (is_const(x, CVal::Int)) {
                    if (x.c.i == std::numeric_limits<int64_t>::min()) return sv_err("integer overflow");
                    return sv_int(-x.c.i);
                }
                if (is_const(x, CVal::Float)) return sv_float(-x.c.f);
                if (is_dyn_int(x)) throw Unsupported{"arithmetic on a request value"};
                return sv_err("'-' requires a number");
            }
            case EX_COND: {
                SVal c = lower(e.kids[0]);
                if (c.k == SVal::ERR) return c;
                if (!is_boolish(c)) return sv_err("conditional requires a Bool");
                if (c.k == SVal::CONST) return lower(e.kids[c.c.b ? 1 : 2]);
                SVal a = lower(e.kids[1]), b = lower(e.kids[2]);
                bool ab = is_boolish(a) || a.k == SVal::ERR, bb = is_boolish(b) || b.k == SVal::ERR;
                if (!ab || !bb) throw Unsupported{"conditional with non-Bool branches selected by the request"};
                TF ta = as_tf(a), tb = as_tf(b);
                TF r{dag.Or(dag.And(c.tf.t, ta.t), dag.And(c.tf.f, tb.t)), dag.Or(dag.And(c.tf.t, ta.f), dag.And(c.tf.f, tb.f))};
                return sv_tf(r);
            }
            case EX_BIN: return binary(e);
        }
        return sv_err("internal");
    }

    SVal select(const SVal &o, const std::string &key) {
        if (o.k == SVal::ERR) return o;
        SVal v;
        if (o.k == SVal::MAP_HTTP) {
            for (int f = 0; f < 5; f++)
                if (key == kFieldNames[f]) { v.k = SVal::FIELD; v.field = f; return v; }
            if (key == "headers") { v.k = SVal::MAP_HEADERS; return v; }
            return sv_err("no such key: " + key);
        }
        if (o.k == SVal::MAP_HEADERS) {
            // EXTENSION (no reference counterpart, pingoo/rules.rs:16-25): one more String field per header name
            v.k = SVal::FIELD;
            v.field = header_lookup(key);
            if (v.field < 0) return sv_err("no such key: " + key);
            return v;
        }
        if (o.k == SVal::MAP_CLIENT) {
            if (key == "ip") { v.k = SVal::IPVAR; return v; }
            if (key == "remote_port") { v.k = SVal::INTVAR; v.field = VAR_PORT; return v; }
            if (key == "asn") { v.k = SVal::INTVAR; v.field = VAR_ASN; return v; }
            if (key == "country") { v.k = SVal::COUNTRY; return v; }
            return sv_err("no such key: " + key);
        }
        if (o.k == SVal::MAP_LISTS) {
            for (size_t k = 0; k < lists.size(); k++)
                if (lists[k].name == key) { v.k = SVal::LISTREF; v.list = (int)k; return v; }
            return sv_err("no such key: " + key);
        }
        if (is_const(o, CVal::Map)) {
            for (auto &p : o.c.pairs)
                if (p.first == key) { v.k = SVal::CONST; v.c = p.second; return v; }
            return sv_err("no such key: " + key);
        }
        return sv_err("member access on a non-map value");
    }

    SVal method(const Ex &e) {
        SVal recv = lower(e.kids[0]);
        if (recv.k == SVal::ERR) return recv;
        std::vector<SVal> args;
        for (size_t k = 1; k < e.kids.size(); k++) {
            args.push_back(lower(e.kids[k]));
            if (args.back().k == SVal::ERR) return args.back();
        }
        const std::string &f = e.text;
        if (f == "contains") {
            if (args.size() != 1) return sv_err("contains: expected 1 argument");
            if (is_dyn_string(recv) || is_const(recv, CVal::Str)) {
                bool arg_str = is_dyn_string(args[0]) || is_const(args[0], CVal::Str);
                if (!arg_str) return sv_err("contains: argument must be a String");
                return string_fn(F_CONTAINS, recv, args[0]);
            }
            if (recv.k == SVal::LISTREF || is_const(recv, CVal::List)) return membership(recv, args[0]);
            if (is_const(recv, CVal::Map) || recv.k == SVal::MAP_HTTP || recv.k == SVal::MAP_CLIENT || recv.k == SVal::MAP_LISTS || recv.k == SVal::MAP_HEADERS) return membership(recv, args[0]);
            return sv_err("contains: unsupported receiver type");
        }
        if (f == "starts_with" || f == "ends_with") {
            if (args.size() != 1) return sv_err("expected 1 argument");
            return string_fn(f == "starts_with" ? F_STARTS : F_ENDS, recv, args[0]);
        }
        if (f == "length") {
            if (!args.empty()) return sv_err("length: expected no arguments");
            if (recv.k == SVal::FIELD) { SVal v; v.k = SVal::LEN; v.field = recv.field; return v; }
            if (recv.k == SVal::COUNTRY) return sv_int(2);
            if (is_const(recv, CVal::Str)) return sv_int((int64_t)recv.c.s.size());
            if (is_const(recv, CVal::List)) return sv_int((int64_t)recv.c.items.size());
            if (is_const(recv, CVal::Map)) return sv_int((int64_t)recv.c.pairs.size());
            if (recv.k == SVal::LISTREF) return sv_int((int64_t)lists[(size_t)recv.list].size());
            if (recv.k == SVal::MAP_HTTP) return sv_int(6);  // host, url, path, method, user_agent + the headers map (extension)
            if (recv.k == SVal::MAP_HEADERS) throw Unsupported{"length() of the headers map (it holds the names the whole rule set mentions)"};
            if (recv.k == SVal::MAP_CLIENT) return sv_int(4);
            if (recv.k == SVal::MAP_LISTS) { std::set<std::string> names; for (auto &l : lists) names.insert(l.name); return sv_int((int64_t)names.size()); }
            return sv_err("length: unsupported receiver type");
        }
        if (f == "matches") {
            if (args.size() != 1) return sv_err("matches: expected 1 argument");
            return regex_fn(recv, args[0]);
        }
        return sv_err("undeclared function '" + f + "'");
    }

    static CmpOp flip(CmpOp op) {
        switch (op) {
            case OP_LT: return OP_GT;
            case OP_LE: return OP_GE;
            case OP_GT: return OP_LT;
            case OP_GE: return OP_LE;
            default: return op;
        }
    }

    SVal equality(const SVal &l, const SVal &r, bool negate) {
        auto fin = [&](TF t) { return sv_tf(negate ? tf_not(t) : t); };
        if (l.k == SVal::CONST && r.k == SVal::CONST) return sv_bool(c_eq(l.c, r.c) != negate);
        // Bool (possibly erroring) vs Bool
        if (is_boolish(l) && is_boolish(r)) return fin(tf_eq(as_tf(l), as_tf(r)));
        if (l.k == SVal::BOOLX || r.k == SVal::BOOLX) {
            // Bool vs a value of another type: never equal, but the Bool side is still evaluated
            const SVal &bx = l.k == SVal::BOOLX ? l : r;
            const SVal &other = l.k == SVal::BOOLX ? r : l;
            if (other.k == SVal::CONST || is_dyn_string(other) || is_dyn_int(other) || other.k == SVal::IPVAR || other.k == SVal::LISTREF || other.k == SVal::NETCONST)
                return fin({0, tf_defined(bx.tf)});
        }
        // field vs constant
        const SVal *dyn = nullptr, *con = nullptr;
        if (r.k == SVal::CONST) { dyn = &l; con = &r; }
        else if (l.k == SVal::CONST) { dyn = &r; con = &l; }
        if (dyn && con) {
            if (is_dyn_string(*dyn)) {
                if (con->c.k != CVal::Str) return sv_bool(negate);  // cross-type equality is false (D4)
                return fin(string_in_set(*dyn, {con->c.s}));
            }
            if (is_dyn_int(*dyn)) {
                if (con->c.k == CVal::Int) return fin(int_cmp_atom(*dyn, OP_EQ, con->c.i));
                if (con->c.k == CVal::Float) return fin(int_cmp_double(*dyn, OP_EQ, con->c.f));
                return sv_bool(negate);
            }
            if (dyn->k == SVal::IPVAR || dyn->k == SVal::LISTREF || dyn->k == SVal::NETCONST || dyn->k == SVal::MAP_HTTP || dyn->k == SVal::MAP_CLIENT || dyn->k == SVal::MAP_LISTS || dyn->k == SVal::MAP_HEADERS) {
                if (dyn->k == SVal::LISTREF && con->c.k == CVal::List) throw Unsupported{"comparison of a configured list with a list literal"};
                if ((dyn->k == SVal::MAP_HTTP || dyn->k == SVal::MAP_CLIENT || dyn->k == SVal::MAP_LISTS || dyn->k == SVal::MAP_HEADERS) && con->c.k == CVal::Map) throw Unsupported{"comparison of a context map with a map literal"};
                return sv_bool(negate);
            }
        }
        // dynamic vs dynamic
        bool ls = is_dyn_string(l), rs = is_dyn_string(r), li = is_dyn_int(l), ri = is_dyn_int(r);
        if ((ls && (ri || r.k == SVal::IPVAR)) || (li && (rs || r.k == SVal::IPVAR)) || (l.k == SVal::IPVAR && (rs || ri))) return sv_bool(negate);  // types differ
        if (l.k == SVal::FIELD && r.k == SVal::FIELD) {
            if (l.field == r.field) return sv_bool(!negate);
            return fin(fcmp_atom(std::min(l.field, r.field), std::max(l.field, r.field), FC_EQ));
        }
        if (l.k == SVal::LEN && r.k == SVal::LEN) {
            if (l.field == r.field) return sv_bool(!negate);
            return fin(fcmp_atom(std::min(l.field, r.field), std::max(l.field, r.field), FC_LEN_EQ));
        }
        throw Unsupported{"comparison between two request values of these kinds"};
    }

    SVal ordering(const SVal &l, const SVal &r, CmpOp op) {
        if (l.k == SVal::CONST && r.k == SVal::CONST) {
            int c = c_cmp(l.c, r.c);
            if (c == 2) return sv_err("values are not comparable");
            switch (op) {
                case OP_LT: return sv_bool(c < 0);
                case OP_LE: return sv_bool(c <= 0);
                case OP_GT: return sv_bool(c > 0);
                default: return sv_bool(c >= 0);
            }
        }
        const SVal *dyn = &l, *con = &r;
        if (l.k == SVal::CONST) { dyn = &r; con = &l; op = flip(op); }
        if (con->k == SVal::CONST) {
            if (is_dyn_int(*dyn)) {
                if (con->c.k == CVal::Int) return sv_tf(int_cmp_atom(*dyn, op, con->c.i));
                if (con->c.k == CVal::Float) return sv_tf(int_cmp_double(*dyn, op, con->c.f));
                return sv_err("values are not comparable");
            }
            if (is_dyn_string(*dyn)) {
                if (con->c.k == CVal::Str) throw Unsupported{"lexicographic ordering of a request field"};
                return sv_err("values are not comparable");
            }
            return sv_err("values are not comparable");
        }
        bool ln = is_dyn_int(l), rn = is_dyn_int(r), ls = is_dyn_string(l), rs = is_dyn_string(r);
        if (l.k == SVal::LEN && r.k == SVal::LEN) {
            // a.length() <op> b.length(): LT / LE as atoms, GT / GE by swapping the operands
            if (l.field == r.field) return sv_bool(op == OP_LE || op == OP_GE);
            if (op == OP_LT) return sv_tf(fcmp_atom(l.field, r.field, FC_LEN_LT));
            if (op == OP_LE) return sv_tf(fcmp_atom(l.field, r.field, FC_LEN_LE));
            if (op == OP_GT) return sv_tf(fcmp_atom(r.field, l.field, FC_LEN_LT));
            return sv_tf(fcmp_atom(r.field, l.field, FC_LEN_LE));
        }
        if ((ln && rn) || (ls && rs)) throw Unsupported{"ordering between two request values of these kinds"};
        return sv_err("values are not comparable");
    }

    SVal binary(const Ex &e) {
        if (e.op == B_OR || e.op == B_AND) {
            SVal l = lower(e.kids[0]);
            if (l.k == SVal::ERR) return l;
            if (!is_boolish(l)) return sv_err("logical operator: Bool operands required");
            if (l.k == SVal::CONST) {
                if (e.op == B_OR && l.c.b) return l;
                if (e.op == B_AND && !l.c.b) return l;
                // the right operand decides; it must be a Bool
                SVal r = lower(e.kids[1]);
                if (r.k == SVal::ERR) return r;
                if (!is_boolish(r)) return sv_err("logical operator: Bool operands required");
                return r;
            }
            SVal r = lower(e.kids[1]);
            TF tr = as_tf(r);  // ERR / non-Bool => error value, reached only when the left side does not decide
            return sv_tf(e.op == B_OR ? tf_or(l.tf, tr) : tf_and(l.tf, tr));
        }
        SVal l = lower(e.kids[0]);
        if (l.k == SVal::ERR) return l;
        SVal r = lower(e.kids[1]);
        if (r.k == SVal::ERR) return r;
        switch (e.op) {
            case B_EQ: return equality(l, r, false);
            case B_NE: return equality(l, r, true);
            case B_LT: return ordering(l, r, OP_LT);
            case B_LE: return ordering(l, r, OP_LE);
            case B_GT: return ordering(l, r, OP_GT);
            case B_GE: return ordering(l, r, OP_GE);
            case B_IN: {
                if (r.k == SVal::LISTREF || is_const(r, CVal::List) || is_const(r, CVal::Map) || r.k == SVal::MAP_HTTP || r.k == SVal::MAP_CLIENT || r.k == SVal::MAP_LISTS || r.k == SVal::MAP_HEADERS)
                    return membership(r, l);
                return sv_err("in: right operand must be a List or Map");
            }
            default: break;
        }
        // arithmetic
        if (l.k == SVal::CONST && r.k == SVal::CONST) return fold_arith(e.op, l.c, r.c);
        bool ld = is_dyn_int(l) || is_dyn_string(l), rd = is_dyn_int(r) || is_dyn_string(r);
        if (ld || rd) {
            // would it be well-typed at run time? (Int op Int, String + String)
            bool ln = is_dyn_int(l) || is_numeric_const(l), rn = is_dyn_int(r) || is_numeric_const(r);
            bool lstr = is_dyn_string(l) || is_const(l, CVal::Str), rstr = is_dyn_string(r) || is_const(r, CVal::Str);
            if ((ln && rn) || (e.op == B_ADD && lstr && rstr)) throw Unsupported{"arithmetic / concatenation on a request value"};
        }
        return sv_err("unsupported operand types");
    }
};

// ---------------------------------------------------------------------------------------------------
// DNF
// ---------------------------------------------------------------------------------------------------
using Term = std::vector<uint32_t>;  // sorted literals: atom << 1 | neg
struct DnfConv {
    const Dag &dag;
    std::map<std::pair<int, bool>, std::vector<Term>> memo;
    static constexpr size_t kMaxTerms = 1024;

    static bool merge(const Term &a, const Term &b, Term &out) {
        out.clear();
        std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(out));
        for (size_t k = 1; k < out.size(); k++) if ((out[k] ^ out[k - 1]) == 1) return false;  // a & !a
        return true;
    }
    static void tidy(std::vector<Term> &v) {
        std::sort(v.begin(), v.end(), [](const Term &a, const Term &b) { return a.size() != b.size() ? a.size() < b.size() : a < b; });
        v.erase(std::unique(v.begin(), v.end()), v.end());
        // absorption: drop terms that contain a shorter term
        std::vector<Term> keep;
        for (auto &t : v) {
            bool absorbed = false;
            for (auto &s : keep)
                if (s.size() < t.size() && std::includes(t.begin(), t.end(), s.begin(), s.end())) { absorbed = true; break; }
            if (!absorbed) keep.push_back(t);
        }
        v.swap(keep);
    }
    const std::vector<Term> &conv(int node, bool neg) {
        auto key = std::make_pair(node, neg);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        const GNode &g = dag.n[(size_t)node];
        std::vector<Term> out;
        switch (g.op) {
            case G_FALSE: if (neg) out.push_back({}); break;
            case G_TRUE: if (!neg) out.push_back({}); break;
            case G_ATOM: out.push_back({(uint32_t)g.a << 1 | (neg ? 1u : 0u)}); break;
            case G_NOT: out = conv(g.a, !neg); break;
            case G_AND: case G_OR: {
                bool is_and = (g.op == G_AND) != neg;  // De Morgan
                const std::vector<Term> &x = conv(g.a, neg), &y = conv(g.b, neg);
                if (is_and) {
                    if (x.size() * y.size() > kMaxTerms * 4) throw Unsupported{"expression too complex (DNF larger than the device limit)"};
                    Term m;
                    for (auto &a : x) for (auto &b : y) if (merge(a, b, m)) out.push_back(m);
                } else {
                    out = x;
                    out.insert(out.end(), y.begin(), y.end());
                }
                tidy(out);
                if (out.size() > kMaxTerms) throw Unsupported{"expression too complex (DNF larger than the device limit)"};
                break;
            }
        }
        return memo.emplace(key, std::move(out)).first->second;
    }
};

static std::string trim_item(const char *s) {
    // str::trim (pingoo/lists.rs:90)
    std::string t = s ? s : "";
    size_t b = 0, e = t.size();
    auto ws = [](char c) { return c == ' ' || (c >= 9 && c <= 13); };
    while (b < e && ws(t[b])) b++;
    while (e > b && ws(t[e - 1])) e--;
    return t.substr(b, e - b);
}

static void set_err(pwaf_compile_error &err, int code, uint32_t rule, const std::string &msg) {
    err.code = code;
    err.rule_index = rule;
    snprintf(err.message, sizeof err.message, "%s", msg.c_str());
}

}  // namespace

int compile_program(const CompileInput &in, std::unique_ptr<Program> &out, pwaf_compile_error &err) {
    auto prog = std::make_unique<Program>();
    Program &P = *prog;
    P.flags = in.opts.flags;
    uint32_t lds_budget = in.opts.lds_table_budget ? in.opts.lds_table_budget : 128 * 1024;
    uint32_t max_states = in.opts.max_dfa_states ? std::min(in.opts.max_dfa_states, kMaxDfaStates) : kMaxDfaStates;
    uint32_t max_table_bytes = in.opts.max_table_bytes ? in.opts.max_table_bytes : 3u << 20;  // L2-resident: 4 MiB per XCD
    if (lds_budget < 1024 || lds_budget > 150 * 1024) {
        set_err(err, PWAF_E_INVALID_ARG, 0xFFFFFFFFu, "lds_table_budget must be within 1 KiB .. 150 KiB");
        return PWAF_E_INVALID_ARG;
    }
    P.lds_hot_budget = lds_budget;
    if (max_table_bytes < 1024) {
        set_err(err, PWAF_E_INVALID_ARG, 0xFFFFFFFFu, "max_table_bytes must be at least 1 KiB");
        return PWAF_E_INVALID_ARG;
    }

    // ---- lists (pingoo/lists.rs:62-113) ----
    std::vector<HostList> lists;
    int n_ip_lists = 0;
    for (size_t k = 0; k < in.n_lists; k++) {
        const pwaf_list_desc &d = in.lists[k];
        HostList hl;
        hl.name = d.name ? d.name : "";
        hl.type = d.type;
        if (d.type > PWAF_LIST_IP) {
            set_err(err, PWAF_E_INVALID_ARG, 0xFFFFFFFFu, "unknown list type for list " + hl.name);
            return PWAF_E_INVALID_ARG;
        }
        for (uint32_t i = 0; i < d.n_items; i++) {
            std::string item = trim_item(d.items[i]);
            if (d.type == PWAF_LIST_STRING) hl.strs.push_back(item);
            else if (d.type == PWAF_LIST_INT) {
                int64_t v;
                if (!parse_i64_text(item, v)) {
                    set_err(err, PWAF_E_LIST, 0xFFFFFFFFu, "error parsing list " + hl.name + " at line " + std::to_string(i + 1) + ": error parsing int");
                    return PWAF_E_LIST;
                }
                hl.ints.push_back(v);
            } else {
                PrefixEntry pe;
                std::string e;
                if (!parse_ipnet_text(item, pe, e)) {
                    set_err(err, PWAF_E_LIST, 0xFFFFFFFFu, "error parsing list " + hl.name + " at line " + std::to_string(i + 1) + ": error parsing IP network: " + e);
                    return PWAF_E_LIST;
                }
                hl.nets.push_back(pe);
            }
        }
        // HashMap insert: a later list with the same name replaces the earlier one (lists.rs:56)
        bool replaced = false;
        for (auto &old : lists)
            if (old.name == hl.name) { int keep = old.ip_list_index; old = std::move(hl); old.ip_list_index = keep; replaced = true; break; }
        if (!replaced) lists.push_back(std::move(hl));
    }
    for (auto &l : lists) {
        if (l.type == PWAF_LIST_IP) l.ip_list_index = n_ip_lists++;
        else l.ip_list_index = -1;
    }

    // ---- atoms[0] = TRUE ----
    {
        Atom t;
        t.kind = ATOM_TRUE;
        t.key = "T";
        P.atoms.push_back(t);
    }
    Dag dag;
    RuleCompiler rc(P, dag, lists);
    struct RuleOut {
        int t_root;
        uint32_t public_idx;
        uint8_t eff_unverified, eff_verified;
    };
    std::vector<RuleOut> routs;

    // ---- pseudo rules: gates A and B (http_listener.rs:196-204) ----
    if (!(P.flags & PWAF_OPT_NO_UA_GATE)) {
        SVal ua;
        ua.k = SVal::LEN;
        ua.field = PWAF_FIELD_USER_AGENT;
        TF empty = rc.int_cmp_atom(ua, OP_EQ, 0), big = rc.int_cmp_atom(ua, OP_GE, 256);
        routs.push_back({dag.Or(empty.t, big.t), PWAF_RULE_UA_GATE, PWAF_ACTION_BLOCK, PWAF_ACTION_BLOCK});
    }
    if (!(P.flags & PWAF_OPT_NO_CAPTCHA_BYPASS)) {
        SVal path;
        path.k = SVal::FIELD;
        path.field = PWAF_FIELD_PATH;
        TF t = rc.string_atom(path, RuleCompiler::anchored(rx_literal("/__pingoo/captcha"), true, false));
        routs.push_back({t.t, PWAF_RULE_CAPTCHA_ENDPOINT, PWAF_ACTION_BYPASS, PWAF_ACTION_BYPASS});
    }

    // ---- user rules ----
    P.n_user_rules = (uint32_t)in.n_rules;
    P.rule_status.assign(in.n_rules, {PWAF_OK, std::string()});
    const bool strict = !(P.flags & PWAF_OPT_LENIENT);
    auto unsupported_rule = [&](size_t k, const std::string &why) {
        P.rule_status[k] = {PWAF_E_UNSUPPORTED, why};
        P.warnings.push_back("rule #" + std::to_string(k) + " is NOT evaluated (it never matches): " + why);
    };
    // A rule the column compiler cannot take is lowered WHOLE to a stack program for the residual interpreter (residual.h): it
    // becomes ONE atom whose column residual_kernel fills per request. Only what that compiler refuses as well stays unsupported.
    ResidualBuilder residual;
    std::vector<ResidualList> rlists;
    for (auto &l : lists) {
        ResidualList rl;
        rl.name = l.name; rl.type = l.type; rl.strs = l.strs; rl.ints = l.ints; rl.nets = l.nets;
        rlists.push_back(std::move(rl));
    }
    std::vector<Syntax> syntaxes(in.n_rules);
    // EXTENSION: the headers map holds the names the WHOLE rule set mentions with a literal key. They are collected
    // before any rule is compiled — the oracle's rule and order (collect_header_names) — so that the set is CLOSED when a rule needs the
    // map as a value (a computed key into http_request / http_request.headers, length() of the headers map: residual.cpp). A syntax
    // error ends the collection: the loop below reports it (or an earlier rule's error) and creation fails.
    bool headers_closed = true;
    {
        std::vector<std::string> names;
        for (size_t k = 0; k < in.n_rules; k++) {
            if (!in.rules[k].expression) continue;
            Syntax syn;
            std::string perr;
            if (!parse_expression(in.rules[k].expression, syn, perr)) { headers_closed = false; break; }
            collect_header_names(syn, names);
        }
        if (names.size() > kMaxHeaders) headers_closed = false;  // (the rule that mentions one name too many is refused below, as before)
        else for (const std::string &nm : names) rc.header_field(nm);
        rc.headers_closed = headers_closed;
    }
    auto try_residual = [&](size_t k, const std::string &col_why, std::string &why) -> int {
        if (P.flags & PWAF_OPT_NO_RESIDUAL) { why = col_why; return -1; }
        std::string rwhy;
        const int idx = residual.compile_rule(syntaxes[k], rlists, [&](const std::string &name) { return rc.header_lookup(name); }, rwhy, headers_closed ? &P.header_names : nullptr);
        if (idx < 0) { why = col_why + "; and the residual interpreter cannot take it either: " + rwhy; return -1; }
        if (P.residual_rule.size() <= (size_t)idx) P.residual_rule.resize((size_t)idx + 1, 0xFFFFFFFFu);
        P.residual_rule[(size_t)idx] = (uint32_t)k;
        Atom a;
        a.kind = ATOM_RESIDUAL;
        a.ref = (uint32_t)idx;
        a.key = "R" + std::to_string(idx);
        P.warnings.push_back("rule #" + std::to_string(k) + " has no column form and is lowered to a residual program (compiled for the device when an engine is created, else run by the per-request residual interpreter): " + col_why);
        return rc.intern_atom(std::move(a));
    };
    for (size_t k = 0; k < in.n_rules; k++) {
        const pwaf_rule_desc &rd = in.rules[k];
        std::string rname = rd.name ? rd.name : ("#" + std::to_string(k));
        uint8_t eff_u = PWAF_ACTION_ALLOW, eff_v = PWAF_ACTION_ALLOW;
        for (uint32_t a = 0; a < rd.n_actions; a++) {
            uint8_t act = rd.actions[a];
            if (act != PWAF_RULE_ACTION_BLOCK && act != PWAF_RULE_ACTION_CAPTCHA) {
                set_err(err, PWAF_E_INVALID_ARG, (uint32_t)k, "rule " + rname + ": unknown action code " + std::to_string(act));
                return PWAF_E_INVALID_ARG;
            }
            // http_listener.rs:253-262: Block returns; Captcha returns only for unverified clients
            if (eff_u == PWAF_ACTION_ALLOW) eff_u = act == PWAF_RULE_ACTION_BLOCK ? PWAF_ACTION_BLOCK : PWAF_ACTION_CAPTCHA;
            if (eff_v == PWAF_ACTION_ALLOW && act == PWAF_RULE_ACTION_BLOCK) eff_v = PWAF_ACTION_BLOCK;
        }
        int t_root = 1;  // expression None => match all (pingoo/rules.rs:48-50)
        if (rd.expression) {
            Syntax &syn = syntaxes[k];
            std::string perr;
            if (!parse_expression(rd.expression, syn, perr)) {
                set_err(err, PWAF_E_SYNTAX, (uint32_t)k, "error parsing rules: Expression is not valid: " + perr + " (rule " + rname + ")");
                return PWAF_E_SYNTAX;
            }
            rc.syn = &syn;
            try {
                SVal v = rc.lower(syn.root);
                if (v.k == SVal::ERR) {
                    P.warnings.push_back("rule " + rname + ": expression always fails at run time (" + v.emsg + "): the rule can never match");
                    t_root = 0;
                } else if (v.k == SVal::BOOLX) {
                    t_root = v.tf.t;
                } else if (RuleCompiler::is_const(v, CVal::Bool)) {
                    t_root = v.c.b ? 1 : 0;
                } else {
                    // a non-Bool result never equals Bool(true) (pingoo/rules.rs:47)
                    P.warnings.push_back("rule " + rname + ": expression does not evaluate to a Bool: the rule can never match");
                    t_root = 0;
                }
            } catch (Unsupported &u) {
                std::string why;
                const int ra = try_residual(k, u.msg, why);
                if (ra >= 0) {
                    t_root = dag.atom(ra);
                } else {
                    if (strict) {
                        set_err(err, PWAF_E_UNSUPPORTED, (uint32_t)k, "rule " + rname + ": " + why);
                        return PWAF_E_UNSUPPORTED;
                    }
                    unsupported_rule(k, "rule " + rname + ": " + why);
                    t_root = 0;
                }
            }
        }
        if (eff_u == PWAF_ACTION_ALLOW && eff_v == PWAF_ACTION_ALLOW) continue;  // no action can ever take effect
        if (t_root == 0) continue;                                                 // can never match
        routs.push_back({t_root, (uint32_t)k, eff_u, eff_v});
    }

    // ---- DNF per rule (over source atom indices) ----
    DnfConv dnf{dag, {}};
    std::vector<std::vector<Term>> rule_terms;
    for (auto &r : routs) {
        try {
            rule_terms.push_back(dnf.conv(r.t_root, false));
        } catch (Unsupported &u) {
            uint32_t idx = r.public_idx < in.n_rules ? r.public_idx : 0xFFFFFFFFu;
            std::string why = u.msg;
            const int ra = idx != 0xFFFFFFFFu ? try_residual(idx, u.msg, why) : -1;
            if (ra >= 0) {
                rule_terms.push_back({Term{(uint32_t)ra << 1}});  // the rule = its residual atom
                continue;
            }
            if (strict || idx == 0xFFFFFFFFu) {
                set_err(err, PWAF_E_UNSUPPORTED, idx, "rule #" + std::to_string(r.public_idx) + ": " + why);
                return PWAF_E_UNSUPPORTED;
            }
            unsupported_rule(idx, why);
            rule_terms.push_back({});  // no term: never matches
        }
    }
    // atoms that survived simplification
    std::vector<uint8_t> used(P.atoms.size(), 0);
    used[0] = 1;
    for (auto &terms : rule_terms) for (auto &t : terms) for (uint32_t l : t) used[l >> 1] = 1;

    // ---- column layout: [0] TRUE, numeric atoms, then each DFA group's atoms (group-local id = column - atom_base) ----
    uint32_t col = 1;
    for (size_t a = 1; a < P.atoms.size(); a++) {
        Atom &at = P.atoms[a];
        if (!used[a] || at.kind == ATOM_SCAN || at.kind == ATOM_FCMP || at.kind == ATOM_RESIDUAL) continue;
        at.id = col++;
    }
    uint32_t n_numeric = col - 1;
    const uint32_t scan_base = col;
    uint32_t next_col = scan_base;
    // DFA groups per field: ideally ONE table per field (the field's bytes are then read once). The state budget is what
    // an L2-resident table allows (hot rows are cached in LDS by the kernel), not what fits LDS. Patterns with an
    // unbounded wide-class repetition in the middle (".*") multiply states with each other, so when the joint DFA
    // explodes they are isolated into small groups of their own.
    auto field_name = [&](int f) { return f < PWAF_N_FIELDS ? std::string(kFieldNames[f]) : "headers[\"" + P.header_names[(size_t)f - PWAF_N_FIELDS] + "\"]"; };
    std::vector<std::pair<uint32_t, std::string>> bad_atoms;  // patterns no DFA could be built for
    for (int f = 0; f < PWAF_N_FIELDS + (int)P.header_names.size(); f++) {
        std::vector<ScanPattern> pats;
        for (size_t a = 1; a < P.atoms.size(); a++)
            if (used[a] && P.atoms[a].kind == ATOM_SCAN && P.atoms[a].field == f) pats.push_back({P.atoms[a].pattern, (uint32_t)a});
        if (pats.empty()) continue;
        struct WorkItem { std::vector<ScanPattern> pats; std::vector<uint32_t> filter; };
        struct PendingGate { std::vector<ScanPattern> pats; std::vector<uint32_t> factors; };
        std::vector<WorkItem> work;
        work.push_back({pats, {}});
        while (!work.empty()) {
            std::vector<ScanPattern> cur = std::move(work.back().pats);
            std::vector<uint32_t> cur_filter = std::move(work.back().filter);
            work.pop_back();
            DfaGroup g;
            std::string derr;
            if (cur.size() <= kMaxLocalAtoms && build_dfa(cur, max_states, max_table_bytes, g, derr)) {
                g.field = (uint8_t)f;
                g.filter_atoms = cur_filter;
                g.confirm_off = (P.flags & PWAF_OPT_NO_CONFIRM) != 0;
                // The R tier (program.h: DfaGroup::rtier): the DFA of the atoms that are NOT plain literals, for the candidates whose
                // regex factor the confirm tier found. (Gated gap passes sit behind no bigram filter: no tiers.)
                if (cur_filter.empty() && !(P.flags & (PWAF_OPT_NO_PREFILTER | PWAF_OPT_NO_CONFIRM))) {
                    std::vector<ScanPattern> rp;
                    std::vector<uint16_t> rmap;
                    for (size_t k = 0; k < cur.size(); k++) {
                        std::string lit;
                        bool s0, e0;
                        if (confirm_literal(*cur[k].rx, lit, s0, e0)) continue;
                        rp.push_back(cur[k]);
                        rmap.push_back((uint16_t)k);
                    }
                    g.n_confirm_literals = (uint32_t)(cur.size() - rp.size());
                    if (!rp.empty() && rp.size() < cur.size()) {
                        auto r = std::make_shared<DfaGroup>();
                        std::string rerr;
                        if (build_dfa(rp, max_states, max_table_bytes, *r, rerr)) {
                            for (auto &x : r->emit_list) x = rmap[x];  // local ids of the FULL group
                            for (auto &x : r->end_list) x = rmap[x];
                            r->field = (uint8_t)f;
                            r->atoms = g.atoms;
                            r->n_local = g.n_local;
                            g.rtier = r;
                        }  // (cannot be larger than the full DFA; if it fails anyway the pass keeps the full table for its walks)
                    }
                }
                g.atom_base = next_col;
                for (size_t k = 0; k < cur.size(); k++) P.atoms[cur[k].atom].id = next_col + (uint32_t)k;
                next_col += g.n_local;
                if (g.rtier) g.rtier->atom_base = g.atom_base;
                P.groups.push_back(std::move(g));
                continue;
            }
            if (cur.size() == 1) {
                // one pattern alone exceeds the budget (two long counted gaps in a row, `a.{0,60}b.{0,60}c`, need the product
                // of both distances; a counted gap with a large MINIMUM, `a.{20,40}b`, needs every subset of the last 20
                // positions): the rules that use it are reported and dropped, the rest of the set is unaffected
                bad_atoms.emplace_back(cur[0].atom, std::string("a pattern on http_request.") + field_name(f) + " needs a DFA beyond the state/table budget: " + derr);
                continue;
            }
            std::vector<ScanPattern> gap, plain;
            for (auto &p : cur) (has_wide_gap(*p.rx) ? gap : plain).push_back(p);
            if (!gap.empty() && (!plain.empty() || gap.size() > 8)) {
                // Isolate the gap patterns in chunks of 8. Each chunk whose patterns all have a necessary prefix factor is
                // GATED: the factors join the plain patterns (as hidden atoms) and the chunk's pass only visits requests where
                // one of them fired. LIFO work list: push the plain rest first so that it is built last, with the factors.
                std::vector<PendingGate> gates;
                for (size_t k = 0; k < gap.size(); k += 8) {
                    std::vector<ScanPattern> chunk(gap.begin() + (long)k, gap.begin() + (long)std::min(gap.size(), k + 8));
                    std::vector<uint32_t> factors;
                    bool all = true;
                    for (auto &p : chunk) {
                        RNodeP x = gap_prefilter(p.rx);
                        if (!x) { all = false; break; }
                        Atom fa;
                        fa.kind = ATOM_SCAN;
                        fa.field = (uint8_t)f;
                        fa.pattern = x;
                        fa.min_len = rx_min_len(*x);
                        fa.key = "S" + std::to_string(f) + ":" + rx_key(*x);
                        uint32_t idx = (uint32_t)rc.intern_atom(std::move(fa));
                        if (idx >= used.size()) used.resize(idx + 1, 0);
                        if (!used[idx]) {
                            used[idx] = 1;
                            plain.push_back({P.atoms[idx].pattern, idx});  // hidden atom: scanned with the plain patterns
                        }
                        factors.push_back(idx);
                    }
                    gates.push_back({chunk, all ? factors : std::vector<uint32_t>()});
                }
                if (!plain.empty()) work.push_back({plain, {}});
                for (auto &g : gates) work.push_back({g.pats, g.factors});
            } else {
                size_t half = cur.size() / 2;
                work.push_back({std::vector<ScanPattern>(cur.begin() + (long)half, cur.end()), cur_filter});
                work.push_back({std::vector<ScanPattern>(cur.begin(), cur.begin() + (long)half), cur_filter});
            }
        }
    }
    if (!bad_atoms.empty()) {
        // every rule with such a pattern in some term: reported by index (strict: creation fails naming the first one)
        for (size_t k = 0; k < routs.size(); k++) {
            const std::string *why = nullptr;
            for (auto &t : rule_terms[k])
                for (uint32_t l : t)
                    for (auto &ba : bad_atoms)
                        if ((l >> 1) == ba.first) why = &ba.second;
            if (!why) continue;
            const uint32_t idx = routs[k].public_idx < in.n_rules ? routs[k].public_idx : 0xFFFFFFFFu;
            std::string why2 = *why;
            const int ra = idx != 0xFFFFFFFFu ? try_residual(idx, *why, why2) : -1;
            if (ra >= 0) {  // (the interpreter walks a DFA of the pattern alone, within a budget of its own)
                rule_terms[k] = {Term{(uint32_t)ra << 1}};
                if ((size_t)ra >= used.size()) used.resize((size_t)ra + 1, 0);
                used[(size_t)ra] = 1;
                continue;
            }
            if (strict || idx == 0xFFFFFFFFu) {
                set_err(err, PWAF_E_UNSUPPORTED, idx, why2);
                return PWAF_E_UNSUPPORTED;
            }
            unsupported_rule(idx, why2);
            rule_terms[k].clear();
        }
    }
    // gated groups run after every ungated one (their factors must have been scanned), and resolve factor columns
    std::stable_partition(P.groups.begin(), P.groups.end(), [](const DfaGroup &g) { return g.filter_atoms.empty(); });
    for (auto &g : P.groups)
        for (uint32_t fa : g.filter_atoms) {
            g.filter_cols.push_back(P.atoms[fa].id);
            P.atoms[fa].gates = true;
        }
    P.n_scan_cols = next_col - scan_base;
    // field-against-field atoms: one more (pseudo) pass after the DFA passes
    P.fcmp_base = next_col;
    {
        std::vector<uint8_t> fields;
        for (size_t a = 1; a < P.atoms.size(); a++) {
            Atom &at = P.atoms[a];
            if (!used[a] || at.kind != ATOM_FCMP) continue;
            for (uint8_t f : {at.field, (uint8_t)at.ref})
                if (std::find(fields.begin(), fields.end(), f) == fields.end()) fields.push_back(f);
            if (P.fcmp.size() >= kMaxFcmpAtoms || fields.size() > kMaxFcmpFields) {
                set_err(err, PWAF_E_UNSUPPORTED, 0xFFFFFFFFu, "more than 32 field-against-field predicates, or more than 8 distinct fields in them");
                return PWAF_E_UNSUPPORTED;
            }
            at.id = next_col++;
            P.fcmp.push_back({at.id, (uint8_t)at.c, at.field, (uint8_t)at.ref, 0});
        }
    }
    // residual rules: the last pseudo pass, column = residual_base + the rule's index among the residual rules (every one that was
    // lowered has a column, used or not: the kernel evaluates the blob's rules by index)
    P.residual_base = next_col;
    P.n_residual = (uint32_t)residual.n_rules();
    next_col += P.n_residual;
    for (size_t a = 1; a < P.atoms.size(); a++)
        if (P.atoms[a].kind == ATOM_RESIDUAL) P.atoms[a].id = P.residual_base + P.atoms[a].ref;
    if (P.n_residual) {
        P.residual_blob = residual.blob();
        P.residual_needs_geo = residual.needs_geo();
    }
    if (P.n_residual > kMaxLocalAtoms) {
        set_err(err, PWAF_E_UNSUPPORTED, 0xFFFFFFFFu, "too many rules for the residual interpreter");
        return PWAF_E_UNSUPPORTED;
    }
    P.n_cols = next_col;
    if (P.groups.size() > kMaxGroups) {
        set_err(err, PWAF_E_UNSUPPORTED, 0xFFFFFFFFu, "the rule set needs more than " + std::to_string(kMaxGroups) + " scan passes");
        return PWAF_E_UNSUPPORTED;
    }
    if (P.n_cols >= LIT_ATOM_MASK) {
        set_err(err, PWAF_E_UNSUPPORTED, 0xFFFFFFFFu, "too many distinct predicates");
        return PWAF_E_UNSUPPORTED;
    }

    // ---- bigram prefilters (filter.cpp): a pass whose patterns all have a literal factor only walks the filter's candidates ----
    for (auto &terms : rule_terms) for (auto &t : terms) for (uint32_t l : t) if (l & 1) P.atoms[l >> 1].neg_used = true;
    if (!(P.flags & PWAF_OPT_NO_PREFILTER))
        for (auto &g : P.groups) {
            if (P.flags & PWAF_OPT_FILTER_STRIDE2) {
                build_group_filter(P.atoms, g, nullptr, g.filter, 2, true);  // (with extended windows: Model::best_window)
                if (g.filter.enabled) continue;
            }
            build_group_filter(P.atoms, g, nullptr, g.filter, 1);
            if (!g.filter.enabled || (P.flags & PWAF_OPT_FILTER_STRIDE2)) continue;
            // Without a traffic sample the choice of the sampling stride rests on the built-in prior over URL / header text: stride 2
            // (half the lookups: the pass streams at HBM speed instead of LDS speed) when the model expects it to flag at most two
            // points more of the requests than stride 1 and under a tenth of them — the rule pwaf_engine_tune applies to measured rates.
            GroupFilter alt;
            build_group_filter(P.atoms, g, nullptr, alt, 2);
#ifdef PWAF_PROFILING
            if (getenv("PWAF_TUNE_DEBUG")) fprintf(stderr, "[create] field %d: model rate stride 1 %.4f, stride 2 %s %.4f\n", g.field, g.filter.est_candidate_rate, alt.enabled ? "built" : alt.note.c_str(), alt.est_candidate_rate);
#endif
            if (alt.enabled && alt.est_candidate_rate <= g.filter.est_candidate_rate + 0.02 && alt.est_candidate_rate <= 0.10) g.filter = alt;
        }
    // pass order: plain passes, then filtered ones, then the gated gap passes — the hit records of every list-driven pass are
    // then contiguous (one memset per batch)
    std::stable_partition(P.groups.begin(), P.groups.end(), [](const DfaGroup &g) { return g.filter_atoms.empty() && !g.filter.enabled; });
    std::stable_partition(P.groups.begin(), P.groups.end(), [](const DfaGroup &g) { return g.filter_atoms.empty(); });

    // ---- numeric atom descriptors ----
    for (size_t a = 1; a < P.atoms.size(); a++) {
        const Atom &at = P.atoms[a];
        if (!used[a] || at.kind == ATOM_SCAN || at.kind == ATOM_FCMP || at.kind == ATOM_RESIDUAL) continue;
        NumAtomDev d{};
        d.col = at.id;
        d.kind = at.kind;
        d.var = at.field;
        d.op = at.op;
        d.c = at.c;
        if (at.kind == ATOM_INTSET) {
            d.ref = (uint32_t)P.int_pool.size();
            const auto &s = P.int_sets[at.ref];
            P.int_pool.insert(P.int_pool.end(), s.begin(), s.end());
            d.ref2 = (uint32_t)P.int_pool.size();
        } else {
            d.ref = at.ref;
        }
        P.num_atoms.push_back(d);
    }
    for (auto &lut : P.country_luts)
        for (int w = 0; w < 22; w++) {
            uint32_t word = 0;
            for (int b = 0; b < 32; b++) if (lut[(size_t)w * 32 + b]) word |= 1u << b;
            P.country_lut_words.push_back(word);
        }

    // ---- rules -> literal lists over device columns ----
    for (size_t k = 0; k < routs.size(); k++) {
        DevRule dr{};
        dr.lit_off = (uint32_t)P.lits.size();
        dr.public_idx = routs[k].public_idx;
        dr.eff_unverified = routs[k].eff_unverified;
        dr.eff_verified = routs[k].eff_verified;
        for (auto &t : rule_terms[k]) {
            if (t.empty()) {
                P.lits.push_back(0u | LIT_TERM_END);  // the TRUE column
                continue;
            }
            for (size_t j = 0; j < t.size(); j++) {
                uint32_t lit = P.atoms[t[j] >> 1].id;
                if (t[j] & 1) lit |= LIT_NEG;
                if (j + 1 == t.size()) lit |= LIT_TERM_END;
                P.lits.push_back(lit);
            }
        }
        dr.lit_cnt = (uint32_t)P.lits.size() - dr.lit_off;
        if (dr.lit_cnt == 0) continue;  // constant false after simplification
        P.rules.push_back(dr);
    }

    // ---- ip lists -> membership-set trie ----
    P.n_ip_lists = (uint32_t)n_ip_lists;
    {
        std::vector<PrefixEntry> all;
        for (auto &l : lists)
            if (l.type == PWAF_LIST_IP)
                for (auto pe : l.nets) { pe.payload = (uint32_t)l.ip_list_index; all.push_back(pe); }
        build_ip_trie(all, 0, (uint32_t)n_ip_lists, P.ipset_trie, P.set_masks, P.set_words);
    }
    // ---- GeoIP -> LPM trie (pingoo/geoip.rs:73-91,111-142) ----
    P.geo_recs.push_back({0, (uint16_t)('X' | 'X' << 8), 0});
    if (in.geoip) {
        P.has_geo = true;
        std::vector<PrefixEntry> all;
        for (size_t k = 0; k < in.geoip->n_entries; k++) {
            const pwaf_geoip_entry &g = in.geoip->entries[k];
            if (g.prefix_len > (g.is_v6 ? 128 : 32)) {
                set_err(err, PWAF_E_INVALID_ARG, 0xFFFFFFFFu, "geoip: invalid prefix length");
                return PWAF_E_INVALID_ARG;
            }
            PrefixEntry pe;
            memcpy(pe.addr, g.addr, 16);
            pe.len = g.prefix_len;
            pe.v6 = g.is_v6 != 0;
            bool valid = g.country[0] >= 'A' && g.country[0] <= 'Z' && g.country[1] >= 'A' && g.country[1] <= 'Z';
            if (valid) {
                pe.payload = (uint32_t)P.geo_recs.size();
                P.geo_recs.push_back({g.asn, (uint16_t)(g.country[0] | g.country[1] << 8), 0});
            } else {
                pe.payload = 0;  // a record that fails to decode makes the lookup fall back to the default (http_listener.rs:148-153)
            }
            all.push_back(pe);
        }
        std::vector<uint32_t> unused_masks;
        uint32_t unused_words = 0;
        build_ip_trie(all, 1, 0, P.geo_trie, unused_masks, unused_words);
    }

    // ---- stats ----
    pwaf_stats &s = P.stats;
    s.n_rules = (uint32_t)P.rules.size();
    s.n_atoms = 0;
    for (size_t a = 0; a < P.atoms.size(); a++) if (used[a]) s.n_atoms++;
    s.n_numeric_atoms = n_numeric;
    s.n_scan_atoms = s.n_atoms - n_numeric - 1;
    s.n_dfa_groups = (uint32_t)P.groups.size();
    for (auto &g : P.groups) {
        s.n_dfa_states_total += g.n_states;
        s.max_dfa_states = std::max(s.max_dfa_states, g.n_states);
        s.dfa_table_bytes_total += g.n_states * g.n_classes * 2;
        if (g.filter.enabled) s.n_filtered_groups++;
        if (g.filter.enabled && g.filter.confirm.enabled) s.n_confirm_literals += g.n_confirm_literals;
        if (!g.filter_atoms.empty()) s.n_gated_groups++;
    }
    s.n_ip_lists = P.n_ip_lists;
    s.ipset_trie_nodes = P.ipset_trie.n_nodes();
    s.geo_trie_nodes = P.geo_trie.n_nodes();
    s.n_dnf_literals = (uint32_t)P.lits.size();
    s.n_warnings = (uint32_t)P.warnings.size();
    out = std::move(prog);
    return PWAF_OK;
}

// ---------------------------------------------------------------------------------------------------
// dump: sequence of sections  [tag:u32][count:u32][bytes:u64][payload, padded to 8]
// ---------------------------------------------------------------------------------------------------
namespace {
struct Writer {
    std::vector<uint8_t> buf;
    void raw(const void *p, size_t n) {
        const uint8_t *b = (const uint8_t *)p;
        buf.insert(buf.end(), b, b + n);
    }
    void section(const char tag[4], uint32_t count, const void *p, size_t n) {
        raw(tag, 4);
        raw(&count, 4);
        uint64_t len = n;
        raw(&len, 8);
        raw(p, n);
        while (buf.size() % 8) buf.push_back(0);
    }
};
}  // namespace

std::vector<uint8_t> dump_program(const Program &p) {
    Writer w;
    w.raw("PWAFPRG1", 8);
    uint32_t head[8] = {p.n_cols, p.n_scan_cols, (uint32_t)p.groups.size(), (uint32_t)p.rules.size(), p.n_ip_lists, p.set_words, p.has_geo ? 1u : 0u, p.flags};
    w.section("HEAD", 8, head, sizeof head);
    for (size_t gi = 0; gi < p.groups.size(); gi++) {
        const DfaGroup &g = p.groups[gi];
        uint32_t gh[8] = {g.field, g.n_states, g.n_classes, 0, 0, g.atom_base, g.n_local, (uint32_t)g.atoms.size()};
        w.section("GHDR", (uint32_t)gi, gh, sizeof gh);
        w.section("GCLS", (uint32_t)gi, g.classmap, 256);
        if (g.umap.on()) {  // scalar mode: the device image of the scalar-value -> class map ([ill class] in the section's count field's place: first byte)
            std::vector<uint8_t> img = scalar_map_image(g.umap);
            img.insert(img.begin(), 8, 0);
            img[0] = g.umap.ill_class;
            w.section("GUMP", (uint32_t)gi, img.data(), img.size());
        }
        w.section("GTRN", (uint32_t)gi, g.trans.data(), g.trans.size() * 2);
        w.section("GEMO", (uint32_t)gi, g.emit_off.data(), g.emit_off.size() * 4);
        w.section("GEML", (uint32_t)gi, g.emit_list.data(), g.emit_list.size() * 2);
        w.section("GENO", (uint32_t)gi, g.end_off.data(), g.end_off.size() * 4);
        w.section("GENL", (uint32_t)gi, g.end_list.data(), g.end_list.size() * 2);
        w.section("GFLT", (uint32_t)gi, g.filter_cols.data(), g.filter_cols.size() * 4);
        if (g.filter.enabled) {
            // bigram prefilter: [init, n_heads | hash multiplier << 16] + heads (20 bytes each), then the 4096-entry table
            std::vector<uint8_t> fh(8 + g.filter.heads.size() * sizeof(FilterHead));
            const uint32_t hdr[2] = {g.filter.init, (uint32_t)g.filter.heads.size() | (g.filter.stride << 8) | (g.filter.mul << 16)};
            memcpy(fh.data(), hdr, 8);
            if (!g.filter.heads.empty()) memcpy(fh.data() + 8, g.filter.heads.data(), g.filter.heads.size() * sizeof(FilterHead));
            w.section("GFHD", (uint32_t)gi, fh.data(), fh.size());
            w.section("GFTB", (uint32_t)gi, g.filter.table.data(), g.filter.table.size() * 4);
            // confirm tier: [enabled, has_walk, entries, literal atoms of the pass]; the R-tier DFA the confirmed candidates walk
            const uint32_t ch[4] = {g.filter.confirm.enabled ? 1u : 0u, g.filter.confirm.has_walk ? 1u : 0u, (uint32_t)g.filter.confirm.entries.size(), g.n_confirm_literals};
            w.section("GCNF", (uint32_t)gi, ch, sizeof ch);
        }
        if (g.rtier) {
            const DfaGroup &r = *g.rtier;
            const uint32_t rh[2] = {r.n_states, r.n_classes};
            w.section("RHDR", (uint32_t)gi, rh, sizeof rh);
            w.section("RCLS", (uint32_t)gi, r.classmap, 256);
            if (r.umap.on()) {
                std::vector<uint8_t> img = scalar_map_image(r.umap);
                img.insert(img.begin(), 8, 0);
                img[0] = r.umap.ill_class;
                w.section("RUMP", (uint32_t)gi, img.data(), img.size());
            }
            w.section("RTRN", (uint32_t)gi, r.trans.data(), r.trans.size() * 2);
            w.section("REMO", (uint32_t)gi, r.emit_off.data(), r.emit_off.size() * 4);
            w.section("REML", (uint32_t)gi, r.emit_list.data(), r.emit_list.size() * 2);
            w.section("RENO", (uint32_t)gi, r.end_off.data(), r.end_off.size() * 4);
            w.section("RENL", (uint32_t)gi, r.end_list.data(), r.end_list.size() * 2);
        }
    }
    {
        std::string names;
        for (auto &h : p.header_names) { names += h; names += '\0'; }
        w.section("HDRS", (uint32_t)p.header_names.size(), names.data(), names.size());
    }
    w.section("FCMP", (uint32_t)p.fcmp.size(), p.fcmp.data(), p.fcmp.size() * sizeof(FcmpAtom));
    w.section("NUMA", (uint32_t)p.num_atoms.size(), p.num_atoms.data(), p.num_atoms.size() * sizeof(NumAtomDev));
    w.section("INTP", (uint32_t)p.int_pool.size(), p.int_pool.data(), p.int_pool.size() * 8);
    w.section("CLUT", (uint32_t)p.country_luts.size(), p.country_lut_words.data(), p.country_lut_words.size() * 4);
    w.section("RULE", (uint32_t)p.rules.size(), p.rules.data(), p.rules.size() * sizeof(DevRule));
    w.section("LITS", (uint32_t)p.lits.size(), p.lits.data(), p.lits.size() * 4);
    w.section("SETM", (uint32_t)(p.set_words ? p.set_masks.size() / p.set_words : 0), p.set_masks.data(), p.set_masks.size() * 4);
    w.section("IR4 ", 0, p.ipset_trie.root4.data(), p.ipset_trie.root4.size() * 4);
    w.section("IR6 ", 0, p.ipset_trie.root6.data(), p.ipset_trie.root6.size() * 4);
    w.section("INOD", p.ipset_trie.n_nodes(), p.ipset_trie.nodes.data(), p.ipset_trie.nodes.size() * 4);
    w.section("GR4 ", 0, p.geo_trie.root4.data(), p.geo_trie.root4.size() * 4);
    w.section("GR6 ", 0, p.geo_trie.root6.data(), p.geo_trie.root6.size() * 4);
    w.section("GNOD", p.geo_trie.n_nodes(), p.geo_trie.nodes.data(), p.geo_trie.nodes.size() * 4);
    w.section("GREC", (uint32_t)p.geo_recs.size(), p.geo_recs.data(), p.geo_recs.size() * sizeof(GeoRec));
    if (p.n_residual) {
        const uint32_t rs[2] = {p.n_residual, p.residual_base};
        w.section("RSDL", 2, rs, sizeof rs);
        w.section("RVMB", p.n_residual, p.residual_blob.data(), p.residual_blob.size());  // the residual interpreter's program image (residual.h)
    }
    return w.buf;
}

}  // namespace pwaf
