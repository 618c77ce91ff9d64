// residual.cpp — compiles a whole rule expression into the stack program of residual.h.
//
// Used by compile.cpp for a rule the column compiler rejected (Unsupported): instead of dropping the rule, its syntax tree is lowered
// to bytecode that residual_kernel interprets per request with the interpreter's dynamic semantics (the reference evaluates every
// valid expression: pingoo/rules.rs:37-51). The lowering is a plain post-order walk: no constant folding beyond what the context
// objects need (http_request / client / lists / http_request.headers are compile-time objects: a member with a literal key becomes
// the variable's own instruction), jumps for the short circuit of && || ?:. What cannot be lowered is reported (the rule then
// keeps its PWAF_E_UNSUPPORTED status): see residual.h.
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

#include <functional>

#include "frontend.h"
#include "program.h"
#include "residual.h"

namespace pwaf {

namespace {
using namespace rvm;

struct Reject {
    std::string why;
};

// what is statically known about the value an expression leaves on the stack
struct Info {
    int ctx = 0;          // 1 http_request, 2 client, 3 lists, 4 http_request.headers: a compile-time object, nothing was emitted
    bool clist = false;   // may be a configured list (lists["x"]): only a receiver of contains / length / [] or the right side of `in`
    uint32_t seg = 1;     // most rope segments a String in (or being) this value can have
    uint32_t nest = 0;    // deepest list / map nesting
    uint32_t len = 0;     // most items if it is a list
    uint32_t inner = 0;   // most items of any list / map INSIDE this value (what an index or a member access can bring to the top)
};
}  // namespace

struct ResidualBuilder::Impl {
    std::vector<Ins> code;
    std::vector<uint32_t> entries;
    std::vector<Val> consts;
    std::string strpool;
    std::vector<ListDesc> lists;      // by configured-list id of this builder
    std::vector<uint32_t> lstr;
    std::vector<int64_t> lints;
    std::vector<NetItem> nets;
    std::vector<RegexDesc> regexes;
    std::vector<std::vector<uint8_t>> regex_tabs;  // trans | classmap | flags per regex, appended to the blob
    std::vector<RegexSetDesc> rxsets;
    std::vector<uint32_t> rxitems;
    std::map<std::string, uint32_t> list_ids, regex_ids;
    bool needs_geo = false;

    // per rule
    const Syntax *syn = nullptr;
    const std::vector<ResidualList> *host_lists = nullptr;
    std::function<int(const std::string &)> header_field;
    const std::vector<std::string> *closed_headers = nullptr;  // every header name of the rule set, in column order (null: not known)
    uint32_t depth = 0, max_depth = 0, heap = 0;
    bool reorder = getenv("PWAF_RESIDUAL_SOURCE_ORDER") == nullptr;  // (measurement switch: operands of && / || in source order)
    uint32_t heap_items = 0;  // the largest heap bound over the accepted rules (Header::heap_items)

    uint32_t str_const(const std::string &s) {
        const uint32_t off = (uint32_t)strpool.size();
        strpool += s;
        return off;
    }
    uint32_t add_const(const Val &v) {
        consts.push_back(v);
        if (consts.size() > 0xFFFF) throw Reject{"more than 65535 constants in the residual programs"};
        return (uint32_t)consts.size() - 1;
    }
    void emit(uint8_t op, uint8_t a = 0, uint32_t b = 0) {
        if (b > 0xFFFF) throw Reject{"residual program too large"};
        code.push_back(Ins{op, a, (uint16_t)b});
        if (code.size() > 0xFFF0) throw Reject{"residual program too large"};
    }
    void push(int n = 1) {
        depth += (uint32_t)n;
        max_depth = std::max(max_depth, depth);
        if (max_depth > kStack) throw Reject{"expression needs more than " + std::to_string(kStack) + " stack slots in the residual interpreter"};
    }
    void pop(int n = 1) { depth -= (uint32_t)n; }
    void use_heap(uint32_t n) {
        heap += n;
        if (heap > kHeap) throw Reject{"expression builds more than " + std::to_string(kHeap) + " list / map / concatenation items per request"};
    }
    void push_err() { emit(R_CONST, 0, add_const(mk(T_ERR))); push(); }
    void push_bool(bool b) { emit(R_CONST, 0, add_const(mk_bool(b))); push(); }
    void push_int(int64_t i) { emit(R_CONST, 0, add_const(mk_int(i))); push(); }
    void push_str(const std::string &s) { emit(R_CONST, 0, add_const(mk(T_STR, (uint32_t)s.size(), ((uint64_t)S_CONST << 48) | str_const(s)))); push(); }

    uint32_t list_id(size_t k) {
        const ResidualList &hl = (*host_lists)[k];
        auto it = list_ids.find(hl.name + "#" + std::to_string(k));
        if (it != list_ids.end()) return it->second;
        ListDesc d{};
        d.type = hl.type;
        if (hl.type == PWAF_LIST_STRING) {
            d.first = (uint32_t)(lstr.size() / 2);
            d.n = (uint32_t)hl.strs.size();
            for (auto &s : hl.strs) { lstr.push_back(str_const(s)); lstr.push_back((uint32_t)s.size()); }
        } else if (hl.type == PWAF_LIST_INT) {
            d.first = (uint32_t)lints.size();
            d.n = (uint32_t)hl.ints.size();
            lints.insert(lints.end(), hl.ints.begin(), hl.ints.end());
        } else {
            d.first = (uint32_t)nets.size();
            d.n = (uint32_t)hl.nets.size();
            for (auto &pe : hl.nets) {
                NetItem ni{};
                memcpy(ni.addr, pe.addr, 16);
                ni.prefix = pe.len;
                ni.v6 = pe.v6 ? 1 : 0;
                nets.push_back(ni);
            }
        }
        lists.push_back(d);
        if (lists.size() > 0xFFFF) throw Reject{"too many configured lists in residual programs"};
        list_ids.emplace(hl.name + "#" + std::to_string(k), (uint32_t)lists.size() - 1);
        return (uint32_t)lists.size() - 1;
    }

    uint32_t regex_id(const std::string &pattern) {
        auto it = regex_ids.find(pattern);
        if (it != regex_ids.end()) return it->second;
        int status;
        std::string err;
        RNodeP rx = regex_parse(pattern, status, err);
        uint32_t id;
        if (status == 1) {
            id = 0xFFFu;  // invalid pattern: an execution error in the reference
        } else {
            if (status == 2) throw Reject{err};
            std::vector<ScanPattern> one{{rx, 0}};
            DfaGroup g;
            std::string e2;
            if (!build_dfa(one, 8192, 2u << 20, g, e2)) throw Reject{"matches(): the pattern's DFA exceeds the residual interpreter's budget: " + e2};
            if (regexes.size() >= 0x7FF) throw Reject{"too many regex literals in residual programs"};
            const std::vector<uint8_t> uimg = scalar_map_image(g.umap);  // (empty: the table reads bytes)
            const size_t uat = (g.trans.size() * 2 + 256 + g.n_states + 3) & ~(size_t)3;
            std::vector<uint8_t> tab(uat + uimg.size());
            if (!uimg.empty()) memcpy(tab.data() + uat, uimg.data(), uimg.size());
            memcpy(tab.data(), g.trans.data(), g.trans.size() * 2);
            memcpy(tab.data() + g.trans.size() * 2, g.classmap, 256);
            uint8_t *fl = tab.data() + g.trans.size() * 2 + 256;
            for (uint32_t s = 0; s < g.n_states; s++)
                fl[s] = (uint8_t)((g.emit_off[s + 1] != g.emit_off[s] ? 1 : 0) | (g.end_off[s + 1] != g.end_off[s] ? 2 : 0));
            // bit 2: DEAD — no deciding or accepting state can be reached from here (an anchored pattern that already failed): the walk
            // stops with "no match" instead of reading the rest of the string
            std::vector<uint8_t> alive(g.n_states, 0);
            for (uint32_t s = 0; s < g.n_states; s++) alive[s] = fl[s] != 0;
            for (bool changed = true; changed;) {
                changed = false;
                for (uint32_t s = 0; s < g.n_states; s++) {
                    if (alive[s]) continue;
                    for (uint32_t c = 0; c < g.n_classes && !alive[s]; c++)
                        if (alive[g.trans[(size_t)s * g.n_classes + c]]) alive[s] = 1, changed = true;
                }
            }
            for (uint32_t s = 0; s < g.n_states; s++)
                if (!alive[s]) fl[s] |= 4u;
            // ... and both facts about the TARGET state ride in the table entry (states < 8192: bits 15 and 14 are free), so a step of the
            // walk is one table load (regex_match of residual.h)
            uint16_t *tr = reinterpret_cast<uint16_t *>(tab.data());
            for (size_t k = 0; k < g.trans.size(); k++) {
                const uint16_t to = g.trans[k];
                tr[k] = (uint16_t)(to | ((fl[to] & 1u) ? 0x8000u : 0u) | ((fl[to] & 4u) ? 0x4000u : 0u));
            }
            RegexDesc d{};
            d.n_classes = g.n_classes;
            d.trans = (uint32_t)g.trans.size() * 2;  // (sizes for now: resolved to blob offsets in blob())
            d.flags = g.n_states;
            d.umap = uimg.empty() ? 0u : (uint32_t)uat;  // (relative to the table: resolved in blob())
            d.ill_class = g.umap.ill_class;
            regexes.push_back(d);
            regex_tabs.push_back(std::move(tab));
            id = (uint32_t)regexes.size() - 1;
        }
        regex_ids.emplace(pattern, id);
        return id;
    }

    static Info merge(const Info &a, const Info &b) {
        Info r;
        r.clist = a.clist || b.clist;
        r.seg = std::max(a.seg, b.seg);
        r.nest = std::max(a.nest, b.nest);
        r.len = std::max(a.len, b.len);
        r.inner = std::max(a.inner, b.inner);
        return r;
    }
    static void no_ctx(const Info &i, const char *what) {
        if (i.ctx) throw Reject{std::string("a context map (http_request / client / lists / headers) used as a value: ") + what};
    }
    static void no_clist(const Info &i, const char *what) {
        if (i.clist) throw Reject{std::string("a configured list used as a plain value: ") + what};
    }

    // Member / index with a literal key on a context object
    Info select_ctx(int ctx, const std::string &key) {
        static const char *const kFields[5] = {"host", "url", "path", "method", "user_agent"};
        Info r;
        if (ctx == 1) {
            for (int f = 0; f < 5; f++)
                if (key == kFields[f]) { emit(R_FIELD, 0, (uint32_t)f); push(); return r; }
            if (key == "headers") { r.ctx = 4; return r; }
            push_err();
            return r;
        }
        if (ctx == 4) {  // EXTENSION: one String column per header name the rule set mentions
            const int f = header_field(key);
            if (f < 0) { push_err(); return r; }  // (the names are closed and this is none of them: an absent key)
            emit(R_FIELD, 0, (uint32_t)f);
            push();
            return r;
        }
        if (ctx == 2) {
            if (key == "ip") { emit(R_IP); push(); return r; }
            if (key == "remote_port") { emit(R_PORT); push(); return r; }
            if (key == "asn") { needs_geo = true; emit(R_ASN); push(); return r; }
            if (key == "country") { needs_geo = true; emit(R_COUNTRY); push(); return r; }
            push_err();
            return r;
        }
        // lists
        for (size_t k = 0; k < host_lists->size(); k++)
            if ((*host_lists)[k].name == key) { emit(R_CLIST, 0, list_id(k)); push(); r.clist = true; r.len = (uint32_t)(*host_lists)[k].size(); return r; }
        push_err();
        return r;
    }
    bool ctx_has(int ctx, const std::string &key) {
        static const char *const kFields[5] = {"host", "url", "path", "method", "user_agent"};
        if (ctx == 1) { for (auto f : kFields) if (key == f) return true; return key == "headers"; }
        if (ctx == 4) return header_field(key) >= 0;  // (the headers map holds exactly the names the rule set mentions with a literal key: open set — asking makes it one; closed — it is one or not)
        if (ctx == 2) return key == "ip" || key == "remote_port" || key == "asn" || key == "country";
        for (auto &l : *host_lists) if (l.name == key) return true;
        return false;
    }

    // A context map as a real Map VALUE, built per request: what a COMPUTED key needs (`client[http_request.method]`,
    // `http_request.host in lists`, `http_request[k]`, `http_request.headers[k]`): the key sets of http_request / client / lists are
    // closed, and so is the headers map's once every rule has been read (closed_headers: the names the WHOLE rule set mentions with a
    // literal key) — the map is a literal of this compiler's own making and the generic Map operations answer: a key that is not a
    // String is an execution error, an absent one an error for [] and false for `in` / contains, exactly as for any map. keys_only: the
    // values are never read (membership tests). A map's entries sit on the stack until R_MKMAP: a rule set that mentions more header
    // names than the stack holds is refused here (stack slots), as is one whose names are not known yet (closed_headers == null).
    void gen_ctx_map(int ctx, bool keys_only, Info &r) {
        const Val v = ctx_map_const(ctx, keys_only, r);
        emit(R_CONST, 0, add_const(v));
        push();
    }
    // (round 6) The map is a CONSTANT of the program: its keys are string constants, its value slots name the request value they stand for
    // (T_REF: read through deref when an index / member / comparison reaches the slot) — one stack slot and no heap item whatever its
    // size. Until then the entries were pushed and popped by R_MKMAP: a rule set that mentions more than ~10 header names had every rule
    // that needs the headers map as a value refused ("stack slots").
    Val ctx_map_const(int ctx, bool keys_only, Info &r) {
        static const char *const kFields[5] = {"host", "url", "path", "method", "user_agent"};
        std::vector<Val> kv;
        auto key = [&](const std::string &k) { kv.push_back(mk(T_STR, (uint32_t)k.size(), ((uint64_t)S_CONST << 48) | str_const(k))); };
        auto val = [&](const Val &v) { kv.push_back(keys_only ? mk_bool(true) : v); };
        Info inner;  // (a value that is itself a map: http_request's "headers" entry)
        if (ctx == 1) {
            Val hm = mk(T_NULL);
            if (!keys_only) hm = ctx_map_const(4, false, inner);  // (its entries first: a map's entries are contiguous constants)
            for (int f = 0; f < 5; f++) { key(kFields[f]); val(mk(T_REF, R_FIELD, (uint64_t)f)); }
            key("headers");
            val(hm);
        } else if (ctx == 2) {
            if (!keys_only) needs_geo = true;
            key("ip"); val(mk(T_REF, R_IP));
            key("remote_port"); val(mk(T_REF, R_PORT));
            key("asn"); val(mk(T_REF, R_ASN));
            key("country"); val(mk(T_REF, R_COUNTRY));
        } else if (ctx == 3) {
            std::set<std::string> seen;  // (names are distinct in the set the host hands over; like select_ctx, the first of a name counts)
            for (size_t k = 0; k < host_lists->size(); k++) {
                const std::string &name = (*host_lists)[k].name;
                if (!seen.insert(name).second) continue;
                key(name);
                if (keys_only) val(mk_bool(true));
                else { val(mk(T_CLIST, list_id(k))); r.clist = true; r.len = std::max(r.len, (uint32_t)(*host_lists)[k].size()); }
            }
        } else {
            if (closed_headers == nullptr) throw Reject{"the headers map with a computed key (it holds the names the whole rule set mentions)"};
            for (const std::string &name : *closed_headers) {
                key(name);
                val(mk(T_REF, R_FIELD, (uint64_t)header_field(name)));
            }
        }
        const uint32_t n = (uint32_t)(kv.size() / 2);
        r.nest = std::max(r.nest, inner.nest + 1);
        if (r.nest > kMaxNest) throw Reject{"lists / maps nested deeper than " + std::to_string(kMaxNest)};
        r.inner = std::max(std::max(r.inner, inner.inner), inner.len);
        if (ctx != 3) r.len = n;
        const uint32_t base = (uint32_t)consts.size();
        for (const Val &v : kv) add_const(v);
        return mk(T_MAP, n, (uint64_t)base);
    }

    // does the node denote a context object? (decided from the syntax alone: nothing is emitted)
    int ctx_of(int ni) const {
        const Ex &e = syn->nodes[(size_t)ni];
        if (e.kind == EX_IDENT) return e.text == "http_request" ? 1 : e.text == "client" ? 2 : e.text == "lists" ? 3 : 0;
        if (e.kind == EX_MEMBER && e.text == "headers" && ctx_of(e.kids[0]) == 1) return 4;
        if (e.kind == EX_INDEX && ctx_of(e.kids[0]) == 1) {
            const Ex &ix = syn->nodes[(size_t)e.kids[1]];
            if (ix.kind == EX_STR && ix.text == "headers") return 4;
        }
        return 0;
    }

    // A list / map literal whose items are all constants is a CONSTANT of the program (no stack slot per item, no heap item: a literal of
    // 40 strings is a value like any other; round 6 — the items used to be pushed and popped like computed ones, 24 at most).
    bool is_const_node(int ni) const {
        const Ex &e = syn->nodes[(size_t)ni];
        switch (e.kind) {
            case EX_INT: case EX_FLOAT: case EX_STR: case EX_BOOL: case EX_NULL: return true;
            case EX_LIST: case EX_MAP:
                for (int k : e.kids) if (!is_const_node(k)) return false;
                return true;
            default: return false;
        }
    }
    Val const_value(int ni, Info &r) {
        const Ex &e = syn->nodes[(size_t)ni];
        r = Info{};
        switch (e.kind) {
            case EX_INT: return mk_int(e.ival);
            case EX_FLOAT: return mk_flt(e.fval);
            case EX_STR: return mk(T_STR, (uint32_t)e.text.size(), ((uint64_t)S_CONST << 48) | str_const(e.text));
            case EX_BOOL: return mk_bool(e.bval);
            case EX_NULL: return mk(T_NULL);
            default: break;
        }
        std::vector<Val> items;
        bool err = false;
        if (e.kind == EX_LIST) {
            for (int k : e.kids) {
                Info it;
                const Val v = const_value(k, it);
                err = err || v.t == T_ERR;
                r = merge(r, it);
                items.push_back(v);
            }
        } else {
            for (size_t k = 0; k + 1 < e.kids.size(); k += 2) {
                Info ki, vi;
                const Val key = const_value(e.kids[k], ki), val = const_value(e.kids[k + 1], vi);
                err = err || key.t != T_STR || val.t == T_ERR;  // (a key that is not a String: the map is an execution error, op_mkmap)
                r = merge(r, merge(ki, vi));
                if (err) continue;
                const std::string &kt = syn->nodes[(size_t)e.kids[k]].text;
                bool dup = false;
                for (size_t j = 0; j + 1 < items.size() && !dup; j += 2)
                    if (strpool.compare((size_t)(items[j].p & 0xFFFFFFFFu), items[j].a, kt) == 0) { items[j + 1] = val; dup = true; }  // later duplicates win
                if (!dup) { items.push_back(key); items.push_back(val); }
            }
        }
        r.nest += 1;
        if (r.nest > kMaxNest) throw Reject{"lists / maps nested deeper than " + std::to_string(kMaxNest)};
        r.inner = std::max(r.inner, r.len);
        r.len = (uint32_t)(e.kind == EX_LIST ? items.size() : items.size() / 2);
        r.clist = false;
        if (err) return mk(T_ERR);
        const uint32_t base = (uint32_t)consts.size();
        for (const Val &v : items) add_const(v);
        return mk(e.kind == EX_LIST ? T_LIST : T_MAP, r.len, (uint64_t)base);
    }

    Info gen(int ni) {
        const Ex &e = syn->nodes[(size_t)ni];
        Info r;
        switch (e.kind) {
            case EX_INT: push_int(e.ival); return r;
            case EX_FLOAT: emit(R_CONST, 0, add_const(mk_flt(e.fval))); push(); return r;
            case EX_STR: push_str(e.text); return r;
            case EX_BOOL: push_bool(e.bval); return r;
            case EX_NULL: emit(R_CONST, 0, add_const(mk(T_NULL))); push(); return r;
            case EX_IDENT:
                if (e.text == "http_request") { r.ctx = 1; return r; }
                if (e.text == "client") { r.ctx = 2; return r; }
                if (e.text == "lists") { r.ctx = 3; return r; }
                push_err();  // undeclared reference
                return r;
            case EX_MEMBER: {
                Info o = gen(e.kids[0]);
                if (o.ctx) return select_ctx(o.ctx, e.text);
                no_clist(o, "member access");
                emit(R_SELECT, 0, add_const(mk(T_STR, (uint32_t)e.text.size(), ((uint64_t)S_CONST << 48) | str_const(e.text))));
                r = o;
                r.len = std::max(o.len, o.inner);  // (a member of a map may be any list nested in it: ADVICE r3 — `{"k": [..]}.k + ..` was charged 0 items)
                return r;
            }
            case EX_INDEX: {
                Info o = gen(e.kids[0]);
                if (o.ctx) {
                    const Ex &ix = syn->nodes[(size_t)e.kids[1]];
                    if (ix.kind == EX_STR) return select_ctx(o.ctx, ix.text);
                    if (ix.kind == EX_INT || ix.kind == EX_FLOAT || ix.kind == EX_BOOL || ix.kind == EX_NULL) { push_err(); return r; }  // map keys are Strings
                    // a computed key: the map as a value, then the generic index (client / lists: closed key sets)
                    Info mr;
                    gen_ctx_map(o.ctx, false, mr);
                    Info ki = gen(e.kids[1]);
                    no_ctx(ki, "as an index");
                    no_clist(ki, "as an index");
                    emit(R_INDEX);
                    pop();
                    r = mr;
                    return r;
                }
                Info i = gen(e.kids[1]);
                no_ctx(i, "as an index");
                no_clist(i, "as an index");
                emit(R_INDEX);
                pop();
                r = o;
                r.clist = false;
                r.len = std::max(o.len, o.inner);  // (an item may be any list nested in the receiver, which can be LONGER than the receiver: `[[1, .., 20]][0]`)
                return r;
            }
            case EX_GCALL: push_err(); return r;  // undeclared function
            case EX_MCALL: return call(e);
            case EX_LIST: {
                if (!e.kids.empty() && is_const_node(ni)) { emit(R_CONST, 0, add_const(const_value(ni, r))); push(); return r; }
                uint32_t n = 0;
                for (int k : e.kids) {
                    Info it = gen(k);
                    no_ctx(it, "as a list element");
                    no_clist(it, "as a list element");
                    r = merge(r, it);
                    n++;
                }
                r.nest += 1;
                if (r.nest > kMaxNest) throw Reject{"lists / maps nested deeper than " + std::to_string(kMaxNest)};
                r.inner = std::max(r.inner, r.len);  // (merge() has left the longest item in r.len)
                r.len = n;
                r.clist = false;
                use_heap(n);
                emit(R_MKLIST, 0, n);
                pop((int)n);
                push();
                return r;
            }
            case EX_MAP: {
                if (!e.kids.empty() && is_const_node(ni)) { emit(R_CONST, 0, add_const(const_value(ni, r))); push(); return r; }
                uint32_t n = 0;
                for (size_t k = 0; k + 1 < e.kids.size(); k += 2) {
                    Info key = gen(e.kids[k]);
                    no_ctx(key, "as a map key");
                    no_clist(key, "as a map key");
                    Info val = gen(e.kids[k + 1]);
                    no_ctx(val, "as a map value");
                    no_clist(val, "as a map value");
                    r = merge(r, merge(key, val));
                    n++;
                }
                r.nest += 1;
                if (r.nest > kMaxNest) throw Reject{"lists / maps nested deeper than " + std::to_string(kMaxNest)};
                r.inner = std::max(r.inner, r.len);
                r.len = n;
                use_heap(2 * n);
                emit(R_MKMAP, 0, n);
                pop((int)(2 * n));
                push();
                return r;
            }
            case EX_NOT: {
                Info x = gen(e.kids[0]);
                if (x.ctx) { push_err(); return r; }  // '!' requires a Bool
                emit(R_NOT);
                return r;
            }
            case EX_NEG: {
                Info x = gen(e.kids[0]);
                if (x.ctx) { push_err(); return r; }
                emit(R_NEG);
                return r;
            }
            case EX_COND: {
                Info c = gen(e.kids[0]);
                if (c.ctx) { push_err(); return r; }  // conditional requires a Bool
                const size_t jc = code.size();
                emit(R_COND);
                pop();
                const uint32_t d0 = depth;
                Info a = gen(e.kids[1]);
                no_ctx(a, "as a conditional branch");
                const size_t ja = code.size();
                emit(R_JMP);
                depth = d0;
                code[jc].b = (uint16_t)code.size();
                Info b = gen(e.kids[2]);
                no_ctx(b, "as a conditional branch");
                code[ja].b = (uint16_t)code.size();
                return merge(a, b);
            }
            case EX_BIN: return binary(e);
        }
        throw Reject{"internal: unknown node"};
    }

    // The strings a pattern expression can evaluate to, when they are finitely many and known now (a superset is fine: the value picks its
    // table at run time): literals, conditionals between such, concatenations, items of a configured String list, items of list / map
    // literals of such. false: not enumerable (a request field, client.country ...), or more than kMaxPatterns candidates.
    static constexpr size_t kMaxPatterns = 64;
    const ResidualList *clist_of(int ni) const {  // lists["name"] / lists.name
        const Ex &e = syn->nodes[(size_t)ni];
        std::string name;
        if (e.kind == EX_MEMBER && ctx_of(e.kids[0]) == 3) name = e.text;
        else if (e.kind == EX_INDEX && ctx_of(e.kids[0]) == 3 && syn->nodes[(size_t)e.kids[1]].kind == EX_STR) name = syn->nodes[(size_t)e.kids[1]].text;
        else return nullptr;
        for (auto &l : *host_lists) if (l.name == name) return &l;
        return nullptr;
    }
    bool enum_strings(int ni, std::set<std::string> &out) const {
        const Ex &e = syn->nodes[(size_t)ni];
        auto add_all = [&](const std::set<std::string> &more) { out.insert(more.begin(), more.end()); return out.size() <= kMaxPatterns; };
        switch (e.kind) {
            case EX_STR: out.insert(e.text); return out.size() <= kMaxPatterns;
            case EX_INT: case EX_FLOAT: case EX_BOOL: case EX_NULL: return true;  // (never a String: the call is an error whatever the set holds)
            case EX_COND: return enum_strings(e.kids[1], out) && enum_strings(e.kids[2], out);
            case EX_BIN: {
                if (e.op != B_ADD) return false;
                std::set<std::string> l, r, both;
                if (!enum_strings(e.kids[0], l) || !enum_strings(e.kids[1], r) || l.size() * r.size() > kMaxPatterns) return false;
                for (auto &a : l) for (auto &b : r) both.insert(a + b);
                return add_all(both);
            }
            case EX_INDEX: case EX_MEMBER: {
                if (e.kind == EX_INDEX) {
                    if (const ResidualList *cl = clist_of(e.kids[0])) {  // any item of the configured list
                        if (cl->type != PWAF_LIST_STRING) return true;
                        return add_all(std::set<std::string>(cl->strs.begin(), cl->strs.end()));
                    }
                }
                const Ex &o = syn->nodes[(size_t)e.kids[0]];
                if (o.kind == EX_LIST && e.kind == EX_INDEX) { for (int k : o.kids) if (!enum_strings(k, out)) return false; return true; }
                if (o.kind == EX_MAP) { for (size_t k = 1; k < o.kids.size(); k += 2) if (!enum_strings(o.kids[k], out)) return false; return true; }
                return false;
            }
            default: return false;
        }
    }
    uint32_t regex_set_id(const std::set<std::string> &patterns) {
        if (rxsets.size() >= 0x7FE) throw Reject{"too many computed matches() patterns in residual programs"};
        RegexSetDesc d{(uint32_t)(rxitems.size() / 4), (uint32_t)patterns.size()};
        for (const std::string &p : patterns) {
            const uint32_t id = regex_id(p);
            rxitems.push_back(str_const(p)); rxitems.push_back((uint32_t)p.size()); rxitems.push_back(id); rxitems.push_back(0);
        }
        rxsets.push_back(d);
        return 0x800u | (uint32_t)(rxsets.size() - 1);
    }

    Info call(const Ex &e) {
        Info r;
        const std::string &f = e.text;
        const size_t argc = e.kids.size() - 1;
        Info recv = gen(e.kids[0]);
        if (recv.ctx) {
            // the context maps answer contains(literal key) / length() at compile time
            if (f == "contains" && argc == 1) {
                const Ex &k = syn->nodes[(size_t)e.kids[1]];
                if (k.kind == EX_STR) { push_bool(ctx_has(recv.ctx, k.text)); return r; }
                if (k.kind == EX_INT || k.kind == EX_FLOAT || k.kind == EX_BOOL || k.kind == EX_NULL) { push_err(); return r; }
                // a computed key: the map's key set as a value, then the generic contains
                Info mr;
                gen_ctx_map(recv.ctx, true, mr);
                Info ka = gen(e.kids[1]);
                if (ka.ctx) throw Reject{"a context map used as a function argument"};
                if (ka.clist) throw Reject{"a configured list used as a function argument"};
                emit(R_CALL, FN_CONTAINS, (uint32_t)(1u << 12));
                pop();
                return r;
            }
            if (f == "length" && argc == 0) {
                if (recv.ctx == 1) { push_int(6); return r; }  // host, url, path, method, user_agent + the headers map (extension)
                if (recv.ctx == 2) { push_int(4); return r; }
                if (recv.ctx == 3) { std::set<std::string> names; for (auto &l : *host_lists) names.insert(l.name); push_int((int64_t)names.size()); return r; }
                if (closed_headers == nullptr) throw Reject{"length() of the headers map (it holds the names the whole rule set mentions)"};
                push_int((int64_t)closed_headers->size());
                return r;
            }
            // any other method on a map: the arguments are evaluated, then the call fails (or the argument count is wrong)
            push_err();
            return r;
        }
        uint8_t fn;
        if (f == "contains") fn = FN_CONTAINS;
        else if (f == "starts_with") fn = FN_STARTS;
        else if (f == "ends_with") fn = FN_ENDS;
        else if (f == "length") fn = FN_LENGTH;
        else if (f == "matches") fn = FN_MATCHES;
        else fn = 0xFF;  // undeclared function: the operands are evaluated, then the call fails
        const bool arity_ok = fn != 0xFF && (fn == FN_LENGTH ? argc == 0 : argc == 1);
        uint32_t aux = 0;
        if (fn == FN_MATCHES && argc == 1) {
            const Ex &p = syn->nodes[(size_t)e.kids[1]];
            if (p.kind != EX_STR) {
                if (p.kind == EX_INT || p.kind == EX_FLOAT || p.kind == EX_BOOL || p.kind == EX_NULL) aux = 0xFFFu;  // String operands required: an error either way
                else {
                    std::set<std::string> cands;
                    if (!enum_strings(e.kids[1], cands)) throw Reject{"matches() with a pattern that is not a String literal (nor one of at most 64 strings known when the rule is compiled)"};
                    aux = regex_set_id(cands);
                }
            } else {
                aux = regex_id(p.text);
            }
        }
        for (size_t k = 1; k <= argc; k++) {
            Info a = gen(e.kids[k]);
            if (a.ctx) {
                // a context map as an argument: no function takes one (contains(list, map) compares values: unsupported as a value)
                throw Reject{"a context map used as a function argument"};
            }
            if (a.clist) throw Reject{"a configured list used as a function argument"};
        }
        if (!arity_ok || argc > 15) {
            emit(R_FAIL, 0, (uint32_t)argc + 1);
            pop((int)argc);
            return r;
        }
        emit(R_CALL, fn, (uint32_t)(argc << 12) | aux);
        pop((int)argc);
        return r;
    }

    // ---- what is statically known about an operand of && / || (round 6: cheapest operand first) --------------------------------------
    // `A && B` matches iff A is Bool(true) and B is Bool(true); the ORDER only decides which execution error (or non-Bool operand) is
    // met first, and the per-rule error counters must agree with the reference's evaluation order (pingoo/rules.rs:41-45 logs each).
    // So a chain of && (or ||) is reordered only when every operand is PURE: statically a Bool whose evaluation cannot fail — then any
    // order gives the same value and no error either way. Integer arithmetic is pure when interval arithmetic over the operands' ranges
    // (lengths and client.asn < 2^32, ports < 2^16, constants) shows that it cannot overflow or divide by zero. The cheapest operand
    // goes first: comparisons of integers and lengths read no string bytes at all, an equality reads them only when the lengths agree,
    // `contains` / `matches` / orderings walk the text — `(host + ":" + method).matches(..) && path + "x" == "/qx"` evaluated the regex for
    // every request (0.23 ms per 10M requests) before the equality that nearly always fails.
    enum { TY_NONE = 0, TY_BOOL, TY_INT, TY_STR };
    struct Pure {
        int ty = TY_NONE;  // TY_NONE: not known to be an error-free Bool / Int / String
        uint32_t cost = 0;
        __int128 lo = 0, hi = 0;  // TY_INT: every value it can take lies in [lo, hi], inside int64
    };
    static Pure mk_pure(int ty, uint32_t cost, __int128 lo = 0, __int128 hi = 0) { Pure p; p.ty = ty; p.cost = cost; p.lo = lo; p.hi = hi; return p; }
    static bool fits64(__int128 v) { return v >= (__int128)INT64_MIN && v <= (__int128)INT64_MAX; }
    Pure pure_ctx(int ctx, const std::string &key) const {
        if (ctx == 1) {
            for (const char *f : {"host", "url", "path", "method", "user_agent"}) if (key == f) return mk_pure(TY_STR, 0);
            return Pure{};
        }
        if (ctx == 4) {
            if (closed_headers == nullptr) return Pure{};
            for (const std::string &n : *closed_headers) if (n == key) return mk_pure(TY_STR, 0);
            return Pure{};
        }
        if (ctx == 2) {
            if (key == "remote_port") return mk_pure(TY_INT, 0, 0, 65535);
            if (key == "asn") return mk_pure(TY_INT, 0, 0, 0xFFFFFFFFll);
            if (key == "country") return mk_pure(TY_STR, 0);
        }
        return Pure{};
    }
    Pure pure(int ni) const {
        const Ex &e = syn->nodes[(size_t)ni];
        switch (e.kind) {
            case EX_INT: return mk_pure(TY_INT, 0, e.ival, e.ival);
            case EX_STR: return mk_pure(TY_STR, 0);
            case EX_BOOL: return mk_pure(TY_BOOL, 0);
            case EX_MEMBER: { const int c = ctx_of(e.kids[0]); return c ? pure_ctx(c, e.text) : Pure{}; }
            case EX_INDEX: {
                const int c = ctx_of(e.kids[0]);
                const Ex &ix = syn->nodes[(size_t)e.kids[1]];
                return (c && ix.kind == EX_STR) ? pure_ctx(c, ix.text) : Pure{};
            }
            case EX_NOT: { const Pure x = pure(e.kids[0]); return x.ty == TY_BOOL ? x : Pure{}; }
            case EX_NEG: {
                const Pure x = pure(e.kids[0]);
                if (x.ty != TY_INT || !fits64(-x.lo) || !fits64(-x.hi)) return Pure{};
                return mk_pure(TY_INT, x.cost, -x.hi, -x.lo);
            }
            case EX_COND: {
                const Pure c = pure(e.kids[0]), a = pure(e.kids[1]), b = pure(e.kids[2]);
                if (c.ty != TY_BOOL || a.ty == TY_NONE || a.ty != b.ty) return Pure{};
                return mk_pure(a.ty, c.cost + std::max(a.cost, b.cost), std::min(a.lo, b.lo), std::max(a.hi, b.hi));
            }
            case EX_MCALL: {
                if (ctx_of(e.kids[0])) return Pure{};
                const Pure r = pure(e.kids[0]);
                if (r.ty != TY_STR) return Pure{};
                const size_t argc = e.kids.size() - 1;
                if (e.text == "length" && argc == 0) return mk_pure(TY_INT, r.cost, 0, 0xFFFFFFFFll);
                if (argc != 1) return Pure{};
                if (e.text == "matches") {
                    const Ex &p = syn->nodes[(size_t)e.kids[1]];
                    if (p.kind != EX_STR) return Pure{};
                    int status;
                    std::string err;
                    regex_parse(p.text, status, err);
                    return status == 0 ? mk_pure(TY_BOOL, r.cost + 8) : Pure{};
                }
                const Pure a = pure(e.kids[1]);
                if (a.ty != TY_STR) return Pure{};
                if (e.text == "contains") return mk_pure(TY_BOOL, r.cost + a.cost + 6);
                if (e.text == "starts_with" || e.text == "ends_with") return mk_pure(TY_BOOL, r.cost + a.cost + 2);
                return Pure{};
            }
            case EX_BIN: {
                if (e.op == B_IN) return Pure{};
                const Pure l = pure(e.kids[0]), r = pure(e.kids[1]);
                if (l.ty == TY_NONE || r.ty == TY_NONE) return Pure{};
                const uint32_t cost = l.cost + r.cost;
                switch (e.op) {
                    case B_OR: case B_AND: return (l.ty == TY_BOOL && r.ty == TY_BOOL) ? mk_pure(TY_BOOL, cost) : Pure{};
                    case B_EQ: case B_NE: return mk_pure(TY_BOOL, cost + ((l.ty == TY_STR && r.ty == TY_STR) ? 1u : 0u));  // (other types: unequal, not an error — D4)
                    case B_LT: case B_LE: case B_GT: case B_GE:
                        if (l.ty == TY_INT && r.ty == TY_INT) return mk_pure(TY_BOOL, cost);
                        if (l.ty == TY_STR && r.ty == TY_STR) return mk_pure(TY_BOOL, cost + 3);
                        return Pure{};
                    case B_ADD:
                        if (l.ty == TY_STR && r.ty == TY_STR) return mk_pure(TY_STR, cost + 1);  // (a rope: its segments are bounded by the compiler)
                        [[fallthrough]];
                    case B_SUB: case B_MUL: {
                        if (l.ty != TY_INT || r.ty != TY_INT) return Pure{};
                        __int128 c[4];
                        if (e.op == B_ADD) { c[0] = c[1] = l.lo + r.lo; c[2] = c[3] = l.hi + r.hi; }
                        else if (e.op == B_SUB) { c[0] = c[1] = l.lo - r.hi; c[2] = c[3] = l.hi - r.lo; }
                        else { c[0] = l.lo * r.lo; c[1] = l.lo * r.hi; c[2] = l.hi * r.lo; c[3] = l.hi * r.hi; }
                        const __int128 lo = std::min(std::min(c[0], c[1]), std::min(c[2], c[3])), hi = std::max(std::max(c[0], c[1]), std::max(c[2], c[3]));
                        return (fits64(lo) && fits64(hi)) ? mk_pure(TY_INT, cost, lo, hi) : Pure{};
                    }
                    case B_DIV: case B_MOD: {
                        if (l.ty != TY_INT || r.ty != TY_INT || r.lo != r.hi || r.lo == 0 || r.lo == -1) return Pure{};  // a constant divisor other than 0 (and -1: INT64_MIN / -1)
                        const __int128 d = r.lo < 0 ? -r.lo : r.lo, m = std::max(l.lo < 0 ? -l.lo : l.lo, l.hi < 0 ? -l.hi : l.hi);
                        return e.op == B_DIV ? mk_pure(TY_INT, cost, -m, m) : mk_pure(TY_INT, cost, l.lo < 0 ? -(d - 1) : 0, d - 1);
                    }
                    default: return Pure{};
                }
            }
            default: return Pure{};
        }
    }
    void chain_operands(int ni, BinOp op, std::vector<int> &out) const {
        const Ex &e = syn->nodes[(size_t)ni];
        if (e.kind == EX_BIN && e.op == op) { chain_operands(e.kids[0], op, out); chain_operands(e.kids[1], op, out); }
        else out.push_back(ni);
    }

    Info binary(const Ex &e) {
        Info r;
        if ((e.op == B_OR || e.op == B_AND) && reorder) {
            std::vector<int> ops;
            chain_operands(e.kids[0], e.op, ops);
            chain_operands(e.kids[1], e.op, ops);
            std::vector<std::pair<uint32_t, int>> byc;
            bool all_pure = true;
            for (int o : ops) {
                const Pure p = pure(o);
                all_pure = all_pure && p.ty == TY_BOOL;
                byc.emplace_back(p.cost, o);
            }
            if (all_pure) {
                std::stable_sort(byc.begin(), byc.end(), [](const std::pair<uint32_t, int> &a, const std::pair<uint32_t, int> &b) { return a.first < b.first; });
                gen(byc[0].second);
                for (size_t k = 1; k < byc.size(); k++) {  // (the left-nested chain `((a && b) && c)`, operand by operand)
                    const size_t j = code.size();
                    emit(e.op == B_AND ? R_AND_L : R_OR_L);
                    pop();
                    gen(byc[k].second);
                    emit(R_BOOL_CHK);
                    code[j].b = (uint16_t)code.size();
                }
                return r;
            }
        }
        if (e.op == B_OR || e.op == B_AND) {
            Info l = gen(e.kids[0]);
            if (l.ctx) { push_err(); return r; }  // Bool operands required (the right side is not evaluated)
            const size_t j = code.size();
            emit(e.op == B_AND ? R_AND_L : R_OR_L);
            pop();
            Info rr = gen(e.kids[1]);
            if (rr.ctx) push_err();
            emit(R_BOOL_CHK);
            code[j].b = (uint16_t)code.size();
            return r;
        }
        if (e.op == B_IN) {
            // x in <context map>: a literal key is answered at compile time
            const int rc = ctx_of(e.kids[1]);
            if (rc) {
                const Ex &k = syn->nodes[(size_t)e.kids[0]];
                if (k.kind == EX_STR) { push_bool(ctx_has(rc, k.text)); return r; }
                if (k.kind == EX_INT || k.kind == EX_FLOAT || k.kind == EX_BOOL || k.kind == EX_NULL) { push_err(); return r; }
                // a computed key: evaluate it, then the map's key set as a value and the generic `in`
                Info kl = gen(e.kids[0]);
                if (kl.ctx) throw Reject{"a context map on the left of `in`"};
                no_clist(kl, "on the left of `in`");
                Info mr;
                gen_ctx_map(rc, true, mr);
                emit(R_BIN, (uint8_t)B_IN);
                pop();
                return r;
            }
            Info l = gen(e.kids[0]);
            if (l.ctx) throw Reject{"a context map on the left of `in`"};
            no_clist(l, "on the left of `in`");
            gen(e.kids[1]);
            emit(R_BIN, (uint8_t)B_IN);
            pop();
            return r;
        }
        Info l = gen(e.kids[0]);
        Info rr = gen(e.kids[1]);
        if (l.ctx || rr.ctx) throw Reject{"a context map as an operand"};
        if (l.clist || rr.clist) throw Reject{"a configured list as an operand of a comparison or of `+`"};
        emit(R_BIN, (uint8_t)e.op);
        pop();
        if (e.op == B_ADD) {
            r.seg = l.seg + rr.seg;
            if (r.seg > kMaxRope) throw Reject{"a concatenation of more than " + std::to_string(kMaxRope) + " strings"};
            r.len = l.len + rr.len;
            r.inner = std::max(l.inner, rr.inner);
            r.nest = std::max(l.nest, rr.nest);
            use_heap(std::max(r.seg, r.len));
        }
        return r;
    }
};

ResidualBuilder::ResidualBuilder() : impl(new Impl) {}
ResidualBuilder::~ResidualBuilder() { delete impl; }
size_t ResidualBuilder::n_rules() const { return impl->entries.size(); }
bool ResidualBuilder::needs_geo() const { return impl->needs_geo; }

void collect_header_names(const Syntax &syn, std::vector<std::string> &names) {
    if (syn.root < 0) return;
    auto node = [&](int i) -> const Ex & { return syn.nodes[(size_t)i]; };
    auto is_headers = [&](int i) {
        const Ex &n = node(i);
        if (n.kind == EX_MEMBER) return n.text == "headers" && n.kids.size() == 1 && node(n.kids[0]).kind == EX_IDENT && node(n.kids[0]).text == "http_request";
        if (n.kind == EX_INDEX)
            return n.kids.size() == 2 && node(n.kids[0]).kind == EX_IDENT && node(n.kids[0]).text == "http_request" && node(n.kids[1]).kind == EX_STR && node(n.kids[1]).text == "headers";
        return false;
    };
    auto add = [&](const std::string &name) {
        for (auto &h : names) if (h == name) return;
        names.push_back(name);
    };
    std::vector<int> st{syn.root};
    while (!st.empty()) {
        const int i = st.back();
        st.pop_back();
        const Ex &n = node(i);
        if (n.kind == EX_MEMBER && n.kids.size() == 1 && is_headers(n.kids[0])) add(n.text);
        if (n.kind == EX_INDEX && n.kids.size() == 2 && is_headers(n.kids[0]) && node(n.kids[1]).kind == EX_STR) add(node(n.kids[1]).text);
        if (n.kind == EX_BIN && n.op == B_IN && n.kids.size() == 2 && is_headers(n.kids[1]) && node(n.kids[0]).kind == EX_STR) add(node(n.kids[0]).text);
        if (n.kind == EX_MCALL && n.text == "contains" && n.kids.size() == 2 && is_headers(n.kids[0]) && node(n.kids[1]).kind == EX_STR) add(node(n.kids[1]).text);
        for (size_t k = n.kids.size(); k-- > 0;) st.push_back(n.kids[k]);
    }
}

int ResidualBuilder::compile_rule(const Syntax &syn, const std::vector<ResidualList> &lists, const std::function<int(const std::string &)> &header_field, std::string &why,
                                  const std::vector<std::string> *closed_headers) {
    Impl &m = *impl;
    m.closed_headers = closed_headers;
    // a failed rule must leave no trace: snapshot the growing tables
    const size_t c0 = m.code.size(), k0 = m.consts.size(), s0 = m.strpool.size(), l0 = m.lists.size(), ls0 = m.lstr.size(), li0 = m.lints.size(), n0 = m.nets.size(), r0 = m.regexes.size(), xs0 = m.rxsets.size(), xi0 = m.rxitems.size();
    const auto list_ids0 = m.list_ids;
    const auto regex_ids0 = m.regex_ids;
    const bool geo0 = m.needs_geo;
    m.syn = &syn;
    m.host_lists = &lists;
    m.header_field = header_field;
    m.depth = m.max_depth = m.heap = 0;
    try {
        const uint32_t entry = (uint32_t)m.code.size();
        Info top = m.gen(syn.root);
        if (top.ctx) m.push_err();  // a map is not Bool(true)
        m.emit(R_END);
        m.entries.push_back(entry);
        m.heap_items = std::max(m.heap_items, m.heap);
        return (int)m.entries.size() - 1;
    } catch (Reject &rj) {
        why = rj.why;
    }
    m.code.resize(c0); m.consts.resize(k0); m.strpool.resize(s0); m.lists.resize(l0); m.lstr.resize(ls0); m.lints.resize(li0); m.nets.resize(n0);
    m.regexes.resize(r0); m.regex_tabs.resize(r0); m.rxsets.resize(xs0); m.rxitems.resize(xi0);
    m.list_ids = list_ids0; m.regex_ids = regex_ids0; m.needs_geo = geo0;
    return -1;
}

std::vector<uint8_t> ResidualBuilder::blob() const {
    const Impl &m = *impl;
    std::vector<uint8_t> out(sizeof(Header));
    auto align = [&](size_t a) { while (out.size() % a) out.push_back(0); };
    auto put = [&](const void *p, size_t n, size_t a) -> uint32_t {
        align(a);
        const uint32_t at = (uint32_t)out.size();
        const uint8_t *b = (const uint8_t *)p;
        out.insert(out.end(), b, b + n);
        return at;
    };
    Header h{};
    h.magic = 0x314D5652u;  // "RVM1"
    h.n_rules = (uint32_t)m.entries.size();
    h.rules = put(m.entries.data(), m.entries.size() * 4, 4);
    h.code = put(m.code.data(), m.code.size() * sizeof(Ins), 4);
    h.consts = put(m.consts.data(), m.consts.size() * sizeof(Val), 8);
    h.strpool = put(m.strpool.data(), m.strpool.size(), 1);
    h.lists = put(m.lists.data(), m.lists.size() * sizeof(ListDesc), 4);
    h.lstr = put(m.lstr.data(), m.lstr.size() * 4, 4);
    h.lints = put(m.lints.data(), m.lints.size() * 8, 8);
    h.nets = put(m.nets.data(), m.nets.size() * sizeof(NetItem), 4);
    std::vector<RegexDesc> rd = m.regexes;
    for (size_t k = 0; k < rd.size(); k++) {
        const uint32_t trans_bytes = rd[k].trans, n_states = rd[k].flags;
        const uint32_t at = put(m.regex_tabs[k].data(), m.regex_tabs[k].size(), 4);
        rd[k].trans = at;
        rd[k].classmap = at + trans_bytes;
        rd[k].flags = at + trans_bytes + 256;
        if (rd[k].umap) rd[k].umap += at;
        (void)n_states;
    }
    h.regexes = put(rd.data(), rd.size() * sizeof(RegexDesc), 4);
    h.rxsets = put(m.rxsets.data(), m.rxsets.size() * sizeof(RegexSetDesc), 4);
    h.rxitems = put(m.rxitems.data(), m.rxitems.size() * 4, 4);
    h.needs_geo = m.needs_geo ? 1u : 0u;
    h.heap_items = m.heap_items;
    align(16);
    out.resize(out.size() + 16);  // (strings are read eight bytes at a time: residual.h load8)
    h.total_bytes = (uint32_t)out.size();
    memcpy(out.data(), &h, sizeof h);
    return out;
}

}  // namespace pwaf
