// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product
// path (pingoo_amd/, libpwaf.so). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it, and only as the checker / the timed CPU "port" baseline.
//
// oracle_engine.cpp — CPU restatement of the reference's per-request verdict logic, structured the
// way the reference is: per request build a context, walk the rules in order, first match wins.
// PARITY UNPINNED by the reference (no tests, un-vendored interpreter: see oracle_expr.h).
//
//   step (reference file:line)                                   here
//   -----------------------------------------------------------  -------------------------------
//   geoip.lookup: loopback|multicast -> not found; miss/error -> {0,"XX"}
//       pingoo/geoip.rs:73-91,111-118; http_listener.rs:143-157   Engine::geo_lookup
//   record decode: country must be 2 x 'A'..'Z' else lookup errors
//       pingoo/geoip.rs:128-142                                   Engine ctor (invalid entries)
//   gate A: ua.is_empty() || ua.len() >= 256 -> Block
//       http_listener.rs:196-198                                  evaluate_one
//   gate B: path.starts_with("/__pingoo/captcha") -> rules skipped
//       http_listener.rs:200-204                                  evaluate_one
//   RequestData / ClientData -> context variables
//       pingoo/rules.rs:16-34; http_listener.rs:207-219,239-249   build_context
//   for rule in rules { if rule.match_request(ctx) { for action ... } }
//       http_listener.rs:251-264; pingoo/rules.rs:37-51           evaluate_one
//   lists: CSV column 0 trimmed; Int -> i64; Ip -> IpNetwork
//       pingoo/lists.rs:90-108,115-125                            Engine ctor
//   field derivation (trim, to_str, heapless<256>, trailing '/')
//       http_listener.rs:159-165,284-296; http_utils.rs:114-116   pwaf_oracle_derive_*
#include <atomic>
#include <deque>
#include <memory>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/pwaf.h"
#include "oracle_expr.h"

using namespace oracle;

namespace {

struct ORule {
    std::string name;
    bool has_expr = false;
    Program prog;
    std::vector<uint8_t> actions;
};

struct GeoNode {
    int32_t child[2] = {-1, -1};
    int32_t rec = -1;
};
struct GeoRec {
    uint32_t asn;
    uint8_t country[2];
    bool valid;
};

struct Engine {
    std::vector<ORule> rules;
    MapVal lists;
    std::deque<ListVal> list_store;
    bool has_geo = false;
    std::vector<GeoNode> geo4, geo6;
    std::vector<GeoRec> georecs;
    uint32_t flags = 0;
    std::vector<std::string> header_names;  // EXTENSION: the names behind the batches' header columns

    static void geo_insert(std::vector<GeoNode> &t, const uint8_t *addr, int plen, int rec) {
        if (t.empty()) t.emplace_back();
        int cur = 0;
        for (int b = 0; b < plen; b++) {
            int bit = (addr[b >> 3] >> (7 - (b & 7))) & 1;
            if (t[cur].child[bit] < 0) {
                t[cur].child[bit] = (int32_t)t.size();
                t.emplace_back();
            }
            cur = t[cur].child[bit];
        }
        t[cur].rec = rec;  // later duplicates override earlier ones
    }
    // maxminddb::Reader::lookup: longest prefix; returns rec index or -1
    static int geo_find(const std::vector<GeoNode> &t, const uint8_t *addr, int nbits) {
        if (t.empty()) return -1;
        int cur = 0, best = t[0].rec;
        for (int b = 0; b < nbits; b++) {
            int bit = (addr[b >> 3] >> (7 - (b & 7))) & 1;
            cur = t[cur].child[bit];
            if (cur < 0) break;
            if (t[cur].rec >= 0) best = t[cur].rec;
        }
        return best;
    }
    void geo_lookup(const uint8_t *ip, bool v6, uint32_t &asn, uint8_t country[2]) const {
        asn = 0;
        country[0] = 'X';
        country[1] = 'X';
        if (!has_geo) return;
        // IpAddr::is_loopback / is_multicast (geoip.rs:74-76)
        if (!v6) {
            if (ip[0] == 127) return;
            if ((ip[0] & 0xF0) == 0xE0) return;
        } else {
            bool lo = true;
            for (int k = 0; k < 15; k++) if (ip[k]) lo = false;
            if (lo && ip[15] == 1) return;
            if (ip[0] == 0xFF) return;
        }
        int r = v6 ? geo_find(geo6, ip, 128) : geo_find(geo4, ip, 32);
        if (r < 0) return;                 // AddressNotFound -> default
        if (!georecs[r].valid) return;     // decode error -> default (http_listener.rs:148-153)
        asn = georecs[r].asn;
        country[0] = georecs[r].country[0];
        country[1] = georecs[r].country[1];
    }
};

static std::string_view trim_ws(std::string_view s) {
    // str::trim on what the csv crate yields; ASCII whitespace + the Unicode White_Space that can
    // appear in UTF-8 is out of scope for list items (IPs / ints / tokens)
    size_t b = 0, e = s.size();
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0B || c == 0x0C; };
    while (b < e && ws(s[b])) b++;
    while (e > b && ws(s[e - 1])) e--;
    return s.substr(b, e - b);
}

static bool parse_i64(std::string_view s, int64_t &out) {
    // str::parse::<i64>: optional sign, decimal digits, no whitespace, overflow is an error
    if (s.empty()) return false;
    size_t p = 0;
    bool neg = false;
    if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; p = 1; }
    if (p == s.size()) return false;
    __int128 v = 0;
    for (; p < s.size(); p++) {
        if (s[p] < '0' || s[p] > '9') return false;
        v = v * 10 + (s[p] - '0');
        if (v > ((__int128)1 << 63)) return false;
    }
    if (neg) v = -v;
    if (v > INT64_MAX || v < INT64_MIN) return false;
    out = (int64_t)v;
    return true;
}

struct ReqView {
    std::string_view host, url, path, method, ua;
    std::vector<std::string_view> headers;  // EXTENSION: values of the engine-declared header names, in that order
    uint8_t ip[16];
    bool v6;
    uint16_t port;
    uint8_t flags;
    bool has_geo;
    uint32_t asn;
    uint8_t country[2];
};

// EXTENSION (the reference's RequestData has no headers, pingoo/rules.rs:16-25): http_request.headers is a
// Map<String, String> holding exactly the header names declared with pwaf_oracle_set_header_names (absent header = "").
static void build_context(const Engine &e, const ReqView &r, uint32_t asn, const uint8_t country[2], MapVal &http, MapVal &client, MapVal &headers, Context &ctx) {
    headers.items.clear();
    for (size_t k = 0; k < e.header_names.size(); k++) headers.items.emplace(e.header_names[k], Val::str(k < r.headers.size() ? r.headers[k] : std::string_view()));
    http.items.clear();
    {
        Val hm; hm.k = Val::Map; hm.map = &headers;
        http.items.emplace("headers", hm);
    }
    http.items.emplace("host", Val::str(r.host));
    http.items.emplace("url", Val::str(r.url));
    http.items.emplace("path", Val::str(r.path));
    http.items.emplace("method", Val::str(r.method));
    http.items.emplace("user_agent", Val::str(r.ua));
    client.items.clear();
    Val ip;
    ip.k = Val::Ip;
    ip.ip.v6 = r.v6;
    memcpy(ip.ip.b, r.ip, r.v6 ? 16 : 4);
    client.items.emplace("ip", ip);
    client.items.emplace("remote_port", Val::integer((int64_t)r.port));
    client.items.emplace("asn", Val::integer((int64_t)asn));
    client.items.emplace("country", Val::str(std::string_view((const char *)country, 2)));
    ctx.vars.clear();
    Val h; h.k = Val::Map; h.map = &http;
    Val c; c.k = Val::Map; c.map = &client;
    Val l; l.k = Val::Map; l.map = &e.lists;
    ctx.vars.emplace("http_request", h);
    ctx.vars.emplace("client", c);
    ctx.vars.emplace("lists", l);
}

static bool match_request(const ORule &rule, const Context &ctx) {
    // pingoo/rules.rs:37-51
    if (!rule.has_expr) return true;
    Val v = execute(rule.prog, ctx);
    if (v.k == Val::Error) return false;
    return v.k == Val::Bool && v.b;
}

static void evaluate_one(const Engine &e, const ReqView &r, pwaf_verdict &out) {
    out.action = PWAF_ACTION_ALLOW;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
    out.rule_idx = PWAF_RULE_NONE;
    uint32_t asn;
    uint8_t country[2];
    if (r.has_geo) {
        asn = r.asn;
        country[0] = r.country[0];
        country[1] = r.country[1];
    } else {
        e.geo_lookup(r.ip, r.v6, asn, country);
    }
    if (!(e.flags & PWAF_OPT_NO_UA_GATE)) {
        if (r.ua.empty() || r.ua.size() >= 256) {
            out.action = PWAF_ACTION_BLOCK;
            out.rule_idx = PWAF_RULE_UA_GATE;
            return;
        }
    }
    if (!(e.flags & PWAF_OPT_NO_CAPTCHA_BYPASS)) {
        static const std::string_view kPrefix = "/__pingoo/captcha";
        if (r.path.size() >= kPrefix.size() && r.path.substr(0, kPrefix.size()) == kPrefix) {
            out.action = PWAF_ACTION_BYPASS;
            out.rule_idx = PWAF_RULE_CAPTCHA_ENDPOINT;
            return;
        }
    }
    bool verified = (r.flags & PWAF_FLAG_CAPTCHA_VERIFIED) != 0;
    MapVal http, client, headers;
    Context ctx;
    build_context(e, r, asn, country, http, client, headers, ctx);
    for (size_t k = 0; k < e.rules.size(); k++) {
        const ORule &rule = e.rules[k];
        if (!match_request(rule, ctx)) continue;
        for (uint8_t a : rule.actions) {
            if (a == PWAF_RULE_ACTION_BLOCK) {
                out.action = PWAF_ACTION_BLOCK;
                out.rule_idx = (uint32_t)k;
                return;
            }
            if (a == PWAF_RULE_ACTION_CAPTCHA && !verified) {
                out.action = PWAF_ACTION_CAPTCHA;
                out.rule_idx = (uint32_t)k;
                return;
            }
        }
    }
}

static void seterr(char *buf, size_t len, const std::string &m) {
    if (!buf || !len) return;
    snprintf(buf, len, "%s", m.c_str());
}

static bool batch_view(const pwaf_batch *b, uint32_t i, ReqView &r) {
    std::string_view *f[5] = {&r.host, &r.url, &r.path, &r.method, &r.ua};
    for (int k = 0; k < 5; k++) {
        uint32_t o0 = b->field[k].offsets[i], o1 = b->field[k].offsets[i + 1];
        if (o1 < o0) return false;
        *f[k] = std::string_view((const char *)b->field[k].data + o0, o1 - o0);
    }
    r.headers.clear();
    for (uint32_t k = 0; k < b->n_headers && b->headers; k++) {
        uint32_t o0 = b->headers[k].offsets[i], o1 = b->headers[k].offsets[i + 1];
        if (o1 < o0) return false;
        r.headers.emplace_back((const char *)b->headers[k].data + o0, o1 - o0);
    }
    memcpy(r.ip, b->ip + 16 * (size_t)i, 16);
    r.v6 = b->ip_is_v6[i] != 0;
    r.port = b->port[i];
    r.flags = b->flags[i];
    r.has_geo = b->asn && b->country;
    if (r.has_geo) {
        r.asn = b->asn[i];
        memcpy(r.country, &b->country[i], 2);
        if (r.country[0] < 'A' || r.country[0] > 'Z' || r.country[1] < 'A' || r.country[1] > 'Z') return false;
    }
    return true;
}

}  // namespace

extern "C" {

int pwaf_oracle_compile_expression(const char *expr, char *errbuf, size_t errlen) {
    // rules::compile_expression (rules/rules.rs:45-53)
    Program p;
    std::string err;
    if (!expr || !compile(expr, p, err)) {
        seterr(errbuf, errlen, "Expression is not valid: " + err);
        return PWAF_E_SYNTAX;
    }
    return PWAF_OK;
}

int pwaf_oracle_validate_expression(const char *expr, char *errbuf, size_t errlen) {
    // rules::validate_expression (rules/rules.rs:55-77)
    if (!expr || !*expr) {
        seterr(errbuf, errlen, "Expression is not valid: expression is empty");
        return PWAF_E_SYNTAX;
    }
    Program p;
    std::string err;
    if (!compile(expr, p, err)) {
        seterr(errbuf, errlen, "Expression is not valid: " + err);
        return PWAF_E_SYNTAX;
    }
    for (auto &f : p.functions) {
        if (f == "@in") {
            seterr(errbuf, errlen, "Expression is not valid: unknown operator: in");
            return PWAF_E_SYNTAX;
        }
    }
    return PWAF_OK;
}

int pwaf_oracle_create(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists, size_t n_lists,
                       const pwaf_geoip_table *geoip, uint32_t flags, void **out, char *errbuf, size_t errlen) {
    auto e = std::make_unique<Engine>();
    e->flags = flags;
    for (size_t k = 0; k < n_rules; k++) {
        e->rules.emplace_back();
        ORule &r = e->rules.back();
        r.name = rules[k].name ? rules[k].name : "";
        for (uint32_t a = 0; a < rules[k].n_actions; a++) {
            uint8_t act = rules[k].actions[a];
            if (act != PWAF_RULE_ACTION_BLOCK && act != PWAF_RULE_ACTION_CAPTCHA) {
                seterr(errbuf, errlen, "rule " + r.name + ": unknown action");
                return PWAF_E_INVALID_ARG;
            }
            r.actions.push_back(act);
        }
        if (rules[k].expression) {
            r.has_expr = true;
            std::string err;
            if (!compile(rules[k].expression, r.prog, err)) {
                seterr(errbuf, errlen, "error parsing rules: Expression is not valid: " + err + " (rule " + r.name + ")");
                return PWAF_E_SYNTAX;
            }
        }
    }
    for (size_t k = 0; k < n_lists; k++) {
        e->list_store.emplace_back();
        ListVal &lv = e->list_store.back();
        lv.owned.reserve(lists[k].n_items);
        for (uint32_t i = 0; i < lists[k].n_items; i++) {
            std::string_view item = trim_ws(lists[k].items[i] ? lists[k].items[i] : "");
            if (lists[k].type == PWAF_LIST_STRING) {
                lv.owned.emplace_back(item);
            } else if (lists[k].type == PWAF_LIST_INT) {
                int64_t v;
                if (!parse_i64(item, v)) {
                    seterr(errbuf, errlen, std::string("error parsing list ") + lists[k].name + " at line " + std::to_string(i + 1) + ": error parsing int");
                    return PWAF_E_LIST;
                }
                lv.items.push_back(Val::integer(v));
            } else if (lists[k].type == PWAF_LIST_IP) {
                IpNet n;
                std::string err;
                if (!parse_ipnet(item, n, err)) {
                    seterr(errbuf, errlen, std::string("error parsing list ") + lists[k].name + " at line " + std::to_string(i + 1) + ": error parsing IP network: " + err);
                    return PWAF_E_LIST;
                }
                Val v;
                v.k = Val::Net;
                v.net = n;
                lv.items.push_back(v);
            } else {
                seterr(errbuf, errlen, "unknown list type");
                return PWAF_E_INVALID_ARG;
            }
        }
        if (lists[k].type == PWAF_LIST_STRING)
            for (auto &s : lv.owned) lv.items.push_back(Val::str(s));
        Val v;
        v.k = Val::List;
        v.list = &lv;
        e->lists.items[lists[k].name] = v;  // HashMap insert: a duplicate name replaces
    }
    if (geoip) {
        e->has_geo = true;
        for (size_t k = 0; k < geoip->n_entries; k++) {
            const pwaf_geoip_entry &g = geoip->entries[k];
            int maxlen = g.is_v6 ? 128 : 32;
            if (g.prefix_len > maxlen) {
                seterr(errbuf, errlen, "geoip: invalid prefix length");
                return PWAF_E_INVALID_ARG;
            }
            GeoRec rec;
            rec.asn = g.asn;
            rec.country[0] = g.country[0];
            rec.country[1] = g.country[1];
            rec.valid = g.country[0] >= 'A' && g.country[0] <= 'Z' && g.country[1] >= 'A' && g.country[1] <= 'Z';
            e->georecs.push_back(rec);
            Engine::geo_insert(g.is_v6 ? e->geo6 : e->geo4, g.addr, g.prefix_len, (int)e->georecs.size() - 1);
        }
    }
    // EXTENSION: the headers map holds exactly the names the rule set mentions with a literal key — http_request.headers["x"],
    // http_request.headers.x, "x" in http_request.headers, http_request.headers.contains("x") — in order of first use
    {
        auto is_headers = [](const Node &n) {
            if (n.k == Node::Member) return n.name == "headers" && n.kids.size() == 1 && n.kids[0]->k == Node::Ident && n.kids[0]->name == "http_request";
            if (n.k == Node::Index)
                return n.kids.size() == 2 && n.kids[0]->k == Node::Ident && n.kids[0]->name == "http_request" && n.kids[1]->k == Node::Lit &&
                       n.kids[1]->lit.k == Val::String && n.kids[1]->lit.s == "headers";
            return false;
        };
        auto add = [&](std::string_view name) {
            for (auto &h : e->header_names) if (h == name) return;
            e->header_names.emplace_back(name);
        };
        std::vector<const Node *> stack;
        for (auto &r : e->rules)
            if (r.has_expr && r.prog.root) stack.push_back(r.prog.root.get());
        std::vector<const Node *> order;  // pre-order, left to right, rule by rule
        {
            std::vector<const Node *> roots(stack.begin(), stack.end());
            for (const Node *root : roots) {
                std::vector<const Node *> st{root};
                while (!st.empty()) {
                    const Node *n = st.back();
                    st.pop_back();
                    order.push_back(n);
                    for (size_t k = n->kids.size(); k-- > 0;) st.push_back(n->kids[k].get());
                }
            }
        }
        for (const Node *n : order) {
            if (n->k == Node::Member && n->kids.size() == 1 && is_headers(*n->kids[0])) add(n->name);
            if (n->k == Node::Index && n->kids.size() == 2 && is_headers(*n->kids[0]) && n->kids[1]->k == Node::Lit && n->kids[1]->lit.k == Val::String)
                add(n->kids[1]->lit.s);
            if (n->k == Node::Bin && n->name == "in" && n->kids.size() == 2 && is_headers(*n->kids[1]) && n->kids[0]->k == Node::Lit && n->kids[0]->lit.k == Val::String)
                add(n->kids[0]->lit.s);
            if (n->k == Node::Call && n->has_receiver && n->name == "contains" && n->kids.size() == 2 && is_headers(*n->kids[0]) && n->kids[1]->k == Node::Lit &&
                n->kids[1]->lit.k == Val::String)
                add(n->kids[1]->lit.s);
        }
    }
    *out = e.release();
    return PWAF_OK;
}

void pwaf_oracle_destroy(void *h) { delete (Engine *)h; }

// EXTENSION: the header names the rule set mentions (first-use order) = the header columns a batch must carry, in this order.
uint32_t pwaf_oracle_header_count(void *h) { return (uint32_t)((Engine *)h)->header_names.size(); }
const char *pwaf_oracle_header_name(void *h, uint32_t i) {
    Engine &e = *(Engine *)h;
    return i < e.header_names.size() ? e.header_names[i].c_str() : "";
}

int pwaf_oracle_evaluate(void *h, const pwaf_batch *b, pwaf_verdict *out, int n_threads) {
    const Engine &e = *(const Engine *)h;
    if (!b || b->memory != PWAF_MEM_HOST) return PWAF_E_INVALID_ARG;
    uint32_t n = b->n;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> bad{0};
    auto work = [&](uint32_t lo, uint32_t hi) {
        ReqView r;
        for (uint32_t i = lo; i < hi; i++) {
            if (!batch_view(b, i, r)) { bad = 1; return; }
            evaluate_one(e, r, out[i]);
        }
    };
    if (n_threads == 1 || n < 64) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) {
            uint32_t lo = (uint32_t)((uint64_t)n * t / n_threads), hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
            th.emplace_back(work, lo, hi);
        }
        for (auto &t : th) t.join();
    }
    return bad ? PWAF_E_BATCH : PWAF_OK;
}

// Evaluates ONE rule's expression for request i: 1 = Bool(true), 0 = Bool(false), 2 = non-Bool, 3 = error.
int pwaf_oracle_execute_rule(void *h, uint32_t rule, const pwaf_batch *b, uint32_t i) {
    const Engine &e = *(const Engine *)h;
    if (rule >= e.rules.size() || i >= b->n) return -1;
    ReqView r;
    if (!batch_view(b, i, r)) return -1;
    uint32_t asn;
    uint8_t country[2];
    if (r.has_geo) { asn = r.asn; country[0] = r.country[0]; country[1] = r.country[1]; }
    else e.geo_lookup(r.ip, r.v6, asn, country);
    MapVal http, client, headers;
    Context ctx;
    build_context(e, r, asn, country, http, client, headers, ctx);
    if (!e.rules[rule].has_expr) return 1;
    Val v = execute(e.rules[rule].prog, ctx);
    if (v.k == Val::Error) return 3;
    if (v.k != Val::Bool) return 2;
    return v.b ? 1 : 0;
}

int pwaf_oracle_geoip_lookup(void *h, const uint8_t ip[16], int v6, uint32_t *asn, uint8_t country[2]) {
    const Engine &e = *(const Engine *)h;
    e.geo_lookup(ip, v6 != 0, *asn, country);
    return 0;
}

// 1 match, 0 no match, -1 pattern does not compile (message in errbuf)
int pwaf_oracle_regex_is_match(const char *pattern, const uint8_t *hay, size_t len, char *errbuf, size_t errlen) {
    Regex re;
    std::string err;
    if (!Regex::compile(pattern, re, err)) {
        seterr(errbuf, errlen, err);
        return -1;
    }
    return re.is_match(std::string_view((const char *)hay, len)) ? 1 : 0;
}

// 1 contains, 0 not, -1 parse error
int pwaf_oracle_ipnet_contains(const char *net, const uint8_t ip[16], int v6) {
    IpNet n;
    std::string err;
    if (!parse_ipnet(net, n, err)) return -1;
    IpAddr a;
    a.v6 = v6 != 0;
    memcpy(a.b, ip, a.v6 ? 16 : 4);
    return n.contains(a) ? 1 : 0;
}

// 4 = parsed as v4, 6 = v6, 0 = invalid; writes the 16-byte form
int pwaf_oracle_parse_ip(const char *s, uint8_t out[16]) {
    memset(out, 0, 16);
    if (parse_ipv4(s, out)) return 4;
    if (parse_ipv6(s, out)) return 6;
    return 0;
}

// ---- field derivation restatements ---------------------------------------------------------------
size_t pwaf_oracle_derive_path(const uint8_t *p, size_t len) {
    // uri.path().trim_end_matches('/') (http_utils.rs:114-116)
    while (len > 0 && p[len - 1] == '/') len--;
    return len;
}

static bool header_to_str_ok(const uint8_t *p, size_t len) {
    // http::HeaderValue::to_str: every byte must be visible ASCII (32..=126) or '\t'
    for (size_t k = 0; k < len; k++) {
        uint8_t c = p[k];
        if (!(c == '\t' || (c >= 32 && c < 127))) return false;
    }
    return true;
}
static void trim_ascii(const uint8_t *p, size_t len, size_t *start, size_t *outlen) {
    // str::trim on a to_str()-validated header: only ' ' and '\t' can occur as whitespace
    size_t b = 0, e = len;
    while (b < e && (p[b] == ' ' || p[b] == '\t')) b++;
    while (e > b && (p[e - 1] == ' ' || p[e - 1] == '\t')) e--;
    *start = b;
    *outlen = e - b;
}

void pwaf_oracle_derive_user_agent(const uint8_t *hdr, size_t len, int present, size_t *out_start, size_t *out_len) {
    // http_listener.rs:159-165
    *out_start = 0;
    *out_len = 0;
    if (!present) return;
    if (!header_to_str_ok(hdr, len)) return;  // to_str().unwrap_or_default()
    size_t s, l;
    trim_ascii(hdr, len, &s, &l);
    if (l > 256) return;  // heapless::String::<256>::from_str(..).unwrap_or_default()
    *out_start = s;
    *out_len = l;
}

void pwaf_oracle_derive_host(const uint8_t *uri_host, size_t uri_host_len, int uri_host_present, const uint8_t *host_hdr,
                             size_t host_hdr_len, int host_hdr_present, int *from_header, size_t *out_start, size_t *out_len) {
    // http_listener.rs:284-296
    *from_header = 0;
    *out_start = 0;
    *out_len = 0;
    if (uri_host_present) {
        // uri.host() is ASCII; str::trim
        size_t s, l;
        size_t b = 0, e = uri_host_len;
        auto ws = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); };
        while (b < e && ws(uri_host[b])) b++;
        while (e > b && ws(uri_host[e - 1])) e--;
        s = b;
        l = e - b;
        if (l > 256) return;
        *out_start = s;
        *out_len = l;
        return;
    }
    if (host_hdr_present) {
        *from_header = 1;
        if (!header_to_str_ok(host_hdr, host_hdr_len)) return;
        size_t s, l;
        trim_ascii(host_hdr, host_hdr_len, &s, &l);
        if (l > 256) return;
        *out_start = s;
        *out_len = l;
    }
}

}  // extern "C"
