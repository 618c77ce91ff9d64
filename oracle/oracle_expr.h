// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product
// path (pingoo_amd/, libpwaf.so). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it, and only as the checker.
//
// oracle_expr.h — CPU restatement of the expression language the reference evaluates per request.
//
// The reference delegates to `bel 0.11.0` (git dependency pingooio/stdx-rs@70c3c14, Cargo.lock:141-154),
// whose source is NOT under /root/reference and cannot be fetched or built here (no network, no Rust
// toolchain: SURVEY.md F3/F4). The reference has zero tests for this path (F5), so:
//
//      *** PARITY UNPINNED by the reference: this file restates the language from ***
//      *** docs/rules.md:35-76 and the reference's call sites, not from bel source. ***
//
// What IS pinned, and where each piece comes from:
//   - variable surface http_request.{host,url,path,method,user_agent}, client.{ip,remote_port,asn,
//     country}, lists[..]                         pingoo/rules.rs:16-34, http_listener.rs:239-249
//   - types Bool String Int Float Ip Regex Array Map; functions contains length starts_with
//     ends_with                                   docs/rules.md:39-76
//   - operators seen in the reference's own examples: == || && ! . [] list literals, method calls
//                                                 assets/pingoo.yml:15, docs/rules.md:20,110,
//                                                 docs/configuration.md:68, docs/getting_started.md:38,48
//   - execution error  => rule does not match; non-Bool result => does not match
//                                                 pingoo/rules.rs:37-51
//   - the `in` operator exists in the language (compiles) but validate_expression rejects it
//                                                 rules/rules.rs:65-71
// Everything else (precedence = CEL's, left-to-right short-circuit with error propagation, strict
// typing, `matches` as the regex entry point, CIDR containment for Array<Ip>.contains) is a documented
// decision: DESIGN.md §3 lists every one (D1..D16).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "oracle_regex.h"

namespace oracle {

struct IpAddr {
    uint8_t b[16] = {0};
    bool v6 = false;
    bool operator==(const IpAddr &o) const;
};
struct IpNet {
    IpAddr addr;
    uint8_t prefix = 0;
    bool contains(const IpAddr &ip) const;  // ipnetwork semantics: family must match, mask compare
};
bool parse_ipv4(std::string_view s, uint8_t out[4]);
bool parse_ipv6(std::string_view s, uint8_t out[16]);
// ipnetwork::IpNetwork::from_str: "addr", "addr/len", v4 also "addr/dotted-mask"
bool parse_ipnet(std::string_view s, IpNet &out, std::string &err);

struct ListVal;
struct MapVal;

struct Val {
    enum K : uint8_t { Error, Null, Bool, Int, Float, String, Ip, Net, List, Map } k = Error;
    bool b = false;
    int64_t i = 0;
    double f = 0;
    std::string_view s;  // String (views into request buffers / AST literals / owned pool)
    IpAddr ip;
    IpNet net;
    const ListVal *list = nullptr;
    const MapVal *map = nullptr;
    const char *emsg = "";

    static Val err(const char *m) { Val v; v.k = Error; v.emsg = m; return v; }
    static Val boolean(bool x) { Val v; v.k = Bool; v.b = x; return v; }
    static Val integer(int64_t x) { Val v; v.k = Int; v.i = x; return v; }
    static Val flt(double x) { Val v; v.k = Float; v.f = x; return v; }
    static Val str(std::string_view x) { Val v; v.k = String; v.s = x; return v; }
};
struct ListVal {
    std::vector<Val> items;
    std::vector<std::string> owned;  // backing store for String items
};
struct MapVal {
    std::map<std::string, Val, std::less<>> items;
};

struct Node;
using NodeP = std::unique_ptr<Node>;
struct Node {
    enum K {
        Lit, Ident, Member, Index, Call, ListLit, MapLit, Not, Neg, Bin, Cond
    } k = Lit;
    Val lit;
    std::string name;  // Ident / Member field / Call function / Bin operator
    std::vector<NodeP> kids;  // Member: [obj]; Index: [obj, idx]; Call: [receiver?, args...]; Bin: [l, r]
    bool has_receiver = false;
    std::string lit_store;                      // owns Lit string bytes
    std::shared_ptr<ListVal> scratch_list;      // ListLit evaluation scratch
    std::shared_ptr<MapVal> scratch_map;
    mutable std::shared_ptr<Regex> regex_cache;  // `matches` with a literal pattern
    mutable bool regex_tried = false;
    mutable std::string regex_err;
};

struct Program {
    NodeP root;
    std::vector<std::string> functions;  // references().functions(): operators appear as "@in", "_==_", ...
};

// bel::Program::compile. Returns false + err on a syntax error.
bool compile(std::string_view src, Program &out, std::string &err);

struct Context {
    std::map<std::string, Val, std::less<>> vars;
};
// bel::Program::execute
Val execute(const Program &p, const Context &ctx);

}  // namespace oracle
