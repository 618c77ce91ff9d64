// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_expr.h header note: PARITY UNPINNED by the reference).
//
// Lexer + recursive-descent parser (CEL precedence, docs/rules.md:35-37 "a subset of CEL") and a
// tree-walking evaluator over dynamically typed values — deliberately the same shape as the
// reference's interpreter (pingoo/rules.rs:37-51 calls bel::Program::execute per rule per request).
#include "oracle_expr.h"

#include <cmath>
#include <cstring>
#include <deque>
#include <limits>

namespace oracle {

// ------------------------------------------------------------------------------------------------
// IP parsing (std::net::{Ipv4Addr,Ipv6Addr}::from_str and ipnetwork::IpNetwork::from_str)
// ------------------------------------------------------------------------------------------------
bool IpAddr::operator==(const IpAddr &o) const {
    if (v6 != o.v6) return false;
    return memcmp(b, o.b, v6 ? 16 : 4) == 0;
}

bool IpNet::contains(const IpAddr &ip) const {
    if (ip.v6 != addr.v6) return false;
    int bits = prefix;
    int nbytes = addr.v6 ? 16 : 4;
    for (int k = 0; k < nbytes && bits > 0; k++) {
        int take = bits >= 8 ? 8 : bits;
        uint8_t mask = (uint8_t)(0xFF << (8 - take));
        if ((ip.b[k] & mask) != (addr.b[k] & mask)) return false;
        bits -= take;
    }
    return true;
}

bool parse_ipv4(std::string_view s, uint8_t out[4]) {
    // strict dotted quad: 4 decimal parts 0..255, no leading zeros (Rust >= 1.59), nothing else
    size_t pos = 0;
    for (int part = 0; part < 4; part++) {
        if (pos >= s.size()) return false;
        size_t b = pos;
        unsigned v = 0;
        while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') {
            v = v * 10 + (unsigned)(s[pos] - '0');
            pos++;
            if (pos - b > 3) return false;
        }
        if (pos == b) return false;
        if (pos - b > 1 && s[b] == '0') return false;
        if (v > 255) return false;
        out[part] = (uint8_t)v;
        if (part < 3) {
            if (pos >= s.size() || s[pos] != '.') return false;
            pos++;
        }
    }
    return pos == s.size();
}

bool parse_ipv6(std::string_view s, uint8_t out[16]) {
    // RFC 4291 text forms: 8 groups, "::" compression once, optional trailing embedded IPv4.
    uint16_t head[8], tail[8];
    int nh = 0, nt = 0;
    bool compressed = false;
    size_t pos = 0;
    if (s.size() < 2) return false;
    if (s[0] == ':') {
        if (s[1] != ':') return false;
        compressed = true;
        pos = 2;
    }
    auto groups = [&](uint16_t *dst, int &n, int cap) -> bool {
        // parse groups separated by single ':' until end or "::"
        while (pos < s.size()) {
            // embedded ipv4?
            size_t e = pos;
            bool dot = false;
            while (e < s.size() && s[e] != ':') {
                if (s[e] == '.') dot = true;
                e++;
            }
            if (dot) {
                if (e != s.size()) return false;
                uint8_t v4[4];
                if (!parse_ipv4(s.substr(pos), v4)) return false;
                if (n + 2 > cap) return false;
                dst[n++] = (uint16_t)((v4[0] << 8) | v4[1]);
                dst[n++] = (uint16_t)((v4[2] << 8) | v4[3]);
                pos = s.size();
                return true;
            }
            if (e == pos || e - pos > 4) return false;
            unsigned v = 0;
            for (size_t k = pos; k < e; k++) {
                char c = s[k];
                int h = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1;
                if (h < 0) return false;
                v = v * 16 + (unsigned)h;
            }
            if (n + 1 > cap) return false;
            dst[n++] = (uint16_t)v;
            pos = e;
            if (pos == s.size()) return true;
            // s[pos] == ':'
            if (pos + 1 < s.size() && s[pos + 1] == ':') return true;  // leave at "::"
            pos++;
            if (pos == s.size()) return false;  // trailing single ':'
        }
        return true;
    };
    if (!compressed) {
        if (!groups(head, nh, 8)) return false;
        if (pos < s.size()) {
            // at "::"
            compressed = true;
            pos += 2;
            if (!groups(tail, nt, 8)) return false;
            if (pos != s.size()) return false;  // second "::"
        }
    } else {
        if (!groups(tail, nt, 8)) return false;
        if (pos != s.size()) return false;
    }
    if (compressed) {
        if (nh + nt > 7) return false;
    } else if (nh != 8) {
        return false;
    }
    uint16_t g[8] = {0};
    for (int k = 0; k < nh; k++) g[k] = head[k];
    for (int k = 0; k < nt; k++) g[8 - nt + k] = tail[k];
    for (int k = 0; k < 8; k++) {
        out[2 * k] = (uint8_t)(g[k] >> 8);
        out[2 * k + 1] = (uint8_t)(g[k] & 0xFF);
    }
    return true;
}

bool parse_ipnet(std::string_view s, IpNet &out, std::string &err) {
    size_t slash = s.find('/');
    std::string_view a = slash == std::string_view::npos ? s : s.substr(0, slash);
    IpNet n;
    if (parse_ipv4(a, n.addr.b)) {
        n.addr.v6 = false;
        n.prefix = 32;
    } else if (parse_ipv6(a, n.addr.b)) {
        n.addr.v6 = true;
        n.prefix = 128;
    } else {
        err = "invalid address: " + std::string(s);
        return false;
    }
    if (slash != std::string_view::npos) {
        std::string_view p = s.substr(slash + 1);
        uint8_t m[4];
        if (!n.addr.v6 && parse_ipv4(p, m)) {
            // dotted netmask: must be contiguous ones
            uint32_t mask = ((uint32_t)m[0] << 24) | ((uint32_t)m[1] << 16) | ((uint32_t)m[2] << 8) | m[3];
            int ones = 0;
            while (ones < 32 && (mask & (0x80000000u >> ones))) ones++;
            if (ones < 32 && (mask << ones) != 0) {
                err = "invalid prefix";
                return false;
            }
            n.prefix = (uint8_t)ones;
        } else {
            if (p.empty() || p.size() > 3) { err = "invalid prefix"; return false; }
            unsigned v = 0;
            for (char c : p) {
                if (c < '0' || c > '9') { err = "invalid prefix"; return false; }
                v = v * 10 + (unsigned)(c - '0');
            }
            if (v > (n.addr.v6 ? 128u : 32u)) { err = "invalid prefix"; return false; }
            n.prefix = (uint8_t)v;
        }
    }
    out = n;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Lexer
// ------------------------------------------------------------------------------------------------
namespace {

struct Tok {
    enum K { End, Ident, Int, Float, Str, Punct } k = End;
    std::string text;  // Ident / Punct / Str bytes
    uint64_t mag = 0;  // Int magnitude
    double f = 0;
    size_t pos = 0;
};

struct Lexer {
    std::string_view s;
    size_t pos = 0;
    std::string err;

    bool fail(const std::string &m, size_t at) {
        if (err.empty()) err = m + " at offset " + std::to_string(at);
        return false;
    }
    static bool id_start(char c) { return c == '_' || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
    static bool id_char(char c) { return id_start(c) || (c >= '0' && c <= '9'); }
    static int hexv(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    static void utf8(std::string &o, uint32_t cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }

    bool string_lit(Tok &t, bool raw) {
        char q = s[pos];
        size_t start = pos;
        if (pos + 2 < s.size() && s[pos + 1] == q && s[pos + 2] == q) return fail("triple-quoted strings are not supported", start);
        pos++;
        std::string out;
        for (;;) {
            if (pos >= s.size()) return fail("unterminated string literal", start);
            char c = s[pos];
            if (c == '\n' || c == '\r') return fail("newline in string literal", pos);
            if (c == q) { pos++; break; }
            if (c == '\\' && !raw) {
                pos++;
                if (pos >= s.size()) return fail("unterminated escape", pos);
                char e = s[pos++];
                switch (e) {
                    case '\\': out += '\\'; break;
                    case '"': out += '"'; break;
                    case '\'': out += '\''; break;
                    case '`': out += '`'; break;
                    case '?': out += '?'; break;
                    case 'a': out += '\a'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'v': out += '\v'; break;
                    case 'x': case 'X': case 'u': case 'U': {
                        int nd = (e == 'u') ? 4 : (e == 'U') ? 8 : 2;
                        uint32_t v = 0;
                        for (int k = 0; k < nd; k++) {
                            if (pos >= s.size() || hexv(s[pos]) < 0) return fail("invalid hex escape", pos);
                            v = v * 16 + (uint32_t)hexv(s[pos++]);
                        }
                        if (v > 0x10FFFF || (v >= 0xD800 && v <= 0xDFFF)) return fail("invalid code point", pos);
                        utf8(out, v);  // in a string literal \xHH denotes a code point (CEL spec)
                        break;
                    }
                    case '0': case '1': case '2': case '3': {
                        uint32_t v = (uint32_t)(e - '0');
                        for (int k = 0; k < 2; k++) {
                            if (pos >= s.size() || s[pos] < '0' || s[pos] > '7') return fail("invalid octal escape", pos);
                            v = v * 8 + (uint32_t)(s[pos++] - '0');
                        }
                        utf8(out, v);
                        break;
                    }
                    default: return fail("invalid escape sequence", pos - 1);
                }
            } else {
                out += c;
                pos++;
            }
        }
        t.k = Tok::Str;
        t.text = std::move(out);
        return true;
    }

    bool next(Tok &t) {
        while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r' || s[pos] == '\f')) pos++;
        if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/') {
            while (pos < s.size() && s[pos] != '\n') pos++;
            return next(t);
        }
        t = Tok();
        t.pos = pos;
        if (pos >= s.size()) { t.k = Tok::End; return true; }
        char c = s[pos];
        if ((c == 'r' || c == 'R') && pos + 1 < s.size() && (s[pos + 1] == '"' || s[pos + 1] == '\'')) {
            pos++;
            return string_lit(t, true);
        }
        if ((c == 'b' || c == 'B') && pos + 1 < s.size() && (s[pos + 1] == '"' || s[pos + 1] == '\'')) return fail("bytes literals are not supported", pos);
        if (id_start(c)) {
            size_t b = pos;
            while (pos < s.size() && id_char(s[pos])) pos++;
            t.k = Tok::Ident;
            t.text = std::string(s.substr(b, pos - b));
            return true;
        }
        if (c == '"' || c == '\'') return string_lit(t, false);
        bool dot_float = c == '.' && pos + 1 < s.size() && s[pos + 1] >= '0' && s[pos + 1] <= '9';
        if ((c >= '0' && c <= '9') || dot_float) {
            size_t b = pos;
            if (c == '0' && pos + 1 < s.size() && (s[pos + 1] == 'x' || s[pos + 1] == 'X')) {
                pos += 2;
                size_t d = pos;
                uint64_t v = 0;
                while (pos < s.size() && hexv(s[pos]) >= 0) {
                    if (v >> 60) return fail("integer literal out of range", b);
                    v = v * 16 + (uint64_t)hexv(s[pos]);
                    pos++;
                }
                if (pos == d) return fail("invalid hex literal", b);
                if (pos < s.size() && (s[pos] == 'u' || s[pos] == 'U')) return fail("unsigned integer literals are not supported", b);
                t.k = Tok::Int;
                t.mag = v;
                return true;
            }
            bool is_float = false;
            while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') pos++;
            if (pos < s.size() && s[pos] == '.' && pos + 1 < s.size() && s[pos + 1] >= '0' && s[pos + 1] <= '9') {
                is_float = true;
                pos++;
                while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') pos++;
            }
            if (pos < s.size() && (s[pos] == 'e' || s[pos] == 'E')) {
                size_t save = pos;
                pos++;
                if (pos < s.size() && (s[pos] == '+' || s[pos] == '-')) pos++;
                size_t d = pos;
                while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') pos++;
                if (pos == d) pos = save; else is_float = true;
            }
            std::string num(s.substr(b, pos - b));
            if (is_float) {
                t.k = Tok::Float;
                t.f = strtod(num.c_str(), nullptr);
                return true;
            }
            if (pos < s.size() && (s[pos] == 'u' || s[pos] == 'U')) return fail("unsigned integer literals are not supported", b);
            uint64_t v = 0;
            for (char ch : num) {
                uint64_t d = (uint64_t)(ch - '0');
                if (v > (std::numeric_limits<uint64_t>::max() - d) / 10) return fail("integer literal out of range", b);
                v = v * 10 + d;
            }
            t.k = Tok::Int;
            t.mag = v;
            return true;
        }
        static const char *two[] = {"||", "&&", "==", "!=", "<=", ">="};
        for (const char *op : two) {
            if (pos + 1 < s.size() && s[pos] == op[0] && s[pos + 1] == op[1]) {
                t.k = Tok::Punct;
                t.text = op;
                pos += 2;
                return true;
            }
        }
        if (strchr("!<>+-*/%?:.,()[]{}", c)) {
            t.k = Tok::Punct;
            t.text = std::string(1, c);
            pos++;
            return true;
        }
        return fail(std::string("unexpected character '") + c + "'", pos);
    }
};

// ------------------------------------------------------------------------------------------------
// Parser
// ------------------------------------------------------------------------------------------------
struct Parser {
    Lexer lx;
    Tok tok;
    std::string err;
    std::vector<std::string> *functions = nullptr;
    int depth = 0;

    bool fail(const std::string &m) {
        if (err.empty()) err = m + " at offset " + std::to_string(tok.pos);
        return false;
    }
    bool advance() {
        if (!lx.next(tok)) { err = lx.err; return false; }
        return true;
    }
    bool is_punct(const char *p) const { return tok.k == Tok::Punct && tok.text == p; }
    void fn(const char *name) { functions->push_back(name); }

    struct Depth {
        Parser &p;
        bool ok;
        explicit Depth(Parser &pp) : p(pp) { ok = ++p.depth <= 200; if (!ok) p.fail("expression nesting too deep"); }
        ~Depth() { p.depth--; }
    };

    NodeP mk(Node::K k) { auto n = std::make_unique<Node>(); n->k = k; return n; }
    NodeP bin(const char *op, NodeP l, NodeP r) {
        auto n = mk(Node::Bin);
        n->name = op;
        n->kids.push_back(std::move(l));
        n->kids.push_back(std::move(r));
        return n;
    }

    NodeP parse_expr() {
        Depth d(*this);
        if (!d.ok) return nullptr;
        NodeP c = parse_or();
        if (!c) return nullptr;
        if (is_punct("?")) {
            if (!advance()) return nullptr;
            NodeP a = parse_or();
            if (!a) return nullptr;
            if (!is_punct(":")) { fail("expected ':' in conditional"); return nullptr; }
            if (!advance()) return nullptr;
            NodeP b = parse_expr();
            if (!b) return nullptr;
            fn("_?_:_");
            auto n = mk(Node::Cond);
            n->kids.push_back(std::move(c));
            n->kids.push_back(std::move(a));
            n->kids.push_back(std::move(b));
            return n;
        }
        return c;
    }
    NodeP parse_or() {
        NodeP l = parse_and();
        while (l && is_punct("||")) {
            if (!advance()) return nullptr;
            NodeP r = parse_and();
            if (!r) return nullptr;
            fn("_||_");
            l = bin("||", std::move(l), std::move(r));
        }
        return l;
    }
    NodeP parse_and() {
        NodeP l = parse_rel();
        while (l && is_punct("&&")) {
            if (!advance()) return nullptr;
            NodeP r = parse_rel();
            if (!r) return nullptr;
            fn("_&&_");
            l = bin("&&", std::move(l), std::move(r));
        }
        return l;
    }
    NodeP parse_rel() {
        NodeP l = parse_add();
        for (;;) {
            if (!l) return nullptr;
            std::string op;
            if (tok.k == Tok::Punct && (tok.text == "==" || tok.text == "!=" || tok.text == "<" || tok.text == "<=" || tok.text == ">" || tok.text == ">=")) op = tok.text;
            else if (tok.k == Tok::Ident && tok.text == "in") op = "in";
            else break;
            if (!advance()) return nullptr;
            NodeP r = parse_add();
            if (!r) return nullptr;
            if (op == "in") fn("@in"); else fn(("_" + op + "_").c_str());
            l = bin(op.c_str(), std::move(l), std::move(r));
        }
        return l;
    }
    NodeP parse_add() {
        NodeP l = parse_mul();
        while (l && (is_punct("+") || is_punct("-"))) {
            std::string op = tok.text;
            if (!advance()) return nullptr;
            NodeP r = parse_mul();
            if (!r) return nullptr;
            fn(("_" + op + "_").c_str());
            l = bin(op.c_str(), std::move(l), std::move(r));
        }
        return l;
    }
    NodeP parse_mul() {
        NodeP l = parse_unary();
        while (l && (is_punct("*") || is_punct("/") || is_punct("%"))) {
            std::string op = tok.text;
            if (!advance()) return nullptr;
            NodeP r = parse_unary();
            if (!r) return nullptr;
            fn(("_" + op + "_").c_str());
            l = bin(op.c_str(), std::move(l), std::move(r));
        }
        return l;
    }
    NodeP parse_unary() {
        Depth d(*this);
        if (!d.ok) return nullptr;
        if (is_punct("!")) {
            int n = 0;
            while (is_punct("!")) { n++; if (!advance()) return nullptr; }
            NodeP m = parse_member();
            if (!m) return nullptr;
            for (int k = 0; k < n; k++) {
                fn("!_");
                auto u = mk(Node::Not);
                u->kids.push_back(std::move(m));
                m = std::move(u);
            }
            return m;
        }
        if (is_punct("-")) {
            int n = 0;
            while (is_punct("-")) { n++; if (!advance()) return nullptr; }
            // INT64_MIN literal
            if (n >= 1 && tok.k == Tok::Int && tok.mag == (uint64_t)1 << 63) {
                auto lit = mk(Node::Lit);
                lit->lit = Val::integer(std::numeric_limits<int64_t>::min());
                if (!advance()) return nullptr;
                NodeP m = parse_member_tail(std::move(lit));
                if (!m) return nullptr;
                for (int k = 0; k < n - 1; k++) {
                    fn("-_");
                    auto u = mk(Node::Neg);
                    u->kids.push_back(std::move(m));
                    m = std::move(u);
                }
                return m;
            }
            NodeP m = parse_member();
            if (!m) return nullptr;
            for (int k = 0; k < n; k++) {
                fn("-_");
                auto u = mk(Node::Neg);
                u->kids.push_back(std::move(m));
                m = std::move(u);
            }
            return m;
        }
        return parse_member();
    }
    bool parse_args(std::vector<NodeP> &out) {
        // after '(' consumed
        if (is_punct(")")) return advance();
        for (;;) {
            NodeP a = parse_expr();
            if (!a) return false;
            out.push_back(std::move(a));
            if (is_punct(",")) { if (!advance()) return false; continue; }
            if (is_punct(")")) return advance();
            return fail("expected ',' or ')' in argument list");
        }
    }
    void prepare_call(Node &call) {
        // pre-compile literal regex patterns so evaluation is allocation-free and thread-safe
        if (call.name == "matches" && call.has_receiver && call.kids.size() == 2 && call.kids[1]->k == Node::Lit && call.kids[1]->lit.k == Val::String) {
            call.regex_tried = true;
            Regex re;
            std::string e;
            if (Regex::compile(call.kids[1]->lit.s, re, e)) call.regex_cache = std::make_shared<Regex>(re);
            else call.regex_err = e;
        }
    }
    NodeP parse_member() {
        NodeP p = parse_primary();
        if (!p) return nullptr;
        return parse_member_tail(std::move(p));
    }
    NodeP parse_member_tail(NodeP p) {
        for (;;) {
            if (is_punct(".")) {
                if (!advance()) return nullptr;
                if (tok.k != Tok::Ident) { fail("expected identifier after '.'"); return nullptr; }
                std::string name = tok.text;
                if (!advance()) return nullptr;
                if (is_punct("(")) {
                    if (!advance()) return nullptr;
                    auto c = mk(Node::Call);
                    c->name = name;
                    c->has_receiver = true;
                    c->kids.push_back(std::move(p));
                    if (!parse_args(c->kids)) return nullptr;
                    fn(name.c_str());
                    prepare_call(*c);
                    p = std::move(c);
                } else {
                    auto m = mk(Node::Member);
                    m->name = name;
                    m->kids.push_back(std::move(p));
                    p = std::move(m);
                }
            } else if (is_punct("[")) {
                if (!advance()) return nullptr;
                NodeP idx = parse_expr();
                if (!idx) return nullptr;
                if (!is_punct("]")) { fail("expected ']'"); return nullptr; }
                if (!advance()) return nullptr;
                fn("_[_]");
                auto m = mk(Node::Index);
                m->kids.push_back(std::move(p));
                m->kids.push_back(std::move(idx));
                p = std::move(m);
            } else {
                return p;
            }
        }
    }
    NodeP parse_primary() {
        Depth d(*this);
        if (!d.ok) return nullptr;
        if (tok.k == Tok::Int) {
            if (tok.mag > (uint64_t)std::numeric_limits<int64_t>::max()) { fail("integer literal out of range"); return nullptr; }
            auto n = mk(Node::Lit);
            n->lit = Val::integer((int64_t)tok.mag);
            if (!advance()) return nullptr;
            return n;
        }
        if (tok.k == Tok::Float) {
            auto n = mk(Node::Lit);
            n->lit = Val::flt(tok.f);
            if (!advance()) return nullptr;
            return n;
        }
        if (tok.k == Tok::Str) {
            auto n = mk(Node::Lit);
            n->lit_store = tok.text;
            n->lit = Val::str(n->lit_store);
            if (!advance()) return nullptr;
            return n;
        }
        if (tok.k == Tok::Ident) {
            std::string name = tok.text;
            if (name == "in") { fail("unexpected 'in'"); return nullptr; }
            if (!advance()) return nullptr;
            if (name == "true" || name == "false") {
                auto n = mk(Node::Lit);
                n->lit = Val::boolean(name == "true");
                return n;
            }
            if (name == "null") {
                auto n = mk(Node::Lit);
                n->lit.k = Val::Null;
                return n;
            }
            if (is_punct("(")) {
                if (!advance()) return nullptr;
                auto c = mk(Node::Call);
                c->name = name;
                c->has_receiver = false;
                if (!parse_args(c->kids)) return nullptr;
                fn(name.c_str());
                return c;
            }
            if (is_punct("{")) { fail("message construction is not supported"); return nullptr; }
            auto n = mk(Node::Ident);
            n->name = name;
            return n;
        }
        if (is_punct("(")) {
            if (!advance()) return nullptr;
            NodeP e = parse_expr();
            if (!e) return nullptr;
            if (!is_punct(")")) { fail("expected ')'"); return nullptr; }
            if (!advance()) return nullptr;
            return e;
        }
        if (is_punct("[")) {
            if (!advance()) return nullptr;
            auto l = mk(Node::ListLit);
            while (!is_punct("]")) {
                NodeP e = parse_expr();
                if (!e) return nullptr;
                l->kids.push_back(std::move(e));
                if (is_punct(",")) { if (!advance()) return nullptr; continue; }
                if (!is_punct("]")) { fail("expected ',' or ']' in list literal"); return nullptr; }
            }
            if (!advance()) return nullptr;
            // constant-fold all-literal lists (immutable => thread-safe, allocation-free evaluation)
            bool all_lit = true;
            for (auto &k : l->kids) if (k->k != Node::Lit) all_lit = false;
            if (all_lit) {
                l->scratch_list = std::make_shared<ListVal>();
                for (auto &k : l->kids) l->scratch_list->items.push_back(k->lit);  // string views point into kid->lit_store (kept alive)
            }
            return l;
        }
        if (is_punct("{")) {
            if (!advance()) return nullptr;
            auto m = mk(Node::MapLit);
            while (!is_punct("}")) {
                NodeP k = parse_expr();
                if (!k) return nullptr;
                if (!is_punct(":")) { fail("expected ':' in map literal"); return nullptr; }
                if (!advance()) return nullptr;
                NodeP v = parse_expr();
                if (!v) return nullptr;
                m->kids.push_back(std::move(k));
                m->kids.push_back(std::move(v));
                if (is_punct(",")) { if (!advance()) return nullptr; continue; }
                if (!is_punct("}")) { fail("expected ',' or '}' in map literal"); return nullptr; }
            }
            if (!advance()) return nullptr;
            return m;
        }
        if (tok.k == Tok::End) { fail("unexpected end of expression"); return nullptr; }
        fail("unexpected token '" + tok.text + "'");
        return nullptr;
    }
};

}  // namespace

bool compile(std::string_view src, Program &out, std::string &err) {
    Parser p;
    p.lx.s = src;
    p.functions = &out.functions;
    if (!p.advance()) { err = p.err; return false; }
    NodeP root = p.parse_expr();
    if (!root) { err = p.err.empty() ? "syntax error" : p.err; return false; }
    if (p.tok.k != Tok::End) {
        p.fail("unexpected trailing input");
        err = p.err;
        return false;
    }
    out.root = std::move(root);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Evaluator
// ------------------------------------------------------------------------------------------------
namespace {

// Scratch storage for values created during one execution (string concatenation, list/map literals with
// computed items). Created lazily: the common predicates allocate nothing.
template <class T>
struct Lazy {
    std::unique_ptr<std::deque<T>> d;
    template <class... A>
    T &emplace_back(A &&...a) {
        if (!d) d = std::make_unique<std::deque<T>>();
        d->emplace_back(std::forward<A>(a)...);
        return d->back();
    }
    T &back() { return d->back(); }
};
struct Exec {
    const Context &ctx;
    Lazy<std::string> strs;
    Lazy<ListVal> lists;
    Lazy<MapVal> maps;
};

static bool val_eq(const Val &a, const Val &b);

static bool list_contains(const ListVal &l, const Val &x) {
    for (const Val &it : l.items) {
        if (it.k == Val::Net && x.k == Val::Ip) {
            if (it.net.contains(x.ip)) return true;  // D12: Array<Ip>.contains == CIDR containment (lists.rs:14,102-108)
        } else if (val_eq(it, x)) {
            return true;
        }
    }
    return false;
}

static bool val_eq(const Val &a, const Val &b) {
    if (a.k == Val::Int && b.k == Val::Float) return (double)a.i == b.f;
    if (a.k == Val::Float && b.k == Val::Int) return a.f == (double)b.i;
    if (a.k != b.k) return false;  // D4: cross-type equality is false, not an error
    switch (a.k) {
        case Val::Null: return true;
        case Val::Bool: return a.b == b.b;
        case Val::Int: return a.i == b.i;
        case Val::Float: return a.f == b.f;
        case Val::String: return a.s == b.s;
        case Val::Ip: return a.ip == b.ip;
        case Val::Net: return a.net.prefix == b.net.prefix && a.net.addr == b.net.addr;
        case Val::List: {
            if (a.list->items.size() != b.list->items.size()) return false;
            for (size_t k = 0; k < a.list->items.size(); k++) if (!val_eq(a.list->items[k], b.list->items[k])) return false;
            return true;
        }
        case Val::Map: {
            if (a.map->items.size() != b.map->items.size()) return false;
            auto ia = a.map->items.begin();
            auto ib = b.map->items.begin();
            for (; ia != a.map->items.end(); ++ia, ++ib) {
                if (ia->first != ib->first || !val_eq(ia->second, ib->second)) return false;
            }
            return true;
        }
        default: return false;
    }
}

// returns -1/0/1, or 2 when not comparable
static int val_cmp(const Val &a, const Val &b) {
    auto sgn = [](auto x, auto y) { return x < y ? -1 : (x > y ? 1 : 0); };
    if (a.k == Val::Int && b.k == Val::Int) return sgn(a.i, b.i);
    if (a.k == Val::Float && b.k == Val::Float) { if (std::isnan(a.f) || std::isnan(b.f)) return 2; return sgn(a.f, b.f); }
    if (a.k == Val::Int && b.k == Val::Float) { if (std::isnan(b.f)) return 2; return sgn((double)a.i, b.f); }
    if (a.k == Val::Float && b.k == Val::Int) { if (std::isnan(a.f)) return 2; return sgn(a.f, (double)b.i); }
    if (a.k == Val::String && b.k == Val::String) { int c = a.s.compare(b.s); return c < 0 ? -1 : (c > 0 ? 1 : 0); }
    return 2;
}

static Val eval(const Node &n, Exec &ex);

static Val eval_call(const Node &n, Exec &ex) {
    if (!n.has_receiver) return Val::err("undeclared function");
    Val recv = eval(*n.kids[0], ex);
    if (recv.k == Val::Error) return recv;
    struct Args {  // calls in this language take 0..1 arguments; keep the common case off the heap
        Val inl[2];
        std::vector<Val> more;
        size_t n = 0;
        void push_back(const Val &v) { if (n < 2) inl[n] = v; else more.push_back(v); n++; }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        const Val &operator[](size_t k) const { return k < 2 ? inl[k] : more[k - 2]; }
    } args;
    for (size_t k = 1; k < n.kids.size(); k++) {
        Val a = eval(*n.kids[k], ex);
        if (a.k == Val::Error) return a;
        args.push_back(a);
    }
    const std::string &f = n.name;
    if (f == "contains") {
        if (args.size() != 1) return Val::err("contains: expected 1 argument");
        if (recv.k == Val::String) {
            if (args[0].k != Val::String) return Val::err("contains: argument must be a String");  // D11 strict typing
            return Val::boolean(recv.s.find(args[0].s) != std::string_view::npos);
        }
        if (recv.k == Val::List) return Val::boolean(list_contains(*recv.list, args[0]));
        if (recv.k == Val::Map) {
            if (args[0].k != Val::String) return Val::err("contains: map keys are Strings");
            return Val::boolean(recv.map->items.find(args[0].s) != recv.map->items.end());
        }
        return Val::err("contains: unsupported receiver type");
    }
    if (f == "starts_with" || f == "ends_with") {
        if (args.size() != 1) return Val::err("expected 1 argument");
        if (recv.k != Val::String || args[0].k != Val::String) return Val::err("starts_with/ends_with: String operands required");
        std::string_view h = recv.s, p = args[0].s;
        if (p.size() > h.size()) return Val::boolean(false);
        if (f == "starts_with") return Val::boolean(h.substr(0, p.size()) == p);
        return Val::boolean(h.substr(h.size() - p.size()) == p);
    }
    if (f == "length") {
        if (!args.empty()) return Val::err("length: expected no arguments");
        if (recv.k == Val::String) return Val::integer((int64_t)recv.s.size());  // D13: bytes (fields are ASCII)
        if (recv.k == Val::List) return Val::integer((int64_t)recv.list->items.size());
        if (recv.k == Val::Map) return Val::integer((int64_t)recv.map->items.size());
        return Val::err("length: unsupported receiver type");
    }
    if (f == "matches") {
        if (args.size() != 1) return Val::err("matches: expected 1 argument");
        if (recv.k != Val::String || args[0].k != Val::String) return Val::err("matches: String operands required");
        if (n.regex_tried) {
            if (!n.regex_cache) return Val::err("matches: invalid regex");
            return Val::boolean(n.regex_cache->is_match(recv.s));
        }
        Regex re;
        std::string e;
        if (!Regex::compile(args[0].s, re, e)) return Val::err("matches: invalid regex");
        return Val::boolean(re.is_match(recv.s));
    }
    return Val::err("undeclared function");
}

static Val eval_bin(const Node &n, Exec &ex) {
    const std::string &op = n.name;
    if (op == "||" || op == "&&") {
        // D6: left-to-right, short-circuit, errors propagate from whatever is evaluated, operands must be Bool
        Val l = eval(*n.kids[0], ex);
        if (l.k == Val::Error) return l;
        if (l.k != Val::Bool) return Val::err("logical operator: Bool operands required");
        if (op == "||" && l.b) return l;
        if (op == "&&" && !l.b) return l;
        Val r = eval(*n.kids[1], ex);
        if (r.k == Val::Error) return r;
        if (r.k != Val::Bool) return Val::err("logical operator: Bool operands required");
        return r;
    }
    Val l = eval(*n.kids[0], ex);
    if (l.k == Val::Error) return l;
    Val r = eval(*n.kids[1], ex);
    if (r.k == Val::Error) return r;
    if (op == "==") return Val::boolean(val_eq(l, r));
    if (op == "!=") return Val::boolean(!val_eq(l, r));
    if (op == "<" || op == "<=" || op == ">" || op == ">=") {
        int c = val_cmp(l, r);
        if (c == 2) return Val::err("values are not comparable");
        if (op == "<") return Val::boolean(c < 0);
        if (op == "<=") return Val::boolean(c <= 0);
        if (op == ">") return Val::boolean(c > 0);
        return Val::boolean(c >= 0);
    }
    if (op == "in") {
        if (r.k == Val::List) return Val::boolean(list_contains(*r.list, l));
        if (r.k == Val::Map) {
            if (l.k != Val::String) return Val::err("in: map keys are Strings");
            return Val::boolean(r.map->items.find(l.s) != r.map->items.end());
        }
        return Val::err("in: right operand must be a List or Map");
    }
    // arithmetic
    if (l.k == Val::Int && r.k == Val::Int) {
        int64_t out;
        if (op == "+") { if (__builtin_add_overflow(l.i, r.i, &out)) return Val::err("integer overflow"); return Val::integer(out); }
        if (op == "-") { if (__builtin_sub_overflow(l.i, r.i, &out)) return Val::err("integer overflow"); return Val::integer(out); }
        if (op == "*") { if (__builtin_mul_overflow(l.i, r.i, &out)) return Val::err("integer overflow"); return Val::integer(out); }
        if (op == "/") {
            if (r.i == 0) return Val::err("division by zero");
            if (l.i == std::numeric_limits<int64_t>::min() && r.i == -1) return Val::err("integer overflow");
            return Val::integer(l.i / r.i);
        }
        if (op == "%") {
            if (r.i == 0) return Val::err("modulo by zero");
            if (l.i == std::numeric_limits<int64_t>::min() && r.i == -1) return Val::integer(0);
            return Val::integer(l.i % r.i);
        }
    }
    if ((l.k == Val::Float || l.k == Val::Int) && (r.k == Val::Float || r.k == Val::Int) && (l.k == Val::Float || r.k == Val::Float)) {
        double a = l.k == Val::Float ? l.f : (double)l.i, b = r.k == Val::Float ? r.f : (double)r.i;
        if (op == "+") return Val::flt(a + b);
        if (op == "-") return Val::flt(a - b);
        if (op == "*") return Val::flt(a * b);
        if (op == "/") return Val::flt(a / b);
        return Val::err("unsupported operand types");
    }
    if (op == "+" && l.k == Val::String && r.k == Val::String) {
        ex.strs.emplace_back(std::string(l.s) + std::string(r.s));
        return Val::str(ex.strs.back());
    }
    if (op == "+" && l.k == Val::List && r.k == Val::List) {
        ex.lists.emplace_back();
        ListVal &o = ex.lists.back();
        o.items = l.list->items;
        o.items.insert(o.items.end(), r.list->items.begin(), r.list->items.end());
        Val v;
        v.k = Val::List;
        v.list = &o;
        return v;
    }
    return Val::err("unsupported operand types");
}

static Val eval(const Node &n, Exec &ex) {
    switch (n.k) {
        case Node::Lit: return n.lit;
        case Node::Ident: {
            auto it = ex.ctx.vars.find(n.name);
            if (it == ex.ctx.vars.end()) return Val::err("undeclared reference");
            return it->second;
        }
        case Node::Member: {
            Val o = eval(*n.kids[0], ex);
            if (o.k == Val::Error) return o;
            if (o.k != Val::Map) return Val::err("member access on a non-map value");
            auto it = o.map->items.find(n.name);
            if (it == o.map->items.end()) return Val::err("no such key");
            return it->second;
        }
        case Node::Index: {
            Val o = eval(*n.kids[0], ex);
            if (o.k == Val::Error) return o;
            Val i = eval(*n.kids[1], ex);
            if (i.k == Val::Error) return i;
            if (o.k == Val::Map) {
                if (i.k != Val::String) return Val::err("map keys are Strings");
                auto it = o.map->items.find(i.s);
                if (it == o.map->items.end()) return Val::err("no such key");
                return it->second;
            }
            if (o.k == Val::List) {
                if (i.k != Val::Int) return Val::err("list index must be an Int");
                if (i.i < 0 || (uint64_t)i.i >= o.list->items.size()) return Val::err("index out of range");
                return o.list->items[(size_t)i.i];
            }
            return Val::err("index on a non-indexable value");
        }
        case Node::Call: return eval_call(n, ex);
        case Node::ListLit: {
            Val v;
            v.k = Val::List;
            if (n.scratch_list) { v.list = n.scratch_list.get(); return v; }
            ex.lists.emplace_back();
            ListVal &o = ex.lists.back();
            for (auto &k : n.kids) {
                Val it = eval(*k, ex);
                if (it.k == Val::Error) return it;
                o.items.push_back(it);
            }
            v.list = &o;
            return v;
        }
        case Node::MapLit: {
            ex.maps.emplace_back();
            MapVal &o = ex.maps.back();
            for (size_t k = 0; k + 1 < n.kids.size(); k += 2) {
                Val key = eval(*n.kids[k], ex);
                if (key.k == Val::Error) return key;
                if (key.k != Val::String) return Val::err("map keys are Strings");
                Val val = eval(*n.kids[k + 1], ex);
                if (val.k == Val::Error) return val;
                o.items[std::string(key.s)] = val;
            }
            Val v;
            v.k = Val::Map;
            v.map = &o;
            return v;
        }
        case Node::Not: {
            Val x = eval(*n.kids[0], ex);
            if (x.k == Val::Error) return x;
            if (x.k != Val::Bool) return Val::err("'!' requires a Bool");
            return Val::boolean(!x.b);
        }
        case Node::Neg: {
            Val x = eval(*n.kids[0], ex);
            if (x.k == Val::Error) return x;
            if (x.k == Val::Int) {
                if (x.i == std::numeric_limits<int64_t>::min()) return Val::err("integer overflow");
                return Val::integer(-x.i);
            }
            if (x.k == Val::Float) return Val::flt(-x.f);
            return Val::err("'-' requires a number");
        }
        case Node::Bin: return eval_bin(n, ex);
        case Node::Cond: {
            Val c = eval(*n.kids[0], ex);
            if (c.k == Val::Error) return c;
            if (c.k != Val::Bool) return Val::err("conditional requires a Bool");
            return eval(*n.kids[c.b ? 1 : 2], ex);
        }
    }
    return Val::err("internal");
}

}  // namespace

Val execute(const Program &p, const Context &ctx) {
    Exec ex{ctx, {}, {}, {}};  // no allocation until a value needs scratch storage
    Val v = eval(*p.root, ex);
    // values backed by Exec storage die here; only scalars / context-backed values may escape.
    if (v.k == Val::String || v.k == Val::List || v.k == Val::Map) {
        Val o;
        o.k = Val::Null;  // a non-Bool result never matches (pingoo/rules.rs:47); identity is irrelevant
        return o;
    }
    return v;
}

}  // namespace oracle
