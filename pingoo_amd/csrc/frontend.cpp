// frontend.cpp — tokenizer + Pratt (precedence-climbing) parser for the rule expression language.
// Grammar / literal forms: DESIGN.md §3.1-3.2. Written against the CEL language definition that the
// reference's docs name as the base language (docs/rules.md:35-37); `bel` itself is un-vendored.
#include "frontend.h"

#include <cstdlib>
#include <cstring>
#include <limits>

namespace pwaf {

namespace {

enum TokKind : uint8_t { T_END, T_IDENT, T_INT, T_FLOAT, T_STR, T_OP };

struct Token {
    TokKind kind = T_END;
    std::string text;
    uint64_t mag = 0;
    double f = 0;
    uint32_t pos = 0;
};

class Scanner {
public:
    explicit Scanner(const std::string &s) : src(s) {}
    std::string error;

    bool next(Token &t) {
        skip_space();
        t = Token();
        t.pos = (uint32_t)at;
        if (at >= src.size()) return true;
        unsigned char c = (unsigned char)src[at];
        if ((c == 'r' || c == 'R') && quote_at(at + 1)) { at++; return quoted(t, true); }
        if ((c == 'b' || c == 'B') && quote_at(at + 1)) return bad("bytes literals are not supported");
        if (c == '_' || isalpha(c)) {
            size_t b = at;
            while (at < src.size() && (src[at] == '_' || isalnum((unsigned char)src[at]))) at++;
            t.kind = T_IDENT;
            t.text = src.substr(b, at - b);
            return true;
        }
        if (c == '"' || c == '\'') return quoted(t, false);
        if (isdigit(c) || (c == '.' && at + 1 < src.size() && isdigit((unsigned char)src[at + 1]))) return number(t);
        static const char *const ops2[] = {"||", "&&", "==", "!=", "<=", ">="};
        for (const char *o : ops2) {
            if (src.compare(at, 2, o) == 0) {
                t.kind = T_OP;
                t.text = o;
                at += 2;
                return true;
            }
        }
        if (strchr("!<>+-*/%?:.,()[]{}", c)) {
            t.kind = T_OP;
            t.text.assign(1, (char)c);
            at++;
            return true;
        }
        return bad(std::string("unexpected character '") + (char)c + "'");
    }

private:
    const std::string &src;
    size_t at = 0;

    bool bad(const std::string &m) {
        if (error.empty()) error = m + " at offset " + std::to_string(at);
        return false;
    }
    bool quote_at(size_t p) const { return p < src.size() && (src[p] == '"' || src[p] == '\''); }
    void skip_space() {
        for (;;) {
            while (at < src.size() && strchr(" \t\n\r\f", src[at])) at++;
            if (at + 1 < src.size() && src[at] == '/' && src[at + 1] == '/') {
                while (at < src.size() && src[at] != '\n') at++;
                continue;
            }
            return;
        }
    }
    static int hexval(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    static void put_utf8(std::string &o, uint32_t cp) {
        if (cp < 0x80) { o.push_back((char)cp); return; }
        if (cp < 0x800) { o.push_back((char)(0xC0 | cp >> 6)); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | cp >> 12)); o.push_back((char)(0x80 | (cp >> 6 & 0x3F))); }
        else { o.push_back((char)(0xF0 | cp >> 18)); o.push_back((char)(0x80 | (cp >> 12 & 0x3F))); o.push_back((char)(0x80 | (cp >> 6 & 0x3F))); }
        o.push_back((char)(0x80 | (cp & 0x3F)));
    }
    bool quoted(Token &t, bool raw) {
        char q = src[at];
        if (at + 2 < src.size() && src[at + 1] == q && src[at + 2] == q) return bad("triple-quoted strings are not supported");
        size_t open = at++;
        std::string out;
        while (true) {
            if (at >= src.size()) { at = open; return bad("unterminated string literal"); }
            char c = src[at];
            if (c == '\n' || c == '\r') return bad("newline in string literal");
            at++;
            if (c == q) break;
            if (c != '\\' || raw) { out.push_back(c); continue; }
            if (at >= src.size()) return bad("unterminated escape");
            char e = src[at++];
            const char *simple = "\\\\\"\"''``??a\ab\bf\fn\nr\rt\tv\v";
            bool done = false;
            for (const char *s = simple; *s; s += 2)
                if (*s == e) { out.push_back(s[1]); done = true; break; }
            if (done) continue;
            int digits = 0;
            if (e == 'x' || e == 'X') digits = 2;
            else if (e == 'u') digits = 4;
            else if (e == 'U') digits = 8;
            if (digits) {
                uint32_t v = 0;
                for (int k = 0; k < digits; k++) {
                    if (at >= src.size() || hexval(src[at]) < 0) return bad("invalid hex escape");
                    v = v * 16 + (uint32_t)hexval(src[at++]);
                }
                if (v > 0x10FFFF || (v >= 0xD800 && v <= 0xDFFF)) return bad("invalid code point");
                put_utf8(out, v);
                continue;
            }
            if (e >= '0' && e <= '3') {
                uint32_t v = (uint32_t)(e - '0');
                for (int k = 0; k < 2; k++) {
                    if (at >= src.size() || src[at] < '0' || src[at] > '7') return bad("invalid octal escape");
                    v = v * 8 + (uint32_t)(src[at++] - '0');
                }
                put_utf8(out, v);
                continue;
            }
            at--;
            return bad("invalid escape sequence");
        }
        t.kind = T_STR;
        t.text = std::move(out);
        return true;
    }
    bool number(Token &t) {
        size_t b = at;
        if (src[at] == '0' && at + 1 < src.size() && (src[at + 1] == 'x' || src[at + 1] == 'X')) {
            at += 2;
            size_t d = at;
            uint64_t v = 0;
            while (at < src.size() && hexval(src[at]) >= 0) {
                if (v >> 60) { at = b; return bad("integer literal out of range"); }
                v = v << 4 | (uint64_t)hexval(src[at++]);
            }
            if (at == d) { at = b; return bad("invalid hex literal"); }
            if (at < src.size() && (src[at] == 'u' || src[at] == 'U')) { at = b; return bad("unsigned integer literals are not supported"); }
            t.kind = T_INT;
            t.mag = v;
            return true;
        }
        bool is_float = false;
        while (at < src.size() && isdigit((unsigned char)src[at])) at++;
        if (at + 1 < src.size() && src[at] == '.' && isdigit((unsigned char)src[at + 1])) {
            is_float = true;
            at++;
            while (at < src.size() && isdigit((unsigned char)src[at])) at++;
        }
        if (at < src.size() && (src[at] == 'e' || src[at] == 'E')) {
            size_t save = at++;
            if (at < src.size() && (src[at] == '+' || src[at] == '-')) at++;
            size_t d = at;
            while (at < src.size() && isdigit((unsigned char)src[at])) at++;
            if (at == d) at = save;
            else is_float = true;
        }
        std::string lit = src.substr(b, at - b);
        if (is_float) {
            t.kind = T_FLOAT;
            t.f = strtod(lit.c_str(), nullptr);
            return true;
        }
        if (at < src.size() && (src[at] == 'u' || src[at] == 'U')) { at = b; return bad("unsigned integer literals are not supported"); }
        uint64_t v = 0;
        for (char ch : lit) {
            uint64_t d = (uint64_t)(ch - '0');
            if (v > (std::numeric_limits<uint64_t>::max() - d) / 10) { at = b; return bad("integer literal out of range"); }
            v = v * 10 + d;
        }
        t.kind = T_INT;
        t.mag = v;
        return true;
    }
};

// binding powers (higher binds tighter)
enum { BP_COND = 1, BP_OR = 2, BP_AND = 3, BP_REL = 4, BP_ADD = 5, BP_MUL = 6 };

class Pratt {
public:
    Pratt(const std::string &s, Syntax &o) : sc(s), out(o) {}
    std::string error;

    bool run() {
        if (!advance()) return false;
        int r = expr(BP_COND);
        if (r < 0) return false;
        if (tok.kind != T_END) return fail("unexpected trailing input");
        out.root = r;
        return true;
    }

private:
    Scanner sc;
    Syntax &out;
    Token tok;
    int depth = 0;

    bool fail(const std::string &m) {
        if (error.empty()) error = m + " at offset " + std::to_string(tok.pos);
        return false;
    }
    bool advance() {
        if (!sc.next(tok)) { error = sc.error; return false; }
        return true;
    }
    bool op(const char *s) const { return tok.kind == T_OP && tok.text == s; }
    int node(ExKind k, uint32_t pos) {
        out.nodes.emplace_back();
        out.nodes.back().kind = k;
        out.nodes.back().pos = pos;
        return (int)out.nodes.size() - 1;
    }
    bool binary_op(BinOp &b, int &bp) const {
        if (tok.kind == T_IDENT && tok.text == "in") { b = B_IN; bp = BP_REL; return true; }
        if (tok.kind != T_OP) return false;
        static const struct { const char *s; BinOp b; int bp; } tbl[] = {
            {"||", B_OR, BP_OR}, {"&&", B_AND, BP_AND}, {"==", B_EQ, BP_REL}, {"!=", B_NE, BP_REL}, {"<", B_LT, BP_REL}, {"<=", B_LE, BP_REL},
            {">", B_GT, BP_REL}, {">=", B_GE, BP_REL}, {"+", B_ADD, BP_ADD}, {"-", B_SUB, BP_ADD}, {"*", B_MUL, BP_MUL}, {"/", B_DIV, BP_MUL}, {"%", B_MOD, BP_MUL},
        };
        for (auto &e : tbl) if (tok.text == e.s) { b = e.b; bp = e.bp; return true; }
        return false;
    }

    int expr(int min_bp) {
        if (++depth > 200) { fail("expression nesting too deep"); return -1; }
        int lhs = unary();
        while (lhs >= 0) {
            BinOp b;
            int bp;
            if (binary_op(b, bp) && bp >= min_bp) {
                uint32_t pos = tok.pos;
                if (!advance()) { lhs = -1; break; }
                int rhs = expr(bp + 1);  // left-associative
                if (rhs < 0) { lhs = -1; break; }
                int n = node(EX_BIN, pos);
                out.nodes[n].op = b;
                out.nodes[n].kids = {lhs, rhs};
                if (b == B_IN) out.uses_in = true;
                lhs = n;
                continue;
            }
            if (op("?") && min_bp <= BP_COND) {
                uint32_t pos = tok.pos;
                if (!advance()) { lhs = -1; break; }
                int a = expr(BP_OR);
                if (a < 0) { lhs = -1; break; }
                if (!op(":")) { fail("expected ':' in conditional"); lhs = -1; break; }
                if (!advance()) { lhs = -1; break; }
                int c = expr(BP_COND);
                if (c < 0) { lhs = -1; break; }
                int n = node(EX_COND, pos);
                out.nodes[n].kids = {lhs, a, c};
                lhs = n;
                continue;
            }
            break;
        }
        depth--;
        return lhs;
    }

    int unary() {
        if (op("!") || op("-")) {
            bool is_not = op("!");
            const char *sym = is_not ? "!" : "-";
            uint32_t pos = tok.pos;
            int count = 0;
            while (op(sym)) { count++; if (!advance()) return -1; }
            int m;
            if (!is_not && tok.kind == T_INT && tok.mag == (uint64_t)1 << 63) {
                // the one literal that only exists negated
                m = node(EX_INT, tok.pos);
                out.nodes[m].ival = std::numeric_limits<int64_t>::min();
                if (!advance()) return -1;
                m = postfix(m);
                count--;
            } else {
                m = postfix(primary());
            }
            if (m < 0) return -1;
            for (int k = 0; k < count; k++) {
                int n = node(is_not ? EX_NOT : EX_NEG, pos);
                out.nodes[n].kids = {m};
                m = n;
            }
            return m;
        }
        return postfix(primary());
    }

    bool arguments(std::vector<int> &args) {
        // '(' already consumed
        if (op(")")) return advance();
        while (true) {
            int a = expr(BP_COND);
            if (a < 0) return false;
            args.push_back(a);
            if (op(",")) { if (!advance()) return false; continue; }
            if (op(")")) return advance();
            return fail("expected ',' or ')' in argument list");
        }
    }

    int postfix(int base) {
        while (base >= 0) {
            if (op(".")) {
                if (!advance()) return -1;
                if (tok.kind != T_IDENT) { fail("expected identifier after '.'"); return -1; }
                std::string name = tok.text;
                uint32_t pos = tok.pos;
                if (!advance()) return -1;
                if (op("(")) {
                    if (!advance()) return -1;
                    std::vector<int> args{base};
                    if (!arguments(args)) return -1;
                    int n = node(EX_MCALL, pos);
                    out.nodes[n].text = name;
                    out.nodes[n].kids = args;
                    base = n;
                } else {
                    int n = node(EX_MEMBER, pos);
                    out.nodes[n].text = name;
                    out.nodes[n].kids = {base};
                    base = n;
                }
            } else if (op("[")) {
                uint32_t pos = tok.pos;
                if (!advance()) return -1;
                int idx = expr(BP_COND);
                if (idx < 0) return -1;
                if (!op("]")) { fail("expected ']'"); return -1; }
                if (!advance()) return -1;
                int n = node(EX_INDEX, pos);
                out.nodes[n].kids = {base, idx};
                base = n;
            } else {
                break;
            }
        }
        return base;
    }

    int primary() {
        if (++depth > 200) { fail("expression nesting too deep"); return -1; }
        int r = primary_inner();
        depth--;
        return r;
    }
    int primary_inner() {
        uint32_t pos = tok.pos;
        switch (tok.kind) {
            case T_INT: {
                if (tok.mag > (uint64_t)std::numeric_limits<int64_t>::max()) { fail("integer literal out of range"); return -1; }
                int n = node(EX_INT, pos);
                out.nodes[n].ival = (int64_t)tok.mag;
                return advance() ? n : -1;
            }
            case T_FLOAT: {
                int n = node(EX_FLOAT, pos);
                out.nodes[n].fval = tok.f;
                return advance() ? n : -1;
            }
            case T_STR: {
                int n = node(EX_STR, pos);
                out.nodes[n].text = tok.text;
                return advance() ? n : -1;
            }
            case T_IDENT: {
                std::string name = tok.text;
                if (name == "in") { fail("unexpected 'in'"); return -1; }
                if (!advance()) return -1;
                if (name == "true" || name == "false") {
                    int n = node(EX_BOOL, pos);
                    out.nodes[n].bval = name == "true";
                    return n;
                }
                if (name == "null") return node(EX_NULL, pos);
                if (op("(")) {
                    if (!advance()) return -1;
                    std::vector<int> args;
                    if (!arguments(args)) return -1;
                    int n = node(EX_GCALL, pos);
                    out.nodes[n].text = name;
                    out.nodes[n].kids = args;
                    return n;
                }
                if (op("{")) { fail("message construction is not supported"); return -1; }
                int n = node(EX_IDENT, pos);
                out.nodes[n].text = name;
                return n;
            }
            case T_OP: {
                if (op("(")) {
                    if (!advance()) return -1;
                    int e = expr(BP_COND);
                    if (e < 0) return -1;
                    if (!op(")")) { fail("expected ')'"); return -1; }
                    return advance() ? e : -1;
                }
                if (op("[")) {
                    if (!advance()) return -1;
                    std::vector<int> items;
                    while (!op("]")) {
                        int e = expr(BP_COND);
                        if (e < 0) return -1;
                        items.push_back(e);
                        if (op(",")) { if (!advance()) return -1; continue; }
                        if (!op("]")) { fail("expected ',' or ']' in list literal"); return -1; }
                    }
                    if (!advance()) return -1;
                    int n = node(EX_LIST, pos);
                    out.nodes[n].kids = items;
                    return n;
                }
                if (op("{")) {
                    if (!advance()) return -1;
                    std::vector<int> kv;
                    while (!op("}")) {
                        int k = expr(BP_COND);
                        if (k < 0) return -1;
                        if (!op(":")) { fail("expected ':' in map literal"); return -1; }
                        if (!advance()) return -1;
                        int v = expr(BP_COND);
                        if (v < 0) return -1;
                        kv.push_back(k);
                        kv.push_back(v);
                        if (op(",")) { if (!advance()) return -1; continue; }
                        if (!op("}")) { fail("expected ',' or '}' in map literal"); return -1; }
                    }
                    if (!advance()) return -1;
                    int n = node(EX_MAP, pos);
                    out.nodes[n].kids = kv;
                    return n;
                }
                fail("unexpected token '" + tok.text + "'");
                return -1;
            }
            case T_END: fail("unexpected end of expression"); return -1;
        }
        return -1;
    }
};

}  // namespace

bool parse_expression(const std::string &src, Syntax &out, std::string &err) {
    out = Syntax();
    Pratt p(src, out);
    if (!p.run()) {
        err = p.error.empty() ? "syntax error" : p.error;
        return false;
    }
    return true;
}

}  // namespace pwaf
