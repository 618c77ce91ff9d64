// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_regex.h header note).
//
// Recursive-descent parser for the Rust-regex syntax subset + Pike-VM (thread-set) simulation.
// Supported: literals, escapes (\. \xHH \x{H..} \n \r \t \f \v \a, punctuation), . (no \n unless
// (?s)), classes [..] with ranges / negation / \d\w\s\D\W\S / [[:posix:]], groups ( ) (?: )
// (?P<n> ) (?<n> ), flags i m s U (inline and scoped), | alternation, * + ? {n} {n,} {n,m} with
// optional lazy '?', anchors ^ $ \A \z, word boundaries \b \B.
// Not supported (compile error): back-references/look-around (also absent from the regex crate),
// (?x), class set operations (&& -- ~~), \p{..} Unicode classes, \< \> word-edge escapes, non-ASCII
// code points in escapes. Haystacks are matched as bytes; the reference's fields are ASCII by
// construction (http_listener.rs:159-165,284-296), where byte and Unicode semantics coincide.
#include "oracle_regex.h"

#include <algorithm>
#include <bitset>
#include <cstring>
#include <functional>

namespace oracle {

namespace {

using ByteSet = std::bitset<256>;

enum class AKind { StartText, EndText, StartLine, EndLine, WordB, NotWordB };

struct Ast {
    enum K { Empty, Set, Cat, Alt, Repeat, Assert, Group } k = Empty;
    ByteSet set;
    std::vector<std::unique_ptr<Ast>> kids;
    int rmin = 0, rmax = -1;  // rmax -1 = unbounded
    AKind ak = AKind::StartText;
};
using AstP = std::unique_ptr<Ast>;

struct Flags {
    bool i = false, m = false, s = false;
};

static bool is_word(uint8_t c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
}

struct Parser {
    std::string_view p;
    size_t pos = 0;
    std::string err;
    int depth = 0;
    size_t budget = 0;  // guards {n,m} explosion

    bool fail(const std::string &m) {
        if (err.empty()) err = m + " at offset " + std::to_string(pos);
        return false;
    }
    bool eof() const { return pos >= p.size(); }
    char peek() const { return p[pos]; }

    static void add_ci(ByteSet &s) {
        for (int c = 'a'; c <= 'z'; c++) {
            if (s[c] || s[c - 32]) {
                s.set(c);
                s.set(c - 32);
            }
        }
    }
    static AstP mk_set(const ByteSet &s) {
        auto a = std::make_unique<Ast>();
        a->k = Ast::Set;
        a->set = s;
        return a;
    }
    static AstP mk_byte(uint8_t c, const Flags &f) {
        ByteSet s;
        s.set(c);
        if (f.i) add_ci(s);
        return mk_set(s);
    }
    static AstP mk_assert(AKind k) {
        auto a = std::make_unique<Ast>();
        a->k = Ast::Assert;
        a->ak = k;
        return a;
    }

    // \p{..} / \pX / \P{..} / \p{^..}: Unicode general categories over ASCII text (the reference's fields are ASCII by construction:
    // http_listener.rs:159-165 keeps only visible ASCII, http::Uri is ASCII) — regex-syntax folds the class under (?i) first, then
    // negates it. pos is at the 'p' / 'P'. Returns false after fail() for anything but a general category (or Latin / ASCII / Any).
    bool unicode_property(ByteSet &s, bool fold_case) {
        bool negated = peek() == 'P';
        pos++;
        std::string nm;
        if (!eof() && peek() == '{') {
            size_t end = p.find('}', pos);
            if (end == std::string::npos) { fail("unterminated \\p{"); return false; }
            nm = p.substr(pos + 1, end - pos - 1);
            pos = end + 1;
        } else if (!eof()) {
            nm = std::string(1, peek());
            pos++;
        } else {
            fail("incomplete \\p");
            return false;
        }
        if (!nm.empty() && nm[0] == '^') { negated = !negated; nm = nm.substr(1); }
        std::string k;
        for (char ch : nm) if (ch != '_' && ch != '-' && ch != ' ') k.push_back((char)std::tolower((unsigned char)ch));
        auto is = [&](std::initializer_list<const char *> names) { for (const char *n : names) if (k == n) return true; return false; };
        auto upper = [](int c) { return c >= 'A' && c <= 'Z'; };
        auto lower = [](int c) { return c >= 'a' && c <= 'z'; };
        auto in = [](int c, const char *set) { return c != 0 && std::strchr(set, c) != nullptr; };
        std::function<bool(int)> pred;
        if (is({"l", "letter", "alphabetic", "alpha", "latin", "latn", "lc", "casedletter"})) pred = [&](int c) { return upper(c) || lower(c); };
        else if (is({"lu", "uppercaseletter", "uppercase", "upper"})) pred = upper;
        else if (is({"ll", "lowercaseletter", "lowercase", "lower"})) pred = lower;
        else if (is({"n", "number", "nd", "decimalnumber", "digit"})) pred = [](int c) { return c >= '0' && c <= '9'; };
        else if (is({"p", "punctuation", "punct"})) pred = [&](int c) { return in(c, "!\"#%&'()*,-./:;?@[\\]_{}"); };
        else if (is({"pc", "connectorpunctuation"})) pred = [](int c) { return c == '_'; };
        else if (is({"pd", "dashpunctuation"})) pred = [](int c) { return c == '-'; };
        else if (is({"ps", "openpunctuation"})) pred = [&](int c) { return in(c, "([{"); };
        else if (is({"pe", "closepunctuation"})) pred = [&](int c) { return in(c, ")]}"); };
        else if (is({"po", "otherpunctuation"})) pred = [&](int c) { return in(c, "!\"#%&'*,./:;?@\\"); };
        else if (is({"s", "symbol"})) pred = [&](int c) { return in(c, "$+<=>^`|~"); };
        else if (is({"sc", "currencysymbol"})) pred = [](int c) { return c == '$'; };
        else if (is({"sm", "mathsymbol"})) pred = [&](int c) { return in(c, "+<=>|~"); };
        else if (is({"sk", "modifiersymbol"})) pred = [&](int c) { return in(c, "^`"); };
        else if (is({"z", "separator", "zs", "spaceseparator"})) pred = [](int c) { return c == ' '; };
        else if (is({"cc", "control", "cntrl", "c", "other"})) pred = [](int c) { return c < 0x20 || c == 0x7F; };
        else if (is({"ascii"})) pred = [](int c) { return c < 0x80; };
        else if (is({"any"})) pred = [](int) { return true; };
        else if (is({"lt", "titlecaseletter", "lm", "modifierletter", "lo", "otherletter", "m", "mark", "mn", "mc", "me", "nl", "letternumber", "no", "othernumber", "pi",
                     "initialpunctuation", "pf", "finalpunctuation", "so", "othersymbol", "zl", "lineseparator", "zp", "paragraphseparator", "cf", "format", "cs", "surrogate",
                     "co", "privateuse", "cn", "unassigned"}))
            pred = [](int) { return false; };  // categories without an ASCII member
        else { fail("unsupported: Unicode property \\p{" + nm + "} (only general categories, ASCII-restricted)"); return false; }
        ByteSet t;
        for (int c = 0; c < 256; c++)
            if ((k == "any" || c < 0x80) && pred(c)) t.set((size_t)c);
        if (fold_case)
            for (int c = 'a'; c <= 'z'; c++) {
                if (t.test((size_t)c)) t.set((size_t)(c - 32));
                if (t.test((size_t)(c - 32))) t.set((size_t)c);
            }
        if (negated) t = ~t;
        s |= t;
        return true;
    }

    static void perl_class(char c, ByteSet &s) {
        ByteSet t;
        switch (c) {
            case 'd': case 'D':
                for (int x = '0'; x <= '9'; x++) t.set(x);
                break;
            case 'w': case 'W':
                for (int x = 0; x < 256; x++) if (is_word((uint8_t)x)) t.set(x);
                break;
            case 's': case 'S':
                t.set('\t'); t.set('\n'); t.set(0x0B); t.set(0x0C); t.set('\r'); t.set(' ');
                break;
        }
        if (c == 'D' || c == 'W' || c == 'S') t = ~t;
        s |= t;
    }

    static int hexv(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    // Parses an escape that denotes a single byte (after the backslash has been consumed and
    // p[pos] is the escape char). Returns -1 on error / not-a-single-byte escape.
    int escape_byte() {
        char c = peek();
        pos++;
        switch (c) {
            case 'n': return '\n';
            case 'r': return '\r';
            case 't': return '\t';
            case 'f': return 0x0C;
            case 'v': return 0x0B;
            case 'a': return 0x07;
            case 'x': {
                if (eof()) { fail("incomplete \\x escape"); return -1; }
                unsigned v = 0;
                if (peek() == '{') {
                    pos++;
                    int n = 0;
                    while (!eof() && peek() != '}') {
                        int h = hexv(peek());
                        if (h < 0) { fail("invalid hex digit"); return -1; }
                        v = v * 16 + h;
                        if (v > 0x10FFFF) { fail("hex escape out of range"); return -1; }
                        pos++; n++;
                    }
                    if (eof() || n == 0) { fail("unclosed \\x{ escape"); return -1; }
                    pos++;
                } else {
                    for (int k = 0; k < 2; k++) {
                        if (eof()) { fail("incomplete \\x escape"); return -1; }
                        int h = hexv(peek());
                        if (h < 0) { fail("invalid hex digit"); return -1; }
                        v = v * 16 + h;
                        pos++;
                    }
                }
                if (v > 0x7F) { fail("unsupported: non-ASCII code point escape"); return -1; }
                return (int)v;
            }
            default:
                if ((c >= '!' && c <= '/') || (c >= ':' && c <= '@') || (c >= '[' && c <= '`') ||
                    (c >= '{' && c <= '~') || c == ' ') {
                    if (c == '<' || c == '>') { fail("unsupported: \\< \\> word-edge assertions"); return -1; }
                    return (uint8_t)c;
                }
                fail(std::string("unrecognized escape sequence \\") + c);
                return -1;
        }
    }

    bool parse_posix(ByteSet &s) {
        // at "[:" ; parse [:name:] or [:^name:]
        size_t save = pos;
        pos += 2;
        bool neg = false;
        if (!eof() && peek() == '^') { neg = true; pos++; }
        size_t b = pos;
        while (!eof() && peek() != ':') pos++;
        if (pos + 1 >= p.size() || p[pos + 1] != ']') { pos = save; return false; }
        std::string name(p.substr(b, pos - b));
        pos += 2;
        ByteSet t;
        auto range = [&](int a, int z) { for (int x = a; x <= z; x++) t.set(x); };
        if (name == "alnum") { range('0', '9'); range('a', 'z'); range('A', 'Z'); }
        else if (name == "alpha") { range('a', 'z'); range('A', 'Z'); }
        else if (name == "ascii") range(0, 127);
        else if (name == "blank") { t.set(' '); t.set('\t'); }
        else if (name == "cntrl") { range(0, 31); t.set(127); }
        else if (name == "digit") range('0', '9');
        else if (name == "graph") range('!', '~');
        else if (name == "lower") range('a', 'z');
        else if (name == "print") range(' ', '~');
        else if (name == "punct") { range('!', '/'); range(':', '@'); range('[', '`'); range('{', '~'); }
        else if (name == "space") { t.set('\t'); t.set('\n'); t.set(0x0B); t.set(0x0C); t.set('\r'); t.set(' '); }
        else if (name == "upper") range('A', 'Z');
        else if (name == "word") { for (int x = 0; x < 256; x++) if (is_word((uint8_t)x)) t.set(x); }
        else if (name == "xdigit") { range('0', '9'); range('a', 'f'); range('A', 'F'); }
        else { fail("unknown POSIX class " + name); return true; }
        if (neg) t = ~t;
        s |= t;
        return true;
    }

    AstP parse_class(const Flags &f) {
        // at '['
        pos++;
        bool neg = false;
        if (!eof() && peek() == '^') { neg = true; pos++; }
        ByteSet s;
        bool first = true;
        for (;;) {
            if (eof()) { fail("unclosed character class"); return nullptr; }
            char c = peek();
            if (c == ']' && !first) { pos++; break; }
            first = false;
            int lo = -1;
            if (c == '[') {
                if (pos + 1 < p.size() && p[pos + 1] == ':') {
                    if (parse_posix(s)) { if (!err.empty()) return nullptr; continue; }
                }
                fail("unsupported: nested character class");
                return nullptr;
            }
            if ((c == '&' || c == '-' || c == '~') && pos + 1 < p.size() && p[pos + 1] == c) {
                fail("unsupported: character class set operation");
                return nullptr;
            }
            if (c == '\\') {
                pos++;
                if (eof()) { fail("incomplete escape"); return nullptr; }
                char e = peek();
                if (strchr("dDwWsS", e)) { pos++; perl_class(e, s); continue; }
                if (e == 'p' || e == 'P') { if (!unicode_property(s, f.i)) return nullptr; continue; }
                if (e == 'b') { pos++; lo = 0x08; }  // inside a class \b is backspace
                else { lo = escape_byte(); if (lo < 0) return nullptr; }
            } else {
                lo = (uint8_t)c;
                pos++;
            }
            int hi = lo;
            if (pos + 1 < p.size() && peek() == '-' && p[pos + 1] != ']') {
                pos++;
                char c2 = peek();
                if (c2 == '\\') {
                    pos++;
                    if (eof()) { fail("incomplete escape"); return nullptr; }
                    if (strchr("dDwWsSpP", peek())) { fail("invalid class range"); return nullptr; }
                    hi = escape_byte();
                    if (hi < 0) return nullptr;
                } else if (c2 == '[') {
                    fail("unsupported: nested character class");
                    return nullptr;
                } else {
                    hi = (uint8_t)c2;
                    pos++;
                }
                if (hi < lo) { fail("invalid character class range"); return nullptr; }
            }
            for (int x = lo; x <= hi; x++) s.set(x);
        }
        if (f.i) add_ci(s);
        if (neg) s = ~s;
        return mk_set(s);
    }

    // returns false on error; sets `scoped`=true when a "(?flags:" group was opened
    bool parse_flags(Flags &f, bool &scoped) {
        // at first flag char after "(?"
        bool on = true;
        bool any = false;
        for (;;) {
            if (eof()) return fail("unclosed flag group");
            char c = peek();
            if (c == ')') { pos++; scoped = false; return any ? true : fail("empty flag group"); }
            if (c == ':') { pos++; scoped = true; return true; }
            pos++;
            any = true;
            switch (c) {
                case '-': on = false; break;
                case 'i': f.i = on; break;
                case 'm': f.m = on; break;
                case 's': f.s = on; break;
                case 'U': break;  // swap-greed: irrelevant for is_match
                case 'u': if (!on) return fail("unsupported: (?-u)"); break;
                case 'R': break;  // CRLF mode off by default; ignore
                case 'x': return fail("unsupported: (?x) verbose mode");
                default: return fail(std::string("unrecognized flag ") + c);
            }
        }
    }

    AstP parse_atom(Flags &f) {
        char c = peek();
        if (c == '(') {
            pos++;
            Flags inner = f;
            if (!eof() && peek() == '?') {
                pos++;
                if (eof()) { fail("unclosed group"); return nullptr; }
                char d = peek();
                if (d == 'P' || d == '<') {
                    if (d == 'P') pos++;
                    if (eof() || peek() != '<') { fail("invalid group syntax"); return nullptr; }
                    pos++;
                    if (!eof() && (peek() == '=' || peek() == '!')) { fail("look-behind is not supported"); return nullptr; }
                    size_t b = pos;
                    while (!eof() && peek() != '>') pos++;
                    if (eof() || pos == b) { fail("invalid capture group name"); return nullptr; }
                    pos++;
                } else if (d == '=' || d == '!') {
                    fail("look-around is not supported");
                    return nullptr;
                } else {
                    bool scoped = false;
                    if (!parse_flags(inner, scoped)) return nullptr;
                    if (!scoped) {
                        f = inner;  // applies to the rest of the enclosing group
                        auto e = std::make_unique<Ast>();
                        e->k = Ast::Empty;
                        return e;
                    }
                }
            }
            if (++depth > 200) { fail("nesting too deep"); return nullptr; }
            AstP a = parse_alt(inner);
            depth--;
            if (!a) return nullptr;
            if (eof() || peek() != ')') { fail("unclosed group"); return nullptr; }
            pos++;
            auto g = std::make_unique<Ast>();
            g->k = Ast::Group;
            g->kids.push_back(std::move(a));
            return g;
        }
        if (c == '[') return parse_class(f);
        if (c == '.') {
            pos++;
            ByteSet s;
            s.set();
            if (!f.s) s.reset('\n');
            return mk_set(s);
        }
        if (c == '^') { pos++; return mk_assert(f.m ? AKind::StartLine : AKind::StartText); }
        if (c == '$') { pos++; return mk_assert(f.m ? AKind::EndLine : AKind::EndText); }
        if (c == '\\') {
            pos++;
            if (eof()) { fail("incomplete escape"); return nullptr; }
            char e = peek();
            if (strchr("dDwWsS", e)) {
                pos++;
                ByteSet s;
                perl_class(e, s);
                // (?i) cannot change these sets (closed under ASCII case)
                return mk_set(s);
            }
            if (e == 'A') { pos++; return mk_assert(AKind::StartText); }
            if (e == 'z') { pos++; return mk_assert(AKind::EndText); }
            if (e == 'b') { pos++; return mk_assert(AKind::WordB); }
            if (e == 'B') { pos++; return mk_assert(AKind::NotWordB); }
            if (e == 'p' || e == 'P') {
                ByteSet s;
                if (!unicode_property(s, f.i)) return nullptr;
                return mk_set(s);
            }
            if (e >= '0' && e <= '9') { fail("backreferences are not supported"); return nullptr; }
            int b = escape_byte();
            if (b < 0) return nullptr;
            return mk_byte((uint8_t)b, f);
        }
        if (c == '*' || c == '+' || c == '?') { fail("repetition operator missing expression"); return nullptr; }
        if (c == '{') { fail("repetition operator missing expression"); return nullptr; }
        pos++;
        return mk_byte((uint8_t)c, f);
    }

    bool parse_counted(int &mn, int &mx) {
        // at '{'
        size_t save = pos;
        pos++;
        auto num = [&](int &v) -> bool {
            size_t b = pos;
            long x = 0;
            while (!eof() && peek() >= '0' && peek() <= '9') {
                x = x * 10 + (peek() - '0');
                if (x > 100000) return false;
                pos++;
            }
            if (pos == b) return false;
            v = (int)x;
            return true;
        };
        if (!num(mn)) { pos = save; return fail("invalid counted repetition"); }
        if (!eof() && peek() == '}') { pos++; mx = mn; return true; }
        if (eof() || peek() != ',') { pos = save; return fail("invalid counted repetition"); }
        pos++;
        if (!eof() && peek() == '}') { pos++; mx = -1; return true; }
        if (!num(mx)) { pos = save; return fail("invalid counted repetition"); }
        if (eof() || peek() != '}') { pos = save; return fail("unclosed counted repetition"); }
        pos++;
        if (mx < mn) return fail("invalid repetition range");
        return true;
    }

    AstP parse_repeat(Flags &f) {
        AstP a = parse_atom(f);
        if (!a) return nullptr;
        while (!eof()) {
            char c = peek();
            int mn = 0, mx = 0;
            if (c == '*') { mn = 0; mx = -1; pos++; }
            else if (c == '+') { mn = 1; mx = -1; pos++; }
            else if (c == '?') { mn = 0; mx = 1; pos++; }
            else if (c == '{') { if (!parse_counted(mn, mx)) return nullptr; }
            else break;
            if (!eof() && peek() == '?') pos++;  // lazy: same language
            if (a->k == Ast::Empty && a->kids.empty()) { fail("repetition operator missing expression"); return nullptr; }
            auto r = std::make_unique<Ast>();
            r->k = Ast::Repeat;
            r->rmin = mn;
            r->rmax = mx;
            r->kids.push_back(std::move(a));
            a = std::move(r);
        }
        return a;
    }

    AstP parse_cat(Flags &f) {
        auto cat = std::make_unique<Ast>();
        cat->k = Ast::Cat;
        while (!eof() && peek() != '|' && peek() != ')') {
            AstP a = parse_repeat(f);
            if (!a) return nullptr;
            cat->kids.push_back(std::move(a));
        }
        return cat;
    }

    AstP parse_alt(Flags f) {
        auto alt = std::make_unique<Ast>();
        alt->k = Ast::Alt;
        for (;;) {
            AstP c = parse_cat(f);
            if (!c) return nullptr;
            alt->kids.push_back(std::move(c));
            if (!eof() && peek() == '|') { pos++; continue; }
            break;
        }
        if (alt->kids.size() == 1) return std::move(alt->kids[0]);
        return alt;
    }
};

}  // namespace

// ---- Pike VM ---------------------------------------------------------------------------------
struct Inst {
    enum Op : uint8_t { Byte, Split, Jmp, Assert, Match } op;
    uint32_t x = 0, y = 0;  // Byte: set index; Split: two targets; Jmp: x
    AKind ak = AKind::StartText;
};
struct RegexProg {
    std::vector<Inst> code;
    std::vector<ByteSet> sets;
};

namespace {

struct Compiler {
    RegexProg &pr;
    std::string err;
    static constexpr size_t kMaxInst = 200000;

    uint32_t emit(Inst i) {
        pr.code.push_back(i);
        return (uint32_t)pr.code.size() - 1;
    }
    bool gen(const Ast &a) {
        if (pr.code.size() > kMaxInst) { err = "regex too large"; return false; }
        switch (a.k) {
            case Ast::Empty: return true;
            case Ast::Set: {
                Inst i; i.op = Inst::Byte; i.x = (uint32_t)pr.sets.size();
                pr.sets.push_back(a.set);
                emit(i);
                return true;
            }
            case Ast::Group: return gen(*a.kids[0]);
            case Ast::Cat:
                for (auto &k : a.kids) if (!gen(*k)) return false;
                return true;
            case Ast::Alt: {
                std::vector<uint32_t> jmps;
                for (size_t n = 0; n < a.kids.size(); n++) {
                    if (n + 1 < a.kids.size()) {
                        Inst s; s.op = Inst::Split;
                        uint32_t si = emit(s);
                        pr.code[si].x = si + 1;
                        if (!gen(*a.kids[n])) return false;
                        Inst j; j.op = Inst::Jmp;
                        jmps.push_back(emit(j));
                        pr.code[si].y = (uint32_t)pr.code.size();
                    } else {
                        if (!gen(*a.kids[n])) return false;
                    }
                }
                for (uint32_t j : jmps) pr.code[j].x = (uint32_t)pr.code.size();
                return true;
            }
            case Ast::Assert: {
                Inst i; i.op = Inst::Assert; i.ak = a.ak;
                emit(i);
                return true;
            }
            case Ast::Repeat: {
                const Ast &k = *a.kids[0];
                for (int n = 0; n < a.rmin; n++) if (!gen(k)) return false;
                if (a.rmax < 0) {
                    // k*  : L: split(L+1, out); k; jmp L
                    Inst s; s.op = Inst::Split;
                    uint32_t si = emit(s);
                    pr.code[si].x = si + 1;
                    if (!gen(k)) return false;
                    Inst j; j.op = Inst::Jmp; j.x = si;
                    emit(j);
                    pr.code[si].y = (uint32_t)pr.code.size();
                } else {
                    std::vector<uint32_t> splits;
                    for (int n = a.rmin; n < a.rmax; n++) {
                        Inst s; s.op = Inst::Split;
                        uint32_t si = emit(s);
                        pr.code[si].x = si + 1;
                        splits.push_back(si);
                        if (!gen(k)) return false;
                    }
                    for (uint32_t si : splits) pr.code[si].y = (uint32_t)pr.code.size();
                }
                return true;
            }
        }
        return true;
    }
};

struct SparseSet {
    std::vector<uint32_t> dense, sparse;
    size_t n = 0;
    explicit SparseSet(size_t cap) : dense(cap), sparse(cap) {}
    bool has(uint32_t v) const { return sparse[v] < n && dense[sparse[v]] == v; }
    void add(uint32_t v) { sparse[v] = (uint32_t)n; dense[n++] = v; }
    void clear() { n = 0; }
};

}  // namespace

bool Regex::compile(std::string_view pattern, Regex &out, std::string &err) {
    Parser ps;
    ps.p = pattern;
    Flags f;
    AstP ast = ps.parse_alt(f);
    if (ast && !ps.eof()) {
        if (ps.peek() == ')') ps.fail("unopened group");
        else ps.fail("unexpected character");
        ast.reset();
    }
    if (!ast) {
        err = "regex parse error: " + ps.err;
        return false;
    }
    auto prog = std::make_shared<RegexProg>();
    Compiler c{*prog, {}};
    if (!c.gen(*ast)) {
        err = "regex compile error: " + c.err;
        return false;
    }
    Inst m;
    m.op = Inst::Match;
    prog->code.push_back(m);
    out.prog = prog;
    return true;
}

bool Regex::is_match(std::string_view h) const {
    const RegexProg &pr = *prog;
    const size_t ninst = pr.code.size();
    SparseSet cur(ninst), nxt(ninst);
    std::vector<uint32_t> stack;
    const size_t n = h.size();

    // follow zero-width instructions from pc at boundary `at` (between h[at-1] and h[at])
    auto addthread = [&](SparseSet &set, uint32_t pc0, size_t at) -> bool {
        stack.clear();
        stack.push_back(pc0);
        bool matched = false;
        while (!stack.empty()) {
            uint32_t pc = stack.back();
            stack.pop_back();
            if (set.has(pc)) continue;
            set.add(pc);
            const Inst &in = pr.code[pc];
            switch (in.op) {
                case Inst::Jmp: stack.push_back(in.x); break;
                case Inst::Split: stack.push_back(in.y); stack.push_back(in.x); break;
                case Inst::Assert: {
                    bool ok = false;
                    bool pw = at > 0 && is_word((uint8_t)h[at - 1]);
                    bool nw = at < n && is_word((uint8_t)h[at]);
                    switch (in.ak) {
                        case AKind::StartText: ok = at == 0; break;
                        case AKind::EndText: ok = at == n; break;
                        case AKind::StartLine: ok = at == 0 || h[at - 1] == '\n'; break;
                        case AKind::EndLine: ok = at == n || h[at] == '\n'; break;
                        case AKind::WordB: ok = pw != nw; break;
                        case AKind::NotWordB: ok = pw == nw; break;
                    }
                    if (ok) stack.push_back(pc + 1);
                    break;
                }
                case Inst::Match: matched = true; break;
                case Inst::Byte: break;
            }
        }
        return matched;
    };

    cur.clear();
    for (size_t at = 0;; at++) {
        // unanchored search: a new thread starts at every boundary
        if (addthread(cur, 0, at)) return true;
        // (threads carried over were closed when they were added below)
        if (at == n) break;
        uint8_t c = (uint8_t)h[at];
        nxt.clear();
        bool matched = false;
        for (size_t k = 0; k < cur.n; k++) {
            const Inst &in = pr.code[cur.dense[k]];
            if (in.op == Inst::Byte && pr.sets[in.x][c]) {
                if (addthread(nxt, cur.dense[k] + 1, at + 1)) matched = true;
            }
        }
        if (matched) return true;
        std::swap(cur, nxt);
    }
    return false;
}

}  // namespace oracle
