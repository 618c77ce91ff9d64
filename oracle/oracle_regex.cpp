// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_regex.h header note).
//
// Recursive-descent parser for the Rust-regex syntax + Pike-VM (thread-set) simulation over SCALAR VALUES.
//
// The reference hands `Display(Uri)` and `uri.path()` to `bel` as Rust `str` (pingoo/rules.rs:16-25, serde_utils.rs:16-18) and pins
// http 1.3.1 (Cargo.lock:824-826: UTF-8 is admitted in path and query) and regex 1.12.2 (Cargo.lock:1694-1700), which is Unicode-aware
// BY DEFAULT: `.` and negated classes consume one scalar value, \d \s \w \b and \p{..} read Unicode tables, (?i) is simple case folding
// ((?i)s matches U+017F, (?i)k matches U+212A). Round 4 matched bytes with ASCII classes and called the difference unreachable; it is
// reachable (`?q=union<U+00A0>select`), so this restatement follows the crate's Unicode semantics (regex-syntax's translation rules):
//   literals / escapes (\xHH \x{H..} \uHHHH \u{H..} \UHHHHHHHH \n \r \t \f \v \a, punctuation) are code points; `.` = any scalar but \n
//   ((?s): any); classes [..] with ranges / negation / \d\w\s\D\W\S / [[:posix:]] / \p{..}, folded under (?i) as a whole and THEN
//   negated; \d = Nd, \s = White_Space, \w = Alphabetic + M + Nd + Pc + Join_Control; \p{..}: general categories, scripts, Alphabetic /
//   White_Space / Lowercase / Uppercase / Any / ASCII / Assigned (tables: unicode_data.inc, Unicode 13.0 from perl's UCD — newer
//   assignments of the crate's tables read as unassigned: DESIGN.md D19); groups, flags i m s U u x (inline and scoped), |, * + ?
//   {n} {n,} {n,m} (+ lazy), ^ $ \A \z, \b \B (Unicode; ASCII under (?-u)); (?x) verbose mode. (?-u) makes \d \s \w \b (?i) ASCII and
//   refuses what the crate's `Regex` (UTF-8 mode) refuses because it could match invalid UTF-8: `.`, negated classes, \xHH >= 0x80.
// Not supported (compile error): back-references / look-around (absent from the crate too), class set operations and nested classes,
// other \p{..} properties (age, Script_Extensions, ...), \< \> \b{start}.., (?R) CRLF mode switched on.
// Haystacks: the reference's strings are valid UTF-8 by type. A byte that is not part of a well-formed sequence is one UNIT that no
// class matches (not even `.` or a negated class — exactly what UTF-8 automata do) and next to which \b and \B are both false (D17).
#include "oracle_regex.h"

#include <algorithm>
#include <cstring>
#include <functional>

namespace oracle {

namespace {

#include "unicode_data.inc"

using Range = std::pair<uint32_t, uint32_t>;
using CpSet = std::vector<Range>;  // sorted, disjoint, non-adjacent, within scalar values
constexpr uint32_t kMaxCp = 0x10FFFF;

static void normalize(CpSet &s) {
    std::sort(s.begin(), s.end());
    CpSet o;
    for (auto r : s) {
        if (!o.empty() && r.first <= o.back().second + 1) o.back().second = std::max(o.back().second, r.second);
        else o.push_back(r);
    }
    // surrogates are not scalar values
    CpSet t;
    for (auto r : o) {
        if (r.second < 0xD800 || r.first > 0xDFFF) { t.push_back(r); continue; }
        if (r.first < 0xD800) t.push_back({r.first, 0xD7FF});
        if (r.second > 0xDFFF) t.push_back({0xE000, r.second});
    }
    s.swap(t);
}
static CpSet negate(const CpSet &s) {
    CpSet o;
    uint32_t at = 0;
    for (auto r : s) {
        if (r.first > at) o.push_back({at, r.first - 1});
        at = r.second + 1;
    }
    if (at <= kMaxCp) o.push_back({at, kMaxCp});
    normalize(o);
    return o;
}
static bool contains(const CpSet &s, uint32_t c) {
    size_t lo = 0, hi = s.size();
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (s[mid].second < c) lo = mid + 1;
        else hi = mid;
    }
    return lo < s.size() && s[lo].first <= c;
}
static void add_table(CpSet &s, const UniTable &t) {
    for (unsigned k = 0; k < t.count; k++) s.push_back({kUniRanges[t.first + k][0], kUniRanges[t.first + k][1]});
}
static const UniTable *find_table(int kind, const std::string &key) {
    for (const UniTable &t : kUniTables) {
        if (t.kind != kind) continue;
        const char *q = t.names;
        while (*q) {
            const char *e = strchr(q, '|');
            size_t len = e ? (size_t)(e - q) : strlen(q);
            if (len == key.size() && !memcmp(q, key.data(), len)) return &t;
            q += len + (e ? 1 : 0);
        }
    }
    return nullptr;
}
// simple case folding closure of a class (regex-syntax: ClassUnicode::case_fold_simple)
static void fold_unicode(CpSet &s) {
    normalize(s);
    CpSet add;
    const size_t n = sizeof(kUniFold) / sizeof(kUniFold[0]);
    for (auto r : s) {
        size_t lo = 0, hi = n;
        while (lo < hi) {
            size_t mid = (lo + hi) / 2;
            if (kUniFold[mid][0] < r.first) lo = mid + 1;
            else hi = mid;
        }
        for (size_t k = lo; k < n && kUniFold[k][0] <= r.second; k++) add.push_back({kUniFold[k][1], kUniFold[k][1]});
    }
    s.insert(s.end(), add.begin(), add.end());
    normalize(s);
}
static void fold_ascii(CpSet &s) {
    normalize(s);
    CpSet add;
    for (uint32_t c = 'a'; c <= 'z'; c++) {
        if (contains(s, c)) add.push_back({c - 32, c - 32});
        if (contains(s, c - 32)) add.push_back({c, c});
    }
    s.insert(s.end(), add.begin(), add.end());
    normalize(s);
}
static CpSet word_set(bool unicode) {
    CpSet s;
    if (!unicode) {
        s = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}};
        return s;
    }
    add_table(s, *find_table(2, "alphabetic"));
    add_table(s, *find_table(0, "m"));
    add_table(s, *find_table(0, "nd"));
    add_table(s, *find_table(0, "pc"));
    add_table(s, *find_table(2, "joincontrol"));
    normalize(s);
    return s;
}
static const CpSet &word_unicode() {
    static const CpSet w = word_set(true);
    return w;
}

enum class AKind { StartText, EndText, StartLine, EndLine, WordB, NotWordB, WordBAscii, NotWordBAscii };

struct Ast {
    enum K { Empty, Set, Cat, Alt, Repeat, Assert, Group } k = Empty;
    CpSet set;
    std::vector<std::unique_ptr<Ast>> kids;
    int rmin = 0, rmax = -1;  // rmax -1 = unbounded
    AKind ak = AKind::StartText;
};
using AstP = std::unique_ptr<Ast>;

struct Flags {
    bool i = false, m = false, s = false, u = true, x = false;
};

struct Parser {
    std::string_view p;
    size_t pos = 0;
    std::string err;
    int depth = 0;

    bool fail(const std::string &m) {
        if (err.empty()) err = m + " at offset " + std::to_string(pos);
        return false;
    }
    bool eof() const { return pos >= p.size(); }
    char peek() const { return p[pos]; }

    // the pattern is a Rust string: one scalar value at `pos` (a malformed pattern byte is refused)
    bool next_char(uint32_t &c) {
        const uint8_t b0 = (uint8_t)p[pos];
        if (b0 < 0x80) { c = b0; pos++; return true; }
        int len = b0 >= 0xF0 ? 4 : b0 >= 0xE0 ? 3 : b0 >= 0xC2 ? 2 : 0;
        if (!len || pos + (size_t)len > p.size()) return fail("pattern is not valid UTF-8");
        uint32_t v = b0 & (0xFFu >> (len + 1));
        for (int k = 1; k < len; k++) {
            const uint8_t b = (uint8_t)p[pos + (size_t)k];
            if ((b & 0xC0) != 0x80) return fail("pattern is not valid UTF-8");
            v = (v << 6) | (b & 0x3Fu);
        }
        if ((len == 3 && v < 0x800) || (len == 4 && (v < 0x10000 || v > kMaxCp)) || (v >= 0xD800 && v <= 0xDFFF)) return fail("pattern is not valid UTF-8");
        c = v;
        pos += (size_t)len;
        return true;
    }
    void skip_space(const Flags &f) {  // (?x): whitespace and # comments between tokens
        if (!f.x) return;
        while (!eof()) {
            const char c = peek();
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') { pos++; continue; }
            if (c == '#') { while (!eof() && peek() != '\n') pos++; continue; }
            break;
        }
    }

    static AstP mk_set(CpSet s) {
        auto a = std::make_unique<Ast>();
        a->k = Ast::Set;
        normalize(s);
        a->set = std::move(s);
        return a;
    }
    AstP mk_char(uint32_t c, const Flags &f) {
        CpSet s{{c, c}};
        if (f.i) { if (f.u) fold_unicode(s); else fold_ascii(s); }
        return mk_set(s);
    }
    static AstP mk_assert(AKind k) {
        auto a = std::make_unique<Ast>();
        a->k = Ast::Assert;
        a->ak = k;
        return a;
    }

    // \p{..} / \pX / \P{..} / \p{^..}. pos is at the 'p' / 'P'. The item's class, case-folded when asked (regex-syntax folds a class
    // item under (?i) and then negates it), is united into `s`.
    bool unicode_property(CpSet &s, const Flags &f) {
        if (!f.u) return fail("Unicode class not allowed without the u flag");
        bool negated = peek() == 'P';
        pos++;
        std::string nm;
        if (!eof() && peek() == '{') {
            size_t end = p.find('}', pos);
            if (end == std::string::npos) return fail("unterminated \\p{");
            nm = std::string(p.substr(pos + 1, end - pos - 1));
            pos = end + 1;
        } else if (!eof()) {
            nm = std::string(1, peek());
            pos++;
        } else {
            return fail("incomplete \\p");
        }
        if (!nm.empty() && nm[0] == '^') { negated = !negated; nm = nm.substr(1); }
        auto loose = [](const std::string &x) {
            std::string k;
            for (char ch : x) if (ch != '_' && ch != '-' && ch != ' ') k.push_back((char)std::tolower((unsigned char)ch));
            return k;
        };
        std::string prop, val = nm;
        size_t eq = nm.find_first_of("=:");
        if (eq != std::string::npos) {
            if (eq > 0 && nm[eq - 1] == '!') return fail("unsupported: \\p{name!=value}");
            prop = loose(nm.substr(0, eq));
            val = nm.substr(eq + 1);
        }
        std::string k = loose(val);
        CpSet t;
        const UniTable *tab = nullptr;
        if (prop.empty()) {
            if (k == "any") t = {{0, kMaxCp}};
            else if (k == "ascii") t = {{0, 0x7F}};
            else if (k == "assigned") { add_table(t, *find_table(0, "cn")); normalize(t); t = negate(t); }
            else if ((tab = find_table(2, k == "alpha" ? "alphabetic" : k == "space" || k == "wspace" ? "whitespace" : k == "lower" ? "lowercase" : k == "upper" ? "uppercase" : k == "joinc" ? "joincontrol" : k))) add_table(t, *tab);
            else if ((tab = find_table(0, k))) add_table(t, *tab);
            else if ((tab = find_table(1, k))) add_table(t, *tab);
            else return fail("unsupported: Unicode property \\p{" + nm + "}");
        } else if (prop == "gc" || prop == "generalcategory") {
            if (!(tab = find_table(0, k))) return fail("unknown general category " + val);
            add_table(t, *tab);
        } else if (prop == "sc" || prop == "script") {
            if (!(tab = find_table(1, k))) return fail("unknown script " + val);
            add_table(t, *tab);
        } else {
            return fail("unsupported: Unicode property \\p{" + nm + "}");
        }
        normalize(t);
        if (f.i) fold_unicode(t);
        if (negated) t = negate(t);
        s.insert(s.end(), t.begin(), t.end());
        return true;
    }

    // \d \s \w and their negations; false (after fail) when the negation cannot be written without the u flag
    bool perl_class(char c, CpSet &s, const Flags &f) {
        CpSet t;
        const char lower = (char)(c | 0x20);
        if (f.u) {
            if (lower == 'd') add_table(t, *find_table(0, "nd"));
            else if (lower == 's') add_table(t, *find_table(2, "whitespace"));
            else t = word_unicode();
        } else {
            if (lower == 'd') t = {{'0', '9'}};
            else if (lower == 's') t = {{'\t', '\r'}, {' ', ' '}};
            else t = word_set(false);
        }
        normalize(t);
        if (c != lower) {
            if (!f.u) return fail("pattern can match invalid UTF-8 (negated ASCII class without the u flag)");
            t = negate(t);
        }
        s.insert(s.end(), t.begin(), t.end());
        return true;
    }

    static int hexv(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    // An escape that denotes one code point (the backslash has been consumed, p[pos] is the escape char); -1 on error.
    long escape_char(const Flags &f) {
        char c = peek();
        pos++;
        switch (c) {
            case 'n': return '\n';
            case 'r': return '\r';
            case 't': return '\t';
            case 'f': return 0x0C;
            case 'v': return 0x0B;
            case 'a': return 0x07;
            case 'x': case 'u': case 'U': {
                if (eof()) { fail("incomplete hexadecimal escape"); return -1; }
                unsigned long v = 0;
                bool byte_form = false;  // regex-syntax: only the fixed two-digit \xHH is a byte literal when the u flag is off
                if (peek() == '{') {
                    pos++;
                    int n = 0;
                    while (!eof() && peek() != '}') {
                        int h = hexv(peek());
                        if (h < 0) { fail("invalid hex digit"); return -1; }
                        v = v * 16 + (unsigned)h;
                        if (v > kMaxCp) { fail("hex escape out of range"); return -1; }
                        pos++; n++;
                    }
                    if (eof() || n == 0) { fail("unclosed hexadecimal escape"); return -1; }
                    pos++;
                } else {
                    const int digits = c == 'x' ? 2 : c == 'u' ? 4 : 8;
                    byte_form = c == 'x';
                    for (int k = 0; k < digits; k++) {
                        if (eof()) { fail("incomplete hexadecimal escape"); return -1; }
                        int h = hexv(peek());
                        if (h < 0) { fail("invalid hex digit"); return -1; }
                        v = v * 16 + (unsigned)h;
                        pos++;
                    }
                }
                if (v > kMaxCp || (v >= 0xD800 && v <= 0xDFFF)) { fail("hex escape is not a scalar value"); return -1; }
                if (!f.u && v > 0x7F && byte_form) { fail("pattern can match invalid UTF-8 (byte escape without the u flag)"); return -1; }
                return (long)v;
            }
            default:
                if ((c >= '!' && c <= '/') || (c >= ':' && c <= '@') || (c >= '[' && c <= '`') || (c >= '{' && c <= '~') || c == ' ') {
                    if (c == '<' || c == '>') { fail("unsupported: \\< \\> word-edge assertions"); return -1; }
                    return (uint8_t)c;
                }
                fail(std::string("unrecognized escape sequence \\") + c);
                return -1;
        }
    }

    bool parse_posix(CpSet &s) {
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
        CpSet t;
        auto range = [&](uint32_t a, uint32_t z) { t.push_back({a, z}); };
        if (name == "alnum") { range('0', '9'); range('a', 'z'); range('A', 'Z'); }
        else if (name == "alpha") { range('a', 'z'); range('A', 'Z'); }
        else if (name == "ascii") range(0, 127);
        else if (name == "blank") { range(' ', ' '); range('\t', '\t'); }
        else if (name == "cntrl") { range(0, 31); range(127, 127); }
        else if (name == "digit") range('0', '9');
        else if (name == "graph") range('!', '~');
        else if (name == "lower") range('a', 'z');
        else if (name == "print") range(' ', '~');
        else if (name == "punct") { range('!', '/'); range(':', '@'); range('[', '`'); range('{', '~'); }
        else if (name == "space") { range('\t', '\r'); range(' ', ' '); }
        else if (name == "upper") range('A', 'Z');
        else if (name == "word") { range('0', '9'); range('a', 'z'); range('A', 'Z'); range('_', '_'); }
        else if (name == "xdigit") { range('0', '9'); range('a', 'f'); range('A', 'F'); }
        else { fail("unknown POSIX class " + name); return true; }
        normalize(t);
        if (neg) t = negate(t);  // (over all scalar values: regex-syntax negates the Unicode class)
        s.insert(s.end(), t.begin(), t.end());
        return true;
    }

    AstP parse_class(const Flags &f) {
        // at '['
        pos++;
        bool neg = false;
        skip_space(f);
        if (!eof() && peek() == '^') { neg = true; pos++; }
        CpSet s;
        bool first = true;
        bool posix_negated = false;
        for (;;) {
            skip_space(f);
            if (eof()) { fail("unclosed character class"); return nullptr; }
            char c = peek();
            if (c == ']' && !first) { pos++; break; }
            first = false;
            long lo = -1;
            if (c == '[') {
                if (pos + 1 < p.size() && p[pos + 1] == ':') {
                    size_t at = pos;
                    if (parse_posix(s)) {
                        if (!err.empty()) return nullptr;
                        if (p[at + 2] == '^') posix_negated = true;
                        continue;
                    }
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
                if (strchr("dDwWsS", e)) { pos++; if (!perl_class(e, s, f)) return nullptr; continue; }
                if (e == 'p' || e == 'P') { if (!unicode_property(s, f)) return nullptr; continue; }
                if (e == 'b') { pos++; lo = 0x08; }  // inside a class \b is backspace
                else { lo = escape_char(f); if (lo < 0) return nullptr; }
            } else {
                uint32_t ch = 0;
                if (!next_char(ch)) return nullptr;
                lo = ch;
            }
            long hi = lo;
            if (pos + 1 < p.size() && peek() == '-' && p[pos + 1] != ']') {
                pos++;
                char c2 = peek();
                if (c2 == '\\') {
                    pos++;
                    if (eof()) { fail("incomplete escape"); return nullptr; }
                    if (strchr("dDwWsSpP", peek())) { fail("invalid class range"); return nullptr; }
                    hi = escape_char(f);
                    if (hi < 0) return nullptr;
                } else if (c2 == '[') {
                    fail("unsupported: nested character class");
                    return nullptr;
                } else {
                    uint32_t ch = 0;
                    if (!next_char(ch)) return nullptr;
                    hi = ch;
                }
                if (hi < lo) { fail("invalid character class range"); return nullptr; }
            }
            if (!f.u && hi > 0x7F) { fail("unsupported: non-ASCII class member without the u flag"); return nullptr; }
            s.push_back({(uint32_t)lo, (uint32_t)hi});
        }
        if (f.i) { if (f.u) fold_unicode(s); else fold_ascii(s); }
        normalize(s);
        if (!f.u && (neg || posix_negated)) { fail("pattern can match invalid UTF-8 (negated class without the u flag)"); return nullptr; }
        if (neg) s = negate(s);
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
                case 'u': f.u = on; break;
                case 'R': if (on) return fail("unsupported: (?R) CRLF mode"); break;
                case 'x': f.x = on; break;
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
            if (!f.u) { fail("pattern can match invalid UTF-8 (. without the u flag)"); return nullptr; }
            CpSet s{{0, kMaxCp}};
            if (!f.s) s = {{0, '\n' - 1}, {'\n' + 1, kMaxCp}};
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
                CpSet s;
                if (!perl_class(e, s, f)) return nullptr;
                return mk_set(s);  // (regex-syntax does not fold Perl classes: they are closed under simple case folding)
            }
            if (e == 'A') { pos++; return mk_assert(AKind::StartText); }
            if (e == 'z') { pos++; return mk_assert(AKind::EndText); }
            if (e == 'b') {
                pos++;
                if (!eof() && peek() == '{') { fail("unsupported: \\b{start} / \\b{end} word-edge assertions"); return nullptr; }
                return mk_assert(f.u ? AKind::WordB : AKind::WordBAscii);
            }
            if (e == 'B') { pos++; return mk_assert(f.u ? AKind::NotWordB : AKind::NotWordBAscii); }
            if (e == 'p' || e == 'P') {
                CpSet s;
                if (!unicode_property(s, f)) return nullptr;
                return mk_set(s);
            }
            if (e >= '0' && e <= '9') { fail("backreferences are not supported"); return nullptr; }
            long b = escape_char(f);
            if (b < 0) return nullptr;
            return mk_char((uint32_t)b, f);
        }
        if (c == '*' || c == '+' || c == '?') { fail("repetition operator missing expression"); return nullptr; }
        if (c == '{') { fail("repetition operator missing expression"); return nullptr; }
        uint32_t ch = 0;
        if (!next_char(ch)) return nullptr;
        return mk_char(ch, f);
    }

    bool parse_counted(int &mn, int &mx, const Flags &f) {
        // at '{'
        size_t save = pos;
        pos++;
        auto num = [&](int &v) -> bool {
            skip_space(f);
            size_t b = pos;
            long x = 0;
            while (!eof() && peek() >= '0' && peek() <= '9') {
                x = x * 10 + (peek() - '0');
                if (x > 100000) return false;
                pos++;
            }
            if (pos == b) return false;
            v = (int)x;
            skip_space(f);
            return true;
        };
        if (!num(mn)) { pos = save; return fail("invalid counted repetition"); }
        if (!eof() && peek() == '}') { pos++; mx = mn; return true; }
        if (eof() || peek() != ',') { pos = save; return fail("invalid counted repetition"); }
        pos++;
        skip_space(f);
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
        for (;;) {
            skip_space(f);
            if (eof()) break;
            char c = peek();
            int mn = 0, mx = 0;
            if (c == '*') { mn = 0; mx = -1; pos++; }
            else if (c == '+') { mn = 1; mx = -1; pos++; }
            else if (c == '?') { mn = 0; mx = 1; pos++; }
            else if (c == '{') { if (!parse_counted(mn, mx, f)) return nullptr; }
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
        for (;;) {
            skip_space(f);
            if (eof() || peek() == '|' || peek() == ')') break;
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
    enum Op : uint8_t { Char, Split, Jmp, Assert, Match } op;
    uint32_t x = 0, y = 0;  // Char: set index; Split: two targets; Jmp: x
    AKind ak = AKind::StartText;
};
struct RegexProg {
    std::vector<Inst> code;
    std::vector<CpSet> sets;
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
                Inst i; i.op = Inst::Char; i.x = (uint32_t)pr.sets.size();
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

// The haystack as units: scalar values of its well-formed UTF-8 sequences; every other byte is a unit of its own, 0x110000 + byte
// (in no class, a word character to nobody, and \b / \B are false next to it).
constexpr uint32_t kIllFormed = 0x110000;
static void decode_units(std::string_view h, std::vector<uint32_t> &out) {
    out.clear();
    out.reserve(h.size());
    size_t i = 0;
    const size_t n = h.size();
    auto cont = [&](size_t k) { return k < n && ((uint8_t)h[k] & 0xC0) == 0x80; };
    while (i < n) {
        const uint8_t b0 = (uint8_t)h[i];
        if (b0 < 0x80) { out.push_back(b0); i++; continue; }
        uint32_t v = 0;
        size_t len = 0;
        if (b0 >= 0xC2 && b0 <= 0xDF && cont(i + 1)) { len = 2; v = ((b0 & 0x1Fu) << 6) | ((uint8_t)h[i + 1] & 0x3Fu); }
        else if (b0 >= 0xE0 && b0 <= 0xEF && cont(i + 1) && cont(i + 2)) {
            v = ((b0 & 0x0Fu) << 12) | (((uint8_t)h[i + 1] & 0x3Fu) << 6) | ((uint8_t)h[i + 2] & 0x3Fu);
            len = (v >= 0x800 && !(v >= 0xD800 && v <= 0xDFFF)) ? 3 : 0;
        } else if (b0 >= 0xF0 && b0 <= 0xF4 && cont(i + 1) && cont(i + 2) && cont(i + 3)) {
            v = ((b0 & 0x07u) << 18) | (((uint8_t)h[i + 1] & 0x3Fu) << 12) | (((uint8_t)h[i + 2] & 0x3Fu) << 6) | ((uint8_t)h[i + 3] & 0x3Fu);
            len = (v >= 0x10000 && v <= kMaxCp) ? 4 : 0;
        }
        if (len) { out.push_back(v); i += len; }
        else { out.push_back(kIllFormed + b0); i++; }
    }
}

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
    if (!ast || !ps.err.empty()) {
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

bool Regex::is_match(std::string_view hay) const {
    const RegexProg &pr = *prog;
    const size_t ninst = pr.code.size();
    SparseSet cur(ninst), nxt(ninst);
    std::vector<uint32_t> stack;
    std::vector<uint32_t> h;
    decode_units(hay, h);
    const size_t n = h.size();
    const CpSet &W = word_unicode();
    auto ascii_word = [](uint32_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; };

    // follow zero-width instructions from pc at boundary `at` (between units h[at-1] and h[at])
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
                    const bool ill = (at > 0 && h[at - 1] >= kIllFormed) || (at < n && h[at] >= kIllFormed);
                    switch (in.ak) {
                        case AKind::StartText: ok = at == 0; break;
                        case AKind::EndText: ok = at == n; break;
                        case AKind::StartLine: ok = at == 0 || h[at - 1] == '\n'; break;
                        case AKind::EndLine: ok = at == n || h[at] == '\n'; break;
                        case AKind::WordB: case AKind::NotWordB: {
                            const bool pw = at > 0 && contains(W, h[at - 1]), nw = at < n && contains(W, h[at]);
                            ok = !ill && ((pw != nw) == (in.ak == AKind::WordB));
                            break;
                        }
                        case AKind::WordBAscii: case AKind::NotWordBAscii: {
                            const bool pw = at > 0 && ascii_word(h[at - 1]), nw = at < n && ascii_word(h[at]);
                            ok = !ill && ((pw != nw) == (in.ak == AKind::WordBAscii));
                            break;
                        }
                    }
                    if (ok) stack.push_back(pc + 1);
                    break;
                }
                case Inst::Match: matched = true; break;
                case Inst::Char: break;
            }
        }
        return matched;
    };

    cur.clear();
    for (size_t at = 0;; at++) {
        // unanchored search: a new thread starts at every unit boundary
        if (addthread(cur, 0, at)) return true;
        if (at == n) break;
        const uint32_t c = h[at];
        nxt.clear();
        bool matched = false;
        for (size_t k = 0; k < cur.n; k++) {
            const Inst &in = pr.code[cur.dense[k]];
            if (in.op == Inst::Char && contains(pr.sets[in.x], c)) {
                if (addthread(nxt, cur.dense[k] + 1, at + 1)) matched = true;
            }
        }
        if (matched) return true;
        std::swap(cur, nxt);
    }
    return false;
}

}  // namespace oracle
