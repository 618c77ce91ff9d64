// pattern.cpp — regex front-end of the device compiler: pattern text -> RNode tree.
//
// The reference evaluates regexes through `bel` -> `regex 1.12.2` (Cargo.lock:1694-1700). Only
// is_match (a boolean) reaches the rule result (pingoo/rules.rs:47), so leftmost-first ordering and
// laziness are irrelevant and a DFA over the pattern's language is exact. The crate is UNICODE-AWARE by
// default and the reference's strings are Rust str (url / path may carry UTF-8: http 1.3.1,
// Cargo.lock:824-826): classes are sets of scalar values here (`.`, negated classes, \d \s \w, \p{..},
// simple case folding under (?i)); dfa.cpp lowers them to UTF-8 byte automata. Supported subset and the
// reasons for every exclusion: DESIGN.md §3.4. Operator-stack parser (one pass, no recursion).
#include <algorithm>
#include <cstring>

#include "program.h"

namespace pwaf {

static constexpr uint32_t kLastScalar = 0x10FFFF;

RNodeP rx_empty() { return std::make_shared<RNode>(); }
RNodeP rx_class(const ByteSet &s) {
    auto n = std::make_shared<RNode>();
    n->k = RNode::CLASS;
    n->cls = s;
    return n;
}
RNodeP rx_byte(uint8_t c) {
    ByteSet s;
    s.set(c);
    return rx_class(s);
}
RNodeP rx_literal(const std::string &bytes) {
    std::vector<RNodeP> kids;
    for (unsigned char c : bytes) kids.push_back(rx_byte(c));
    return rx_cat(std::move(kids));
}
RNodeP rx_cat(std::vector<RNodeP> kids) {
    if (kids.empty()) return rx_empty();
    if (kids.size() == 1) return kids[0];
    auto n = std::make_shared<RNode>();
    n->k = RNode::CAT;
    n->kids = std::move(kids);
    return n;
}
RNodeP rx_alt(std::vector<RNodeP> kids) {
    if (kids.size() == 1) return kids[0];
    auto n = std::make_shared<RNode>();
    n->k = RNode::ALT;
    n->kids = std::move(kids);
    return n;
}
RNodeP rx_assert(AssertKind k) {
    auto n = std::make_shared<RNode>();
    n->k = RNode::ASSERT;
    n->ak = k;
    return n;
}

static void key_rec(const RNode &n, std::string &o) {
    switch (n.k) {
        case RNode::EMPTY: o += "e"; break;
        case RNode::CLASS: {
            o += "[";
            // run-length over the 256 bits
            int run = -1;
            for (int b = 0; b <= 256; b++) {
                bool on = b < 256 && n.cls[b];
                if (on && run < 0) run = b;
                if (!on && run >= 0) {
                    char buf[16];
                    snprintf(buf, sizeof buf, "%02x-%02x", run, b - 1);
                    o += buf;
                    run = -1;
                }
            }
            o += "]";
            break;
        }
        case RNode::CAT: case RNode::ALT:
            o += n.k == RNode::CAT ? "(." : "(|";
            for (auto &k : n.kids) key_rec(*k, o);
            o += ")";
            break;
        case RNode::REPEAT:
            o += "{" + std::to_string(n.rmin) + "," + std::to_string(n.rmax) + ":";
            key_rec(*n.kids[0], o);
            o += "}";
            break;
        case RNode::ASSERT: o += "@" + std::to_string((int)n.ak); break;
        case RNode::UCLASS: {
            o += "U[";
            char buf[32];
            for (auto &r : n.ucls) { snprintf(buf, sizeof buf, "%x-%x,", r.first, r.second); o += buf; }
            o += "]";
            break;
        }
    }
}
std::string rx_key(const RNode &n) {
    std::string o;
    key_rec(n, o);
    return o;
}

// ---- sets of scalar values ----
void cp_canon(CpSet &s) {
    std::sort(s.begin(), s.end());
    size_t w = 0;
    for (size_t k = 0; k < s.size(); k++) {
        if (w && s[k].first <= s[w - 1].second + 1) s[w - 1].second = std::max(s[w - 1].second, s[k].second);
        else s[w++] = s[k];
    }
    s.resize(w);
    // no surrogates: a Rust char is a scalar value
    CpSet o;
    for (auto &r : s) {
        if (r.first <= 0xD7FF) o.push_back({r.first, std::min<uint32_t>(r.second, 0xD7FF)});
        if (r.second >= 0xE000) o.push_back({std::max<uint32_t>(r.first, 0xE000), r.second});
    }
    s.swap(o);
}
CpSet cp_complement(const CpSet &s) {
    CpSet o;
    uint32_t from = 0;
    for (auto &r : s) {
        if (r.first > from) o.push_back({from, r.first - 1});
        from = r.second + 1;
    }
    if (from <= kLastScalar) o.push_back({from, kLastScalar});
    cp_canon(o);
    return o;
}
CpSet cp_intersect(const CpSet &a, const CpSet &b) {
    CpSet o;
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) {
        const uint32_t lo = std::max(a[i].first, b[j].first), hi = std::min(a[i].second, b[j].second);
        if (lo <= hi) o.push_back({lo, hi});
        if (a[i].second < b[j].second) i++;
        else j++;
    }
    return o;
}

namespace {

#include "unicode_data.inc"

bool cp_has(const CpSet &s, uint32_t c) {
    auto it = std::upper_bound(s.begin(), s.end(), CpRange{c, 0xFFFFFFFFu});
    return it != s.begin() && (it - 1)->second >= c;
}
void cp_add_table(CpSet &s, int kind, const char *name) {
    const size_t len = strlen(name);
    for (const UniTable &t : kUniTables) {
        if (t.kind != kind) continue;
        for (const char *q = t.names; *q;) {
            const char *e = strchr(q, '|');
            const size_t l = e ? (size_t)(e - q) : strlen(q);
            if (l == len && !memcmp(q, name, len)) {
                for (unsigned k = 0; k < t.count; k++) s.push_back({kUniRanges[t.first + k][0], kUniRanges[t.first + k][1]});
                return;
            }
            q += l + (e ? 1 : 0);
        }
    }
}
bool cp_table_exists(int kind, const std::string &name) {
    CpSet t;
    // (an existing table may be empty — Cs over scalar values — so look the name up rather than test the result)
    for (const UniTable &u : kUniTables) {
        if (u.kind != kind) continue;
        for (const char *q = u.names; *q;) {
            const char *e = strchr(q, '|');
            const size_t l = e ? (size_t)(e - q) : strlen(q);
            if (l == name.size() && !memcmp(q, name.data(), l)) return true;
            q += l + (e ? 1 : 0);
        }
    }
    return false;
}
// the members' simple case folding orbits (regex-syntax: case_fold_simple); ASCII-only without the u flag
void cp_fold(CpSet &s, bool unicode) {
    cp_canon(s);
    CpSet extra;
    if (unicode) {
        const size_t n = sizeof kUniFold / sizeof kUniFold[0];
        for (auto &r : s) {
            size_t at = (size_t)(std::lower_bound(kUniFold, kUniFold + n, r.first, [](const unsigned(&e)[2], uint32_t v) { return e[0] < v; }) - kUniFold);
            for (; at < n && kUniFold[at][0] <= r.second; at++) extra.push_back({kUniFold[at][1], kUniFold[at][1]});
        }
    } else {
        for (uint32_t c = 'A'; c <= 'Z'; c++) {
            if (cp_has(s, c)) extra.push_back({c + 32, c + 32});
            if (cp_has(s, c + 32)) extra.push_back({c, c});
        }
    }
    s.insert(s.end(), extra.begin(), extra.end());
    cp_canon(s);
}

struct RxFlags {
    bool icase = false, multiline = false, dotall = false, unicode = true, verbose = false;
};

struct Item {
    enum T { NODE, LPAREN, VBAR } t = NODE;
    RNodeP node;
    RxFlags saved;             // LPAREN: flags to restore at ')'
    bool flag_marker = false;  // NODE produced by a bare (?flags) group: not quantifiable
    Item(T tt, RNodeP n, RxFlags s) : t(tt), node(std::move(n)), saved(s) {}
};

struct RxParser {
    const std::string &p;
    size_t i = 0;
    std::vector<Item> st;
    RxFlags fl;
    int status = 0;  // 0 ok, 1 invalid, 2 unsupported
    std::string err;

    explicit RxParser(const std::string &s) : p(s) {}

    bool invalid(const std::string &m) {
        if (!status) { status = 1; err = m + " (regex offset " + std::to_string(i) + ")"; }
        return false;
    }
    bool unsupported(const std::string &m) {
        if (!status) { status = 2; err = "unsupported regex feature: " + m + " (regex offset " + std::to_string(i) + ")"; }
        return false;
    }
    static int hex(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        c |= 0x20;
        return (c >= 'a' && c <= 'f') ? c - 'a' + 10 : -1;
    }
    // One scalar value of the pattern text (a Rust string) at i; -1 after invalid().
    long take_char() {
        const unsigned char b = (unsigned char)p[i];
        if (b < 0x80) { i++; return b; }
        const int n = b >= 0xF0 ? 4 : b >= 0xE0 ? 3 : b >= 0xC2 ? 2 : 0;
        if (!n || i + (size_t)n > p.size()) { invalid("pattern is not valid UTF-8"); return -1; }
        uint32_t v = b & (0x7Fu >> n);
        for (int k = 1; k < n; k++) {
            const unsigned char c = (unsigned char)p[i + (size_t)k];
            if ((c & 0xC0) != 0x80) { invalid("pattern is not valid UTF-8"); return -1; }
            v = (v << 6) | (c & 0x3Fu);
        }
        if ((n == 3 && v < 0x800) || (n == 4 && v < 0x10000) || v > kLastScalar || (v >= 0xD800 && v <= 0xDFFF)) { invalid("pattern is not valid UTF-8"); return -1; }
        i += (size_t)n;
        return (long)v;
    }
    void skip_verbose() {  // (?x): whitespace and # comments between tokens are syntax
        if (!fl.verbose) return;
        while (i < p.size()) {
            const char c = p[i];
            if (c == ' ' || (c >= '\t' && c <= '\r')) i++;
            else if (c == '#') while (i < p.size() && p[i] != '\n') i++;
            else break;
        }
    }
    static void posix(const char *name, size_t len, CpSet &out, bool &ok) {
        struct { const char *n; const char *ranges; } tbl[] = {
            {"alnum", "09AZaz"}, {"alpha", "AZaz"}, {"ascii", "\x01\x7f"}, {"blank", "  \t\t"}, {"cntrl", "\x01\x1f\x7f\x7f"},
            {"digit", "09"}, {"graph", "!~"}, {"lower", "az"}, {"print", " ~"}, {"punct", "!/:@[`{~"},
            {"space", "\t\r  "}, {"upper", "AZ"}, {"word", "09AZaz__"}, {"xdigit", "09AFaf"},
        };
        ok = false;
        for (auto &e : tbl) {
            if (strlen(e.n) == len && !memcmp(e.n, name, len)) {
                for (const char *r = e.ranges; r[0]; r += 2) out.push_back({(unsigned char)r[0], (unsigned char)r[1]});
                if (!strcmp(e.n, "ascii") || !strcmp(e.n, "cntrl")) out.push_back({0, 0});  // NUL cannot sit in the range string
                ok = true;
                return;
            }
        }
    }
    // \p{Name} / \pX / \P{Name} / \p{^Name} / \p{gc=..} / \p{sc=..}: general categories, scripts and the binary properties the crate's
    // Perl classes are made of, from the Unicode tables (unicode_data.inc). The item is folded under (?i) and then negated
    // (regex-syntax's order). i points at the 'p' / 'P'. Other properties (age, Script_Extensions, ...): unsupported.
    bool unicode_class(CpSet &out) {
        if (!fl.unicode) return invalid("Unicode class without the u flag");
        const bool neg_outer = p[i] == 'P';
        i++;
        std::string name;
        if (i < p.size() && p[i] == '{') {
            const size_t close = p.find('}', i);
            if (close == std::string::npos) return invalid("unterminated \\p{");
            name = p.substr(i + 1, close - i - 1);
            i = close + 1;
        } else if (i < p.size()) {
            name = std::string(1, p[i++]);
        } else {
            return invalid("incomplete \\p");
        }
        bool neg = neg_outer;
        if (!name.empty() && name[0] == '^') { neg = !neg; name.erase(0, 1); }
        auto squash = [](const std::string &x) {
            std::string k;
            for (char ch : x)
                if (ch != '_' && ch != ' ' && ch != '-') k += (char)tolower((unsigned char)ch);
            return k;
        };
        std::string prop, value = name;
        const size_t sep = name.find_first_of("=:");
        if (sep != std::string::npos) {
            if (sep > 0 && name[sep - 1] == '!') return unsupported("\\p{name!=value}");
            prop = squash(name.substr(0, sep));
            value = name.substr(sep + 1);
        }
        const std::string key = squash(value);
        CpSet t;
        if (prop.empty()) {
            static const struct { const char *alias, *table; } bins[] = {{"alphabetic", "alphabetic"}, {"alpha", "alphabetic"}, {"whitespace", "whitespace"}, {"space", "whitespace"},
                                                                          {"wspace", "whitespace"}, {"lowercase", "lowercase"}, {"lower", "lowercase"}, {"uppercase", "uppercase"},
                                                                          {"upper", "uppercase"}, {"joincontrol", "joincontrol"}, {"joinc", "joincontrol"}};
            const char *bin = nullptr;
            for (auto &b : bins) if (key == b.alias) bin = b.table;
            if (key == "any") t.push_back({0, kLastScalar});
            else if (key == "ascii") t.push_back({0, 0x7F});
            else if (key == "assigned") { cp_add_table(t, 0, "cn"); cp_canon(t); t = cp_complement(t); }
            else if (bin) cp_add_table(t, 2, bin);
            else if (cp_table_exists(0, key)) cp_add_table(t, 0, key.c_str());
            else if (cp_table_exists(1, key)) cp_add_table(t, 1, key.c_str());
            else return unsupported("Unicode property \\p{" + name + "}");
        } else if (prop == "gc" || prop == "generalcategory") {
            if (!cp_table_exists(0, key)) return invalid("unknown general category " + value);
            cp_add_table(t, 0, key.c_str());
        } else if (prop == "sc" || prop == "script") {
            if (!cp_table_exists(1, key)) return invalid("unknown script " + value);
            cp_add_table(t, 1, key.c_str());
        } else {
            return unsupported("Unicode property \\p{" + name + "}");
        }
        cp_canon(t);
        if (fl.icase) cp_fold(t, true);
        if (neg) t = cp_complement(t);
        out.insert(out.end(), t.begin(), t.end());
        return true;
    }
    bool shorthand(char k, CpSet &out) {
        CpSet t;
        const char lower = (char)(k | 0x20);
        if (fl.unicode) {
            if (lower == 'd') cp_add_table(t, 0, "nd");
            else if (lower == 's') cp_add_table(t, 2, "whitespace");
            else t = unicode_word_set(true);
        } else {
            if (lower == 'd') t = {{'0', '9'}};
            else if (lower == 's') t = {{9, 13}, {32, 32}};
            else t = unicode_word_set(false);
        }
        cp_canon(t);
        if (k != lower) {
            if (!fl.unicode) return invalid("pattern can match invalid UTF-8: negated ASCII class without the u flag");
            t = cp_complement(t);
        }
        out.insert(out.end(), t.begin(), t.end());
        return true;
    }

    // escape that denotes one scalar value; i points just past the backslash
    long one_char_escape() {
        char c = p[i++];
        switch (c) {
            case 'a': return 7;
            case 'f': return 12;
            case 't': return 9;
            case 'n': return 10;
            case 'r': return 13;
            case 'v': return 11;
            case 'x': case 'u': case 'U': {
                unsigned long v = 0;
                bool byte_form = false;  // only the fixed two-digit \xHH denotes a BYTE without the u flag (regex-syntax: Literal::byte() is Some for HexFixed(X) alone)
                if (i < p.size() && p[i] == '{') {
                    size_t j = i + 1;
                    int nd = 0;
                    while (j < p.size() && p[j] != '}') {
                        if (hex(p[j]) < 0) { invalid("invalid hexadecimal digit"); return -1; }
                        v = v * 16 + (unsigned)hex(p[j]);
                        if (v > kLastScalar) { invalid("hexadecimal escape out of range"); return -1; }
                        j++; nd++;
                    }
                    if (j >= p.size() || !nd) { invalid("unclosed hexadecimal escape"); return -1; }
                    i = j + 1;
                } else {
                    const size_t digits = c == 'x' ? 2 : c == 'u' ? 4 : 8;
                    byte_form = c == 'x';
                    if (i + digits > p.size()) { invalid("invalid hexadecimal escape"); return -1; }
                    for (size_t k = 0; k < digits; k++) {
                        if (hex(p[i + k]) < 0) { invalid("invalid hexadecimal escape"); return -1; }
                        v = v * 16 + (unsigned)hex(p[i + k]);
                    }
                    i += digits;
                }
                if (v > kLastScalar || (v >= 0xD800 && v <= 0xDFFF)) { invalid("hexadecimal escape is not a scalar value"); return -1; }
                // (\x{..}, \uHHHH and \UHHHHHHHH denote the scalar value with or without the u flag: its UTF-8 encoding, like a raw non-ASCII character)
                if (!fl.unicode && v > 0x7F && byte_form) { invalid("pattern can match invalid UTF-8: byte escape without the u flag"); return -1; }
                return (long)v;
            }
            default: break;
        }
        bool punct = (c >= '!' && c <= '/') || (c >= ':' && c <= '@') || (c >= '[' && c <= '`') || (c >= '{' && c <= '~') || c == ' ';
        if (punct) {
            if (c == '<' || c == '>') { unsupported("\\< \\> word-edge assertions"); return -1; }
            return (unsigned char)c;
        }
        invalid(std::string("unrecognized escape sequence \\") + c);
        return -1;
    }

    bool bracket() {
        // p[i] == '['
        i++;
        skip_verbose();
        bool negate = false, posix_negated = false;
        if (i < p.size() && p[i] == '^') { negate = true; i++; }
        CpSet s;
        bool first = true;
        while (true) {
            skip_verbose();
            if (i >= p.size()) return invalid("unclosed character class");
            char c = p[i];
            if (c == ']' && !first) { i++; break; }
            first = false;
            if (c == '[') {
                if (i + 1 < p.size() && p[i + 1] == ':') {
                    size_t j = i + 2;
                    bool neg = j < p.size() && p[j] == '^';
                    if (neg) j++;
                    size_t b = j;
                    while (j < p.size() && p[j] != ':') j++;
                    if (j + 1 < p.size() && p[j + 1] == ']') {
                        CpSet t;
                        bool ok;
                        posix(p.data() + b, j - b, t, ok);
                        if (!ok) return invalid("unknown POSIX class");
                        cp_canon(t);
                        if (neg) { t = cp_complement(t); posix_negated = true; }  // (over all scalar values, as regex-syntax negates the Unicode class)
                        s.insert(s.end(), t.begin(), t.end());
                        i = j + 2;
                        continue;
                    }
                }
                return unsupported("nested character class");
            }
            if ((c == '&' || c == '-' || c == '~') && i + 1 < p.size() && p[i + 1] == c) return unsupported("character class set operation");
            long lo;
            if (c == '\\') {
                i++;
                if (i >= p.size()) return invalid("incomplete escape");
                char e = p[i];
                if (strchr("dDwWsS", e)) { i++; if (!shorthand(e, s)) return false; continue; }
                if (e == 'p' || e == 'P') { if (!unicode_class(s)) return false; continue; }  // (the crate folds, then negates, per item)
                if (e == 'b') { lo = 8; i++; }
                else { lo = one_char_escape(); if (lo < 0) return false; }
            } else {
                lo = take_char();
                if (lo < 0) return false;
            }
            long hi = lo;
            if (i + 1 < p.size() && p[i] == '-' && p[i + 1] != ']') {
                i++;
                char c2 = p[i];
                if (c2 == '[') return unsupported("nested character class");
                if (c2 == '\\') {
                    i++;
                    if (i >= p.size()) return invalid("incomplete escape");
                    if (strchr("dDwWsSpP", p[i])) return invalid("invalid character class range");
                    hi = one_char_escape();
                    if (hi < 0) return false;
                } else {
                    hi = take_char();
                    if (hi < 0) return false;
                }
                if (hi < lo) return invalid("invalid character class range");
            }
            if (!fl.unicode && hi > 0x7F) return unsupported("non-ASCII class member without the u flag");
            s.push_back({(uint32_t)lo, (uint32_t)hi});
        }
        if (fl.icase) cp_fold(s, fl.unicode);
        cp_canon(s);
        if (!fl.unicode && (negate || posix_negated)) return invalid("pattern can match invalid UTF-8: negated class without the u flag");
        if (negate) s = cp_complement(s);
        push(rx_scalars(s));
        return true;
    }

    void push(RNodeP n) { st.emplace_back(Item::NODE, std::move(n), RxFlags{}); }
    void push_char(uint32_t c) {
        CpSet s{{c, c}};
        if (fl.icase) cp_fold(s, fl.unicode);
        push(rx_scalars(s));
    }

    // concatenates the NODE items above the nearest marker into one node
    void collapse_cat() {
        size_t b = st.size();
        while (b > 0 && st[b - 1].t == Item::NODE) b--;
        std::vector<RNodeP> kids;
        for (size_t k = b; k < st.size(); k++) kids.push_back(st[k].node);
        st.erase(st.begin() + (long)b, st.end());
        push(rx_cat(std::move(kids)));
    }
    // after collapse_cat: folds "a VBAR b VBAR c" above the nearest LPAREN into one ALT node
    void collapse_alt() {
        std::vector<RNodeP> alts;
        while (!st.empty() && st.back().t != Item::LPAREN) {
            if (st.back().t == Item::NODE) alts.push_back(st.back().node);
            st.pop_back();
        }
        std::reverse(alts.begin(), alts.end());
        push(rx_alt(std::move(alts)));
    }

    bool quantify() {
        char c = p[i];
        int mn = 0, mx = -1;
        if (c == '*') { i++; }
        else if (c == '+') { mn = 1; i++; }
        else if (c == '?') { mx = 1; i++; }
        else {
            // '{'
            size_t j = i + 1;
            auto blanks = [&]() { if (fl.verbose) while (j < p.size() && (p[j] == ' ' || (p[j] >= '\t' && p[j] <= '\r'))) j++; };
            auto number = [&](int &v) {
                blanks();
                size_t b = j;
                long x = 0;
                while (j < p.size() && p[j] >= '0' && p[j] <= '9') {
                    x = x * 10 + (p[j] - '0');
                    if (x > 100000) return false;
                    j++;
                }
                v = (int)x;
                const bool any = j > b;
                blanks();
                return any;
            };
            if (!number(mn)) return invalid("invalid counted repetition");
            if (j < p.size() && p[j] == '}') { mx = mn; j++; }
            else if (j < p.size() && p[j] == ',') {
                j++;
                blanks();
                if (j < p.size() && p[j] == '}') { mx = -1; j++; }
                else {
                    if (!number(mx)) return invalid("invalid counted repetition");
                    if (j >= p.size() || p[j] != '}') return invalid("unclosed counted repetition");
                    j++;
                    if (mx < mn) return invalid("invalid repetition range");
                }
            } else {
                return invalid("invalid counted repetition");
            }
            i = j;
        }
        if (i < p.size() && p[i] == '?') i++;  // lazy marker: same language
        if (st.empty() || st.back().t != Item::NODE || st.back().flag_marker) return invalid("repetition operator missing expression");
        auto r = std::make_shared<RNode>();
        r->k = RNode::REPEAT;
        r->rmin = mn;
        r->rmax = mx;
        r->kids.push_back(st.back().node);
        st.back().node = r;
        return true;
    }

    bool group_open() {
        // p[i] == '('
        i++;
        RxFlags saved = fl;
        if (i < p.size() && p[i] == '?') {
            i++;
            if (i >= p.size()) return invalid("unclosed group");
            char d = p[i];
            if (d == 'P' || d == '<') {
                if (d == 'P') i++;
                if (i >= p.size() || p[i] != '<') return invalid("invalid group syntax");
                i++;
                if (i < p.size() && (p[i] == '=' || p[i] == '!')) return invalid("look-around is not supported");
                size_t b = i;
                while (i < p.size() && p[i] != '>') i++;
                if (i >= p.size() || i == b) return invalid("invalid capture group name");
                i++;
            } else if (d == '=' || d == '!') {
                return invalid("look-around is not supported");
            } else {
                bool on = true, any = false;
                RxFlags nf = fl;
                while (true) {
                    if (i >= p.size()) return invalid("unclosed flag group");
                    char f = p[i];
                    if (f == ')') {
                        if (!any) return invalid("empty flag group");
                        i++;
                        fl = nf;  // bare (?flags): applies to the rest of the enclosing group
                        st.emplace_back(Item::NODE, rx_empty(), RxFlags{});
                        st.back().flag_marker = true;
                        return true;
                    }
                    if (f == ':') { i++; break; }
                    i++;
                    any = true;
                    switch (f) {
                        case '-': on = false; break;
                        case 'i': nf.icase = on; break;
                        case 'm': nf.multiline = on; break;
                        case 's': nf.dotall = on; break;
                        case 'U': break;
                        case 'R': if (on) return unsupported("(?R) CRLF mode"); break;
                        case 'u': nf.unicode = on; break;
                        case 'x': nf.verbose = on; break;
                        default: return invalid(std::string("unrecognized flag ") + f);
                    }
                }
                st.emplace_back(Item::LPAREN, nullptr, saved);
                fl = nf;
                return true;
            }
        }
        st.emplace_back(Item::LPAREN, nullptr, saved);
        return true;
    }

    bool group_close() {
        i++;
        collapse_cat();
        bool found = false;
        for (auto &it : st) if (it.t == Item::LPAREN) found = true;
        if (!found) return invalid("unopened group");
        collapse_alt();
        // stack: ... LPAREN NODE
        RNodeP g = st.back().node;
        st.pop_back();
        fl = st.back().saved;
        st.pop_back();
        push(g);
        return true;
    }

    RNodeP run() {
        if (p.size() > 4096) { unsupported("pattern longer than 4096 bytes"); return nullptr; }
        int depth = 0;
        while (!status) {
            skip_verbose();
            if (i >= p.size()) break;
            char c = p[i];
            switch (c) {
                case '(':
                    if (++depth > 200) { invalid("nesting too deep"); break; }
                    group_open();
                    if (!st.empty() && st.back().t == Item::NODE) depth--;  // it was a bare flag group
                    break;
                case ')': depth--; group_close(); break;
                case '|':
                    i++;
                    collapse_cat();
                    st.emplace_back(Item::VBAR, nullptr, RxFlags{});
                    break;
                case '[': bracket(); break;
                case '.': {
                    i++;
                    if (!fl.unicode) { invalid("pattern can match invalid UTF-8: . without the u flag"); break; }
                    CpSet s{{0, kLastScalar}};
                    if (!fl.dotall) s = {{0, 9}, {11, kLastScalar}};
                    cp_canon(s);
                    push(rx_scalars(s));
                    break;
                }
                case '^': i++; push(rx_assert(fl.multiline ? A_LINE_START : A_TEXT_START)); break;
                case '$': i++; push(rx_assert(fl.multiline ? A_LINE_END : A_TEXT_END)); break;
                case '*': case '+': case '?': quantify(); break;
                case '{': quantify(); break;
                case '\\': {
                    i++;
                    if (i >= p.size()) { invalid("incomplete escape"); break; }
                    char e = p[i];
                    if (strchr("dDwWsS", e)) { CpSet s; i++; if (shorthand(e, s)) { cp_canon(s); push(rx_scalars(s)); } break; }
                    if (e == 'A') { i++; push(rx_assert(A_TEXT_START)); break; }
                    if (e == 'z') { i++; push(rx_assert(A_TEXT_END)); break; }
                    if (e == 'b') {
                        i++;
                        if (i < p.size() && p[i] == '{') { unsupported("\\b{start} / \\b{end} word-edge assertions"); break; }
                        push(rx_assert(fl.unicode ? A_WORD_B : A_WORD_B_ASCII));
                        break;
                    }
                    if (e == 'B') { i++; push(rx_assert(fl.unicode ? A_NOT_WORD_B : A_NOT_WORD_B_ASCII)); break; }
                    if (e == 'p' || e == 'P') { CpSet s; if (unicode_class(s)) { cp_canon(s); push(rx_scalars(s)); } break; }
                    if (e >= '0' && e <= '9') { invalid("backreferences are not supported"); break; }
                    long b = one_char_escape();
                    if (b < 0) break;
                    push_char((uint32_t)b);
                    break;
                }
                default: {
                    long ch = take_char();
                    if (ch >= 0) push_char((uint32_t)ch);
                }
            }
        }
        if (status) return nullptr;
        collapse_cat();
        for (auto &it : st) if (it.t == Item::LPAREN) { invalid("unclosed group"); return nullptr; }
        // fold top-level alternation
        std::vector<RNodeP> alts;
        for (auto &it : st) if (it.t == Item::NODE) alts.push_back(it.node);
        return rx_alt(std::move(alts));
    }
};

}  // namespace

RNodeP rx_scalars(const CpSet &s) {
    auto n = std::make_shared<RNode>();
    bool beyond = false;
    for (auto &r : s) {
        for (uint32_t c = r.first; c <= std::min<uint32_t>(r.second, 0x7F); c++) n->cls.set(c);
        if (r.second > 0x7F) beyond = true;
    }
    if (!beyond) {
        n->k = RNode::CLASS;
        return n;
    }
    n->k = RNode::UCLASS;
    n->ucls = s;
    return n;
}

const CpSet &unicode_word_set(bool unicode) {
    static const CpSet ascii{{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}};
    static const CpSet uni = [] {
        CpSet w;
        cp_add_table(w, 2, "alphabetic");
        cp_add_table(w, 0, "m");
        cp_add_table(w, 0, "nd");
        cp_add_table(w, 0, "pc");
        cp_add_table(w, 2, "joincontrol");
        cp_canon(w);
        return w;
    }();
    return unicode ? uni : ascii;
}

// The scalar values >= 0x80 of `s` as UTF-8: sequences of byte ranges, each matching exactly the encodings of one run of code
// points (the construction of utf8-ranges / RE2: split a range at the encoding-length boundaries, then wherever a lower byte
// does not span its whole 0x80..0xBF, so that every position of a sequence is a contiguous byte range).
void utf8_sequences(const CpSet &s, std::vector<std::vector<std::pair<uint8_t, uint8_t>>> &out) {
    auto enc = [](uint32_t c, uint8_t *b) -> int {
        if (c < 0x800) { b[0] = (uint8_t)(0xC0 | (c >> 6)); b[1] = (uint8_t)(0x80 | (c & 0x3F)); return 2; }
        if (c < 0x10000) { b[0] = (uint8_t)(0xE0 | (c >> 12)); b[1] = (uint8_t)(0x80 | ((c >> 6) & 0x3F)); b[2] = (uint8_t)(0x80 | (c & 0x3F)); return 3; }
        b[0] = (uint8_t)(0xF0 | (c >> 18)); b[1] = (uint8_t)(0x80 | ((c >> 12) & 0x3F)); b[2] = (uint8_t)(0x80 | ((c >> 6) & 0x3F)); b[3] = (uint8_t)(0x80 | (c & 0x3F));
        return 4;
    };
    std::vector<CpRange> work;
    for (auto &r : s) {
        if (r.second < 0x80) continue;
        work.push_back({std::max<uint32_t>(r.first, 0x80), r.second});
    }
    while (!work.empty()) {
        CpRange r = work.back();
        work.pop_back();
        // encoding-length boundaries
        bool split = false;
        for (uint32_t edge : {0x7FFu, 0xFFFFu}) {
            if (r.first <= edge && r.second > edge) {
                work.push_back({edge + 1, r.second});
                work.push_back({r.first, edge});
                split = true;
                break;
            }
        }
        if (split) continue;
        // continuation-byte alignment: the low k*6 bits must run 0..all-ones over the whole range, or lo and hi share what is above
        for (int k = 1; k < 4 && !split; k++) {
            const uint32_t m = (1u << (6 * k)) - 1;
            if ((r.first & ~m) != (r.second & ~m)) {
                if ((r.first & m) != 0) {
                    work.push_back({(r.first | m) + 1, r.second});
                    work.push_back({r.first, r.first | m});
                    split = true;
                } else if ((r.second & m) != m) {
                    work.push_back({r.second & ~m, r.second});
                    work.push_back({r.first, (r.second & ~m) - 1});
                    split = true;
                }
            }
        }
        if (split) continue;
        uint8_t a[4], b[4];
        const int n = enc(r.first, a);
        enc(r.second, b);
        std::vector<std::pair<uint8_t, uint8_t>> seq;
        for (int k = 0; k < n; k++) seq.push_back({a[k], b[k]});
        out.push_back(std::move(seq));
    }
}

RNodeP regex_parse(const std::string &pattern, int &status, std::string &err) {
    RxParser ps(pattern);
    RNodeP r = ps.run();
    status = ps.status;
    err = ps.err;
    if (status) return nullptr;
    return r;
}

}  // namespace pwaf
