// pattern.cpp — regex front-end of the device compiler: pattern text -> RNode tree.
//
// The reference evaluates regexes through `bel` -> `regex 1.12.2` (Cargo.lock:1694-1700). Only
// is_match (a boolean) reaches the rule result (pingoo/rules.rs:47), so leftmost-first ordering and
// laziness are irrelevant and a DFA over the pattern's language is exact. Supported subset and the
// reasons for every exclusion: DESIGN.md §3.4. Operator-stack parser (one pass, no recursion).
#include <algorithm>
#include <cstring>

#include "program.h"

namespace pwaf {

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
    }
}
std::string rx_key(const RNode &n) {
    std::string o;
    key_rec(n, o);
    return o;
}

namespace {

struct RxFlags {
    bool icase = false, multiline = false, dotall = false;
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
    static bool word(int c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; }
    static int hex(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        c |= 0x20;
        return (c >= 'a' && c <= 'f') ? c - 'a' + 10 : -1;
    }
    static void fold(ByteSet &s) {
        for (int c = 'A'; c <= 'Z'; c++)
            if (s[c] || s[c + 32]) { s.set(c); s.set(c + 32); }
    }
    static void named(const char *name, size_t len, ByteSet &out, bool &ok) {
        struct { const char *n; const char *ranges; } tbl[] = {
            {"alnum", "09AZaz"}, {"alpha", "AZaz"}, {"ascii", "\x01\x7f"}, {"blank", "  \t\t"}, {"cntrl", "\x01\x1f\x7f\x7f"},
            {"digit", "09"}, {"graph", "!~"}, {"lower", "az"}, {"print", " ~"}, {"punct", "!/:@[`{~"},
            {"space", "\t\r  "}, {"upper", "AZ"}, {"word", "09AZaz__"}, {"xdigit", "09AFaf"},
        };
        ok = false;
        for (auto &e : tbl) {
            if (strlen(e.n) == len && !memcmp(e.n, name, len)) {
                for (const char *r = e.ranges; r[0]; r += 2)
                    for (int c = (unsigned char)r[0]; c <= (unsigned char)r[1]; c++) out.set(c);
                if (!strcmp(e.n, "ascii") || !strcmp(e.n, "cntrl")) out.set(0);  // NUL cannot sit in the range string
                ok = true;
                return;
            }
        }
    }
    // \p{Name} / \pX / \P{Name} / \p{^Name}: Unicode general categories (and the script Latin), RESTRICTED TO ASCII — the reference's fields
    // are ASCII by construction (HeaderValue::to_str, http::Uri), where \p{L} is [A-Za-z], \p{N} is [0-9] and so on; under (?i) a cased
    // category takes its other case too (the regex crate folds classes). i points at the 'p' / 'P'. Other property names: unsupported.
    bool unicode_class(ByteSet &out, bool icase) {
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
        std::string key;
        for (char ch : name)
            if (ch != '_' && ch != ' ' && ch != '-') key += (char)tolower((unsigned char)ch);
        struct Cat { const char *names; const char *ranges; };  // names separated by '|', ranges as byte pairs
        static const Cat cats[] = {
            {"l|letter|alphabetic|alpha|latin|latn|lc|casedletter", "AZaz"}, {"lu|uppercaseletter|uppercase|upper", "AZ"}, {"ll|lowercaseletter|lowercase|lower", "az"},
            {"n|number|nd|decimalnumber|digit", "09"}, {"p|punctuation|punct", "!#%*,/:;?@[]__{{}}"}, {"pc|connectorpunctuation", "__"}, {"pd|dashpunctuation", "--"},
            {"ps|openpunctuation", "(([[{{"}, {"pe|closepunctuation", "))]]}}"}, {"po|otherpunctuation", "!#%'**,,./:;?@\\\\"}, {"s|symbol", "$$++<>^^``||~~"},
            {"sc|currencysymbol", "$$"}, {"sm|mathsymbol", "++<>||~~"}, {"sk|modifiersymbol", "^^``"}, {"z|separator|zs|spaceseparator", "  "},
            {"cc|control|cntrl|c|other", "\x01\x1f\x7f\x7f"}, {"ascii", "\x01\x7f"}, {"any", "\x01\xff"},
            {"lt|titlecaseletter|lm|modifierletter|lo|otherletter|m|mark|mn|mc|me|nl|letternumber|no|othernumber|pi|initialpunctuation|pf|finalpunctuation|so|othersymbol|zl|lineseparator|zp|paragraphseparator|cf|format|cs|surrogate|co|privateuse|cn|unassigned", ""},
        };
        const Cat *hit = nullptr;
        for (const Cat &c : cats) {
            const char *q = c.names;
            while (*q && !hit) {
                const char *e = strchr(q, '|');
                const size_t len = e ? (size_t)(e - q) : strlen(q);
                if (len == key.size() && !memcmp(q, key.data(), len)) hit = &c;
                q += len + (e ? 1 : 0);
            }
            if (hit) break;
        }
        if (!hit) return unsupported("Unicode property \\p{" + name + "} (only general categories, ASCII-restricted)");
        ByteSet t;
        for (const char *r = hit->ranges; r[0]; r += 2)
            for (int c = (unsigned char)r[0]; c <= (unsigned char)r[1]; c++) t.set((size_t)c);
        if (key == "cc" || key == "control" || key == "cntrl" || key == "c" || key == "other" || key == "ascii" || key == "any") t.set(0);  // NUL cannot sit in the range string
        if (key == "p" || key == "punctuation" || key == "punct") for (char c : std::string("\"&'()-.\\")) t.set((size_t)(unsigned char)c);
        if (icase) fold(t);
        if (neg) t.flip();
        out |= t;
        return true;
    }
    static void shorthand(char k, ByteSet &out) {
        ByteSet t;
        char lower = (char)(k | 0x20);
        if (lower == 'd') for (int c = '0'; c <= '9'; c++) t.set(c);
        if (lower == 'w') for (int c = 0; c < 128; c++) if (word(c)) t.set(c);
        if (lower == 's') for (int c : {9, 10, 11, 12, 13, 32}) t.set(c);
        if (k != lower) t.flip();
        out |= t;
    }

    // escape that denotes one byte; i points just past the backslash
    int one_byte_escape() {
        char c = p[i++];
        switch (c) {
            case 'a': return 7;
            case 'f': return 12;
            case 't': return 9;
            case 'n': return 10;
            case 'r': return 13;
            case 'v': return 11;
            case 'x': {
                unsigned v = 0;
                if (i < p.size() && p[i] == '{') {
                    size_t j = i + 1;
                    int nd = 0;
                    while (j < p.size() && p[j] != '}') {
                        if (hex(p[j]) < 0) { invalid("invalid hexadecimal digit"); return -1; }
                        v = v * 16 + (unsigned)hex(p[j]);
                        if (v > 0x10FFFF) { invalid("hexadecimal escape out of range"); return -1; }
                        j++; nd++;
                    }
                    if (j >= p.size() || !nd) { invalid("unclosed hexadecimal escape"); return -1; }
                    i = j + 1;
                } else {
                    if (i + 2 > p.size() || hex(p[i]) < 0 || hex(p[i + 1]) < 0) { invalid("invalid hexadecimal escape"); return -1; }
                    v = (unsigned)(hex(p[i]) * 16 + hex(p[i + 1]));
                    i += 2;
                }
                if (v > 0x7F) { unsupported("non-ASCII code point escape"); return -1; }
                return (int)v;
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
        bool negate = false;
        if (i < p.size() && p[i] == '^') { negate = true; i++; }
        ByteSet s;
        bool first = true;
        while (true) {
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
                        ByteSet t;
                        bool ok;
                        named(p.data() + b, j - b, t, ok);
                        if (!ok) return invalid("unknown POSIX class");
                        if (neg) t.flip();
                        s |= t;
                        i = j + 2;
                        continue;
                    }
                }
                return unsupported("nested character class");
            }
            if ((c == '&' || c == '-' || c == '~') && i + 1 < p.size() && p[i + 1] == c) return unsupported("character class set operation");
            int lo;
            if (c == '\\') {
                i++;
                if (i >= p.size()) return invalid("incomplete escape");
                char e = p[i];
                if (strchr("dDwWsS", e)) { shorthand(e, s); i++; continue; }
                if (e == 'p' || e == 'P') { if (!unicode_class(s, fl.icase)) return false; continue; }  // (the crate folds, then negates, per item)
                if (e == 'b') { lo = 8; i++; }
                else { lo = one_byte_escape(); if (lo < 0) return false; }
            } else {
                lo = (unsigned char)c;
                i++;
            }
            int hi = lo;
            if (i + 1 < p.size() && p[i] == '-' && p[i + 1] != ']') {
                i++;
                char c2 = p[i];
                if (c2 == '[') return unsupported("nested character class");
                if (c2 == '\\') {
                    i++;
                    if (i >= p.size()) return invalid("incomplete escape");
                    if (strchr("dDwWsSpP", p[i])) return invalid("invalid character class range");
                    hi = one_byte_escape();
                    if (hi < 0) return false;
                } else {
                    hi = (unsigned char)c2;
                    i++;
                }
                if (hi < lo) return invalid("invalid character class range");
            }
            for (int b = lo; b <= hi; b++) s.set(b);
        }
        if (fl.icase) fold(s);
        if (negate) s.flip();
        push(rx_class(s));
        return true;
    }

    void push(RNodeP n) { st.emplace_back(Item::NODE, std::move(n), RxFlags{}); }

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
            auto number = [&](int &v) {
                size_t b = j;
                long x = 0;
                while (j < p.size() && p[j] >= '0' && p[j] <= '9') {
                    x = x * 10 + (p[j] - '0');
                    if (x > 100000) return false;
                    j++;
                }
                v = (int)x;
                return j > b;
            };
            if (!number(mn)) return invalid("invalid counted repetition");
            if (j < p.size() && p[j] == '}') { mx = mn; j++; }
            else if (j < p.size() && p[j] == ',') {
                j++;
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
                        case 'U': case 'R': break;
                        case 'u': if (!on) return unsupported("(?-u)"); break;
                        case 'x': return unsupported("(?x) verbose mode");
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
        while (i < p.size() && !status) {
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
                    ByteSet s;
                    s.set();
                    if (!fl.dotall) s.reset('\n');
                    push(rx_class(s));
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
                    if (strchr("dDwWsS", e)) { ByteSet s; shorthand(e, s); i++; push(rx_class(s)); break; }
                    if (e == 'A') { i++; push(rx_assert(A_TEXT_START)); break; }
                    if (e == 'z') { i++; push(rx_assert(A_TEXT_END)); break; }
                    if (e == 'b') { i++; push(rx_assert(A_WORD_B)); break; }
                    if (e == 'B') { i++; push(rx_assert(A_NOT_WORD_B)); break; }
                    if (e == 'p' || e == 'P') { ByteSet s; if (unicode_class(s, fl.icase)) push(rx_class(s)); break; }
                    if (e >= '0' && e <= '9') { invalid("backreferences are not supported"); break; }
                    int b = one_byte_escape();
                    if (b < 0) break;
                    ByteSet s;
                    s.set(b);
                    if (fl.icase) fold(s);
                    push(rx_class(s));
                    break;
                }
                default: {
                    i++;
                    ByteSet s;
                    s.set((unsigned char)c);
                    if (fl.icase) fold(s);
                    push(rx_class(s));
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

RNodeP regex_parse(const std::string &pattern, int &status, std::string &err) {
    RxParser ps(pattern);
    RNodeP r = ps.run();
    status = ps.status;
    err = ps.err;
    if (status) return nullptr;
    return r;
}

}  // namespace pwaf
