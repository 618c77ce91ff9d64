// dfa.cpp — multi-pattern DFA construction for one (field, group): the table the scan kernel walks.
//
// All string predicates of the rule language over one request field — contains / starts_with /
// ends_with / == / matches(regex) / membership in a string list — become patterns of one joint
// automaton, so the field's bytes are read ONCE regardless of how many rules mention it
// ("each input byte is counted once regardless of rule count").
//
// Construction: Thompson NFA per pattern (continuation-passing, no patch lists) -> subset
// construction over "core sets" with zero-width assertions resolved at byte boundaries:
//   state  D = (core C, prev-kind pk, delayed emits Ed)
//   entry  eager closure from C ∪ {pattern starts} under pk: EPS and the assertions decidable from the
//          previous byte alone (\A, (?m)^) are taken; assertions needing the NEXT byte ($, \z, \b, \B,
//          (?m)$) park as pending. Accepts reached here are emitted on entering D.
//   step   on byte class cl (kind nk): pending assertions that hold under (pk, nk) are released,
//          closure continues; accepts reached now become the successor's delayed emits; the successor
//          core is move(live BYTE states, cl).
//   end    pending assertions are released with next = END; accepts reached are D's end-emits.
// For literal-only pattern sets no assertion needs the next byte, Ed is always empty and the result is
// exactly the Aho-Corasick DFA of the literals.
#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <unordered_map>

#include "program.h"
#include "utf8.h"

namespace pwaf {

namespace {

// The automaton's ALPHABET. A table whose patterns are all made of bytes (literals, ASCII classes) reads bytes: symbols 0..255, as
// in rounds 1-4. As soon as a pattern of the table holds a class with members beyond ASCII (RNode::UCLASS: `.`, [^x], \\w \\s \\d,
// \\p{..}, (?i)s ...) the table is in SCALAR MODE: the text is read one scalar value at a time — an ASCII byte is its own symbol; a
// scalar beyond ASCII is ONE symbol, the ATOM (symbol 256 + k) of the partition of the non-ASCII scalar values that the table's sets
// (and, when a \\b is there, the word characters) induce; continuation bytes are skipped, a byte that begins no well-formed sequence
// is the atom ILL that no class holds. The walkers decode at the lead byte (rare: csrc/utf8.h) and look the atom's class up in a
// two-stage table (DfaGroup::umap); everything else — tables, kernels, the bigram filter over raw bytes — is what it was for ASCII
// traffic. (The round's first version compiled every class into UTF-8 byte automata instead: exact too, but \\w alone is a 300-state
// decoder that every context of a pattern needs a copy of — 19 gap passes instead of 6 for the 1k-rule set, rows of 140 byte classes,
// 6.0 -> 5.2 G requests/s on traffic that holds no byte above 0x7F at all.)
using SymSet = std::bitset<512>;
static constexpr int kAtom0 = 256, kMaxAtoms = 240, kAtomIll = 0;  // atom 0 = a byte outside every well-formed sequence
enum : uint8_t { N_EPS, N_BYTE, N_ASSERT, N_ACCEPT };
struct NState {
    uint8_t type;
    AssertKind ak;
    int out = -1, out2 = -1;
    int cls = -1;   // N_BYTE: distinct symbol-set id
    int atom = -1;  // N_ACCEPT: local atom id
};
static inline bool is_word_assert(AssertKind a) { return a == A_WORD_B || a == A_NOT_WORD_B || a == A_WORD_B_ASCII || a == A_NOT_WORD_B_ASCII; }

// kinds of a symbol (what assertions ask of the previous / next one): K_WORD = an ASCII word character, K_UWORD = a word character
// beyond ASCII (a word character to \\b, none to (?-u:\\b)), K_ILL = an ill-formed byte (\\b and \\B are both false next to it: D17)
enum : uint8_t { K_EDGE = 0 /* START as prev, END as next */, K_OTHER = 1, K_WORD = 2, K_NEWLINE = 3, K_UWORD = 4, K_ILL = 5, K_COUNT = 6 };

// The partition of the scalar values beyond ASCII into atoms (scalar mode).
struct ScalarAlphabet {
    bool on = false;
    std::vector<uint32_t> starts;  // elementary intervals [starts[i], starts[i + 1]) over [0x80, 0x110000)
    std::vector<uint16_t> atom;    // ... and the atom each belongs to (>= 1; 0 is ILL)
    std::vector<uint8_t> word;     // per atom: a \\w scalar (meaningful when the table holds a Unicode \\b)
    int n_atoms = 1;
    std::vector<CpSet> sets;       // the distinct non-ASCII parts that were partitioned
    void build(const std::vector<CpSet> &all, bool with_words) {
        on = true;
        auto beyond_ascii = [](const CpSet &x) {
            CpSet t;
            for (auto &r : x)
                if (r.second >= 0x80) t.push_back({std::max<uint32_t>(r.first, 0x80), r.second});
            return t;
        };
        auto intern = [&](const CpSet &t) -> int {
            if (t.empty()) return -1;
            auto it = std::find(sets.begin(), sets.end(), t);
            if (it == sets.end()) { sets.push_back(t); return (int)sets.size() - 1; }
            return (int)(it - sets.begin());
        };
        for (auto &x : all) intern(beyond_ascii(x));
        const int words = with_words ? intern(beyond_ascii(unicode_word_set(true))) : -1;
        std::vector<uint32_t> cuts{0x80, 0x110000};
        for (auto &x : sets)
            for (auto &r : x) { cuts.push_back(r.first); cuts.push_back(r.second + 1); }
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        std::map<std::vector<bool>, int> ids;
        auto in = [](const CpSet &x, uint32_t c) {
            auto it = std::upper_bound(x.begin(), x.end(), CpRange{c, 0xFFFFFFFFu});
            return it != x.begin() && (it - 1)->second >= c;
        };
        word.assign(1, 0);
        for (size_t i = 0; i + 1 < cuts.size(); i++) {
            std::vector<bool> sig(sets.size());
            for (size_t k = 0; k < sets.size(); k++) sig[k] = in(sets[k], cuts[i]);
            auto it = ids.find(sig);
            if (it == ids.end()) {
                it = ids.emplace(sig, n_atoms++).first;
                word.push_back(words >= 0 && sig[(size_t)words]);
            }
            if (!atom.empty() && atom.back() == it->second) continue;  // (the same atom goes on)
            starts.push_back(cuts[i]);
            atom.push_back((uint16_t)it->second);
        }
        starts.push_back(0x110000);
    }
    int atom_of(uint32_t cp) const { return atom[(size_t)(std::upper_bound(starts.begin(), starts.end(), cp) - starts.begin()) - 1]; }
    SymSet symbols(const RNode &n) const {  // a UCLASS as a set of symbols: its ASCII members and the atoms it holds (an atom lies inside a set or outside it)
        SymSet o;
        for (int c = 0; c < 128; c++) o[(size_t)c] = n.cls[(size_t)c];
        for (auto &r : n.ucls) {
            if (r.second < 0x80) continue;
            size_t i = (size_t)(std::upper_bound(starts.begin(), starts.end(), std::max<uint32_t>(r.first, 0x80)) - starts.begin()) - 1;
            for (; i + 1 < starts.size() && starts[i] <= r.second; i++) o.set((size_t)(kAtom0 + atom[i]));
        }
        return o;
    }
};

struct Nfa {
    std::vector<NState> st;
    std::vector<SymSet> sets;
    std::map<std::string, int> set_ids;
    std::vector<int> entries;  // per-pattern entry states
    const ScalarAlphabet *alpha = nullptr;
    bool uses_word = false, uses_uword = false, uses_line = false;
    size_t cap = 0;
    bool overflow = false;
    // Counted repetitions of ONE byte class (`.{0,40}`, `[^>]{0,64}`, `\s{1,8}`) unroll into a chain of optional
    // copies; chain_rank[s] = copies still available at entry state s of chain chain_id[s] (0 = not a chain entry).
    // A thread with more copies left accepts a superset of what a thread of the same chain with fewer copies (or the
    // chain's continuation, chain_tail) accepts from the same position on, which is what lets prune_core() keep the
    // subset construction polynomial for `a.{0,n}b` instead of tracking every subset of the n gap positions.
    std::vector<int> chain_id, chain_rank;
    std::vector<std::vector<int>> chain_tail;  // per NFA state: chains whose continuation it is
    int n_chains = 0;
    void tag(int s, int chain, int rank) {
        if ((size_t)s >= chain_id.size()) { chain_id.resize(st.size(), -1); chain_rank.resize(st.size(), 0); }
        chain_id[s] = chain;
        chain_rank[s] = rank;
    }

    int add(NState s) {
        if (st.size() >= cap) { overflow = true; return 0; }
        st.push_back(s);
        return (int)st.size() - 1;
    }
    int set_id(const SymSet &b) {
        std::string k = b.to_string();
        auto it = set_ids.find(k);
        if (it != set_ids.end()) return it->second;
        int id = (int)sets.size();
        sets.push_back(b);
        set_ids.emplace(std::move(k), id);
        return id;
    }
    // compile `n` so that a match continues at state `next`; returns the entry state
    int build(const RNode &n, int next) {
        if (overflow) return next;
        switch (n.k) {
            case RNode::EMPTY: return next;
            case RNode::CLASS: {
                NState s{N_BYTE, A_TEXT_START};
                SymSet b;
                // (scalar mode reads no raw byte above 0x7F: what is left of a byte class up there — a literal that is not UTF-8 — matches nothing)
                for (int c = 0; c < (alpha->on ? 128 : 256); c++) b[(size_t)c] = n.cls[(size_t)c];
                s.cls = set_id(b);
                s.out = next;
                return add(s);
            }
            case RNode::UCLASS: {
                NState s{N_BYTE, A_TEXT_START};
                s.cls = set_id(alpha->symbols(n));
                s.out = next;
                return add(s);
            }
            case RNode::CAT: {
                int cur = next;
                for (size_t k = n.kids.size(); k-- > 0;) cur = build(*n.kids[k], cur);
                return cur;
            }
            case RNode::ALT: {
                int cur = build(*n.kids.back(), next);
                for (size_t k = n.kids.size() - 1; k-- > 0;) {
                    NState s{N_EPS, A_TEXT_START};
                    s.out = build(*n.kids[k], next);
                    s.out2 = cur;
                    cur = add(s);
                }
                return cur;
            }
            case RNode::REPEAT: {
                const RNode &k = *n.kids[0];
                int cur;
                if (n.rmax < 0) {
                    NState loop{N_EPS, A_TEXT_START};
                    int L = add(loop);
                    int body = build(k, L);
                    st[L].out = body;
                    st[L].out2 = next;
                    if (n.rmin == 0) cur = L;
                    else {
                        cur = body;  // x+ : enter through the body, loop state decides
                        for (int c = 1; c < n.rmin; c++) cur = build(k, cur);
                    }
                } else {
                    cur = next;
                    bool chain = (k.k == RNode::CLASS || k.k == RNode::UCLASS) && n.rmax - n.rmin >= 2;
                    int id = chain ? n_chains++ : -1;
                    for (int c = n.rmin; c < n.rmax; c++) {
                        NState opt{N_EPS, A_TEXT_START};
                        opt.out = build(k, cur);
                        opt.out2 = next;
                        cur = add(opt);
                        if (chain && !overflow) tag(cur, id, c - n.rmin + 1);
                    }
                    if (chain && !overflow) {
                        if ((size_t)next >= chain_tail.size()) chain_tail.resize(st.size());
                        chain_tail[next].push_back(id);
                    }
                    for (int c = 0; c < n.rmin; c++) cur = build(k, cur);
                }
                return cur;
            }
            case RNode::ASSERT: {
                NState s{N_ASSERT, n.ak};
                s.out = next;
                if (is_word_assert(n.ak)) uses_word = true;
                if (n.ak == A_WORD_B || n.ak == A_NOT_WORD_B) uses_uword = true;
                if (n.ak == A_LINE_START || n.ak == A_LINE_END) uses_line = true;
                return add(s);
            }
        }
        return next;
    }
};


static inline bool needs_next(AssertKind a) { return a == A_TEXT_END || a == A_LINE_END || is_word_assert(a); }
static inline bool holds(AssertKind a, uint8_t pk, uint8_t nk) {
    // pk: K_EDGE = start of text; nk: K_EDGE = end of text
    switch (a) {
        case A_TEXT_START: return pk == K_EDGE;
        case A_TEXT_END: return nk == K_EDGE;
        case A_LINE_START: return pk == K_EDGE || pk == K_NEWLINE;
        case A_LINE_END: return nk == K_EDGE || nk == K_NEWLINE;
        // \\b \\B: the regex crate's Unicode-aware assertions look at SCALAR VALUES — the symbols of a table in scalar mode; under (?-u)
        // only ASCII word characters count. Next to an ill-formed byte neither holds.
        case A_WORD_B: case A_NOT_WORD_B: {
            if (pk == K_ILL || nk == K_ILL) return false;
            const bool pw = pk == K_WORD || pk == K_UWORD, nw = nk == K_WORD || nk == K_UWORD;
            return (pw != nw) == (a == A_WORD_B);
        }
        case A_WORD_B_ASCII: case A_NOT_WORD_B_ASCII: {
            if (pk == K_ILL || nk == K_ILL) return false;
            return ((pk == K_WORD) != (nk == K_WORD)) == (a == A_WORD_B_ASCII);
        }
    }
    return false;
}

struct DKey {
    std::vector<int> core;
    std::vector<uint16_t> delayed;
    uint8_t pk;
    bool operator<(const DKey &o) const {
        if (pk != o.pk) return pk < o.pk;
        if (core != o.core) return core < o.core;
        return delayed < o.delayed;
    }
};

struct Builder {
    const Nfa &nfa;
    std::vector<uint32_t> mark;
    uint32_t stamp = 0;
    std::vector<int> stack;  // also used directly by build_dfa's filtered phase-A walk

    std::vector<int> best, touched;  // per chain: highest rank present in the core being pruned

    explicit Builder(const Nfa &n) : nfa(n), mark(n.st.size(), 0), best(n.n_chains, 0) {}

    // Drop the states of a (sorted, unique) core that another state of the same core subsumes: lower-ranked entries of
    // a counted-class chain, and the chain's continuation state when any entry of that chain is present (its closure
    // contains the continuation). The language of the DFA state is unchanged; only its identity becomes canonical.
    void prune_core(std::vector<int> &core) {
        if (!nfa.n_chains) return;
        for (int s : core)
            if (nfa.chain_id[s] >= 0) {
                int c = nfa.chain_id[s];
                if (!best[c]) touched.push_back(c);
                best[c] = std::max(best[c], nfa.chain_rank[s]);
            }
        if (touched.empty()) return;
        size_t w = 0;
        for (int s : core) {
            bool keep = true;
            if (nfa.chain_id[s] >= 0) keep = nfa.chain_rank[s] == best[nfa.chain_id[s]];
            else for (int c : nfa.chain_tail[s]) if (best[c] > 0) { keep = false; break; }
            if (keep) core[w++] = s;
        }
        for (int c : touched) best[c] = 0;
        touched.clear();
        core.resize(w);
    }

    // Closure from `seeds` (already-unvisited check by stamp). When nk_known is false, assertions that need
    // the next byte are parked in `pending`; otherwise every assertion is decided under (pk, nk).
    void closure(const std::vector<int> &seeds, uint8_t pk, bool nk_known, uint8_t nk, std::vector<int> &live, std::vector<uint16_t> &accepts,
                 std::vector<int> *pending) {
        stack.clear();
        for (int s : seeds) stack.push_back(s);
        while (!stack.empty()) {
            int s = stack.back();
            stack.pop_back();
            if (mark[s] == stamp) continue;
            mark[s] = stamp;
            const NState &n = nfa.st[s];
            switch (n.type) {
                case N_EPS:
                    if (n.out2 >= 0) stack.push_back(n.out2);
                    if (n.out >= 0) stack.push_back(n.out);
                    break;
                case N_BYTE: live.push_back(s); break;
                case N_ACCEPT: accepts.push_back((uint16_t)n.atom); break;
                case N_ASSERT:
                    if (!nk_known && needs_next(n.ak)) { pending->push_back(s); break; }
                    if (holds(n.ak, pk, nk)) stack.push_back(n.out);
                    break;
            }
        }
    }
};

}  // namespace

// Scalar mode reads scalar values, so the byte literals of contains / starts_with / == ... (rx_literal: one CLASS per byte) are turned
// into scalars too: a run of single-byte classes that spells one well-formed sequence becomes the UCLASS of that scalar. (A byte above
// 0x7F that is part of no such run — a literal that is not UTF-8: no Rust str can hold it — matches nothing in scalar mode.)
static RNodeP scalarize(const RNodeP &n) {
    if (!n) return n;
    if (n->k == RNode::CAT) {
        auto single = [](const RNodeP &x) -> int { return (x->k == RNode::CLASS && x->cls.count() == 1) ? (int)x->cls._Find_first() : -1; };
        std::vector<RNodeP> kids;
        for (size_t i = 0; i < n->kids.size();) {
            const int b0 = single(n->kids[i]);
            const int len = b0 >= 0xF0 ? 4 : b0 >= 0xE0 ? 3 : b0 >= 0xC2 ? 2 : 0;
            bool done = false;
            if (len && i + (size_t)len <= n->kids.size()) {
                uint32_t v = (uint32_t)b0 & (0x7Fu >> len);
                bool ok = true;
                for (int k = 1; k < len && ok; k++) {
                    const int b = single(n->kids[i + (size_t)k]);
                    ok = b >= 0x80 && b <= 0xBF;
                    v = (v << 6) | ((uint32_t)b & 0x3Fu);
                }
                if (ok && !((len == 3 && v < 0x800) || (len == 4 && v < 0x10000) || v > 0x10FFFF || (v >= 0xD800 && v <= 0xDFFF))) {
                    kids.push_back(rx_scalars(CpSet{{v, v}}));
                    i += (size_t)len;
                    done = true;
                }
            }
            if (!done) kids.push_back(scalarize(n->kids[i++]));
        }
        auto o = std::make_shared<RNode>(*n);
        o->kids = std::move(kids);
        return o;
    }
    if (n->kids.empty()) return n;
    auto o = std::make_shared<RNode>(*n);
    for (auto &k : o->kids) k = scalarize(k);
    return o;
}
static void collect_uclasses(const RNode &n, std::vector<CpSet> &out, bool &uword) {
    if (n.k == RNode::UCLASS) out.push_back(n.ucls);
    if (n.k == RNode::ASSERT && (n.ak == A_WORD_B || n.ak == A_NOT_WORD_B)) uword = true;
    for (auto &k : n.kids) collect_uclasses(*k, out, uword);
}
static bool has_uclass(const RNode &n) {  // ... or a Unicode-aware \\b / \\B: both ask what a SCALAR VALUE is
    if (n.k == RNode::UCLASS || (n.k == RNode::ASSERT && (n.ak == A_WORD_B || n.ak == A_NOT_WORD_B))) return true;
    for (auto &k : n.kids) if (has_uclass(*k)) return true;
    return false;
}
static bool has_high_single_bytes(const RNode &n) {  // a literal byte above 0x7F (what scalarize() would turn into a scalar)
    if (n.k == RNode::CLASS && n.cls.count() == 1 && n.cls._Find_first() >= 0x80) return true;
    for (auto &k : n.kids) if (has_high_single_bytes(*k)) return true;
    return false;
}

bool build_dfa(const std::vector<ScanPattern> &pats, uint32_t max_states, uint32_t max_table_bytes, DfaGroup &out, std::string &err) {
    Nfa nfa;
    nfa.cap = 400000;
    // ---- the alphabet: bytes, or (a class beyond ASCII somewhere) scalar values ----
    ScalarAlphabet alpha;
    std::vector<RNodeP> rxs;
    {
        bool scalar = false;
        for (auto &p : pats) scalar = scalar || has_uclass(*p.rx);
        for (auto &p : pats) rxs.push_back(scalar && has_high_single_bytes(*p.rx) ? scalarize(p.rx) : p.rx);
        if (scalar) {
            std::vector<CpSet> sets;
            bool uword = false;
            for (auto &r : rxs) collect_uclasses(*r, sets, uword);
            alpha.build(sets, uword);
            if (alpha.n_atoms > kMaxAtoms) {
                err = "too many distinct classes of non-ASCII scalar values in one table";
                return false;
            }
        }
    }
    nfa.alpha = &alpha;
    for (size_t k = 0; k < pats.size(); k++) {
        NState acc{N_ACCEPT, A_TEXT_START};
        acc.atom = (int)k;
        int a = nfa.add(acc);
        size_t before = nfa.st.size();
        nfa.entries.push_back(nfa.build(*rxs[k], a));
        if (nfa.overflow || nfa.st.size() - before > 20000) {
            err = "pattern too large (more than 20000 NFA states)";
            return false;
        }
    }
    nfa.chain_id.resize(nfa.st.size(), -1);
    nfa.chain_rank.resize(nfa.st.size(), 0);
    nfa.chain_tail.resize(nfa.st.size());
    // ---- symbol classes: symbols are equivalent when no set and no assertion kind tells them apart ----
    // symbols: 0..255 the bytes; scalar mode: kAtom0 + k the atoms (bytes 0x80..0xBF are then continuation bytes — skipped: the
    // class CONT, whose transitions all stay — and bytes 0xC0..0xFF lead bytes, which the walkers replace by the scalar's atom)
    const int n_sym = alpha.on ? kAtom0 + alpha.n_atoms : 256;
    auto kind_of = [&](int b) -> uint8_t {
        if (b >= kAtom0) {
            if (b - kAtom0 == kAtomIll) return nfa.uses_word ? K_ILL : K_OTHER;
            return (nfa.uses_uword && alpha.word[(size_t)(b - kAtom0)]) ? K_UWORD : K_OTHER;
        }
        bool w = (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_';
        if (nfa.uses_word && w) return K_WORD;
        if (nfa.uses_line && b == '\n') return K_NEWLINE;
        return K_OTHER;
    };
    std::vector<int> cls_of((size_t)n_sym);
    int n_cls = 0;
    {
        // the first split: kind, and in scalar mode what a symbol IS (an ASCII byte, a continuation byte, a lead byte, an atom) — the
        // walkers treat these differently, so no class may mix them
        auto role = [&](int b) { return !alpha.on ? 0 : b >= kAtom0 ? 3 : b >= 0xC0 ? 2 : b >= 0x80 ? 1 : 0; };
        std::map<std::pair<int, int>, int> first;
        for (int b = 0; b < n_sym; b++) {
            auto key = std::make_pair((int)kind_of(b), role(b));
            auto it = first.find(key);
            if (it == first.end()) it = first.emplace(key, n_cls++).first;
            cls_of[(size_t)b] = it->second;
        }
        for (size_t si = 0; si < nfa.sets.size(); si++) {
            const SymSet &s = nfa.sets[si];
            std::map<std::pair<int, bool>, int> split;
            int next_id = 0;
            std::vector<int> nc((size_t)n_sym);
            for (int b = 0; b < n_sym; b++) {
                auto key = std::make_pair(cls_of[(size_t)b], (bool)s[(size_t)b]);
                auto it = split.find(key);
                if (it == split.end()) it = split.emplace(key, next_id++).first;
                nc[(size_t)b] = it->second;
            }
            cls_of = nc;
            n_cls = next_id;
        }
    }
    std::vector<int> rep((size_t)n_cls, -1);  // representative symbol per class
    for (int b = 0; b < n_sym; b++) if (rep[(size_t)cls_of[(size_t)b]] < 0) rep[(size_t)cls_of[(size_t)b]] = b;
    // classes of the continuation / lead BYTES of scalar mode: no transition of theirs is ever computed — they stay
    std::vector<uint8_t> cls_stays((size_t)n_cls, 0);
    if (alpha.on)
        for (int c = 0; c < n_cls; c++) cls_stays[(size_t)c] = rep[(size_t)c] >= 0x80 && rep[(size_t)c] < kAtom0;
    // set membership per class
    std::vector<std::vector<uint8_t>> set_has(nfa.sets.size(), std::vector<uint8_t>((size_t)n_cls));
    for (size_t s = 0; s < nfa.sets.size(); s++)
        for (int c = 0; c < n_cls; c++) set_has[s][(size_t)c] = nfa.sets[s][(size_t)rep[(size_t)c]];
    std::vector<uint8_t> cls_kind((size_t)n_cls);
    for (int c = 0; c < n_cls; c++) cls_kind[(size_t)c] = kind_of(rep[(size_t)c]);
    if (n_cls > 250) { err = "more than 250 symbol classes in one table"; return false; }

    if ((uint64_t)n_cls * 2 > max_table_bytes) { err = "LDS table budget too small"; return false; }
    uint32_t state_cap = std::min<uint64_t>(max_states, (uint64_t)max_table_bytes / (2ull * n_cls));
    state_cap = std::min<uint32_t>(state_cap, kMaxDfaStates);

    // ---- subset construction ----
    // The pattern entry states are seeds of EVERY state's closure (unanchored search restarts at each
    // boundary), so their closure ("root") is computed once per prev-kind and shared.
    Builder bl(nfa);
    struct Root {
        std::vector<uint8_t> in;  // visited in phase A
        std::vector<int> liveA, pending;
        std::vector<uint16_t> accA;
        std::vector<int> liveB[K_COUNT];      // extra live states released under next-kind nk
        std::vector<uint16_t> accB[K_COUNT];  // accepts released under nk
        std::vector<std::vector<int>> move;  // per class: sorted targets of liveA ∪ liveB[kind(c)]
        bool ready = false;
    };
    Root roots[K_COUNT];
    auto get_root = [&](uint8_t pk) -> Root & {
        Root &r = roots[pk];
        if (r.ready) return r;
        r.ready = true;
        r.in.assign(nfa.st.size(), 0);
        bl.stamp++;
        bl.closure(nfa.entries, pk, false, 0, r.liveA, r.accA, &r.pending);
        for (size_t s = 0; s < nfa.st.size(); s++) r.in[s] = bl.mark[s] == bl.stamp;
        for (uint8_t nk = 0; nk < K_COUNT; nk++) {
            std::vector<int> rel;
            for (int s : r.pending)
                if (holds(nfa.st[s].ak, pk, nk)) rel.push_back(nfa.st[s].out);
            if (rel.empty()) continue;
            bl.stamp++;
            for (int s : r.liveA) bl.mark[s] = bl.stamp;
            bl.closure(rel, pk, true, nk, r.liveB[nk], r.accB[nk], nullptr);
        }
        r.move.assign(n_cls, {});
        for (int c = 0; c < n_cls; c++) {
            std::vector<int> &m = r.move[c];
            for (int s : r.liveA) if (set_has[nfa.st[s].cls][c]) m.push_back(nfa.st[s].out);
            for (int s : r.liveB[cls_kind[c]]) if (set_has[nfa.st[s].cls][c]) m.push_back(nfa.st[s].out);
            std::sort(m.begin(), m.end());
            m.erase(std::unique(m.begin(), m.end()), m.end());
        }
        return r;
    };

    struct DState {
        DKey key;
        std::vector<uint16_t> emits;      // eager accepts ∪ delayed
        std::vector<uint16_t> end_emits;  // accepts released at END
        std::vector<int> next;            // per class
    };
    std::vector<DState> ds;
    std::map<DKey, int> index;
    auto intern = [&](DKey &&k) -> int {
        auto it = index.find(k);
        if (it != index.end()) return it->second;
        int id = (int)ds.size();
        ds.emplace_back();
        ds.back().key = k;
        index.emplace(std::move(k), id);
        return id;
    };
    auto uniq16 = [](std::vector<uint16_t> &v) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    };
    DKey k0;
    k0.pk = K_EDGE;
    intern(std::move(k0));
    std::vector<int> liveA, liveB, pending, rel;
    std::vector<uint16_t> accA, accB, entry_emits;
    for (size_t d = 0; d < ds.size(); d++) {
        if (ds.size() > state_cap) {
            err = "DFA state limit exceeded";
            return false;
        }
        DKey key = ds[d].key;  // copy: ds may reallocate
        Root &root = get_root(key.pk);
        // phase A over the core seeds only (root part is shared); skip what the root closure covers
        liveA.clear(); accA.clear(); pending.clear();
        bl.stamp++;
        uint32_t stampA = bl.stamp;
        {
            std::vector<int> seeds;
            for (int s : key.core) if (!root.in[s]) seeds.push_back(s);
            // closure() does not know about root.in: pre-mark nothing, filter on the fly instead
            bl.stack.clear();
            for (int s : seeds) bl.stack.push_back(s);
            while (!bl.stack.empty()) {
                int s = bl.stack.back();
                bl.stack.pop_back();
                if (bl.mark[s] == stampA || root.in[s]) continue;
                bl.mark[s] = stampA;
                const NState &n = nfa.st[s];
                switch (n.type) {
                    case N_EPS:
                        if (n.out2 >= 0) bl.stack.push_back(n.out2);
                        if (n.out >= 0) bl.stack.push_back(n.out);
                        break;
                    case N_BYTE: liveA.push_back(s); break;
                    case N_ACCEPT: accA.push_back((uint16_t)n.atom); break;
                    case N_ASSERT:
                        if (needs_next(n.ak)) { pending.push_back(s); break; }
                        if (holds(n.ak, key.pk, 0)) bl.stack.push_back(n.out);
                        break;
                }
            }
        }
        entry_emits = accA;
        entry_emits.insert(entry_emits.end(), root.accA.begin(), root.accA.end());
        entry_emits.insert(entry_emits.end(), key.delayed.begin(), key.delayed.end());
        uniq16(entry_emits);
        ds[d].emits = entry_emits;
        ds[d].next.assign(n_cls, (int)d);  // (the classes that stay — continuation and lead bytes of scalar mode — keep this)
        // phase B for next-kind nk: release this state's own pending assertions (the root's are precomputed)
        auto phaseB = [&](uint8_t nk) {
            liveB.clear();
            accB.clear();
            rel.clear();
            for (int s : pending)
                if (holds(nfa.st[s].ak, key.pk, nk)) rel.push_back(nfa.st[s].out);
            if (!rel.empty()) {
                // keep phase-A marks (same stamp): only NEW states are entered. States of the root closure may be
                // re-entered here; duplicates are removed when targets/accepts are sorted.
                std::vector<int> extra;
                bl.closure(rel, key.pk, true, nk, extra, accB, nullptr);
                liveB = extra;
            }
            accB.insert(accB.end(), root.accB[nk].begin(), root.accB[nk].end());
            uniq16(accB);
            std::vector<uint16_t> diff;
            std::set_difference(accB.begin(), accB.end(), entry_emits.begin(), entry_emits.end(), std::back_inserter(diff));
            accB.swap(diff);
        };
        phaseB(K_EDGE);
        ds[d].end_emits = accB;
        for (uint8_t nk : {K_OTHER, K_WORD, K_NEWLINE, K_UWORD, K_ILL}) {
            bool used = false;
            for (int c = 0; c < n_cls; c++) if (cls_kind[c] == nk && !cls_stays[(size_t)c]) used = true;
            if (!used) continue;
            // phase B marks must not leak between next-kinds: restart from the phase-A marks
            bl.stamp++;
            for (int s : liveA) bl.mark[s] = bl.stamp;
            phaseB(nk);
            for (int c = 0; c < n_cls; c++) {
                if (cls_kind[c] != nk || cls_stays[(size_t)c]) continue;
                DKey nkey;
                nkey.pk = nk;
                nkey.delayed = accB;
                nkey.core = root.move[c];
                for (int s : liveA) if (set_has[nfa.st[s].cls][c]) nkey.core.push_back(nfa.st[s].out);
                for (int s : liveB) if (set_has[nfa.st[s].cls][c]) nkey.core.push_back(nfa.st[s].out);
                std::sort(nkey.core.begin(), nkey.core.end());
                nkey.core.erase(std::unique(nkey.core.begin(), nkey.core.end()), nkey.core.end());
                bl.prune_core(nkey.core);
                int t = intern(std::move(nkey));
                ds[d].next[c] = t;
            }
        }
    }
    if (ds.size() > state_cap) {
        err = "DFA state limit exceeded";
        return false;
    }

    // ---- output: states keep their BFS discovery order (start = 0) ----
    uint32_t S = (uint32_t)ds.size();
    out.n_states = S;
    out.n_classes = (uint32_t)n_cls;
    for (int b = 0; b < 256; b++) out.classmap[b] = (uint8_t)cls_of[(size_t)b];
    out.class_stays.assign(cls_stays.begin(), cls_stays.end());
    out.umap = ScalarMap();
    if (alpha.on) {
        // scalar value -> class, two stages of 128 (equal blocks shared): what a walker looks up at a lead byte
        ScalarMap &m = out.umap;
        m.ill_class = (uint8_t)cls_of[(size_t)(kAtom0 + kAtomIll)];
        m.stage1.assign(kScalarBlocks, 0);
        std::map<std::vector<uint8_t>, uint16_t> blocks;
        std::vector<uint8_t> blk(128);
        size_t iv = 0;
        for (uint32_t bi = 0; bi < kScalarBlocks; bi++) {
            for (uint32_t k = 0; k < 128; k++) {
                const uint32_t cp = bi * 128 + k;
                uint8_t c = m.ill_class;  // (ASCII is never looked up; surrogates are never decoded)
                if (cp >= 0x80 && !(cp >= 0xD800 && cp <= 0xDFFF)) {
                    while (alpha.starts[iv + 1] <= cp) iv++;
                    c = (uint8_t)cls_of[(size_t)(kAtom0 + alpha.atom[iv])];
                }
                blk[k] = c;
            }
            auto it = blocks.find(blk);
            if (it == blocks.end()) {
                it = blocks.emplace(blk, (uint16_t)(m.stage2.size() / 128)).first;
                m.stage2.insert(m.stage2.end(), blk.begin(), blk.end());
            }
            m.stage1[bi] = it->second;
        }
    }
    out.trans.assign((size_t)S * n_cls, 0);
    out.emit_off.assign(1, 0);
    out.emit_list.clear();
    out.end_off.assign(1, 0);
    out.end_list.clear();
    for (uint32_t k = 0; k < S; k++) {
        const DState &d = ds[k];
        for (int c = 0; c < n_cls; c++) out.trans[(size_t)k * n_cls + c] = (uint16_t)d.next[c];
        out.emit_list.insert(out.emit_list.end(), d.emits.begin(), d.emits.end());
        out.emit_off.push_back((uint32_t)out.emit_list.size());
        out.end_list.insert(out.end_list.end(), d.end_emits.begin(), d.end_emits.end());
        out.end_off.push_back((uint32_t)out.end_list.size());
    }
    out.atoms.clear();
    for (auto &p : pats) out.atoms.push_back(p.atom);
    out.n_local = (uint32_t)pats.size();
    return true;
}

// `.*`, `[^>]+`, and counted gaps wide enough to multiply states with the other patterns of a table (`.{0,40}`)
static bool is_wide_gap(const RNode &n) {
    return n.k == RNode::REPEAT && (n.rmax < 0 || n.rmax - n.rmin >= 8) && (n.kids[0]->k == RNode::CLASS || n.kids[0]->k == RNode::UCLASS) && n.kids[0]->cls.count() >= 64;
}

bool has_wide_gap(const RNode &n) {
    if (is_wide_gap(n)) return true;
    for (auto &k : n.kids) if (has_wide_gap(*k)) return true;
    return false;
}

static bool rx_nullable(const RNode &n) {
    switch (n.k) {
        case RNode::EMPTY: case RNode::ASSERT: return true;
        case RNode::CLASS: case RNode::UCLASS: return false;
        case RNode::CAT: for (auto &k : n.kids) if (!rx_nullable(*k)) return false; return true;
        case RNode::ALT: for (auto &k : n.kids) if (rx_nullable(*k)) return true; return false;
        case RNode::REPEAT: return n.rmin == 0 || rx_nullable(*n.kids[0]);
    }
    return true;
}

uint32_t rx_min_len(const RNode &n) {
    switch (n.k) {
        case RNode::EMPTY: case RNode::ASSERT: return 0;
        case RNode::CLASS: return 1;
        case RNode::UCLASS: return n.cls.any() ? 1 : 2;  // (bytes: a scalar beyond ASCII takes at least two)
        case RNode::CAT: { uint32_t s = 0; for (auto &k : n.kids) s += rx_min_len(*k); return s; }
        case RNode::ALT: { uint32_t m = 0xFFFFFFFFu; for (auto &k : n.kids) m = std::min(m, rx_min_len(*k)); return m == 0xFFFFFFFFu ? 0 : m; }
        case RNode::REPEAT: return (uint32_t)n.rmin * rx_min_len(*n.kids[0]);
    }
    return 0;
}

RNodeP gap_prefilter(const RNodeP &rx) {
    if (!rx || rx->k != RNode::CAT) return nullptr;
    size_t g = 0;
    while (g < rx->kids.size() && !is_wide_gap(*rx->kids[g])) {
        if (has_wide_gap(*rx->kids[g])) return nullptr;  // a gap nested deeper comes first: no clean prefix
        g++;
    }
    if (g == rx->kids.size()) return nullptr;
    std::vector<RNodeP> pre(rx->kids.begin(), rx->kids.begin() + (long)g);
    while (!pre.empty() && pre.back()->k == RNode::ASSERT) pre.pop_back();  // trailing look-arounds only strengthen X
    RNodeP x = rx_cat(pre);
    if (pre.empty() || rx_nullable(*x)) return nullptr;  // would fire everywhere: useless as a filter
    return x;
}

// the device image of a ScalarMap (what utf8_class reads): stage 1, then stage 2
std::vector<uint8_t> scalar_map_image(const ScalarMap &m) {
    std::vector<uint8_t> img;
    if (!m.on()) return img;
    img.resize(kUmapStage2 + m.stage2.size());
    memcpy(img.data(), m.stage1.data(), kUmapStage2);
    memcpy(img.data() + kUmapStage2, m.stage2.data(), m.stage2.size());
    return img;
}
uint32_t dfa_class_at(const DfaGroup &g, const uint8_t *bytes, size_t i, size_t n) {
    const uint32_t b = bytes[i];
    if (b < 0x80u || !g.umap.on()) return g.classmap[b];
    if (b < 0xC0u) {
        // a continuation byte: the table's own class (stays) when a well-formed sequence holds it, else ill-formed (utf8.h: utf8_cont_covered)
        const uint32_t back = (uint32_t)std::min<size_t>(3, i);
        uint32_t prev = 0, self_next = 0;
        for (uint32_t d = 1; d <= back; d++) prev |= (uint32_t)bytes[i - d] << (8 * (d - 1));
        for (size_t k = 0; k < 4 && i + k < n; k++) self_next |= (uint32_t)bytes[i + k] << (8 * k);
        return utf8_cont_covered(prev, self_next, back, (uint32_t)std::min<size_t>(n - i, 0xFFFFu)) ? g.classmap[b] : g.umap.ill_class;
    }
    uint32_t next = 0;
    for (size_t k = 1; k < 4 && i + k < n; k++) next |= (uint32_t)bytes[i + k] << (8 * (k - 1));
    // (a walk that looks classes up often builds the image once; this is the simple form for the test hooks)
    const uint32_t len = b >= 0xF0u ? 4u : b >= 0xE0u ? 3u : 2u;
    if (b < 0xC2u || b > 0xF4u || n - i < len) return g.umap.ill_class;
    const uint32_t c1 = next & 0xFFu, c2 = (next >> 8) & 0xFFu, c3 = (next >> 16) & 0xFFu;
    if ((c1 & 0xC0u) != 0x80u || (len > 2u && (c2 & 0xC0u) != 0x80u) || (len > 3u && (c3 & 0xC0u) != 0x80u)) return g.umap.ill_class;
    uint32_t cp = len == 2u ? ((b & 0x1Fu) << 6) | (c1 & 0x3Fu) : len == 3u ? ((b & 0x0Fu) << 12) | ((c1 & 0x3Fu) << 6) | (c2 & 0x3Fu) : ((b & 0x07u) << 18) | ((c1 & 0x3Fu) << 12) | ((c2 & 0x3Fu) << 6) | (c3 & 0x3Fu);
    if ((len == 3u && (cp < 0x800u || (cp >= 0xD800u && cp <= 0xDFFFu))) || (len == 4u && (cp < 0x10000u || cp > 0x10FFFFu))) return g.umap.ill_class;
    return g.umap.stage2[(size_t)g.umap.stage1[cp >> 7] * 128 + (cp & 127u)];
}

void dfa_run_host(const DfaGroup &g, const uint8_t *bytes, size_t n, std::vector<uint16_t> &out_atoms) {
    uint32_t s = 0;
    auto emit = [&](uint32_t st) {
        for (uint32_t k = g.emit_off[st]; k < g.emit_off[st + 1]; k++) out_atoms.push_back(g.emit_list[k]);
    };
    emit(s);
    for (size_t i = 0; i < n; i++) {
        const uint32_t before = s;
        s = g.trans[(size_t)s * g.n_classes + dfa_class_at(g, bytes, i, n)];
        if (s != before || !g.class_stays[dfa_class_at(g, bytes, i, n)]) emit(s);
    }
    for (uint32_t k = g.end_off[s]; k < g.end_off[s + 1]; k++) out_atoms.push_back(g.end_list[k]);
    std::sort(out_atoms.begin(), out_atoms.end());
    out_atoms.erase(std::unique(out_atoms.begin(), out_atoms.end()), out_atoms.end());
}

}  // namespace pwaf
