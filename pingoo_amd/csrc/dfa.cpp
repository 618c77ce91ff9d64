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

namespace pwaf {

namespace {

// N_SCALAR: consumes one SCALAR VALUE of a set with members beyond ASCII (RNode::UCLASS); exists only while a pattern is being built —
// lower_scalars() replaces every one by the UTF-8 byte sequences of its set before the subset construction sees the automaton.
enum : uint8_t { N_EPS, N_BYTE, N_ASSERT, N_ACCEPT, N_SCALAR };
struct NState {
    uint8_t type;
    AssertKind ak;
    int out = -1, out2 = -1;
    int cls = -1;   // N_BYTE: distinct class-set id; N_SCALAR: index into Nfa::usets
    int atom = -1;  // N_ACCEPT: local atom id
};
static inline bool is_word_assert(AssertKind a) { return a == A_WORD_B || a == A_NOT_WORD_B || a == A_WORD_B_ASCII || a == A_NOT_WORD_B_ASCII; }

enum : uint8_t { K_EDGE = 0 /* START as prev, END as next */, K_OTHER = 1, K_WORD = 2, K_NEWLINE = 3 };

struct Nfa {
    std::vector<NState> st;
    std::vector<ByteSet> sets;
    std::map<std::string, int> set_ids;
    std::vector<int> entries;  // per-pattern entry states
    std::vector<CpSet> usets;  // N_SCALAR sets
    bool uses_word = false /* (never set any more: eliminate_word_asserts rewrites \\b \\B away) */, uses_line = false;
    bool pattern_has_word = false;  // the pattern being built holds a \\b / \\B
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
    int set_id(const ByteSet &b) {
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
                s.cls = set_id(n.cls);
                s.out = next;
                return add(s);
            }
            case RNode::UCLASS: {
                NState s{N_SCALAR, A_TEXT_START};
                s.cls = (int)usets.size();
                usets.push_back(n.ucls);
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
                if (is_word_assert(n.ak)) pattern_has_word = true;
                if (n.ak == A_LINE_START || n.ak == A_LINE_END) uses_line = true;
                return add(s);
            }
        }
        return next;
    }
};


// ---- \b and \B, Unicode-aware, without look-around in the subset construction ---------------------------------------------------
// The regex crate's \b looks at the SCALAR VALUES on both sides (is the previous one a word character? is the next one?). A byte DFA
// cannot look a whole multi-byte scalar ahead, so the assertion is compiled away per pattern, as a product of its NFA with
//   last = what the thread consumed last (nothing known / a non-word scalar / a word scalar [/ an ASCII word character, when the
//          pattern also holds (?-u:\b)]), and
//   need = what the next scalar must be (a set of those kinds, or the end of the text):
// a \b turns `last` into a `need`; a consuming state under a `need` keeps the members of its set that satisfy it; an accept under a
// `need` first consumes one scalar of a permitted kind (or asserts \z) — reporting a match one scalar late is the same match for a
// boolean. What precedes the FIRST scalar a thread consumes is known by consuming it: when a \b can be reached before anything was
// consumed the pattern gets a prelude (\A | one scalar of either kind) — the search is unanchored, so a thread starting one scalar
// earlier is the same search. Next to a byte that is not part of a well-formed sequence both \b and \B are false: such
// a byte is never consumed, so it is the `last = unknown` of a thread that starts behind it and it satisfies no `need`.
struct WordElim {
    Nfa &nfa;
    size_t lo, hi;  // the pattern's states: [lo, hi) of the NFA as built
    CpSet P[3];     // the kinds: 0 = no word character, 1 = word character [beyond ASCII when kind 2 exists], 2 = ASCII word character
    bool has_uni = false, has_ascii = false;
    std::vector<uint8_t> reach;  // [s - lo]: a word assertion can be reached from s
    std::map<std::tuple<int, int, int>, int> memo;
    std::map<std::tuple<int, int, int>, int> chains;
    struct Item { int id, s, last, need; };
    std::vector<Item> work;
    bool unknown_last_met = false;
    static constexpr int ANY = 15, END = 8;

    WordElim(Nfa &n, size_t lo_, size_t hi_) : nfa(n), lo(lo_), hi(hi_) {
        for (size_t s = lo; s < hi; s++)
            if (nfa.st[s].type == N_ASSERT) {
                const AssertKind a = nfa.st[s].ak;
                if (a == A_WORD_B || a == A_NOT_WORD_B) has_uni = true;
                if (a == A_WORD_B_ASCII || a == A_NOT_WORD_B_ASCII) has_ascii = true;
            }
        const CpSet &wu = unicode_word_set(true), &wa = unicode_word_set(false);
        P[0] = cp_complement(has_uni ? wu : wa);
        if (has_uni) P[1] = has_ascii ? cp_intersect(wu, cp_complement(wa)) : wu;
        if (has_ascii) P[2] = wa;
        // which states can reach a word assertion (backwards over the pattern's edges, to a fixpoint)
        reach.assign(hi - lo, 0);
        for (bool grew = true; grew;) {
            grew = false;
            for (size_t s = lo; s < hi; s++) {
                if (reach[s - lo]) continue;
                const NState &n2 = nfa.st[s];
                bool r = n2.type == N_ASSERT && is_word_assert(n2.ak);
                for (int t : {n2.out, n2.out2})
                    if (t >= (int)lo && t < (int)hi && reach[(size_t)t - lo]) r = true;
                if (r) { reach[s - lo] = 1; grew = true; }
            }
        }
    }
    bool reaches(int s) const { return s >= (int)lo && s < (int)hi && reach[(size_t)s - lo]; }
    int mask_word(AssertKind a) const { return (a == A_WORD_B || a == A_NOT_WORD_B) ? 6 : 4; }  // kinds that are word characters to this assertion
    CpSet set_of(const NState &n2) const {
        if (n2.type == N_SCALAR) return nfa.usets[(size_t)n2.cls];
        CpSet o;
        const ByteSet &b = nfa.sets[(size_t)n2.cls];
        for (uint32_t c = 0; c < 128; c++)
            if (b[c]) o.push_back({c, c});
        cp_canon(o);
        return o;
    }
    int consume(const CpSet &set, int out) {  // a state that consumes one scalar of `set`
        bool beyond = false;
        ByteSet b;
        for (auto &r : set) {
            for (uint32_t c = r.first; c <= std::min<uint32_t>(r.second, 127); c++) b.set(c);
            if (r.second > 127) beyond = true;
        }
        NState s{beyond ? N_SCALAR : N_BYTE, A_TEXT_START};
        if (beyond) { s.cls = (int)nfa.usets.size(); nfa.usets.push_back(set); }
        else s.cls = nfa.set_id(b);
        s.out = out;
        return nfa.add(s);
    }
    int alt_of(const std::vector<int> &heads) {  // -1: no alternative is left
        int cur = -1;
        for (size_t k = heads.size(); k-- > 0;) {
            if (cur < 0) { cur = heads[k]; continue; }
            NState e{N_EPS, A_TEXT_START};
            e.out = heads[k];
            e.out2 = cur;
            cur = nfa.add(e);
        }
        return cur;
    }
    int chain_of(int c, int last, int need) {
        auto it = chains.find({c, last, need});
        if (it == chains.end()) it = chains.emplace(std::make_tuple(c, last, need), nfa.n_chains++).first;
        return it->second;
    }
    // the product state (s, last, need): -1 = dead
    int get(int s, int last, int need) {
        if (s < 0) return -1;
        if (need == ANY && !reaches(s)) return s;  // nothing ahead depends on `last`: the pattern's own states go on
        const NState n2 = nfa.st[(size_t)s];
        if (n2.type == N_ASSERT && is_word_assert(n2.ak)) {
            if (last == 0) { unknown_last_met = true; return -1; }
            const int mw = mask_word(n2.ak);
            const bool prev_word = (mw >> (last - 1)) & 1;
            const bool boundary = n2.ak == A_WORD_B || n2.ak == A_WORD_B_ASCII;
            const bool next_word = boundary ? !prev_word : prev_word;
            const int nn = need & (next_word ? mw : (ANY & ~mw));
            return nn ? get(n2.out, last, nn) : -1;
        }
        auto key = std::make_tuple(s, last, need);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        NState ph{N_EPS, A_TEXT_START};
        const int id = nfa.add(ph);
        memo.emplace(key, id);
        work.push_back({id, s, last, need});
        return id;
    }
    void drain() {
        while (!work.empty() && !nfa.overflow) {
            const Item w = work.back();
            work.pop_back();
            const NState n2 = nfa.st[(size_t)w.s];
            NState r{N_EPS, A_TEXT_START};
            switch (n2.type) {
                case N_EPS:
                    r.out = get(n2.out, w.last, w.need);
                    r.out2 = get(n2.out2, w.last, w.need);
                    break;
                case N_ASSERT:  // (\A \z ^ $: the subset construction decides them on bytes)
                    r.type = N_ASSERT;
                    r.ak = n2.ak;
                    r.out = get(n2.out, w.last, w.need);
                    if (r.out < 0) r = NState{N_EPS, A_TEXT_START};
                    break;
                case N_BYTE:
                case N_SCALAR: {
                    CpSet set = set_of(n2);
                    if (w.need != ANY) {
                        CpSet allowed;
                        for (int k = 0; k < 3; k++)
                            if ((w.need >> k) & 1) allowed.insert(allowed.end(), P[k].begin(), P[k].end());
                        cp_canon(allowed);
                        set = cp_intersect(set, allowed);
                    }
                    std::vector<int> heads;
                    if (!reaches(n2.out)) {
                        if (!set.empty()) heads.push_back(consume(set, n2.out));
                    } else {
                        for (int k = 0; k < 3; k++) {
                            const CpSet part = cp_intersect(set, P[k]);
                            if (part.empty()) continue;
                            const int t = get(n2.out, k + 1, ANY);
                            if (t >= 0) heads.push_back(consume(part, t));
                        }
                    }
                    r.out = alt_of(heads);
                    break;
                }
                case N_ACCEPT: {  // under a need: one more scalar of a permitted kind, or the end of the text
                    std::vector<int> heads;
                    for (int k = 0; k < 3; k++)
                        if (((w.need >> k) & 1) && !P[k].empty()) heads.push_back(consume(P[k], w.s));
                    if (w.need & END) {
                        NState e{N_ASSERT, A_TEXT_END};
                        e.out = w.s;
                        heads.push_back(nfa.add(e));
                    }
                    r.out = alt_of(heads);
                    break;
                }
            }
            nfa.st[(size_t)w.id] = r;
            // counted-class chains keep their pruning, each (last, need) variant as a chain of its own
            if ((size_t)w.s < nfa.chain_id.size() && nfa.chain_id[(size_t)w.s] >= 0) nfa.tag(w.id, chain_of(nfa.chain_id[(size_t)w.s], w.last, w.need), nfa.chain_rank[(size_t)w.s]);
            if ((size_t)w.s < nfa.chain_tail.size() && !nfa.chain_tail[(size_t)w.s].empty()) {
                std::vector<int> tails;
                for (int c : nfa.chain_tail[(size_t)w.s]) tails.push_back(chain_of(c, w.last, w.need));
                if ((size_t)w.id >= nfa.chain_tail.size()) nfa.chain_tail.resize(nfa.st.size());
                nfa.chain_tail[(size_t)w.id] = tails;
            }
        }
    }
    int run(int entry) {
        const int plain = get(entry, 0, ANY);
        drain();
        if (!unknown_last_met) return plain;
        std::vector<int> heads;
        if (plain >= 0) heads.push_back(plain);
        {
            const int t = get(entry, 1, ANY);  // the start of the text counts as "no word character before"
            drain();
            if (t >= 0) {
                NState e{N_ASSERT, A_TEXT_START};
                e.out = t;
                heads.push_back(nfa.add(e));
            }
        }
        for (int k = 0; k < 3; k++) {
            if (P[k].empty()) continue;
            const int t = get(entry, k + 1, ANY);
            drain();
            if (t >= 0) heads.push_back(consume(P[k], t));
        }
        const int e = alt_of(heads);
        if (e >= 0) return e;
        NState dead{N_EPS, A_TEXT_START};
        return nfa.add(dead);
    }
};

// Every N_SCALAR state becomes the alternatives of its set: one byte state for the ASCII members, a chain of byte-range states per
// UTF-8 sequence of the others (suffixes shared), joined by epsilon states; the state keeps its number (it may be a chain's tail).
static std::vector<uint8_t> reachable_states(const Nfa &nfa) {
    std::vector<uint8_t> seen(nfa.st.size(), 0);
    std::vector<int> stack(nfa.entries.begin(), nfa.entries.end());
    while (!stack.empty()) {
        const int s = stack.back();
        stack.pop_back();
        if (s < 0 || seen[(size_t)s]) continue;
        seen[(size_t)s] = 1;
        stack.push_back(nfa.st[(size_t)s].out);
        stack.push_back(nfa.st[(size_t)s].out2);
    }
    return seen;
}
static void lower_scalars(Nfa &nfa) {
    const size_t n0 = nfa.st.size();
    const std::vector<uint8_t> live = reachable_states(nfa);  // (what eliminate_word_asserts replaced is still there, unreachable)
    for (size_t s = 0; s < n0 && !nfa.overflow; s++) {
        if (nfa.st[s].type != N_SCALAR || !live[s]) continue;
        const CpSet set = nfa.usets[(size_t)nfa.st[s].cls];
        const int out = nfa.st[s].out;
        std::vector<int> heads;
        ByteSet ascii;
        for (auto &r : set)
            for (uint32_t c = r.first; c <= std::min<uint32_t>(r.second, 127); c++) ascii.set(c);
        auto byte_state = [&](const ByteSet &b, int to) {
            NState x{N_BYTE, A_TEXT_START};
            x.cls = nfa.set_id(b);
            x.out = to;
            return nfa.add(x);
        };
        if (ascii.any()) heads.push_back(byte_state(ascii, out));
        std::vector<std::vector<std::pair<uint8_t, uint8_t>>> seqs;
        utf8_sequences(set, seqs);
        std::map<std::tuple<int, int, int>, int> shared;  // (lo, hi, to) -> state
        for (auto &seq : seqs) {
            int cur = out;
            for (size_t j = seq.size(); j-- > 0;) {
                auto key = std::make_tuple((int)seq[j].first, (int)seq[j].second, cur);
                auto it = shared.find(key);
                if (it == shared.end()) {
                    ByteSet b;
                    for (int c = seq[j].first; c <= seq[j].second; c++) b.set((size_t)c);
                    it = shared.emplace(key, byte_state(b, cur)).first;
                }
                cur = it->second;
            }
            if (std::find(heads.begin(), heads.end(), cur) == heads.end()) heads.push_back(cur);
        }
        NState e{N_EPS, A_TEXT_START};
        int cur = -1;
        for (size_t k = heads.size(); k-- > 1;) {
            NState x{N_EPS, A_TEXT_START};
            x.out = heads[k];
            x.out2 = cur;
            cur = nfa.add(x);
        }
        if (!heads.empty()) { e.out = heads[0]; e.out2 = cur; }
        nfa.st[s] = e;
    }
}

static inline bool needs_next(AssertKind a) { return a == A_TEXT_END || a == A_LINE_END || a == A_WORD_B || a == A_NOT_WORD_B; }
static inline bool holds(AssertKind a, uint8_t pk, uint8_t nk) {
    // pk: K_EDGE = start of text; nk: K_EDGE = end of text
    switch (a) {
        case A_TEXT_START: return pk == K_EDGE;
        case A_TEXT_END: return nk == K_EDGE;
        case A_LINE_START: return pk == K_EDGE || pk == K_NEWLINE;
        case A_LINE_END: return nk == K_EDGE || nk == K_NEWLINE;
        case A_WORD_B: return (pk == K_WORD) != (nk == K_WORD);
        case A_NOT_WORD_B: return (pk == K_WORD) == (nk == K_WORD);
        default: break;  // (word assertions never reach the subset construction: eliminate_word_asserts)
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

bool build_dfa(const std::vector<ScanPattern> &pats, uint32_t max_states, uint32_t max_table_bytes, DfaGroup &out, std::string &err) {
    Nfa nfa;
    nfa.cap = 400000;
    for (size_t k = 0; k < pats.size(); k++) {
        NState acc{N_ACCEPT, A_TEXT_START};
        acc.atom = (int)k;
        int a = nfa.add(acc);
        size_t before = nfa.st.size();
        nfa.pattern_has_word = false;
        int entry = nfa.build(*pats[k].rx, a);
        if (nfa.pattern_has_word && !nfa.overflow) {
            nfa.chain_id.resize(nfa.st.size(), -1);
            nfa.chain_rank.resize(nfa.st.size(), 0);
            nfa.chain_tail.resize(nfa.st.size());
            WordElim we(nfa, (size_t)a, nfa.st.size());
            entry = we.run(entry);
        }
        nfa.entries.push_back(entry);
        if (nfa.overflow || nfa.st.size() - before > 20000) {
            err = "pattern too large (more than 20000 NFA states)";
            return false;
        }
    }
    lower_scalars(nfa);
    if (nfa.overflow) {
        err = "pattern set too large (NFA state limit)";
        return false;
    }
    nfa.chain_id.resize(nfa.st.size(), -1);
    nfa.chain_rank.resize(nfa.st.size(), 0);
    nfa.chain_tail.resize(nfa.st.size());
    // ---- byte classes: bytes are equivalent when no class set and no assertion kind tells them apart ----
    auto kind_of = [&](int b) -> uint8_t {
        bool w = (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_';
        if (nfa.uses_word && w) return K_WORD;
        if (nfa.uses_line && b == '\n') return K_NEWLINE;
        return K_OTHER;
    };
    std::vector<int> cls_of(256);
    int n_cls = 0;
    {
        std::map<int, int> first;
        for (int b = 0; b < 256; b++) {
            int k = kind_of(b);
            auto it = first.find(k);
            if (it == first.end()) it = first.emplace(k, n_cls++).first;
            cls_of[b] = it->second;
        }
        // (only the sets of states a pattern can reach: what the word-assertion rewrite left behind must not split byte classes)
        std::vector<uint8_t> set_used(nfa.sets.size(), 0);
        {
            const std::vector<uint8_t> live = reachable_states(nfa);
            for (size_t q = 0; q < nfa.st.size(); q++)
                if (live[q] && nfa.st[q].type == N_BYTE) set_used[(size_t)nfa.st[q].cls] = 1;
        }
        for (size_t si = 0; si < nfa.sets.size(); si++) {
            if (!set_used[si]) continue;
            const ByteSet &s = nfa.sets[si];
            std::map<std::pair<int, bool>, int> split;
            int next_id = 0;
            std::vector<int> nc(256);
            for (int b = 0; b < 256; b++) {
                auto key = std::make_pair(cls_of[b], (bool)s[b]);
                auto it = split.find(key);
                if (it == split.end()) it = split.emplace(key, next_id++).first;
                nc[b] = it->second;
            }
            cls_of = nc;
            n_cls = next_id;
        }
    }
    std::vector<int> rep(n_cls, -1);  // representative byte per class
    for (int b = 0; b < 256; b++) if (rep[cls_of[b]] < 0) rep[cls_of[b]] = b;
    // set membership per class
    std::vector<std::vector<uint8_t>> set_has(nfa.sets.size(), std::vector<uint8_t>(n_cls));
    for (size_t s = 0; s < nfa.sets.size(); s++)
        for (int c = 0; c < n_cls; c++) set_has[s][c] = nfa.sets[s][rep[c]];
    std::vector<uint8_t> cls_kind(n_cls);
    for (int c = 0; c < n_cls; c++) cls_kind[c] = kind_of(rep[c]);

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
        std::vector<int> liveB[4];      // extra live states released under next-kind nk
        std::vector<uint16_t> accB[4];  // accepts released under nk
        std::vector<std::vector<int>> move;  // per class: sorted targets of liveA ∪ liveB[kind(c)]
        bool ready = false;
    };
    Root roots[4];
    auto get_root = [&](uint8_t pk) -> Root & {
        Root &r = roots[pk];
        if (r.ready) return r;
        r.ready = true;
        r.in.assign(nfa.st.size(), 0);
        bl.stamp++;
        bl.closure(nfa.entries, pk, false, 0, r.liveA, r.accA, &r.pending);
        for (size_t s = 0; s < nfa.st.size(); s++) r.in[s] = bl.mark[s] == bl.stamp;
        for (uint8_t nk = 0; nk < 4; nk++) {
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
        ds[d].next.assign(n_cls, 0);
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
        for (uint8_t nk : {K_OTHER, K_WORD, K_NEWLINE}) {
            bool used = false;
            for (int c = 0; c < n_cls; c++) if (cls_kind[c] == nk) used = true;
            if (!used) continue;
            // phase B marks must not leak between next-kinds: restart from the phase-A marks
            bl.stamp++;
            for (int s : liveA) bl.mark[s] = bl.stamp;
            phaseB(nk);
            for (int c = 0; c < n_cls; c++) {
                if (cls_kind[c] != nk) continue;
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
    for (int b = 0; b < 256; b++) out.classmap[b] = (uint8_t)cls_of[b];
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

void dfa_run_host(const DfaGroup &g, const uint8_t *bytes, size_t n, std::vector<uint16_t> &out_atoms) {
    uint32_t s = 0;
    auto emit = [&](uint32_t st) {
        for (uint32_t k = g.emit_off[st]; k < g.emit_off[st + 1]; k++) out_atoms.push_back(g.emit_list[k]);
    };
    emit(s);
    for (size_t i = 0; i < n; i++) {
        s = g.trans[(size_t)s * g.n_classes + g.classmap[bytes[i]]];
        emit(s);
    }
    for (uint32_t k = g.end_off[s]; k < g.end_off[s + 1]; k++) out_atoms.push_back(g.end_list[k]);
    std::sort(out_atoms.begin(), out_atoms.end());
    out_atoms.erase(std::unique(out_atoms.begin(), out_atoms.end()), out_atoms.end());
}

}  // namespace pwaf
