// filter.cpp — the bigram prefilter of a scan pass (program.h: GroupFilter; kernels.hip: filter_kernel).
//
// The reference evaluates every string predicate of every rule on every request (pingoo/rules.rs:37-51). Nearly all of them are
// false for nearly all requests — that is what a WAF rule set looks like — so the device first asks a much cheaper question per
// field: "could ANY pattern of this pass match here?". Every pattern is reduced to a set of necessary literal FACTORS (every match
// contains one of them), every factor to a window of <= 4 consecutive bigrams, and the windows are spread over 8 buckets of a
// shift-or automaton whose whole transition function is one 16 KiB table indexed by a hashed, case-folded byte pair. The filter
// has no false negatives by construction; what it flags is decided exactly by the pass's DFA.
//
//   1. factor extraction: the usual prefix / suffix / exact-set algebra over the regex tree, on byte SETS per position so that
//      (?i) letters and small classes ([0-9], \s) stay inside a factor;
//   2. window choice: the 4-bigram window of each factor that is least likely in traffic (bin probabilities from a traffic
//      sample when the engine was tuned, else a built-in prior for URL / header text);
//   3. bucket assignment: greedy, minimising the summed false-positive probability of the 8 buckets (a bucket accepts a
//      position when EVERY one of its window positions is hit by SOME member, so members should be few and of equal length);
//   4. heads: anchored literals that most requests satisfy are taken out of the filter and compared directly.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

#include "confirm.h"
#include "program.h"

namespace pwaf {

namespace {

using CStr = std::vector<ByteSet>;  // a "class string": one byte set per position
struct Cover {
    bool ok = false;
    std::vector<CStr> s;
};
// E: every match IS one of these (exact cover); P / S: every match starts / ends with one of these (always known: {""} is trivial);
// B: every match CONTAINS one of these somewhere (best inner factor set found so far).
struct Ext {
    Cover E, P, S, B;
};

constexpr size_t kMaxStrs = 24, kMaxLen = 64, kMaxClass = 16;

Cover trivial() {
    Cover c;
    c.ok = true;
    c.s.push_back({});
    return c;
}

static inline uint32_t fold_pair(uint8_t a, uint8_t b) { return filter_fold(a) | (filter_fold(b) << 8); }  // (program.h: the fold filter_bin applies)

struct Model {
    const double *pairw;  // 65536 probabilities of the case-folded bigrams
    size_t stride;        // sampling stride of the filter being built
    bool extend;          // short windows may reach one bigram beyond their factor (best_window)
    // the distinct folded pairs of two byte sets
    static void pairs_of(const ByteSet &a, const ByteSet &b, std::vector<uint16_t> &out) {
        std::set<uint32_t> ps;
        for (int x = 0; x < 256; x++)
            if (a[(size_t)x])
                for (int y = 0; y < 256; y++)
                    if (b[(size_t)y]) ps.insert(fold_pair((uint8_t)x, (uint8_t)y));
        out.assign(ps.begin(), ps.end());
    }
    double pair_weight(const ByteSet &a, const ByteSet &b) {
        // probability that a text bigram is one of the (folded) pairs of the two sets: independent of the hash, which only adds
        // the pairs that happen to share a bin
        std::vector<uint16_t> ps;
        pairs_of(a, b, ps);
        double w = 0;
        for (uint16_t x : ps) w += pairw[x];
        return std::min(1.0, w);
    }
    // Cheapest window of <= 4 sampled bigrams for a factor whose first byte sits `o` bytes past a sampling point (o < stride):
    // sampled bigrams start at o' = (stride - o) % stride ... i.e. at internal offsets a, a + stride, ...; returns its cost, the
    // internal offset of its first bigram and the bigram count (0 = the factor is too short for this alignment).
    // A factor with fewer than FOUR sampled bigrams of its own ("../" at stride 2: ".." or "./", whichever the alignment samples) is
    // EXTENDED by the sampled bigram that straddles its start — (any byte, s[0]): internal offset -1 — and / or its end —
    // (s[len - 1], any byte): offset len - 1 — when the alignment samples those: the text around an occurrence is part of the same
    // byte stream, so the straddling bigrams are always there (fields lie back to back in the arena, PWAF_ARENA_PAD behind the last;
    // in front only at stride 2, where offset -1 is sampled for factors at ODD arena positions: there is a byte before those). A window of
    // one bigram made every other request a candidate (the URL pass of the 1k-rule set: 33 % of the sample at stride 2 — which is what
    // kept a third of the streamed bytes on the LDS-bound stride-1 path); two positions, one of them a class of 256 pairs, flag a few
    // requests in a thousand. The window's LAST bigram always starts inside the factor (the confirm tier finds the factor from there).
    double best_window(const CStr &s, size_t a, long &start, size_t &k) {
        std::vector<long> pos;
        for (size_t j = a; j + 1 < s.size(); j += stride) pos.push_back((long)j);
        if (extend && stride > 1 && pos.size() < 4 && !s.empty()) {  // (stride 1: neighbouring bigrams overlap — the bigram behind a factor's last one is implied by it, an extension there filters nothing)
            const long last = (long)s.size() - 1;
            if (a + 1 == stride) pos.insert(pos.begin(), -1L);
            if (last >= (long)a && ((size_t)last - a) % stride == 0) pos.push_back(last);
            if (pos.size() == 1 && pos[0] < 0) pos.clear();  // (only the bigram in front: no position inside the factor to find it from)
        }
        k = std::min<size_t>(4, pos.size());
        start = 0;
        if (k == 0) return INFINITY;
        ByteSet any;
        any.set();
        std::vector<double> w(pos.size());
        for (size_t q = 0; q < pos.size(); q++) w[q] = pair_weight(pos[q] < 0 ? any : s[(size_t)pos[q]], (size_t)(pos[q] + 1) < s.size() ? s[(size_t)pos[q] + 1] : any);
        double best = 2;
        for (size_t st = 0; st + k <= pos.size(); st++) {
            double c = 1;
            for (size_t j = 0; j < k; j++) c *= w[st + j];
            if (c < best) { best = c; start = pos[st]; }
        }
        return best;
    }
    // expected false-positive contribution of a factor set (every string at every alignment); +inf when it is not usable
    double score(const Cover &c) {
        if (!c.ok || c.s.empty()) return INFINITY;
        double t = 0;
        for (auto &s : c.s) {
            if (s.size() < 1 + stride) return INFINITY;
            for (size_t a = 0; a < stride; a++) {
                long st;
                size_t k;
                t += best_window(s, a, st, k);
            }
        }
        return t;
    }
};

bool cross(const Cover &a, const Cover &b, Cover &out, int keep /* 0: all (fail when too long), 1: first kMaxLen, 2: last kMaxLen */) {
    out = Cover();
    if (!a.ok || !b.ok) return false;
    if (a.s.size() * b.s.size() > kMaxStrs) return false;
    std::set<std::string> seen;
    for (auto &x : a.s)
        for (auto &y : b.s) {
            CStr z = x;
            z.insert(z.end(), y.begin(), y.end());
            if (z.size() > kMaxLen) {
                if (keep == 0) return false;
                if (keep == 1) z.resize(kMaxLen);
                else z.erase(z.begin(), z.end() - (long)kMaxLen);
            }
            std::string key;
            for (auto &bs : z) key += bs.to_string();
            if (seen.insert(key).second) out.s.push_back(std::move(z));
        }
    out.ok = true;
    return true;
}

bool unite(const Cover &a, const Cover &b, Cover &out) {
    out = Cover();
    if (!a.ok || !b.ok || a.s.size() + b.s.size() > 2 * kMaxStrs) return false;
    out.ok = true;
    out.s = a.s;
    out.s.insert(out.s.end(), b.s.begin(), b.s.end());
    return true;
}

// The cheaper cover (expected false positives).
const Cover &better(Model &m, const Cover &a, const Cover &b) { return m.score(b) < m.score(a) ? b : a; }

Ext cat2(Model &m, const Ext &a, const Ext &b) {
    Ext r;
    Cover t;
    if (cross(a.E, b.E, t, 0)) r.E = t;
    if (a.E.ok && cross(a.E, b.P, t, 1)) r.P = t;
    else r.P = a.P;
    if (b.E.ok && cross(a.S, b.E, t, 2)) r.S = t;
    else r.S = b.S;
    r.B = better(m, a.B, b.B);
    if (cross(a.S, b.P, t, 1)) r.B = better(m, r.B, t);
    if (!std::isfinite(m.score(r.B))) r.B = Cover();
    return r;
}

Ext extract(Model &m, const RNode &n) {
    Ext r;
    switch (n.k) {
        case RNode::EMPTY:
        case RNode::ASSERT:
            r.E = r.P = r.S = trivial();
            return r;
        case RNode::CLASS:
            if (n.cls.count() >= 1 && n.cls.count() <= kMaxClass) {
                r.E.ok = true;
                r.E.s.push_back({n.cls});
                r.P = r.S = r.E;
            } else {
                r.P = r.S = trivial();
            }
            return r;
        case RNode::UCLASS: {
            // One scalar value of a set with members beyond ASCII (`.`, \s, (?i)s = s | S | U+017F ...): one class string per
            // encoded LENGTH — the ASCII members, and per length 2..4 the union of the sequences' byte ranges position by position
            // (a superset of the encodings: a cover may over-approximate, it must not miss). A position with more than kMaxClass
            // bytes makes the class wide: it then only breaks factors, as `.` always did.
            std::vector<std::vector<std::pair<uint8_t, uint8_t>>> seqs;
            utf8_sequences(n.ucls, seqs);
            CStr by_len[5];
            for (auto &q : seqs) {
                CStr &c = by_len[q.size()];
                c.resize(q.size());
                for (size_t j = 0; j < q.size(); j++)
                    for (int b = q[j].first; b <= q[j].second; b++) c[j].set((size_t)b);
            }
            bool wide = n.cls.count() > kMaxClass;
            for (int L = 2; L <= 4; L++)
                for (auto &bs : by_len[L]) wide = wide || bs.count() > kMaxClass;
            if (wide) {
                r.P = r.S = trivial();
                return r;
            }
            r.E.ok = true;
            if (n.cls.any()) r.E.s.push_back({n.cls});
            for (int L = 2; L <= 4; L++)
                if (!by_len[L].empty()) r.E.s.push_back(by_len[L]);
            r.P = r.S = r.E;
            return r;
        }
        case RNode::CAT: {
            r.E = r.P = r.S = trivial();
            for (auto &k : n.kids) r = cat2(m, r, extract(m, *k));
            return r;
        }
        case RNode::ALT: {
            bool first = true, b_ok = true;
            Cover bset;
            for (auto &k : n.kids) {
                Ext c = extract(m, *k);
                // what a whole alternative guarantees: its best inner factor, or the alternative itself
                Cover cand = c.B;
                cand = better(m, cand, c.E);
                cand = better(m, cand, c.P);
                cand = better(m, cand, c.S);
                if (!std::isfinite(m.score(cand))) b_ok = false;
                if (first) {
                    r.E = c.E;
                    r.P = c.P;
                    r.S = c.S;
                    bset = cand;
                    first = false;
                    continue;
                }
                Cover t;
                if (unite(r.E, c.E, t) && t.s.size() <= kMaxStrs) r.E = t; else r.E = Cover();
                if (unite(r.P, c.P, t) && t.s.size() <= kMaxStrs) r.P = t; else r.P = trivial();
                if (unite(r.S, c.S, t) && t.s.size() <= kMaxStrs) r.S = t; else r.S = trivial();
                if (b_ok && !unite(bset, cand, t)) b_ok = false;
                else if (b_ok) bset = t;
            }
            if (b_ok && !first) r.B = bset;
            return r;
        }
        case RNode::REPEAT: {
            const Ext c = extract(m, *n.kids[0]);
            if (n.rmin <= 0) {
                if (n.rmax == 0) r.E = trivial();
                r.P = r.S = trivial();
                return r;
            }
            const int copies = std::min(n.rmin, 4);
            r = c;
            for (int i = 1; i < copies; i++) r = cat2(m, r, c);
            if (!(n.rmax == n.rmin && n.rmin <= 4)) {
                r.E = Cover();  // more may follow the copies taken
                r.S = c.S;      // ... but the last repetition still ends the match
            }
            return r;
        }
    }
    return r;
}

// pattern == (\A)? literal (\z)?  with literal made of single bytes
bool anchored_literal(const RNode &n, std::string &lit, bool &at_start, bool &at_end) {
    std::vector<const RNode *> flat;
    std::vector<const RNode *> stack{&n};
    while (!stack.empty()) {
        const RNode *x = stack.back();
        stack.pop_back();
        if (x->k == RNode::CAT) {
            for (size_t k = x->kids.size(); k-- > 0;) stack.push_back(x->kids[k].get());
        } else if (x->k != RNode::EMPTY) {
            flat.push_back(x);
        }
    }
    lit.clear();
    at_start = at_end = false;
    size_t i = 0, e = flat.size();
    if (i < e && flat[i]->k == RNode::ASSERT && flat[i]->ak == A_TEXT_START) { at_start = true; i++; }
    if (e > i && flat[e - 1]->k == RNode::ASSERT && flat[e - 1]->ak == A_TEXT_END) { at_end = true; e--; }
    for (; i < e; i++) {
        if (flat[i]->k != RNode::CLASS || flat[i]->cls.count() != 1) return false;
        for (int b = 0; b < 256; b++)
            if (flat[i]->cls[(size_t)b]) lit += (char)b;
    }
    return true;
}

// Prior over the bytes of URL / header text, used when no traffic sample is available. Only relative magnitudes matter:
// it decides which window of a factor is taken and how factors are bucketed, never a result.
void default_pair_prob(double *pairw) {
    double p[256];
    for (int b = 0; b < 256; b++) p[b] = 0.0002;
    for (int b = 'a'; b <= 'z'; b++) p[b] = 0.024;
    for (int b = 'A'; b <= 'Z'; b++) p[b] = 0.003;
    for (int b = '0'; b <= '9'; b++) p[b] = 0.012;
    const char *common = "/.-_=&% ";
    for (const char *c = common; *c; c++) p[(unsigned char)*c] = 0.02;
    p['/'] = 0.05;
    const char *some = "?;():,+*~@!$'\"";
    for (const char *c = some; *c; c++) p[(unsigned char)*c] = 0.004;
    double tot = 0;
    for (int b = 0; b < 256; b++) tot += p[b];
    for (uint32_t x = 0; x < 65536; x++) pairw[x] = 0;
    for (int a = 0; a < 256; a++)
        for (int b = 0; b < 256; b++) pairw[fold_pair((uint8_t)a, (uint8_t)b)] += p[a] / tot * p[b] / tot;
}

struct Window {
    size_t k = 0;
    std::vector<uint16_t> pairs[4];  // folded byte pairs per window position
    double cost = 0;
};

}  // namespace

bool confirm_literal(const RNode &n, std::string &lit, bool &at_start, bool &at_end) {
    return anchored_literal(n, lit, at_start, at_end) && lit.size() >= 2 && lit.size() <= kMaxLen;
}

namespace {
// One factor occurrence the confirm tier can verify: the factor string, where its window ends, what it decides.
struct ConfirmSeed {
    CStr s;
    uint32_t d;     // the window's last bigram starts d bytes into the factor
    uint16_t atom;  // local atom (a confirm literal), or kConfirmWalk
    uint8_t flags;
};

// Lays the seeds out as the table confirm.h reads (program.h: ConfirmTable). false: a limit of the encoding was exceeded — the
// pass then keeps the round-3 path (every candidate walked through the full DFA).
bool build_confirm_table(std::vector<ConfirmSeed> &seeds, uint32_t mul, ConfirmTable &out) {
    out = ConfirmTable();
    // identical (factor, window end, decision) seeds — regexes sharing a factor — once
    {
        std::map<std::string, size_t> seen;
        std::vector<ConfirmSeed> uniq;
        for (auto &sd : seeds) {
            std::string key = std::to_string(sd.d) + ":" + std::to_string(sd.atom) + ":" + std::to_string(sd.flags) + ":";
            for (auto &bs : sd.s) key += bs.to_string();
            if (seen.emplace(key, uniq.size()).second) uniq.push_back(sd);
        }
        seeds.swap(uniq);
    }
    std::map<std::string, uint32_t> class_ids;
    struct Placed { uint32_t bin; ConfirmEntry e; };
    std::vector<Placed> placed;
    for (const ConfirmSeed &sd : seeds) {
        const size_t len = sd.s.size();
        if (len < 2 || len > kMaxLen || (size_t)sd.d + 1 > len) return false;  // (the window's last bigram starts inside the factor; d = len - 1: its second byte is the byte behind the factor)
        const size_t l4 = (len + 3) & ~(size_t)3;
        ConfirmEntry e{};
        e.bytes_off = (uint32_t)out.bytes.size();
        e.len = (uint16_t)len;
        e.d = (uint16_t)sd.d;
        e.atom = sd.atom;
        e.flags = sd.flags;
        std::vector<uint8_t> val(l4, 0), msk(l4, 0), cls;
        for (size_t j = 0; j < len; j++) {
            const ByteSet &bs = sd.s[j];
            const size_t cnt = bs.count();
            int x = -1, y = -1;
            for (int b = 0; b < 256; b++)
                if (bs[(size_t)b]) { if (x < 0) x = b; else if (y < 0) y = b; }
            if (cnt == 1) {
                val[j] = (uint8_t)x;
                msk[j] = 0xFF;
            } else if (cnt == 2 && ((x ^ y) & ((x ^ y) - 1)) == 0) {  // two bytes one bit apart ((?i) letters): compared under a mask
                msk[j] = (uint8_t)~(x ^ y);
                val[j] = (uint8_t)(x & msk[j]);
            } else {  // a byte class: position + class id after the masks
                const std::string key = bs.to_string();
                auto it = class_ids.find(key);
                if (it == class_ids.end()) {
                    it = class_ids.emplace(key, (uint32_t)class_ids.size()).first;
                    for (int w = 0; w < 8; w++) {
                        uint32_t word = 0;
                        for (int b = 0; b < 32; b++)
                            if (bs[(size_t)(w * 32 + b)]) word |= 1u << b;
                        out.classes.push_back(word);
                    }
                }
                if (it->second > 255 || cls.size() >= 2 * 255) return false;
                cls.push_back((uint8_t)j);
                cls.push_back((uint8_t)it->second);
            }
        }
        e.n_cls = (uint8_t)(cls.size() / 2);
        out.bytes.insert(out.bytes.end(), val.begin(), val.end());
        out.bytes.insert(out.bytes.end(), msk.begin(), msk.end());
        out.bytes.insert(out.bytes.end(), cls.begin(), cls.end());
        while (out.bytes.size() & 3) out.bytes.push_back(0);
        // the bins the window's LAST bigram can fall into: the entry is listed under each
        std::vector<uint16_t> pairs;
        ByteSet any;
        any.set();
        Model::pairs_of(sd.s[sd.d], (size_t)sd.d + 1 < len ? sd.s[sd.d + 1] : any, pairs);
        std::set<uint32_t> bins;
        for (uint16_t pr : pairs) bins.insert(filter_bin((uint8_t)pr, (uint8_t)(pr >> 8), mul));
        for (uint32_t bn : bins) placed.push_back({bn, e});
        if (sd.atom == kConfirmWalk) out.has_walk = true;
    }
    std::stable_sort(placed.begin(), placed.end(), [](const Placed &a, const Placed &b) { return a.bin < b.bin; });
    if (placed.size() >= (1u << 20)) return false;
    out.head.assign(kFilterEntries, 0u);
    for (size_t k = 0; k < placed.size();) {
        size_t j = k;
        while (j < placed.size() && placed[j].bin == placed[k].bin) j++;
        if (j - k > 4095) return false;
        out.head[placed[k].bin] = (uint32_t)k | ((uint32_t)(j - k) << 20);
        k = j;
    }
    for (auto &pl : placed) out.entries.push_back(pl.e);
    for (int pad = 0; pad < 4; pad++) out.bytes.push_back(0);  // (the last entry's dword loads)
    while (out.bytes.size() & 3) out.bytes.push_back(0);
    if (out.classes.empty()) out.classes.assign(8, 0u);
    out.enabled = true;
    return true;
}
}  // namespace

void build_group_filter(const std::vector<Atom> &atoms, const DfaGroup &g, const FilterHints *hints, GroupFilter &out, uint32_t stride, bool extend) {
    out = GroupFilter();
    out.stride = stride;
    std::vector<double> prior;
    const double *pairw0 = hints ? hints->pair_prob : nullptr;
    if (!pairw0) {
        prior.resize(65536);
        default_pair_prob(prior.data());
        pairw0 = prior.data();
    }
    // smoothed: a bigram the sample never showed is still possible
    std::vector<double> pw(65536);
    for (uint32_t x = 0; x < 65536; x++) pw[x] = pairw0[x] * 0.98 + 0.02 / 65536;
    Model m{pw.data(), stride, extend && stride > 1};

    if (g.field == PWAF_FIELD_METHOD) { out.note = "method: a handful of bytes per request, the DFA pass is already cheaper than a filter + confirmation"; return; }
    if (hints && hints->mean_len > 0 && hints->mean_len < 8) { out.note = "mean field length below 8 bytes"; return; }
    if (!g.filter_atoms.empty()) { out.note = "gated pass"; return; }
    if (g.emit_off.size() > 1 && g.emit_off[1] > g.emit_off[0]) { out.note = "a pattern matches the empty string"; return; }

    // ---- heads: hot anchored literals ----
    const uint64_t n_req = hints && hints->atom_hits ? hints->n_requests : 0;
    auto rate = [&](uint32_t local) -> double {
        if (n_req && hints->atom_hits && local < hints->atom_hits->size()) return (double)(*hints->atom_hits)[local] / (double)n_req;
        return -1;  // unknown
    };
    struct HeadCand { uint32_t local; double hot; std::string lit; bool exact; };
    std::vector<HeadCand> hc;
    std::vector<uint8_t> is_head(g.atoms.size(), 0);
    for (uint32_t l = 0; l < g.atoms.size(); l++) {
        const Atom &at = atoms[g.atoms[l]];
        std::string lit;
        bool s, e;
        if (!at.pattern || !anchored_literal(*at.pattern, lit, s, e) || !s || lit.empty() || lit.size() > 16) continue;
        // A prefilter factor of a gap pass (`\A/api/` of `^/api/.*foo`) must not become a head: a head is compared by the filter kernel
        // and only lands in the hit record — the request is neither a bigram candidate nor enqueued for the gap pass, whose `.*` rule
        // would then silently never match for requests outside the candidate list. It stays in the filter windows;
        // if that makes most of the sample a candidate, tune drops the filter and the pass streams every request.
        if (at.gates) continue;
        const double r = rate(l);
        const double hot = r >= 0 ? r : (at.neg_used ? 0.5 : 0.0);
        if (hot >= 0.02) hc.push_back({l, hot, lit, e});
    }
    std::stable_sort(hc.begin(), hc.end(), [](const HeadCand &a, const HeadCand &b) { return a.hot > b.hot; });
    if (hc.size() > 2) hc.resize(2);
    for (auto &h : hc) {
        FilterHead fh{};
        memcpy(fh.bytes, h.lit.data(), h.lit.size());
        fh.len = (uint8_t)h.lit.size();
        fh.exact = h.exact ? 1 : 0;
        fh.local = (uint16_t)h.local;
        if (h.local >= 0x7FFF) continue;  // (does not fit an inline record slot)
        out.heads.push_back(fh);
        is_head[h.local] = 1;
    }

    // ---- factors -> windows ----
    std::map<std::string, size_t> index;
    std::vector<Window> wins;
    std::vector<ConfirmSeed> seeds;  // every (factor, alignment) entered into the filter, for the confirm tier
    const bool l_fits_records = g.atoms.size() < kConfirmWalk;
    for (uint32_t l = 0; l < g.atoms.size(); l++) {
        if (is_head[l]) continue;
        const Atom &at = atoms[g.atoms[l]];
        if (!at.pattern) { out.note = "atom without a pattern"; out.heads.clear(); return; }
        const double r = rate(l);
        if (r >= 0.25) {
            out.note = "a pattern that is not an anchored literal holds for " + std::to_string((int)(r * 100)) + " % of the sample";
            out.heads.clear();
            return;
        }
        // A confirm literal's factor is the literal itself, whole (what the confirm tier compares decides the atom); any other
        // pattern takes the cheapest necessary factor set.
        Cover f;
        std::string c_lit;
        bool c_start = false, c_end = false;
        const bool is_lit = confirm_literal(*at.pattern, c_lit, c_start, c_end);
        if (is_lit) {
            CStr cs;
            for (unsigned char ch : c_lit) { ByteSet b1; b1.set(ch); cs.push_back(b1); }
            f.ok = true;
            f.s.push_back(std::move(cs));
        } else {
            Ext x = extract(m, *at.pattern);
            f = x.B;
            f = better(m, f, x.E);
            f = better(m, f, x.P);
            f = better(m, f, x.S);
        }
        if (!std::isfinite(m.score(f))) {
            out.note = "pattern without a literal factor of " +  std::to_string(1 + stride) + " or more bytes: " + at.key.substr(0, 80);
            out.heads.clear();
            return;
        }
        for (auto &s : f.s)
            for (size_t al = 0; al < stride; al++) {
                long st;
                size_t k;
                Window wd;
                wd.cost = m.best_window(s, al, st, k);
                wd.k = k;
                std::string key = std::to_string(k) + ":";
                ByteSet any;
                any.set();
                for (size_t j = 0; j < k; j++) {
                    const long at = st + (long)(j * stride);  // (-1 / len - 1: the bigram that straddles the factor's start / end)
                    Model::pairs_of(at < 0 ? any : s[(size_t)at], (size_t)(at + 1) < s.size() ? s[(size_t)at + 1] : any, wd.pairs[j]);
                    for (uint16_t pr : wd.pairs[j]) key += std::to_string(pr) + ",";
                    key += ";";
                }
#ifdef PWAF_PROFILING
                if (getenv("PWAF_FILTER_DEBUG") && (wd.cost > 1e-5 || (k <= 2 && stride > 1))) {
                    std::string txt;
                    for (auto &bs : s) { int c = -1, cnt = 0; for (int b = 0; b < 256; b++) if (bs[(size_t)b]) { c = b; cnt++; } txt += cnt == 1 ? (char)c : '#'; }
                    fprintf(stderr, "filter field %d: factor '%s' align %zu window at %zu x%zu cost %.2e (atom %s)\n", g.field, txt.c_str(), al, st, k, wd.cost, at.key.substr(0, 60).c_str());
                }
#endif
                if (index.emplace(key, wins.size()).second) wins.push_back(std::move(wd));
                seeds.push_back(ConfirmSeed{s, (uint32_t)(st + (long)((k - 1) * stride)), is_lit ? (uint16_t)l : kConfirmWalk,
                                            (uint8_t)(is_lit ? ((c_start ? kConfirmAtStart : 0) | (c_end ? kConfirmAtEnd : 0)) : 0)});
            }
    }
    if (wins.empty() && out.heads.empty()) { out.note = "no patterns"; return; }

    // ---- buckets, once per candidate multiplier of the bigram hash: the one whose bins keep the factor windows away from the
    //      traffic's frequent bigrams wins (a three-byte factor such as "../" owns only two positions: one unlucky collision with a
    //      common bigram and every other request is a candidate) ----
    struct Bucket {
        size_t kmin = 5;  // 5 = empty
        std::bitset<kFilterEntries> set[4];
        double wsum[4] = {0, 0, 0, 0};
        double fp() const {
            if (kmin > 4) return 0;
            double f = 1;
            for (size_t j = 4 - kmin; j < 4; j++) f *= std::min(1.0, wsum[j]);
            return f;
        }
    };
    std::vector<size_t> order(wins.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return wins[a].k != wins[b].k ? wins[a].k > wins[b].k : wins[a].cost > wins[b].cost; });
    std::vector<size_t> order_short_first = order;
    std::stable_sort(order_short_first.begin(), order_short_first.end(), [&](size_t a, size_t b) { return wins[a].k != wins[b].k ? wins[a].k < wins[b].k : wins[a].cost > wins[b].cost; });
    static const uint32_t kMuls[] = {kFilterMul, 0x85EBu, 0xC2B3u, 0x27D5u, 0x165Bu, 0xB5A7u, 0x6F4Fu, 0x93D7u, 0xE995u, 0x4F1Du, 0xA3C1u, 0x7A6Bu,
                                     0x3C6Fu, 0xD1B5u, 0x5BD1u, 0xE6A9u, 0x2F8Du, 0x9A4Bu, 0x7F4Bu, 0xC34Fu};
    // Two things depend on the multiplier: how far the factor windows stay from the traffic's frequent bigrams (false positives ->
    // candidates) and how evenly the traffic's bigrams spread over the 32 LDS banks (filter_kernel is bound by the bank conflicts
    // of its table gather: a ds_read_b32 takes as many cycles as its most loaded bank; lanes reading the SAME bin are a broadcast).
    // Pass 1 finds the lowest false-positive estimate; pass 2 takes, among the multipliers within 25 % of it, the one with the
    // lowest probability that two text positions read different bins of one bank (measured spread: 0.037 .. 0.074, uniform 0.031).
    auto bank_collision = [&](const std::vector<double> &w) {
        double bank[32] = {0}, same_bin = 0;
        for (uint32_t bn = 0; bn < kFilterEntries; bn++) { bank[bn & 31] += w[bn]; same_bin += w[bn] * w[bn]; }
        double c = 0;
        for (double x : bank) c += x * x;
        return c - same_bin;
    };
    double lowest_fp = INFINITY;
    const bool have_sample = hints && hints->pair_prob;
    std::vector<double> w(kFilterEntries);
    std::vector<double> mul_fp, mul_bank;
    double best_fp = INFINITY, best_bank = INFINITY;
    for (int pass = 0; pass < 2; pass++) {
    size_t mul_index = 0;
    for (uint32_t mul : kMuls) {
        // (without a traffic sample the estimate rests on a generic prior, which cannot tell the multipliers apart: keep the default)
        if (!have_sample && mul != kFilterMul) continue;
        const size_t mi = mul_index++;
        if (pass == 1 && !(mul_fp[mi] <= lowest_fp * 1.25 + 1e-12 && mul_bank[mi] < best_bank)) continue;
        std::fill(w.begin(), w.end(), 0.0);
        for (uint32_t pr = 0; pr < 65536; pr++) w[filter_bin((uint8_t)pr, (uint8_t)(pr >> 8), mul)] += pw[pr];
        Bucket bk[8];
        auto add = [&](Bucket &t, const Window &wd) {
            t.kmin = std::min(t.kmin, wd.k);
            for (size_t j = 0; j < wd.k; j++) {
                const size_t pos = 4 - wd.k + j;
                for (uint16_t pr : wd.pairs[j]) {
                    const uint32_t bn = filter_bin((uint8_t)pr, (uint8_t)(pr >> 8), mul);
                    if (!t.set[pos][bn]) { t.set[pos].set(bn); t.wsum[pos] += w[bn]; }
                }
            }
        };
        auto assign = [&](const std::vector<size_t> &ord, Bucket (&out_bk)[8]) {
            for (size_t wi : ord) {
                int best_b = 0;
                double best_d = INFINITY;
                for (int b = 0; b < 8; b++) {
                    Bucket t = out_bk[b];
                    add(t, wins[wi]);
                    const double d = t.fp() - out_bk[b].fp();
                    if (d < best_d) { best_d = d; best_b = b; }
                }
                add(out_bk[best_b], wins[wi]);
            }
            double fp = 0;
            for (int b = 0; b < 8; b++) fp += out_bk[b].fp();
            return fp;
        };
        double fp_pos = assign(order, bk);
        if (extend && stride > 1) {
            // Short windows FIRST: a bucket tests only its last kmin positions, so a two-bigram window that arrives when all eight
            // buckets are taken cuts some bucket's members down to their last two bigrams (the URL pass: "../" joined a bucket of
            // three- and four-bigram windows, whose cross products with its 256-pair class then flagged 14 % of the sample). Placed
            // first, the short windows keep buckets of their own and the long ones share the rest. Whichever order estimates lower.
            Bucket alt[8];
            const double fp_alt = assign(order_short_first, alt);
            if (fp_alt < fp_pos) {
                fp_pos = fp_alt;
                for (int b = 0; b < 8; b++) bk[b] = alt[b];
            }
        }
        if (pass == 0) {
            mul_fp.push_back(fp_pos);
            mul_bank.push_back(bank_collision(w));
            lowest_fp = std::min(lowest_fp, fp_pos);
            continue;
        }
        best_bank = mul_bank[mi];
        best_fp = fp_pos;
#ifdef PWAF_PROFILING
        if (getenv("PWAF_FILTER_DEBUG")) {
            fprintf(stderr, "filter field %d stride %u mul %#x: fp per position %.3e;", g.field, stride, mul, fp_pos);
            for (int b = 0; b < 8; b++) fprintf(stderr, " bucket %d kmin %zu fp %.2e (w %.3f %.3f %.3f %.3f)", b, bk[b].kmin, bk[b].fp(), bk[b].wsum[0], bk[b].wsum[1], bk[b].wsum[2], bk[b].wsum[3]);
            fprintf(stderr, "\n");
        }
#endif
        out.mul = mul;
        out.table.assign(kFilterEntries, 0xFFFFFFFFu);
        out.init = 0xFFFFFFFFu;
        for (int b = 0; b < 8; b++) {
            if (bk[b].kmin > 4) continue;
            for (size_t j = 0; j < 4; j++) {
                const uint32_t bit = 1u << (8 * j + (size_t)b);
                if (j < 4 - bk[b].kmin) {  // wildcard position of this bucket
                    for (auto &e : out.table) e &= ~bit;
                    out.init &= ~bit;
                } else {
                    for (uint32_t bn = 0; bn < kFilterEntries; bn++)
                        if (bk[b].set[j][bn]) out.table[bn] &= ~bit;
                }
            }
        }
    }
    }  // passes
    const double len = (hints && hints->mean_len > 0 ? hints->mean_len : 64.0) / stride;
    out.est_candidate_rate = 1.0 - std::pow(std::max(0.0, 1.0 - std::min(1.0, best_fp)), len);
    out.enabled = true;
    if (l_fits_records && !g.confirm_off) build_confirm_table(seeds, out.mul, out.confirm);  // (failure leaves confirm.enabled false: the round-3 path)
}

bool short_literal_atom(const RNode &n, std::string &lit, bool &exact) {
    bool at_start = false, at_end = false;
    if (!anchored_literal(n, lit, at_start, at_end) || !at_start || lit.size() > 8) return false;
    exact = at_end;
    return true;
}

bool filter_candidate_host(const GroupFilter &f, const uint8_t *bytes, size_t n, size_t phase) {
    uint32_t st = f.init;
    for (size_t i = phase; i + 1 < n; i += f.stride) {
        st = (st << 8) | f.table[filter_bin(bytes[i], bytes[i + 1], f.mul)];
        if ((~st) & 0xFF000000u) return true;
    }
    return false;
}

// The filter over the field [fs, fe) of an arena as the device's flat stream sees it: the state at the field's first sampled position
// is what the (up to four) sampled bigrams in front of it left behind — the previous fields' bytes, or the init state at the arena's
// start — and the field's last position pairs its last byte with the byte behind the field (the windows of short factors reach one
// bigram beyond the factor on either side: Model::best_window). flag(i) for every position of the field that completes a window;
// bytes at or beyond `readable` read as zero.
template <class Flag>
static void filter_field_positions(const GroupFilter &f, const uint8_t *arena, uint32_t fs, uint32_t fe, uint64_t readable, Flag &&flag) {
    const uint32_t s = f.stride ? f.stride : 1u;
    const uint32_t i0 = s == 2 ? (fs + 1u) & ~1u : fs;  // (bigrams are sampled at the even bytes of the ARENA)
    uint32_t st = f.init;
    auto at = [&](uint64_t i) -> uint8_t { return i < readable ? arena[i] : (uint8_t)0; };
    for (uint32_t i = i0 >= 4u * s ? i0 - 4u * s : i0 % s; i < fe; i += s) {
        st = (st << 8) | f.table[filter_bin(at(i), at((uint64_t)i + 1), f.mul)];
        if (i >= i0 && ((~st) & 0xFF000000u)) flag(i);
    }
}

bool filter_candidate_arena(const GroupFilter &f, const uint8_t *arena, uint32_t fs, uint32_t fe, uint64_t readable) {
    bool any = false;
    filter_field_positions(f, arena, fs, fe, readable, [&](uint32_t) { any = true; });
    return any;
}

uint32_t filter_heads_host(const GroupFilter &f, const uint8_t *bytes, size_t n) {
    uint32_t r = 0;
    for (size_t k = 0; k < f.heads.size(); k++) {
        const FilterHead &h = f.heads[k];
        if (n < h.len || (h.exact && n != h.len)) continue;
        if (memcmp(bytes, h.bytes, h.len) == 0) r |= 1u << k;
    }
    return r;
}

}  // namespace pwaf

namespace pwaf {

// Host model of filter + confirm tier over the field [fs, fe) of an arena with PWAF_ARENA_PAD readable bytes behind it: the chunks
// the filter flags at the field's positions (filter_field_positions: as the device's stream sees them, neighbours included), then
// confirm.h over each of them. Appends the confirmed literal atoms (local ids, possibly
// repeated) to `lits`; returns true when the request must be walked through the DFA (always, for a flagged field of a pass without
// a confirm table). `flagged` (optional): the filter flagged the field at all.
bool confirm_field_host(const GroupFilter &f, const uint8_t *arena, uint32_t fs, uint32_t fe, std::vector<uint16_t> &lits, bool *flagged) {
    std::vector<uint32_t> chunks;
    filter_field_positions(f, arena, fs, fe, (uint64_t)fe + PWAF_ARENA_PAD, [&](uint32_t i) {
        if (chunks.empty() || chunks.back() != (i >> 4)) chunks.push_back(i >> 4);
    });
    if (flagged) *flagged = !chunks.empty();
    if (chunks.empty()) return false;
    if (!f.confirm.enabled) return true;
    const ConfirmTable &t = f.confirm;
    const ConfirmView cv{t.head.data(), t.entries.data(), t.bytes.data(), t.classes.data(), f.mul, f.stride, f.init};
    bool walk = false;
    for (uint32_t c : chunks) {
        const bool wk = confirm_chunk(cv, arena, fs, fe, c, [&](uint32_t bin) { return f.table[bin]; }, [&](uint32_t bin) { return t.head[bin]; },
                                      [&](uint32_t atom) { lits.push_back((uint16_t)atom); });
        walk = walk || wk;
    }
    return walk;
}

}  // namespace pwaf
