// iptrie.cpp — CIDR lists and GeoIP prefixes -> multibit radix tries for the device lookup kernel.
//
// Replaces, for the whole batch at once:
//   - `lists["x"].contains(client.ip)`: the reference scans a Vec<IpNetwork> per rule per request
//     (pingoo/lists.rs:14,102-108,119-121). Here ALL ip lists are merged into one trie whose leaf is
//     the id of the request's membership set (which lists contain the address), so one walk answers
//     every ip-list predicate of every rule.
//   - GeoipDB::lookup (pingoo/geoip.rs:73-91): longest-prefix match -> record id.
// Layout: 16-bit root (65536 entries) then 8-bit strides; IPv4 needs <= 3 dependent loads, IPv6
// <= 15 (<= 7 for prefixes up to /64). Entry = leaf flag | value, or child node index.
// Built from sorted, properly nested CIDR ranges: sweep into disjoint elementary intervals, then
// recursive descent that emits a leaf whenever a slot lies inside one interval.
#include <algorithm>
#include <cstring>
#include <map>

#include "program.h"

namespace pwaf {

namespace {

typedef unsigned __int128 u128;

struct Range {
    u128 lo, hi;
    uint32_t payload;
    uint32_t order;
    uint8_t len;
};

struct Breaks {
    std::vector<u128> start;     // sorted interval starts; interval k = [start[k], start[k+1]-1]
    std::vector<uint32_t> value;  // value per interval
};

struct TrieBuilder {
    IpTrie &trie;
    const Breaks &br;
    int total_bits;

    // value of the interval containing x
    size_t find(u128 x) const {
        size_t k = std::upper_bound(br.start.begin(), br.start.end(), x) - br.start.begin();
        return k - 1;
    }
    // fills `slots` (count entries) covering [base, base + count * span) where span = 2^(bits_below)
    void fill(uint32_t *slots_ptr, size_t slots_index, bool in_nodes, u128 base, int count_bits, int bits_below) {
        size_t count = (size_t)1 << count_bits;
        for (size_t s = 0; s < count; s++) {
            u128 lo = base + ((u128)s << bits_below);
            u128 hi = bits_below == 0 ? lo : lo + (((u128)1 << bits_below) - 1);
            size_t a = find(lo), b = find(hi);
            uint32_t entry;
            if (a == b || bits_below == 0) {
                entry = TRIE_LEAF | br.value[a];
            } else {
                // all intervals in [a, b] might still share one value
                bool same = true;
                for (size_t k = a + 1; k <= b; k++) if (br.value[k] != br.value[a]) { same = false; break; }
                if (same) {
                    entry = TRIE_LEAF | br.value[a];
                } else {
                    uint32_t node = trie.n_nodes();
                    trie.nodes.resize(trie.nodes.size() + 256, 0);
                    int stride = bits_below >= 8 ? 8 : bits_below;
                    fill(nullptr, (size_t)node * 256, true, lo, stride, bits_below - stride);
                    entry = node;
                }
            }
            if (in_nodes) trie.nodes[slots_index + s] = entry;  // re-index: nodes may have been reallocated
            else slots_ptr[s] = entry;
        }
    }
};

static void family(const std::vector<PrefixEntry> &prefixes, bool v6, int mode, uint32_t n_lists, IpTrie &trie, std::vector<uint32_t> &root,
                   std::vector<uint32_t> &set_masks, uint32_t set_words, std::map<std::vector<uint32_t>, uint32_t> &set_ids) {
    const int bits = v6 ? 128 : 32;
    std::vector<Range> rs;
    uint32_t ord = 0;
    for (const PrefixEntry &p : prefixes) {
        if (p.v6 != v6) { ord++; continue; }
        u128 a = 0;
        for (int k = 0; k < bits / 8; k++) a = (a << 8) | p.addr[k];
        u128 mask = p.len == 0 ? 0 : (~(u128)0 << (bits - p.len));
        if (bits < 128) mask &= (((u128)1 << bits) - 1);
        Range r;
        r.lo = a & mask;
        u128 hostmask = p.len == bits ? 0 : ((((u128)1 << (bits - p.len - 1)) << 1) - 1);
        r.hi = r.lo | hostmask;
        r.payload = p.payload;
        r.order = ord++;
        r.len = p.len;
        rs.push_back(r);
    }
    if (rs.empty()) { root.clear(); return; }
    // outer ranges first; among identical ranges the later input wins (is innermost)
    std::sort(rs.begin(), rs.end(), [](const Range &x, const Range &y) {
        if (x.lo != y.lo) return x.lo < y.lo;
        if (x.len != y.len) return x.len < y.len;
        return x.order < y.order;
    });
    // sweep: CIDR ranges nest properly, so a stack of open ranges describes the cover at every point
    Breaks br;
    std::vector<const Range *> open;
    std::vector<uint32_t> cur_mask(set_words, 0);
    auto value_now = [&]() -> uint32_t {
        if (mode == 1) return open.empty() ? 0 : open.back()->payload;
        std::fill(cur_mask.begin(), cur_mask.end(), 0);
        for (const Range *r : open) cur_mask[r->payload >> 5] |= 1u << (r->payload & 31);
        auto it = set_ids.find(cur_mask);
        if (it != set_ids.end()) return it->second;
        uint32_t id = (uint32_t)set_ids.size();
        set_ids.emplace(cur_mask, id);
        set_masks.insert(set_masks.end(), cur_mask.begin(), cur_mask.end());
        return id;
    };
    auto emit = [&](u128 at) {
        uint32_t v = value_now();
        if (!br.start.empty() && br.start.back() == at) { br.value.back() = v; return; }
        if (!br.value.empty() && br.value.back() == v) return;
        br.start.push_back(at);
        br.value.push_back(v);
    };
    const u128 maxv = bits == 128 ? ~(u128)0 : (((u128)1 << bits) - 1);
    emit(0);
    size_t i = 0;
    while (i < rs.size() || !open.empty()) {
        // next event: either the next range opens, or the innermost open range closes
        bool open_next = i < rs.size() && (open.empty() || rs[i].lo <= open.back()->hi);
        if (open_next) {
            open.push_back(&rs[i]);
            emit(rs[i].lo);
            i++;
        } else {
            u128 end = open.back()->hi;
            open.pop_back();
            if (end != maxv) emit(end + 1);
            else break;  // closed at the very top of the space: nothing follows
        }
    }
    (void)n_lists;
    root.assign(65536, 0);
    TrieBuilder tb{trie, br, bits};
    tb.fill(root.data(), 0, false, 0, 16, bits - 16);
}

}  // namespace

void build_ip_trie(const std::vector<PrefixEntry> &prefixes, int mode, uint32_t n_lists, IpTrie &trie, std::vector<uint32_t> &set_masks, uint32_t &set_words) {
    trie.root4.clear();
    trie.root6.clear();
    trie.nodes.clear();
    std::map<std::vector<uint32_t>, uint32_t> set_ids;
    if (mode == 0) {
        set_words = std::max<uint32_t>(1, (n_lists + 31) / 32);
        set_masks.assign(set_words, 0);  // set 0 = member of nothing
        set_ids.emplace(std::vector<uint32_t>(set_words, 0), 0);
    }
    family(prefixes, false, mode, n_lists, trie, trie.root4, set_masks, set_words, set_ids);
    family(prefixes, true, mode, n_lists, trie, trie.root6, set_masks, set_words, set_ids);
}

// ---- text parsing of list items (pingoo/lists.rs:90-108) ------------------------------------------------
static bool v4_text(const char *s, size_t n, uint8_t out[4]) {
    size_t p = 0;
    for (int part = 0; part < 4; part++) {
        size_t b = p;
        unsigned v = 0;
        while (p < n && s[p] >= '0' && s[p] <= '9' && p - b < 4) v = v * 10 + (unsigned)(s[p++] - '0');
        size_t digits = p - b;
        if (digits == 0 || digits > 3 || v > 255) return false;
        if (digits > 1 && s[b] == '0') return false;  // Ipv4Addr::from_str rejects leading zeros
        out[part] = (uint8_t)v;
        if (part < 3) {
            if (p >= n || s[p] != '.') return false;
            p++;
        }
    }
    return p == n;
}

static bool v6_text(const char *s, size_t n, uint8_t out[16]) {
    // Ipv6Addr::from_str: up to 8 hextets, one "::", optional dotted-quad tail
    uint16_t g[8];
    int ng = 0, gap = -1;
    size_t p = 0;
    if (n >= 2 && s[0] == ':' && s[1] == ':') { gap = 0; p = 2; }
    else if (n >= 1 && s[0] == ':') return false;
    while (p < n) {
        if (ng == 8) return false;
        // dotted quad tail?
        size_t e = p;
        bool dotted = false;
        while (e < n && s[e] != ':') { if (s[e] == '.') dotted = true; e++; }
        if (dotted) {
            uint8_t q[4];
            if (e != n || ng > 6 || !v4_text(s + p, n - p, q)) return false;
            g[ng++] = (uint16_t)(q[0] << 8 | q[1]);
            g[ng++] = (uint16_t)(q[2] << 8 | q[3]);
            p = n;
            break;
        }
        if (e == p || e - p > 4) return false;
        unsigned v = 0;
        for (size_t k = p; k < e; k++) {
            char c = s[k];
            int h = (c >= '0' && c <= '9') ? c - '0' : ((c | 0x20) >= 'a' && (c | 0x20) <= 'f') ? (c | 0x20) - 'a' + 10 : -1;
            if (h < 0) return false;
            v = v << 4 | (unsigned)h;
        }
        g[ng++] = (uint16_t)v;
        p = e;
        if (p == n) break;
        // s[p] == ':'
        if (p + 1 < n && s[p + 1] == ':') {
            if (gap >= 0) return false;
            gap = ng;
            p += 2;
        } else {
            p++;
            if (p == n) return false;  // trailing ':'
        }
    }
    if (gap < 0 && ng != 8) return false;
    if (gap >= 0 && ng > 7) return false;
    uint16_t full[8] = {0};
    if (gap < 0) memcpy(full, g, sizeof full);
    else {
        for (int k = 0; k < gap; k++) full[k] = g[k];
        for (int k = gap; k < ng; k++) full[8 - (ng - k)] = g[k];
    }
    for (int k = 0; k < 8; k++) { out[2 * k] = (uint8_t)(full[k] >> 8); out[2 * k + 1] = (uint8_t)full[k]; }
    return true;
}

bool parse_ipnet_text(const std::string &s, PrefixEntry &out, std::string &err) {
    size_t slash = s.find('/');
    std::string a = slash == std::string::npos ? s : s.substr(0, slash);
    memset(out.addr, 0, 16);
    if (v4_text(a.data(), a.size(), out.addr)) { out.v6 = false; out.len = 32; }
    else if (v6_text(a.data(), a.size(), out.addr)) { out.v6 = true; out.len = 128; }
    else { err = "invalid address: " + s; return false; }
    if (slash == std::string::npos) return true;
    std::string p = s.substr(slash + 1);
    uint8_t m[4];
    if (!out.v6 && v4_text(p.data(), p.size(), m)) {
        // ipnetwork accepts a dotted netmask for IPv4 ("10.0.0.0/255.0.0.0")
        uint32_t mask = (uint32_t)m[0] << 24 | (uint32_t)m[1] << 16 | (uint32_t)m[2] << 8 | m[3];
        uint32_t inv = ~mask;
        if ((inv & (inv + 1)) != 0) { err = "invalid prefix"; return false; }
        out.len = (uint8_t)__builtin_popcount(mask);
        return true;
    }
    if (p.empty() || p.size() > 3) { err = "invalid prefix"; return false; }
    unsigned v = 0;
    for (char c : p) {
        if (c < '0' || c > '9') { err = "invalid prefix"; return false; }
        v = v * 10 + (unsigned)(c - '0');
    }
    if (v > (out.v6 ? 128u : 32u)) { err = "invalid prefix"; return false; }
    out.len = (uint8_t)v;
    return true;
}

bool parse_i64_text(const std::string &s, int64_t &out) {
    // i64::from_str: optional sign, >= 1 digit, no whitespace, overflow is an error
    size_t p = 0;
    bool neg = false;
    if (!s.empty() && (s[0] == '+' || s[0] == '-')) { neg = s[0] == '-'; p = 1; }
    if (p >= s.size()) return false;
    uint64_t mag = 0;
    for (; p < s.size(); p++) {
        if (s[p] < '0' || s[p] > '9') return false;
        unsigned d = (unsigned)(s[p] - '0');
        if (mag > (UINT64_MAX - d) / 10) return false;
        mag = mag * 10 + d;
    }
    if (neg) {
        if (mag > (uint64_t)1 << 63) return false;
        out = mag == (uint64_t)1 << 63 ? INT64_MIN : -(int64_t)mag;
    } else {
        if (mag > (uint64_t)INT64_MAX) return false;
        out = (int64_t)mag;
    }
    return true;
}

}  // namespace pwaf
