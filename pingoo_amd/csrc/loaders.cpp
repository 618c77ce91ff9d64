// loaders.cpp — host-side readers for the data files that feed the rule path (SURVEY.md §8f "next" rows):
//   * a MaxMind DB (.mmdb) reader that flattens the database into the pwaf_geoip_entry prefix table the engine consumes
//     (the reference opens the same file with the `maxminddb` crate and looks addresses up one by one:
//     pingoo/geoip.rs:43-72,73-91; record = {asn: "AS1234" string, country: 2 x 'A'..'Z'}: geoip.rs:17-23, serde_utils.rs:1-9);
//   * the list-file parser (CSV, first column, trimmed: pingoo/lists.rs:62-117).
// Written from the public MaxMind DB file format specification v2.0; no code of the reference or of libmaxminddb is used.
// Plain C++17, no device code.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

#include "pwaf.h"

namespace pwaf {
int fail(int code, const std::string &msg);  // engine.cpp: records pwaf_last_error()
}
using pwaf::fail;

namespace {

struct Mmdb {
    const uint8_t *p = nullptr;
    size_t len = 0;
    size_t data_start = 0, data_end = 0;  // data section [start, end)
    uint32_t node_count = 0, record_size = 0, ip_version = 0;
};

enum : uint32_t { T_POINTER = 1, T_STRING = 2, T_DOUBLE = 3, T_BYTES = 4, T_U16 = 5, T_U32 = 6, T_MAP = 7, T_I32 = 8, T_U64 = 9, T_U128 = 10,
                  T_ARRAY = 11, T_CACHE = 12, T_END = 13, T_BOOL = 14, T_FLOAT = 15 };

struct Field {
    uint32_t type = 0;
    size_t size = 0;     // payload bytes, or entry count for map / array, or the value for bool
    size_t payload = 0;  // absolute offset of the payload
    size_t next = 0;     // absolute offset just past this field's control bytes (+ payload for scalar types) IN THE STREAM IT WAS READ
                         // FROM: after a pointer this is the byte after the pointer, not after the pointed-to value
};

// Decodes the control bytes at `at` (absolute offset, inside [lo, hi)). Pointers are followed once (`base` = start of the
// section pointers are relative to). Returns false on a malformed field.
bool read_field(const Mmdb &db, size_t at, size_t lo, size_t hi, size_t base, Field &f, int depth = 0) {
    if (at < lo || at >= hi) return false;
    const uint8_t ctrl = db.p[at++];
    uint32_t type = ctrl >> 5;
    if (type == T_POINTER) {
        const uint32_t ss = (ctrl >> 3) & 3u, v = ctrl & 7u;
        if (at + ss + 1 > hi) return false;
        size_t ptr;
        if (ss == 0) ptr = ((size_t)v << 8) | db.p[at];
        else if (ss == 1) ptr = 2048 + (((size_t)v << 16) | ((size_t)db.p[at] << 8) | db.p[at + 1]);
        else if (ss == 2) ptr = 526336 + (((size_t)v << 24) | ((size_t)db.p[at] << 16) | ((size_t)db.p[at + 1] << 8) | db.p[at + 2]);
        else ptr = ((size_t)db.p[at] << 24) | ((size_t)db.p[at + 1] << 16) | ((size_t)db.p[at + 2] << 8) | db.p[at + 3];
        const size_t after = at + ss + 1;
        if (depth > 0) return false;  // a pointer must not point at a pointer
        if (!read_field(db, base + ptr, lo, hi, base, f, depth + 1)) return false;
        f.next = after;
        return true;
    }
    if (type == 0) {  // extended type
        if (at >= hi) return false;
        type = 7u + db.p[at++];
    }
    size_t size = ctrl & 0x1Fu;
    if (size == 29) {
        if (at + 1 > hi) return false;
        size = 29 + (size_t)db.p[at];
        at += 1;
    } else if (size == 30) {
        if (at + 2 > hi) return false;
        size = 285 + (((size_t)db.p[at] << 8) | db.p[at + 1]);
        at += 2;
    } else if (size == 31) {
        if (at + 3 > hi) return false;
        size = 65821 + (((size_t)db.p[at] << 16) | ((size_t)db.p[at + 1] << 8) | db.p[at + 2]);
        at += 3;
    }
    f.type = type;
    f.size = size;
    f.payload = at;
    if (type == T_MAP || type == T_ARRAY || type == T_BOOL || type == T_END || type == T_CACHE) {
        f.next = at;  // containers: the entries follow; bool / end marker: no payload
    } else {
        if (at + size > hi) return false;
        f.next = at + size;
    }
    return true;
}

// Skips one complete value (recursing into containers) starting at `at`; returns the offset after it, or 0 on error.
size_t skip_value(const Mmdb &db, size_t at, size_t lo, size_t hi, size_t base, int depth = 0) {
    if (depth > 32) return 0;
    Field f;
    if (!read_field(db, at, lo, hi, base, f)) return 0;
    if ((db.p[at] >> 5) == T_POINTER) return f.next;  // the pointed-to value lives elsewhere; the stream continues after the pointer
    if (f.type == T_MAP) {
        size_t cur = f.next;
        for (size_t k = 0; k < f.size; k++) {
            cur = skip_value(db, cur, lo, hi, base, depth + 1);  // key
            if (!cur) return 0;
            cur = skip_value(db, cur, lo, hi, base, depth + 1);  // value
            if (!cur) return 0;
        }
        return cur;
    }
    if (f.type == T_ARRAY) {
        size_t cur = f.next;
        for (size_t k = 0; k < f.size; k++) {
            cur = skip_value(db, cur, lo, hi, base, depth + 1);
            if (!cur) return 0;
        }
        return cur;
    }
    return f.next;
}

bool field_uint(const Mmdb &db, const Field &f, uint64_t &out) {
    if (f.type != T_U16 && f.type != T_U32 && f.type != T_U64) return false;
    if (f.size > 8) return false;
    uint64_t v = 0;
    for (size_t k = 0; k < f.size; k++) v = (v << 8) | db.p[f.payload + k];
    out = v;
    return true;
}

// Looks `key` up in the map whose control bytes are at `at`; on success `val` is the (pointer-resolved) value field.
bool map_get(const Mmdb &db, size_t at, size_t lo, size_t hi, size_t base, const char *key, Field &val) {
    Field m;
    if (!read_field(db, at, lo, hi, base, m) || m.type != T_MAP) return false;
    size_t cur = m.payload;  // entries follow the map's control bytes (the pointed-to ones, if the map was reached through a pointer)
    const size_t klen = strlen(key);
    for (size_t k = 0; k < m.size; k++) {
        Field kf;
        if (!read_field(db, cur, lo, hi, base, kf) || kf.type != T_STRING) return false;
        const size_t after_key = kf.next;
        Field vf;
        if (!read_field(db, after_key, lo, hi, base, vf)) return false;
        if (kf.size == klen && memcmp(db.p + kf.payload, key, klen) == 0) {
            val = vf;
            return true;
        }
        cur = skip_value(db, after_key, lo, hi, base);
        if (!cur) return false;
    }
    return false;
}

bool open_mmdb(const uint8_t *p, size_t len, Mmdb &db, std::string &err) {
    static const uint8_t marker[] = {0xAB, 0xCD, 0xEF, 'M', 'a', 'x', 'M', 'i', 'n', 'd', '.', 'c', 'o', 'm'};
    if (!p || len < sizeof marker + 16) { err = "file too short"; return false; }
    size_t meta = 0;
    bool found = false;
    for (size_t k = len - sizeof marker + 1; k-- > 0;) {  // LAST occurrence
        if (memcmp(p + k, marker, sizeof marker) == 0) { meta = k + sizeof marker; found = true; break; }
        if (len - k > 128 * 1024 + sizeof marker) break;  // the metadata block is at most 128 KiB
    }
    if (!found) { err = "metadata marker not found"; return false; }
    db.p = p;
    db.len = len;
    Field f;
    uint64_t v;
    if (!map_get(db, meta, meta, len, meta, "node_count", f) || !field_uint(db, f, v)) { err = "metadata: node_count missing"; return false; }
    db.node_count = (uint32_t)v;
    if (!map_get(db, meta, meta, len, meta, "record_size", f) || !field_uint(db, f, v)) { err = "metadata: record_size missing"; return false; }
    db.record_size = (uint32_t)v;
    if (!map_get(db, meta, meta, len, meta, "ip_version", f) || !field_uint(db, f, v)) { err = "metadata: ip_version missing"; return false; }
    db.ip_version = (uint32_t)v;
    if (db.record_size != 24 && db.record_size != 28 && db.record_size != 32) { err = "unsupported record_size"; return false; }
    if (db.ip_version != 4 && db.ip_version != 6) { err = "unsupported ip_version"; return false; }
    const size_t tree = (size_t)db.node_count * db.record_size / 4;  // two records per node
    if (tree + 16 > meta - sizeof marker) { err = "search tree larger than the file"; return false; }
    db.data_start = tree + 16;
    db.data_end = meta - sizeof marker;
    return true;
}

inline uint32_t record_of(const Mmdb &db, uint32_t node, int side) {
    const uint8_t *n = db.p + (size_t)node * db.record_size / 4;
    if (db.record_size == 24) return side == 0 ? ((uint32_t)n[0] << 16) | ((uint32_t)n[1] << 8) | n[2] : ((uint32_t)n[3] << 16) | ((uint32_t)n[4] << 8) | n[5];
    if (db.record_size == 28)
        return side == 0 ? ((uint32_t)(n[3] >> 4) << 24) | ((uint32_t)n[0] << 16) | ((uint32_t)n[1] << 8) | n[2]
                         : ((uint32_t)(n[3] & 0x0F) << 24) | ((uint32_t)n[4] << 16) | ((uint32_t)n[5] << 8) | n[6];
    return side == 0 ? ((uint32_t)n[0] << 24) | ((uint32_t)n[1] << 16) | ((uint32_t)n[2] << 8) | n[3]
                     : ((uint32_t)n[4] << 24) | ((uint32_t)n[5] << 16) | ((uint32_t)n[6] << 8) | n[7];
}

// The record as the reference deserialises it (geoip.rs:17-23): both fields must be strings, the country two upper-case ASCII
// letters; the asn keeps its digits after any leading "AS"s and is 0 when they do not parse (serde_utils.rs:5-8). Anything else
// makes `lookup` fail, and the caller then uses the default record {0, "XX"} (http_listener.rs:143-157).
void decode_record(const Mmdb &db, size_t off, uint32_t &asn, uint8_t country[2]) {
    asn = 0;
    country[0] = country[1] = 'X';
    Field fa, fc;
    const size_t lo = db.data_start, hi = db.data_end;
    if (!map_get(db, off, lo, hi, lo, "asn", fa) || fa.type != T_STRING) return;
    if (!map_get(db, off, lo, hi, lo, "country", fc) || fc.type != T_STRING) return;
    if (fc.size != 2) return;
    const uint8_t c0 = db.p[fc.payload], c1 = db.p[fc.payload + 1];
    if (c0 < 'A' || c0 > 'Z' || c1 < 'A' || c1 > 'Z') return;
    std::string s((const char *)db.p + fa.payload, fa.size);
    size_t b = 0;
    while (s.compare(b, 2, "AS") == 0) b += 2;  // trim_start_matches("AS"): every leading repetition
    uint64_t v = 0;
    bool ok = b < s.size();
    size_t k = b;
    if (ok && s[k] == '+') k++;  // Rust's u32::from_str accepts one leading '+'
    ok = ok && k < s.size();
    for (; ok && k < s.size(); k++) {
        if (s[k] < '0' || s[k] > '9') ok = false;
        else {
            v = v * 10 + (uint64_t)(s[k] - '0');
            if (v > 0xFFFFFFFFull) ok = false;
        }
    }
    country[0] = c0;
    country[1] = c1;
    asn = ok ? (uint32_t)v : 0u;
}

struct Walker {
    const Mmdb &db;
    std::vector<pwaf_geoip_entry> out;
    uint8_t bits[16] = {0};
    uint32_t max_depth;
    bool overflow = false;

    void leaf(uint32_t depth, uint32_t rec) {
        const size_t off = db.data_start + (size_t)(rec - db.node_count - 16);
        uint32_t asn;
        uint8_t cc[2];
        if (rec < db.node_count + 16 || off >= db.data_end) { asn = 0; cc[0] = cc[1] = 'X'; }
        else decode_record(db, off, asn, cc);
        pwaf_geoip_entry e{};
        e.asn = asn;
        e.country[0] = cc[0];
        e.country[1] = cc[1];
        if (db.ip_version == 4) {
            memcpy(e.addr, bits, 4);
            e.prefix_len = (uint8_t)depth;
            e.is_v6 = 0;
            out.push_back(e);
            return;
        }
        memcpy(e.addr, bits, 16);
        e.prefix_len = (uint8_t)depth;
        e.is_v6 = 1;
        out.push_back(e);
        // IPv4 addresses are looked up below ::/96 (the `maxminddb` crate starts an IPv4 lookup at the node reached by 96 zero
        // bits, or at the record that ends the walk earlier)
        bool zero96 = true;
        for (int k = 0; k < 12; k++) zero96 = zero96 && bits[k] == 0;
        if (!zero96) return;
        pwaf_geoip_entry v4{};
        v4.asn = asn;
        v4.country[0] = cc[0];
        v4.country[1] = cc[1];
        v4.is_v6 = 0;
        if (depth >= 96) {
            memcpy(v4.addr, bits + 12, 4);
            v4.prefix_len = (uint8_t)(depth - 96);
        } else {
            v4.prefix_len = 0;  // a record above ::/96 on the all-zero path covers every IPv4 address
        }
        out.push_back(v4);
    }

    void walk(uint32_t node, uint32_t depth) {
        if (overflow) return;
        for (int side = 0; side < 2; side++) {
            const uint32_t rec = record_of(db, node, side);
            if (rec == db.node_count) continue;  // no data below
            if (side) bits[depth >> 3] |= (uint8_t)(0x80u >> (depth & 7));
            if (rec > db.node_count) {
                leaf(depth + 1, rec);
            } else if (depth + 1 >= max_depth) {
                overflow = true;  // a node where the address has run out of bits: corrupt tree
            } else {
                walk(rec, depth + 1);
            }
            if (side) bits[depth >> 3] &= (uint8_t)~(0x80u >> (depth & 7));
            // a tree with node_count inner nodes has at most node_count + 1 leaves (each may add one IPv4 alias): more means shared
            // children, i.e. a crafted DAG that would expand exponentially
            if (out.size() > 2 * ((size_t)db.node_count + 1)) overflow = true;
        }
    }
};

}  // namespace

extern "C" {

int pwaf_geoip_from_mmdb(const uint8_t *mmdb, size_t len, pwaf_geoip_entry **entries_out, size_t *n_out) {
    if (!entries_out || !n_out) return fail(PWAF_E_INVALID_ARG, "NULL output argument");
    *entries_out = nullptr;
    *n_out = 0;
    Mmdb db;
    std::string err;
    if (!open_mmdb(mmdb, len, db, err)) return fail(PWAF_E_INVALID_ARG, "mmdb file is not valid: " + err);
    if (db.node_count == 0) return PWAF_OK;
    Walker w{db};
    w.max_depth = db.ip_version == 4 ? 32 : 128;
    w.walk(0, 0);
    if (w.overflow) return fail(PWAF_E_INVALID_ARG, "mmdb file is not valid: search tree is deeper than the address or unreasonably large");
    pwaf_geoip_entry *mem = (pwaf_geoip_entry *)malloc(std::max<size_t>(1, w.out.size()) * sizeof(pwaf_geoip_entry));
    if (!mem) return fail(PWAF_E_NOMEM, "out of memory");
    if (!w.out.empty()) memcpy(mem, w.out.data(), w.out.size() * sizeof(pwaf_geoip_entry));
    *entries_out = mem;
    *n_out = w.out.size();
    return PWAF_OK;
}

void pwaf_geoip_free(pwaf_geoip_entry *entries) { free(entries); }

// zstd::decode_all (pingoo/geoip.rs:49-55: a database whose path ends in .zst is ZSTD-compressed; the reference's Docker image ships
// geoip.mmdb.zst, config.rs:31-36). libzstd is loaded at run time (this image has libzstd.so.1 but no headers): the stable
// streaming API, one frame after another until the input is consumed.
int pwaf_zstd_decompress(const uint8_t *src, size_t len, uint8_t **out, size_t *out_len) {
    if (!out || !out_len || (!src && len)) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    *out = nullptr;
    *out_len = 0;
    struct InBuf { const void *src; size_t size, pos; };
    struct OutBuf { void *dst; size_t size, pos; };
    static void *lib = nullptr;
    static void *(*create)() = nullptr;
    static size_t (*destroy)(void *) = nullptr;
    static size_t (*run)(void *, OutBuf *, InBuf *) = nullptr;
    static unsigned (*is_error)(size_t) = nullptr;
    static const char *(*error_name)(size_t) = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
        create = (void *(*)())dlsym(lib, "ZSTD_createDStream");
        destroy = (size_t (*)(void *))dlsym(lib, "ZSTD_freeDStream");
        run = (size_t (*)(void *, OutBuf *, InBuf *))dlsym(lib, "ZSTD_decompressStream");
        is_error = (unsigned (*)(size_t))dlsym(lib, "ZSTD_isError");
        error_name = (const char *(*)(size_t))dlsym(lib, "ZSTD_getErrorName");
    });
    if (!lib || !create || !destroy || !run || !is_error) return fail(PWAF_E_UNSUPPORTED, "zstd is not available: libzstd.so.1 could not be loaded");
    void *ds = create();
    if (!ds) return fail(PWAF_E_NOMEM, "ZSTD_createDStream failed");
    std::vector<uint8_t> buf;
    try {
        buf.resize(std::max<size_t>(1 << 20, len * 4));
    } catch (const std::bad_alloc &) {
        destroy(ds);
        return fail(PWAF_E_NOMEM, "out of memory");
    }
    InBuf in{src, len, 0};
    size_t produced = 0, last = 1;
    while (in.pos < in.size || last != 0) {
        if (produced == buf.size()) {
            if (buf.size() > ((size_t)4 << 30)) { destroy(ds); return fail(PWAF_E_INVALID_ARG, "error decompressing geoip database: output larger than 4 GiB"); }
            try { buf.resize(buf.size() * 2); } catch (const std::bad_alloc &) { destroy(ds); return fail(PWAF_E_NOMEM, "out of memory"); }
        }
        OutBuf ob{buf.data(), buf.size(), produced};
        const size_t before_in = in.pos;
        last = run(ds, &ob, &in);
        if (is_error(last)) {
            const std::string why = error_name ? error_name(last) : "corrupt frame";
            destroy(ds);
            return fail(PWAF_E_INVALID_ARG, "error decompressing geoip database: " + why);
        }
        const bool progressed = ob.pos != produced || in.pos != before_in;
        produced = ob.pos;
        if (in.pos == in.size && last == 0) break;  // every frame complete
        if (!progressed && in.pos == in.size) { destroy(ds); return fail(PWAF_E_INVALID_ARG, "error decompressing geoip database: truncated frame"); }
    }
    destroy(ds);
    uint8_t *mem = (uint8_t *)malloc(std::max<size_t>(1, produced));
    if (!mem) return fail(PWAF_E_NOMEM, "out of memory");
    memcpy(mem, buf.data(), produced);
    *out = mem;
    *out_len = produced;
    return PWAF_OK;
}
void pwaf_buffer_free(uint8_t *p) { free(p); }

// GeoipDB::load on a file image: `.zst` names are decompressed first (geoip.rs:49-55), then the MMDB is flattened.
int pwaf_geoip_from_file_image(const char *path_for_suffix, const uint8_t *content, size_t len, pwaf_geoip_entry **entries_out, size_t *n_out) {
    const std::string p = path_for_suffix ? path_for_suffix : "";
    if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".zst") == 0) {
        uint8_t *raw = nullptr;
        size_t raw_len = 0;
        int rc = pwaf_zstd_decompress(content, len, &raw, &raw_len);
        if (rc) return rc;
        rc = pwaf_geoip_from_mmdb(raw, raw_len, entries_out, n_out);
        free(raw);
        return rc;
    }
    return pwaf_geoip_from_mmdb(content, len, entries_out, n_out);
}

// lists.rs:62-117: CSV without headers, 1 or 2 columns per record ("flexible"), the FIRST column trimmed is the item. Quoted
// fields follow RFC 4180 ("" is an escaped quote). Empty lines are skipped (the csv crate does not yield them).
int pwaf_list_parse_csv(const char *text, size_t len, char ***items_out, size_t *n_out) {
    if (!items_out || !n_out || (!text && len)) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    *items_out = nullptr;
    *n_out = 0;
    std::vector<std::string> items;
    size_t pos = 0, line = 0;
    while (pos < len) {
        // one record
        std::vector<std::string> cols;
        std::string cur;
        bool in_quotes = false, any = false, record_done = false;
        while (pos < len && !record_done) {
            const char c = text[pos];
            if (in_quotes) {
                if (c == '"') {
                    if (pos + 1 < len && text[pos + 1] == '"') { cur.push_back('"'); pos += 2; }
                    else { in_quotes = false; pos++; }
                } else { cur.push_back(c); pos++; }
                continue;
            }
            if (c == '"' && cur.empty()) { in_quotes = true; any = true; pos++; }
            else if (c == ',') { cols.push_back(cur); cur.clear(); any = true; pos++; }
            else if (c == '\n' || c == '\r') {
                pos++;
                if (c == '\r' && pos < len && text[pos] == '\n') pos++;
                record_done = true;
            } else { cur.push_back(c); any = true; pos++; }
        }
        if (in_quotes) return fail(PWAF_E_LIST, "error parsing list at line " + std::to_string(line + 1) + ": unterminated quoted field");
        if (!any && cur.empty()) continue;  // empty line
        cols.push_back(cur);
        line++;
        if (cols.size() > 2 || cols.empty())
            return fail(PWAF_E_LIST, "error parsing list at line " + std::to_string(line) + ": invalid number of columns. Min: 1, Max: 2");
        std::string &v = cols[0];
        size_t b = 0, e = v.size();
        auto ws = [](unsigned char ch) { return ch == ' ' || (ch >= 9 && ch <= 13); };
        while (b < e && ws((unsigned char)v[b])) b++;
        while (e > b && ws((unsigned char)v[e - 1])) e--;
        items.push_back(v.substr(b, e - b));
    }
    char **arr = (char **)malloc(std::max<size_t>(1, items.size()) * sizeof(char *));
    if (!arr) return fail(PWAF_E_NOMEM, "out of memory");
    for (size_t k = 0; k < items.size(); k++) {
        arr[k] = (char *)malloc(items[k].size() + 1);
        if (!arr[k]) {
            for (size_t q = 0; q < k; q++) free(arr[q]);
            free(arr);
            return fail(PWAF_E_NOMEM, "out of memory");
        }
        memcpy(arr[k], items[k].c_str(), items[k].size() + 1);
    }
    *items_out = arr;
    *n_out = items.size();
    return PWAF_OK;
}

void pwaf_list_free(char **items, size_t n) {
    if (!items) return;
    for (size_t k = 0; k < n; k++) free(items[k]);
    free(items);
}

}  // extern "C"
