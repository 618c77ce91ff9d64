// TEST-ONLY host build of the residual interpreter (pingoo_amd/csrc/residual.h + residual.cpp): the same interpreter source that
// residual_kernel runs per request on the device, compiled with g++ so that the CPU suite can fuzz it against the oracle without a GPU.
// Built into tests/_build/librvm_host.so by tests/test_residual.py; the product library (libpwaf.so) exports nothing like it.
#include <cstring>
#include <string>
#include <vector>

#include "../pingoo_amd/csrc/frontend.h"
#include "../pingoo_amd/csrc/program.h"
#include "../pingoo_amd/csrc/residual.h"

namespace pwaf {
bool rvm_specialize(const uint8_t *blob, size_t len, std::string &out, std::string &why);  // residual_jit.cpp
}

using namespace pwaf;

struct Handle {
    std::vector<uint8_t> blob;
    std::vector<std::string> header_names;
    size_t n_rules = 0;
};

static std::string trim_item(const char *s) {
    std::string t = s ? s : "";
    size_t b = 0, e = t.size();
    auto ws = [](unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); };
    while (b < e && ws((unsigned char)t[b])) b++;
    while (e > b && ws((unsigned char)t[e - 1])) e--;
    return t.substr(b, e - b);
}

extern "C" {

// Compiles n expressions as residual rules. Returns a handle, or nullptr with the reason of the first rule that cannot be lowered
// (rule index in *bad).
void *rvmh_compile(const char *const *exprs, size_t n, const pwaf_list_desc *lists, size_t n_lists, char *why, size_t why_len, int *bad) {
    std::vector<ResidualList> hl;
    for (size_t k = 0; k < n_lists; k++) {
        ResidualList l;
        l.name = lists[k].name;
        l.type = lists[k].type;
        for (uint32_t i = 0; i < lists[k].n_items; i++) {
            std::string item = trim_item(lists[k].items[i]);
            if (l.type == PWAF_LIST_STRING) l.strs.push_back(item);
            else if (l.type == PWAF_LIST_INT) { int64_t v = 0; parse_i64_text(item, v); l.ints.push_back(v); }
            else { PrefixEntry pe; std::string e; if (parse_ipnet_text(item, pe, e)) l.nets.push_back(pe); }
        }
        bool replaced = false;
        for (auto &old : hl) if (old.name == l.name) { old = l; replaced = true; break; }
        if (!replaced) hl.push_back(l);
    }
    auto *h = new Handle;
    ResidualBuilder rb;
    // the headers map's names: what ALL the expressions mention with a literal key, collected first (compile.cpp does the same); once
    // closed, any other name is an absent key (-1)
    bool closed = true;
    auto header_field = [&](const std::string &name) -> int {
        for (size_t k = 0; k < h->header_names.size(); k++) if (h->header_names[k] == name) return PWAF_N_FIELDS + (int)k;
        if (closed) return -1;
        h->header_names.push_back(name);
        return PWAF_N_FIELDS + (int)h->header_names.size() - 1;
    };
    for (size_t k = 0; k < n && closed; k++) {
        Syntax syn;
        std::string perr;
        if (!parse_expression(exprs[k], syn, perr)) closed = false;
        else collect_header_names(syn, h->header_names);
    }
    for (size_t k = 0; k < n; k++) {
        Syntax syn;
        std::string perr, reason;
        if (!parse_expression(exprs[k], syn, perr)) { reason = "syntax: " + perr; }
        else if (rb.compile_rule(syn, hl, header_field, reason, closed ? &h->header_names : nullptr) >= 0) continue;
        snprintf(why, why_len, "%s", reason.c_str());
        *bad = (int)k;
        delete h;
        return nullptr;
    }
    h->blob = rb.blob();
    h->n_rules = rb.n_rules();
    return h;
}
// A handle over a program image taken from pwaf_program_dump (section RVMB): what compile.cpp actually produced for a rule set.
void *rvmh_from_blob(const uint8_t *blob, size_t len) {
    auto *h = new Handle;
    h->blob.assign(blob, blob + len);
    h->n_rules = reinterpret_cast<const rvm::Header *>(h->blob.data())->n_rules;
    return h;
}
// The SPECIALIZED form of the handle's rules (csrc/residual_jit.cpp: the rule functions as portable C++ over residual.h), for
// tests/test_residual_jit.py to compile with g++ and run beside the interpreter. Returns the text's length (0 + why on failure).
size_t rvmh_specialize(void *hv, char *buf, size_t cap, char *why, size_t why_len) {
    Handle *h = (Handle *)hv;
    std::string text, reason;
    if (!rvm_specialize(h->blob.data(), h->blob.size(), text, reason)) { snprintf(why, why_len, "%s", reason.c_str()); return 0; }
    if (buf && cap) { const size_t k = text.size() < cap - 1 ? text.size() : cap - 1; memcpy(buf, text.data(), k); buf[k] = 0; }
    return text.size();
}
const uint8_t *rvmh_blob(void *hv, size_t *len) { Handle *h = (Handle *)hv; *len = h->blob.size(); return h->blob.data(); }
void rvmh_free(void *h) { delete (Handle *)h; }
size_t rvmh_header_count(void *h) { return ((Handle *)h)->header_names.size(); }
const char *rvmh_header_name(void *h, size_t k) { return ((Handle *)h)->header_names[k].c_str(); }

// Evaluates rule `rule` for request r of a host batch (string columns: 5 fields then the handle's header columns, each data + offsets).
int rvmh_eval(void *hv, uint32_t rule, const uint8_t *const *data, const uint32_t *const *off, uint32_t r, const uint8_t *ip16, uint32_t v6, uint32_t port, uint32_t asn, uint32_t country) {
    Handle *h = (Handle *)hv;
    rvm::Machine m;
    m.blob = h->blob.data();
    m.h = reinterpret_cast<const rvm::Header *>(h->blob.data());
    m.q.data = data;
    m.q.off = off;
    m.q.r = r;
    m.q.ip = ip16;
    m.q.v6 = v6;
    m.q.port = port;
    m.q.asn = asn;
    m.q.country = country;
    m.heap_n = 0;
    return (int)rvm::run_rule(m, rule);  // 1 match, 2 execution error, 0 otherwise
}
}
