// residual_jit.cpp — the SPECIALIZED form of the residual rules: the stack programs of residual.h translated to straight-line code.
//
// The reference evaluates any valid expression per request (Program::execute, pingoo/rules.rs:37-51); rules outside the column
// compiler's subset are lowered to stack programs (residual.cpp) that residual_kernel INTERPRETS with one lane per request: decode,
// dispatch, a dynamically indexed value stack in private memory — 17.6 ms per 10M requests for 8 rules (round 4), 10x the whole column
// pipeline. Interpretation is the cost, not the rules: the same programs, translated once per rule set into calls of residual.h's own
// operation functions with every stack slot a named local and every constant a literal, are ordinary device code — the types of most
// values fold at compile time, the stack lives in registers, lanes of a wave run the same instruction stream. This file is that
// translation (plain C++: no device needed; the CPU suite compiles its output with g++ and fuzzes it against the oracle); engine
// creation hands the text to hiprtc for the device it runs on (rtc.cpp) and falls back to the interpreter kernel when that is not
// possible. Semantics are shared by construction: both forms call the same op_* functions of residual.h.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "program.h"
#include "residual.h"

namespace pwaf {

namespace {
using namespace rvm;

const char kResidualHeaderText[] =
#include "residual_h.inc"
    ;

std::string u(uint64_t v) { return std::to_string(v) + "ull"; }
std::string u32(uint32_t v) { return std::to_string(v) + "u"; }
std::string slot(int d) { return "s" + std::to_string(d); }

// One rule: instructions [entry, its R_END]. The stack depth at every instruction is static (the compiler produced the program by a
// post-order walk; both arms of && || ?: leave the same depth), so slot k of the stack is the local variable s<k>.
bool translate_rule(const Header &h, const uint8_t *blob, uint32_t rule, std::string &out, std::string &why) {
    const Ins *code = reinterpret_cast<const Ins *>(blob + h.code);
    const Val *consts = reinterpret_cast<const Val *>(blob + h.consts);
    const uint32_t entry = reinterpret_cast<const uint32_t *>(blob + h.rules)[rule];
    const uint32_t n_code = (h.consts - h.code) / (uint32_t)sizeof(Ins);  // (sections follow each other: an upper bound)
    // pass 1: extent, jump targets and the depth they are reached with
    uint32_t end = entry;
    while (end < n_code && code[end].op != R_END) end++;
    if (end >= n_code) { why = "residual program without an end"; return false; }
    std::vector<int> depth_at(end + 2, -1);
    std::vector<uint8_t> is_target(end + 2, 0);
    int max_depth = 0;
    {
        int d = 0;
        bool live = true;  // false right after an unconditional jump: the next instruction is only reached through its label
        for (uint32_t pc = entry; pc <= end; pc++) {
            const Ins in = code[pc];
            if (depth_at[pc] >= 0) {
                if (live && depth_at[pc] != d) { why = "residual program: inconsistent stack depth"; return false; }
                d = depth_at[pc];
                live = true;
            } else if (!live) {
                why = "residual program: unreachable code";
                return false;
            }
            auto target = [&](uint32_t t, int dt) -> bool {
                if (t <= pc || t > end) return false;  // forward jumps only
                if (depth_at[t] >= 0 && depth_at[t] != dt) return false;
                depth_at[t] = dt;
                is_target[t] = 1;
                return true;
            };
            bool ok = true;
            switch (in.op) {
                case R_END: if (d != 1) ok = false; break;
                case R_CONST: case R_FIELD: case R_COUNTRY: case R_PORT: case R_ASN: case R_IP: case R_CLIST: d += 1; break;
                case R_NOT: case R_NEG: case R_BOOL_CHK: case R_SELECT: ok = d >= 1; break;
                case R_INDEX: case R_BIN: ok = d >= 2; d -= 1; break;
                case R_CALL: { const int argc = in.b >> 12; ok = argc <= 1 && d >= argc + 1; d -= argc; break; }
                case R_AND_L: case R_OR_L: ok = d >= 1 && target(in.b, d); d -= 1; break;  // decided: the operand (or an error) is back on the stack at the target
                case R_COND: {
                    ok = d >= 1 && in.b >= 1 && in.b - 1 > pc && code[in.b - 1].op == R_JMP;
                    d -= 1;
                    ok = ok && target(in.b, d) && target(code[in.b - 1].b, d + 1);  // else branch; the end (an error goes straight there with one value pushed)
                    break;
                }
                case R_JMP: ok = target(in.b, d); live = false; break;
                case R_FAIL: ok = d >= (int)in.b; d -= (int)in.b; d += 1; break;
                case R_MKLIST: ok = d >= (int)in.b; d -= (int)in.b; d += 1; break;
                case R_MKMAP: ok = d >= 2 * (int)in.b; d -= 2 * (int)in.b; d += 1; break;
                default: ok = false;
            }
            if (!ok || d < 0 || d > (int)kStack) { why = "residual program: malformed instruction at " + std::to_string(pc); return false; }
            if (d > max_depth) max_depth = d;
        }
    }
    // pass 2: text
    std::string s;
    // (a heap of its own per rule: a rule that reads its heap through an index the compiler cannot resolve then keeps ITS heap in
    // scratch memory, not every rule's — the others' stay in registers)
    s += "PWAF_RVM_RULE uint32_t rvm_rule_" + std::to_string(rule) + "(const Machine &m0) {\n";
    s += "    Machine m;\n    m.blob = m0.blob;\n    m.h = m0.h;\n    m.q = m0.q;\n    m.heap_n = 0;\n";
    s += "    Val ";
    for (int k = 0; k < std::max(1, max_depth); k++) s += (k ? ", " : "") + slot(k);
    s += ";\n";
    int d = 0;
    for (uint32_t pc = entry; pc <= end; pc++) {
        const Ins in = code[pc];
        if (depth_at[pc] >= 0) d = depth_at[pc];
        if (is_target[pc]) s += "L" + std::to_string(pc) + ":;\n";
        const std::string top = d >= 1 ? slot(d - 1) : std::string(), below = d >= 2 ? slot(d - 2) : std::string();
        switch (in.op) {
            case R_END: s += "    return rule_result(" + slot(0) + ");\n"; break;
            case R_CONST: {
                const Val &c = consts[in.b];
                s += "    " + slot(d) + " = mk(" + u32(c.t) + ", " + u32(c.a) + ", " + u(c.p) + ");\n";
                d++;
                break;
            }
            case R_FIELD: s += "    " + slot(d) + " = op_field(m, " + u32(in.b) + ");\n"; d++; break;
            case R_COUNTRY: s += "    " + slot(d) + " = op_country(m);\n"; d++; break;
            case R_PORT: s += "    " + slot(d) + " = op_port(m);\n"; d++; break;
            case R_ASN: s += "    " + slot(d) + " = op_asn(m);\n"; d++; break;
            case R_IP: s += "    " + slot(d) + " = mk(T_IP);\n"; d++; break;
            case R_CLIST: s += "    " + slot(d) + " = mk(T_CLIST, " + u32(in.b) + ");\n"; d++; break;
            case R_NOT: s += "    " + top + " = op_not(" + top + ");\n"; break;
            case R_NEG: s += "    " + top + " = op_neg(" + top + ");\n"; break;
            case R_BOOL_CHK: s += "    " + top + " = op_bool_chk(" + top + ");\n"; break;
            case R_AND_L: case R_OR_L:
                // (decided: the slot keeps the result and the right operand is skipped; else the right operand overwrites the slot)
                s += "    { Val o; if (op_logic_left(" + std::string(in.op == R_OR_L ? "true" : "false") + ", " + top + ", o)) { " + top + " = o; goto L" + std::to_string(in.b) + "; } }\n";
                d--;
                break;
            case R_COND:
                s += "    { const uint32_t c = op_cond(" + top + "); if (c == 2u) { " + top + " = mk(T_ERR); goto L" + std::to_string(code[in.b - 1].b) + "; } if (c == 1u) goto L" +
                     std::to_string(in.b) + "; }\n";
                d--;
                break;
            case R_JMP: s += "    goto L" + std::to_string(in.b) + ";\n"; break;
            case R_FAIL: d -= (int)in.b; s += "    " + slot(d) + " = mk(T_ERR);\n"; d++; break;
            case R_MKLIST: case R_MKMAP: {
                const int n = in.op == R_MKLIST ? (int)in.b : 2 * (int)in.b;
                d -= n;
                s += "    { const Val it[" + std::to_string(std::max(1, n)) + "] = {";
                for (int k = 0; k < n; k++) s += (k ? ", " : "") + slot(d + k);
                if (n == 0) s += "mk(T_ERR)";
                s += "}; " + slot(d) + " = " + (in.op == R_MKLIST ? "op_mklist" : "op_mkmap") + "(m, it, " + u32(in.b) + "); }\n";
                d++;
                break;
            }
            case R_INDEX: s += "    " + below + " = op_index(m, " + below + ", " + top + ");\n"; d--; break;
            case R_SELECT: {
                const Val &c = consts[in.b];
                s += "    " + top + " = op_select(m, " + top + ", mk(" + u32(c.t) + ", " + u32(c.a) + ", " + u(c.p) + "));\n";
                break;
            }
            case R_CALL: {
                const uint32_t argc = in.b >> 12, aux = in.b & 0xFFFu;
                const std::string recv = slot(d - 1 - (int)argc);
                if (in.a == FN_MATCHES && argc == 1 && aux != 0xFFFu && !(aux & 0x800u)) {
                    // the pattern's table reads bytes or scalar values (dfa.cpp): known now, so the walk is compiled for exactly that
                    const RegexDesc &rd = reinterpret_cast<const RegexDesc *>(blob + h.regexes)[aux];
                    s += "    " + recv + " = op_matches<" + (rd.umap ? "true" : "false") + ">(m, " + u32(aux) + ", " + recv + ", " + slot(d - 1) + ");\n";
                } else {
                    s += "    " + recv + " = op_call(m, " + u32(in.a) + ", " + u32(argc) + ", " + u32(aux) + ", " + recv + ", " + (argc ? slot(d - 1) : recv) + ");\n";
                }
                d -= (int)argc;
                break;
            }
            case R_BIN: s += "    " + below + " = op_bin(m, " + u32(in.a) + ", " + below + ", " + top + ");\n"; d--; break;
            default: why = "residual program: unknown instruction"; return false;
        }
    }
    s += "}\n";
    out += s;
    return true;
}

}  // namespace

// The rule functions rvm_rule_<k>(Machine &) and the dispatcher rvm_rule_dispatch(Machine &, k): portable C++ over residual.h
// (PWAF_RVM_RULE = the functions' attributes: device code on the device, plain inline functions in the test-only host build).
bool rvm_specialize(const uint8_t *blob, size_t len, std::string &out, std::string &why) {
    if (len < sizeof(Header)) { why = "residual program image too short"; return false; }
    Header h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != 0x314D5652u || h.total_bytes > len) { why = "not a residual program image"; return false; }
    std::string s = "namespace pwaf {\nnamespace rvm {\n";
    for (uint32_t k = 0; k < h.n_rules; k++)
        if (!translate_rule(h, blob, k, s, why)) return false;
    s += "PWAF_RVM_RULE uint32_t rvm_rule_dispatch(const Machine &m, uint32_t k) {\n    switch (k) {\n";
    for (uint32_t k = 0; k < h.n_rules; k++) s += "        case " + std::to_string(k) + ": return rvm_rule_" + std::to_string(k) + "(m);\n";
    s += "        default: return 0u;\n    }\n}\n}  // namespace rvm\n}  // namespace pwaf\n";
    out += s;
    return true;
}

// The whole device program: residual.h, the rule functions, and the kernel that runs every rule for every request of a batch —
// one lane per request, results as one match bit per (request, rule): match_words[w * n + r] bit k = rule 32 w + k, which the verdict
// kernel reads in place of hit records (VerdictArgs::res_match); execution errors are counted per rule as they happen.
bool rvm_jit_program(const uint8_t *blob, size_t len, std::string &out, std::string &why) {
    Header h;
    if (len < sizeof h) { why = "residual program image too short"; return false; }
    memcpy(&h, blob, sizeof h);
    std::string s;
    s += "#define PWAF_RVM_RULE static __device__\n";
    s += "#define PWAF_RVM_HEAP " + std::to_string(std::max(1u, std::min(h.heap_items, 64u))) + "\n";
    s += kResidualHeaderText;
    s += "\n";
    if (!rvm_specialize(blob, len, s, why)) return false;
    s += R"KRN(
// (mirrors kernels.h: ResidualJitArgs)
struct RvmJitArgs {
    const uint8_t *const *data;
    const uint32_t *const *off;
    const uint8_t *blob;
    uint32_t n, n_rules;
    const uint8_t *ip;
    const uint8_t *ip_is_v6;
    const uint16_t *port;
    const uint32_t *asn;
    const uint16_t *country;
    uint32_t has_geo, pad;
    const uint32_t *geo_root4, *geo_root6, *geo_nodes;
    const uint2 *geo_recs;  // {asn, country (two bytes, memory order) | pad << 16}
    uint32_t *match_words;
    unsigned long long *rule_errors;
};
extern "C" __global__ __launch_bounds__(256) void rvm_jit_kernel(RvmJitArgs a) {
    using namespace pwaf::rvm;
    Machine m;
    m.blob = a.blob;
    m.h = reinterpret_cast<const Header *>(a.blob);
    m.q.data = a.data;
    m.q.off = a.off;
    m.heap_n = 0;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += gridDim.x * blockDim.x) {
        m.q.r = r;
        m.q.ip = a.ip + (size_t)r * 16;
        m.q.v6 = a.ip_is_v6[r];
        m.q.port = a.port[r];
        uint32_t asn = 0, country = (uint32_t)'X' | ((uint32_t)'X' << 8);  // the default record {0, "XX"} (http_listener.rs:148-156)
        if (a.asn != nullptr) {
            asn = a.asn[r];
            const uint32_t cc = a.country[r], c0 = (cc & 0xFFu) - 'A', c1 = (cc >> 8) - 'A';
            if (c0 < 26u && c1 < 26u) country = cc;
        } else if (a.has_geo) {
            // GeoipDB::lookup (pingoo/geoip.rs:73-91): loopback / multicast are "not found"
            const uint8_t *ip = m.q.ip;
            bool walk;
            if (!m.q.v6) walk = !(ip[0] == 127u || (ip[0] & 0xF0u) == 0xE0u);
            else {
                bool loopback = ip[15] == 1;
                for (int k = 0; k < 15; k++) loopback = loopback && ip[k] == 0;
                walk = !(loopback || ip[0] == 0xFFu);
            }
            if (walk) {
                uint32_t e = (m.q.v6 ? a.geo_root6 : a.geo_root4)[((uint32_t)ip[0] << 8) | ip[1]];
                for (uint32_t k = 2; !(e & 0x80000000u); k++) e = a.geo_nodes[(size_t)e * 256 + ip[k]];
                const uint2 g = a.geo_recs[e & 0x7FFFFFFFu];
                asn = g.x;
                country = g.y & 0xFFFFu;
            }
        }
        m.q.asn = asn;
        m.q.country = country;
)KRN";
    // Per 32 rules one result word per request (bit k = rule 32 w + k matched: the verdict kernel turns the words of a 64-request group
    // into its column bitmasks); execution errors (the reference logs each, pingoo/rules.rs:41-45) are counted per rule: one atomic per
    // wave and rule that saw any.
    const uint32_t words = (h.n_rules + 31) / 32;
    for (uint32_t w = 0; w < words; w++) {
        s += "        {\n            uint32_t mt = 0, er = 0, res;\n";
        for (uint32_t k = 32 * w; k < std::min(h.n_rules, 32 * w + 32); k++) {
            const std::string bit = std::to_string(k & 31u);
            s += "            res = rvm_rule_" + std::to_string(k) + "(m); mt |= (uint32_t)(res == 1u) << " + bit + "; er |= (uint32_t)(res == 2u) << " + bit + ";\n";
        }
        s += "            a.match_words[(size_t)" + std::to_string(w) + " * a.n + r] = mt;\n";
        s += "            if (__builtin_amdgcn_ballot_w64(er != 0u) != 0ull) {\n"
             "                for (uint32_t k = 0; k < 32u; k++) {\n"
             "                    const unsigned long long em = __builtin_amdgcn_ballot_w64(((er >> k) & 1u) != 0u);\n"
             "                    if (em != 0ull && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(em))\n"
             "                        __hip_atomic_fetch_add(&a.rule_errors[" + std::to_string(32 * w) + "u + k], (unsigned long long)__builtin_popcountll(em), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n"
             "                }\n            }\n        }\n";
    }
    s += "    }\n}\n";
    out = std::move(s);
    return true;
}

}  // namespace pwaf
