// program.h — the compiled form of a rule set: what the host compiler produces and the device consumes.
//
// Pipeline:  expression text --front-end--> typed boolean DAG over ATOMS
//   --DNF--> per-rule literal lists;  string atoms --regex/literal patterns--> per-field DFA groups
//   (LDS-resident transition tables);  ip-list atoms --> one multibit radix trie (membership-set ids);
//   GeoIP prefixes --> one multibit radix trie (record ids).
#pragma once
#include <array>
#include <functional>
#include <bitset>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pwaf.h"

namespace pwaf {

// ---- regex / literal patterns ------------------------------------------------------------------
using ByteSet = std::bitset<256>;

// A_WORD_B / A_NOT_WORD_B: the crate's Unicode-aware \b \B (the default); the _ASCII forms: under (?-u)
enum AssertKind : uint8_t { A_TEXT_START, A_TEXT_END, A_LINE_START, A_LINE_END, A_WORD_B, A_NOT_WORD_B, A_WORD_B_ASCII, A_NOT_WORD_B_ASCII };
using CpRange = std::pair<uint32_t, uint32_t>;
using CpSet = std::vector<CpRange>;  // scalar values: sorted, disjoint, non-adjacent ranges

struct RNode;
using RNodeP = std::shared_ptr<RNode>;
struct RNode {
    // CLASS: one BYTE of a set (literal bytes of contains / == / ..., and every regex class within ASCII). UCLASS: one SCALAR VALUE of
    // a set with members beyond ASCII — `.`, negated classes, \w \d \s \p{..}, (?i)s (= s, S, U+017F) — as the regex crate matches
    // a Rust str (regex 1.12.2 is Unicode-aware by default, Cargo.lock:1694-1700; url / path may carry UTF-8, http 1.3.1): `ucls` holds
    // the whole set (ASCII members too), `cls` its ASCII members; dfa.cpp lowers it to UTF-8 byte sequences.
    enum K : uint8_t { EMPTY, CLASS, CAT, ALT, REPEAT, ASSERT, UCLASS } k = EMPTY;
    ByteSet cls;
    CpSet ucls;
    std::vector<RNodeP> kids;
    int rmin = 0, rmax = -1;  // REPEAT; rmax < 0 = unbounded
    AssertKind ak = A_TEXT_START;
};
RNodeP rx_empty();
RNodeP rx_class(const ByteSet &s);
void cp_canon(CpSet &s);  // sorts, merges, drops the surrogates
CpSet cp_complement(const CpSet &s);
CpSet cp_intersect(const CpSet &a, const CpSet &b);
RNodeP rx_scalars(const CpSet &s);  // CLASS when the set lies within ASCII, else UCLASS
void utf8_sequences(const CpSet &s, std::vector<std::vector<std::pair<uint8_t, uint8_t>>> &out);  // the non-ASCII part of `s` as sequences of byte ranges
const CpSet &unicode_word_set(bool unicode);  // \w: Alphabetic + M + Nd + Pc + Join_Control (regex-syntax), or [0-9A-Za-z_]
RNodeP rx_byte(uint8_t c);
RNodeP rx_literal(const std::string &bytes);
RNodeP rx_cat(std::vector<RNodeP> kids);
RNodeP rx_alt(std::vector<RNodeP> kids);
RNodeP rx_assert(AssertKind k);
std::string rx_key(const RNode &n);  // canonical text (atom de-duplication)

// Parses the supported Rust-regex subset. status: 0 ok, 1 invalid pattern
// (a run-time error in the reference => the rule can never match), 2 valid-but-unsupported.
RNodeP regex_parse(const std::string &pattern, int &status, std::string &err);

// ---- atoms ---------------------------------------------------------------------------------------
enum AtomKind : uint8_t {
    ATOM_TRUE = 0,   // column of all ones
    ATOM_SCAN,       // string field matched by a pattern (DFA)           -> scan kernels
    ATOM_LEN,        // byte length of a string field  <op> constant       -> verdict kernel
    ATOM_INT,        // client.remote_port / client.asn <op> constant
    ATOM_INTSET,     // client.remote_port / client.asn in sorted set
    ATOM_IPSET,      // client.ip contained in CIDR list L
    ATOM_COUNTRY,    // client.country in 676-bit table
    ATOM_FCMP,       // one request string field against ANOTHER (== / contains / starts_with / ends_with, or their lengths)
    ATOM_RESIDUAL,   // a whole rule evaluated per request by the residual interpreter (residual.h): ref = its index among the residual rules
};
// ATOM_FCMP operators (Atom::c): field `field` against field `ref`
enum FcmpOp : uint8_t { FC_EQ = 0, FC_CONTAINS, FC_STARTS, FC_ENDS, FC_LEN_EQ, FC_LEN_LT, FC_LEN_LE };
struct FcmpAtom {
    uint32_t col;  // device column
    uint8_t op, a, b, pad;
};
static constexpr uint32_t kMaxFcmpAtoms = 32, kMaxFcmpFields = 8;
enum CmpOp : uint8_t { OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE };
enum IntVar : uint8_t { VAR_PORT = 0, VAR_ASN = 1 };

struct Atom {
    AtomKind kind = ATOM_TRUE;
    uint8_t field = 0;  // SCAN/LEN: PWAF_FIELD_*; INT/INTSET: IntVar
    CmpOp op = OP_EQ;
    int64_t c = 0;             // LEN / INT constant
    RNodeP pattern;            // SCAN
    uint32_t ref = 0;          // INTSET: index into int_sets; IPSET: ip list index; COUNTRY: lut index
    uint32_t id = 0;           // device column id (assigned at layout time)
    uint32_t min_len = 0;      // SCAN: length of the shortest string the pattern matches (a proxy for how rare a hit is)
    bool neg_used = false;     // the atom occurs negated in some rule term (a hint that most requests satisfy it)
    bool gates = false;        // SCAN: a prefilter factor of some gap pass (DfaGroup::filter_atoms): a request that satisfies it must reach
                               // that pass's request list, which only the DFA walk of the owning pass does — never a filter HEAD
    std::string key;           // canonical form for de-duplication
};

// ---- bigram prefilter (filter.cpp) ------------------------------------------------
// A pass whose patterns ALL have a necessary literal factor does not walk its DFA over every request. The filter kernel
// streams the field once with a bucketed shift-or over hashed BIGRAMS (case-folded byte pairs): one independent 4-byte LDS
// lookup per input byte, no state-dependent address. 8 buckets x 4 bigram positions = one 32-bit state per lane;
// state' = (state << 8) | table[bin(b[i], b[i+1])], and a zero bit in the top byte says "the last <= 4 bigrams are
// consistent with some factor window of bucket b". Requests with such a position are CANDIDATES; only they are walked
// through the pass's DFA, which decides exactly. The filter is conservative by construction (a factor is necessary for
// a match; extra positions past a field's end can only add candidates), so verdicts never depend on it.
static constexpr uint32_t kFilterBits = 12, kFilterEntries = 1u << kFilterBits;
// Bigrams are sampled at every stride-th byte of the arena stream; a factor is entered once per alignment it can have relative to
// the sampling grid. Stride 2 halves the lookups per input byte (the filter kernel is bound by LDS gathers: ~8 LDS cycles per wave
// lookup, three quarters of them bank conflicts) at the price of windows that span up to 8 bytes: factors shorter than 3 bytes
// cannot be filtered, and a 3-4 byte factor has a single sampled bigram of its own per alignment.
// GroupFilter::stride: 1, or 2 for a pass whose factors stay selective with half the bigrams sampled — half the table lookups per
// byte. (Stride 2 for EVERY pass was tried first: a 3-byte factor such as "../" then owns a single sampled position and floods the
// candidates — 34 % of the URL sample. Round 5: such a window is EXTENDED by the sampled bigrams that straddle the factor's start and
// end — (any byte, first byte), (last byte, any byte) — and short windows get buckets of their own: 2.8 %, filter.cpp
// Model::best_window.) pwaf_engine_tune decides per pass from the sample, with and without extended windows.
static constexpr uint32_t kFilterMul = 0x9E37u;  // default 16-bit multiplier of the bigram hash (v_pk_mul_lo_u16 on the device); a pass picks its own from a
                                                 // few candidates so that its factor windows avoid the bins frequent bigrams fall into (GroupFilter::mul)
#if defined(__HIPCC__)
#define PWAF_HOST_DEVICE __host__ __device__
#else
#define PWAF_HOST_DEVICE
#endif
// The filter's case folding: bit 5 cleared in the bytes that have bit 6 set (0x40 - 0x7F, 0xC0 - 0xFF) — letters lose their case, and
// nothing below 0x40 moves. Clearing bit 5 everywhere (rounds 1 - 4) also folded '+' onto \v, '-' onto \r, ',' onto \f, ')' onto \t and
// '*' onto \n: a `\s` position of a factor then let `union+select` and `select+` through the filter — every request of the hostile
// stream that spells its blanks as '+' was a candidate of the url pass (88 % of its completed windows: tools/hostile_flags.py).
PWAF_HOST_DEVICE static inline uint32_t filter_fold(uint32_t b) { return b & ~((b >> 1) & 0x20u); }             // one byte
PWAF_HOST_DEVICE static inline uint32_t filter_fold4(uint32_t x) { return x & ~((x >> 1) & 0x20202020u); }       // four packed bytes (bit 6 of a byte lands on ITS bit 5: nothing crosses a byte)
PWAF_HOST_DEVICE static inline uint32_t filter_bin(uint8_t b0, uint8_t b1, uint32_t mul = kFilterMul) {
    const uint32_t p = filter_fold(b0) | (filter_fold(b1) << 8);  // ASCII case folding
    // top 12 bits of the 16-bit product: they mix all 8 bits of the second byte (bits [2, 14) keep only 6 of them, and digits then
    // alias letters: measured 2.6x the candidates on URLs)
    return ((p * mul) & 0xFFFFu) >> (16 - kFilterBits);
}
// An anchored literal (starts_with / ==, <= 16 bytes) that most requests satisfy (e.g. a browser User-Agent prefix under a
// negation) cannot go through the filter — every request would be a candidate. Up to two such HEADS per pass are compared
// directly against the first 16 bytes of the field by the filter kernel and land in the request's hit record.
struct FilterHead {
    uint8_t bytes[16];
    uint8_t len;
    uint8_t exact;   // 1: field == literal, 0: field starts with literal
    uint16_t local;  // local atom id in the pass
};

// ---- the CONFIRM tier of a filtered pass (round 4; filter.cpp builds it, confirm.h evaluates it on the host and on the device) ----
// The bigram filter flags a 16-byte chunk when some position of it completes a WINDOW (<= 4 sampled bigrams) of some factor; a window
// covers five to eight bytes of a factor that may be sixty long, so traffic made of near misses of the rule literals — what an
// attacker sends — passes the filter and used to be walked through the pass's DFA, deep into states no LDS copy holds. The confirm
// tier looks at the flagged position itself: the bigram there selects (by its filter bin) the few factors whose window can END
// there, and each is compared in full, byte for byte, at the place the window implies. What that decides:
//   * an atom that IS a literal (contains / starts_with / ends_with / == of a constant: `confirm_literal`) is decided exactly —
//     it never needs a DFA — and lands in the request's hit record;
//   * any other atom (a regex) is known NOT to match unless one of its necessary factors was confirmed; only then is the request
//     walked, through a DFA of the pass's non-literal atoms alone (DfaGroup::rtier), which is a fraction of the full table.
// Entry e of bin b (head[b] = first | count << 20): the factor `bytes` (len value bytes, len mask bytes — ((text ^ value) & mask) == 0
// per position, a zero mask where the position is a byte CLASS — then n_cls x {position, class id}) begins d bytes before the
// flagged position; atom = the local atom it decides (0xFFFF: an R entry — "a regex factor is here: walk").
struct ConfirmEntry {
    uint32_t bytes_off;  // into ConfirmTable::bytes (4-byte aligned)
    uint16_t len, d;
    uint16_t atom;       // local atom id, or kConfirmWalk
    uint8_t flags;       // kConfirmAtStart: the factor must begin at the field's first byte; kConfirmAtEnd: it must end at its last
    uint8_t n_cls;
};
static_assert(sizeof(ConfirmEntry) == 12, "ConfirmEntry layout");
static constexpr uint16_t kConfirmWalk = 0xFFFFu;
static constexpr uint8_t kConfirmAtStart = 1, kConfirmAtEnd = 2;
struct ConfirmTable {
    bool enabled = false;
    bool has_walk = false;               // some entry is an R entry: the pass keeps a DFA (of its R atoms)
    std::vector<uint32_t> head;          // kFilterEntries
    std::vector<ConfirmEntry> entries;   // grouped by bin
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> classes;       // 8 words (256 bits) per class id
};

struct GroupFilter {
    bool enabled = false;
    std::vector<uint32_t> table;  // kFilterEntries masks: bit 8*j + b = 0 <=> bucket b accepts the bin at window position j
    uint32_t init = 0xFFFFFFFFu;  // state at the start of a field (zero at a bucket's wildcard positions)
    uint32_t mul = kFilterMul;    // multiplier of the bigram hash chosen for this pass
    uint32_t stride = 1;          // bigrams are sampled at every stride-th byte of the arena stream (1 or 2); factors are entered once per alignment
    std::vector<FilterHead> heads;
    double est_candidate_rate = 0;  // expected fraction of requests flagged by chance (model or sample)
    std::string note;               // why the pass is not filtered, for stats / warnings
    ConfirmTable confirm;           // built with the windows it mirrors (every rebuild of the filter rebuilds it)
};
struct FilterHints {  // from a traffic sample (pwaf_engine_tune); all optional
    const double *pair_prob = nullptr;             // 65536 entries indexed by fold(b0) | fold(b1) << 8: probability of the (case-folded) bigram in traffic
    const std::vector<uint64_t> *atom_hits = nullptr;  // per local atom: sample requests it holds for
    uint64_t n_requests = 0;
    double mean_len = 0;                           // mean field length in the sample (0 = unknown)
};

// ---- DFA groups ----------------------------------------------------------------------------------
// One multi-pattern DFA = one scan pass over one field. States are numbered in BFS order from the start
// state (id 0), so low ids are the shallow, frequently visited states: the device keeps rows [0, n_hot)
// in LDS and reads colder rows from the L2-resident global copy.
// Scalar mode of a table (dfa.cpp): the class of a scalar value beyond ASCII, looked up at its lead byte — stage1[cp >> 7] names a
// block of 128 classes in stage2 (equal blocks shared: a few KiB even when \\w is in the table). Empty: the table reads bytes.
static constexpr uint32_t kScalarBlocks = 0x110000 >> 7;
struct ScalarMap {
    std::vector<uint16_t> stage1;
    std::vector<uint8_t> stage2;
    uint8_t ill_class = 0;  // a byte that begins no well-formed sequence
    bool on() const { return !stage1.empty(); }
};
struct DfaGroup {
    uint8_t field = 0;
    uint32_t n_states = 0, n_classes = 0;
    uint8_t classmap[256] = {0};          // class of a BYTE (scalar mode: 0x80..0xBF continuation bytes — a class that stays — and 0xC0..0xFF lead bytes, which the walkers replace by umap's class of the scalar)
    ScalarMap umap;
    std::vector<uint8_t> class_stays;    // per class: every transition stays where it is and emits nothing (continuation / lead bytes of scalar mode)
    std::vector<uint16_t> trans;         // n_states * n_classes, row-major; state 0 = start
    std::vector<uint32_t> emit_off;      // n_states + 1 offsets into emit_list (columns set on ENTERING the state)
    std::vector<uint16_t> emit_list;     // LOCAL atom ids (column = atom_base + local)
    std::vector<uint32_t> end_off;       // n_states + 1
    std::vector<uint16_t> end_list;      // LOCAL atom ids true if the field ends in this state
    uint32_t atom_base = 0;              // first device column of this group
    uint32_t n_local = 0;                // columns owned by this group (= atoms.size())
    std::vector<uint32_t> atoms;         // indices into Program::atoms, local id order
    // Prefilter gating: when non-empty, a request's field can only match a pattern of this group if at
    // least one of these columns (owned by EARLIER, ungated groups of the same field) is set, so the pass only needs to
    // visit those requests.
    std::vector<uint32_t> filter_atoms;  // indices into Program::atoms
    std::vector<uint32_t> filter_cols;   // their device columns (filled at layout time)
    // Bigram prefilter of an otherwise ungated pass (enabled = the pass only walks the filter's candidates)
    GroupFilter filter;
    // The DFA of the pass's NON-literal atoms alone (local atom ids are the full group's): what a confirmed candidate is walked
    // through. Null when every atom is a confirm literal (no walk at all) or none is (the full DFA is the R DFA).
    std::shared_ptr<DfaGroup> rtier;
    uint32_t n_confirm_literals = 0;  // atoms of the pass the confirm tier decides without a DFA
    bool confirm_off = false;         // PWAF_OPT_NO_CONFIRM: build_group_filter leaves the confirm table out
};
static constexpr uint32_t kMaxDfaStates = 32767;   // 15-bit state ids: bit 15 of a table entry flags "target state emits"
static constexpr uint32_t kMaxLocalAtoms = 32766;  // 15-bit (+1) atom slots in a hit record

// ---- radix tries ----------------------------------------------------------------------------------
// Entry: bit31 set => leaf, low 31 bits = value; else child node index (node n occupies nodes[n*256 ..]).
struct IpTrie {
    std::vector<uint32_t> root4, root6;  // 65536 entries each (empty vector when the family has no prefixes)
    std::vector<uint32_t> nodes;         // 256-entry nodes, shared by both families
    uint32_t n_nodes() const { return (uint32_t)(nodes.size() / 256); }
};
static constexpr uint32_t TRIE_LEAF = 0x80000000u;

struct GeoRec {
    uint32_t asn;
    uint16_t country;  // two bytes in memory order
    uint16_t pad;
};

// ---- per-rule device records -----------------------------------------------------------------------
struct DevRule {
    uint32_t lit_off, lit_cnt;  // into Program::lits
    uint32_t public_idx;        // index in the caller's rule array, or PWAF_RULE_* pseudo index
    uint8_t eff_unverified;     // PWAF_ACTION_* applied when this rule matches and the client is not captcha-verified
    uint8_t eff_verified;       //   ... and when it is (ALLOW here means "no effect: keep going")
    uint8_t pad[2];
};
// literal encoding in Program::lits
static constexpr uint32_t LIT_NEG = 1u << 30;       // negated atom
static constexpr uint32_t LIT_TERM_END = 1u << 31;  // last literal of its conjunction
static constexpr uint32_t LIT_ATOM_MASK = (1u << 24) - 1;
// DEVICE copy of the literals only (engine.cpp; never in Program::lits): a LAZY comparison atom — `length / port op constant` that is no rule's
// trigger — is not evaluated per group by the attribute kernel; its literal carries an index into VerdictArgs::lazy instead of a column and the
// verdict kernel evaluates it for the few rules whose other literals already hold for somebody (kernels.hip: verdict2_kernel)
static constexpr uint32_t LIT_LAZY = 1u << 29;  // then bits [15:0] = the constant (<= 65534... any 16-bit value), bit 16 = operator (0: ==, 1: <=), bit 17 = slot of the variable (VerdictArgs::lazy_var), bit 18 = the atom is evaluated complemented (!=, >)

struct NumAtomDev {  // numeric atom descriptor consumed by the verdict kernel
    uint32_t col;    // device column
    uint8_t kind, var, op, pad;
    uint32_t ref, ref2;  // INTSET: [begin,end) into int_pool; IPSET: list bit; COUNTRY: lut index
    int64_t c;
};

static constexpr uint32_t kMaxHeaders = 120;   // header columns (field ids 5 .. 5 + kMaxHeaders - 1)
static constexpr uint32_t kMaxGroups = 250;    // scan passes per program

struct Program {
    // source-level
    std::vector<Atom> atoms;                     // atoms[0] = TRUE
    // EXTENSION: the names the rule set uses as http_request.headers["name"], in first-use order; field id of
    // column k = PWAF_N_FIELDS + k. The host supplies one string column per name (absent header = empty string).
    std::vector<std::string> header_names;
    std::vector<std::vector<int64_t>> int_sets;  // sorted, unique
    std::vector<std::bitset<704>> country_luts;  // index = (c0-'A')*26 + (c1-'A')
    std::vector<std::string> warnings;
    // per caller rule: PWAF_OK, or why the device compiler could not take it (PWAF_E_UNSUPPORTED + text). Such a rule never matches
    // (and says so here and in the warnings) instead of failing the whole rule set, unless PWAF_OPT_STRICT asks for the failure.
    std::vector<std::pair<int, std::string>> rule_status;

    // device-level
    uint32_t n_cols = 0;          // total columns: [0] TRUE, numeric atoms, then each DFA group's atoms
    uint32_t n_scan_cols = 0;     // columns owned by DFA groups
    std::vector<DfaGroup> groups;
    std::vector<NumAtomDev> num_atoms;
    std::vector<FcmpAtom> fcmp;   // field-against-field atoms: one more (pseudo) pass, columns [fcmp_base, fcmp_base + fcmp.size())
    uint32_t fcmp_base = 0;
    // residual rules (residual.h): one more pseudo pass after the field-against-field one, one column per rule, evaluated by residual_kernel
    std::vector<uint8_t> residual_blob;  // rvm::Header + sections (empty: none)
    uint32_t n_residual = 0, residual_base = 0;
    bool residual_needs_geo = false;     // some residual rule reads client.asn / client.country
    std::vector<uint32_t> residual_rule; // per residual rule: the caller's rule index (execution-error counters are reported per caller rule)
    std::vector<int64_t> int_pool;
    std::vector<uint32_t> country_lut_words;  // 22 words per lut
    std::vector<DevRule> rules;               // pseudo rules first, then the caller's rules with an effect
    std::vector<uint32_t> lits;
    uint32_t n_user_rules = 0;

    // ip lists -> membership sets
    uint32_t n_ip_lists = 0;
    uint32_t set_words = 0;                // 32-bit words per membership set
    std::vector<uint32_t> set_masks;       // n_sets * set_words ; set 0 = empty
    IpTrie ipset_trie;
    bool has_geo = false;
    IpTrie geo_trie;
    std::vector<GeoRec> geo_recs;          // rec 0 = default {0,"XX"}

    uint32_t flags = 0;
    uint32_t lds_hot_budget = 64 * 1024;  // LDS bytes the scan kernel may use for the hot rows of one table
    pwaf_stats stats{};
};

struct CompileInput {
    const pwaf_rule_desc *rules;
    size_t n_rules;
    const pwaf_list_desc *lists;
    size_t n_lists;
    const pwaf_geoip_table *geoip;
    pwaf_options opts;
};

// Returns PWAF_OK or an error code; fills err (rule index + message) on failure.
int compile_program(const CompileInput &in, std::unique_ptr<Program> &out, pwaf_compile_error &err);

// Little-endian self-describing dump used by tests to inspect compiler output.
std::vector<uint8_t> dump_program(const Program &p);

// ---- pieces (exposed for unit tests through the dump) -------------------------------------------------
struct ScanPattern {
    RNodeP rx;
    uint32_t atom;  // index into Program::atoms
};
// Builds one DFA for `pats` (local ids = positions in pats). Returns false when the state limit is exceeded.
// class of the symbol that begins at bytes[i] (csrc/utf8.h: scalar mode decodes at a lead byte)
std::vector<uint8_t> scalar_map_image(const ScalarMap &m);  // [stage1 u16 x kScalarBlocks][stage2]: what csrc/utf8.h reads
uint32_t dfa_class_at(const DfaGroup &g, const uint8_t *bytes, size_t i, size_t n);
bool build_dfa(const std::vector<ScanPattern> &pats, uint32_t max_states, uint32_t max_table_bytes, DfaGroup &out, std::string &err);
// true when the pattern has an unbounded repetition of a wide byte class (".*", "[^x]*", ...): such patterns multiply DFA
// states with each other (each adds an independent "prefix seen" bit), so the grouping heuristic isolates them.
bool has_wide_gap(const RNode &n);
// For a pattern that is a top-level concatenation X · gap · REST (gap = unbounded repetition of a wide class): returns X
// without trailing zero-width assertions — every match of the pattern contains a match of X — or null when no useful
// (non-nullable) prefix exists.
RNodeP gap_prefilter(const RNodeP &rx);
uint32_t rx_min_len(const RNode &n);
// Runs a DFA on the host over `bytes`, returning local atom ids that hold. COMPILE-TIME USE ONLY
// (folding predicates over the 676 possible country codes into a lookup table).
void dfa_run_host(const DfaGroup &g, const uint8_t *bytes, size_t n, std::vector<uint16_t> &out_atoms);

// Builds the bigram prefilter of pass `g` (filter.cpp). `atoms` = Program::atoms. Leaves filter.enabled false (with a note)
// when some pattern has no usable literal factor.
// extend (stride 2 only): windows with fewer than four sampled bigrams may reach one bigram beyond their factor on either side
// (filter.cpp: Model::best_window). Their selectivity depends on the text AROUND factors, which only a traffic sample can tell:
// creation decides without them, pwaf_engine_tune builds both forms and keeps the one that flags less of the sample.
void build_group_filter(const std::vector<Atom> &atoms, const DfaGroup &g, const FilterHints *hints, GroupFilter &out, uint32_t stride = 1, bool extend = false);
// An atom the confirm tier decides by itself: (\A)? literal (\z)? with a literal of 2 .. 64 single bytes.
bool confirm_literal(const RNode &n, std::string &lit, bool &at_start, bool &at_end);
// Host model of the filter kernel over one field value: true = candidate. (Used by tune to measure the candidate rate on the
// sample; the device may flag MORE requests — it also looks at the bytes just past a field's end — never fewer.)
// `\A literal` / `\A literal \z` with a literal of at most 8 single bytes (kernels.h: ShortAtom)
bool short_literal_atom(const RNode &n, std::string &lit, bool &exact);
bool filter_candidate_host(const GroupFilter &f, const uint8_t *bytes, size_t n, size_t phase = 0);  // phase: offset of the first sampled byte (< stride)
bool filter_candidate_arena(const GroupFilter &f, const uint8_t *arena, uint32_t fs, uint32_t fe, uint64_t readable);  // the same question for the field [fs, fe) of an arena, its neighbours' bytes included (what the device's flat stream sees); bytes at or beyond `readable` read as zero
// Host model of filter + confirm tier over one field [fs, fe) of an arena with PWAF_ARENA_PAD readable bytes behind its end: the literal
// atoms confirmed (appended to lits), returns "walk the request through the DFA" (filter.cpp; tune and the CPU test hook use it).
bool confirm_field_host(const GroupFilter &f, const uint8_t *arena, uint32_t fs, uint32_t fe, std::vector<uint16_t> &lits, bool *flagged = nullptr);
// Which heads hold for the field value: bit k = heads[k].
uint32_t filter_heads_host(const GroupFilter &f, const uint8_t *bytes, size_t n);

struct PrefixEntry {
    uint8_t addr[16];
    uint8_t len;
    bool v6;
    uint32_t payload;  // list index (sets) or record id (LPM)
};
// mode 0: membership sets (payload = list bit, result value = set id into set_masks)
// mode 1: longest-prefix match (result value = payload of the most specific prefix; later duplicates win)
void build_ip_trie(const std::vector<PrefixEntry> &prefixes, int mode, uint32_t n_lists, IpTrie &trie, std::vector<uint32_t> &set_masks, uint32_t &set_words);

bool parse_ipnet_text(const std::string &s, PrefixEntry &out, std::string &err);
bool parse_i64_text(const std::string &s, int64_t &out);

// ---- residual rules (residual.h / residual.cpp) ---------------------------------------------------------------------------------
// A rule the column compiler cannot take is lowered WHOLE to a stack program evaluated per request by residual_kernel.
struct Syntax;
struct ResidualList {  // a configured list as the residual compiler sees it (parsed items, lists.rs:90-108)
    std::string name;
    uint32_t type = 0;
    std::vector<std::string> strs;
    std::vector<int64_t> ints;
    std::vector<PrefixEntry> nets;
    size_t size() const { return type == PWAF_LIST_STRING ? strs.size() : type == PWAF_LIST_INT ? ints.size() : nets.size(); }
};
// EXTENSION: the names of the headers map = what the rule set asks of it with a LITERAL key — http_request.headers["x"],
// http_request.headers.x, "x" in http_request.headers, http_request.headers.contains("x") (and the same behind http_request["headers"]) —
// appended to `names` in order of first use, pre-order, left to right (the oracle's rule: oracle/oracle_engine.cpp, pwaf_oracle_create).
void collect_header_names(const Syntax &syn, std::vector<std::string> &names);

class ResidualBuilder {
public:
    ResidualBuilder();
    ~ResidualBuilder();
    ResidualBuilder(const ResidualBuilder &) = delete;
    ResidualBuilder &operator=(const ResidualBuilder &) = delete;
    // Lowers one rule; returns its index among the residual rules, or -1 with the reason (nothing of the rule is kept then).
    // header_field(name) = string column of a header name (registers the name with the program on first use).
    // closed_headers (may be null): EVERY header name the whole rule set mentions with a literal key, in the order their columns have
    // (collect_header_names over all rules, before any is compiled) — what the headers map IS as a value: needed by a computed key into
    // http_request / http_request.headers and by length() of the headers map; without it such a rule is refused.
    int compile_rule(const Syntax &syn, const std::vector<ResidualList> &lists, const std::function<int(const std::string &)> &header_field, std::string &why,
                     const std::vector<std::string> *closed_headers = nullptr);
    size_t n_rules() const;
    bool needs_geo() const;  // some rule reads client.asn / client.country
    std::vector<uint8_t> blob() const;  // the device image (rvm::Header + sections)
private:
    struct Impl;
    Impl *impl;
};

}  // namespace pwaf
