// kernels.h — argument blocks and launch entry points of the gfx950 kernels (kernels.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "program.h"

namespace pwaf {

static constexpr int kMaxPasses = 250;  // (= program.h kMaxGroups)
static constexpr int kVerdictPre = 12;  // hit records per request the verdict kernel requests one group ahead
static constexpr uint32_t kGapLists = 32;     // gated gap passes own list slots [0, 32) (one bit each in the factor masks); filtered passes follow
static constexpr uint32_t kMaxHeaderLens = 8; // header columns whose LENGTH rules compare
static constexpr uint32_t kDirChunkWords = 32; // one 128-byte line per /16 of the compressed DIR-24 table (VerdictArgs::dir_chunks)

// Hit record of one (scan pass, request): what the request's field matched in that pass's DFA.
//   bit 31 = 0: bits [14:0] = first local atom + 1 (0 = none), bits [29:15] = second local atom + 1 (0 = none)
//   bit 31 = 1: bits [30:0] = head of a chain in the overflow pool holding ALL atoms of the request for this pass
static constexpr uint32_t REC_OVERFLOW = 0x80000000u;
struct PoolEntry {
    uint32_t atom;  // local atom id
    uint32_t next;  // 0xFFFFFFFF = end of chain
};

// Device table of one DFA group. n_states rows of `stride` = n_classes + 3 uint16 cells, row r starting at
// cell index r * stride. Row 0 is the start state; rows [0, n_hot) are copied into LDS ("hot": the most visited states when
// the engine was tuned on a traffic sample, else the shallowest), followed by one SENTINEL row of 0xFFFF cells. Among the hot
// rows the ones that emit matches come last: rows [n_plain, n_hot).
//   [0, n_classes)   transition cell        [n_classes]      STAY cell (= own row; used for lanes past their last byte)
//   [n_classes + 1]  END cell, [n_classes + 2]  EMIT cell: 0x8000 | atom for a single match, or 1 + list id (0 = none)
// A transition cell c <  emit_base    -> cell index of the next row: hot, nothing to emit (next lookup address = c * 2 + class
//                                        offset: one v_lshl_add)
//                   c <  special_base -> same, and entering that row emits what its EMIT cell says
//                   c >= special_base -> index c - special_base into `special`: the target row is cold.
// emit_base = n_plain * stride, special_base = (n_hot + 1) * stride (just past the sentinel row). A lane whose current row is
// cold parks on the sentinel row, whose cells are all 0xFFFF, so the same single compare (c >= emit_base) routes it to the
// careful path that reads the real row from the L2-resident table.
struct SpecialCell {
    uint32_t next_off;  // byte offset of the target row in the full table
    uint32_t emit;      // 1 + emit-list id, 0 = none
};
struct ScanArgs {
    const uint8_t *data;      // field arena
    const uint32_t *off;      // n + 1 offsets
    uint32_t n;
    const uint16_t *tab;      // full table (all rows), padded to 16 bytes
    const uint8_t *classmap;  // 256 bytes
    const uint8_t *umap;      // scalar mode (dfa.cpp): the scalar-value -> class map a lead byte is looked up in (csrc/utf8.h); null: the table reads bytes
    uint32_t ill_class;       // ... and the class of a byte that begins no well-formed sequence
    const SpecialCell *special;
    const uint32_t *list_off; // shared by end- and emit-lists
    const uint16_t *list;     // local atom ids
    uint32_t n_states, stride, n_classes, n_hot;
    uint32_t start_emit;      // 1 + emit-list id of the start state
    uint32_t emit_base, special_base;
    uint32_t n_cus;           // CUs of the device (grid cap of a full pass)
    uint32_t chunks;          // 16-byte chunks per loop iteration: 1 or 2 (ungated passes; chosen from the tuning sample)
    uint32_t *rec;            // n hit records of this pass
    PoolEntry *pool;
    uint32_t *pool_count;     // atomic allocator
    uint32_t pool_cap;
    uint32_t *status;         // device status word: bit 0 = overflow pool exhausted
    // a pass that owns prefilter factors (else null): per local atom, the bitmask of gated passes it triggers, and their lists
    const uint32_t *colmask_local;
    uint32_t n_local;         // atoms of this pass (entries of colmask_local)
    uint32_t *gate_lists;     // [n_gated][n]
    uint32_t *gate_count;     // [n_gated], zeroed by the host per batch
};

// A list-driven pass (behind a bigram prefilter, or gated by prefilter factors): lscan_kernel walks ONE listed request per lane
// through the pass's DFA. A list holds a few percent of the batch, so what matters is latency per step: the table's first n_hot rows
// (states are numbered by how often the tuning sample's candidates visit them, start state first) are staged in LDS — a step there
// is two dependent LDS reads (class, cell) instead of an L2 round trip — and the rest is read from the L2-resident flat table.
//   flat[s * (n_classes + 3) + c] = next state | 0x8000 when entering it emits; cell n_classes of a row = what entering that state
//   emits (0, 0x8000 | the single local atom, or 1 = a list), cell n_classes + 1 = the state itself (the STAY cell: the "transition"
//   of a lane past its field's end), cell n_classes + 2 = 1 when the field ending in this state emits; emit / end lists are indexed
//   by (renumbered) state.
static constexpr uint32_t kListThreads = 512;
static constexpr uint32_t kListWalks = 1;      // listed requests a lane walks in lockstep (2 was measured: see lscan_kernel)
static constexpr uint32_t kListHotBytes = 48 * 1024;  // 3 workgroups (24 waves) per CU
struct ListScanArgs {
    const uint8_t *data;
    const uint32_t *off;
    uint32_t n;
    const uint16_t *flat;
    const uint8_t *classmap;  // 256 bytes
    const uint8_t *umap;      // scalar mode: see ScanArgs
    uint32_t ill_class;
    uint32_t n_classes;
    uint32_t n_hot;            // rows staged in LDS: n_hot * (n_classes + 3) * 2 <= the launch's ListShape::hot_bytes
    // DELTA records (engine.cpp: build_flat_group): states [n_hot, n_hot + n_delta) live in LDS as 8 bytes each — base row (16 bits), two
    // exception classes (8 bits each), their two cells (16 bits each); every other class reads the base row's cell. Null: none.
    const uint64_t *delta;
    uint32_t n_delta;
    const uint32_t *emit_off;  // [n_states + 1]
    const uint16_t *emit_list;
    const uint32_t *end_off;
    const uint16_t *end_list;
    uint32_t *rec;
    PoolEntry *pool;
    uint32_t *pool_count;
    uint32_t pool_cap;
    uint32_t *status;
    const uint32_t *colmask_local;  // a pass that owns prefilter factors (else null)
    uint32_t n_local;
    uint32_t *gate_lists;
    uint32_t *gate_count;
    uint32_t *enq_bits;        // [gap list slot][enq_words]: a request is appended to a gap pass's list ONCE (the list holds n entries) — this is also that pass's "record valid" bitmap
    uint32_t enq_words;
    uint32_t behind_filter;    // the list is a bigram prefilter's candidate list (hostile traffic makes it long: lscan_async)
    const uint32_t *req_list;  // the requests to visit and, on the device, how many (req_list null: every request, n_list ignored)
    const uint32_t *n_list;
    uint32_t *visited;         // bitmap: set bit r for every visited request (null: not needed — the list IS the pass's bitmap, or all)
    // Gap passes whose prefilter factors all belong to ONE filtered pass share that pass's candidate list instead of getting lists
    // of their own through atomics (a returned same-address atomic per enqueued request was 0.5 ms per batch): the owner writes,
    // per list entry, the mask of gap passes its hits call for (need_out); a sharing pass skips entries without its bit.
    uint32_t *need_out;        // owner: [list entry] -> gap-pass mask (null: none shares)
    uint32_t shared_bits;      // owner: the gap passes that read need_out (the other bits of a hit's mask are enqueued with atomics)
    const uint32_t *need_in;   // sharing pass: the owner's masks (null: the list is its own)
    uint32_t need_bit;
    // The walk of a pass with a confirm tier: the request's record already holds what the filter heads and the confirm tier found
    // (literal atoms are not in the R-tier DFA); the walk starts from it, and enqueues only what its own hits add.
    uint32_t merge_rec;
    uint32_t n_cus;
    // The flag-density switch (round 6; FilterArgs::dense_flag): a filtered pass whose prefilter flagged more than half of the arena's chunks —
    // an attacker can make it flag all of them — is not confirmed chunk by chunk (5.6 x slower than no filter at all, measured) but walked
    // whole, every request through the pass's FULL DFA. The decision is per pass and per batch, on the device (the flagged-chunk count filter_kernel leaves), so both
    // forms are in the launch and the plan kernel gives the untaken one no work:
    //   dense_mode 1  the dense alternative itself: every request (req_list null), the full table — runs only when *dense_flag > dense_thresh
    //   dense_mode 2  the pass's R-tier walk over the confirm tier's walk list — nothing to do when it does
    //   dense_mode 3  a gap pass that shares the pass's list through need masks: every request when it does (the dense walk writes the masks by request)
    const uint32_t *dense_flag;  // the pass's flagged-chunk count on the device (filter_kernel adds each slab's); the pass is dense when it exceeds dense_thresh
    uint32_t dense_thresh;
    uint32_t dense_mode;
};
// the list a pass walks in this batch: (req_list, entries) after the flag-density switch
struct ListOf {
    const uint32_t *req_list;
    uint32_t n_l;
};

// ---- confirm tier (program.h: ConfirmTable; confirm.h) ---------------------------------------------------------------------------
// confirm_kernel: one lane per (request, flagged chunk) pair of a filtered pass (resolve_kernel lists them). The lane runs the filter's
// automaton over its chunk, compares the entries of the windows that completed there, and only acts when something is confirmed
// (rare: most pairs are chance hits of the bigram hash, or near misses): a literal atom is merged into the request's hit record
// (compare-and-swap: another chunk of the same request may be merging too), a factor of a non-literal atom puts the request on the
// pass's walk list (once: walk_bits) for the R-tier DFA.
struct ConfirmArgs {
    const uint8_t *data;
    const uint32_t *off;
    uint32_t n;
    uint32_t total;             // bytes in the arena (readable: total + PWAF_ARENA_PAD)
    const uint2 *pairs;         // {request, arena chunk}
    const uint32_t *pair_count; // on the device
    uint32_t pair_cap;
    uint32_t mul, stride, init; // of the pass's bigram hash / sampling / automaton (GroupFilter)
    const uint32_t *ftable;     // the pass's filter table (kFilterEntries masks)
    const uint32_t *c_head;     // ConfirmTable on the device
    const ConfirmEntry *c_entries;
    const uint8_t *c_bytes;
    const uint32_t *c_classes;
    uint32_t n_entries, n_bytes, n_class_words;  // table sizes (bytes: a multiple of 4): a pass whose entries + bytes + classes fit the launch's LDS pool is compared from LDS
    uint32_t *rec;              // the pass's hit records, ZEROED by the host per batch (hits are merged in)
    uint32_t *valid_bits;       // bit r: record r holds something (the verdict kernel reads no other record of the pass)
    uint32_t *walk_bits;        // bit r: request r is on the walk list
    uint32_t *walk_list;        // requests to walk through the R-tier DFA, and how many (null: the pass has no non-literal atom)
    uint32_t *walk_count;
    PoolEntry *pool;
    uint32_t *pool_count;
    uint32_t pool_cap;
    uint32_t *status;
    const uint32_t *colmask_local;  // a pass that owns prefilter factors of gap passes (else null): see ListScanArgs
    uint32_t n_local;
    uint32_t *gate_lists;
    uint32_t *gate_count;
    uint32_t *enq_bits;
    uint32_t enq_words;
    uint32_t shared_bits;       // gap passes that share this pass's walk list (ListScanArgs::need_out): a literal hit that calls for one sends the request through the walk
    const uint32_t *dense_flag; // the pass's flagged-chunk count; above dense_thresh the pass is walked whole this batch (ListScanArgs::dense_flag): no pair of it is confirmed
    uint32_t dense_thresh;
};
static constexpr uint32_t kConfirmThreads = 1024;
static constexpr uint32_t kConfirmPoolBytes = 40 * 1024;  // LDS for a pass's entries + bytes + classes, next to the two 16 KiB tables: two 1024-thread workgroups per CU
struct ConfirmTableDev {
    const ConfirmArgs *c;  // device
    uint32_t count;
};
// `plan`: count + 1 words of device scratch (work-item prefix sums, written on the same stream)
int launch_confirm(const ConfirmArgs *host, uint32_t count, const ConfirmArgs *dev, uint32_t *plan, uint32_t n_cus, void *stream);

// ---- bigram prefilter (program.h: GroupFilter) ------------------------------------------------------------------------------
// The shift-or state only remembers the last four bigrams, so a field's arena is processed as ONE flat byte stream: a wave takes a
// slab of kStreamSlab consecutive bytes and walks it kStreamIter bytes per iteration — four rows of 1 KiB, every lane one 16-byte
// chunk of each row, the state a chunk starts from handed over by its neighbour lane (kernels.hip: filter_rows) — with no
// per-request logic at all in the loop: every load instruction is one contiguous KiB (every line is fetched exactly once; the
// per-request-lane version fetched 2x the arena because 7 MB of half-consumed lines per XCD thrashed L2), no lane ever idles,
// and a window that straddles a request boundary can only ADD a candidate.
//   filter_kernel   per 16-byte chunk a hit bit, written as the pass's dense chunk bitmap (a row's ballot is its 64-bit word: no
//                   lists, no atomics). Heads (anchored literals) are compared at request starts, which the wave finds by walking
//                   the offsets column alongside the bytes.
//   resolve_kernel  flagged chunks -> requests (rank queries over the slab's bitmap), one bit per request in `bitmap`.
//   bitcount_kernel / compact_kernel   bitmap -> dense ascending request list + its length for the confirming lscan_kernel.
static constexpr uint32_t kStreamSlab = 128 * 1024;   // bytes per wave
static constexpr uint32_t kStreamIter = 4096;          // bytes per iteration of a wave: four rows of 64 lanes x 16 bytes
static constexpr uint32_t kFilterWaves = 4;           // waves (slabs) per workgroup
static constexpr uint32_t kCompactWords = 2048;       // bitmap words per compact workgroup (65536 requests)
struct FilterArgs {
    const uint8_t *data;
    const uint32_t *off;
    uint32_t n;
    uint32_t total;           // bytes in the arena (= off[n])
    uint32_t slab0;           // first slab to stream (a slab view of a larger arena starts at off[0]: the bytes before it are not the batch's)
    uint32_t init;            // state of a stream with no history
    uint32_t mul;             // multiplier of the bigram hash (GroupFilter::mul)
    uint32_t stride;          // 1: a bigram at every byte; 2: at the even bytes of the arena stream (GroupFilter::stride)
    const uint32_t *table;    // kFilterEntries masks
    uint32_t n_heads;
    uint32_t head_w[2][4];    // head literal, little-endian dwords
    uint32_t head_m[2][4];    // byte masks of the literal's length
    uint32_t head_len[2];     // length | exact << 8
    uint32_t head_code[2];    // hit-record bits of the head's atom
    uint32_t *rec;            // n hit records of the pass, zeroed by the host: written only where a head holds
    uint32_t *chunk_bits;     // the pass's chunk bitmap: bit c = the 16-byte chunk c of the streamed slabs holds a position that completed a window (kStreamSlab / 512 words per slab)
    uint32_t *sub_count;      // [slabs]: flagged chunks of the slab
    uint32_t *bitmap;         // [(n + 31) / 32], zeroed by the host: candidate requests
    uint32_t *block_count;    // [(words + kCompactWords - 1) / kCompactWords]: candidates per compact workgroup
    uint32_t *list;           // n: dense candidate list
    uint32_t *list_count;     // its length
    // A pass with a confirm tier: resolve_kernel writes no candidate bitmap but the (request, flagged chunk) PAIRS of the pass — one per
    // flagged chunk inside a request's own bytes — for confirm_kernel to take one lane each (null: the pass has no confirm tier)
    uint2 *pairs;
    uint32_t *pair_count;
    uint32_t pair_cap;
    uint32_t *pair_base;      // [slabs]: where each slab's pairs begin in the list (filter_kernel: one atomic on pair_count per slab)
    // the flag-density switch (ListScanArgs::dense_flag): when the pass's flagged chunks (*dense_flag = its pair count) exceed dense_thresh,
    // resolve_kernel leaves the pass alone (null: the pass has no dense alternative)
    const uint32_t *dense_flag;
    uint32_t dense_thresh;
    uint32_t first_block;     // first workgroup of this pass in the fused filter launch
    uint32_t debug;           // -DPWAF_PROFILING timing experiments only (wrong results): 1 = no table lookups, 2 = no loads after a slab's first iteration
};
// The descriptors of all passes of a launch live in DEVICE memory (a 4096-rule set over 64 header fields has ~70 filtered passes:
// 15 KB of descriptors, kernel arguments hold 8). The host builds the descriptors of EVERY launch of a batch before the first one, writes
// them into a page-locked slot and one copy launch on the batch's stream moves the block (engine.cpp: Scratch::args): stream-ordered,
// one launch per batch whatever the number of passes.
int upload_args_block(const void *host_pinned, void *dev, size_t bytes, void *zero /* nullable: a block to clear in the same launch */, size_t zero_bytes, void *stream);
struct FilterTable {
    const FilterArgs *f;  // device
    uint32_t count;
};
struct FilterMix {  // the fused filter launch: the stride-1 passes and the stride-2 passes, each class with first_block numbered from 0
    const FilterArgs *f1, *f2;  // device
    uint32_t count1, count2, blocks1, blocks2;
};
// `host` = the same `count` descriptors as `dev` (launch geometry). launch_filter: ONE launch, the passes of stride 1 and of stride 2
// as two tables (either may be empty), first_block numbered within each by the caller.
int launch_filter(const FilterArgs *host1, uint32_t count1, const FilterArgs *dev1, const FilterArgs *host2, uint32_t count2, const FilterArgs *dev2, void *stream);
int launch_resolve(const FilterArgs *host, uint32_t count, const FilterArgs *dev, void *stream);
int launch_compact(const FilterArgs *host, uint32_t count, const FilterArgs *dev, void *stream);  // bitcount_kernel, then compact_kernel
// Sets the dynamic-LDS limit of every kernel on the CURRENT device (once per device and process; engines on several
// devices of one process each need it).
int configure_kernels(int device);

// Source words of the membership atoms of one request (bit_col maps (source word, bit) -> column): ip-list sets, country tables,
// port sets, asn sets, asn comparisons.
static constexpr uint32_t kSrcSet = 0, kSetWordsMax = 16, kSrcCc = 16, kCcWordsMax = 8, kSrcPort = 24, kSrcAsn = 28, kIntWordsMax = 4, kSrcAcmp = 32, kAcmpWordsMax = 4,
                          kSrcWords = 36;
struct PassInfo {
    uint32_t base;       // first device column of the pass
    uint32_t kind_slot;  // kind << 24 | slot
};
// Field-against-field atoms (program.h: ATOM_FCMP): one lane per request compares the two strings (or their lengths); the results
// are written as the hit record of one more pass whose local atom k is atoms[k].
struct FcmpArgs {
    const uint8_t *data[kMaxFcmpFields];
    const uint32_t *off[kMaxFcmpFields];
    uint32_t atoms[kMaxFcmpAtoms];  // op | field slot a << 8 | field slot b << 16
    uint32_t n, n_atoms;
    uint32_t *rec;
    PoolEntry *pool;
    uint32_t *pool_count;
    uint32_t pool_cap;
    uint32_t *status;
};
int launch_fcmp(const FcmpArgs &a, void *stream);
// Residual rules (residual.h): one lane per request interprets every residual rule's stack program; rule k that ends in Bool(true) is
// local atom k of the pseudo pass whose hit records `rec` holds.
struct ResidualArgs {
    const uint8_t *const *data;   // device arrays: per string column (5 fields, then the header columns) arena ...
    const uint32_t *const *off;   // ... and offsets
    const uint8_t *blob;          // rvm::Header + sections
    uint32_t n, n_rules;
    const uint8_t *ip;
    const uint8_t *ip_is_v6;
    const uint16_t *port;
    const uint32_t *asn;          // nullable: then the GeoIP record of the address supplies client.asn / client.country
    const uint16_t *country;
    uint32_t has_geo;             // the engine has a GeoIP table AND some rule reads asn / country
    const uint32_t *geo_root4, *geo_root6, *geo_nodes;  // the LPM trie with RECORD leaves
    const GeoRec *geo_recs;
    uint32_t *rec;
    PoolEntry *pool;
    uint32_t *pool_count;
    uint32_t pool_cap;
    uint32_t *status;
    unsigned long long *rule_errors;  // [n_rules], accumulated over the engine's batches: requests for which the rule's evaluation ended in an error
};
int launch_residual(const ResidualArgs &a, void *stream);
// The SPECIALIZED form (residual_jit.cpp: the same programs as straight-line device code, compiled by hiprtc when the engine is created —
// rtc.cpp): rvm_jit_kernel evaluates every rule for every request and writes one match bit per (request, rule) — match_words[w * n + r]
// bit k = rule 32 w + k — which the verdict kernel reads in place of the pseudo pass's hit records (VerdictArgs::res_match), and counts
// execution errors per rule. (The struct is mirrored in the generated program's text: plain pointers and words only.)
struct ResidualJitArgs {
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
    const GeoRec *geo_recs;
    uint32_t *match_words;
    unsigned long long *rule_errors;
};
struct JitKernel {
    void *module = nullptr, *function = nullptr;  // hipModule_t / hipFunction_t
};
static constexpr uint32_t kMaxJitRules = 256;  // (result words per request: ceil(rules / 32))
bool rvm_specialize(const uint8_t *blob, size_t len, std::string &out, std::string &why);   // residual_jit.cpp
bool rvm_jit_program(const uint8_t *blob, size_t len, std::string &out, std::string &why);
bool rtc_compile(const std::string &source, const std::string &arch, std::vector<char> &code, std::string &why);  // rtc.cpp
bool jit_load(const std::vector<char> &code, JitKernel &out, std::string &why);
void jit_release(JitKernel &k);
int launch_residual_jit(const JitKernel &k, const ResidualJitArgs &a, uint32_t n_cus, void *stream);
// The batch's string-column pointer table (ResidualArgs::data / off): part of the batch's descriptor block.
struct ColPtrChunk {
    const void *p[2 * (PWAF_N_FIELDS + kMaxHeaders)];
    uint32_t count;
};
struct CmpAtomDev {
    uint32_t col, c;
};
// A string atom that is an anchored literal of at most 8 bytes (`method == "POST"`, `method.starts_with("P")`) on a field whose
// pass consists of such atoms only: evaluated by the attribute kernel from the field's first 8 bytes instead of a DFA pass over
// every request (the pass then does not exist on the device: no walk, no hit records for the verdict kernel to read).
struct ShortAtom {
    uint32_t col;        // device column
    uint32_t len_exact;  // literal length | exact << 8 (exact: the field IS the literal; else it starts with it)
    uint32_t lit_lo, lit_hi;
};
struct VerdictArgs {
    uint32_t n, n_groups;
    uint32_t debug_skip;  // profiling aid (PWAF_DEBUG_SKIP env): bit k disables section k of the kernel; 0 in production
    uint32_t force_global_tables;  // PWAF_OPT_GLOBAL_VERDICT_TABLES: the verdict kernel variant for programs whose tables do not fit LDS (same results)
    uint32_t sparse_mode;          // verdict_shape's mode: 3 entry list (default), 4 entry list with 8 slots; 0 dense column file, 1 sparse, 2 sparse with 8 value slots
    uint32_t v_cap;                // sparse: value slots per wave in LDS; entry list: entries per wave in LDS
    unsigned long long *spill;     // sparse, v_cap < n_cols: [workgroups x waves][n_cols - v_cap] words for the dirty columns beyond v_cap; entry list: [workgroups x waves][n_cols], by column
    const uint32_t *off[PWAF_N_FIELDS];
    const uint8_t *ip;
    const uint8_t *ip_is_v6;
    const uint16_t *port;
    const uint8_t *flags;
    const uint32_t *asn;      // nullable
    const uint16_t *country;  // nullable
    // scan results
    uint32_t n_passes;
    const uint32_t *rec;        // [n_passes][n]
    // Per pass (device table, read once per wave): first column, and where its VISITED bitmap lives. A list-driven pass writes hit
    // records only for the requests it visits; bit r of the bitmap says record r is valid (everything else reads as "nothing
    // matched": no memset of 4 bytes per request and pass, no read of them here). kind 0 = the pass writes (or the host zeroes)
    // every record; 1 = bitmap `slot` of cand_bits (a filter's candidates); 2 = bitmap `slot` of visit_bits (a gap pass).
    const PassInfo *passes;
    // the residual rules' results when their specialized program ran (ResidualJitArgs::match_words; their pseudo pass then has no
    // records: kind 3): res_words words per request, bit k of word w = column res_base + 32 w + k
    const uint32_t *res_match;
    uint32_t res_words, res_base;
    const uint32_t *cand_bits, *visit_bits;
    uint32_t bit_words;  // words per bitmap (2 per 64-request group)
    // EXTENSION: header columns whose length is compared (comparison variable 7 + k)
    const uint32_t *hoff[kMaxHeaderLens];
    const ShortAtom *short_atoms;  // atoms of the short-literal field (see ShortAtom), n_short of them (0: none)
    uint32_t n_short;
    const uint8_t *short_data;     // that field's arena and offsets
    const uint32_t *short_off;
    uint32_t n_hlen;
    const PoolEntry *pool;
    // compiled program
    uint32_t n_cols;
    const CmpAtomDev *cmp;        // comparison atoms (LEN / INT against a constant): col = column | code << 24 with
                                  // code = 2 * variable (0-4 field lengths, 5 port, 6 asn) + operator (0: ==, 1: <=)
    uint32_t n_cmp;
    uint32_t cmp_vars;            // bit v: some (eager) comparison atom reads variable v — the attribute kernel fetches no other length
    // LAZY comparison atoms (program.h: LIT_LAZY), evaluated by the verdict kernel on demand: col = (2 * slot + operator) << 24, c = the
    // constant; slot s = comparison variable lazy_var[s] (as above: 0-4 field lengths, 5 port, 7 + k header lengths; at most two), whose raw
    // values the verdict kernel fetches with a group's other inputs
    const CmpAtomDev *lazy;
    uint32_t n_lazy;
    uint32_t lazy_var[4], n_lazy_var;
    const uint32_t *bit_col;      // [kSrcWords source words][32 bits] -> column of the membership atom, 0 = none
    // integer sets, merged per variable (0 = remote_port, 1 = asn): sorted distinct values + membership rows (row 0 = miss)
    const int64_t *iu_vals[2];
    const uint32_t *iu_masks[2];
    uint32_t iu_n[2], iu_words[2];
    const uint32_t *country_masks;  // [676][cc_words]: bit t of row c = country table t contains country code c
    uint32_t cc_words;
    const DevRule *rules;
    uint32_t n_rules;
    const uint32_t *lits;
    uint32_t n_lits, n_trig;
    const uint32_t *trig_off;     // [n_cols + 1]: rules triggered by a non-zero column (one positive literal per term)
    const uint16_t *trig_rules;
    const uint32_t *always_rules; // bitmap over rules with a term made of negations only
    // tries
    const uint32_t *ip_root4, *ip_root6, *ip_nodes;  // membership sets (null roots => set 0)
    const uint32_t *set_masks;
    uint32_t set_words;
    uint32_t n_ip_lists;
    const uint32_t *geo_root4, *geo_root6, *geo_nodes;
    // (roots are never null on the device: a family without prefixes gets an all-leaf root. GeoIP trie leaves are CLASS ids.)
    const uint32_t *dir24;        // BUILD TIME ONLY (dir24_kernel's output): first 24 bits of both IPv4 tries flattened: class | set << 16, or DIR_ESCAPE | index into dir_esc
    // ... compressed for the lookups: one 128-byte chunk per /16 (index = the address's top 16 bits) = 8 records of 4 words, one per
    // group of 32 /24s: {bitmap of the /24s where a run of equal entries starts, the entry in force when the group begins, the
    // entry of the first run starting inside the group, index in dir_vals of the entries of the group's further runs}.
    // dir_chunks null = walk from the roots.
    const uint32_t *dir_chunks, *dir_vals;
    // SUMMARY in front of the table (round 6): one bit per block of 2^dir_sum_shift /24s — 0 = every /24 of the block holds the table's
    // most common entry `dir_common` (for a WAF rule set: "no predicate cares about this address") and the lookup ends there. The bitmap
    // is a few hundred KiB (L2-resident), the table 8 MiB (94 % of 10M uniform lookups missed the XCD's 4 MiB L2: 1.2 GB of 128-byte
    // lines over the fabric per batch). Null: no summary (most blocks differ: the bit would cost a load and save nothing).
    const uint32_t *dir_summary;
    uint32_t dir_sum_shift, dir_common;
    const uint2 *dir_esc;         // {geo trie entry, ip-list trie entry} of the escaped /24s
    const uint32_t *class_rows;   // per GeoIP class: country-table words, asn-set words, asn-comparison words (class 0 = all zero)
    uint32_t class_words;
    uint32_t acmp_words;          // asn-comparison words per class row
    uint32_t geo_default;         // class of the default record {0, "XX"}
    uint32_t *ipres;              // per-batch scratch, ipres_kernel -> attr_kernel: GeoIP class and membership-set id of every request
    uint32_t ipres_packed;        // 1: one word per request (class | set << 16: both fit 16 bits), 0: two words
    uint32_t has_geo;
    // attribute kernel -> verdict kernel: per 64-request group the (column, mask) pairs of every non-scan atom that holds for
    // some request of the group. gpairs[g * pair_stride + k] = {column, 0, mask lo, mask hi}, ghdr[g] = number of pairs
    uint4 *gpairs;
    uint32_t *ghdr;
    uint32_t pair_stride;
    uint32_t attr_blocks;  // the device's CU count: grid of the attribute kernel and of the LDS-table verdict kernel
    // outputs
    pwaf_verdict *out;
    unsigned long long *counts;  // 4, accumulated (nullable)
    uint32_t *match_idx;         // nullable
    uint32_t *n_matches;         // nullable
};

// Launchers (hipStream_t passed as void*). Return hipError_t as int.
int launch_scan(const ScanArgs &a, void *stream);
struct GatedTable {
    const ListScanArgs *g;  // device
    uint32_t count;
    uint32_t debug;  // -DPWAF_PROFILING timing experiments only: 1 = never, 2 = always the asynchronous loop for a list (same results)
};
// Workgroup shape of the list scan: threads per workgroup, LDS bytes of hot rows per workgroup, workgroups per CU.
struct ListShape {
    uint32_t threads, hot_bytes, wg_per_cu;
};
ListShape list_shape(uint32_t variant);  // 0 = default
// LDS bytes for hot rows in a launch of this shape
uint32_t list_hot_bytes(const ListShape &shape);
// `plan`: count + 1 words of device scratch (the work-item prefix sums, written by lscan_plan_kernel on the same stream)
int launch_scan_gated(const ListScanArgs *host, uint32_t count, const ListScanArgs *dev, uint32_t *plan, const ListShape &shape, void *stream);
int launch_verdict(const VerdictArgs &a, void *stream);
int launch_ipres(const VerdictArgs &a, void *stream);  // address lookups -> a.ipres; then, on the same stream:
int launch_attr(const VerdictArgs &a, void *stream);
int launch_dir24(const VerdictArgs &a, void *out /* 2^24 x u32, or null: count only */, void *esc, void *esc_count, void *stream);
uint32_t scan_lds_bytes(uint32_t n_hot, uint32_t stride, uint32_t n_gate_atoms);
struct VerdictShape {
    uint32_t waves, lds_bytes, lds_tables;
    uint32_t sparse, v_cap, per_cu;  // sparse: 1 = the sparse column file, 2 = the entry list (verdict2_kernel); (kernels.hip: verdict_kernel<.., SP>): value slots per wave, workgroups per CU
};
// force_global (PWAF_OPT_GLOBAL_VERDICT_TABLES): take the variant whose program tables stay in global memory even when they would fit LDS
// mode: 3 = entry list (default), 4 = entry list with 8 entry slots (PWAF_OPT_TINY_VERDICT_SLOTS: the spill path's test hook); 0 = the dense column file
// (PWAF_OPT_DENSE_VERDICT), 1 = the sparse column file of round 5 (PWAF_OPT_SPARSE_VERDICT), 2 = that with 8 value slots (both flags)
VerdictShape verdict_shape(uint32_t n_cols, uint32_t n_rules, uint32_t n_trig, uint32_t n_lits, bool force_global = false, int mode = 3, uint32_t n_passes = 0);
uint32_t verdict_blocks_sp(const VerdictShape &sh, uint32_t n_cus);

}  // namespace pwaf
