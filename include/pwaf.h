/*
 * pwaf.h — C ABI of the MI355X batched WAF rule-matching engine (libpwaf.so).
 *
 * This is the drop-in boundary for ONE hot path of pingooio/pingoo: the per-request
 * rule evaluation that the reference runs inline in its hyper request closure.
 * The reference has no FFI/plugin seam for this path, so every entry
 * point below cites the reference code it replaces. All citations are relative to
 * the reference tree.
 *
 *   reference construct                                        replaced by
 *   ---------------------------------------------------------  ---------------------------
 *   rules::compile_expression        rules/rules.rs:45-53      pwaf_compile_expression
 *   rules::validate_expression       rules/rules.rs:55-77      pwaf_validate_expression
 *   Vec<Rule> + lists + GeoipDB      pingoo/server.rs:40-47,76 pwaf_engine_create
 *   rule loop + gates                http_listener.rs:196-264  pwaf_evaluate_batch / _one
 *   Rule::match_request              pingoo/rules.rs:37-51     (inside the batch evaluation)
 *   GeoipDB::lookup                  pingoo/geoip.rs:73-91     (device trie, or caller-supplied asn/country)
 *   get_host / get_path / UA derive  http_listener.rs:140-165,284-296; http_utils.rs:114-116
 *                                                              pwaf_derive_host / _path / _user_agent
 *
 * Plain pointers and sizes only; no C++/torch types. Never aborts across the ABI:
 * every failure is an int status (<0) plus pwaf_last_error().
 */
#ifndef PWAF_H
#define PWAF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWAF_ABI_VERSION 3u /* 2: pwaf_batch carries header columns; strict rule compilation by default (PWAF_OPT_LENIENT, PWAF_W_PARTIAL);
                            * residual rules; pwaf_request carries header values; pwaf_node_evaluate_device; pwaf_program_tune
                            * 3: page-locked host memory for batch columns (pwaf_host_alloc / _register); no struct changed */

/* ---- status codes ------------------------------------------------------------------ */
#define PWAF_OK 0
#define PWAF_E_INVALID_ARG (-1)  /* NULL pointer, bad struct_size, bad enum value            */
#define PWAF_E_SYNTAX (-2)       /* == rules::Error::ExpressionIsNotValid (rules/rules.rs:41) */
#define PWAF_E_UNSUPPORTED (-3)  /* valid expression outside the device-compilable subset     */
#define PWAF_E_LIST (-4)         /* list item does not parse (pingoo/lists.rs:93-108)         */
#define PWAF_E_DEVICE (-5)       /* HIP error / no GPU: host may fail open                    */
#define PWAF_E_BATCH (-6)        /* malformed batch (offsets not monotone, bad country, ...)  */
#define PWAF_E_NOMEM (-7)
#define PWAF_W_PARTIAL 1          /* (positive: a warning, the object WAS created) with PWAF_OPT_LENIENT some rule is not evaluated */

/* ---- verdict vocabulary (rules::Action, rules/rules.rs:30-35) ------------------------ */
#define PWAF_ACTION_ALLOW 0u   /* no rule fired: proceed to routing (http_listener.rs:266)   */
#define PWAF_ACTION_BLOCK 1u   /* Action::Block {}   -> 403 (http_listener.rs:255)           */
#define PWAF_ACTION_CAPTCHA 2u /* Action::Captcha {} -> captcha page (http_listener.rs:256)  */
#define PWAF_ACTION_BYPASS 3u  /* path under /__pingoo/captcha: rules skipped (:200-204)     */

#define PWAF_RULE_NONE 0xFFFFFFFFu             /* verdict not caused by a rule (Allow)       */
#define PWAF_RULE_UA_GATE 0xFFFFFFFEu          /* UA empty or >=256 B (http_listener.rs:196) */
#define PWAF_RULE_CAPTCHA_ENDPOINT 0xFFFFFFFDu /* http_listener.rs:200                        */

/* action kinds inside a rule description (serde tag "action": block | captcha) */
#define PWAF_RULE_ACTION_BLOCK 1u
#define PWAF_RULE_ACTION_CAPTCHA 2u

typedef struct pwaf_engine pwaf_engine;   /* compiled rules + device tables (immutable, thread-safe) */
typedef struct pwaf_program pwaf_program; /* host-side compiled form only (no GPU needed)            */

/* One rule == pingoo::rules::Rule {name, expression: Option<Program>, actions} (pingoo/rules.rs:9-14).
 * Array order == evaluation order (first match wins, http_listener.rs:251-264). */
typedef struct pwaf_rule_desc {
    const char *name;       /* NUL-terminated, for diagnostics                         */
    const char *expression; /* NUL-terminated; NULL == match-all (pingoo/rules.rs:48-50) */
    const uint8_t *actions; /* PWAF_RULE_ACTION_*, in order                            */
    uint32_t n_actions;
    uint32_t reserved;
} pwaf_rule_desc;

/* One list == pingoo/lists.rs:11-15. Items are the raw CSV column-0 strings; the engine trims
 * them and parses Int / IpNetwork exactly where the reference does (lists.rs:90-108). */
#define PWAF_LIST_STRING 0u
#define PWAF_LIST_INT 1u
#define PWAF_LIST_IP 2u
typedef struct pwaf_list_desc {
    const char *name;
    uint32_t type;
    uint32_t n_items;
    const char *const *items;
} pwaf_list_desc;

/* GeoIP prefix table: the decoded content of a MaxMind DB as the reference consumes it
 * (pingoo/geoip.rs:17-23: asn u32, country 2 x 'A'..'Z'). Longest prefix wins. */
typedef struct pwaf_geoip_entry {
    uint8_t addr[16];  /* v4: first 4 bytes, network order; v6: all 16 */
    uint8_t prefix_len;
    uint8_t is_v6;
    uint8_t country[2];
    uint32_t asn;
} pwaf_geoip_entry;
typedef struct pwaf_geoip_table {
    const pwaf_geoip_entry *entries;
    size_t n_entries;
} pwaf_geoip_table;

/* ---- data-file readers (the callers' data formats on either side of the path) ----------------------------- */
/* Flattens a MaxMind DB (the uncompressed content of geoip.mmdb; for .zst see pwaf_geoip_from_file_image) into the prefix table above: one entry per network of the search tree, IPv4 networks of an IPv6 database (those
 * below ::/96) emitted as IPv4 entries as well. Records are read the way the reference deserialises them (pingoo/geoip.rs:17-23,
 * serde_utils.rs:1-9): {"asn": "AS<digits>" string, "country": two upper-case letters}; a record that would make the
 * reference's lookup fail yields the default {0, "XX"} for its network (http_listener.rs:143-157). Replaces
 * maxminddb::Reader::from_source + lookup (geoip.rs:57,73-91). *entries_out is malloc'ed: release with pwaf_geoip_free. */
int pwaf_geoip_from_mmdb(const uint8_t *mmdb, size_t len, pwaf_geoip_entry **entries_out, size_t *n_out);
void pwaf_geoip_free(pwaf_geoip_entry *entries);
/* zstd::decode_all for `.zst` databases (pingoo/geoip.rs:49-55; the reference's Docker image ships geoip.mmdb.zst): libzstd is loaded
 * at run time; PWAF_E_UNSUPPORTED when it is not installed. *out is malloc'ed: release with pwaf_buffer_free. */
int pwaf_zstd_decompress(const uint8_t *src, size_t len, uint8_t **out, size_t *out_len);
void pwaf_buffer_free(uint8_t *);
/* GeoipDB::load on a file image: decompresses when `path` ends in ".zst" (geoip.rs:49-55), then pwaf_geoip_from_mmdb. */
int pwaf_geoip_from_file_image(const char *path, const uint8_t *content, size_t len, pwaf_geoip_entry **entries_out, size_t *n_out);
/* The list-file format of pingoo/lists.rs:62-117: CSV without header, 1 or 2 columns, the first column trimmed is the item.
 * Returns the items as malloc'ed NUL-terminated strings for pwaf_list_desc.items: release with pwaf_list_free. */
int pwaf_list_parse_csv(const char *text, size_t len, char ***items_out, size_t *n_out);
void pwaf_list_free(char **items, size_t n);

#define PWAF_OPT_NO_UA_GATE 1u        /* skip gate A (http_listener.rs:196-198)             */
#define PWAF_OPT_NO_CAPTCHA_BYPASS 2u /* skip gate B (http_listener.rs:200-204)             */
#define PWAF_OPT_STRICT 8u            /* (deprecated: strict is the default since ABI 2) */
#define PWAF_OPT_LENIENT 32u          /* a rule neither the column compiler nor the residual interpreter can take does NOT fail creation:
                                       * that rule alone never matches, pwaf_program_rule_status / the warnings say why, and creation
                                       * returns PWAF_W_PARTIAL instead of PWAF_OK. Default (ABI 2): creation fails with the rule's index —
                                       * the reference evaluates every valid expression (pingoo/rules.rs:37-51), a silently dropped Block
                                       * rule is a fail-open hole */
#define PWAF_OPT_GLOBAL_VERDICT_TABLES 128u /* testing: the verdict kernel variant for programs whose tables do not fit LDS (same verdicts) */
#define PWAF_OPT_SPARSE_VERDICT 16384u /* A-B / testing: the verdict kernel with the round-5 SPARSE column file (dirty bits + rank per 32 columns) instead of the entry list (same verdicts) */
#define PWAF_OPT_DENSE_VERDICT 2048u  /* A-B / testing: the verdict kernel with the round-4 DENSE column file (8 bytes per column and wave) instead of the sparse one (same verdicts) */
#define PWAF_OPT_TINY_VERDICT_SLOTS 4096u /* testing: 8 entry slots per wave (with PWAF_OPT_SPARSE_VERDICT: 8 value slots of the sparse column file), so that every group takes the spill path (same verdicts) */
#define PWAF_OPT_NO_DENSE_SWITCH 65536u /* A-B / testing: a pass whose prefilter flags most of the arena is still confirmed chunk by chunk (round 5), never walked whole (same verdicts) */
#define PWAF_OPT_EAGER_CMP 32768u      /* A-B / testing: every length / port comparison is evaluated per group by the attribute kernel (round 5), none lazily by the verdict kernel (same verdicts) */
#define PWAF_OPT_NO_DIR_SUMMARY 8192u  /* A-B / testing: IPv4 lookups always gather from the compressed DIR-24 table, without the summary bitmap in front of it (same verdicts) */
#define PWAF_OPT_NO_RESIDUAL 64u      /* do not use the per-request residual interpreter (testing / benchmarking the column path alone) */
#define PWAF_OPT_NO_RESIDUAL_JIT 1024u /* residual rules are INTERPRETED per request (residual_kernel) instead of running as the specialized
                                       * device program compiled by hiprtc when the engine is created (the default): same verdicts */
#define PWAF_OPT_NO_PREFILTER 4u      /* every scan pass walks its DFA over every request (no bigram prefilter): same verdicts */
#define PWAF_OPT_FILTER_STRIDE2 16u   /* prefilters sample every second byte wherever a pass's patterns allow it (default: stride 1
                                       * until pwaf_engine_tune decides per pass from the traffic sample): same verdicts */
#define PWAF_OPT_NO_CONFIRM 512u      /* testing / A-B: no confirm tier — every prefilter candidate is walked through the pass's full DFA
                                       * (the round-3 path): same verdicts */
typedef struct pwaf_options {
    uint32_t struct_size; /* sizeof(pwaf_options) */
    uint32_t flags;
    int32_t device;            /* HIP device ordinal, -1 = current device                 */
    uint32_t lds_table_budget; /* LDS bytes for the hot rows of one DFA table; 0 = default (128 KiB) */
    uint32_t max_dfa_states;   /* per DFA group, <= 32767; 0 = default (32767)                       */
    uint32_t max_table_bytes;  /* per DFA group (L2-resident transition table); 0 = default (3 MiB)  */
    uint32_t reserved[2];
} pwaf_options;

/* Per-rule diagnostics of engine creation. */
typedef struct pwaf_compile_error {
    int32_t code;        /* PWAF_E_* */
    uint32_t rule_index; /* offending rule, or 0xFFFFFFFF */
    char message[248];
} pwaf_compile_error;

/* ---- the batch: struct-of-arrays, one request per index --------------------------------
 * String field i of request r is data[offsets[r] .. offsets[r+1]). offsets has n+1 entries and is
 * non-decreasing, so each field's bytes are contiguous in request order (field-major arena).
 * `data` must be readable for 16 bytes past offsets[n] (PWAF_ARENA_PAD) when memory == DEVICE.
 * Field meaning == RequestData / ClientData (pingoo/rules.rs:16-34) AFTER the reference's own
 * derivation (pwaf_derive_* below). */
#define PWAF_FIELD_HOST 0
#define PWAF_FIELD_URL 1
#define PWAF_FIELD_PATH 2
#define PWAF_FIELD_METHOD 3
#define PWAF_FIELD_USER_AGENT 4
#define PWAF_N_FIELDS 5
#define PWAF_ARENA_PAD 16u

#define PWAF_MEM_HOST 0u
#define PWAF_MEM_DEVICE 1u

#define PWAF_FLAG_CAPTCHA_VERIFIED 1u /* http_listener.rs:222-236, decided on the host (JWT) */

typedef struct pwaf_strcol {
    const uint8_t *data;
    const uint32_t *offsets; /* n+1 */
} pwaf_strcol;

typedef struct pwaf_batch {
    uint32_t struct_size; /* sizeof(pwaf_batch) */
    uint32_t n;
    uint32_t memory; /* PWAF_MEM_HOST | PWAF_MEM_DEVICE: where EVERY pointer below lives */
    uint32_t n_headers; /* header columns below (extension, see `headers`); 0 = none */
    pwaf_strcol field[PWAF_N_FIELDS];
    const uint8_t *ip;       /* n x 16; IPv4 in bytes 0..3 (network order), rest ignored  */
    const uint8_t *ip_is_v6; /* n                                                        */
    const uint16_t *port;    /* n; client.remote_port                                    */
    const uint8_t *flags;    /* n; PWAF_FLAG_*                                           */
    /* Optional pre-computed GeoIP (both or neither). When NULL the engine looks the ip up in its
     * own table on the device, or uses {0,"XX"} if it has none (geoip.rs:111-118). */
    const uint32_t *asn;     /* n */
    const uint16_t *country; /* n; two bytes 'A'..'Z' in memory order */
    /* Byte size of each field arena (= offsets[n]). The bigram prefilter streams an arena as one flat byte range, so the launch
     * geometry depends on it. HOST batches: ignored (read from the offsets). DEVICE batches: the caller that built the arenas
     * knows it; 0 = unknown, the engine then reads offsets[n] back with one small synchronous copy per evaluate call. */
    uint32_t field_bytes[PWAF_N_FIELDS];
    uint32_t reserved;
    /* EXTENSION (no reference counterpart: pingoo/rules.rs:16-25 exposes no headers — BASELINE.json configs[4]):
     * one string column per header name the rule set mentions as http_request.headers["name"], in the order of
     * pwaf_engine_header_name(); a request without the header carries the empty string. */
    const pwaf_strcol *headers;    /* n_headers column descriptors (the ARRAY is in host memory; the pointers inside follow `memory`) */
    const uint32_t *header_bytes;  /* n_headers arena sizes (HOST memory, same rule as field_bytes), or NULL */
} pwaf_batch;

typedef struct pwaf_verdict {
    uint8_t action; /* PWAF_ACTION_* */
    uint8_t pad[3];
    uint32_t rule_idx; /* index into the rule array, or PWAF_RULE_* */
} pwaf_verdict;

typedef struct pwaf_counts {
    uint64_t by_action[4]; /* indexed by PWAF_ACTION_* */
} pwaf_counts;

typedef struct pwaf_span {
    const char *data; /* not NUL-terminated */
    uint32_t len;
    uint32_t reserved;
} pwaf_span;

/* One request, for the evaluate(Request)->Action convenience wrapper. */
typedef struct pwaf_request {
    const char *host, *url, *path, *method, *user_agent; /* not NUL-terminated */
    uint32_t host_len, url_len, path_len, method_len, user_agent_len;
    uint8_t ip[16];
    uint8_t ip_is_v6;
    uint8_t flags;
    uint16_t port;
    uint8_t has_geoip; /* 1: asn/country below are valid */
    uint8_t country[2];
    uint8_t pad;
    uint32_t asn;
    /* EXTENSION (ABI 2): header values for rule sets that read http_request.headers["name"]: headers[k] is the value of header name k of
     * the engine (pwaf_engine_header_name(k); an absent header = {NULL, 0}), n_headers <= pwaf_engine_header_count — the rest reads as "".
     * The reference's RequestData has no headers (pingoo/rules.rs:16-25); the call shape served is http_listener.rs:206-264. */
    uint32_t n_headers;
    const struct pwaf_span *headers;
} pwaf_request;

/* ---- expression front-end (no GPU needed) ------------------------------------------------ */
/* rules::compile_expression (rules/rules.rs:45-53): syntax only. 0 or PWAF_E_SYNTAX. */
int pwaf_compile_expression(const char *expression, char *errbuf, size_t errbuf_len);
/* rules::validate_expression (rules/rules.rs:55-77): also rejects "" and the `in` operator. */
int pwaf_validate_expression(const char *expression, char *errbuf, size_t errbuf_len);

/* ---- host-side compilation only (used by engine_create; exported for inspection/tests) ---- */
int pwaf_program_compile(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists,
                         size_t n_lists, const pwaf_geoip_table *geoip, const pwaf_options *opts,
                         pwaf_program **out, pwaf_compile_error *err);
void pwaf_program_destroy(pwaf_program *);
/* Serialises the compiled tables as a self-describing little-endian blob;
 * returns the byte size; copies at most `cap` bytes into `buf` (buf may be NULL to query). */
size_t pwaf_program_dump(const pwaf_program *, uint8_t *buf, size_t cap);
/* Number of per-rule warnings (statically-erroring expressions that can never match,
 * pingoo/rules.rs:41-45) and their text. */
/* Per caller rule: PWAF_OK, or PWAF_E_UNSUPPORTED with the reason in `msg` when the device compiler cannot evaluate its expression
 * (DESIGN.md §3.5). The reference evaluates any expression (pingoo/rules.rs:37-51); an unsupported rule here never matches — the
 * reference's own behaviour for a rule whose execution fails (rules.rs:41-45) — and the host can see which ones. */
int pwaf_program_rule_status(const pwaf_program *, uint32_t rule_index, char *msg, size_t msg_len);
size_t pwaf_program_warning_count(const pwaf_program *);
const char *pwaf_program_warning(const pwaf_program *, size_t i);

/* ---- engine ------------------------------------------------------------------------------ */
int pwaf_engine_create(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists,
                       size_t n_lists, const pwaf_geoip_table *geoip /* nullable */,
                       const pwaf_options *opts /* nullable */, pwaf_engine **out,
                       pwaf_compile_error *err /* nullable */);
void pwaf_engine_destroy(pwaf_engine *);
const pwaf_program *pwaf_engine_program(const pwaf_engine *);
/* EXTENSION (header fields): the header names the rule set mentions as http_request.headers["name"] (exact, case-sensitive
 * strings — HTTP/2 and the `http` crate present them in lower case), in the column order pwaf_batch.headers must use. */
uint32_t pwaf_engine_header_count(const pwaf_engine *);
const char *pwaf_engine_header_name(const pwaf_engine *, uint32_t i);
uint32_t pwaf_program_header_count(const pwaf_program *);
const char *pwaf_program_header_name(const pwaf_program *, uint32_t i);
void *pwaf_engine_stream(const pwaf_engine *); /* hipStream_t the synchronous entry points run on */

/* Synchronous batch evaluation. HOST batches are copied to the device, evaluated, and verdicts are
 * copied back into `out` (host, n entries). DEVICE batches are evaluated in place and `out`/`counts`
 * must be device pointers too. `counts` is nullable. Callable concurrently from several threads: an engine owns a small ring of
 * per-call contexts (scratch, staging buffers, stream), so one caller's copies overlap another's kernels. A batch that exhausts
 * the scan overflow pool (more than two distinct matches per pass for very many requests) is evaluated again with a larger pool
 * before the call returns: PWAF_E_NOMEM is only reported when that does not help. */
int pwaf_evaluate_batch(pwaf_engine *, const pwaf_batch *in, pwaf_verdict *out, pwaf_counts *counts);

/* Page-locked host memory for the columns of HOST batches (ABI 3). Where the reference's listener has the request's bytes when the
 * rule loop runs (http_listener.rs:206-219) is host memory; a host that parses into arenas from pwaf_host_alloc (or registers its own
 * with pwaf_host_register) lets pwaf_evaluate_batch's copies run at the link's speed: the copy engine reads such memory directly and
 * the calling thread validates the batch meanwhile. Pageable columns still work (the runtime stages them: slower). Small host batches
 * (under 1 MiB: the micro-batcher's) are packed into one page-locked block inside the engine whatever the caller's memory is. */
int pwaf_host_alloc(size_t bytes, void **out);
void pwaf_host_free(void *);
int pwaf_host_register(void *p, size_t bytes);
int pwaf_host_unregister(void *p);

/* Asynchronous device-resident evaluation on a caller-supplied HIP stream (hipStream_t as void*, passed through
 * unchanged: NULL is HIP's default stream; pwaf_engine_stream() returns the engine's own non-blocking stream). All pointers (batch columns, out, counts, match_idx, n_matches)
 * are device pointers. `match_idx`/`n_matches` (nullable) receive the compacted indices of
 * non-Allow requests (wavefront ballot + prefix-sum compaction; order unspecified).
 * `counts` and `n_matches` are ACCUMULATED into (caller zeroes them). */
int pwaf_evaluate_device(pwaf_engine *, const pwaf_batch *in, pwaf_verdict *out, pwaf_counts *counts,
                         uint32_t *match_idx, uint32_t *n_matches, void *stream);

/* pwaf_evaluate_device is re-entrant: calls from several threads / on several streams each take the next per-call context of the
 * engine's ring (the stream first waits, on the device, for that context's previous user), so two in-flight batches never share
 * scratch. After pwaf_evaluate_device: pwaf_engine_device_status waits for the device and returns PWAF_OK, or PWAF_E_NOMEM when
 * some device-resident batch since the last status call ran out of scan overflow scratch (its verdicts are incomplete: evaluate it
 * again — the pool has been grown to what it asked for; cannot happen below 8 overflowing hits per request on average). */
int pwaf_engine_device_status(pwaf_engine *);

/* Optional tuning from a traffic sample (HOST memory; at most the first 65536 requests are used). Each DFA pass keeps its most
 * visited states in LDS; without a profile those are the shallowest states (BFS order), with one they are the states the sample
 * actually visits. Only speed depends on this: verdicts are identical with any profile. Synchronises the device, then rebuilds
 * and re-uploads the scan tables. No reference counterpart (the reference interprets each rule per request:
 * pingoo/rules.rs:37-51); the closest analogue is warming its caches. */
int pwaf_engine_tune(pwaf_engine *, const pwaf_batch *sample);
/* The host half of the same tuning on a compiled PROGRAM (no device needed): the program's bigram prefilters are replaced by the
 * ones rebuilt for the sample, and show up in pwaf_program_dump. Lets a deployment (and the CPU tests) inspect what tuning would
 * do to the tables before any engine exists; an engine is not affected. */
int pwaf_program_tune(pwaf_program *, const pwaf_batch *sample);

/* evaluate(Request) -> Action: a batch of one (north_star's RuleEngine::evaluate façade). */
int pwaf_evaluate_one(pwaf_engine *, const pwaf_request *req, pwaf_verdict *out);

/* ---- one process, every GPU of the node ---------------------------------------------
 * The reference builds its rule state once per process and shares it with every listener (pingoo/server.rs:40-47,76,111-134);
 * a pwaf_node is that object for a multi-GPU host: one engine replica per device (tables are tens of MB), a HOST batch is cut
 * into contiguous 64-aligned slabs (pwaf_node_shard_bounds), each slab is evaluated by its device on its own host thread and
 * stream, verdicts land in `out` at the slab's position and the four action counters are summed on the host. No data-path
 * exchange between devices. (bench.py's process-per-GPU mode does the same with an RCCL all-reduce of the counters.) */
typedef struct pwaf_node pwaf_node;
int pwaf_node_create(const pwaf_rule_desc *rules, size_t n_rules, const pwaf_list_desc *lists, size_t n_lists,
                     const pwaf_geoip_table *geoip /* nullable */, const pwaf_options *opts /* nullable; .device is ignored */,
                     const int *devices, size_t n_devices, pwaf_node **out, pwaf_compile_error *err /* nullable */);
void pwaf_node_destroy(pwaf_node *);
size_t pwaf_node_device_count(const pwaf_node *);
pwaf_engine *pwaf_node_engine(const pwaf_node *, size_t i);
int pwaf_node_tune(pwaf_node *, const pwaf_batch *sample);
int pwaf_node_evaluate_batch(pwaf_node *, const pwaf_batch *in /* HOST */, pwaf_verdict *out /* n */, pwaf_counts *counts /* nullable */);
/* DEVICE-resident slabs, one per device of the node (ABI 2): batches[r] lives on device r (its slab of the request stream:
 * pwaf_node_shard_bounds), outs[r] / counts[r] (nullable, ACCUMULATED into like pwaf_evaluate_device's) are device pointers on device
 * r, streams[r] (nullable array / entries) is a HIP stream of device r. No host staging: the enqueue runs on the node's persistent
 * per-device host threads in parallel and returns without waiting for the devices. Replaces, for a single-process host, what
 * pingoo/server.rs:111-134 does when it hands the shared rule state to every listener. */
int pwaf_node_evaluate_device(pwaf_node *, const pwaf_batch *const *batches, pwaf_verdict *const *outs, pwaf_counts *const *counts, void *const *streams);
/* Waits for every device; PWAF_E_NOMEM when a device-resident slab since the last call ran out of scan scratch (see pwaf_engine_device_status). */
int pwaf_node_synchronize(pwaf_node *);
/* The path's only cross-device exchange — the sum of the four action counters — as an RCCL all-reduce over xGMI, in place on the
 * per-device counters: comms[r] = the caller's ncclComm_t for device r (librccl.so is loaded on first use). Hosts that add the 4 x 8
 * bytes per device up themselves do not need it. */
int pwaf_node_allreduce_counts(pwaf_node *, void *const *comms, pwaf_counts *const *counts, void *const *streams);
/* Slab [lo, hi) of device `rank` out of `world` for a batch of n requests. */
void pwaf_node_shard_bounds(uint32_t n, uint32_t rank, uint32_t world, uint32_t *lo, uint32_t *hi);

/* ---- deadline micro-batcher ------------------------------------------------------ */
/* The reference evaluates rules once per request on the tokio worker that owns the connection (http_listener.rs:196-264); a
 * drop-in RuleEngine::evaluate(Request) -> Action keeps that call shape. The batcher gathers concurrent callers: each call blocks
 * until its batch — closed when it holds max_batch requests, when its oldest request has waited max_delay_us, or (early close) when
 * every caller currently inside the call is already waiting in a batch and the oldest has waited max_delay_us / 8: callers block, so
 * nobody else can join before somebody is answered — has gone through pwaf_evaluate_batch. Thread-safe; two dispatcher threads per batcher (one gathers and submits the next batch while the other waits for its own on the device). Destroy it before the engine. */
typedef struct pwaf_batcher pwaf_batcher;
int pwaf_batcher_create(pwaf_engine *engine, uint32_t max_batch, uint32_t max_delay_us, pwaf_batcher **out);
int pwaf_batcher_evaluate(pwaf_batcher *, const pwaf_request *req, pwaf_verdict *out);
int pwaf_batcher_stats(pwaf_batcher *, uint64_t *n_batches, uint64_t *n_requests);
void pwaf_batcher_destroy(pwaf_batcher *);

/* ---- measurement ------------------------------------------------------------------------- */
typedef struct pwaf_kernel_time {
    char name[48];
    float ms;            /* HIP-event duration of the last profiled evaluate call */
    uint64_t alg_bytes;  /* algorithmic bytes this launch is credited with */
} pwaf_kernel_time;
/* on = 1: every evaluate call brackets each kernel with hipEvents on the launch stream; on = 2: only the launches that STREAM the request
 * bytes (filter_kernel, scan_kernel: what the HBM roofline is quoted on) — an event between two kernels keeps the second from starting
 * while the first drains, ~20 us per batch of a dozen launches. Calling it (any value) starts a new measurement window. */
int pwaf_engine_set_profiling(pwaf_engine *, int on);
/* Blocks until the last profiled launch finished; fills up to cap entries, one per kernel launch since the window
 * started (in launch order); returns the count. */
int pwaf_engine_kernel_times(pwaf_engine *, pwaf_kernel_time *out, int cap);

typedef struct pwaf_stats {
    uint32_t n_rules, n_atoms, n_scan_atoms, n_numeric_atoms;
    uint32_t n_dfa_groups, n_dfa_states_total, max_dfa_states, dfa_table_bytes_total;
    uint32_t n_ip_lists, ipset_trie_nodes, geo_trie_nodes, n_dnf_literals;
    uint32_t n_warnings, n_filtered_groups /* passes behind a bigram prefilter */,
        n_gated_groups /* gap passes visited only by requests whose prefix factor was found */,
        n_confirm_literals /* string predicates decided by the confirm tier of their pass's prefilter, without a DFA */;
} pwaf_stats;
int pwaf_engine_stats(const pwaf_engine *, pwaf_stats *out);
/* Execution-error visibility (the reference logs every rule whose execution errs and treats it as "no match": pingoo/rules.rs:41-45).
 * Here an error is a no-match too; errors the compiler can see — unknown names, type errors, invalid patterns: nearly all of them,
 * the context's schema being fixed — are reported ONCE at creation (pwaf_program_warning: "can never match"). What remains are
 * run-time errors of rules on the per-request interpreter (checked arithmetic: overflow, division by zero; an index computed from a
 * request value): counted on the device. counts[i] = requests, over every batch this engine has evaluated, for which caller rule i
 * ended in an execution error. Waits for the device. */
int pwaf_engine_rule_errors(pwaf_engine *, uint64_t *counts, size_t n_rules);
/* How the engine evaluates the rules outside the column compiler's subset (the reference: Program::execute on every request,
 * pingoo/rules.rs:37-51): 0 = the rule set has none, 1 = interpreted per request on the device (residual_kernel),
 * 2 = SPECIALIZED — the rules' stack programs translated to straight-line device code and compiled for this device by hiprtc at
 * creation (csrc/residual_jit.cpp, rtc.cpp). When 1 was not asked for (PWAF_OPT_NO_RESIDUAL_JIT), pwaf_engine_residual_fallback
 * says why ("" otherwise; the string lives as long as the engine): no libhiprtc.so on the host, a compile error, too many rules. */
int pwaf_engine_residual_mode(const pwaf_engine *);
const char *pwaf_engine_residual_fallback(const pwaf_engine *);
/* Inspection / test hooks of the specialized form (CPU, no device). _source: kind 0 = the rule functions alone (portable C++ over
 * csrc/residual.h: the CPU suite compiles them with g++ and fuzzes them against the oracle), kind 1 = the whole device program as
 * handed to hiprtc. Returns the text's length (0: the program has no residual rules, or they cannot be specialized); copies at most
 * cap - 1 bytes + NUL. _compile: runs hiprtc for `arch` ("gfx950"); returns the code object's size, or a negative PWAF_E_* with the
 * compiler's log in err. */
size_t pwaf_program_residual_source(const pwaf_program *, int kind, char *buf, size_t cap);
long pwaf_program_residual_compile(const pwaf_program *, const char *arch, char *err, size_t err_len);
/* TEST HOOK (CPU, no device): the bigram prefilter + confirm tier of scan pass `group` over ONE field value placed `arena_offset`
 * bytes into an arena, as the device evaluates it (csrc/confirm.h is the code both run). Writes the local atom ids of the literal
 * predicates confirmed (at most cap; possibly repeated), *n_atoms, *flagged (the filter flagged the field) and *walk (a factor of a
 * non-literal predicate was confirmed: the request is walked through the pass's R-tier DFA). PWAF_E_INVALID_ARG when the pass has
 * no prefilter. The reference has no counterpart: it evaluates every predicate on every request (pingoo/rules.rs:37-51). */
int pwaf_program_confirm_field(const pwaf_program *, uint32_t group, const uint8_t *bytes, size_t len, uint32_t arena_offset,
                               uint16_t *atoms, size_t cap, size_t *n_atoms, int *flagged, int *walk);
int pwaf_program_stats(const pwaf_program *, pwaf_stats *out);

/* ---- host-side field derivation (what the reference does before building RequestData) ------ */
/* get_path: uri.path().trim_end_matches('/') (http_utils.rs:114-116). Returns the kept length. */
size_t pwaf_derive_path(const uint8_t *uri_path, size_t len);
/* User-Agent (http_listener.rs:159-165): header bytes -> to_str() (fails -> "" unless every byte is
 * visible ASCII 0x20..0x7E or TAB) -> trim -> heapless::String<256> (len > 256 -> ""). Writes the
 * start offset and length of the kept slice. `present` = 0 when the header is absent. */
void pwaf_derive_user_agent(const uint8_t *hdr, size_t len, int present, size_t *out_start, size_t *out_len);
/* get_host (http_listener.rs:284-296): uri.host() if any (trim), else Host header to_str() (trim);
 * longer than 256 -> "". Same output convention. */
void pwaf_derive_host(const uint8_t *uri_host, size_t uri_host_len, int uri_host_present,
                      const uint8_t *host_hdr, size_t host_hdr_len, int host_hdr_present,
                      int *out_from_header, size_t *out_start, size_t *out_len);

/* Thread-local message of the last failing call on this thread. Never NULL. */
const char *pwaf_last_error(void);
uint32_t pwaf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PWAF_H */
