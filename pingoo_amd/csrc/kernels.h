// kernels.h — argument blocks and launch entry points of the gfx950 kernels (kernels.hip).
#pragma once
#include <cstdint>

#include "program.h"

namespace pwaf {

// Device copy of one DFA group's tables.
// tab: n_states rows of (n_classes + 2) uint16: [0, n_classes) = next state PRE-MULTIPLIED by the row
// stride, [n_classes] = 1 + end-list id (0 = none), [n_classes + 1] = 1 + emit-list id (0 = none).
struct ScanArgs {
    const uint8_t *data;      // field arena
    const uint32_t *off;      // n + 1 offsets
    uint32_t n;
    uint32_t n_groups;        // ceil(n / 64): one group = 64 consecutive requests = one bit column word
    const uint16_t *tab;      // global copy of the table (staged into LDS by every block)
    const uint8_t *classmap;  // 256 bytes
    const uint32_t *list_off; // shared by end- and emit-lists
    const uint16_t *list;     // local atom ids
    uint32_t n_states, stride /* n_classes + 2 */, n_classes;
    uint32_t first_emit_pm;   // first_emit * stride
    uint32_t start_pm;        // start * stride
    uint32_t n_local;         // columns owned by this group (multiple of 64)
    uint32_t col_rel;         // atom_base - scan_base (multiple of 64)
    uint32_t scan_cols;       // total scan columns (row length of M)
    uint32_t scan_words;      // scan_cols / 64 (row length of S)
    uint64_t *S;              // [n_groups][scan_words]  bit a%64 of word a/64: column a has a hit in this group
    uint64_t *M;              // [n_groups][scan_cols]   64-request hit masks; valid only where S says so
};

struct VerdictArgs {
    uint32_t n, n_groups;
    const uint32_t *off[PWAF_N_FIELDS];
    const uint8_t *ip;
    const uint8_t *ip_is_v6;
    const uint16_t *port;
    const uint8_t *flags;
    const uint32_t *asn;      // nullable
    const uint16_t *country;  // nullable
    // compiled program
    uint32_t n_cols, scan_base, scan_cols, scan_words;
    const uint64_t *S;
    const uint64_t *M;
    const NumAtomDev *num_atoms;
    uint32_t n_num_atoms;
    const int64_t *int_pool;
    const uint32_t *country_luts;  // 22 words per lut
    const DevRule *rules;
    uint32_t n_rules;
    const uint32_t *lits;
    // tries
    const uint32_t *ip_root4, *ip_root6, *ip_nodes;  // membership sets (null roots => set 0)
    const uint32_t *set_masks;
    uint32_t set_words;
    uint32_t n_ip_lists;
    const uint32_t *geo_root4, *geo_root6, *geo_nodes;
    const GeoRec *geo_recs;
    uint32_t has_geo;
    // outputs
    pwaf_verdict *out;
    unsigned long long *counts;  // 4, accumulated (nullable)
    uint32_t *match_idx;         // nullable
    uint32_t *n_matches;         // nullable
};

// Launchers (hipStream_t passed as void*). Return hipError_t as int.
int launch_scan(const ScanArgs &a, void *stream);
int launch_verdict(const VerdictArgs &a, void *stream);
// LDS bytes the scan kernel needs for a group with these dimensions (table + classmap + wave matrices)
uint32_t scan_lds_bytes(uint32_t n_states, uint32_t stride, uint32_t n_local);
uint32_t verdict_lds_bytes(uint32_t n_cols);

}  // namespace pwaf
