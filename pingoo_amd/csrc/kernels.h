// kernels.h — argument blocks and launch entry points of the gfx950 kernels (kernels.hip).
#pragma once
#include <cstdint>

#include "program.h"

namespace pwaf {

static constexpr int kMaxPasses = 64;

// Hit record of one (scan pass, request): what the request's field matched in that pass's DFA.
//   bit 31 = 0: bits [14:0] = first local atom + 1 (0 = none), bits [29:15] = second local atom + 1 (0 = none)
//   bit 31 = 1: bits [30:0] = head of a chain in the overflow pool holding ALL atoms of the request for this pass
static constexpr uint32_t REC_OVERFLOW = 0x80000000u;
struct PoolEntry {
    uint32_t atom;  // local atom id
    uint32_t next;  // 0xFFFFFFFF = end of chain
};

// Device table of one DFA group: n_states rows of stride = n_classes + 2 uint16:
//   [0, n_classes)  next state id | 0x8000 when the next state has an emit list
//   [n_classes]     1 + end-list id (0 = none)        [n_classes + 1]  1 + emit-list id (0 = none)
// Rows [0, n_hot) are staged into LDS by every workgroup; colder rows are read from this (L2-resident) copy.
struct ScanArgs {
    const uint8_t *data;      // field arena
    const uint32_t *off;      // n + 1 offsets
    uint32_t n;
    const uint16_t *tab;
    const uint8_t *classmap;  // 256 bytes
    const uint32_t *list_off; // shared by end- and emit-lists
    const uint16_t *list;     // local atom ids
    uint32_t n_states, stride, n_classes, n_hot;
    uint32_t *rec;            // n hit records of this pass
    PoolEntry *pool;
    uint32_t *pool_count;     // atomic allocator
    uint32_t pool_cap;
    uint32_t *status;         // device status word: bit 0 = overflow pool exhausted
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
    // scan results
    uint32_t n_passes;
    const uint32_t *rec;        // [n_passes][n]
    const uint32_t *pass_base;  // first column of each pass
    const PoolEntry *pool;
    // compiled program
    uint32_t n_cols;
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
uint32_t scan_lds_bytes(uint32_t n_hot, uint32_t stride);
uint32_t verdict_lds_bytes(uint32_t n_cols);

}  // namespace pwaf
