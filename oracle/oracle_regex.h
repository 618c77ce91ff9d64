// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product
// path (pingoo_amd/, libpwaf.so). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it, and only as the checker.
//
// oracle_regex.h — regex is_match for the oracle, restating the semantics of the `regex` crate
// 1.12.2 that the reference's `bel` interpreter depends on (Cargo.lock:1694-1700; SURVEY.md §7
// "Regex parity"). The crate's source is NOT under /root/reference (un-vendored dependency), so this
// follows its published syntax/semantics for the ASCII/byte subset documented in DESIGN.md §3:
// leftmost-first search reduces to plain existence for is_match, so a Pike-VM thread-set
// simulation is exact. PARITY UNPINNED by the reference itself (it has no tests);
// tests/test_oracle_regex.py pins this file against CPython's `re` on the shared syntax subset.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

namespace oracle {

struct RegexProg;  // Pike-VM program

struct Regex {
    std::shared_ptr<RegexProg> prog;
    // Compile; on failure returns false and sets err. `unsupported` distinguishes syntax this
    // restatement deliberately does not cover (still an error for the oracle).
    static bool compile(std::string_view pattern, Regex &out, std::string &err);
    bool is_match(std::string_view haystack) const;
};

}  // namespace oracle
