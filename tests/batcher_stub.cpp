// TEST-ONLY: the micro-batcher (pingoo_amd/csrc/batcher.cpp, the product source, compiled here with g++) over a STUB engine, so that its
// threading — who waits on what, who wakes whom, who may touch the object when — is exercised on the CPU, also under ThreadSanitizer.
// The stub's pwaf_evaluate_batch takes ~150 us (what a small batch costs on the device: tools/small_batch_timeline.py) and answers
// every request with a function of ITS OWN bytes, so a caller that is handed somebody else's verdict is caught.
// Built and run by tests/test_batcher_cpu.py: `batcher_stub <threads> <calls per thread> <deadline us> [max_batch]`.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/pwaf.h"

struct pwaf_engine { int unused; };
namespace pwaf {
int fail(int code, const std::string &) { return code; }
}
static std::atomic<uint64_t> g_batches{0}, g_requests{0};
static uint32_t mix(const uint8_t *p, uint32_t n, uint32_t seed) {
    uint32_t h = 2166136261u ^ seed;
    for (uint32_t i = 0; i < n; i++) h = (h ^ p[i]) * 16777619u;
    return h;
}
extern "C" {
uint32_t pwaf_engine_header_count(const pwaf_engine *) { return 0; }
const char *pwaf_last_error(void) { return "stub"; }
int pwaf_evaluate_batch(pwaf_engine *, const pwaf_batch *b, pwaf_verdict *out, pwaf_counts *) {
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(150);
    for (uint32_t i = 0; i < b->n; i++) {
        const pwaf_strcol &c = b->field[2];  // path
        const uint32_t h = mix(c.data + c.offsets[i], c.offsets[i + 1] - c.offsets[i], b->port[i]);
        out[i].action = (uint8_t)(h & 3u);
        out[i].rule_idx = h >> 2;
    }
    g_batches++;
    g_requests += b->n;
    while (std::chrono::steady_clock::now() < until) std::this_thread::yield();
    return PWAF_OK;
}
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 64, per = argc > 2 ? atoi(argv[2]) : 200, deadline = argc > 3 ? atoi(argv[3]) : 200;
    const int max_batch = argc > 4 ? atoi(argv[4]) : 4096;
    pwaf_engine eng{};
    pwaf_batcher *b = nullptr;
    if (pwaf_batcher_create(&eng, (uint32_t)max_batch, (uint32_t)deadline, &b) != PWAF_OK) return 2;
    std::vector<double> lat((size_t)threads * per);
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            for (int j = 0; j < per; j++) {
                char path[64];
                const int n = snprintf(path, sizeof path, "/t%d/call%d", t, j);
                pwaf_request r{};
                r.host = "h"; r.host_len = 1;
                r.url = path; r.url_len = (uint32_t)n;
                r.path = path; r.path_len = (uint32_t)n;
                r.method = "GET"; r.method_len = 3;
                r.user_agent = "ua"; r.user_agent_len = 2;
                r.port = (uint16_t)(t * 131 + j);
                r.has_geoip = (t & 1) ? 1 : 0;  // both slots are used
                r.country[0] = 'F'; r.country[1] = 'R';
                pwaf_verdict v{};
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = pwaf_batcher_evaluate(b, &r, &v);
                lat[(size_t)t * per + j] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                const uint32_t h = mix((const uint8_t *)path, (uint32_t)n, r.port);
                if (rc != PWAF_OK || v.action != (uint8_t)(h & 3u) || v.rule_idx != (h >> 2)) bad++;
            }
        });
    for (auto &x : th) x.join();
    uint64_t nb = 0, nr = 0;
    pwaf_batcher_stats(b, &nb, &nr);
    pwaf_batcher_destroy(b);
    std::sort(lat.begin(), lat.end());
    printf("{\"threads\": %d, \"requests\": %llu, \"batches\": %llu, \"bad\": %d, \"p50_us\": %.1f, \"p99_us\": %.1f, \"max_us\": %.1f}\n", threads, (unsigned long long)nr,
           (unsigned long long)nb, bad.load(), lat[lat.size() / 2], lat[lat.size() * 99 / 100], lat.back());
    return (bad.load() == 0 && nr == (uint64_t)threads * per && nb == g_batches.load()) ? 0 : 1;
}
