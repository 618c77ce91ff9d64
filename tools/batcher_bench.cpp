// tools/batcher_bench.cpp — what a NATIVE per-request host (the reference's tokio workers, pingoo/listeners/http_listener.rs:206-264)
// sees through the deadline micro-batcher: N threads call pwaf_batcher_evaluate back to back, per-call latency is recorded.
// Measurement harness only (the Python figure was GIL-bound). Built by __graft_entry__.build() into
// tools/libbatcher_bench.so and driven by pingoo_amd.engine.native_batcher_latency; it calls the product through the C ABI entry point
// it is handed, nothing else.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

#include "../include/pwaf.h"

extern "C" {
typedef int (*evaluate_fn)(pwaf_batcher *, const pwaf_request *, pwaf_verdict *);

// Runs `threads` native threads, each issuing `per_thread` blocking calls over the request pool (round robin from a per-thread
// offset). lat_ms_out: threads * per_thread latencies in milliseconds. actions_out (optional): the action of every call, same order.
// Returns the number of failed calls; *seconds = wall time of the whole run.
int bb_run(void *fn, pwaf_batcher *b, const pwaf_request *reqs, size_t n_reqs, int threads, int per_thread, double *lat_ms_out, uint8_t *actions_out, double *seconds) {
    evaluate_fn eval = (evaluate_fn)fn;
    std::atomic<int> failed{0}, ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            ready++;
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (int j = 0; j < per_thread; j++) {
                const pwaf_request &r = reqs[((size_t)t * 7 + (size_t)j) % n_reqs];
                pwaf_verdict v{};
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = eval(b, &r, &v);
                const auto t1 = std::chrono::steady_clock::now();
                lat_ms_out[(size_t)t * per_thread + j] = std::chrono::duration<double, std::milli>(t1 - t0).count();
                if (actions_out) actions_out[(size_t)t * per_thread + j] = v.action;
                if (rc != PWAF_OK) failed++;
            }
        });
    while (ready.load() < threads) std::this_thread::yield();
    const auto w0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto &x : th) x.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    return failed.load();
}
}
