// batcher.cpp — deadline micro-batcher: the piece that lets a per-request caller use a batch engine.
//
// The reference evaluates rules inline, once per request, on whichever tokio worker owns the connection
// (pingoo/listeners/http_listener.rs:196-264). A drop-in `RuleEngine::evaluate(Request) -> Action` keeps that call shape, so
// concurrent callers have to be gathered into batches somewhere: here. Callers block in pwaf_batcher_evaluate; a dispatcher
// thread closes a batch when it is full or when its oldest request has waited max_delay_us, runs it through
// pwaf_evaluate_batch (the same kernels as everything else) and wakes the callers with their verdicts.
// Plain C++17 on top of the public C ABI: no device code, no access to engine internals.
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pwaf.h"

namespace pwaf {
int fail(int code, const std::string &msg);
}
using pwaf::fail;

namespace {

#ifdef PWAF_BATCHER_SYSTEM_CLOCK
// (the ThreadSanitizer build of tests/test_batcher_cpu.py: timed waits on the steady clock go through pthread_cond_clockwait, which
// this toolchain's sanitizer does not intercept — it then misses the wait's unlock / lock and reports every access around it)
using Clock = std::chrono::system_clock;
#else
using Clock = std::chrono::steady_clock;
#endif

// One batch: the callers that joined it share this. `done` is the only thing they wait on — a futex word of the batch's own, so a
// finished batch wakes ITS callers, all at once, and none of them needs a lock to read its verdict (the fields above `done` are
// written before it is set and never afterwards). Round 4's first form had one condition variable for every caller of the batcher
// under the batcher's mutex: each finished batch woke all 64 callers, one mutex hand-over at a time — most of the 0.46 ms a
// request took on a pipeline that answers a small batch in 0.15 ms.
struct Generation {
    std::vector<pwaf_verdict> verdicts;
    int status = PWAF_OK;
    std::string error;
    std::atomic<uint32_t> done{0};
    void wait_done() {
        for (int spin = 0; spin < 64 && !done.load(std::memory_order_acquire); spin++) __builtin_ia32_pause();
        while (!done.load(std::memory_order_acquire)) syscall(SYS_futex, reinterpret_cast<uint32_t *>(&done), FUTEX_WAIT_PRIVATE, 0u, nullptr, nullptr, 0);
    }
    void set_done() {
        done.store(1u, std::memory_order_release);
        syscall(SYS_futex, reinterpret_cast<uint32_t *>(&done), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    }
};
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the futex word");

struct Slot {  // a batch being filled (struct of arrays, exactly the pwaf_batch layout); columns: the 5 fields, then the engine's header columns
    std::vector<std::vector<uint8_t>> data;
    std::vector<std::vector<uint32_t>> offs;
    std::vector<uint8_t> ip, v6, flags;
    std::vector<uint16_t> port, country;
    std::vector<uint32_t> asn;
    uint32_t n = 0;
    Clock::time_point deadline, first;  // first: when the batch's oldest request arrived
    std::shared_ptr<Generation> gen = std::make_shared<Generation>();
    void reset(size_t n_cols) {
        data.assign(n_cols, {});
        offs.assign(n_cols, std::vector<uint32_t>(1, 0u));
        ip.clear(); v6.clear(); flags.clear(); port.clear(); country.clear(); asn.clear();
        n = 0;
        gen = std::make_shared<Generation>();
    }
};

}  // namespace

struct pwaf_batcher {
    pwaf_engine *engine = nullptr;
    size_t n_cols = PWAF_N_FIELDS;  // 5 + pwaf_engine_header_count(engine)
    uint32_t max_batch = 0;
    std::chrono::microseconds max_delay{0};
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    Slot slot[2];  // [0]: requests without GeoIP columns, [1]: requests that bring asn / country (a batch has them for all or none)
    bool stop = false;
    uint64_t n_batches = 0, n_requests = 0;
    std::atomic<uint32_t> active{0};  // callers that are placing a request or waiting for its answer (they leave without the lock)
    std::atomic<uint32_t> inside{0};  // callers anywhere inside pwaf_batcher_evaluate: destroy waits for them (their last access to *this)
    uint32_t in_flight = 0;  // requests of the batches being evaluated right now
    // Early close: callers BLOCK in pwaf_batcher_evaluate, so once every caller inside the call sits in a slot or in a batch under
    // evaluation, nobody else can join until somebody is answered — waiting out the deadline then only adds latency. Such a batch is
    // closed after a short gather window (an eighth of the deadline: woken callers re-enter over a few tens of microseconds), a batch
    // that others may still join at its deadline, as before.
    std::chrono::microseconds grace{0};
    // two dispatchers: while one waits for its batch on the device, the other closes and submits the next one (the engine's per-call
    // contexts let their copies and kernels overlap)
    std::thread worker[2];

    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            // which slot is due?
            int due = -1;
            Clock::time_point wake = Clock::time_point::max();
            const auto now = Clock::now();
            const bool all_here = slot[0].n + slot[1].n + in_flight >= active.load(std::memory_order_acquire);
            for (int s = 0; s < 2; s++) {
                if (slot[s].n == 0) continue;
                Clock::time_point close_at = slot[s].deadline;
                if (all_here) close_at = std::min(close_at, slot[s].first + grace);
                if (slot[s].n >= max_batch || close_at <= now || stop) { due = s; break; }
                if (close_at < wake) wake = close_at;
                // (callers leave — `active` falls — without the lock and their notification can slip in before this thread waits: a
                // pending batch is looked at again after a gather window at the latest, not at its deadline)
                if (!all_here && now + grace < wake) wake = now + grace;
            }
            if (due < 0) {
                if (stop) return;
                if (wake == Clock::time_point::max()) cv_work.wait(lk);
                else cv_work.wait_until(lk, wake);
                continue;
            }
            // the batch leaves its slot and a fresh one takes its place. Building the fresh slot allocates (columns, its Generation): done
            // FIRST and inside a try, so that a std::bad_alloc leaves the gathered batch where it is and is tried again — nothing may
            // escape a worker thread (std::terminate: ADVICE r4)
            std::unique_ptr<Slot> fresh;
            try {
                fresh.reset(new Slot());
                fresh->reset(n_cols);
            } catch (const std::exception &) {
                cv_work.wait_for(lk, std::chrono::milliseconds(1));
                continue;
            }
            Slot b = std::move(slot[due]);
            slot[due] = std::move(*fresh);
            in_flight += b.n;
            lk.unlock();
            // evaluate outside the lock: callers keep filling the next batch meanwhile
            pwaf_batch pb{};
            std::vector<pwaf_strcol> hcols(n_cols - PWAF_N_FIELDS);
            std::vector<uint32_t> hbytes(n_cols - PWAF_N_FIELDS);
            std::vector<pwaf_verdict> out;
            int rc = PWAF_OK;
            std::string err;
            try {
                pb.struct_size = sizeof pb;
                pb.n = b.n;
                pb.memory = PWAF_MEM_HOST;
                for (size_t f = 0; f < n_cols; f++) {
                    const uint32_t bytes = (uint32_t)b.data[f].size();
                    b.data[f].resize(b.data[f].size() + PWAF_ARENA_PAD, 0);  // (may throw: inside the worker's try block)
                    pwaf_strcol &c = f < PWAF_N_FIELDS ? pb.field[f] : hcols[f - PWAF_N_FIELDS];
                    c.data = b.data[f].data();
                    c.offsets = b.offs[f].data();
                    if (f >= PWAF_N_FIELDS) hbytes[f - PWAF_N_FIELDS] = bytes;
                }
                if (n_cols > PWAF_N_FIELDS) {
                    pb.n_headers = (uint32_t)(n_cols - PWAF_N_FIELDS);
                    pb.headers = hcols.data();
                    pb.header_bytes = hbytes.data();
                }
                pb.ip = b.ip.data();
                pb.ip_is_v6 = b.v6.data();
                pb.port = b.port.data();
                pb.flags = b.flags.data();
                if (due == 1) {
                    pb.asn = b.asn.data();
                    pb.country = b.country.data();
                }
                out.resize(b.n);
                rc = pwaf_evaluate_batch(engine, &pb, out.data(), nullptr);
                if (rc) err = pwaf_last_error();
            } catch (const std::exception &ex) {
                rc = PWAF_E_NOMEM;
                err = std::string("micro-batcher: ") + ex.what();
            }
            b.gen->verdicts = std::move(out);
            b.gen->status = rc;
            b.gen->error = err;
            lk.lock();
            n_batches++;  // (before the callers are woken: a caller that returns sees its batch in pwaf_batcher_stats)
            n_requests += b.n;
            in_flight -= b.n;
            lk.unlock();
            b.gen->set_done();  // (the batch's callers wake and leave on their own: no lock, no shared condition variable)
            cv_done.notify_all();  // (callers waiting for room in a full slot)
            lk.lock();
        }
    }
};

extern "C" {

int pwaf_batcher_create(pwaf_engine *engine, uint32_t max_batch, uint32_t max_delay_us, pwaf_batcher **out) {
    if (!engine || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    if (max_batch == 0) return fail(PWAF_E_INVALID_ARG, "max_batch must be at least 1");
    auto *b = new pwaf_batcher();
    b->engine = engine;
    b->max_batch = max_batch;
    b->max_delay = std::chrono::microseconds(max_delay_us);
    b->grace = std::chrono::microseconds(max_delay_us / 8);
    b->n_cols = PWAF_N_FIELDS + (size_t)pwaf_engine_header_count(engine);
    b->slot[0].reset(b->n_cols);
    b->slot[1].reset(b->n_cols);
    for (auto &w : b->worker) w = std::thread([b] { b->run(); });
    *out = b;
    return PWAF_OK;
}

int pwaf_batcher_evaluate(pwaf_batcher *b, const pwaf_request *r, pwaf_verdict *out) {
    if (!b || !r || !out) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    if (r->n_headers && !r->headers) return fail(PWAF_E_INVALID_ARG, "n_headers without a headers array");
    const size_t n_cols = b->n_cols;
    std::vector<const char *> ptr{r->host, r->url, r->path, r->method, r->user_agent};
    std::vector<uint32_t> len{r->host_len, r->url_len, r->path_len, r->method_len, r->user_agent_len};
    for (size_t k = 0; k + PWAF_N_FIELDS < n_cols; k++) {  // header values in the engine's header order; an absent one reads as ""
        const bool have = k < r->n_headers;
        ptr.push_back(have ? r->headers[k].data : nullptr);
        len.push_back(have ? r->headers[k].len : 0u);
    }
    for (size_t f = 0; f < n_cols; f++)
        if (len[f] && !ptr[f]) return fail(PWAF_E_INVALID_ARG, "NULL field with non-zero length");
    // what would fail the SHARED batch is refused here, for this caller only (pingoo/geoip.rs:128-142: two letters A-Z)
    if (r->has_geoip && (r->country[0] < 'A' || r->country[0] > 'Z' || r->country[1] < 'A' || r->country[1] > 'Z'))
        return fail(PWAF_E_BATCH, "country is not two letters A-Z (pingoo/geoip.rs:128-142)");
    std::shared_ptr<Generation> gen;
    uint32_t idx = 0;
    int rc = PWAF_OK;
    std::string emsg;
    // Counted BEFORE the mutex is touched and until nothing of *b is touched any more, on every way out: pwaf_batcher_destroy waits
    // for `inside` to reach zero before it deletes the object (a caller blocked on the mutex used to be invisible to it).
    struct Inside {
        std::atomic<uint32_t> &c;
        explicit Inside(std::atomic<uint32_t> &x) : c(x) { c.fetch_add(1, std::memory_order_acq_rel); }
        ~Inside() { c.fetch_sub(1, std::memory_order_acq_rel); }
    } inside_guard(b->inside);
    {
        std::unique_lock<std::mutex> lk(b->mu);
        if (b->stop) return fail(PWAF_E_INVALID_ARG, "batcher is shutting down");
        b->active.fetch_add(1, std::memory_order_acq_rel);
        try {
            // a full slot the dispatcher has not picked up yet: wait for it to be taken
            while (b->slot[r->has_geoip ? 1 : 0].n >= b->max_batch && !b->stop) {
                b->cv_work.notify_one();
                b->cv_done.wait_until(lk, Clock::now() + std::chrono::microseconds(50));
            }
            if (b->stop) {
                rc = PWAF_E_INVALID_ARG;
                emsg = "batcher is shutting down";
            } else {
                Slot &t = b->slot[r->has_geoip ? 1 : 0];
                for (size_t f = 0; f < n_cols && rc == PWAF_OK; f++)
                    if ((uint64_t)t.data[f].size() + len[f] > 0xFFFFFFF0ull) { rc = PWAF_E_BATCH; emsg = "batch field arena would exceed 4 GiB"; }
                if (rc == PWAF_OK) {
                    // Transactional append: every column first gets the CAPACITY it needs — the only step that can throw —
                    // and only then the request's values; a std::bad_alloc can no longer leave the shared batch's columns with
                    // different lengths (every later request of that batch would have been evaluated with shifted offsets).
                    // (capacity grows GEOMETRICALLY: libstdc++'s reserve(n) allocates exactly n, so asking for size + len per request
                    // reallocated and copied every column on every append — O(n^2) bytes per batch under the batcher's mutex)
                    auto grow = [](auto &v, size_t extra) {
                        const size_t need = v.size() + extra;
                        if (v.capacity() < need) v.reserve(std::max(need, 2 * v.capacity()));
                    };
                    for (size_t f = 0; f < n_cols; f++) {
                        grow(t.data[f], len[f] + PWAF_ARENA_PAD);
                        grow(t.offs[f], 1);
                    }
                    grow(t.ip, 16);
                    grow(t.v6, 1);
                    grow(t.flags, 1);
                    grow(t.port, 1);
                    if (r->has_geoip) {
                        grow(t.asn, 1);
                        grow(t.country, 1);
                    }
                    for (size_t f = 0; f < n_cols; f++) {
                        t.data[f].insert(t.data[f].end(), (const uint8_t *)ptr[f], (const uint8_t *)ptr[f] + len[f]);
                        t.offs[f].push_back((uint32_t)t.data[f].size());
                    }
                    t.ip.insert(t.ip.end(), r->ip, r->ip + 16);
                    t.v6.push_back(r->ip_is_v6);
                    t.flags.push_back(r->flags);
                    t.port.push_back(r->port);
                    if (r->has_geoip) {
                        t.asn.push_back(r->asn);
                        t.country.push_back((uint16_t)(r->country[0] | (r->country[1] << 8)));
                    }
                    idx = t.n++;
                    gen = t.gen;
                    if (idx == 0) {
                        t.first = Clock::now();
                        t.deadline = t.first + b->max_delay;
                    }
                    // the dispatcher is woken when this arrival starts a batch's clock, fills it, or completes "everyone is here"
                    // (a wake-up per arrival had a dispatcher thread competing with the callers for the lock 64 times per batch)
                    if (idx == 0 || t.n >= b->max_batch || b->slot[0].n + b->slot[1].n + b->in_flight >= b->active.load(std::memory_order_acquire))
                        b->cv_work.notify_one();
                }
            }
        } catch (const std::exception &ex) {  // (std::bad_alloc while appending: nothing may escape the C ABI)
            rc = PWAF_E_NOMEM;
            emsg = std::string("micro-batcher: ") + ex.what();
        }
    }
    // outside the lock: wait for THIS batch, read the verdict, leave
    int status = rc;
    if (rc == PWAF_OK && gen) {
        gen->wait_done();
        status = gen->status;
        if (status == PWAF_OK) *out = gen->verdicts[idx];
        else emsg = gen->error;
    }
    b->active.fetch_sub(1, std::memory_order_acq_rel);
    b->cv_work.notify_one();  // (a caller leaving can complete the "everyone is here" condition of the batch being gathered; nothing of *b is touched after this but the guard's counter)
    if (status != PWAF_OK) return fail(status, emsg);
    return PWAF_OK;
}

int pwaf_batcher_stats(pwaf_batcher *b, uint64_t *n_batches, uint64_t *n_requests) {
    if (!b) return fail(PWAF_E_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(b->mu);
    if (n_batches) *n_batches = b->n_batches;
    if (n_requests) *n_requests = b->n_requests;
    return PWAF_OK;
}

void pwaf_batcher_destroy(pwaf_batcher *b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;  // pending requests are still evaluated (the dispatcher drains both slots before it returns)
    }
    b->cv_work.notify_all();
    for (auto &w : b->worker)
        if (w.joinable()) w.join();
    {
        // callers still inside pwaf_batcher_evaluate (woken by their batch, or refused because of `stop`) leave before the object goes
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv_done.notify_all();
    }
    while (b->inside.load(std::memory_order_acquire) != 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    delete b;
}

}  // extern "C"
