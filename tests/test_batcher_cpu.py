"""The deadline micro-batcher's threading (pingoo_amd/csrc/batcher.cpp: the product source) on the CPU, over a STUB engine whose
pwaf_evaluate_batch takes what a small batch takes on the device (~150 us) and answers every request with a function of its own bytes
(tests/batcher_stub.cpp). What the reference does inline on the connection's worker (http_listener.rs:239-264), per request, this
serves from many blocked callers at once: every caller must get ITS verdict, nothing may race (ThreadSanitizer build), a full slot and
a tiny max_batch must not deadlock, and the per-request latency of the model is reported (the device adds nothing the stub does not
model: tools/small_batch_timeline.py measured 145 us per 32-request call)."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BUILD = os.path.join(HERE, "_build")
SRCS = [os.path.join(HERE, "batcher_stub.cpp"), os.path.join(ROOT, "pingoo_amd", "csrc", "batcher.cpp")]


def build(name, *flags):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, name)
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in SRCS):
        subprocess.run(["g++", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), *flags, *SRCS, "-o", out], check=True)
    return out


def run(exe, *args, timeout=300):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout)
    return r, (json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else None)


def test_every_caller_gets_its_own_verdict_and_latency_of_the_model():
    exe = build("batcher_stub", "-O2")
    r, out = run(exe, 64, 200, 200)
    assert r.returncode == 0 and out["bad"] == 0 and out["requests"] == 64 * 200, (r.returncode, out, r.stderr[-500:])
    assert out["batches"] < out["requests"] / 4, out  # 64 native callers really are batched
    print("micro-batcher over the stub engine (150 us per batch):", out)
    # not a timing assertion a loaded CI host could fail on, only a sanity bound: a request is answered within a few batch times
    # (a host running a dozen other processes measured 8 ms: the default bound only catches a hang; PWAF_TEST_QUIET_HOST=1 keeps the tight one)
    assert out["p50_us"] < (5000 if os.environ.get("PWAF_TEST_QUIET_HOST") else 50000), out


@pytest.mark.parametrize("threads,per,deadline,max_batch", [(16, 100, 200, 4), (8, 50, 50, 1), (3, 200, 1000, 4096), (1, 50, 200, 4096)])
def test_full_slots_tiny_batches_and_few_callers(threads, per, deadline, max_batch):
    exe = build("batcher_stub", "-O2")
    r, out = run(exe, threads, per, deadline, max_batch)
    assert r.returncode == 0 and out["bad"] == 0 and out["requests"] == threads * per, (r.returncode, out, r.stderr[-500:])


def test_no_data_race_under_thread_sanitizer():
    exe = build("batcher_stub_tsan", "-O1", "-g", "-fsanitize=thread", "-DPWAF_BATCHER_SYSTEM_CLOCK")
    for args in ((32, 60, 200), (12, 60, 200, 3)):
        r, out = run(exe, *args, timeout=600)
        if "unexpected memory mapping" in r.stderr:  # (the sanitizer runtime cannot start under this kernel's address-space layout: nothing was tested)
            pytest.skip("ThreadSanitizer cannot run here: " + r.stderr.strip().splitlines()[0])
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0 and out["bad"] == 0, (r.returncode, out)
