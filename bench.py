#!/usr/bin/env python3
"""bench.py — requests/s matched by the MI355X WAF batch matcher on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path (all scan kernels + the verdict kernel) over one batch of synthetic
requests that is already resident in HBM. Workload: BASELINE.json configs[2] — 10M requests x 1024 rules
(600 literal + 200 regex + 124 CIDR lists + 100 GeoIP/ASN rules, 600k-prefix GeoIP table) — the configuration the
metric ("1k-rule WAF") is quoted on; it fits one GPU. With N GPUs every rank evaluates its own 10M-request slab of
the same seeded stream (weak scaling); the only collective is the RCCL all-reduce of the four action counters.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline      HBM roofline of the dominant kernel (scan_kernel), from HIP events recorded on the launch stream
                during the timed steps; algorithmic bytes per SURVEY.md §8(d) / DESIGN.md §6
  cpu_baseline  the CPU oracle ("port": a restatement of the reference's interpreter loop, NOT the Rust binary)
                timed on this box's host cores over a bounded sample of the same request stream (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, help="synthetic config id (BASELINE.json configs[id-1]); default 3")
    ap.add_argument("--requests", type=int, default=0, help="requests per GPU (default: the config's batch size)")
    ap.add_argument("--lds-budget", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie-inclusive", action="store_true", help="also time a 1M-request HOST batch through pwaf_evaluate_batch (extra launches: keep it out of profiled runs)")
    ap.add_argument("--verbose", action="store_true", help="per-kernel timings on stderr")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    from pingoo_amd import shard
    from pingoo_amd.engine import DeviceBatch, RuleEngine
    from synth import pysynth

    rank, world, local = shard.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    shard.init_process_group()
    dev = torch.device("cuda", local)

    default_n = {1: 10_000, 2: 1_000_000, 3: 10_000_000, 5: 10_000_000}.get(args.config, 100_000)
    n = args.requests or default_n
    threads = max(1, (os.cpu_count() or 1) // world)

    t0 = time.time()
    wl = pysynth.Workload(args.config)
    batch = wl.batch(rank * n, n, threads=threads)  # this rank's slab of the global seeded request stream
    t_gen = time.time() - t0
    t0 = time.time()
    opts = {"lds_table_budget": args.lds_budget} if args.lds_budget else {}
    eng = RuleEngine(wl.rules, wl.lists, wl.geoip, **opts)
    # profile-guided LDS residency: the DFA rows kept in LDS are chosen from a traffic sample DISJOINT from the timed batch
    # (a deployment would sample live traffic); verdicts do not depend on it
    tune_n = 0 if os.environ.get("PWAF_BENCH_NO_TUNE") else 32768
    if tune_n:
        eng.tune(wl.batch(world * n + rank * tune_n, tune_n, threads=threads))
    t_compile = time.time() - t0
    stats = eng.stats()
    dbatch = DeviceBatch(batch, dev)
    if os.environ.get("PWAF_BENCH_FILL"):  # diagnostic only: constant field bytes => no DFA ever leaves its root (pure hot-loop time)
        for d in dbatch.data:
            d.fill_(int(os.environ["PWAF_BENCH_FILL"]))
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        counts.zero_()
        eng.evaluate_device(dbatch, out=out, counts=counts, stream=stream.cuda_stream)
        shard.allreduce_counts(counts)  # the path's only exchange: 4 counters over RCCL/xGMI

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    eng.set_profiling(not os.environ.get("PWAF_BENCH_NO_EVENTS"))  # HIP events around every kernel launch, on the launch stream, during the timed steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ktimes = eng.kernel_times()
    eng.set_profiling(False)

    total_requests = n * world * args.steps
    value = total_requests / elapsed
    final_counts = counts.cpu().tolist()

    result = {
        "metric": "requests/sec matched (whole node), 1k-rule WAF",
        "value": value,
        "unit": "requests/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE.json configs[{args.config - 1}]: {n} requests/GPU x {len(wl.rules)} rules "
                        f"({stats['n_scan_atoms']} string/regex predicates in {stats['n_dfa_groups']} DFA passes, {stats['n_filtered_groups']} of them behind a bigram prefilter, {stats['n_ip_lists']} CIDR lists, "
                        f"{0 if wl.geoip is None else len(wl.geoip)} GeoIP prefixes), seed 0x50494E47^{args.config}",
            "requests_per_gpu": n,
            "rules": len(wl.rules),
            "hot_rows": f"tuned on {tune_n} sample requests disjoint from the timed batch" if tune_n else "BFS order (untuned)",
            "parallelism": f"requests sharded over {world} GPU(s), tables replicated, RCCL all-reduce of 4 counters",
            "action_counts_allow_block_captcha_bypass": final_counts,
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (scan_kernel: the launches that stream the request bytes) ----
        field_bytes = dbatch.field_bytes
        # Launches that stream request bytes: filter_kernel (every pass behind a bigram prefilter, all fields in ONE launch) and
        # scan_kernel (passes whose DFA walks every request). Algorithmic bytes: every byte of a streamed field ONCE per launch
        # + its n+1 offsets. The roofline object describes whichever of the two takes more time per step.
        kinds = {"filter_kernel": [0.0, 0, 0], "scan_kernel": [0.0, 0, 0]}  # ms, launches, algorithmic bytes
        verdict_ms, attr_ms, other_ms = 0.0, 0.0, {}
        fnames = ["host", "url", "path", "method", "user_agent"]
        for name, ms, tag in ktimes:
            if name == "filter":
                k = kinds["filter_kernel"]
                k[0] += ms
                k[1] += 1
                k[2] += sum(field_bytes[f] + 4 * (n + 1) for f in range(5) if (tag >> f) & 1)
            elif name.startswith("scan_"):
                k = kinds["scan_kernel"]
                k[0] += ms
                k[1] += 1
                k[2] += field_bytes[tag] + 4 * (n + 1)
            elif name == "verdict":
                verdict_ms += ms
            elif name == "attr":
                attr_ms += ms  # side stream, beside the scans
            else:
                other_ms[name.split("_x")[0]] = other_ms.get(name.split("_x")[0], 0.0) + ms
        dom = max(kinds, key=lambda kk: kinds[kk][0])
        scan_ms, n_scan_launches, scan_alg = kinds[dom]
        stream_ms = sum(v[0] for v in kinds.values())
        if args.verbose:
            per = {}
            for name, ms, tag in ktimes:
                per.setdefault(name, []).append(ms)
            for name, v in per.items():
                print(f"  {name:<24} avg {sum(v) / len(v):8.3f} ms  x{len(v)}", file=sys.stderr)
            print("  " + json.dumps(stats), file=sys.stderr)
        scan_s = scan_ms / 1000.0
        achieved = scan_alg / scan_s / 1e9 if scan_s > 0 else 0.0
        pipeline_alg = dbatch.algorithmic_bytes * args.steps
        # HBM traffic cannot be counted from inside this process: it comes from the separate rocprofv3 --pmc passes over this very
        # command (tools/profile_round.sh), committed under profiles/ and only quoted for the workload they were measured on
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r2_traffic.json")
        if args.config == 3 and n == 10_000_000 and os.path.exists(tpath):
            try:
                tk = json.load(open(tpath))["kernels"]
                sk = [v for k, v in tk.items() if "::" + dom in k]  # one entry per template instantiation
                traffic = sum(sum(v["fetch_bytes"]) + sum(v["write_bytes"]) for v in sk) // max(1, sum(v["launches"] for v in sk))
                traffic_src = "profiles/r2_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 per MI355X_MICROARCH.md)"
            except Exception:  # a malformed profile file must not break the bench line
                traffic, traffic_src = None, None
        kernel_s = (stream_ms + verdict_ms + sum(other_ms.values())) / 1000.0  # (the attribute kernel runs beside these on a side stream)
        result["roofline"] = {
            "bound": "hbm",
            "kernel": dom,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,  # HBM bytes per scan launch from the committed rocprofv3 PMC passes (profiles/), or null
            "traffic_source": traffic_src,
            "alg_bytes_per_launch": scan_alg // max(1, n_scan_launches),
            "launches_per_step": n_scan_launches // max(1, args.steps),
            "avg_launch_ms": scan_ms / max(1, n_scan_launches),
            "alg_bytes_per_step": scan_alg // args.steps,
            "pipeline": {
                "alg_bytes_per_request": dbatch.algorithmic_bytes / n,
                "achieved": pipeline_alg / kernel_s / 1e9 if kernel_s > 0 else 0.0,
                "frac": (pipeline_alg / kernel_s / 1e9 / HBM_PEAK_GBS) if kernel_s > 0 else 0.0,
                "filter_ms_per_step": kinds["filter_kernel"][0] / args.steps,
                "scan_ms_per_step": kinds["scan_kernel"][0] / args.steps,
                "verdict_ms_per_step": verdict_ms / args.steps,
                "attr_ms_per_step_side_stream": attr_ms / args.steps,
                "other_ms_per_step": {k: v / args.steps for k, v in other_ms.items()},
            },
        }
        # ---- SURVEY §8(d) extras: the part's measured copy bandwidth, bytes per clock and CU, and the PCIe-inclusive rate ----
        try:
            src = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize(dev)
            copy_gbs = 8 * 2 * src.numel() / (e0.elapsed_time(e1) / 1e3) / 1e9  # read + write traffic of a device-to-device copy
            del src, dst
            props = torch.cuda.get_device_properties(dev)
            result["roofline"]["peak_measured_copy_gbs"] = copy_gbs
            result["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
            # (2.4 GHz: the MI355X peak engine clock of MI355X_MICROARCH.md; torch does not expose the running clock)
            result["roofline"]["bytes_per_clk_per_cu_at_2p4ghz"] = achieved * 1e9 / (props.multi_processor_count * 2.4e9)
        except Exception as exc:  # informational only
            result["roofline"]["peak_measured_copy_gbs"] = None
            print(f"copy-bandwidth probe failed: {exc}", file=sys.stderr)
        if world == 1 and args.pcie_inclusive:
            # host batch in, verdicts out through pwaf_evaluate_batch: H2D + kernels + D2H, on a bounded slice of the same batch
            m = min(n, 1_000_000)
            hb = batch.slice(0, m) if m < n else batch
            eng.evaluate_batch(hb)
            t0 = time.perf_counter()
            hv = eng.evaluate_batch(hb)
            dt = time.perf_counter() - t0
            gv = out[:m].cpu().numpy().view(np.uint32)
            result["pcie_inclusive"] = {"value": m / dt, "unit": "requests/s", "sample": f"{m} requests from host memory, synchronous pwaf_evaluate_batch (H2D of ~324 B/request, kernels, D2H of 8 B/request)",
                                        "verdicts_match_device_resident_run": bool((hv["action"] == gv[:, 0]).all() and (hv["rule_idx"] == gv[:, 1]).all())}
        # ---- CPU baseline: the oracle (port of the reference's per-request interpreter loop) on host cores ----
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle

            cores = os.cpu_count() or 1
            orc = pyoracle.Oracle(wl.rules, wl.lists, wl.geoip)
            probe = wl.batch(0, 2000, threads=cores)
            t0 = time.perf_counter()
            orc.evaluate(probe, threads=cores)
            rate = 2000 / max(1e-6, time.perf_counter() - t0)
            sample_n = int(min(n, max(2000, rate * args.cpu_seconds)))
            sample = batch.slice(0, sample_n) if sample_n < n else batch
            t0 = time.perf_counter()
            cpu_v = orc.evaluate(sample, threads=cores)
            cpu_t = time.perf_counter() - t0
            gpu_v = out[:sample_n].cpu().numpy().view(np.uint32)
            same = bool((gpu_v[:, 0] == cpu_v["action"]).all() and (gpu_v[:, 1] == cpu_v["rule_idx"]).all())
            result["cpu_baseline"] = {
                "value": sample_n / cpu_t,
                "unit": "requests/s",
                "cores": cores,
                "kind": "port",
                "sample": f"first {sample_n} requests of the same batch, same rules; CPU restatement of the reference's interpreter loop "
                          f"(oracle/), {cores} threads; verdicts {'identical to' if same else 'DIFFERENT from'} the GPU's",
                "verdicts_match_gpu": same,
            }
        result["timing_notes"] = {"generate_s": round(t_gen, 2), "compile_and_upload_tables_s": round(t_compile, 2)}
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
