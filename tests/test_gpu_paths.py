"""Targeted GPU parity cases for device-code paths the random fuzzers reach rarely: many comparison atoms (> 64: chunked), many scan
passes (> 12: beyond the verdict kernel's prefetched records), hit-record overflow chains, prefixes longer than /24 behind the DIR-24
table, and the verdict kernel variant that keeps the program tables in global memory."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi, geoip_entries
from pingoo_amd.engine import DeviceBatch, RuleEngine

pytestmark = pytest.mark.gpu
B, CAP = _abi.RULE_ACTION_BLOCK, _abi.RULE_ACTION_CAPTCHA
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(rules, lists, geo, reqs, label, **opts):
    eng = RuleEngine(rules, lists, geo, **opts)
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules, lists, geo).evaluate(batch, threads=8)
    got, counts = eng.evaluate_batch(batch, with_counts=True)
    H.assert_verdicts_equal(got, want, batch, label)
    assert counts.tolist() == np.bincount(want["action"], minlength=4).tolist()
    stats = eng.stats()
    eng.close()
    return want, stats


def test_more_than_64_comparison_atoms():
    rng = random.Random(1)
    rules = [(f"p{k}", f"client.remote_port == {1000 + 7 * k}", [B]) for k in range(90)]
    rules += [(f"l{k}", f"http_request.url.length() > {300 - k} && client.remote_port < {2000 + k}", [CAP]) for k in range(40)]
    rules += [("neg", "client.remote_port < -5 || client.remote_port == 4294967296 || http_request.path.length() <= -1", [B]), ("all", "client.remote_port <= 4294967296", [CAP])]
    reqs = [Request(remote_port=rng.choice([1000 + 7 * rng.randrange(95), rng.randrange(65536)]), url="/" + "u" * rng.randrange(400), captcha_verified=rng.random() < 0.3) for _ in range(2000)]
    want, stats = check(rules, {}, None, reqs, "many comparisons")
    assert stats["n_numeric_atoms"] > 130 and len(set(want["rule_idx"].tolist())) > 60


def test_more_than_12_scan_passes_and_overflowing_hit_records():
    rng = random.Random(2)
    words = ["".join(rng.choice("abcdefgh") for _ in range(rng.randint(3, 6))) for _ in range(260)]
    rules = [(f"w{k}", f'http_request.path.contains("{w}")', [B]) for k, w in enumerate(words[:200])]
    # rules that need SEVERAL atoms of the same pass: the request's hit record overflows into the pool chain
    rules += [(f"m{k}", " && ".join(f'http_request.url.contains("{w}")' for w in words[200 + 6 * k:206 + 6 * k]), [CAP, B]) for k in range(10)]
    reqs = []
    for _ in range(3000):
        p = "/" + "/".join(rng.choice(words + ["zzz"] * 300) for _ in range(rng.randint(0, 4)))
        k = rng.randrange(10)
        u = "/" + "-".join(rng.sample(words[200 + 6 * k:206 + 6 * k], rng.choice([6, 6, 5, 3, 0]))) + rng.choice(["", "?x=" + rng.choice(words)])
        reqs.append(Request(path=p, url=u, captcha_verified=rng.random() < 0.5))
    want, stats = check(rules, {}, None, reqs, "many passes", max_table_bytes=1024)
    assert stats["n_dfa_groups"] > 12, stats
    assert (want["action"] == _abi.ACTION_CAPTCHA).sum() > 50 and (want["action"] == _abi.ACTION_BLOCK).sum() > 50


def test_prefixes_longer_than_24_bits_and_ipv6_depth():
    rng = random.Random(3)
    geo_rows = [("10.0.0.0/8", 100, "AA"), ("10.1.2.0/24", 200, "BB"), ("10.1.2.128/25", 300, "CC"), ("10.1.2.200/30", 400, "DD"), ("10.1.2.201/32", 500, "EE"),
                ("2001:db8::/32", 600, "FF"), ("2001:db8:1:2::/64", 700, "GG"), ("2001:db8:1:2:3:4:5:0/112", 800, "HH"), ("2001:db8:1:2:3:4:5:6/128", 900, "II")]
    lists = {"v4": (_abi.LIST_IP, ["10.1.2.202/31", "10.1.2.129", "192.168.0.0/16", "192.168.7.7/32"]), "v6": (_abi.LIST_IP, ["2001:db8:1:2:3:4:5:6", "2001:db8:1:2:3::/80", "fe80::/10"])}
    rules = [(c, f'client.country == "{c}"', [B]) for _, _, c in geo_rows[::2]] + [("asn", "client.asn >= 300 && client.asn < 800", [CAP]), ("l4", 'lists["v4"].contains(client.ip)', [B]),
                                                                                    ("l6", 'lists["v6"].contains(client.ip)', [B]), ("xx", 'client.country == "XX"', [CAP])]
    v4 = ["10.1.2.%d" % k for k in range(120, 256)] + ["10.1.3.1", "10.9.9.9", "11.0.0.1", "192.168.7.7", "192.168.7.8", "127.0.0.1", "224.1.1.1"]
    v6 = ["2001:db8:1:2:3:4:5:%x" % k for k in range(0, 12)] + ["2001:db8:1:2:3:4:6:1", "2001:db8:1:2:3:5::1", "2001:db8:1:3::1", "2001:db9::1", "fe80::1", "::1", "ff02::1", "::10.1.2.201"]
    reqs = [Request(ip=rng.choice(v4 + v6), captcha_verified=rng.random() < 0.3) for _ in range(2000)]
    want, _ = check(rules, lists, geoip_entries(geo_rows), reqs, "deep prefixes")
    assert len(set(want["rule_idx"].tolist())) >= 8


def test_verdict_kernel_with_program_tables_in_global_memory():
    """The LT = false variant (programs whose tables do not fit LDS) on the fuzz cases, through PWAF_OPT_GLOBAL_VERDICT_TABLES."""
    code = """
import random, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import helpers as H
from oracle import pyoracle
from pingoo_amd import RequestBatch, _abi
from pingoo_amd.engine import RuleEngine, CompiledProgram, UnsupportedExpression
for seed in range(12):
    rng = random.Random(9000 + seed)
    lists = H.fuzz_lists(rng)
    geo = H.fuzz_geoip(rng)
    rules = []
    for k in range(rng.randint(3, 14)):
        e = H.rexpr(rng, lists)
        rules.append((f"r{k}", e, H.fuzz_actions(rng)))
    eng = RuleEngine(rules, lists, geo, flags=_abi.OPT_LENIENT | _abi.OPT_GLOBAL_VERDICT_TABLES)
    rules, _ = H.as_the_engine_sees(rules, eng.program)
    batch = RequestBatch.from_requests(H.fuzz_requests(rng, 500, seed %% 3 == 0))
    want = pyoracle.Oracle(rules, lists, geo).evaluate(batch)
    got, counts = eng.evaluate_batch(batch, with_counts=True)
    H.assert_verdicts_equal(got, want, batch, f"global tables, seed {seed}")
    assert counts.tolist() == np.bincount(want["action"], minlength=4).tolist()
    eng.close()
print("GLOBAL-TABLES-OK")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "GLOBAL-TABLES-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_service_routing_is_first_match_over_route_expressions():
    """http_listener.rs:266-271: the first service whose route matches (or that has no route) takes the request; none -> 404."""
    from pingoo_amd.engine import ServiceRouter

    rng = random.Random(4)
    routes = [("api", 'http_request.host.starts_with("api.")'), ("static", 'http_request.path.starts_with("/static/") || http_request.path.ends_with(".css")'),
              ("admin", 'http_request.host == "admin.example.com" && lists["office"].contains(client.ip)'), ("broken", "http_request.path"), ("v2", 'http_request.url.matches("^/v2/[0-9]+")')]
    lists = {"office": (_abi.LIST_IP, ["10.0.0.0/8"])}
    hosts = ["api.example.com", "www.example.com", "admin.example.com", "api", ""]
    paths = ["/static/a.js", "/x/y.css", "/v2/123/items", "/v2/abc", "", "/index.html"]
    reqs = [Request(host=rng.choice(hosts), path=(p := rng.choice(paths)), url=p or "/", ip=rng.choice(["10.1.1.1", "8.8.8.8"]), user_agent="") for _ in range(800)]
    batch = RequestBatch.from_requests(reqs)
    rules = [(n, e, [B]) for n, e in routes]
    flags = _abi.OPT_NO_UA_GATE | _abi.OPT_NO_CAPTCHA_BYPASS
    want = pyoracle.Oracle(rules, lists, None, flags=flags).evaluate(batch)
    expect = np.where(want["action"] == _abi.ACTION_BLOCK, want["rule_idx"].astype(np.int64), -1)
    router = ServiceRouter(routes, lists)
    got = router.route_batch(batch)
    assert (got == expect).all()
    assert set(got.tolist()) == {-1, 0, 1, 2, 4}  # "broken" (a non-bool route) never matches; an empty User-Agent is not gated here
    # with a catch-all service at the end nothing is unrouted
    r2 = ServiceRouter(routes + [("default", None)], lists)
    got2 = r2.route_batch(batch)
    assert (got2[got >= 0] == got[got >= 0]).all() and (got2[got < 0] == len(routes)).all()
    router.close()
    r2.close()


def test_node_api_two_engines_from_one_thread_and_concurrent_device_calls():
    """The single-process multi-device entry point (pwaf_node_*) with two replicas (both on device 0 here: the driver box has one GPU;
    what matters is two engines created from ONE thread — per-device kernel configuration — and slabs evaluated on two host threads),
    then two threads issuing pwaf_evaluate_device on two streams of one engine at the same time (per-call scratch contexts)."""
    import threading

    import torch

    from pingoo_amd.engine import NodeEngine
    from synth import pysynth

    w = pysynth.Workload(0)  # tiny mixed config incl. header fields
    batch = w.batch(0, 20001)
    # (+ two rules outside the column compiler's subset: every replica runs their specialized program, compiled once per process)
    rules = list(w.rules) + [("res_a", "http_request.path.length() * 2 > http_request.url.length() + 3 && client.remote_port % 5 == 0", [B]),
                             ("res_b", '(http_request.method + " " + http_request.path).starts_with("POST /a")', [CAP])]
    want = pyoracle.Oracle(rules, w.lists, w.geoip).evaluate(batch, threads=8)
    assert len({int(x) for x in want["rule_idx"]} & {len(rules) - 2, len(rules) - 1}) >= 1, "the residual rules never decide a request: weak test"
    node = NodeEngine(rules, w.lists, w.geoip, devices=[0, 0])
    assert node.n_devices == 2
    got, counts = node.evaluate_batch(batch, with_counts=True)
    H.assert_verdicts_equal(got, want, batch, "node, 2 replicas")
    assert counts.tolist() == np.bincount(want["action"], minlength=4).tolist()
    node.tune(w.batch(100000, 2000))
    H.assert_verdicts_equal(node.evaluate_batch(batch), want, batch, "node, tuned")
    # DEVICE-resident slabs (pwaf_node_evaluate_device): one slab per replica, already in HBM, enqueued by the node's persistent
    # per-device host threads; the per-device counters are accumulated on the device
    from pingoo_amd import shard
    bounds = [shard.shard_bounds(batch.n, r, 2) for r in range(2)]
    slabs = [DeviceBatch(batch.slice(lo, hi)) for lo, hi in bounds]
    outs_d = [torch.empty((hi - lo, 2), dtype=torch.int32, device="cuda") for lo, hi in bounds]
    cnts_d = [torch.zeros(4, dtype=torch.int64, device="cuda") for _ in bounds]
    streams_d = [torch.cuda.Stream() for _ in bounds]
    for _ in range(3):
        for c in cnts_d:
            c.zero_()
        torch.cuda.synchronize()
        node.evaluate_device(slabs, outs_d, cnts_d, [s.cuda_stream for s in streams_d])
        node.synchronize()
    got_d = np.concatenate([o.cpu().numpy().view(np.uint32) for o in outs_d])
    assert (got_d[:, 0] == want["action"]).all() and (got_d[:, 1] == want["rule_idx"]).all(), "node, device-resident slabs"
    assert (cnts_d[0] + cnts_d[1]).cpu().tolist() == np.bincount(want["action"], minlength=4).tolist()
    node.close()

    eng = RuleEngine(rules, w.lists, w.geoip)
    halves = [batch.slice(0, 10000), batch.slice(10000, 20001)]
    dbs = [DeviceBatch(h) for h in halves]
    outs, errs = [None, None], []

    def worker(k):
        try:
            s = torch.cuda.Stream()
            for _ in range(5):  # several in-flight calls per stream: the ring of contexts is reused under load
                outs[k] = eng.evaluate_device(dbs[k], stream=s.cuda_stream)
            s.synchronize()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    eng.device_status()
    for k, (lo, hi) in enumerate([(0, 10000), (10000, 20001)]):
        g = outs[k].cpu().numpy().view(np.uint32)
        assert (g[:, 0] == want["action"][lo:hi]).all() and (g[:, 1] == want["rule_idx"][lo:hi]).all(), f"stream {k}"
    eng.close()


def test_overflow_pool_exhaustion_is_retried_not_reported():
    """A hostile batch whose requests each match many patterns of one pass (more than the two inline hit-record slots) at a size that
    exhausts the default overflow pool: the synchronous entry point grows the pool and runs the batch again."""
    toks = ["aa1", "bb2", "cc3", "dd4", "ee5", "ff6", "gg7", "hh8", "ii9", "jj0", "kk1", "ll2"]
    rules = [(f"r{k}", f'http_request.url.contains("{t}") && http_request.path.length() > 1000', [B]) for k, t in enumerate(toks)]
    rules.append(("all", " && ".join(f'http_request.url.contains("{t}")' for t in toks), [CAP]))
    eng = RuleEngine(rules)
    n = 200_000  # pool = max(1M, 8 n) = 1.6M entries; every request needs 12
    batch = RequestBatch.from_requests([Request(url="/" + "-".join(toks), path="/p", host="h")]).tile(n)
    got = eng.evaluate_batch(batch)
    assert (got["action"] == 2).all() and (got["rule_idx"] == len(toks)).all()
    eng.close()


def test_a_retried_batch_counts_its_execution_errors_once():
    """pwaf_engine_rule_errors counts REQUESTS whose evaluation of a rule failed (the reference logs each, pingoo/rules.rs:41-45). A batch that
    exhausts the overflow pool is run again by the synchronous entry point: the second run's errors go to a sink, not to the counters."""
    toks = ["aa1", "bb2", "cc3", "dd4", "ee5", "ff6", "gg7", "hh8", "ii9", "jj0", "kk1", "ll2"]
    rules = [(f"r{k}", f'http_request.url.contains("{t}") && http_request.path.length() > 1000', [B]) for k, t in enumerate(toks)]
    rules.append(("div", "10 / (http_request.path.length() - 2) > 100", [B]))  # path "/p": a division by zero for every request, residual
    rules.append(("all", " && ".join(f'http_request.url.contains("{t}")' for t in toks), [CAP]))
    eng = RuleEngine(rules)
    n = 200_000
    batch = RequestBatch.from_requests([Request(url="/" + "-".join(toks), path="/p", host="h")]).tile(n)
    got = eng.evaluate_batch(batch)
    assert (got["action"] == 2).all() and (got["rule_idx"] == len(rules) - 1).all()
    errs = eng.rule_errors(len(rules))
    assert errs[len(toks)] == n and sum(errs) == n, errs[len(toks)]
    eng.close()


def test_field_against_field_predicates_and_per_rule_unsupported():
    """One request field against another on the device (pingoo/rules.rs:37-51 evaluates any expression), and a rule the device compiler
    cannot take (a bounded gap far beyond the DFA budget) failing alone: it is reported by index, never matches, the rest is unaffected."""
    rules = [("refl", "http_request.url.contains(http_request.host)", [B]),
             ("fwd", 'http_request.host == http_request.headers["x-forwarded-host"] && http_request.host != ""', [CAP]),
             ("len", "http_request.path.length() > http_request.url.length()", [B]),
             ("pre", "http_request.url.starts_with(http_request.path) && !http_request.url.ends_with(http_request.path)", [CAP]),
             ("gap", 'http_request.url.matches("select.{0,60}from.{0,60}where")', [B]),
             ("tail", 'http_request.user_agent.ends_with(http_request.method)', [B])]
    eng = RuleEngine(rules, flags=_abi.OPT_LENIENT)
    assert eng.partial
    prog = eng.program
    assert prog.unsupported_rules(len(rules)) == [4] and "budget" in prog.rule_status(4)[1]
    seen, bad = H.as_the_engine_sees(rules, prog, allow=1)
    assert bad == {4}
    rng = random.Random(3)
    words = ["a", "ex.com", "/p", "/p/q", "GET", "x", "", "select 1 from t where", "/p?host=ex.com"]
    reqs = [Request(host=rng.choice(words), url=rng.choice(words) + rng.choice(words), path=rng.choice(words), method=rng.choice(["GET", "POST"]),
                    user_agent=rng.choice(["curl GET", "xPOST", "Mozilla"]), headers={"x-forwarded-host": rng.choice(words)} if rng.random() < 0.6 else None)
            for _ in range(4000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(seen).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "field against field")
    assert len(set(want["rule_idx"].tolist())) >= 5
    with pytest.raises(Exception) as ei:  # the default: a rule nobody can evaluate fails creation, naming the rule
        RuleEngine(rules)
    assert ei.value.rule_index == 4
    eng.close()


def test_mmdb_database_and_config_files_drive_the_device_engine(tmp_path):
    """SURVEY §8f on the GPU: a (zstd-compressed) MaxMind DB image -> pwaf_geoip_from_file_image -> HIP engine, and a pingoo.yml + rules
    folder + CSV lists loaded by pingoo_amd.config -> HIP engine; verdicts against the oracle fed with the same decoded inputs."""
    import ipaddress

    from mmdb_writer import write_mmdb
    from pingoo_amd import config
    from test_loaders import zstd_compress

    rng = random.Random(77)
    nets = [("203.0.113.0/24", {"asn": "AS64500", "country": "NL"}), ("198.51.100.128/25", {"asn": "AS7", "country": "KP"}), ("10.0.0.0/8", {"asn": "AS1", "country": "US"}),
            ("2001:db8::/32", {"asn": "AS64501", "country": "FR"}), ("2001:db8:ff00::/40", {"asn": "AS9", "country": "KP"}), ("192.0.2.0/28", {"asn": "bogus", "country": "DE"}),
            ("100.64.0.0/10", {"asn": "AS5", "country": "zz"})]  # (a record whose country is not A-Z makes the lookup fall back to the default)
    (tmp_path / "geoip.mmdb.zst").write_bytes(zstd_compress(write_mmdb(nets, ip_version=6)))
    geo = config.load_geoip([str(tmp_path / "geoip.mmdb.zst")])
    (tmp_path / "rules").mkdir()
    (tmp_path / "ips.csv").write_text("203.0.113.7\n10.9.0.0/16, lab\n2001:db8:1::/48\n")
    (tmp_path / "pingoo.yml").write_text(f"""
lists:
  blocked: {{file: "{tmp_path}/ips.csv", type: Ip}}
rules:
  kp:
    expression: client.country == "KP"
    actions: [{{action: block}}]
  asn:
    expression: client.asn == 64500 || client.asn == 64501
    actions: [{{action: captcha}}]
""")
    (tmp_path / "rules" / "10-lists.yml").write_text('listed:\n  expression: lists["blocked"].contains(client.ip)\n  actions:\n    - action: block\n')
    (tmp_path / "rules" / "20-default.yml").write_text('unknown_country:\n  expression: client.country == "XX" && http_request.path.starts_with("/admin")\n  actions:\n    - action: block\n')
    rules, lists = config.load_rule_config(str(tmp_path / "pingoo.yml"), str(tmp_path / "rules"))
    assert [r[0] for r in rules] == ["kp", "asn", "listed", "unknown_country"]
    ips = ["203.0.113.7", "203.0.113.9", "198.51.100.129", "198.51.100.1", "10.9.1.1", "10.1.1.1", "2001:db8::1", "2001:db8:ff00::1", "2001:db8:1::5", "192.0.2.3", "100.64.1.1",
           "8.8.8.8", "::ffff:203.0.113.7", "127.0.0.1", "2001:db9::1"]
    reqs = [Request(ip=rng.choice(ips), path=rng.choice(["/admin", "/x"]), host="h", captcha_verified=rng.random() < 0.3) for _ in range(3000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules, lists, geo).evaluate(batch)
    eng = RuleEngine(rules, lists, geo)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "mmdb + config files")
    assert len(set(want["rule_idx"].tolist())) == 5  # every rule decides something, and some requests pass
    eng.close()


def test_counted_gap_patterns_on_the_device():
    """The counted-gap signatures of test_compiler (a.{0,n}b and friends) through the HIP engine: they are isolated into gated passes
    (their necessary prefix gates the confirmation scan); verdicts against the oracle, with other rules sharing the field."""
    from test_compiler import COUNTED_GAPS
    rules = [(f"g{k}", f'http_request.url.matches("{p}")', [B]) for k, p in enumerate(COUNTED_GAPS)]
    rules += [("lit", 'http_request.url.contains("zzzzzzzzzzz")', [CAP]), ("ua", 'http_request.user_agent.matches("(?i)sqlmap.{0,20}[0-9]")', [B])]
    eng = RuleEngine(rules)
    assert eng.program.unsupported_rules(len(rules)) == []
    rng = random.Random(5)
    pieces = ["select", "from", "where", "union", "a", "b", "c", "x", "ab", "<script", ">", " ", "\n", "=", "on", "load", "/p", ".php", "-" * 7, "q" * 13, "q", "A", "SeLeCt", "z" * 11, "z" * 29]
    reqs = [Request(url="".join(rng.choice(pieces) for _ in range(rng.randrange(0, 12))), path="/", host="h", user_agent=rng.choice(["sqlmap/1.7", "SQLMap " + "-" * 21 + "7", "curl"]))
            for _ in range(20000)]
    for n in (39, 40, 41, 42):
        reqs += [Request(url="select" + "y" * n + "from"), Request(url="a" + "y" * n + "b"), Request(url="a" + "y" * (n // 2) + "\n" + "y" * (n - n // 2 - 1) + "b")]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "counted gaps on the device")
    eng.tune(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "counted gaps on the device, tuned")
    eng.close()


def test_short_literal_atoms_are_answered_by_the_attribute_kernel():
    """A pass whose atoms are all anchored literals of <= 8 bytes (what rules ask of the method) is not walked: the attribute kernel compares
    the field's first 8 bytes (kernels.h: ShortAtom). Exact / prefix / empty / 8-byte literals, values longer than 8 bytes, an engine whose
    method pass also has a regex (then it stays a DFA pass), and the same rules with PWAF_OPT_NO_PREFILTER (DFA pass): all against the oracle."""
    short = [("post", 'http_request.method == "POST"', [B]), ("p", 'http_request.method.starts_with("P") && http_request.path.contains("x")', [CAP]),
             ("pf", 'http_request.method == "PROPFIND"', [B]), ("empty", 'http_request.method == ""', [CAP]), ("notget", '!(http_request.method == "GET") && http_request.url.contains("zz")', [B]),
             ("pre8", 'http_request.method.starts_with("PROPFIND")', [CAP]), ("any", 'http_request.method.starts_with("")  && http_request.host == "never"', [B])]
    methods = ["GET", "POST", "PUT", "PATCH", "PROPFIND", "PROPFINDX", "PROPFIN", "", "P", "post", "OPTIONS", "DELETE", "G", "GETT", "MKCALENDAR"]
    rng = random.Random(11)
    reqs = [Request(method=rng.choice(methods), url=rng.choice(["/", "/zz", "/a?zz=1"]), path=rng.choice(["/", "/x", "/ax"]), host=rng.choice(["h", "never"])) for _ in range(5000)]
    batch = RequestBatch.from_requests(reqs)
    for rules in (short, short + [("rx", 'http_request.method.matches("^(PUT|DEL)")', [B])], short + [("long", 'http_request.method == "MKCALENDAR"', [B])]):
        want = pyoracle.Oracle(rules).evaluate(batch)
        for flags in (0, _abi.OPT_NO_PREFILTER):
            eng = RuleEngine(rules, flags=flags)
            H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, f"short literals, {len(rules)} rules, flags {flags}")
            eng.tune(batch.slice(0, 2000))
            H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, f"short literals tuned, {len(rules)} rules, flags {flags}")
            eng.close()
    assert len(set(pyoracle.Oracle(short).evaluate(batch)["rule_idx"].tolist())) >= 6


def test_header_values_through_evaluate_one_and_the_micro_batcher():
    """ABI 2: pwaf_request carries header values (in the engine's header order), so rule sets over http_request.headers[...] give the
    same verdicts through evaluate(Request) and the deadline micro-batcher as through a batch (the call shape is
    http_listener.rs:206-264). The native harness (tools/batcher_bench.cpp) drives the batcher from std::threads."""
    import threading
    from pingoo_amd.engine import MicroBatcher, native_batcher_latency

    rules = [("tok", 'http_request.headers["x-token"].contains("evil") && http_request.method == "POST"', [B]),
             ("ref", 'http_request.headers["referer"].starts_with("http://spam.") || http_request.headers["cookie"].matches("sid=[0-9]{4}")', [CAP]),
             ("len", 'http_request.headers["x-token"].length() > 20', [B]),
             ("mix", 'http_request.headers["cookie"] + http_request.path == "a=1/p"', [B])]  # (a residual rule over a header column)
    eng = RuleEngine(rules)
    assert set(eng.header_names) == {"x-token", "referer", "cookie"}
    rng = random.Random(12)
    vals = ["", "evil", "xx evil yy", "http://spam.example", "sid=1234", "sid=12a4", "a=1", "t" * 25, "benign"]
    reqs = [Request(path=rng.choice(["/p", "/q"]), url="/p", host="h", method=rng.choice(["GET", "POST"]), user_agent="ua",
                    headers={k: rng.choice(vals) for k in ("x-token", "referer", "cookie", "unused") if rng.random() < 0.7} or None) for _ in range(600)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "headers, batch")
    assert len(set(want["rule_idx"].tolist())) >= 4
    for i in range(0, 60):
        v = eng.evaluate(reqs[i])
        assert int(v.decision) == int(want[i]["action"]) and (v.rule_idx is None or v.rule_idx == int(want[i]["rule_idx"])), (i, v)
    mb = MicroBatcher(eng, max_batch=64, max_delay_us=300)
    got = [None] * len(reqs)

    def caller(k):
        for i in range(k, len(reqs), 16):
            got[i] = mb.evaluate(reqs[i])
    th = [threading.Thread(target=caller, args=(k,)) for k in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    mb.close()
    for i, v in enumerate(got):
        assert int(v.decision) == int(want[i]["action"]), (i, reqs[i])
    stats = native_batcher_latency(eng, batch, threads=16, per_thread=40, max_batch=256, max_delay_us=200, pool=128)
    assert stats["failed"] == 0 and stats["requests"] == 640 and stats["latency_ms"]["p50"] > 0
    eng.close()


def test_rccl_allreduce_of_the_counters_on_a_one_device_communicator():
    """pwaf_node_allreduce_counts (csrc/node.cpp: librccl.so loaded at run time, ncclUint64 / ncclSum passed as plain integers, one
    RCCL group around the per-device calls) really runs: a communicator over device [0] from ncclCommInitAll, the counters of an
    evaluated batch all-reduced in place — over one rank the sum is the counters themselves."""
    import ctypes as C

    import torch

    from pingoo_amd.engine import NodeEngine

    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(name)
            break
        except OSError:
            continue
    assert rccl is not None, "librccl.so not found on a ROCm box"
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    comm = C.c_void_p()
    assert rccl.ncclCommInitAll(C.byref(comm), 1, (C.c_int * 1)(0)) == 0 and comm.value
    rules = [("a", 'http_request.path.starts_with("/.env")', [B]), ("b", 'client.remote_port < 1024', [CAP])]
    node = NodeEngine(rules, devices=[0])
    rng = random.Random(5)
    batch = RequestBatch.from_requests([Request(path=rng.choice(["/.env", "/x"]), remote_port=rng.choice([80, 5000]), host="h") for _ in range(5000)])
    want = pyoracle.Oracle(rules).evaluate(batch)
    hist = np.bincount(want["action"], minlength=4).tolist()
    out = torch.empty((batch.n, 2), dtype=torch.int32, device="cuda")
    counts = torch.zeros(4, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    node.evaluate_device([DeviceBatch(batch)], [out], [counts], [s.cuda_stream])
    node.allreduce_counts([comm.value], [counts], [s.cuda_stream])  # enqueued behind the batch on the same stream
    node.synchronize()
    s.synchronize()
    assert counts.cpu().tolist() == hist and hist[1] > 0 and hist[2] > 0
    node.close()
    assert rccl.ncclCommDestroy(comm) == 0


def test_lenient_node_keeps_its_engines_when_a_rule_is_refused():
    """PWAF_OPT_LENIENT makes pwaf_engine_create return PWAF_W_PARTIAL (+1) with an engine; pwaf_node_create used to treat
    that as a failure (and leak the engine). The node now exists, says `partial`, and the refused rule alone never matches."""
    from pingoo_amd.engine import NodeEngine, PwafError

    rules = [("dyn", 'http_request.path.matches(http_request.host)', [B]), ("ok", 'http_request.path == "/a"', [B])]  # (a regex compiled per request: refused)
    with pytest.raises(PwafError):
        NodeEngine(rules, devices=[0, 0])
    node = NodeEngine(rules, devices=[0, 0], flags=_abi.OPT_LENIENT)
    assert node.partial and node.n_devices == 2
    batch = RequestBatch.from_requests([Request(path="/a", host="h"), Request(path="/b", host="h")] * 100)
    got = node.evaluate_batch(batch)
    assert got["action"].tolist() == [1, 0] * 100 and set(got["rule_idx"][::2].tolist()) == {1}
    node.close()


def test_host_batches_small_and_large_pageable_and_page_locked():
    """pwaf_evaluate_batch's two staging paths (one packed page-locked block for small batches, column by column for large ones), from
    pageable memory and from page-locked columns (pwaf_host_register) into a page-locked result array (pwaf_host_alloc): the same
    verdicts as the oracle either way, and the same engine serves all of them in turn."""
    from pingoo_amd.engine import PinnedVerdicts
    rng = random.Random(99)
    lists = H.fuzz_lists(rng)
    rules = [(f"r{k}", H.rexpr(rng, lists), H.fuzz_actions(rng)) for k in range(12)]
    eng = RuleEngine(rules, lists, flags=_abi.OPT_LENIENT)
    seen, _ = H.as_the_engine_sees(rules, eng.program)
    orc = pyoracle.Oracle(seen, lists)
    pool = H.fuzz_requests(rng, 700, with_geo=True)
    for n in (1, 300, 60000):  # (60000 requests exceed the 1 MiB packed block: the column-by-column path)
        batch = RequestBatch.from_requests([pool[i % len(pool)] for i in range(n)])
        want = orc.evaluate(batch, threads=8)
        H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, f"pageable, n={n}")
        pv = PinnedVerdicts(n)
        batch.pin()
        try:
            got, counts = eng.evaluate_batch(batch, with_counts=True, out=pv.array)
            assert got is pv.array
            H.assert_verdicts_equal(got, want, batch, f"page-locked, n={n}")
            assert counts.tolist() == np.bincount(want["action"], minlength=4).tolist()
            H.assert_verdicts_equal(eng.evaluate_batch(batch, out=pv.array), want, batch, f"page-locked again, n={n}")
        finally:
            batch.unpin()
            pv.free()
    eng.close()


def test_verdict_sparse_column_file_dense_and_spill_agree():
    """The verdict kernel's ENTRY LIST (round 6: verdict2_kernel, the default) against the round-5 sparse column file
    (PWAF_OPT_SPARSE_VERDICT), the round-4 dense file (PWAF_OPT_DENSE_VERDICT) and both with 8 slots per wave
    (PWAF_OPT_TINY_VERDICT_SLOTS: every group spills what lies beyond the eighth entry / dirty column into global memory) — the 1k-rule
    set on benign, hostile and UTF-8 traffic, a rule set where most requests hit many atoms at once (overflowing hit records: the
    entry list's merge path), and the oracle on a prefix."""
    from synth import pysynth

    w = pysynth.Workload(3)
    engines = {name: RuleEngine(w.rules, w.lists, w.geoip, flags=fl) for name, fl in (("sparse", 0), ("dense", _abi.OPT_DENSE_VERDICT), ("tiny", _abi.OPT_TINY_VERDICT_SLOTS), ("file", _abi.OPT_SPARSE_VERDICT),
                                                                                    ("file_tiny", _abi.OPT_SPARSE_VERDICT | _abi.OPT_TINY_VERDICT_SLOTS))}
    orc = pyoracle.Oracle(w.rules, w.lists, w.geoip)
    for label, kw in (("benign", {}), ("hostile", {"adversarial": True}), ("utf8", {"utf8": True})):
        batch = w.batch(300_000, 200_000, **kw)
        got = {name: e.evaluate_batch(batch, with_counts=True) for name, e in engines.items()}
        for name in ("dense", "tiny", "file", "file_tiny"):
            H.assert_verdicts_equal(got[name][0], got["sparse"][0], batch, f"{label}: {name} vs the entry list")
            assert got[name][1].tolist() == got["sparse"][1].tolist()
        pre = batch.slice(0, 6000)
        H.assert_verdicts_equal(got["sparse"][0][:6000], orc.evaluate(pre, threads=16), pre, f"{label}: sparse vs oracle")
    for e in engines.values():
        e.close()
    # many atoms per request, many distinct dirty columns per group
    rng = random.Random(77)
    words = ["".join(rng.choice("abcdefgh") for _ in range(3)) for _ in range(300)]
    rules = [(f"r{k}", f'http_request.url.contains("{wd}") && http_request.path.contains("{words[(k * 7) % 300]}")', [B if k % 3 else CAP]) for k, wd in enumerate(words)]
    reqs = [Request(url="/" + "".join(rng.choice(words) for _ in range(rng.randint(0, 40))), path="/" + "".join(rng.choice(words) for _ in range(rng.randint(0, 12))), host="h",
                    captcha_verified=rng.random() < 0.3) for _ in range(20_000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch, threads=16)
    for fl in (0, _abi.OPT_DENSE_VERDICT, _abi.OPT_TINY_VERDICT_SLOTS, _abi.OPT_SPARSE_VERDICT, _abi.OPT_SPARSE_VERDICT | _abi.OPT_TINY_VERDICT_SLOTS):
        e = RuleEngine(rules, flags=fl)
        H.assert_verdicts_equal(e.evaluate_batch(batch), want, batch, f"many atoms per request, flags {fl}")
        e.close()
    assert len(set(want["action"].tolist())) >= 2


def test_field_against_field_overflow_runs_as_residual_rules_on_the_device():
    """48 rules of field-against-field predicates over 12 header columns: the device's table holds 32 predicates over 8 fields; the rest
    is lowered to residual programs (round 5; creation used to fail). Same verdicts as the oracle, nothing refused."""
    names = [f"x-h{k}" for k in range(12)]
    rules = [(f"r{k}", f'http_request.headers["{names[k % 12]}"] {["==", "!="][k % 2]} http_request.{["host", "path", "method", "url"][k % 4]}' +
              (f' && http_request.headers["{names[(k + 5) % 12]}"].contains(http_request.host)' if k % 3 == 0 else ""), [B if k % 2 else CAP]) for k in range(48)]
    eng = RuleEngine(rules)
    assert not eng.partial and 8 <= sum("residual" in w for w in eng.program.warnings()) < 48
    rng = random.Random(3)
    reqs = [Request(host=rng.choice(["a", "b", ""]), path="/" + rng.choice(["a", "b"]), url="/" + rng.choice(["a", "b"]), method=rng.choice(["a", "GET"]), user_agent="ua",
                    headers={nm: rng.choice(["a", "b", "/a", "/b", "GET", ""]) for nm in names if rng.random() < 0.7}) for _ in range(3000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, "field-against-field overflow on the device")
    assert len(set(want["rule_idx"].tolist())) >= 4
    eng.close()


@pytest.mark.parametrize("what", ["asn_comparisons", "header_lengths", "country_tables", "port_sets"])
def test_rule_sets_beyond_a_device_row_width_create_and_match_on_the_device(what):
    """tests/test_compiler.py: test_rule_sets_beyond_a_device_table_width_fall_to_residual_programs, on the device: the engine is CREATED
    (it used to refuse such a rule set as a whole) and gives the oracle's verdicts; the residual rules run in the interpreter kernel here
    (no hiprtc compile: the specialized form of the same programs is covered by tests/test_gpu_residual.py)."""
    rng = random.Random(17)
    if what == "asn_comparisons":
        rules = [(f"r{k}", f"client.asn == {1000 + k}", [B]) for k in range(200)]
    elif what == "header_lengths":
        rules = [(f"r{k}", f'http_request.headers["x-h{k}"].length() > {k % 4}', [B]) for k in range(12)]
    elif what == "country_tables":
        cc = [chr(65 + a) + chr(65 + b) for a in range(26) for b in range(26)]
        rules = [(f"r{k}", f'["{cc[k]}", "{cc[(7 * k + 3) % 676]}"].contains(client.country) && client.remote_port > 5', [B]) for k in range(300)]
    else:
        rules = [(f"r{k}", f"[{k + 2}, {k + 70000 % 60000}, 9].contains(client.remote_port)", [B]) for k in range(150)]
    eng = RuleEngine(rules, flags=_abi.OPT_NO_RESIDUAL_JIT)
    assert not eng.partial and eng.residual_mode == 1
    reqs = [Request(host="h", path="/", url="/", user_agent="ua", remote_port=rng.choice([2, 3, 9, 50, 150, 10000]), asn=rng.choice([1000, 1100, 1150, 1199, 5]),
                    country=rng.choice(["AA", "AD", "KX", "ZZ", "FR"]), headers={f"x-h{k}": "a" * rng.randint(0, 5) for k in range(12) if rng.random() < 0.6}) for _ in range(2000)]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules).evaluate(batch)
    H.assert_verdicts_equal(eng.evaluate_batch(batch), want, batch, what)
    assert len(set(want["rule_idx"].tolist())) >= 3
    eng.close()


def test_lazy_comparison_atoms_agree_with_eager_ones_and_the_oracle():
    """Round 6: a length / port comparison that is no rule's trigger is not evaluated per group by the attribute kernel any more — the verdict
    kernel evaluates it for the rules whose other literals already hold for somebody (program.h: LIT_LAZY). Rule sets that mix such atoms
    every way the DNF allows — beside a literal, negated, in several terms, under a conditional, beside a membership atom, alone (then the
    atom IS the trigger and stays eager), over header lengths — against the oracle and against PWAF_OPT_EAGER_CMP (round 5's evaluation)."""
    rng = random.Random(23)
    words = ["admin", "login", "wp-", ".php", "select", "etc/passwd", "api/v", "..%2f"] + ["".join(rng.choice("bcdfgklmnprstvz") for _ in range(5)) for _ in range(40)]
    neutral = ["".join(rng.choice("aeiouy") for _ in range(rng.randrange(1, 9))) for _ in range(300)] + ["index.html", "x" * 25]
    F = ["http_request.path", "http_request.url", "http_request.host", "http_request.user_agent", 'http_request.headers["x-tok"]']

    def cmp_atom():
        k = rng.randrange(6)
        if k == 0:
            return f"{rng.choice(F)}.length() {rng.choice(['>', '>=', '<', '<=', '==', '!='])} {rng.choice([0, 1, 5, 12, 20, 40, 64, 255, 256])}"
        if k == 1:
            return f"client.remote_port {rng.choice(['>', '>=', '<', '<=', '==', '!='])} {rng.choice([0, 80, 1024, 30000, 65535, 70000])}"
        if k == 2:
            return f"!({rng.choice(F)}.length() > {rng.choice([3, 10, 30])})"
        if k == 3:
            return f"client.asn {rng.choice(['==', '<', '>='])} {rng.choice([0, 64512, 15169])}"
        if k == 4:
            return f"({rng.choice(F)}.length() < 8 || client.remote_port % 2 == 0)"  # (a residual atom beside a lazy one)
        return f"{rng.choice(F)}.length() + 0 == {rng.choice([5, 12])}"

    def lit():
        return f'{rng.choice(F[:2])}.contains("{rng.choice(words)}")'

    rules = []
    for k in range(120):
        shape = rng.randrange(8)
        if shape == 0: e = f"{lit()} && {cmp_atom()}"
        elif shape == 1: e = f"({lit()} || {lit()}) && {cmp_atom()} && !({cmp_atom()})"
        elif shape == 2: e = f"{cmp_atom()} ? {lit()} : ({lit()} && {cmp_atom()})"
        elif shape == 3: e = f"{cmp_atom()} && {cmp_atom()} && {rng.choice(F[:2])}.length() == {rng.randrange(3, 60)}"  # no other literal: one of the atoms is the trigger
        elif shape == 4: e = f'lists["nets"].contains(client.ip) && {cmp_atom()}'
        elif shape == 5: e = f"({lit()} && {cmp_atom()}) || ({lit()} && {cmp_atom()}) || ({cmp_atom()} && {lit()} && {lit()})"
        elif shape == 6: e = f"!({lit()}) ? false : {cmp_atom()}"
        else: e = f'http_request.method == "POST" && {cmp_atom()} && {lit()}'
        rules.append((f"r{k}", e, [B if k % 3 else CAP]))
    lists = {"nets": (_abi.LIST_IP, ["10.0.0.0/8", "1.2.3.0/24", "2001:db8::/32"])}
    geo = geoip_entries([("8.8.8.0/24", 15169, "US"), ("5.5.0.0/16", 64512, "KP")])
    reqs = []
    for _ in range(30_000):
        segs = "/".join(rng.choice(words) if rng.random() < 0.12 else rng.choice(neutral) for _ in range(rng.randrange(1, 5)))
        reqs.append(Request(path="/" + segs, url="/" + segs + rng.choice(["", "?q=1", "?id=" + "9" * rng.randrange(0, 40)]), host=rng.choice(["a.b", "example.com", "h" * 20]),
                            method=rng.choice(["GET", "POST"]), user_agent=rng.choice(["Mozilla/5.0 (X11)", "curl/8", "x" * 64]), ip=rng.choice(["8.8.8.8", "10.1.2.3", "5.5.1.1", "9.9.9.9", "2001:db8::1"] + ["9.9.9.9"] * 20),
                            remote_port=rng.choice([80, 1023, 1024, 30000, 65535]), captcha_verified=rng.random() < 0.3, headers={"x-tok": "t" * rng.randrange(0, 16)} if rng.random() < 0.5 else None))
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules, lists, geo).evaluate(batch, threads=8)
    assert len(set(want["rule_idx"].tolist())) > 20
    for fl in (0, _abi.OPT_EAGER_CMP, _abi.OPT_TINY_VERDICT_SLOTS, _abi.OPT_SPARSE_VERDICT):
        eng = RuleEngine(rules, lists, geo, flags=fl)
        got, counts = eng.evaluate_batch(batch, with_counts=True)
        H.assert_verdicts_equal(got, want, batch, f"lazy comparison atoms, flags {fl}")
        assert counts.tolist() == np.bincount(want["action"], minlength=4).tolist()
        eng.close()
    # the batch supplies asn / country itself: client.asn comparisons are comparison atoms then (never lazy)
    for r in reqs[:5000]:
        r.asn, r.country = rng.choice([0, 64512, 15169, 7]), "US"
    b2 = RequestBatch.from_requests(reqs[:5000])
    eng = RuleEngine(rules, lists, geo)
    H.assert_verdicts_equal(eng.evaluate_batch(b2), pyoracle.Oracle(rules, lists, geo).evaluate(b2, threads=8), b2, "lazy comparison atoms, asn in the batch")
    eng.close()
