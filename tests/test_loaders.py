"""the data files on the caller's side of the rule path — MaxMind DB reader, list CSV parser, rule/list
config loader. CPU only (no kernels involved); the MMDB files come from the independent test-only writer in mmdb_writer.py."""
import ipaddress
import os
import random

import numpy as np
import pytest

import helpers as H
from mmdb_writer import write_mmdb
from oracle import pyoracle
from pingoo_amd import Request, RequestBatch, _abi, config
from pingoo_amd.engine import zstd_decompress, CompiledProgram, PwafError, geoip_from_mmdb, parse_list_csv
from table_walker import Tables

B = _abi.RULE_ACTION_BLOCK


def disjoint_networks(rng, version, count):
    """random partition cells of the address space: no network contains another (MMDB trees are flat)."""
    bits = 32 if version == 4 else 128
    cells = [(0, 0)]
    while len(cells) < count:
        k = rng.randrange(len(cells))
        v, l = cells.pop(k)
        if l >= (30 if version == 4 else 64):
            cells.append((v, l))
            continue
        cells += [(v, l + 1), (v | (1 << (bits - 1 - l)), l + 1)]
    rng.shuffle(cells)
    keep = [c for c in cells if c[1] > 0][: max(1, count * 2 // 3)]  # leave holes ("not found")
    mk = ipaddress.IPv4Network if version == 4 else ipaddress.IPv6Network
    return [str(mk((v, l))) for v, l in keep]


def table_set(t):
    return {(bytes(e["addr"][: 16 if e["is_v6"] else 4]), int(e["prefix_len"]), int(e["is_v6"]), int(e["asn"]), bytes(e["country"])) for e in t}


@pytest.mark.parametrize("ip_version,record_size,pointers", [(4, 24, False), (4, 28, True), (4, 32, False), (6, 24, True), (6, 28, False), (6, 32, True)])
def test_mmdb_reader_enumerates_every_network(ip_version, record_size, pointers):
    rng = random.Random(ip_version * 100 + record_size)
    nets = disjoint_networks(rng, 4, 120)
    if ip_version == 6:
        nets += [n for n in disjoint_networks(rng, 6, 80) if not ipaddress.ip_network(n).subnet_of(ipaddress.ip_network("::/96")) and
                 not ipaddress.ip_network("::/96").subnet_of(ipaddress.ip_network(n))]
    recs = [(n, {"asn": f"AS{rng.randrange(1, 400000)}", "country": rng.choice(["FR", "US", "DE", "CN", "BR"]), "extra": {"ignored": [1, 2, "x"]}}) for n in nets]
    got = table_set(geoip_from_mmdb(write_mmdb(recs, ip_version, record_size, pointers)))
    want = set()
    for n, r in recs:
        net = ipaddress.ip_network(n)
        asn, cc = int(r["asn"][2:]), r["country"].encode()
        if net.version == 4:
            want.add((net.network_address.packed, net.prefixlen, 0, asn, cc))
            if ip_version == 6:  # the same network as the IPv6 database stores it (::a.b.c.d/96+len)
                want.add((b"\0" * 12 + net.network_address.packed, net.prefixlen + 96, 1, asn, cc))
        else:
            want.add((net.network_address.packed, net.prefixlen, 1, asn, cc))
    assert got == want


def test_mmdb_records_are_read_like_the_reference_deserialises_them():
    """geoip.rs:17-23 + serde_utils.rs:1-9: asn is a STRING with optional AS prefixes, unparsable -> 0; country must be two upper-case
    letters; any other shape makes the reference's lookup fail, which the request path turns into {0, "XX"}."""
    cases = [({"asn": "AS64500", "country": "FR"}, (64500, b"FR")), ({"asn": "64501", "country": "US"}, (64501, b"US")),
             ({"asn": "ASAS77", "country": "DE"}, (77, b"DE")), ({"asn": "AS12AS3", "country": "DE"}, (0, b"DE")), ({"asn": "", "country": "DE"}, (0, b"DE")),
             ({"asn": "AS", "country": "DE"}, (0, b"DE")), ({"asn": "AS4294967296", "country": "DE"}, (0, b"DE")), ({"asn": "AS4294967295", "country": "DE"}, (4294967295, b"DE")),
             ({"asn": "AS-5", "country": "DE"}, (0, b"DE")), ({"asn": "+5", "country": "DE"}, (5, b"DE")), ({"asn": " 5", "country": "DE"}, (0, b"DE")),
             ({"asn": 64500, "country": "FR"}, (0, b"XX")), ({"asn": "AS1", "country": "fr"}, (0, b"XX")), ({"asn": "AS1", "country": "FRA"}, (0, b"XX")),
             ({"asn": "AS1"}, (0, b"XX")), ({"country": "FR"}, (0, b"XX")), ("not a map", (0, b"XX")), ({"asn": "AS1", "country": "F1"}, (0, b"XX")),
             ({"country": "JP", "z": 1.5, "asn": "AS9", "b": True, "n": -3, "big": 2 ** 40, "bytes": b"\x00\x01"}, (9, b"JP"))]
    recs = [(f"10.{k}.0.0/16", r) for k, (r, _) in enumerate(cases)]
    for pointers in (False, True):
        t = geoip_from_mmdb(write_mmdb(recs, 4, 24, pointers))
        by_net = {bytes(e["addr"][:4]): (int(e["asn"]), bytes(e["country"])) for e in t}
        for k, (_, want) in enumerate(cases):
            assert by_net[bytes([10, k, 0, 0])] == want, cases[k][0]


def test_mmdb_large_fields_and_rejects_garbage():
    long_name = "x" * 70000  # 3-byte extended size
    recs = [("192.0.2.0/24", {"asn": "AS1", "country": "FR", "note": long_name, "mid": "y" * 300, "arr": list(range(40))})]
    t = geoip_from_mmdb(write_mmdb(recs, 4, 28, True))
    assert table_set(t) == {(bytes([192, 0, 2, 0]), 24, 0, 1, b"FR")}
    for blob in (b"", b"\0" * 100, b"\xab\xcd\xefMaxMind.com", write_mmdb(recs)[:-3], b"junk" + b"\xab\xcd\xefMaxMind.com" + b"\xe0"):
        with pytest.raises(PwafError) as ei:
            geoip_from_mmdb(blob)
        assert ei.value.code == _abi.E_INVALID_ARG and "mmdb file is not valid" in ei.value.message
    assert len(geoip_from_mmdb(write_mmdb([], 4, 24))) == 0  # an empty tree is a valid database


def test_mmdb_table_drives_the_engine_like_the_original_table():
    rng = random.Random(99)
    nets = disjoint_networks(rng, 4, 200)
    rows = [(n, rng.randrange(1, 70000), rng.choice(["FR", "US", "CN", "RU"])) for n in nets]
    from pingoo_amd.batch import geoip_entries
    direct = geoip_entries(rows)
    via_mmdb = geoip_from_mmdb(write_mmdb([(n, {"asn": f"AS{a}", "country": c}) for n, a, c in rows], 6, 28, True))
    rules = [("cn", 'client.country == "CN"', [B]), ("asn", "client.asn < 20000 && client.asn > 0", [_abi.RULE_ACTION_CAPTCHA]), ("xx", 'client.country == "XX" && client.asn == 0', [B])]
    reqs = [Request(ip=str(ipaddress.IPv4Address(rng.randrange(2 ** 32)))) for _ in range(300)]
    reqs += [Request(ip="::" + str(ipaddress.IPv4Address(rng.randrange(2 ** 32)))) for _ in range(40)]  # IPv4-compatible IPv6: the ::/96 subtree
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules, {}, via_mmdb).evaluate(batch)
    t = Tables(CompiledProgram(rules, {}, via_mmdb))
    got = np.array([t.evaluate(batch, i) for i in range(batch.n)], dtype=[("action", np.uint8), ("rule_idx", np.uint32)])
    H.assert_verdicts_equal(got, want, batch, "mmdb table")
    # IPv4 requests see exactly what the directly built table gives
    v4 = RequestBatch.from_requests(reqs[:300])
    a = pyoracle.Oracle(rules, {}, direct).evaluate(v4)
    b = pyoracle.Oracle(rules, {}, via_mmdb).evaluate(v4)
    assert (a["action"] == b["action"]).all() and (a["rule_idx"] == b["rule_idx"]).all()
    assert len(set(a["action"].tolist())) >= 2


def test_list_csv_parser_follows_lists_rs():
    text = 'a\n  b  \r\n"c,d",comment\n\n"e ""q"""\n f,x\n10.0.0.0/8 , office\n'
    assert parse_list_csv(text) == ["a", "b", "c,d", 'e "q"', "f", "10.0.0.0/8"]
    assert parse_list_csv("") == [] and parse_list_csv("\n\n") == [] and parse_list_csv("last") == ["last"]
    for bad in ("a,b,c\n", '"open\n'):
        with pytest.raises(PwafError) as ei:
            parse_list_csv(bad)
        assert ei.value.code == _abi.E_LIST and "line 1" in ei.value.message
    with pytest.raises(PwafError) as ei:
        parse_list_csv("ok\nfine,x\nno,no,no\n")
    assert "line 3" in ei.value.message and "invalid number of columns" in ei.value.message


def test_rule_and_list_config_loader(tmp_path):
    (tmp_path / "rules").mkdir()
    (tmp_path / "blocked_ips.csv").write_text("10.0.0.0/8, office\n192.168.1.7\n2001:db8::/32\n")
    (tmp_path / "bad_ports.csv").write_text("23\n 2323 \n")
    (tmp_path / "pingoo.yml").write_text(f"""
listeners:
  http: {{address: "http://0.0.0.0:8080"}}
services: {{}}
lists:
  blocked_ips: {{file: "{tmp_path}/blocked_ips.csv", type: Ip}}
  bad_ports: {{file: "{tmp_path}/bad_ports.csv", type: Int}}
rules:
  block_listed:
    expression: lists["blocked_ips"].contains(client.ip)
    actions:
      - action: block
  telnet:
    expression: lists["bad_ports"].contains(client.remote_port)
    actions: [{{action: captcha}}, {{action: block}}]
""")
    (tmp_path / "rules" / "10-admin.yml").write_text('admin:\n  expression: http_request.path.starts_with("/admin")\n  actions:\n    - action: block\n')
    (tmp_path / "rules" / "20-all.yml").write_text("everything:\n  actions:\n    - action: captcha\n")
    (tmp_path / "rules" / "notes.txt").write_text("ignored: not a .yml file")
    # (the reference always reads /etc/pingoo/rules, config.rs:381: the folder next to a relocated config file is passed explicitly)
    with pytest.warns(UserWarning, match="2 files define rules.*sorted order"):  # first match wins: the order of rule FILES is policy
        rules, lists = config.load_rule_config(str(tmp_path / "pingoo.yml"), str(tmp_path / "rules"))
    assert [r[0] for r in rules] == ["block_listed", "telnet", "admin", "everything"]
    # the reference's own order (read_dir, config.rs:383-404) on request: same rules, in whatever order the OS lists the folder
    with pytest.warns(UserWarning, match="read_dir order"):
        rd, _ = config.load_rule_config(str(tmp_path / "pingoo.yml"), str(tmp_path / "rules"), folder_order="read_dir")
    assert [r[0] for r in rd][:2] == ["block_listed", "telnet"] and sorted(r[0] for r in rd[2:]) == ["admin", "everything"]
    want_rd = [x for x in os.listdir(tmp_path / "rules") if x.endswith(".yml")]
    assert [r[0] for r in rd[2:]] == [{"10-admin.yml": "admin", "20-all.yml": "everything"}[x] for x in want_rd]
    assert [r[0] for r in config.load_rule_config(str(tmp_path / "pingoo.yml"), str(tmp_path / "nowhere"))[0]] == ["block_listed", "telnet"]
    assert rules[1][2] == [_abi.RULE_ACTION_CAPTCHA, B] and rules[3][1] is None
    assert lists == {"blocked_ips": (_abi.LIST_IP, ["10.0.0.0/8", "192.168.1.7", "2001:db8::/32"]), "bad_ports": (_abi.LIST_INT, ["23", "2323"])}
    # the loaded configuration compiles and behaves
    reqs = [Request(ip="10.1.2.3"), Request(ip="8.8.8.8", remote_port=23), Request(ip="8.8.8.8", path="/admin/x"), Request(ip="8.8.8.8", captcha_verified=True),
            Request(ip="2001:db8::1")]
    batch = RequestBatch.from_requests(reqs)
    want = pyoracle.Oracle(rules, lists).evaluate(batch)
    t = Tables(CompiledProgram(rules, lists))
    got = [t.evaluate(batch, i) for i in range(batch.n)]
    assert [(int(a), int(r)) for a, r in got] == [(int(v["action"]), int(v["rule_idx"])) for v in want]
    assert [int(a) for a, _ in got] == [1, 2, 1, 0, 1]
    # duplicate names across the file and the folder, unknown actions, missing files
    (tmp_path / "rules" / "30-dup.yml").write_text("telnet:\n  actions: []\n")
    with pytest.raises(config.ConfigError, match="duplicate rule name: telnet"):
        config.load_rule_config(str(tmp_path / "pingoo.yml"), str(tmp_path / "rules"))
    os.remove(tmp_path / "rules" / "30-dup.yml")
    (tmp_path / "rules" / "30-bad.yml").write_text("x:\n  actions:\n    - action: allow\n")
    with pytest.raises(config.ConfigError, match="unknown action"):
        config.load_rule_config(str(tmp_path / "pingoo.yml"), str(tmp_path / "rules"))
    with pytest.raises(config.ConfigError, match="error reading config file"):
        config.load_rule_config(str(tmp_path / "missing.yml"))
    with pytest.raises(config.ConfigError, match="not a valid ListType"):
        config.load_list(str(tmp_path / "bad_ports.csv"), "Float")


def test_geoip_path_search(tmp_path):
    assert config.load_geoip([str(tmp_path / "a.mmdb"), str(tmp_path / "b.mmdb")]) is None
    (tmp_path / "b.mmdb").write_bytes(write_mmdb([("203.0.113.0/24", {"asn": "AS7", "country": "NL"})]))
    t = config.load_geoip([str(tmp_path / "a.mmdb"), str(tmp_path / "b.mmdb")])
    assert len(t) == 1 and int(t[0]["asn"]) == 7 and bytes(t[0]["country"]) == b"NL"
    # .zst databases (geoip.rs:49-55; the reference's Docker image ships geoip.mmdb.zst): decompressed through libzstd at run time
    raw = write_mmdb([("198.51.100.0/24", {"asn": "AS64500", "country": "SE"}), ("2001:db8::/32", {"asn": "AS9", "country": "FI"})], ip_version=6)
    (tmp_path / "a.mmdb.zst").write_bytes(zstd_compress(raw))
    t = config.load_geoip([str(tmp_path / "a.mmdb.zst"), str(tmp_path / "b.mmdb")])
    assert sorted((int(e["asn"]), bytes(e["country"])) for e in t) == sorted([(64500, b"SE"), (64500, b"SE"), (9, b"FI")])
    assert zstd_decompress(zstd_compress(raw) + zstd_compress(b"tail")) == raw + b"tail"  # frame after frame, like zstd::decode_all
    (tmp_path / "c.mmdb.zst").write_bytes(b"\x28\xb5\x2f\xfd")
    with pytest.raises(config.ConfigError, match="decompressing"):
        config.load_geoip([str(tmp_path / "c.mmdb.zst")])
    with pytest.raises(config.ConfigError, match="decompressing"):
        config.load_geoip([str(tmp_path / "b.mmdb").replace("b.mmdb", "d.mmdb.zst")] if (tmp_path / "d.mmdb.zst").write_bytes(raw) else [])


def zstd_compress(data: bytes) -> bytes:
    """TEST-ONLY: ZSTD_compress of the system libzstd (the product only ever decompresses)."""
    import ctypes

    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(buf, cap, data, len(data), 3)
    assert not z.ZSTD_isError(n)
    return buf.raw[:n]


@pytest.mark.parametrize("name", ["handmade_v4.mmdb", "handmade_v4_rs28.mmdb", "handmade_v4_rs32.mmdb"])
def test_mmdb_reader_against_a_hand_assembled_file(name):
    """tests/golden/handmade_v4*.mmdb were written byte by byte from the public MaxMind DB spec (tests/golden/make_handmade_mmdb.py cites
    the section per field), not by the repo's own writer — with 24-, 28- and 32-bit search-tree records (geoip.rs:57 reads whatever
    the file declares): 128.0.0.0/1 -> AS64500 FR, 64.0.0.0/3 -> AS15169 US, 96.0.0.0/3 -> a record whose country fails validation
    (-> the default record), 0.0.0.0/2 -> no data."""
    here = os.path.dirname(os.path.abspath(__file__))
    blob = open(os.path.join(here, "golden", name), "rb").read()
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_handmade_mmdb", os.path.join(here, "golden", "make_handmade_mmdb.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert blob == mod.FIXTURES[name], "the committed fixture is not what the script assembles"
    got = table_set(geoip_from_mmdb(blob))
    assert got == {(bytes([128, 0, 0, 0]), 1, 0, 64500, b"FR"), (bytes([64, 0, 0, 0]), 3, 0, 15169, b"US"), (bytes([96, 0, 0, 0]), 3, 0, 0, b"XX")}
    # and through the oracle's longest-prefix lookup: addresses inside / outside the three networks
    orc = pyoracle.Oracle([("r", "true", [B])], None, geoip_from_mmdb(blob))
    for ip, want in [("200.1.2.3", (64500, b"FR")), ("65.0.0.1", (15169, b"US")), ("100.0.0.1", (0, b"XX")), ("10.0.0.1", (0, b"XX")), ("127.0.0.1", (0, b"XX"))]:
        assert orc.geoip_lookup(ipaddress.ip_address(ip).packed + b"\0" * 12, False) == want, ip
