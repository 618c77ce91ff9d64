"""Host-side loaders for the files the reference builds its rule state from — what sits between
`/etc/pingoo` and `RuleEngine(rules, lists, geoip)`:

  load_rule_config(config_file, rules_folder)   pingoo.yml `rules:` / `lists:` sections + rules/*.yml   (config/config.rs:199-213,
                                                 255-269,378-421; config_file.rs:25-33,98-101)
  load_list(path, type)                          CSV list files                                          (lists.rs:62-117)
  load_geoip(paths)                              first existing geoip.mmdb                               (geoip.rs:43-72,94-110)

Only the rule path's inputs are read: listeners, services, TLS, service discovery are ignored (out of scope).
YAML parsing uses PyYAML (the reference uses serde_yaml); the MaxMind DB and CSV readers are the C++ ones in libpwaf.so.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import yaml

from . import _abi
from .engine import geoip_from_mmdb, parse_list_csv

DEFAULT_CONFIG_FILE = "/etc/pingoo/pingoo.yml"  # config/config.rs:25
DEFAULT_CONFIG_FOLDER = "/etc/pingoo"           # config/config.rs:26
GEOIP_DATABASE_PATHS = ("/etc/pingoo/geoip.mmdb", "/etc/pingoo/geoip.mmdb.zst", "/usr/share/pingoo/geoip.mmdb", "/usr/share/pingoo/geoip.mmdb.zst")  # :31-36

ACTIONS = {"block": _abi.RULE_ACTION_BLOCK, "captcha": _abi.RULE_ACTION_CAPTCHA}  # rules/rules.rs:30-36 (serde tag = "action", snake_case)
LIST_TYPES = {"String": _abi.LIST_STRING, "Int": _abi.LIST_INT, "Ip": _abi.LIST_IP}   # lists.rs:17-22


class ConfigError(Exception):
    pass


def _rules_from_mapping(mapping, where: str) -> List[Tuple[str, Optional[str], List[int]]]:
    if mapping is None:
        return []
    if not isinstance(mapping, dict):
        raise ConfigError(f"error parsing {where}: rules must be a map of name -> rule")
    out = []
    for name, body in mapping.items():  # (PyYAML keeps file order, like IndexMap)
        if not isinstance(body, dict) or "actions" not in body:
            raise ConfigError(f"error parsing {where}: rule {name}: missing field `actions`")
        expr = body.get("expression")
        if expr is not None and not isinstance(expr, str):
            raise ConfigError(f"error parsing {where}: rule {name}: expression must be a string")
        acts = []
        for a in body["actions"] or []:
            key = a.get("action") if isinstance(a, dict) else None  # internally tagged: `- action: block` (rules/rules.rs:30-36)
            if key not in ACTIONS:
                raise ConfigError(f"error parsing {where}: rule {name}: unknown action {a!r}")
            acts.append(ACTIONS[key])
        out.append((str(name), expr, acts))
    return out


def load_list(path: str, type_: str) -> Tuple[int, List[str]]:
    """-> (list type, items as text) for RuleEngine's `lists` argument; item syntax errors surface at engine creation with the
    line number, as the reference reports them at load time (lists.rs:90-111)."""
    if type_ not in LIST_TYPES:
        raise ConfigError(f"{type_} is not a valid ListType")  # lists.rs:45
    try:
        with open(path, "rb") as f:
            content = f.read()
    except OSError as err:
        raise ConfigError(f"error reading list {path}: {err}") from err
    return LIST_TYPES[type_], parse_list_csv(content)


def load_rule_config(config_file: str = DEFAULT_CONFIG_FILE, rules_folder: Optional[str] = None, folder_order: str = "sorted"):
    """-> (rules, lists): rules = [(name, expression | None, [action, ...])] in evaluation order (config file first, then the
    rules folder), lists = {name: (type, items)}.

    folder_order: "sorted" (default: file names in sorted order) or "read_dir" (whatever order the OS lists the folder in — what the
    reference does, pingoo/config/config.rs:383-404). First match wins makes the order of rule FILES part of the policy: when more
    than one file of the folder defines rules a warning says which order was taken, so that a deployment whose behaviour depended on
    its directory order notices the switch."""
    try:
        with open(config_file, "rb") as f:
            raw = yaml.safe_load(f) or {}
    except OSError as err:
        raise ConfigError(f"error reading config file ({config_file}): {err}") from err
    except yaml.YAMLError as err:
        raise ConfigError(f"error parsing config file ({config_file}): {err}") from err
    rules = _rules_from_mapping(raw.get("rules"), config_file)
    # the reference always reads DEFAULT_CONFIG_FOLDER/rules, whatever the config file's path (config.rs:381)
    folder = rules_folder if rules_folder is not None else os.path.join(DEFAULT_CONFIG_FOLDER, "rules")
    from_folder: List[Tuple[str, Optional[str], List[int]]] = []
    if os.path.isdir(folder):
        # The reference visits the folder in read_dir order (config.rs:383-404), which the OS does not define; first match wins makes
        # that order meaningful, so files are taken in sorted name order here — name them so that sorted order is the intended one.
        if folder_order not in ("sorted", "read_dir"):
            raise ConfigError(f"folder_order must be 'sorted' or 'read_dir', not {folder_order!r}")
        entries = [x for x in os.listdir(folder) if x.endswith(".yml")]  # (os.listdir = the directory's own order, like read_dir)
        if folder_order == "sorted":
            entries.sort()
        files_with_rules = []
        for entry in entries:
            p = os.path.join(folder, entry)
            try:
                with open(p, "rb") as f:
                    more = _rules_from_mapping(yaml.safe_load(f), p)
            except yaml.YAMLError as err:
                raise ConfigError(f"error parsing rules file {p!r}: {err}") from err
            seen = {r[0] for r in from_folder}
            for r in more:
                if r[0] in seen:
                    raise ConfigError(f"duplicate rule name: {r[0]}")
            from_folder.extend(more)
            if more:
                files_with_rules.append(entry)
        if len(files_with_rules) > 1:
            import warnings

            warnings.warn(f"rules folder {folder!r}: {len(files_with_rules)} files define rules; they are evaluated in {folder_order} order "
                          f"({', '.join(files_with_rules)}). The reference takes read_dir order (config.rs:383-404), which the OS does not define: "
                          "pass folder_order='read_dir' to keep that, or name the files so that sorted order is the intended one.", stacklevel=2)
    names = {r[0] for r in rules}
    for r in from_folder:
        if r[0] in names:
            raise ConfigError(f"duplicate rule name: {r[0]}")
    rules.extend(from_folder)
    lists: Dict[str, Tuple[int, List[str]]] = {}
    for name, lc in (raw.get("lists") or {}).items():
        if not isinstance(lc, dict) or "file" not in lc or "type" not in lc:
            raise ConfigError(f"error parsing config file ({config_file}): list {name}: `file` and `type` are required")
        lists[str(name)] = load_list(lc["file"], lc["type"])
    return rules, lists


def load_geoip(paths: Sequence[str] = GEOIP_DATABASE_PATHS):
    """First existing database of `paths` as a prefix table, or None (geoip.rs:94-110); a path ending in `.zst` is ZSTD-compressed
    (geoip.rs:49-55; libzstd is loaded at run time)."""
    for p in paths:
        if os.path.exists(p):
            with open(p, "rb") as f:
                content = f.read()
            try:
                return geoip_from_mmdb(content, p)
            except Exception as err:  # noqa: BLE001
                raise ConfigError(f"error loading geoip database ({p}): {err}") from err
    return None
