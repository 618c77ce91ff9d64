"""Builds libpwaf.so (host compiler + gfx950 kernels + C ABI) in-tree with hipcc.

`python -m pingoo_amd.build` or `pingoo_amd.build.build()`. The .so is git-ignored but travels to the
GPU box with the repo snapshot. hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpwaf.so")
SOURCES = ["frontend.cpp", "pattern.cpp", "dfa.cpp", "iptrie.cpp", "filter.cpp", "residual.cpp", "residual_jit.cpp", "rtc.cpp", "compile.cpp", "loaders.cpp", "batcher.cpp", "node.cpp", "engine.cpp", "kernels.hip"]
HEADERS = ["frontend.h", "program.h", "kernels.h", "residual.h", "confirm.h", "unicode_data.inc", os.path.join("..", "..", "include", "pwaf.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build pingoo_amd/libpwaf.so)")


EMBED = os.path.join(CSRC, "residual_h.inc")


def embed() -> str:
    """csrc/residual_h.inc: residual.h as one raw string literal. residual_jit.cpp puts it in front of the specialized residual
    program it hands to hiprtc when an engine is created (the device has no include path to find the header in)."""
    text = open(os.path.join(CSRC, "residual.h")).read()
    assert ')RVMH"' not in text
    out = 'R"RVMH(' + text + ')RVMH"\n'
    if not os.path.exists(EMBED) or open(EMBED).read() != out:
        with open(EMBED, "w") as f:
            f.write(out)
    return EMBED


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    """variant "prof": libpwaf_prof.so, the -DPWAF_PROFILING build (timing-experiment switches for tools/*.sh; never the product).
    variant "asan": libpwaf_asan.so — the HOST translation units (rule compiler, loaders, batcher, node) under AddressSanitizer + UBSan, the
    HIP ones as usual: for the CPU fuzz tools and the CPU suite (PWAF_LIB_VARIANT=asan, LD_PRELOAD = the toolchain's
    libclang_rt.asan-x86_64.so, ASAN_OPTIONS=detect_leaks=0; tests/test_batcher_cpu.py's ThreadSanitizer leg cannot share a process with it)."""
    lib = LIB if not variant else os.path.join(HERE, f"libpwaf_{variant}.so")
    if not force and not is_stale(lib):
        return lib
    embed()
    objs = []
    objdir = os.path.join(HERE, "build" + ("_" + variant if variant else ""))
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", os.path.join(HERE, "..", "include")]
    # PWAF_EXTRA_CXXFLAGS=-DPWAF_PROFILING builds the timing-experiment variant (env switches that change results); never the default
    common += os.environ.get("PWAF_EXTRA_CXXFLAGS", "").split()
    if variant == "prof":
        common.append("-DPWAF_PROFILING")
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), *(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)):
            continue
        cmd = [cc, *common, "-c", path, "-o", obj]
        if src.endswith(".hip") or src in ("engine.cpp", "rtc.cpp"):
            cmd[1:1] = ["-x", "hip", "--offload-arch=gfx950"]
        else:
            # host-only translation units (no HIP headers): plain C++
            cmd[1:1] = ["-x", "c++"] + (["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"] if variant == "asan" else [])
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *(["-fsanitize=address,undefined", "-shared-libsan"] if variant == "asan" else []), "-o", lib, *objs]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, variant="prof" if "--prof" in sys.argv else "asan" if "--asan" in sys.argv else ""))
