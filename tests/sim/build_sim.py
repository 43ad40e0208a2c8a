"""Builds tests/sim/_build/libsvdx_sim.so: the kernel sources of svd_xtend_amd/csrc compiled for the HOST against the wave64 functional
simulator (sim_rt.h / sim_rt.cpp).  TEST INFRASTRUCTURE -- the product never loads this library.

The sources are used as they are; a handful of textual substitutions replace what only the gfx950 back end understands:
  __builtin_amdgcn_*            -> sim_amdgcn_*   (functions of sim_rt.h)
  asm volatile("s_waitcnt ...") -> sim_waitcnt_vm(N) / sim_waitcnt_lgkm(N)
  the buffer_store_dwordx4 asm  -> sim_buffer_store
  extern __shared__ T name[];   -> T* name = the workgroup's dynamic LDS
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.environ.get("SVDX_SIM_CSRC") or os.path.join(ROOT, "svd_xtend_amd", "csrc")      # mutation tests point this at an edited copy
OUT = os.environ.get("SVDX_SIM_OUT") or os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libsvdx_sim.so")
CXX = os.environ.get("SVDX_SIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")
OPT = os.environ.get("SVDX_SIM_OPT", "-O0")      # kernel sources: -O0 compiles gemm.hip's instantiations in seconds (-O1: most of a minute); the scheduler is -O2
FLAGS = ["-std=c++17", "-fPIC", "-pthread", "-Wno-unused-value", "-Wno-pass-failed", "-Wno-unknown-pragmas", "-Wno-deprecated-declarations",
         "-ffp-contract=off", "-I", HERE]

SUBS = [
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\(%0\)"\s*::\s*"n"\((.*?)\)\s*:\s*"memory"\)'), r"sim_waitcnt_vm(\1)"),
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)"\s*:::\s*"memory"\)'), r"sim_waitcnt_vm(\1)"),
    (re.compile(r'asm volatile\("s_waitcnt lgkmcnt\((\d+)\)"\s*:::\s*"memory"\)'), r"sim_waitcnt_lgkm(\1)"),
    (re.compile(r'asm volatile\("buffer_store_dwordx4 %0, %1, %2, %3 offen\\n\\ts_nop 1"\s*::\s*"v"\((\w+)\),\s*"v"\((\w+)\),\s*"s"\((\w+)\),\s*"s"\((\w+)\)\s*:\s*"memory"\)'),
     r"sim_buffer_store(&\1, 16, \3, \2, \4)"),
    (re.compile(r"/\*SIM-BEGIN\*/.*?/\*SIM-END (.*?)\*/", re.S), r"\1"),          # a region the simulator models as one call (hidden_dma)
    (re.compile(r"extern __shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];"), r"\1* \2 = reinterpret_cast<\1*>(sim_dyn_lds());"),
    (re.compile(r"__builtin_amdgcn_"), "sim_amdgcn_"),
    (re.compile(r'#include "\.\./\.\./include/svdx\.h"'), f'#include "{os.path.join(ROOT, "include", "svdx.h")}"'),
]


def transform(text: str) -> str:
    for rx, rep in SUBS:
        text = rx.sub(rep, text)
    left = re.findall(r'asm volatile\("[^"]+"', text)
    if left:
        raise RuntimeError(f"inline assembly the simulator has no model for: {sorted(set(left))}")
    return text


def sources():
    sys.path.insert(0, ROOT)
    from svd_xtend_amd import build as b
    return list(b.SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    srcs = sources()
    hdrs = [f for f in os.listdir(CSRC) if f.endswith(".h")]
    deps = [os.path.join(CSRC, f) for f in srcs + hdrs] + [os.path.join(HERE, f) for f in ("sim_rt.h", "sim_rt.cpp", "build_sim.py")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    for f in srcs + hdrs:
        with open(os.path.join(CSRC, f)) as fh:
            t = transform(fh.read())
        out = os.path.join(OUT, os.path.splitext(f)[0] + (".cpp" if not f.endswith(".h") else ".h"))
        if not os.path.exists(out) or open(out).read() != t:
            with open(out, "w") as fh:
                fh.write(t)

    def cc(src):
        o = os.path.join(OUT, os.path.splitext(os.path.basename(src))[0] + ".o")
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in deps):
            cmd = [CXX] + FLAGS + ["-O2" if os.path.basename(src) == "sim_rt.cpp" else OPT] + ["-c", src, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"{src}:\n{r.stderr[-6000:]}")
        return o

    units = [os.path.join(OUT, os.path.splitext(f)[0] + ".cpp") for f in srcs] + [os.path.join(HERE, "sim_rt.cpp")]
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, units))
    r = subprocess.run([CXX, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
