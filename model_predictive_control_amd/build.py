"""Builds libmpcx.so (HIP, gfx950 only) in-tree: python -m model_predictive_control_amd.build"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "mpcx_kernels.hip")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))) + [os.path.join(ROOT, "include", "mpcx.h")]
OUT = os.path.join(HERE, "lib", "libmpcx.so")
# the same sources with -DMPCX_TEST_HOOKS: adds the first-generation one-lane-per-instance kernel
# and the environment variables that route solves to a chosen kernel; loaded only by
# tests/test_gpu_cross_kernels.py (through MPCX_LIB), never by the package itself
OUT_HOOKS = os.path.join(HERE, "lib", "libmpcx_hooks.so")
LOG = os.path.join(HERE, "lib", "build_resource_usage.txt")


def up_to_date() -> bool:
    return all(os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in DEPS) for o in (OUT, OUT_HOOKS))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc")]
    # both libraries at once: the compile is one translation unit each, about a minute
    hooks = subprocess.Popen(cmd + ["-DMPCX_TEST_HOOKS", "-o", OUT_HOOKS, SRC], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    r = subprocess.run(cmd + ["-o", OUT, SRC], capture_output=True, text=True)
    _, hooks_err = hooks.communicate()
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise RuntimeError("hipcc failed building libmpcx.so")
    if hooks.returncode != 0:
        sys.stderr.write(hooks_err)
        raise RuntimeError("hipcc failed building libmpcx_hooks.so")
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = {"name": m.group(2)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(1).split(" ")[0]] = m.group(2)
    with open(LOG, "w") as f:
        for c in rows:
            f.write(" ".join(f"{k}={v}" for k, v in c.items()) + "\n")
    if verbose:
        print(open(LOG).read())
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
