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
LOG = os.path.join(HERE, "lib", "build_resource_usage.txt")


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-o", OUT, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise RuntimeError("hipcc failed building libmpcx.so")
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
