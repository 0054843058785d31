#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; csv output) of bench.py into
profiles/hbm_traffic.json: HBM bytes per launch of the solve kernel.

usage: summarise_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <batch>

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 bytes by rocprofv3; on
gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes, so it is doubled.
"""
import csv
import json
import os
import sys


def mean_counter(path, kernel, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} in {path}")
    # per-dispatch rows may be split per XCD/dimension: sum rows of one dispatch first
    by_dispatch = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                by_dispatch.setdefault(row["Dispatch_Id"], 0.0)
                by_dispatch[row["Dispatch_Id"]] += float(row["Counter_Value"])
    d = sorted(by_dispatch.values())
    return d[len(d) // 2], len(d)


def main():
    fetch_csv, write_csv, kernel, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    fetch, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
    write, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
    rec = {"kernel": kernel, "batch": batch,
           "fetch_size_kb_raw": fetch, "write_size_kb_raw": write,
           "fetch_bytes": 2.0 * fetch * 1024.0, "write_bytes": write * 1024.0,
           "bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
           "dispatches": [nf, nw],
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; "
                  "median over dispatches; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B)"}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
