#!/bin/bash
# Kernel time and HBM traffic of the stage-wise solve kernel at T = 30, 65,536 instances:
# one kernel-trace pass, then FETCH_SIZE and WRITE_SIZE in passes of their own.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/stage30
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 30 > $OUT/trace.log 2>&1 \
&& timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 30 > $OUT/fetch.log 2>&1 \
&& timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 30 > $OUT/write.log 2>&1 \
&& python3 - <<PY
import csv, glob, json
def per_dispatch(path, counter):
    d = {}
    for r in csv.DictReader(open(path)):
        if 'k_solve_stage' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            d[r['Dispatch_Id']] = d.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    v = sorted(d.values()); return v[len(v) // 2]
fetch = per_dispatch(glob.glob('$OUT/fetch/**/p_counter_collection.csv', recursive=True)[0], 'FETCH_SIZE')
write = per_dispatch(glob.glob('$OUT/write/**/p_counter_collection.csv', recursive=True)[0], 'WRITE_SIZE')
dur = None
for r in csv.DictReader(open(glob.glob('$OUT/trace/**/p_kernel_stats.csv', recursive=True)[0])):
    if 'k_solve_stage' in r['Name']: dur = float(r['AverageNs']); calls = int(r['Calls'])
# FETCH_SIZE / WRITE_SIZE come in units of 1024 bytes; gfx950 counts a 128-byte read request as 64 bytes
rd, wr = 2 * fetch * 1024, write * 1024
rec = {"kernel": "k_solve_stage<dual steer>", "variant": "dsr_ackermann", "T": 30, "batch": 65536, "avg_ns": dur, "launches": calls,
       "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_GBps": (rd + wr) / dur, "algorithmic_bytes": 65536 * 3449}
print(json.dumps(rec))
json.dump(rec, open('$OUT/stage30.json', 'w'), indent=1)
PY
