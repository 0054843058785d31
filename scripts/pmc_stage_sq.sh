#!/bin/bash
# SQ counters of the stage-wise solve kernel at T = 30, 65,536 instances: where a wave's cycles go.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/stage30sq
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 3 30 > $OUT/a.log 2>&1 \
&& python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob('$OUT/a/**/p_counter_collection.csv', recursive=True)[0])):
    if 'k_solve_stage' in r['Kernel_Name']:
        acc[(r['Dispatch_Id'], r['Counter_Name'])].append(float(r['Counter_Value']))
per = collections.defaultdict(list)
for (d, c), v in acc.items(): per[c].append(sum(v))
rec = {c: sorted(v)[len(v) // 2] for c, v in per.items()}
w = rec.get('SQ_WAVES', 1.0)
out = {"kernel": "k_solve_stage<dual steer>", "T": 30, "batch": 65536, "per_wave": {c: v / w for c, v in rec.items()}, "waves": w}
print(json.dumps(out))
json.dump(out, open('$OUT/stage30_sq.json', 'w'), indent=1)
PY
