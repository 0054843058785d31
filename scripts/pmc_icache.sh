#!/bin/bash
# Instruction-cache counters of the solve kernel (library chosen by MPCX_LIB).
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/$1
timeout -s KILL 90 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$1/a -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 > gpurun_out/$1/a.log 2>&1 \
&& python3 - <<PY
import csv,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/$1/a/p_counter_collection.csv')):
    if 'k_solve' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value'])); g=int(r['Grid_Size'])//64
for k,v in sorted(acc.items()): print(k,round(sum(v)/len(v)/g,1),'per wave')
PY
