#!/bin/bash
# Two SQ counter passes over a few launches of the headline solve (library chosen by MPCX_LIB).
set -o pipefail
TAG=$1
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/$TAG/a -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 > gpurun_out/$TAG/a.log 2>&1 \
&& timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/$TAG/b -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 > gpurun_out/$TAG/b.log 2>&1 \
&& python3 - <<PY
import csv,collections
for d in ['a','b']:
    acc=collections.defaultdict(list); waves=None
    for r in csv.DictReader(open(f'gpurun_out/$TAG/{d}/p_counter_collection.csv')):
        if 'k_solve' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); g=int(r['Grid_Size'])//64
    for k,v in sorted(acc.items()): print(d,k,round(sum(v)/len(v)/g,1))
PY
