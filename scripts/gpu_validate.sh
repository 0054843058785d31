#!/bin/bash
# Round validation on the GPU box: GPU tests, smoke, bench, kernel-trace stats, and the counter
# passes (each in a run of its own) of the headline kernel, its chained instantiation, the T = 10
# four-lane kernel (32,768 and 65,536 instances), the wavefront-per-instance kernel (T = 30, 1,024, both variants)
# and the stage-wise kernel (T = 30 at 65,536, both variants).
# Every GPU step has its own time limit and the chain stops at the first failure.
# Afterwards, here: python scripts/summarise_profiles.py <tag>  (writes profiles/).
set -o pipefail
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
pmc() {  # pmc <dir> <counters...> -- <script args...>
  local d=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 150 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/$d -o p -- python3 "$@" > $OUT/$d.log 2>&1
}
SQ_A="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES"
SQ_F64="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
SQ_B="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 300 --timeout-method thread > $OUT/gputests.log 2>&1 \
 && tail -3 $OUT/gputests.log \
 && timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 \
 && tail -1 $OUT/smoke.log \
 && timeout -k 10 300 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err \
 && timeout -k 10 300 python bench.py --workload mixed4m > $OUT/mixed4m.json 2> $OUT/mixed4m.err \
 && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/trace.log 2>&1 \
 && pmc pmc_fetch FETCH_SIZE -- bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs \
 && pmc pmc_write WRITE_SIZE -- bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs \
 && pmc sq5_a $SQ_A -- scripts/prof_solve.py dsr_ackermann 65536 5 \
 && pmc sq5_b $SQ_B -- scripts/prof_solve.py dsr_ackermann 65536 5 \
 && pmc sq5_f64 $SQ_F64 -- scripts/prof_solve.py dsr_ackermann 65536 5 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ch_trace -o p -- python3 scripts/prof_solve.py sim_ackermann 65536 20 > $OUT/ch_trace.log 2>&1 \
 && pmc ch_sq_a $SQ_A -- scripts/prof_solve.py sim_ackermann 65536 5 \
 && pmc ch_sq_b $SQ_B -- scripts/prof_solve.py sim_ackermann 65536 5 \
 && pmc ch_fetch FETCH_SIZE -- scripts/prof_solve.py sim_ackermann 65536 5 \
 && pmc ch_write WRITE_SIZE -- scripts/prof_solve.py sim_ackermann 65536 5 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wv_trace -o p -- python3 scripts/prof_solve.py dsr_ackermann 1024 20 30 > $OUT/wv_trace.log 2>&1 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wd_trace -o p -- python3 scripts/prof_solve.py dsr_diff 1024 20 30 > $OUT/wd_trace.log 2>&1 \
 && pmc wv_sq_a $SQ_A -- scripts/prof_solve.py dsr_ackermann 1024 5 30 \
 && pmc wv_sq_b $SQ_B -- scripts/prof_solve.py dsr_ackermann 1024 5 30 \
 && pmc q10_fetch FETCH_SIZE -- scripts/prof_solve.py mpc_py 32768 5 \
 && pmc q10_write WRITE_SIZE -- scripts/prof_solve.py mpc_py 32768 5 \
 && pmc q10_sq_a $SQ_A -- scripts/prof_solve.py mpc_py 32768 5 \
 && pmc q10_sq_b $SQ_B -- scripts/prof_solve.py mpc_py 32768 5 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q10_trace -o p -- python3 scripts/prof_solve.py mpc_py 65536 200 > $OUT/q10_trace.log 2>&1 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q10b_trace -o p -- python3 scripts/prof_solve.py mpc_py 32768 200 > $OUT/q10b_trace.log 2>&1 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_trace -o p -- python3 scripts/prof_solve.py dsr_ackermann 65536 5 30 > $OUT/st_trace.log 2>&1 \
 && pmc st_fetch FETCH_SIZE -- scripts/prof_solve.py dsr_ackermann 65536 5 30 \
 && pmc st_write WRITE_SIZE -- scripts/prof_solve.py dsr_ackermann 65536 5 30 \
 && pmc st_sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU -- scripts/prof_solve.py dsr_ackermann 65536 3 30 \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lp_trace -o p -- python3 scripts/prof_loop.py > $OUT/lp_trace.log 2>&1 \
 && pmc lp_fetch FETCH_SIZE -- scripts/prof_loop.py \
 && pmc lp_write WRITE_SIZE -- scripts/prof_loop.py \
 && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sd_trace -o p -- python3 scripts/prof_solve.py dsr_diff 65536 5 30 > $OUT/sd_trace.log 2>&1 \
 && pmc sd_fetch FETCH_SIZE -- scripts/prof_solve.py dsr_diff 65536 3 30 \
 && pmc sd_write WRITE_SIZE -- scripts/prof_solve.py dsr_diff 65536 3 30 \
 && echo validate-done
