#!/bin/bash
# Round validation on the GPU box: GPU tests, smoke, bench, kernel-trace stats and the two
# HBM counter passes of the same bench command.  Every GPU step has its own time limit and the
# chain stops at the first failure.
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > $OUT/gputests.log 2>&1 \
 && tail -3 $OUT/gputests.log \
 && timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 \
 && tail -1 $OUT/smoke.log \
 && timeout -k 10 300 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err \
 && cat $OUT/bench.json \
 && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1 \
 && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/pmc_fetch.log 2>&1 \
 && timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/pmc_write.log 2>&1 \
 && echo validate-done
