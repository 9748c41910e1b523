#!/bin/bash
# Collects the numbers kept under profiles/: the default bench line, rocprofv3 kernel statistics of the
# same command, and three counter passes (SQ, FETCH_SIZE, WRITE_SIZE) over one step of the alignment
# workload.  Run on the GPU box from the repository root: bash tools/profile_round.sh <tag>
set -e
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -k 10 500 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
n=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc$n -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-minimizers > $OUT/pmc$n.json 2> $OUT/pmc$n.err
done
find $OUT -name "*.csv" | head -20
