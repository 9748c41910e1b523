#!/bin/bash
# Collects the numbers kept under profiles/: the default bench line, rocprofv3 kernel statistics of the
# same command, and three counter passes (SQ, FETCH_SIZE, WRITE_SIZE; separate runs, as the TCC block cannot
# hold both size counters at once) over one short run of every workload of bench.py.
# Run on the GPU box from the repository root: bash tools/profile_round.sh <tag> <commit>   (the box has no .git)
set -e
TAG=${1:-final}
COMMIT=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
echo "[profile] bench" && timeout -k 10 500 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "[profile] kernel trace" && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
n=0
export SA_NO_OVERLAP=1   # a step's fill in one launch, so that a kernel's entry is its traffic per step
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  n=$((n+1))
  echo "[profile] counters pass $n"
  timeout -s KILL 400 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc$n -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 1 --leg-steps 1 --cpu-sample 0 --no-minimizers --legs nw_1kx1k,sg_10kx11k,nw_100kx100k > $OUT/pmc$n.json 2> $OUT/pmc$n.err
  for leg in reads genome; do
    timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc$leg$n -o pmc -- python3 $ROOT/bench.py --only-minimizers $leg > $OUT/pmc$leg$n.json 2> $OUT/pmc$leg$n.err
  done
done
unset SA_NO_OVERLAP
python3 $ROOT/tools/summarize_pmc.py $OUT $COMMIT > $OUT/pmc_per_launch.json
find $OUT -name "*.csv" | head -20
