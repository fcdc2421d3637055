#!/bin/bash
# does K(X*,X) have to go to HBM?  The headline step at several chunk-workspace sizes: step time from
# bench.py itself, HBM bytes of post_kernel / ksx_kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE,
# GRBM_GUI_ACTIVE), summed over ALL launches of one profiled run (1 warm-up + 1 timed step) so that different chunk
# sizes compare.   usage (on the GPU box, from the repo root): bash tools/ws_experiment.sh "4096 1024 256 128"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ws
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ws in $1; do
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workspace-mib $ws > $OUT/bench_$ws.json 2> $OUT/bench_$ws.err || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_${ws}_$ctr --output-format csv -- \
      python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --workspace-mib $ws > $OUT/pmc_${ws}_$ctr.out 2> $OUT/pmc_${ws}_$ctr.err || exit 1
  done
  echo "ws $ws done"
done
python3 $R/tools/ws_summary.py $OUT "$1" > $OUT/summary.json
cat $OUT/summary.json
