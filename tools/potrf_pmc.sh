#!/bin/bash
# MFMA-busy share and trailing-update utilisation of potrf_persist_kernel in the three shapes the path runs
# it in (single fit, ten fits side by side, sixteen likelihoods).   usage (GPU box, repo root): bash tools/potrf_pmc.sh [N]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-2048}
OUT=$R/gpurun_out/potrf_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in single s10 nll16; do
  python3 $R/tools/potrf_shapes.py $s $N 20 > $OUT/${s}_plain.out 2>&1 || exit 1
  cat $OUT/${s}_plain.out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/${s}_trace --output-format csv -- python3 $R/tools/potrf_shapes.py $s $N 8 \
    > $OUT/${s}_trace.out 2> $OUT/${s}_trace.err || echo "$s trace failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/$s --output-format csv -- \
    python3 $R/tools/potrf_shapes.py $s $N 8 > $OUT/${s}_pmc.out 2> $OUT/${s}_pmc.err || echo "$s pmc failed"
done
python3 $R/tools/potrf_pmc_summary.py $OUT $N > $OUT/summary.json
cat $OUT/summary.json
