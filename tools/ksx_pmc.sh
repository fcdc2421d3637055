#!/bin/bash
# a PMC pass per covariance-kernel width class (DPAD 32 / 48 / 64 / 96): MFMA-busy share, VALU-busy share,
# LDS conflict share, wave-cycle breakdown, HBM write bytes, effective clock.  Workload: tools/ksx_rate.py (N = 2048,
# 262144 candidates per launch, d = 32, 39, 64, 96).   usage (GPU box, repo root): bash tools/ksx_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ksx_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/ksx_rate.py 32 39 64 96 > $OUT/rate.txt 2>&1 || exit 1
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pass$i --output-format csv -- python3 $R/tools/ksx_rate.py 32 39 64 96 > $OUT/pass$i.out 2> $OUT/pass$i.err || echo "pass $i failed"
done
python3 $R/tools/ksx_pmc_summary.py $OUT > $OUT/summary.json
cat $OUT/rate.txt
cat $OUT/summary.json
