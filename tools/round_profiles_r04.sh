#!/bin/bash
# Round 4's evidence on top of tools/round_profiles.sh: the reference's default regime (bench line, kernel stats, MFMA-busy of
# its kernels, latencies, phase stamps of the one-workgroup fit / likelihood kernel), the persistent Cholesky's stamps at
# N = 2048, the bit comparison of the small-problem kernels, the single-process group rehearsal.
# usage (GPU box, repo root): bash tools/round_profiles_r04.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload default --steps 3 --warmup 1 > $OUT/default_regime.json 2> $OUT/default_regime.err || echo "default failed"
echo "default regime done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/default_trace --output-format csv -- python3 $R/bench.py --workload default --steps 1 --warmup 0 --no-cpu-baseline \
  > $OUT/default_under_rocprof.json 2> $OUT/default_trace.err || echo "default trace failed"
cp $OUT/default_trace/*/*kernel_stats.csv $OUT/default_kernel_stats.csv 2>/dev/null
head -8 $OUT/default_kernel_stats.csv
python3 $R/tools/nominate_default_trace.py > $OUT/nominate_latency.txt 2>&1
python3 $R/tools/nll_latency.py > $OUT/nll_latency.txt 2>&1
cat $OUT/nominate_latency.txt $OUT/nll_latency.txt
export B7_TRACE_CASE=1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/default_pmc --output-format csv -- \
  python3 $R/tools/nominate_default_trace.py > $OUT/default_pmc.out 2> $OUT/default_pmc.err || echo "default pmc failed"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_nom100 -- python3 $R/tools/nominate_default_trace.py > /dev/null 2>&1
unset B7_TRACE_CASE
python3 $R/tools/default_pmc_summary.py $OUT/default_pmc $OUT/default_pmc_summary.json > /dev/null
cat $OUT/default_pmc_summary.json
python3 $R/tools/timeline.py $OUT/tl_nom100 8 > $OUT/default_nomination_timeline.txt 2>&1
cat $OUT/default_nomination_timeline.txt
timeout -k 10 200 python3 $R/tools/gp_small_stamps.py run > $OUT/gp_small_stamps.txt 2>&1 || echo "stamps failed"
timeout -k 10 300 python3 $R/tools/small_fit_bits.py > $OUT/small_fit_bits.txt 2>&1 || echo "bits differ"
tail -2 $OUT/small_fit_bits.txt
timeout -k 10 200 python3 $R/tools/bits_fingerprint.py > $OUT/bits_fingerprint.txt 2>&1
tail -1 $OUT/bits_fingerprint.txt
B7_PERSIST_STAMPS=1 timeout -k 10 200 python3 $R/tools/persist_stamps.py 2048 > $OUT/persist_stamps_N2048.txt 2>&1 || echo "persist stamps failed"
head -30 $OUT/persist_stamps_N2048.txt
timeout -k 10 200 python3 $R/tools/fit_by_n.py > $OUT/fit_by_N.txt 2>&1
cat $OUT/fit_by_N.txt
python3 $R/bench.py --gpus 2 --virtual-ranks --steps 3 > $OUT/group2_virtual.json 2> $OUT/group2.err || echo "group failed"
python3 $R/bench.py --gpus 8 --virtual-ranks --steps 3 --candidates 131072 > $OUT/group8_virtual.json 2> $OUT/group8.err || echo "group8 failed"
python3 $R/bench.py --workload cfg5 --steps 20 --samples 10 --no-cpu-baseline > $OUT/cfg5_s10.json 2>/dev/null
echo "r04 extras done"
