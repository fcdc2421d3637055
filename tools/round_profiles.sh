#!/bin/bash
# Round-end evidence: bench lines of every BASELINE config, rocprofv3 kernel stats of the default bench command, and the
# PMC passes behind bench.py's roofline.traffic.   usage (GPU box, repo root): bash tools/round_profiles.sh r02
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_metric.json 2> $OUT/bench_metric.err || exit 1
echo "metric done"
: > $OUT/bench_other_configs.jsonl
for w in cfg2 cfg3 cfg4 cfg5; do
  sleep 3   # the previous process freed tens of GB: let the driver finish before a sub-millisecond workload is timed
  python3 $R/bench.py --workload $w --steps 20 --warmup 5 >> $OUT/bench_other_configs.jsonl 2> $OUT/bench_$w.err || echo "$w failed"
done
echo "configs done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-extras \
  > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || echo "trace failed"
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $ctrs | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/pmc/$n --output-format csv -- \
    python3 $R/bench.py --steps 1 --warmup 1 --candidates 262144 --no-cpu-baseline --no-extras > $OUT/pmc_$n.out 2> $OUT/pmc_$n.err || echo "pmc $n failed"
done
# in-kernel clock and MFMA issue efficiency of the dominant kernel (diagnostic build: python tools/post_clock.py build, here)
if [ -f $R/tools/_build/libbot7hip_stamps.so ]; then
  timeout -k 10 200 python3 $R/tools/post_clock.py run > $OUT/post_clock.txt 2> $OUT/post_clock.err || echo "post_clock failed"
  cat $OUT/post_clock.txt
fi
python3 $R/tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json "round $TAG final code." > /dev/null
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
head -12 $OUT/kernel_stats.csv
python3 - <<PY
import json
for l in open("$OUT/bench_other_configs.jsonl"):
    d = json.loads(l)
    print(d["metric"], "%.3g cand/s" % d["value"], "%.3f ms/step (%.3f without phase events)" % (d["ms_per_step"], d["ms_per_step_without_phase_events"]), "fit %.3f ms" % d["gp_fit_ms"], "ksx frac", d.get("ksx_hbm_frac"), "best", d["best"])
d = json.load(open("$OUT/bench_metric.json"))
print("metric", d["value"], d["ms_per_step"], d["gp_fit_ms"], d["gp_fit_ms_by_N"], d["roofline"]["frac"], d.get("marginalised"))
PY
