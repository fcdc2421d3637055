"""Summary of tools/ws_experiment.sh: per workspace size, step time and per-STEP HBM bytes of post_kernel / ksx_kernel
(FETCH_SIZE doubled per MI355X_MICROARCH.md for wide coalesced reads, WRITE_SIZE as is; both in KiB), effective clock
= GRBM_GUI_ACTIVE / 8 / kernel time.  Each profiled run has 2 steps (1 warm-up + 1 timed): totals are halved."""
import csv
import glob
import json
import sys

root, sizes = sys.argv[1], sys.argv[2].split()
res = {}
for ws in sizes:
    b = json.loads(open("%s/bench_%s.json" % (root, ws)).read().strip().splitlines()[-1])
    e = {"ms_per_step": b["ms_per_step"], "post_ms_avg": b["phases"]["post"]["ms_avg"],
         "post_launches_per_step": b["phases"]["post"]["launches"] // b["steps"],
         "post_TFLOPs": b["roofline"]["achieved"], "ksx_ms_per_step": b["phases"]["ksx"]["ms_total"] / b["steps"],
         "post_ms_per_step": b["phases"]["post"]["ms_total"] / b["steps"]}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
        tot, dur = {}, {}
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (root, ws, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                k = "post" if "post_kernel" in r["Kernel_Name"] else ("ksx" if "ksx_kernel" in r["Kernel_Name"] else None)
                if k is None or r["Counter_Name"] != ctr:
                    continue
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
                dur[k] = dur.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k in tot:
            if ctr == "FETCH_SIZE":
                e[k + "_hbm_read_GB_per_step"] = tot[k] * 1024 * 2 / 2 / 1e9
            elif ctr == "WRITE_SIZE":
                e[k + "_hbm_write_GB_per_step"] = tot[k] * 1024 / 2 / 1e9
            else:
                e[k + "_clock_GHz"] = tot[k] / 8 / dur[k]
    res[ws + " MiB"] = e
print(json.dumps(res, indent=1))
