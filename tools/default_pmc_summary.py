"""MFMA-busy share and average duration per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_BUSY_CYCLES) of the default-regime nomination loop (tools/nominate_default_trace.py, B7_TRACE_CASE=1: N = 100, d = 6,
2e4 candidates).   usage: default_pmc_summary.py <dir given to rocprofv3 -d> [out.json]
SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD-quad (x4 per CU cycle at full use); MFMA-busy = counter / (4 x CUs x cycles), with the
kernel's cycles from GRBM_GUI_ACTIVE / 8 XCDs, as tools/potrf_pmc_summary.py."""
import csv
import glob
import json
import sys

root = sys.argv[1]
rows = []
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
per = {}
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    k = per.setdefault(name, {"disp": {}, "ctr": {}})
    k["disp"][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k["ctr"].setdefault(r["Counter_Name"], {})[r["Dispatch_Id"]] = float(r["Counter_Value"])
out = {"note": "GRBM_GUI_ACTIVE also counts the dispatch's front and back end, which is not small against kernels of tens of "
               "microseconds: effective_clock_GHz comes out above the real 2.4 and mfma_busy is UNDER-stated by the same factor; "
               "kpost_small_kernel's launches are half S = 1 and half S = 10 nominations",
       "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python3 tools/nominate_default_trace.py "
                  "(B7_TRACE_CASE=1: N = 100, d = 6, 20000 candidates, S = 1 and S = 10 nominations)", "kernels": {}}
for name, k in sorted(per.items(), key=lambda kv: -sum(kv[1]["disp"].values())):
    n = len(k["disp"])
    if n < 5:
        continue
    avg_us = sum(k["disp"].values()) / n / 1e3
    e = {"launches": n, "avg_duration_us_under_pmc": round(avg_us, 2)}
    gui = k["ctr"].get("GRBM_GUI_ACTIVE")
    busy = k["ctr"].get("SQ_VALU_MFMA_BUSY_CYCLES")
    if gui and busy:
        cyc = sum(gui.values()) / 8.0            # per XCD
        e["effective_clock_GHz"] = round(cyc / (sum(k["disp"].values())), 3)
        e["mfma_busy_frac_of_simd_cycles"] = round(sum(busy.values()) / (4.0 * 256 * cyc), 4)
    out["kernels"][name] = e
txt = json.dumps(out, indent=1)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
