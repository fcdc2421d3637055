"""Summary of the potrf_persist_kernel rows of tools/potrf_pmc.sh's rocprofv3 passes.
usage: potrf_pmc_summary.py <out-dir> > profiles/rNN_potrf_pmc.json

Per shape (single / s10 / nll16): fits per launch, launch duration, SQ_VALU_MFMA_BUSY_CYCLES against the SIMD cycles the
launch was resident (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), and the ALGORITHMIC utilisation: flop of the trailing updates
(N^3/3 per fit; with the inverse another N^3/3) / duration / 78.6 TFLOP/s.  MFMA-busy counts every v_mfma issued, i.e. the
trailing updates, the triangular solves and the inverse's tile products alike."""
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
PEAK = 78.6e12
res = {"n_obs": N, "peak_tflops": 78.6, "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
       "SQ_BUSY_CYCLES -d <dir> --output-format csv -- python3 tools/potrf_shapes.py <shape> %d" % N, "shapes": {}}
fits = {"single": 1, "s10": 10, "nll16": 16}
with_inverse = {"single": True, "s10": True, "nll16": False}
for shape in ("single", "s10", "nll16"):
    rows = []
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, shape), recursive=True):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if "potrf_persist_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    by = {}
    for r in rows:
        e = by.setdefault(r["Dispatch_Id"], {"grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                                             "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # the launches of interest: the largest grid of the run (the fit batch; jitter-free runs have nothing else)
    g = max(e["grid"] for e in by.values())
    sel = [e for e in by.values() if e["grid"] == g][1:] or [e for e in by.values() if e["grid"] == g]
    n = len(sel)
    dur = sum(e["ns"] for e in sel) / n
    mfma = sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for e in sel) / n
    gui = sum(e.get("GRBM_GUI_ACTIVE", 0.0) for e in sel) / n
    per_fit = N ** 3 / 3.0
    flop_update = fits[shape] * per_fit
    flop_all = flop_update * (2 if with_inverse[shape] else 1)
    trace = glob.glob("%s/%s_trace/**/*kernel_stats.csv" % (root, shape), recursive=True)
    unprofiled = None
    for f in trace:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "potrf_persist_kernel" in r["Name"]:
                    unprofiled = float(r["AverageNs"]) / 1e3
    e = {"fits_per_launch": fits[shape], "with_inverse": with_inverse[shape], "launches_averaged": n,
         "workgroups": g // sel[0]["wg"], "duration_us_under_pmc": dur / 1e3,
         "duration_us_kernel_trace_only": unprofiled,
         "mfma_busy_frac_of_simd_cycles": mfma / (gui / 8 * 1024) if gui else None,
         "effective_clock_GHz_under_pmc": gui / 8 / dur if gui else None,
         "trailing_update_flop": flop_update, "all_gemm_flop": flop_all}
    for key, d_us in (("under_pmc", dur / 1e3), ("kernel_trace_only", unprofiled)):
        if d_us:
            e["trailing_update_util_" + key] = flop_update / (d_us * 1e-6) / PEAK
            e["chol_plus_inverse_util_" + key] = flop_all / (d_us * 1e-6) / PEAK
    res["shapes"][shape] = e
wall = {}
for shape in fits:
    try:
        m = re.search(r"([\d.]+) ms per call", open("%s/%s_plain.out" % (root, shape)).read())
        wall[shape] = float(m.group(1))
    except Exception:
        pass
res["wall_ms_per_call_unprofiled"] = wall
res["note"] = ("MFMA-busy is SQ_VALU_MFMA_BUSY_CYCLES summed over the chip's 1024 SIMDs divided by (GRBM_GUI_ACTIVE / 8 XCDs) x 1024; "
               "the profiler's counter passes lower the clock (DESIGN section 8), so durations under PMC are longer than the "
               "kernel-trace-only ones; utilisation = N^3/3 flop per fit (trailing updates only) / duration / 78.6 TFLOP/s.")
print(json.dumps(res, indent=1))
