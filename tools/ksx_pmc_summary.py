"""Per-class summary of tools/ksx_pmc.sh: counters per launch of ksx_kernel<DPAD> (largest grid only = the K(X*,X) launches)."""
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
acc, dur, meta = {}, {}, {}
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"ksx_kernel<(\d+)", r["Kernel_Name"])
        if not m or int(r["Grid_Size"]) < 262144 // 64 * 256:
            continue
        k = "DPAD%s" % m.group(1)
        a = acc.setdefault((k, r["Counter_Name"]), [0.0, set()])
        a[0] += float(r["Counter_Value"])
        a[1].add((f, r["Dispatch_Id"]))
        dur.setdefault(k, {})[(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        meta[k] = {"LDS_bytes_per_workgroup": int(r.get("LDS_Block_Size", 0) or 0), "VGPRs": int(r.get("VGPR_Count", 0) or 0),
                   "accum_VGPRs": int(r.get("Accum_VGPR_Count", 0) or 0), "SGPRs": int(r.get("SGPR_Count", 0) or 0)}
res = {}
for k in sorted(meta):
    per = {c: v[0] / max(1, len(v[1])) for (kk, c), v in acc.items() if kk == k}
    t_ns = sum(dur[k].values()) / len(dur[k])
    e = dict(meta[k], counters_per_launch=per, duration_ms_under_pmc=t_ns / 1e6)
    lds = max(1, e["LDS_bytes_per_workgroup"])
    e["workgroups_per_CU_by_LDS"] = min(8, 163840 // lds)
    if "GRBM_GUI_ACTIVE" in per:
        e["effective_clock_GHz"] = per["GRBM_GUI_ACTIVE"] / 8 / t_ns
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            e["mfma_busy_frac_of_simd_cycles"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (per["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
    if per.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if c in per:
                e[c.lower() + "_share_of_wave_cycles"] = per[c] / per["SQ_WAVE_CYCLES"]
    if "WRITE_SIZE" in per:
        e["hbm_write_bytes"] = per["WRITE_SIZE"] * 1024
        e["hbm_write_TBps"] = per["WRITE_SIZE"] * 1024 / t_ns / 1e3
    res[k] = e
print(json.dumps(res, indent=1))
