"""Per-launch PMC summary of the two dominant kernels from separate rocprofv3 --pmc passes of bench.py.
usage: pmc_summary.py <dir-of-pass-dirs> <out.json> [note]
Each pass:  rocprofv3 --kernel-trace --pmc <COUNTERS> -d <dir>/<pass> --output-format csv --
            python3 bench.py --steps 1 --warmup 1 --candidates 262144 --no-cpu-baseline --no-extras
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide
coalesced reads), WRITE_SIZE is taken as is.  GRBM_GUI_ACTIVE / 8 XCDs / kernel time = effective clock."""
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]


def kind(name):
    return "post_kernel" if "post_kernel" in name else ("ksx_kernel" if "ksx_kernel" in name else None)


rows = []
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        rows += [(f, r) for r in csv.DictReader(fh) if kind(r["Kernel_Name"])]
# ksx_kernel also assembles K(X,X) (a much smaller grid): keep only the launches with the largest grid per kernel
big = {}
for f, r in rows:
    k = kind(r["Kernel_Name"])
    big[k] = max(big.get(k, 0), int(r["Grid_Size"]))
acc = {}      # (kernel, counter) -> [sum, dispatches]
dur = {}      # kernel -> [sum_ns, n]
for f, r in rows:
    k = kind(r["Kernel_Name"])
    if int(r["Grid_Size"]) != big[k]:
        continue
    a = acc.setdefault((k, r["Counter_Name"]), [0.0, set()])
    a[0] += float(r["Counter_Value"])
    a[1].add((f, r["Dispatch_Id"]))
    d = dur.setdefault(k, {})
    d[(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
dur = {k: [sum(v.values()), len(v)] for k, v in dur.items()}
res = {"command": __doc__.split("Each pass:")[1].split("FETCH_SIZE")[0].strip(), "kernels": {}}
for k in ("post_kernel", "ksx_kernel"):
    per = {c: v[0] / max(1, len(v[1])) for (kk, c), v in acc.items() if kk == k}
    e = {"counters_per_launch": per}
    if k in dur:
        e["duration_ms_under_pmc"] = dur[k][0] / dur[k][1] / 1e6
    if "FETCH_SIZE" in per:
        e["hbm_read_bytes_corrected"] = per["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in per:
        e["hbm_write_bytes"] = per["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    if "GRBM_GUI_ACTIVE" in per and k in dur:
        e["effective_clock_GHz"] = per["GRBM_GUI_ACTIVE"] / 8 / (dur[k][0] / dur[k][1])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "GRBM_GUI_ACTIVE" in per:
        # MFMA-busy cycles summed over the 1024 SIMDs against the cycles the kernel was resident (GUI_ACTIVE is
        # summed over the 8 XCDs)
        e["mfma_busy_frac_of_simd_cycles"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (per["GRBM_GUI_ACTIVE"] / 8 * 1024)
    res["kernels"][k] = e
res.update({"rows_per_launch": 262144, "n_obs": 2048, "d": 32,
            "note": (sys.argv[3] if len(sys.argv) > 3 else "") + " FETCH_SIZE/WRITE_SIZE are in KiB and count the L2's fabric-side "
                    "requests (Infinity-Cache hits included: profiles/r02_workspace_experiment.json); FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE taken as is; made by "
                    "tools/pmc_summary.py"})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
