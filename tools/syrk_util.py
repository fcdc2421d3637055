"""MFMA utilisation of the Cholesky trailing updates from a rocprofv3 kernel-trace of tools/fit_once.py.
usage: syrk_util.py <trace-dir> <N>; the trace must come from the pair schedule with stand-alone update launches
(B7_POTRF_SCHED=0 B7_POTRF_DEFER=0 in the environment of fit_once.py).  Launch order per pair of panels (group size 2): narrow (K = 64, one block
column), bulk (K = 128, everything right of the pair); tiles = workgroups; flop = tiles * 64*64*K*2."""
import csv
import glob
import sys
PEAK = 78.6e12
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
rows.sort()
start = [i for i, r in enumerate(rows) if "copy_lower" in r[2]][-1]
syrk = [(e - s, g) for s, e, n, g in rows[start:] if "potrf_syrk" in n]
N = int(sys.argv[2])
nb = (N + 127) // 128 * 2
tot_f = tot_t = bulk_f = bulk_t = 0.0
best = 0.0
for i, (dt, tiles) in enumerate(syrk):
    narrow = (i % 2 == 0)
    K = 64 if narrow else 128
    fl = tiles * 64 * 64 * K * 2.0
    tot_f += fl
    tot_t += dt * 1e-9
    if not narrow:
        bulk_f += fl
        bulk_t += dt * 1e-9
        best = max(best, fl / (dt * 1e-9) / PEAK)
print("N = %d: %d trailing-update launches, %.2f GFLOP in %.1f us" % (N, len(syrk), tot_f / 1e9, tot_t * 1e6))
print("  MFMA utilisation: all launches %.1f %%, K=128 launches %.1f %%, best launch %.1f %%" %
      (100 * tot_f / tot_t / PEAK, 100 * bulk_f / bulk_t / PEAK, 100 * best))
