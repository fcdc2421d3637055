"""Timeline of the LAST fit in a rocprofv3 kernel-trace CSV: per kernel name count, busy time, and the idle gaps
between consecutive kernels.  usage: trace_timeline.py <dir-with-*_kernel_trace.csv>"""
import csv
import glob
import sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last fit starts at the last copy_lower_kernel (jitter-free fit: one potrf attempt) preceded by kxx
idx = [i for i, r in enumerate(rows) if "copy_lower" in r[2]]
start = idx[-1]
while start > 0 and ("kxx" in rows[start - 1][2] or "prep_obs" in rows[start - 1][2] or "k_generic" in rows[start - 1][2]):
    start -= 1
fit = rows[start:]
t0, t1 = fit[0][0], fit[-1][1]
print("kernels in last fit: %d, span %.1f us" % (len(fit), (t1 - t0) / 1e3))
agg = {}
gap_total = 0
prev_end = None
for s, e, n in fit:
    key = n.replace("void ", "").replace("(anonymous namespace)::", "")
    key = key.split("(")[0]
    if "potrf_diag_kernel" in key or "potrf_syrk_kernel" in key:
        key = key.replace(", false", "").replace("false", "")
    else:
        key = key.split("<")[0]
    a = agg.setdefault(key, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    if prev_end is not None:
        a[2] += s - prev_end
        gap_total += s - prev_end
    prev_end = e
for k, (n, busy, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s x%3d  busy %8.1f us (avg %6.2f)  gap before %7.1f us (avg %5.2f)" % (k[:34], n, busy / 1e3, busy / n / 1e3, gap / 1e3, gap / n / 1e3))
print("total busy %.1f us, total gaps %.1f us" % (sum(a[1] for a in agg.values()) / 1e3, gap_total / 1e3))
