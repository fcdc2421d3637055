"""The launches of the LAST `n` kernels of a rocprofv3 --kernel-trace run as a timeline (start offset, duration, name, grid).
usage: python tools/timeline.py <dir given to rocprofv3 -d> [n]"""
import csv
import glob
import sys

root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f us  +%6.1f us  %-60s grid %s x %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                       r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"]))
