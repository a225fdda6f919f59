"""Average duration per kernel from a rocprofv3 --kernel-trace CSV directory: python tools/kernel_times.py DIR"""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size_X", "?")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    print("%-100s n=%3d avg %.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
