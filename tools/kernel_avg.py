"""average duration per kernel name from a rocprofv3 kernel-trace csv: python tools/kernel_avg.py <csv> [substring]"""
import collections
import csv
import sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:100], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-100s grid=%-8s n=%-4d median %7.2f us  min %7.2f" % (k.replace("void lg::", ""), grid, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
