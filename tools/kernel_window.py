"""Print a window of consecutive kernels from the middle of a rocprofv3 kernel-trace csv, with durations and the idle gaps
between them:  python tools/kernel_window.py <kernel_trace.csv> [count] [name-substring to start at]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
start = len(rows) // 2
if len(sys.argv) > 3:
    start = next(i for i in range(start, len(rows)) if sys.argv[3] in rows[i]["Kernel_Name"])
prev, tot = None, 0
for r in rows[start:start + count]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    gap = (int(r["Start_Timestamp"]) - prev) / 1e3 if prev is not None else 0.0
    prev = int(r["End_Timestamp"])
    tot += d
    print("%-92s grid=%-7s %6.1f us   idle before %5.1f us" % (r["Kernel_Name"][:92].replace("void lg::", "").replace("lg::", ""), r["Grid_Size_X"], d / 1e3, gap))
span = (int(rows[start + count - 1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])) / 1e3
print("%d kernels: %.1f us of kernel time, %.1f us from first start to last end" % (count, tot / 1e3, span))
