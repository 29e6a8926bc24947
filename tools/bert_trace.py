"""kernel summary of ONE steady-state tiny-BERT iteration from a rocprofv3 kernel trace of `bench.py` or examples/bert.py:
    python tools/bert_trace.py <kernel_trace.csv> [--list]"""
import collections
import csv
import sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "gather_rows" in n]
starts = [m for k, m in enumerate(marks) if k == 0 or m - marks[k - 1] > 20]      # first gather of each iteration
s, e = starts[-2], starts[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[s:e]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"].replace("void lg::", "").replace("lg::", "").split("(")[0][:60]
    agg[k][0] += 1
    agg[k][1] += d
    if "--list" in sys.argv:
        print("%-70s %8.1f us grid=%s" % (k, d, r["Grid_Size_X"]))
tot = sum(v[1] for v in agg.values())
print("kernels per iteration: %d   sum of kernel time: %.1f us   span: %.1f us" % (e - s, tot, (int(rows[e]["Start_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%-62s n=%-4d %8.1f us  %5.1f%%" % (k, n, t, 100 * t / tot))
