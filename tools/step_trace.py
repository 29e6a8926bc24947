"""Print the kernel sequence of ONE steady-state MLP training step from a rocprofv3 kernel trace csv:
    python tools/step_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step of the flat-bucket path ends with its ONE multi-tensor optimizer launch; the per-parameter path has four
marks = [i for i, n in enumerate(names) if "adam_multi_dev" in n]
if len(marks) < 60:
    # the update applied by the backward kernels (bench.py's default at N = 1): a step ends with the launch that makes dW2, dW1 and dx
    marks = [i for i, n in enumerate(names) if "sgemm_triple_wgrad2_xgrad" in n]
if len(marks) < 60:
    marks = [i for i, n in enumerate(names) if "adam" in n][3::4]
s, e = marks[50] + 1, marks[51] + 1
tot = 0
for r in rows[s:e]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print("%-88s grid=%-8s %6.1f us" % (r["Kernel_Name"][:88].replace("void lg::", "").replace("lg::", ""), r["Grid_Size_X"], d / 1e3))
span = (int(rows[e - 1]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3
print("kernels per step: %d   sum of kernel time: %.1f us   first-start to last-end: %.1f us" % (e - s, tot / 1e3, span))
