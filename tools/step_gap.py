"""From a rocprofv3 kernel trace of bench.py: the idle time between the last kernel of one replayed step and the first
kernel of the next (graph-to-graph launch gap), next to the per-step kernel time:  python tools/step_gap.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_multi_dev" in r["Kernel_Name"]]
if len(ends) < 141:      # the update applied by the backward kernels (bench.py's default at N = 1): a step ends with the three-product launch
    ends = [i for i, r in enumerate(rows) if "sgemm_triple_wgrad2_xgrad" in r["Kernel_Name"]]
gaps, spans = [], []
for a, b in zip(ends[40:140], ends[41:141]):
    if b - a > 12:
        continue
    gaps.append(int(rows[a + 1]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]))
    spans.append(int(rows[b]["End_Timestamp"]) - int(rows[a + 1]["Start_Timestamp"]))
gaps.sort()
spans.sort()
n = len(gaps)
print("steps %d: gap between steps median %.2f us (min %.2f, p90 %.2f); step span median %.2f us -> %.2f us per step"
      % (n, gaps[n // 2] / 1e3, gaps[0] / 1e3, gaps[int(n * .9)] / 1e3, spans[n // 2] / 1e3, (gaps[n // 2] + spans[n // 2]) / 1e3))
