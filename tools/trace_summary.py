"""Per-(kernel, grid) durations and the launch timeline gaps of a rocprofv3 kernel trace (csv): python tools/trace_summary.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
gaps = []
prev_end = None
for r in rows:
    t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and t0 - prev_end < 50_000:  # inside one enqueued sequence
        gaps.append((t0 - prev_end) / 1e3)
    prev_end = t1
    n = r["Kernel_Name"]
    if flt in n:
        d[(n.split("(")[0][-52:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((t1 - t0) / 1e3)
for k in sorted(d, key=lambda k: -sum(d[k])):
    v = d[k]
    print(f"{k[0]:54s} grid {k[1]:>7s} {k[2]:>4s} {k[3]:>4s}  calls {len(v):5d}  total {sum(v) / 1e3:8.3f} ms  avg {sum(v) / len(v):7.1f} us  min {min(v):6.1f}  max {max(v):6.1f}")
if gaps:
    gaps.sort()
    print(f"gaps between consecutive kernels (< 50 us): n {len(gaps)}  median {gaps[len(gaps) // 2]:.2f} us  mean {sum(gaps) / len(gaps):.2f} us  total {sum(gaps) / 1e3:.3f} ms")
