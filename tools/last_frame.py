"""Per-kernel totals of the LAST P-frame in a rocprofv3 kernel trace (frames are delimited by patch_match_kernel, one per frame).
usage: python3 tools/last_frame.py <kernel_trace.csv> [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "patch_match_kernel" in r["Kernel_Name"]]
a, b = marks[-2] + 1, marks[-1] + 1
step = rows[a:b]
tot = collections.Counter()
cnt = collections.Counter()
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:100]
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
wall = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
print(f"last frame: {len(step)} kernels, busy {sum(tot.values()) / 1e6:.3f} ms, span {wall / 1e6:.3f} ms")
for n, t in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    print(f"{t / 1e6:8.3f} ms  n={cnt[n]:4d}  avg={t / cnt[n] / 1e3:8.1f} us  {n}")
