"""Timeline of the LAST training step in a rocprofv3 kernel trace (steps are delimited by pack_batch_kernel): per queue (= HIP stream) the
busy time and the idle gaps, the union-busy time of the GPU, and the largest kernels per queue.  Tells a host-bound step (gaps between short
kernels on every queue) from a GPU-bound one.  python tools/step_timeline.py kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "pack_batch_kernel" in r["Kernel_Name"]]
a, b = marks[-2] + 1, marks[-1] + 1
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
qkey = "Queue_Id" if "Queue_Id" in step[0] else ("Stream_Id" if "Stream_Id" in step[0] else None)
print(f"last step: {len(step)} kernels, span {(t1 - t0) / 1e6:.2f} ms, queue column: {qkey}")
byq = collections.defaultdict(list)
for r in step:
    byq[r[qkey] if qkey else "0"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv = sorted((s, e) for v in byq.values() for s, e, _ in v)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"GPU busy (union over queues) {busy / 1e6:.2f} ms, idle {((t1 - t0) - busy) / 1e6:.2f} ms")
for q, v in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    v.sort()
    kb = sum(e - s for s, e, _ in v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    big = sum(g for g in gaps if g > 20000)
    print(f"queue {q}: {len(v)} kernels, busy {kb / 1e6:.2f} ms, first start +{(v[0][0] - t0) / 1e6:.2f} ms, last end +{(v[-1][1] - t0) / 1e6:.2f} ms, "
          f"gaps > 20 us: {sum(g > 20000 for g in gaps)} totalling {big / 1e6:.2f} ms, median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us")
    tot = collections.Counter()
    for s, e, n in v:
        tot[n.replace('(anonymous namespace)::', '').replace('void ', '')[:70]] += e - s
    for n, t in tot.most_common(6):
        print(f"      {t / 1e6:7.3f} ms  {n}")
