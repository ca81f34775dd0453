#!/bin/bash
# usage: tools/pmc_pair.sh TAG "CTR CTR ..." ["CTR CTR ..."]... ; one rocprofv3 --pmc pass per quoted group on the fused
# 2 x (3x3 64->64) pair at 1088x1920 (or SIZE="H W"); prints per-launch averages for conv_pair_kernel
set -e
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  out=gpurun_out/pmc_${tag}_g$i
  rm -rf "$out"
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o r -- python3 tools/one_pair.py ${SIZE:-1088 1920} 6 > "$out.log" 2>&1
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_pair" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v) / len(v), "launches", len(v))
PY
  i=$((i+1))
done
