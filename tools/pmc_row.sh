#!/bin/bash
# usage: tools/pmc_row.sh TAG "CTR CTR ..." ["CTR CTR ..."]... ; one rocprofv3 --pmc pass per quoted group on a single conv launch loop
# (CONV="cin cout k stride H W", default the 3x3 128->128 @544x960 layer); prints per-launch averages for conv_row_kernel
set -e
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  out=gpurun_out/pmc_${tag}_g$i
  rm -rf "$out"
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o r -- python3 tools/one_conv.py ${CONV:-128 128 3 1 544 960} 6 > "$out.log" 2>&1
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_row" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v) / len(v), "launches", len(v))
PY
  i=$((i+1))
done
