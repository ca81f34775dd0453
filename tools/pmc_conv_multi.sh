#!/bin/bash
# usage: tools/pmc_conv_multi.sh TAG "CTR CTR ..." ["CTR CTR ..."]... ; one rocprofv3 --pmc pass per quoted group (<= 8 SQ
# counters per pass) on the 3x3 64->64 1080p layer (or LAYER="cin cout k stride H W"); prints per-launch averages for the conv kernel
set -e
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  out=gpurun_out/pmc_${tag}_g$i
  rm -rf "$out"
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o r -- python3 tools/one_conv.py ${LAYER:-64 64 3 1 1088 1920} 6 > "$out.log" 2>&1
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
name = ""
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_mfma" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        name = r["Kernel_Name"][:48]
for k, v in acc.items():
    print(k, sum(v) / len(v), "launches", len(v), name)
PY
  i=$((i+1))
done
