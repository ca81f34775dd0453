#!/bin/bash
# usage: tools/pmc_dcn.sh COUNTER... ; one rocprofv3 --pmc pass per counter on the fused DCN forward at 1080p
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for c in "$@"; do
  out=gpurun_out/pmc_dcn_$c
  rm -rf "$out"
  rocprofv3 --pmc "$c" --output-format csv -d "$out" -o r -- python3 tools/one_dcn.py 4 > "$out.log" 2>&1 || true
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dcn_" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
v = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], sum(v) / max(len(v), 1), "launches", len(v))
PY
done
