#!/bin/bash
# usage: tools/pmc_layer.sh "cin cout k stride H W" COUNTER... ; one rocprofv3 --pmc pass per counter on one conv layer
set -e
shape=$1; shift
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for c in "$@"; do
  out=gpurun_out/pmc_layer_$c
  rm -rf "$out"
  rocprofv3 --pmc "$c" --output-format csv -d "$out" -o r -- python3 tools/one_conv.py $shape 4 > "$out.log" 2>&1 || true
  f=$(find "$out" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_mfma" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
v = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], sum(v) / max(len(v), 1), "launches", len(v), rows[0]["Kernel_Name"][:50] if rows else "")
PY
done
