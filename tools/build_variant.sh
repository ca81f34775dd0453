#!/bin/bash
# A/B builds of ONE kernel file: the shipped library with FILE.hip.o replaced by a build with extra -D flags, written to
# tdvc_amd/variants/libtdvc_NAME.so (git-ignored *.so; they travel with gpurun).  Use with TDVC_LIB=... (tools/ab_row.py, tools/bench_pair.py).
# usage: tools/build_variant.sh FILE NAME "-DFLAG=0 ..."
set -e -o pipefail
file="$1"; name="$2"; flags="$3"
cd "$(dirname "$0")/../tdvc_amd/csrc"
make -j6 > /dev/null
mkdir -p ../variants ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  -fno-honor-nans $flags -c $file.hip -o ../../build/variants/${file}_$name.o 2> >(sed '/packed-fp32-ops.*not a recognized feature/d' >&2)
objs=$(ls ../../build/csrc/*.o | grep -v "/$file.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libtdvc_$name.so $objs ../../build/variants/${file}_$name.o
echo "built tdvc_amd/variants/libtdvc_$name.so ($file: $flags)"
