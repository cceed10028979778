#!/bin/bash
# Diagnostic library with in-kernel cycle stamps (never shipped): build/librfi_diag.so = the product's objects with the
# named sources rebuilt under -DRFI_DIAG_STAMPS=1.  Select it with RFI_HIP_LIB=build/librfi_diag.so.
#   tools/build_diag.sh conv_ws.hip [more sources]
set -e
cd "$(dirname "$0")/.."
python -m rfi_toolbox_amd.build
objs=$(ls build/obj/*.o)
for s in "$@"; do
  o=build/${s%.*}_diag.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DRFI_DIAG_STAMPS=1 -x hip -c rfi_toolbox_amd/csrc/$s -o $o
  objs=$(echo "$objs" | grep -v "/${s//./_}.o")
  objs="$objs"$'\n'"$o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o build/librfi_diag.so -ldl
echo built build/librfi_diag.so
