#!/bin/bash
# Build a whole-library measurement variant: tools/build_variant.sh <name> <file.hip> [-DMACRO ...]
# -> morig_amd/lib/variants/lib_<name>.so (select with MORIG_HIP_LIB). The other objects come from the regular build.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../morig_amd/csrc"
make -s >/dev/null
mkdir -p ../lib/variants /tmp/morig_variants
obj=/tmp/morig_variants/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $src -o $obj
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -ldl -o ../lib/variants/lib_$name.so
echo built lib_$name.so
