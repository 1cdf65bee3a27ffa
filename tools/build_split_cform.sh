#!/bin/bash
# Build the measurement variant for tools/gpu_split_mix_ab.sh: edge_pp.hip and tile_gemm.hip with -DMORIG_SPLIT_C_FORM (the fp32 ->
# (hi, lo) conversion as the plain C expression instead of v_fma_mix) -> morig_amd/lib/variants/lib_split_cform.so
set -e
cd "$(dirname "$0")/../morig_amd/csrc"
make -s >/dev/null
mkdir -p ../lib/variants /tmp/morig_variants
for f in edge_pp tile_gemm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DMORIG_SPLIT_C_FORM -c $f.hip -o /tmp/morig_variants/cform_$f.o
done
others=$(ls *.o | grep -v "^edge_pp.o$\|^tile_gemm.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/morig_variants/cform_edge_pp.o /tmp/morig_variants/cform_tile_gemm.o -o ../lib/variants/lib_split_cform.so
echo built lib_split_cform.so
