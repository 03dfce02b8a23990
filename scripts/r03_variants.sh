#!/bin/bash
# Builds variants of the library that differ in k_align2's policy constants
# (gpurun_variants/libfalcon_amd_<name>.so), for scripts/r03_sweep.sh on the GPU box.
#   bash scripts/r03_variants.sh name:"-DA2_FREE_JOIN=10 ..." ...
set -e
cd "$(dirname "$0")/../falcon_amd/csrc"
mkdir -p ../../gpurun_variants
OBJS="k_pack_index.o k_chain.o k_trimwin.o k_align.o k_align_wide.o k_msa.o engine.o legacy_abi.o reader.o fasta.o pack_host.o"
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags -c k_align2.hip -o /tmp/k_align2_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/libfalcon_amd_$name.so $OBJS /tmp/k_align2_$name.o
    echo "built $name ($flags)"
done
