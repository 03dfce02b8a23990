#!/bin/bash
# Builds variants of the library that differ in k_align2's placement policy (A2_FREE_MIN: free lanes two
# running tracks need to stay paired; A2_FREE_JOIN: ... to form a pair; A2_LOOK_EVERY) into gpurun_variants/
# (git-ignored; they travel to the GPU box).  usage: scripts/r05_policy_build.sh "0:6:8 2:6:8 4:6:8 ..."
cd $(dirname $0)/../falcon_amd/csrc
mkdir -p ../../gpurun_variants
OBJS="k_pack.o k_seed_index.o k_chain.o k_trimwin.o k_align.o k_align2_shadow.o k_align_wide.o k_msa.o k_links2.o k_score1.o k_score2.o engine.o legacy_abi.o reader.o fasta.o pack_host.o"
for v in $1; do
  IFS=: read m j l <<< "$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DA2_FREE_MIN=$m -DA2_FREE_JOIN=$j -DA2_LOOK_EVERY=$l -c k_align2.hip -o /tmp/k_align2_$m.$j.$l.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/libfalcon_amd_m${m}j${j}l${l}.so /tmp/k_align2_$m.$j.$l.o $OBJS
  echo built m$m j$j l$l
done
