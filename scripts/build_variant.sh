#!/bin/bash
# A build of libfalcon_amd.so with extra compiler flags on k_align2.hip, under gpurun_variants/<name>/ (travels to the
# GPU box; FALCON_AMD_LIB selects it).   usage: scripts/build_variant.sh <name> [flags...]   (SRC=<tree> for another tree)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
SRC=${SRC:-$R}
D=$R/gpurun_variants/$NAME; mkdir -p $D
cd $SRC/falcon_amd/csrc
[ "$SRC" != "$R" ] && make -s -j8 >/dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-pass-failed "$@" -c k_align2.hip -o $D/k_align2.o
OBJS=$(ls *.o | grep -v '^k_align2.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libfalcon_amd.so $D/k_align2.o $OBJS
rm -f $D/k_align2.o
echo built $D/libfalcon_amd.so
