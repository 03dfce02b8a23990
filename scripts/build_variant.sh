#!/bin/bash
# A build of libfalcon_amd.so with extra compiler flags on one source (FILE=k_chain.hip; default k_align2.hip), under gpurun_variants/<name>/ (travels to the
# GPU box; FALCON_AMD_LIB selects it).   usage: scripts/build_variant.sh <name> [flags...]   (SRC=<tree> for another tree)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
SRC=${SRC:-$R}
FILES=${FILE:-k_align2.hip}   # (several: FILE="k_chain.hip k_seed_index.hip" -- the flags go to each)
D=$R/gpurun_variants/$NAME; mkdir -p $D
cd $SRC/falcon_amd/csrc
[ "$SRC" != "$R" ] && make -s -j8 >/dev/null 2>&1
MINE=""; SKIP=""
for f in $FILES; do   # (a path with a slash: that file instead of the tree's source of the same name, e.g. an older revision)
  o=$(basename ${f%.hip}.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-pass-failed -I. "$@" -c $f -o $D/$o
  MINE="$MINE $D/$o"; SKIP="$SKIP -e ^$o\$"
done
OBJS=$(ls *.o | grep -v $SKIP)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libfalcon_amd.so $MINE $OBJS
rm -f $MINE
echo built $D/libfalcon_amd.so
