#!/bin/bash
# run the differential campaign with each library variant under build_variants/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp falcon_amd/libfalcon_amd.so /tmp/lib_head.so
for v in /tmp/lib_head.so build_variants/*.so; do
  cp $v falcon_amd/libfalcon_amd.so
  echo "== $v"; timeout 200 python scripts/gpu_differential_campaign.py piles 0 72 2>&1 | tail -1 | cut -c1-260
done
cp /tmp/lib_head.so falcon_amd/libfalcon_amd.so
