#!/bin/bash
# dev aid (GPU box): stage times of the scan kernel for each development build given as tag
for tag in "$@"; do
  for fmt in sc16 uc8; do
    MODES_LIB=$GRAFT_REPO_ROOT/dump1090_amd/libmodes_hip_$tag.so timeout -k 10 100 python tools/stage_time.py $fmt 2>/dev/null | awk -v t=$tag '{printf "%s %s %s %s | ", t, $1, $3, $5} END {print ""}' || exit 1
  done
done
