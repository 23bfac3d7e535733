#!/bin/bash
# A/B of library variants on one box: scripts/astar_ab.sh <out dir> <reads> <seeds> <product seeds> variant [variant ...]   ("default" = the shipped library)
out=$1; reads=$2; seeds=$3; prod=$4; shift 4
mkdir -p "$out"
export MGTA_ASTAR_GROUP=8
for v in "$@"; do
  if [ "$v" = default ]; then unset MEGAGTA_HIP_LIB; else export MEGAGTA_HIP_LIB=$PWD/megagta_amd/libmegagta_hip_$v.so; fi
  echo "== $v" | tee -a "$out/summary.log"
  timeout -k 10 420 python scripts/astar_ab.py "$reads" "$seeds" "$prod" 1 > "$out/$v.log" 2>&1 || { echo "variant $v failed"; tail -n 5 "$out/$v.log"; exit 1; }
  grep -E "COLD|PRODUCT|astar-prof|LONE" "$out/$v.log" | tail -n 30 | tee -a "$out/summary.log"
done
