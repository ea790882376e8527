#!/bin/bash
# Builds timing-experiment variants of libcnf_hip.so into gan-flow_amd/variants/ (git-ignored): each variant compiles
# cnf_split.hip with one or more -DCNF_EXP_* switches (results are wrong on purpose; only kernel time matters).
# usage: [SRC=cnf_fwd] tools/build_variants.sh NAME "-DFLAG1 -DFLAG2" [NAME2 "..."] ...   (SRC: which file gets the flags)
set -e
cd "$(dirname "$0")/../gan-flow_amd/csrc"
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
SRC=${SRC:-cnf_split}
# the files the Makefile compiles without packed-fp32 instructions (NOPK_FLAGS there)
case $SRC in cnf_fwd|cnf_wide_bwd) FLAGS="$FLAGS -Xclang -target-feature -Xclang -packed-fp32-ops";; esac
# the files the Makefile compiles with the max-ILP scheduler (ILP_FLAGS there)
case $SRC in cnf_split_fwd) FLAGS="$FLAGS -mllvm -amdgpu-sched-strategy=max-ilp";; esac
# cnf_split goes through the hazard pass in the Makefile; a variant of it built here does NOT (timing experiments only)
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -c $SRC.hip -o ../variants/${SRC}_$name.o
  objs=""
  for o in cnf_fwd cnf_bwd cnf_split cnf_split_fwd cnf_split16 cnf_wide cnf_wide_bwd cnf_train cnf_data cnf_api; do
    if [ $o = $SRC ]; then objs="$objs ../variants/${SRC}_$name.o"; else objs="$objs $o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcnf_hip_$name.so $objs
  rm -f ../variants/${SRC}_$name.o
  echo built $name
done
