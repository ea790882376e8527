#!/bin/bash
# Builds timing-experiment variants of libcnf_hip.so into gan-flow_amd/variants/ (git-ignored): each variant compiles
# cnf_split.hip with one or more -DCNF_EXP_* switches (results are wrong on purpose; only kernel time matters).
# usage: tools/build_variants.sh NAME "-DFLAG1 -DFLAG2" [NAME2 "..."] ...
set -e
cd "$(dirname "$0")/../gan-flow_amd/csrc"
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -c cnf_split.hip -o ../variants/cnf_split_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcnf_hip_$name.so cnf_fwd.o cnf_bwd.o ../variants/cnf_split_$name.o cnf_wide.o cnf_wide_bwd.o cnf_train.o cnf_api.o
  rm -f ../variants/cnf_split_$name.o
  echo built $name
done
