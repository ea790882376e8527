#!/bin/bash
# A second build of the four kernel libraries in which the H = 256 files (cnf_wide.hip, cnf_wide_bwd.hip) are compiled with
# the max-ILP instruction scheduler, into gan-flow_amd/variants_ilp/ (git-ignored; CNF_HIP_LIBDIR points the loader at it).
# The default build's objects of the other files are reused, so run `make -C gan-flow_amd/csrc` first.
# usage: tools/build_wide_ilp.sh [extra -D flags for the two files]
set -e
cd "$(dirname "$0")/../gan-flow_amd/csrc"
OUT=../variants_ilp
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp $*"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
build_one() {   # $1: suffix ("" | .softplus | .elu | .relu), $2: -DCNF_ACT=..., $3: library name
  /opt/rocm/bin/hipcc $FLAGS $2 -c cnf_wide.hip -o $OUT/cnf_wide$1.o &
  /opt/rocm/bin/hipcc $FLAGS $NOPK $2 -c cnf_wide_bwd.hip -o $OUT/cnf_wide_bwd$1.o &
  wait
  objs=""
  for o in cnf_fwd cnf_bwd cnf_split cnf_split_fwd cnf_split16 cnf_train cnf_data cnf_api; do objs="$objs $o$1.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$3 $objs $OUT/cnf_wide$1.o $OUT/cnf_wide_bwd$1.o
  echo built $OUT/$3
}
build_one "" "" libcnf_hip.so &
build_one .softplus -DCNF_ACT=1 libcnf_hip_softplus.so &
wait
build_one .elu -DCNF_ACT=2 libcnf_hip_elu.so &
build_one .relu -DCNF_ACT=3 libcnf_hip_relu.so &
wait
