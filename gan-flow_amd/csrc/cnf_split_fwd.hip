// cnf_split_fwd.hip -- the forward solve kernels of cnf_split.hip (dopri5_split_kernel, launch_split_fwd) as a
// translation unit of their own, so that they can be compiled with another instruction scheduler than the adjoint
// kernels (Makefile: -mllvm -amdgpu-sched-strategy=max-ilp).  All of the code is in cnf_split.hip.
#define CNF_SPLIT_PART 1
#include "cnf_split.hip"
