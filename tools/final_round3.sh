#!/bin/bash
# Round-3 closing run on the MI355X box: the whole GPU suite, tools/final_profiles.sh TAG, rocprofv3 kernel stats of the rk4
# training leg, and the two-rank rehearsal of bench.py on the one GPU.  usage: tools/final_round3.sh TAG
tag=${1:-r03h}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/test.log 2>&1 || { tail -20 $out/test.log; exit 1; }
tail -2 $out/test.log
bash tools/final_profiles.sh $tag || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_rk4 -o ${tag}_rk4 -- python3 bench.py --steps 30 --warmup 5 --extras rk4 > $out/bench_rk4_under_rocprof.json 2> $out/rocprof_rk4_err.log || exit 1
BENCH_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 > $out/bench_2ranks.json 2> $out/bench_2ranks_err.log || exit 1
echo done
