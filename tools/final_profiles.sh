#!/bin/bash
# Round-end measurement set (run on the MI355X box from the repo root): HBM-traffic and SQ counter passes over
# tools/time_kernels.py (each --pmc pass a run of its own), then the default bench line and the rocprofv3 kernel stats of
# the training bench (both read the traffic figure of the passes before them).  usage: tools/final_profiles.sh TAG
set -e
tag=${1:-r01f}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 tools/time_kernels.py > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 tools/time_kernels.py > $out/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq -o s -- python3 tools/time_kernels.py > $out/sq.log 2>&1
python3 tools/bwd_traffic.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv $out/bwd_traffic.json > $out/bwd_traffic.log 2>&1
cp $out/bwd_traffic.json profiles/${tag}_bwd_traffic.json
python3 tools/pmc_summary.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv > $out/hbm_pmc_summary.txt 2>&1 || true
python3 tools/pmc_summary.py $out/sq/s_counter_collection.csv > $out/sq_pmc_summary.txt 2>&1 || true
t0=$(date +%s)
timeout -k 10 600 python -u bench.py > $out/bench_line.json 2> $out/bench_err.log
echo "default bench.py wall seconds: $(( $(date +%s) - t0 ))" > $out/bench_time.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ${tag}_train -- python3 bench.py --steps 30 --warmup 5 --extras roofline > $out/bench_under_rocprof.json 2> $out/rocprof_err.log
cat $out/bench_time.txt
find $out -name "*.csv"
