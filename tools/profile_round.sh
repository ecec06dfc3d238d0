#!/bin/bash
# One GPU call: bench line, rocprofv3 kernel trace (summarised over the TIMED steps only), and the two PMC passes for HBM traffic.
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/<tag>/)
set -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline > $out/trace.log 2>&1 || exit 1
python3 tools/prof_summary.py $out/trace/b_kernel_trace.csv 5 3 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline; the 5 TIMED steps only (warm-up, one-off weight packing and the live-profiled steps cut away)" > $out/bench_n1_kernel_summary.txt || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $out/pmc_fetch/b_counter_collection.csv $out/pmc_write/b_counter_collection.csv "void lsam::tailpair_kernel<14, 28" $out/traffic.json || exit 1
python3 tools/pmc_per_kernel.py $out/pmc_fetch/b_counter_collection.csv $out/pmc_write/b_counter_collection.csv > $out/pmc_hbm_per_kernel.txt || exit 1
head -12 $out/bench_n1_kernel_summary.txt
