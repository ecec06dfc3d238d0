#!/bin/bash
# One GPU call: bench line, rocprofv3 kernel stats, and the two PMC passes for HBM traffic.
# usage: tools/profile_round.sh <tag>      (outputs under gpurun_out/<tag>/)
set -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o b -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
ls $out $out/pmc_fetch
