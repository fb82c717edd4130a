#!/bin/bash
# On the GPU box: kernel-trace stats + HBM traffic counters for the default bench.  usage: tools/profile_round.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/trace_err.log || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
tools/pmc_pass.sh $out/fetch "FETCH_SIZE" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline || exit 1
tools/pmc_pass.sh $out/write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline || exit 1
python3 tools/pmc_summary.py $out/fetch delta > $out/pmc_fetch.txt
python3 tools/pmc_summary.py $out/write delta > $out/pmc_write.txt
head -4 $out/kernel_stats.csv | cut -c1-200; cat $out/pmc_fetch.txt $out/pmc_write.txt
