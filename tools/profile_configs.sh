#!/bin/bash
# On the GPU box: kernel-trace stats and HBM traffic counters for the secondary configurations
# (dictionary pages, run-heavy keys, nested V1 pages, bit-unpack sweep).  usage: tools/profile_configs.sh <tag> [pages]
tag=$1; pages=${2:-8000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profcfg_$tag; mkdir -p $out
timeout -k 10 600 python3 tools/bench_configs.py --pages $pages > $out/configs.jsonl 2> $out/configs.err || { tail -3 $out/configs.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_configs.py --only c1,c1b,c4,plain,bytes,walk,levels --pages $pages > $out/under_rocprof.jsonl 2> $out/trace_err.log || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
tools/pmc_pass.sh $out/fetch "FETCH_SIZE" python3 tools/bench_configs.py --only c1,c1b,c4 --pages $pages || exit 1
tools/pmc_pass.sh $out/write "WRITE_SIZE" python3 tools/bench_configs.py --only c1,c1b,c4 --pages $pages || exit 1
python3 tools/pmc_summary.py $out/fetch wave_kernel > $out/pmc_fetch.txt
python3 tools/pmc_summary.py $out/write wave_kernel > $out/pmc_write.txt
cut -c1-220 $out/configs.jsonl; head -6 $out/kernel_stats.csv | cut -c1-160; cat $out/pmc_fetch.txt $out/pmc_write.txt
