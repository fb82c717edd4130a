#!/bin/bash
# On the GPU box: everything profiles/r02 holds.  usage: tools/profile_r02.sh   (results under gpurun_out/prof_r02)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r02; rm -rf $out; mkdir -p $out
step() { echo "== $1" >> $out/progress.log; }
step bench; timeout -k 10 400 python3 bench.py > $out/bench_c2_full.json 2> $out/bench.err || { tail -3 $out/bench.err; exit 1; }
step trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/delta_c2_bench_under_rocprof.json 2> $out/trace_err.log || exit 1
cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" $out/delta_c2_kernel_stats.csv || exit 1
step fetch; tools/pmc_pass.sh $out/fetch "FETCH_SIZE" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline || exit 1
step write; tools/pmc_pass.sh $out/write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline || exit 1
python3 tools/pmc_summary.py $out/fetch delta_wave > $out/pmc_fetch.txt
python3 tools/pmc_summary.py $out/write delta_wave > $out/pmc_write.txt
step configs; timeout -k 10 600 python3 tools/bench_configs.py --pages 8000 > $out/configs_8000pages.jsonl 2> $out/configs.err || { tail -3 $out/configs.err; exit 1; }
step large; timeout -k 10 600 python3 tools/bench_configs.py --only c2seg,c4big > $out/configs_large_pages.jsonl 2> $out/large.err || { tail -3 $out/large.err; exit 1; }
step segtrace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/segtrace -- python3 tools/bench_configs.py --only c2seg,c1 --pages 8000 > $out/segments_under_rocprof.jsonl 2> $out/segtrace_err.log || exit 1
cp "$(find $out/segtrace -name '*kernel_stats.csv' | head -1)" $out/segments_kernel_stats.csv || exit 1
step cfgtrace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfgtrace -- python3 tools/bench_configs.py --only c1,c1b,c4,plain,bytes,walk,levels --pages 8000 > $out/configs_under_rocprof.jsonl 2> $out/cfgtrace_err.log || exit 1
cp "$(find $out/cfgtrace -name '*kernel_stats.csv' | head -1)" $out/configs_kernel_stats.csv || exit 1
step chunks; timeout -k 10 300 python3 tools/bench_chunks.py > $out/c5_one_gpu.json 2> $out/c5.err || { tail -3 $out/c5.err; exit 1; }
step done
rm -rf $out/trace $out/segtrace $out/cfgtrace
cat $out/pmc_fetch.txt $out/pmc_write.txt; head -4 $out/delta_c2_kernel_stats.csv | cut -c1-200
