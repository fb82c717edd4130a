#!/bin/bash
# On the GPU box: everything profiles/r03 holds, in two calls (a box serves 20 minutes at most).
# usage: tools/profile_r03.sh main | c3        (results under gpurun_out/prof_r03 and prof_r03_c3)
# Writes the sha256 of the library the passes run on next to them; tools/collect_profiles.py refuses
# to tie the counters to any other build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step() { echo "== $1" >> $out/progress.log; }
if [ "$1" = "c3" ]; then
  out=gpurun_out/prof_r03_c3; rm -rf $out; mkdir -p $out
  sha256sum parquet-go_amd/libpqdecode.so | cut -d' ' -f1 > $out/library.sha256
  step c3; timeout -k 10 1000 python3 -u tools/bench_configs.py --only c3full > $out/c3_sweep.jsonl 2> $out/c3.err || { tail -3 $out/c3.err; exit 1; }
  step done; tail -2 $out/c3_sweep.jsonl | cut -c1-200; exit 0
fi
out=gpurun_out/prof_r03; rm -rf $out; mkdir -p $out
sha256sum parquet-go_amd/libpqdecode.so | cut -d' ' -f1 > $out/library.sha256
step bench; timeout -k 10 400 python3 -u bench.py > $out/bench_c2_full.json 2> $out/bench.err || { tail -3 $out/bench.err; exit 1; }
step trace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 -u bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/delta_c2_bench_under_rocprof.json 2> $out/trace_err.log || exit 1
cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" $out/delta_c2_kernel_stats.csv || exit 1
step fetch; tools/pmc_pass.sh $out/fetch "FETCH_SIZE" python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline || exit 1
step write; tools/pmc_pass.sh $out/write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline || exit 1
python3 tools/pmc_summary.py $out/fetch delta_wave > $out/pmc_fetch.txt
python3 tools/pmc_summary.py $out/write delta_wave > $out/pmc_write.txt
step configs; timeout -k 10 600 python3 -u tools/bench_configs.py --pages 8000 > $out/configs_8000pages.jsonl 2> $out/configs.err || { tail -3 $out/configs.err; exit 1; }
step c4full; timeout -k 10 300 python3 -u tools/bench_configs.py --only c4 --pages 25000 > $out/configs_c4_25000pages.jsonl 2> $out/c4full.err || { tail -3 $out/c4full.err; exit 1; }
step large; timeout -k 10 600 python3 -u tools/bench_configs.py --only c2seg,c2stride,c2pages,c4big,hybig > $out/configs_large_pages.jsonl 2> $out/large.err || { tail -3 $out/large.err; exit 1; }
step segtrace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/segtrace -- python3 -u tools/bench_configs.py --only c2seg,c2pages,c4big,hybig > $out/segments_under_rocprof.jsonl 2> $out/segtrace_err.log || exit 1
cp "$(find $out/segtrace -name '*kernel_stats.csv' | head -1)" $out/segments_kernel_stats.csv || exit 1
step cfgtrace; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfgtrace -- python3 -u tools/bench_configs.py --only c1,c1b,c4,plain,bytes,walk,levels --pages 8000 > $out/configs_under_rocprof.jsonl 2> $out/cfgtrace_err.log || exit 1
cp "$(find $out/cfgtrace -name '*kernel_stats.csv' | head -1)" $out/configs_kernel_stats.csv || exit 1
# instruction counters: the fused segment kernel (per 512 values of 100 M) and the byte-array kernels
step segpmc; tools/pmc_pass.sh $out/segpmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python3 -u tools/bench_configs.py --only c2seg || exit 1
python3 tools/pmc_dispatches.py $out/segpmc seg_fused 100000000 > $out/segments_pmc.txt
step c4pmc; tools/pmc_pass.sh $out/c4pmc "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python3 -u tools/bench_configs.py --only c1,c1b,c4,plain --pages 8000 || exit 1
python3 tools/pmc_dispatches.py $out/c4pmc _wave_kernel 160000000 > $out/hybrid_c4_pmc.txt
# HBM traffic of DELTA_BYTE_ARRAY, one pass a counter block
step dbafetch; tools/pmc_pass.sh $out/dbafetch "FETCH_SIZE" python3 -u tools/bench_configs.py --only dba --pages 8000 || exit 1
step dbawrite; tools/pmc_pass.sh $out/dbawrite "WRITE_SIZE" python3 -u tools/bench_configs.py --only dba --pages 8000 || exit 1
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 -u tools/bench_configs.py --only dba --pages 8000"; python3 tools/pmc_summary.py $out/dbafetch byte_array_kernel; python3 tools/pmc_summary.py $out/dbawrite byte_array_kernel; } > $out/byte_array_pmc.txt
step chunks; timeout -k 10 300 python3 tools/bench_chunks.py > $out/c5_one_gpu.json 2> $out/c5.err || { tail -3 $out/c5.err; exit 1; }
step file; timeout -k 10 300 python3 -u tools/bench_parquet_file.py --rows 16000000 --row-groups 8 > $out/parquet_file_16M_rows.json 2> $out/file.err || { tail -3 $out/file.err; exit 1; }
step done
rm -rf $out/trace $out/segtrace $out/cfgtrace $out/fetch $out/write $out/segpmc $out/c4pmc $out/dbafetch $out/dbawrite
cat $out/pmc_fetch.txt $out/pmc_write.txt $out/byte_array_pmc.txt; head -4 $out/delta_c2_kernel_stats.csv | cut -c1-200
