cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo step trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$1_trace -- python3 -u tools/bench_configs.py --only $1 > gpurun_out/$1_trace.jsonl 2> gpurun_out/$1_trace.err || exit 1
f=$(ls gpurun_out/$1_trace/*/*kernel_stats.csv | head -1)
cut -c1-60,200-330 $f | head -14
