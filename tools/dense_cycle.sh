#!/bin/bash
# On the GPU box: the GPU parity suite, then the hybrid / dictionary / nested-page / boolean configurations
# and the controlled run shapes.  usage: tools/dense_cycle.sh <tag> [pages]
tag=$1; pages=${2:-8000}
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/t_$tag.log 2>&1; rc=$?
tail -4 gpurun_out/t_$tag.log | cut -c1-250
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u tools/bench_configs.py --only c1,c1b,c4,plain --pages $pages > gpurun_out/configs_$tag.jsonl 2> gpurun_out/configs_$tag.err || { tail -5 gpurun_out/configs_$tag.err; exit 1; }
cut -c1-160 gpurun_out/configs_$tag.jsonl
timeout -k 10 300 python -u tools/hybrid_probe.py --pages $pages > gpurun_out/probe_$tag.jsonl 2> gpurun_out/probe_$tag.err || { tail -5 gpurun_out/probe_$tag.err; exit 1; }
cut -c1-160 gpurun_out/probe_$tag.jsonl
