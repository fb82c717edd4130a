#!/bin/bash
# On the GPU box: delta-focused tests, the default bench, and an SQ counter pass.  usage: tools/gpu_cycle.sh <tag> [pytest -k expr]
tag=$1; kexpr=${2:-"delta or single"}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread -k "$kexpr" > gpurun_out/t_$tag.log 2>&1; rc=$?
tail -4 gpurun_out/t_$tag.log | cut -c1-250
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("value=%.1f GB/s  ms/step=%.4f  roofline=%.1f GB/s frac=%.4f" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"]))
PY
tools/pmc_pass.sh gpurun_out/pmc_$tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline && python3 tools/pmc_summary.py gpurun_out/pmc_$tag delta_wave
