#!/bin/bash
# On the GPU box: the GPU parity suite, the hybrid / dictionary / nested-page configurations, and an
# SQ counter pass over them.  usage: tools/hybrid_cycle.sh <tag> [pages]
tag=$1; pages=${2:-3000}
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/t_$tag.log 2>&1; rc=$?
tail -4 gpurun_out/t_$tag.log | cut -c1-250
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u tools/bench_configs.py --only c1,c1b,c4 --pages $pages > gpurun_out/configs_$tag.jsonl 2> gpurun_out/configs_$tag.err || { tail -5 gpurun_out/configs_$tag.err; exit 1; }
cut -c1-200 gpurun_out/configs_$tag.jsonl
tools/pmc_pass.sh gpurun_out/pmc_$tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR" python3 tools/bench_configs.py --only c1,c1b,c4 --pages $pages || exit 1
python3 - <<PY
import csv,glob,collections
rows=collections.defaultdict(dict)
for f in glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wave_kernel' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value']); rows[int(r['Dispatch_Id'])]['k']=r['Kernel_Name'][4:20]
seen=set()
for d in sorted(rows):
    key=(rows[d]['k'], round(rows[d]['SQ_INSTS_VALU'],-5))
    if key in seen: continue
    seen.add(key)
    print(d, ' '.join('%s=%s'%(k.replace('SQ_',''),(v if isinstance(v,str) else '%.3g'%v)) for k,v in sorted(rows[d].items())))
PY
