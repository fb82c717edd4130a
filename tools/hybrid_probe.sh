#!/bin/bash
# On the GPU box: timing of the probe shapes, then an SQ counter pass over the same program.
# usage: tools/hybrid_probe.sh <tag> [shapes]
tag=$1; shapes=${2:-rle16,rle32,rle64,rle128,rle512,bp1,bp2,bp4,bp8,bp16,mix8,mix24,mix56}
mkdir -p gpurun_out
timeout -k 10 300 python3 -u tools/hybrid_probe.py --shapes $shapes > gpurun_out/probe_$tag.jsonl 2> gpurun_out/probe_$tag.err || { tail -5 gpurun_out/probe_$tag.err; exit 1; }
cat gpurun_out/probe_$tag.jsonl
tools/pmc_pass.sh gpurun_out/probe_pmc_$tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM" python3 tools/hybrid_probe.py --shapes $shapes --steps 1 || exit 1
python3 - <<PY
import csv,glob,collections
rows=collections.defaultdict(dict)
for f in glob.glob('gpurun_out/probe_pmc_$tag/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stream_wave_kernel' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
shapes="$shapes".split(',')
tiles=8000*20000/512
for s,d in zip(shapes,sorted(rows)):
    r=rows[d]
    print('%-8s SALU/tile %6.0f VALU/tile %6.0f LDS/tile %5.0f SMEM/tile %5.1f'%(s,r['SQ_INSTS_SALU']/tiles,r['SQ_INSTS_VALU']/tiles,r['SQ_INSTS_LDS']/tiles,r.get('SQ_INSTS_SMEM',0)/tiles))
PY
