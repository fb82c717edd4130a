#!/bin/bash
# usage (on the GPU box): tools/pmc_one.sh <config list for bench_configs --only> <kernel name part> <values per dispatch> [more bench_configs arguments]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_one; rm -rf $out
echo step pmc $1
cfg=$1; pat=$2; per=$3; shift 3
tools/pmc_pass.sh $out "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python3 -u tools/bench_configs.py --only $cfg "$@" || exit 1
python3 tools/pmc_dispatches.py $out $pat $per | cut -c1-220
