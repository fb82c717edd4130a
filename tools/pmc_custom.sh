#!/bin/bash
# usage (on the GPU box): tools/pmc_custom.sh "<counters>" <config> <kernel name part> <values per dispatch> [more bench_configs arguments]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_custom; rm -rf $out
ctrs=$1; cfg=$2; pat=$3; per=$4; shift 4
echo step pmc $cfg: $ctrs
tools/pmc_pass.sh $out "$ctrs" python3 -u tools/bench_configs.py --only $cfg "$@" || exit 1
tail -3 $out/stderr.log | cut -c1-200
python3 tools/pmc_dispatches.py $out $pat $per | cut -c1-260
