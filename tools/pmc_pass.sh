#!/bin/bash
# usage: tools/pmc_pass.sh <outdir> "<counters>" <program args...>   (run on the GPU box)
out=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/stdout.log" 2> "$out/stderr.log"
echo "pmc exit=$?"
