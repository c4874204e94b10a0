#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <python args>   (one --pmc pass; run on the GPU box)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $out -- python3 "$@" > $out.log 2>&1
