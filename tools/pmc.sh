#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <counters...> -- <kernel_micro args...>   (run on the GPU box)
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $R/gpurun_out/$out -- python3 $R/tools/kernel_micro.py "$@" > $R/gpurun_out/$out.log 2>&1
tail -1 $R/gpurun_out/$out.log
