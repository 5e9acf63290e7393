#!/bin/bash
# Per-kernel statistics of an arbitrary command (run on the GPU box):
#   bash nksr_amd/tools/kprof.sh <tag> <command ...>
# writes gpurun_out/kprof_<tag>.md
tag=${1:-run}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out && rm -rf /tmp/prof_$tag
cd /tmp && export TMPDIR=/tmp
(cd $root && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- "$@" > $root/gpurun_out/kprof_$tag.out 2>/dev/null)
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
cd $root && KSTATS_ROWS=${KSTATS_ROWS:-30} python -m nksr_amd.tools.prof_summary $db gpurun_out/kprof_$tag.md | head -${KSTATS_TOP:-24}
