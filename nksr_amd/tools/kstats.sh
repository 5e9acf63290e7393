#!/bin/bash
# Per-kernel statistics of the default bench workload (run on the GPU box):
#   bash nksr_amd/tools/kstats.sh <tag> [extra bench flags]
# writes gpurun_out/kstats_<tag>.md (+ the bench line in gpurun_out/kstats_<tag>.json)
tag=${1:-run}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out && rm -rf /tmp/prof_$tag
cd /tmp && export TMPDIR=/tmp
(cd $root && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' > $root/gpurun_out/kstats_$tag.json)
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
cd $root && python -m nksr_amd.tools.prof_summary $db gpurun_out/kstats_$tag.md | head -${KSTATS_TOP:-24}
python -m nksr_amd.tools.prof_gaps $db gpurun_out/kgaps_$tag.md 40 ${KGAPS_TAIL_MS:-0} > /dev/null
python -m nksr_amd.tools.prof_timeline $db gpurun_out/ktimeline_$tag.md ${KGAPS_TAIL_MS:-0} ${KTIMELINE_BIG_US:-300}
