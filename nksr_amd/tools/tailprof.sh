root=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/prof_tail; cd /tmp && export TMPDIR=/tmp
(cd $root && NKSR_TAIL_STEPS_ONLY=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_tail -o r -- python -m nksr_amd.tools.prof_rank_tail 8 3 > $root/gpurun_out/tail_prof.out 2>/dev/null)
db=$(find /tmp/prof_tail -name '*.db' | head -1)
cd $root && python -m nksr_amd.tools.prof_gaps $db gpurun_out/kgaps_tail.md 60 66 > /dev/null
python -m nksr_amd.tools.prof_timeline $db gpurun_out/ktimeline_tail.md 66 100 > /dev/null
tail -4 gpurun_out/tail_prof.out
