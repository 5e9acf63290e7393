# kernel trace of ONE rank's step at 8 ranks (tools/prof_rank_tail.py with the merge part skipped): idle gaps, timeline, and the
# dispatch counts of the last <tail> ms (default 66 = the whole step; 8 = the mesher)
root=${GRAFT_REPO_ROOT:-$(pwd)}
tail_ms=${1:-66}
rm -rf /tmp/prof_tail; cd /tmp && export TMPDIR=/tmp
(cd $root && NKSR_TAIL_STEPS_ONLY=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_tail -o r -- python -m nksr_amd.tools.prof_rank_tail 8 3 > $root/gpurun_out/tail_prof.out 2>/dev/null)
db=$(find /tmp/prof_tail -name '*.db' | head -1)
cd $root && python -m nksr_amd.tools.prof_gaps $db gpurun_out/kgaps_tail.md 60 $tail_ms > /dev/null
python -m nksr_amd.tools.prof_timeline $db gpurun_out/ktimeline_tail.md $tail_ms 100 > /dev/null
python - <<PY
import sqlite3
db=sqlite3.connect("$db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
t_end=max(r[1] for r in rows); rows=[r for r in rows if r[0] >= t_end - $tail_ms*1e6]
agg={}
for s,e,n in rows:
    a=agg.setdefault(n[:64],[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e3
print('dispatches in the last $tail_ms ms:', len(rows), 'busy us %.0f' % sum(v[1] for v in agg.values()))
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]:
    print('%5d %8.1f us  %s' % (c,t,n))
PY
tail -3 gpurun_out/tail_prof.out | cut -c1-200
