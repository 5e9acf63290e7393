# kernel trace of the last reconstruct + mesh of configs[1] (tools/stage_small.py): dispatch list, idle gaps
root=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/prof_small; cd /tmp && export TMPDIR=/tmp
(cd $root && rocprofv3 --kernel-trace --stats -d /tmp/prof_small -o r -- python -m nksr_amd.tools.stage_small > $root/gpurun_out/small_prof.out 2>/dev/null)
db=$(find /tmp/prof_small -name '*.db' | head -1)
cd $root && python -m nksr_amd.tools.prof_gaps $db gpurun_out/kgaps_small.md 60 ${1:-11} > /dev/null
python -m nksr_amd.tools.prof_timeline $db gpurun_out/ktimeline_small.md ${1:-11} 30 > /dev/null
python - <<PY
import sqlite3
db=sqlite3.connect("$db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows=list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
t_end=max(r[1] for r in rows); rows=[r for r in rows if r[0] >= t_end - ${1:-11}*1e6]
agg={}
for s,e,n in rows:
    a=agg.setdefault(n[:60],[0,0.0]); a[0]+=1; a[1]+=(e-s)/1e3
print('dispatches in the last ${1:-11} ms:', len(rows), 'busy us', sum(v[1] for v in agg.values()))
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:40]:
    print('%5d %8.1f us  %s' % (c,t,n))
PY
tail -2 gpurun_out/small_prof.out
