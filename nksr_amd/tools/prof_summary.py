"""Summarise a rocprofv3 rocpd SQLite database: per-kernel calls / total / average time.
Usage: python -m nksr_amd.tools.prof_summary <results.db> [out.md]"""
import sqlite3
import sys


def summarise(db_path, top=40, tail_ms=0.0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    where = ''
    if tail_ms > 0:      # only the last tail_ms milliseconds (the last timed step of a bench run)
        t_end = list(cur.execute('select max(end) from %s' % kd))[0][0]
        where = 'where d.start >= %d ' % int(t_end - tail_ms * 1e6)
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, "
         "max(d.end-d.start)/1e3, s.arch_vgpr_count, s.sgpr_count, s.group_segment_size, s.private_segment_size "
         "from %s d join %s s on d.kernel_id=s.id %sgroup by s.kernel_name order by 3 desc" % (kd, ks, where))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    lines = ['| kernel | calls | total ms | % | avg us | min us | max us | vgpr | sgpr | lds B | scratch B |', '|---|---|---|---|---|---|---|---|---|---|---|']
    for r in rows[:top]:
        lines.append('| `%s` | %d | %.2f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |' % (
            r[0][:90], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
    return 'total GPU kernel time: %.2f ms over %d kernels\n\n' % (tot, len(rows)) + '\n'.join(lines)


if __name__ == '__main__':
    import os
    out = summarise(sys.argv[1], top=int(os.environ.get('KSTATS_ROWS', 40)), tail_ms=float(os.environ.get('KSTATS_TAIL_MS', 0)))
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out + '\n')
    print(out)
