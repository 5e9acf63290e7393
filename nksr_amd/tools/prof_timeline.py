"""Kernel timeline of the tail of a rocprofv3 rocpd database: every dispatch of at least `big_us` in order, the smaller ones between
them folded into one line each (count, total, the three heaviest names).  Usage:
python -m nksr_amd.tools.prof_timeline <results.db> [out.md] [tail_ms] [big_us]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^_Z\d+', '', name)
    return name[:60]


def timeline(db_path, tail_ms=0.0, big_us=300.0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
    if not rows:
        return 'no dispatches'
    t_end = max(r[1] for r in rows)
    if tail_ms > 0:
        rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
    t0 = rows[0][0]
    lines = ['| at ms | ms | kernel(s) |', '|---|---|---|']
    small = []

    def flush():
        if not small:
            return
        tot = sum(e - s for s, e, _ in small) / 1e6
        span = (small[-1][1] - small[0][0]) / 1e6
        agg = {}
        for s, e, n in small:
            agg[short(n)] = agg.get(short(n), 0.0) + (e - s) / 1e6
        top = sorted(agg.items(), key=lambda kv: -kv[1])[:3]
        lines.append('| %.1f | %.2f | (%d small, span %.2f: %s) |' % ((small[0][0] - t0) / 1e6, tot, len(small), span,
                                                                   ', '.join('%s %.2f' % kv for kv in top)))
        del small[:]

    for s, e, n in rows:
        if (e - s) / 1e3 >= big_us:
            flush()
            lines.append('| %.1f | %.2f | `%s` |' % ((s - t0) / 1e6, (e - s) / 1e6, short(n)))
        else:
            small.append((s, e, n))
    flush()
    return '\n'.join(lines)


if __name__ == '__main__':
    res = timeline(sys.argv[1], float(sys.argv[3]) if len(sys.argv) > 3 else 0.0, float(sys.argv[4]) if len(sys.argv) > 4 else 300.0)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(res + '\n')
    else:
        print(res)
