"""Where the GPU idles: the largest gaps between consecutive kernel dispatches of a rocprofv3 rocpd database, with the kernels
on either side, and the busy fraction.  Usage: python -m nksr_amd.tools.prof_gaps <results.db> [out.md] [top] [tail_ms]"""
import sqlite3
import sys


def gaps(db_path, top=40, min_gap_us=200.0, tail_ms=0.0):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
    if not rows:
        return 'no dispatches'
    if tail_ms > 0:          # only the last tail_ms milliseconds (the last timed step of a bench run)
        t_end = max(r[1] for r in rows)
        rows = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy = 0
    cur_end = rows[0][0]
    out = []
    for i, (s, e, n) in enumerate(rows):
        if s > cur_end:
            g = (s - cur_end) / 1e3
            if g >= min_gap_us:
                out.append((g, (cur_end - t0) / 1e6, rows[i - 1][2][:70], n[:70]))
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
    out.sort(reverse=True)
    span = (t1 - t0) / 1e6
    lines = ['span %.1f ms, busy %.1f ms (%.1f %%), %d dispatches, idle in gaps >= %.0f us: %.1f ms' % (
        span, busy / 1e6, 100 * busy / 1e6 / span, len(rows), min_gap_us, sum(o[0] for o in out) / 1e3), '',
        '| gap ms | at ms | after kernel | before kernel |', '|---|---|---|---|']
    for g, at, a, b in out[:top]:
        lines.append('| %.2f | %.1f | `%s` | `%s` |' % (g / 1e3, at, a, b))
    return '\n'.join(lines)


if __name__ == '__main__':
    res = gaps(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 40, tail_ms=float(sys.argv[4]) if len(sys.argv) > 4 else 0.0)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(res + '\n')
    print(res)
