"""HBM traffic of the matrix-free operator's kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
tools/fused_probe.py.  Run on the GPU box:  python -m nksr_amd.tools.fused_pmc <out.json> [points]
Same corrections as tools/spmv_pmc.py (MI355X_MICROARCH.md, HBM / rocprofv3 section): KiB units, FETCH_SIZE doubled on gfx950.
Launches that the PCG's done flag turned into no-ops are excluded (duration < 20 us)."""
import csv
import glob
import json
import os
import subprocess
import sys

KERNELS = ('k_fz_cells', 'k_fz_cellsum', 'k_fz_gather')


def _match(k, name):
    """operator instantiations only (MODE 0), mangled or demangled kernel names"""
    if k not in name:
        return False
    if k in ('k_fz_cells', 'k_fz_gather'):
        return (k + 'ILi0') in name or (k + '<0') in name
    return True


def run_pass(counter, points, tag):
    out = '/tmp/pmc_%s' % tag
    subprocess.run(['rm', '-rf', out])
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--',
           sys.executable, '-m', 'nksr_amd.tools.fused_probe', str(points), '10']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
    desc = [l for l in r.stdout.splitlines() if l.startswith('M=')]
    dur = {}
    for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row.get('Dispatch_Id')] = (row.get('Kernel_Name', ''), (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    vals = {k: [] for k in KERNELS}
    durs = {k: [] for k in KERNELS}
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            name = row.get('Kernel_Name', '')
            for k in KERNELS:
                if _match(k, name):
                    d = dur.get(row.get('Dispatch_Id'), (name, 1e9))[1]
                    if d >= 20.0:
                        vals[k].append(float(row['Counter_Value']))
                        durs[k].append(d)
    return vals, durs, (desc[0] if desc else '')


def main():
    out = sys.argv[1]
    points = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    fv, fd, desc = run_pass('FETCH_SIZE', points, 'ffetch')
    wv, wd, _ = run_pass('WRITE_SIZE', points, 'fwrite')
    from nksr_amd import build
    rec = {'probe': 'python -m nksr_amd.tools.fused_probe %d 10' % points, 'system': desc, 'kernel_source_hash': build.kernel_hash('fused'),
           'correction': 'KiB units; gfx950 FETCH_SIZE counts the 128-B requests of a coalesced stream at 64 B: doubled; WRITE_SIZE uncorrected',
           'kernels': {}}
    tot = 0.0
    for k in KERNELS:
        if not fv[k] or not wv[k]:
            continue
        fetch = 2.0 * 1024.0 * sum(fv[k]) / len(fv[k])
        write = 1024.0 * sum(wv[k]) / len(wv[k])
        us = sum(fd[k]) / len(fd[k])
        rec['kernels'][k] = {'launches': len(fv[k]), 'fetch_bytes': fetch, 'write_bytes': write, 'avg_us_under_pmc': us,
                             'hbm_TBps_under_pmc': (fetch + write) / us / 1e6}
        tot += fetch + write
    rec['hbm_bytes_per_application'] = tot
    import re
    m = re.search(r'M=(\d+) rows=(\d+) partial blocks=(\d+)', desc)
    if m:      # same formula as csrc/fused.hip FusedOperator::bytes (depth 4 on the probe workload)
        M, rows, blocks = int(m.group(1)), int(m.group(2)), int(m.group(3))
        rec['physical_bytes_per_application'] = 4 * 27 * 4 * rows + 4 * 4 * rows + 2 * 128 * blocks + (128 + 3 * 108 + 12) * M
        rec['algorithmic_note'] = 'algorithmic minimum (DESIGN.md section 3.5): 4 B per stored entry + 4 B per row and level + 116 B per unknown'
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
