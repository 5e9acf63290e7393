"""HBM traffic of the SpMV kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
tools/spmv_probe.py.  Run on the GPU box:  python -m nksr_amd.tools.spmv_pmc <out.json> [points]
Corrections as prescribed by MI355X_MICROARCH.md (HBM / rocprofv3 section): counters in KiB-like units of
1024 B... FETCH_SIZE counts the 128-byte requests of a wide coalesced stream at 64 B on gfx950 -> doubled."""
import csv
import glob
import json
import os
import subprocess
import sys


PROBE = 'nksr_amd.tools.spmv_probe'


def run_pass(counter, points, tag, probe_args=('0',)):
    out = '/tmp/pmc_%s' % tag
    subprocess.run(['rm', '-rf', out])
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--',
           sys.executable, '-m', PROBE, str(points)] + list(probe_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=os.environ.get('GRAFT_REPO_ROOT', os.getcwd()))
    line = [l for l in r.stdout.splitlines() if l.startswith('M=')]
    is_spmv = lambda n: 'k_spmv' in n and 'fixup' not in n and 'plan' not in n
    dur = {}
    for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if is_spmv(row.get('Kernel_Name', '')):
                dur[row.get('Dispatch_Id')] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    vals, durs = [], []
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if is_spmv(row.get('Kernel_Name', '')) and row.get('Counter_Name') == counter:
                d = dur.get(row.get('Dispatch_Id'), 1e9)
                if d >= 20.0:       # launches the PCG's done flag turned into no-ops move nothing: not part of the average
                    vals.append(float(row['Counter_Value']))
                    durs.append(d)
    return vals, durs, (line[0] if line else '')


def main():
    out = sys.argv[1]
    points = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    fv, fd, desc = run_pass('FETCH_SIZE', points, 'fetch')
    wv, wd, _ = run_pass('WRITE_SIZE', points, 'write')
    if not fv or not wv:
        print('no counter rows found', file=sys.stderr)
        sys.exit(1)
    import re
    m = re.search(r'M=(\d+) nnz=(\d+)', desc)
    M, nnz = int(m.group(1)), int(m.group(2))
    fetch_kb, write_kb = sum(fv) / len(fv), sum(wv) / len(wv)
    hbm = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
    alg = 8 * nnz + 12 * M + 4
    from nksr_amd import build
    rec = {
        'kernel_source_hash': build.kernel_hash('spmv'),
        'kernel': 'k_spmv<0> (tools/spmv_probe.py, bench matrix: M=%d nnz=%d)' % (M, nnz),
        'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python -m nksr_amd.tools.spmv_probe %d 0  '
                   '(second pass: --pmc WRITE_SIZE); driver: python -m nksr_amd.tools.spmv_pmc' % points,
        'FETCH_SIZE_KB_per_launch': fetch_kb, 'WRITE_SIZE_KB_per_launch': write_kb, 'launches': len(fv), 'note': 'no-op launches (< 20 us: the done flag of the probe\'s own PCG) excluded',
        'correction': 'gfx950 rocprofv3 FETCH_SIZE counts 128-B requests of a wide coalesced stream at 64 B: doubled '
                      '(MI355X_MICROARCH.md section HBM); WRITE_SIZE uncorrected',
        'hbm_bytes_per_launch': hbm, 'algorithmic_bytes_per_launch': alg,
        'avg_kernel_us_under_pmc': sum(fd) / max(len(fd), 1),
        'traffic_over_algorithmic': hbm / alg,
    }
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
