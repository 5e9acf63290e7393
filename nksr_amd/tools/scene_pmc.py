"""Hardware counters of the heavy kernels of the 64-chunk scene step (or any bench flags), one rocprofv3 --pmc pass per counter
group (never combined with a trace domain other than --kernel-trace).  Run on the GPU box:
    python -m nksr_amd.tools.scene_pmc <out.json> [bench flags ...]
Per kernel (by name fragment) the dispatch with the LONGEST duration is reported: its duration under the counters and every
counter of every group.  FETCH_SIZE is in KiB and, on gfx950, counts the 128-byte requests of a coalesced stream at 64 bytes
(MI355X_MICROARCH.md, HBM section): fetch_bytes = 2 * 1024 * FETCH_SIZE."""
import csv
import glob
import json
import os
import subprocess
import sys

GROUPS = [
    ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD'],
    ['SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_MFMA', 'SQ_INST_LEVEL_VMEM', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS',
     'GRBM_GUI_ACTIVE'],
    ['FETCH_SIZE'],
    ['WRITE_SIZE'],
    ['TCC_HIT_sum', 'TCC_MISS_sum'],
    ['TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_PENDING_STALL_CYCLES_sum', 'TA_TA_BUSY_sum'],
    ['TA_ADDR_STALLED_BY_TC_CYCLES_sum', 'TA_DATA_STALLED_BY_TC_CYCLES_sum', 'TCP_TCP_TA_DATA_STALL_CYCLES_sum', 'TD_TD_BUSY_sum'],
    ['SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS', 'SQ_INSTS_SMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_FLAT'],
]
if os.environ.get('NKSR_PMC_GROUPS'):       # e.g. "0,1,7": a quick look at the issue / wait split only
    GROUPS = [GROUPS[int(i)] for i in os.environ['NKSR_PMC_GROUPS'].split(',')]
LAST_BENCH_LINE = None
KERNELS = ['k_cell_blocks', 'k_fz_cells', 'k_cheb16_step', 'k_kernel_rows', 'k_sparse_conv3', 'k_splat_mean32', 'k_splat_trilinear',
           'k_fz_gather', 'k_fz_cellsum', 'k_evaluate_f', 'k_build_nbr', 'k_row_count', 'k_row_fill']


def short(name):
    i = name.find('k_')
    return name[i:i + 40] if i >= 0 else name[:40]


def run_pass(counters, flags, tag):
    import shutil
    import tempfile
    out = tempfile.mkdtemp(prefix='pmc_%s_' % tag, dir='/tmp')      # (per invocation: two runs at once must not share counter files)
    env = dict(os.environ, TMPDIR='/tmp')
    root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
    cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.join(root, 'bench.py')] + flags
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd='/tmp')
    global LAST_BENCH_LINE
    for line in r.stdout.splitlines():
        if line.startswith('{'):
            LAST_BENCH_LINE = line
    dur = {}
    for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row.get('Dispatch_Id')] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    per = {}          # kernel -> dispatch id -> {counter: value}
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '')
            if not any(k in name for k in KERNELS):
                continue
            d = per.setdefault(short(name), {}).setdefault(row.get('Dispatch_Id'), {})
            d[row.get('Counter_Name')] = d.get(row.get('Counter_Name'), 0.0) + float(row['Counter_Value'])
    res = {}
    for k, disp in per.items():
        best = max(disp, key=lambda i: dur.get(i, 0.0))
        res[k] = dict(disp[best], us=dur.get(best, 0.0), dispatches=len(disp))
    if not res:
        sys.stderr.write('pass %s gave no counters: %s\n' % (tag, r.stderr[-600:]))
    shutil.rmtree(out, ignore_errors=True)
    return res


def main():
    out = sys.argv[1]
    flags = sys.argv[2:] or ['--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-cloud', '--no-small-inputs', '--no-live-pmc']
    rec = {'command': 'bench.py ' + ' '.join(flags), 'kernels': {}}
    for gi, g in enumerate(GROUPS):
        res = run_pass(g, flags, 'g%d' % gi)
        for k, v in res.items():
            e = rec['kernels'].setdefault(k, {})
            e['us_pass%d' % gi] = v['us']
            if 'FETCH_SIZE' in g:
                e['us_fetch_pass'] = v['us']
            v.pop('us')
            e['dispatches'] = v.pop('dispatches')
            e.update(v)
    for k, e in rec['kernels'].items():
        if 'FETCH_SIZE' in e:
            e['fetch_bytes'] = 2.0 * 1024.0 * e['FETCH_SIZE']
            e['fetch_TBps'] = e['fetch_bytes'] / e.get('us_fetch_pass', 1e9) / 1e6
        if 'WRITE_SIZE' in e:
            e['write_bytes'] = 1024.0 * e['WRITE_SIZE']
        if e.get('SQ_WAVE_CYCLES'):
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
                if c in e:
                    e[c + '_frac'] = e[c] / e['SQ_WAVE_CYCLES']
        if e.get('TCC_HIT_sum') is not None and e.get('TCC_MISS_sum') is not None and e['TCC_HIT_sum'] + e['TCC_MISS_sum'] > 0:
            e['l2_hit'] = e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum'])
    # the operator application with every chunk still iterating = the longest dispatch of each of its three kernels
    op = [k for k in rec['kernels'] if k.startswith(('k_fz_cells<0', 'k_fz_cellsILi0', 'k_fz_cellsum', 'k_fz_gather<0', 'k_fz_gatherILi0'))]
    if len(op) == 3 and all('fetch_bytes' in rec['kernels'][k] and 'write_bytes' in rec['kernels'][k] for k in op):
        rec['operator_application'] = {k: {'fetch_bytes': rec['kernels'][k]['fetch_bytes'], 'write_bytes': rec['kernels'][k]['write_bytes'],
                                           'us': rec['kernels'][k].get('us_fetch_pass')} for k in op}
        rec['hbm_bytes_per_application'] = sum(rec['kernels'][k]['fetch_bytes'] + rec['kernels'][k]['write_bytes'] for k in op)
        if LAST_BENCH_LINE:
            try:
                rec['physical_bytes_per_application'] = json.loads(LAST_BENCH_LINE)['roofline']['physical_bytes_per_launch']
                rec['algorithmic_bytes_per_application'] = json.loads(LAST_BENCH_LINE)['roofline']['bytes_per_launch']
            except Exception:
                pass
        from nksr_amd import build
        rec['kernel_source_hash'] = build.kernel_hash('fused')
        rec['correction'] = 'KiB units; gfx950 FETCH_SIZE counts the 128-B requests of a coalesced stream at 64 B: doubled; WRITE_SIZE uncorrected'
    json.dump(rec, open(out, 'w'), indent=1, sort_keys=True)
    for k in sorted(rec['kernels'], key=lambda k: -rec['kernels'][k].get('us_pass0', 0)):
        e = rec['kernels'][k]
        if os.environ.get('NKSR_PMC_GROUPS'):
            print(k, {c: v for c, v in sorted(e.items())})
        print('%-42s %8.0f us  fetch %.2f TB/s  l2hit %.2f  wait %.2f  stall %.2f  active %.2f  waves %d' % (
            k, e.get('us_pass0', 0), e.get('fetch_TBps', 0), e.get('l2_hit', 0), e.get('SQ_WAIT_ANY_frac', 0), e.get('SQ_WAIT_INST_ANY_frac', 0),
            e.get('SQ_ACTIVE_INST_ANY_frac', 0), e.get('SQ_WAVES', 0)))


if __name__ == '__main__':
    main()
