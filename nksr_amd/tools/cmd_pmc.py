"""Hardware counters of selected kernels of an ARBITRARY command, one rocprofv3 --pmc pass per counter group of tools/scene_pmc.py
(--kernel-trace only).  Run on the GPU box:
    [NKSR_PMC_GROUPS=0,1,7] python -m nksr_amd.tools.cmd_pmc <out.json> <kernel name fragments, comma separated> -- <command ...>
Per kernel name the LONGEST dispatch is reported with every counter and a few derived figures."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

from .scene_pmc import GROUPS


def run_pass(counters, cmd, frags):
    out = tempfile.mkdtemp(prefix='pmc_')
    env = dict(os.environ, TMPDIR='/tmp')
    r = subprocess.run(['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--'] + cmd,
                       env=env, capture_output=True, text=True)
    dur, per = {}, {}
    for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row.get('Dispatch_Id')] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get('Kernel_Name', '')
            if not any(k in name for k in frags):
                continue
            i = name.find('k_')
            d = per.setdefault(name[i:i + 48] if i >= 0 else name[:48], {}).setdefault(row.get('Dispatch_Id'), {})
            d[row.get('Counter_Name')] = d.get(row.get('Counter_Name'), 0.0) + float(row['Counter_Value'])
    shutil.rmtree(out, ignore_errors=True)
    res = {}
    for k, disp in per.items():
        best = max(disp, key=lambda i: dur.get(i, 0.0))
        res[k] = dict(disp[best], us=dur.get(best, 0.0), dispatches=len(disp))
    if not res:
        sys.stderr.write('pass %s gave no counters: %s\n' % (counters, r.stderr[-800:]))
    return res


def main():
    out, frags = sys.argv[1], sys.argv[2].split(',')
    cmd = sys.argv[sys.argv.index('--') + 1:]
    groups = GROUPS          # (scene_pmc applies NKSR_PMC_GROUPS at import)
    rec = {}
    for g in groups:
        for k, v in run_pass(g, cmd, frags).items():
            e = rec.setdefault(k, {})
            e.setdefault('us', v['us'])
            if 'FETCH_SIZE' in g:
                e['us_fetch_pass'] = v['us']
            if 'WRITE_SIZE' in g:
                e['us_write_pass'] = v['us']
            v.pop('us')
            e['dispatches'] = v.pop('dispatches')
            e.update(v)
    for k, e in rec.items():
        if 'FETCH_SIZE' in e:
            e['fetch_GB'] = 2.0 * 1024.0 * e['FETCH_SIZE'] / 1e9
        if 'WRITE_SIZE' in e:
            e['write_GB'] = 1024.0 * e['WRITE_SIZE'] / 1e9
        if e.get('SQ_WAVE_CYCLES'):
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS'):
                if c in e:
                    e[c + '_frac'] = e[c] / e['SQ_WAVE_CYCLES']
        if e.get('SQ_WAVES'):
            for c in ('SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES'):
                if c in e:
                    e[c + '_per_wave'] = e[c] / e['SQ_WAVES']
        if e.get('TCC_HIT_sum') is not None and e.get('TCC_MISS_sum') is not None and e['TCC_HIT_sum'] + e['TCC_MISS_sum'] > 0:
            e['l2_hit'] = e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum'])
    json.dump(rec, open(out, 'w'), indent=1, sort_keys=True)
    for k in sorted(rec, key=lambda k: -rec[k].get('us', 0)):
        print(k)
        for c, v in sorted(rec[k].items()):
            print('    %-36s %s' % (c, ('%.4g' % v) if isinstance(v, float) else v))


if __name__ == '__main__':
    main()
