"""Kernel-rows micro-benchmark on the bench scene (configs[4]): records the two nksr_kernel_rows calls of one reconstruct() (position
rows, gradient rows), then replays them under the kernel / probe switches of csrc/kfield.hip (NKSR_ROWS_KERNEL, NKSR_ROWS_LEVELS,
NKSR_ROWS_DBG) with HIP events, and compares the rows of the two kernels bit for bit.
python -m nksr_amd.tools.rows_probe [scene points] [reps]"""
import os
import sys

import torch

import nksr_amd
from nksr_amd import configs
from nksr_amd.fields.kernel_field import KernelField

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))


def main():
    import bench
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device('cuda:0')
    rec = nksr_amd.Reconstructor(dev, hparams=configs.get_hparams('ks', tree_depth=5))
    xyz, nrm, scale, owner, bounds, n_scene, _ = bench.terrain_setup(rec, dev, n, 0, 1)
    calls = []
    orig = KernelField.kernel_rows_level_major

    def rec_call(self, xyz_, grad, scale_, out, level_stride, row_index=None, row_cells=None, site_scale=None):
        calls.append((self, xyz_, grad, scale_, out, level_stride, row_index, row_cells, site_scale))
        return orig(self, xyz_, grad, scale_, out, level_stride, row_index, row_cells, site_scale)

    KernelField.kernel_rows_level_major = rec_call
    os.environ['NKSR_ROWS_KERNEL'] = 'site'
    rec.reconstruct(xyz, nrm, detail_level=None, chunk_size=bench.TILE * scale, sharded_input=True, chunk_owner=owner, chunk_bounds=bounds)
    KernelField.kernel_rows_level_major = orig
    torch.cuda.synchronize()
    print('recorded %d calls' % len(calls))
    for c in calls:
        print('  grad=%s sites=%d level_stride=%d L=%d' % (c[2], c[1].shape[0], c[5], c[0].svh.depth))
    L = calls[0][0].svh.depth

    def run(env, which=None):
        for k in ('NKSR_ROWS_KERNEL', 'NKSR_ROWS_LEVELS', 'NKSR_ROWS_DBG'):
            os.environ.pop(k, None)
        os.environ.update(env)
        out = []
        for ci, c in enumerate(calls):
            if which is not None and ci != which:
                out.append(None)
                continue
            orig(*c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                orig(*c)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps)
        return out

    def show(tag, r):
        print('%-34s %s' % (tag, '  '.join('%s %8.3f ms' % ('grad' if c[2] else 'pos ', v) for c, v in zip(calls, r) if v is not None)))
        sys.stdout.flush()

    only = os.environ.get('ROWS_PROBE_ONLY')
    # bitwise comparison of the two kernels on the full scene
    rows = calls[0][4]
    nrows = calls[0][5]
    if not only:
        run({'NKSR_ROWS_KERNEL': 'site'})
        ref = rows.clone()
        rc_ref = calls[0][7].clone()
        rows.fill_(float('nan'))
        calls[0][7].fill_(-7)
        run({'NKSR_ROWS_KERNEL': 'coop'})
        a, b = rows[:L * nrows * 27].view(torch.int32), ref[:L * nrows * 27].view(torch.int32)
        written = ~torch.isnan(rows[:L * nrows * 27])                  # (pad rows are written by neither kernel)
        ndiff = int(((a != b) & written).sum().item())
        print('coop vs site: %d differing words of %d written (%d unwritten); row_cells equal where written: %s' % (
            ndiff, int(written.sum().item()), int((~written).sum().item()), bool(torch.equal(rc_ref[calls[0][7] != -7], calls[0][7][calls[0][7] != -7]))))
        del written, ref, a, b
    # the merged launch (one lane per row) on the same row list
    import ctypes as C
    from nksr_amd._lib import call, ptr, stream
    fld = calls[0][0]
    row_src = torch.full((nrows,), -1, dtype=torch.int32, device=dev)
    margs = {}
    for c in calls:
        ncomp = 3 if c[2] else 1
        call('nksr_row_sources', ptr(c[6]), c[1].shape[0], ncomp, 0 if ncomp == 1 else 1, ptr(row_src), stream())
        margs[ncomp] = (c[1], c[8], c[3])

    def merged():
        (xa, sa, fa), (xb, sb, fb) = margs[1], margs[3]
        pre = os.environ.get('NKSR_ROWS_PRECELLS', '1') != '0'
        if pre:
            call('nksr_row_cells_merged', C.byref(fld._hier), ptr(xa), ptr(xb), ptr(row_src), nrows, ptr(calls[0][7]), stream())
        call('nksr_kernel_rows_merged', C.byref(fld._hier), ptr(xa), ptr(sa), float(fa), ptr(xb), ptr(sb), float(fb), int(fld.approx_kernel_grad),
             ptr(row_src), nrows, ptr(calls[0][7]), int(pre), None, ptr(rows), stream())

    def run_merged(env):
        for k in ('NKSR_ROWS_KERNEL', 'NKSR_ROWS_LEVELS', 'NKSR_ROWS_DBG', 'NKSR_ROWS_PRECELLS'):
            os.environ.pop(k, None)
        os.environ.update(env)
        merged()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            merged()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    if only == 'merged':
        print('merged %.3f ms' % run_merged({}))
        return
    run({'NKSR_ROWS_KERNEL': 'site'})
    ref = rows.clone()
    rc_ref = calls[0][7].clone()
    rows.fill_(float('nan'))
    calls[0][7].fill_(-7)
    run_merged({})
    a, b = rows[:L * nrows * 27].view(torch.int32), ref[:L * nrows * 27].view(torch.int32)
    pad = (row_src < 0)
    live = (~pad)[None, :, None].expand(L, nrows, 27).reshape(-1)
    ndiff = int(((a != b) & live).sum().item())
    padbad = int(((rows[:L * nrows * 27] != 0) & ~live).sum().item())
    print('merged vs site: %d differing words in %d live rows; %d non-zero words in %d pad rows; row_cells equal on live rows: %s, -1 on pads: %s' % (
        ndiff, int((~pad).sum().item()), padbad, int(pad.sum().item()), bool(torch.equal(rc_ref[:, ~pad], calls[0][7][:, ~pad])),
        bool((calls[0][7][:, pad] == -1).all().item())))
    del ref, a, b, live
    r = run({'NKSR_ROWS_KERNEL': 'site'})
    show('site (one lane per site)', r)
    print('%-34s sum  %8.3f ms' % ('', sum(r)))
    print('%-34s both %8.3f ms' % ('merged, cells looked up in the kernel', run_merged({'NKSR_ROWS_PRECELLS': '0'})))
    print('%-34s both %8.3f ms' % ('merged, cells given (incl. their pass)', run_merged({'NKSR_ROWS_PRECELLS': '1'})))
    for d in range(L):
        print('%-34s both %8.3f ms' % ('merged level %d' % d, run_merged({'NKSR_ROWS_LEVELS': str(d)})))
    show('coop', run({'NKSR_ROWS_KERNEL': 'coop'}))


if __name__ == '__main__':
    main()
