"""Do the fields die with their last reference?  Reconstructs a small cloud (single and chunked), drops the results and lists
what only the cyclic collector could free (tensors held by a reference cycle stay on the device until it runs).
Usage (GPU box): python -m nksr_amd.tools.cycle_probe"""
import collections
import gc

import torch

import nksr
from nksr_amd import utils


def garbage_after(fn):
    gc.collect()
    gc.disable()
    gc.set_debug(gc.DEBUG_SAVEALL)
    fn()
    gc.collect()
    found = collections.Counter(type(o).__module__ + '.' + type(o).__name__ for o in gc.garbage)
    nbytes = sum(o.numel() * o.element_size() for o in gc.garbage if isinstance(o, torch.Tensor))
    gc.set_debug(0)
    del gc.garbage[:]
    gc.enable()
    return found, nbytes


def main():
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_terrain_patch(60000, seed=1, extent=(10.0, 10.0))
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr.Reconstructor(dev)

    def single():
        f = rec.reconstruct(xyz, nrm, detail_level=None)
        f.extract_dual_mesh(mise_iter=1)

    def chunked():
        f = rec.reconstruct(xyz, nrm, detail_level=None, chunk_size=5.01)
        f.extract_dual_mesh(mise_iter=1)
        list(f.fields.keys())

    for name, fn in (('single', single), ('chunked', chunked)):
        fn()
        found, nbytes = garbage_after(fn)
        ours = {k: v for k, v in found.items() if k.startswith(('nksr', 'torch.Tensor', 'oracle'))}
        print('%s: %d objects in cycles, %.1f MB of tensors; %s' % (name, sum(found.values()), nbytes / 1e6, ours or 'none of ours'))


if __name__ == '__main__':
    main()
