"""Wall-clock (synchronised) split of the network / hierarchy stage on the bench cloud.
python -m nksr_amd.tools.prof_stages [points]"""
import sys
import time

import torch

import nksr_amd
from nksr_amd import utils
from nksr_amd.density import scale_for_detail_level
from nksr_amd.nn.network import sort_cloud
from nksr_amd.svh import SparseFeatureHierarchy, inv_w0_f32


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device('cuda:0')
    xyz, nrm = utils.synth_scene(n, seed=0)
    xyz, nrm = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    rec = nksr_amd.Reconstructor(dev)
    hp = rec.hparams
    net = rec.network
    acc = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    reps = 4
    for it in range(reps + 1):
        if it == 1:
            acc.clear()
        torch.cuda.synchronize()
        t = time.perf_counter()
        scale = scale_for_detail_level(xyz, 1.0, hp.voxel_size)
        t = tick('detail-level scale', t)
        xs = (xyz * scale).contiguous()
        ks, xs, ns = sort_cloud(xs, nrm, inv_w0_f32(hp.voxel_size))
        t = tick('sort_cloud', t)
        enc_svh = SparseFeatureHierarchy(hp.voxel_size, hp.tree_depth, dev).build_point_splatting_sorted(xs, ks)
        t = tick('encoder hierarchy', t)
        enc = net.encoder(xs, ns, enc_svh, 0, sorted_keys=ks)
        t = tick('point encoder', t)
        cand = SparseFeatureHierarchy(hp.voxel_size, hp.tree_depth, dev).build_point_neighborhood_sorted(enc.keys)
        t = tick('candidate hierarchy', t)
        feat, dec_svh, _ = net.unet(enc, enc_svh, adaptive_depth=hp.adaptive_depth, gt_decoder_svh=cand)
        t = tick('unet (convs, pruning, heads)', t)
    for k, v in acc.items():
        print('%-32s %7.3f ms' % (k, v / reps * 1e3))
    print('%-32s %7.3f ms' % ('total', sum(acc.values()) / reps * 1e3))


if __name__ == '__main__':
    main()
