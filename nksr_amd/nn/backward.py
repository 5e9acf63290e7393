"""Backward of ``network.encoder`` / ``network.unet`` -- the training path (models/nksr_net.py:73-78: the loss back-propagates
through the kernel solve into the U-Net's heads, trunk and the point encoder).  The forward stays the HIP forward of
nn/network.py; here its reverse sweep, wrapped as two ``torch.autograd.Function``s so that ``feat.basis_features`` /
``normal_features`` / ``structure_features`` / ``udf_features`` carry a graph when the network is in training mode:

  * sparse convolution, data gradient: the SAME fp32-MFMA kernel (csrc/nn.hip k_sparse_conv3) with the taps mirrored and the
    weight tiles transposed -- the neighbour relation of a grid is symmetric (nbr[i][s] = j  <=>  nbr[j][26 - s] = i);
  * sparse convolution, weight gradient: csrc/nn.hip k_conv3_wgrad (fp32 MFMA over pairs of voxels, deterministic per-chunk
    partials); the linear heads: plain library GEMMs (rocBLAS through torch.matmul);
  * pooling / gathers / splats: the transposes of their index maps (index_add over the tables the forward recorded); the
    trilinear splat of the encoder is transposed through the points' eight-corner tables (hash queries);
  * ReLU masks and the normal target's normalisation: element-wise.
Discrete decisions (structure pruning, argmax) are constants of the sweep, as in the reference (the structure head is trained
by its own cross-entropy term, models/loss.py:152-160).  Checked against torch autograd through a torch statement of the same
forward (tests/test_gpu_network.py)."""
import torch

from .._lib import call, ptr, stream


def conv3_dgrad(gz, nbr, weight):
    """d(input) of out[i] = sum_s W[s]^T in[nbr[i][s]]: the forward kernel on gz with W'[s] = W[26 - s]^T, no bias, no activation."""
    n, C = gz.shape
    wt = weight.detach().flip(0).transpose(1, 2).contiguous()
    zero = torch.zeros(C, dtype=torch.float32, device=gz.device)
    out = torch.empty_like(gz)
    if n:
        call('nksr_sparse_conv3', ptr(gz.contiguous()), ptr(nbr), n, C, ptr(wt), ptr(zero), None, 0, ptr(out), stream())
    return out


def conv3_wgrad(x, nbr, gz):
    """d(weight)[s] = sum_i in[nbr[i][s]]^T gz[i]: csrc/nn.hip k_conv3_wgrad (fp32 MFMA, two voxels per instruction, one wavefront per
    (chunk of voxels, tap), per-chunk partials added here in chunk order)."""
    from .._lib import lib
    n, C = x.shape
    if n == 0:
        return x.new_zeros(27, C, C)
    nch = int(lib.nksr_conv3_wgrad_chunks(n))
    part = torch.empty((nch, 27, C, C), dtype=torch.float32, device=x.device)
    call('nksr_conv3_wgrad', ptr(x.contiguous()), ptr(nbr), n, C, ptr(gz.contiguous()), ptr(part), stream())
    return part.sum(0)


def conv3_backward(x_in, nbr, weight, out_post, g_out):
    """relu(b + conv(x_in)) backwards: (d x_in, d weight, d bias)."""
    gz = (g_out * (out_post > 0).to(g_out.dtype)).contiguous()
    return conv3_dgrad(gz, nbr, weight), conv3_wgrad(x_in, nbr, gz), gz.sum(0)


def linear_backward(x, weight, g):
    """out = x W^T + b backwards: (d x, d W, d b)."""
    return g @ weight.detach(), g.t() @ x, g.sum(0)


def point_corners(grid, level, inv_w0, xyz):
    """The eight level-``level`` voxels around every point and their trilinear weights (0 where the voxel is absent): the index
    map of splat_mean / splat_trilinear (oracle/network.py: splat), [n, 8] each."""
    inv_w = float(inv_w0) * 2.0 ** (-level)
    p = xyz.to(torch.float32) * inv_w
    base = torch.floor(p - 0.5).to(torch.int32)
    idx, wts = [], []
    for c in range(8):
        co = torch.tensor([c >> 2, (c >> 1) & 1, c & 1], dtype=torch.int32, device=xyz.device)
        ijk = (base + co[None]).contiguous()
        w = torch.prod(1.0 - torch.abs(p - (ijk.to(torch.float32) + 0.5)), dim=1)
        j = grid.ijk_to_index(ijk).long()
        ok = (j >= 0) & (w > 0)
        idx.append(torch.where(ok, j, torch.zeros_like(j)))
        wts.append(torch.where(ok, w, torch.zeros_like(w)))
    return torch.stack(idx, 1), torch.stack(wts, 1)


class EncoderFunction(torch.autograd.Function):
    """voxel_feat = splat_mean(point_mlp(xyz, feat)) as a function of the encoder's parameters."""

    @staticmethod
    def forward(ctx, module, enc, svh, depth, W1, b1, W2, b2):
        ctx.module, ctx.enc, ctx.svh, ctx.depth = module, enc, svh, depth
        ctx.save_for_backward(W1, b1, W2, b2)
        return module._forward_hip(enc, svh, depth)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gv):
        W1, b1, W2, b2 = [t.detach() for t in ctx.saved_tensors]
        enc, svh, d = ctx.enc, ctx.svh, ctx.depth
        grid = svh.level(d)
        idx, w = point_corners(grid, d, svh.inv_w0, enc.xyz)                          # [n, 8]
        ws = torch.zeros(grid.num_voxels, dtype=torch.float32, device=gv.device).index_add_(0, idx.reshape(-1), w.reshape(-1))
        coef = w / ws.clamp_min(1e-30)[idx] * (w > 0)                                 # mean: acc / ws where ws > 0
        g_g = (coef[..., None] * gv[idx]).sum(1)                                      # [n, C]
        p0 = enc.xyz.to(torch.float32) * float(svh.inv_w0)
        inp = torch.cat([(p0 - torch.floor(p0)) - 0.5, enc.feat.to(torch.float32)], 1)
        a1 = inp @ W1.t() + b1
        h = torch.relu(a1)
        ga = (g_g @ W2) * (a1 > 0).to(torch.float32)
        return None, None, None, None, ga.t() @ inp, ga.sum(0), g_g.t() @ h, g_g.sum(0)


class UNetFunction(torch.autograd.Function):
    """The U-Net's feature outputs as functions of the encoded voxel features and of its parameters.  ``tape`` (dict) receives what
    the reverse sweep needs from the forward; the flat output tuple is laid out by ``tape['layout']``."""

    @staticmethod
    def forward(ctx, module, enc, enc_svh, adaptive_depth, gt_decoder_svh, tape, voxel_feat, *params):
        ctx.module, ctx.tape, ctx.adaptive_depth = module, tape, adaptive_depth
        feat, dec_svh, _ = module._forward_impl(enc, enc_svh, adaptive_depth, gt_decoder_svh, tape)
        tape['feat'], tape['dec_svh'] = feat, dec_svh
        outs, layout = [], []
        D = enc_svh.depth
        for name, src in (('basis', feat.basis_features), ('normal', feat.normal_features), ('udf', feat.udf_features)):
            for d in range(D):
                if src[d] is not None:
                    layout.append((name, d))
                    outs.append(src[d])
        for d in range(D):
            layout.append(('structure', d))
            outs.append(feat.structure_features[d])
        tape['layout'] = layout
        ctx.n_params = len(params)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        m, tape = ctx.module, ctx.tape
        D = len(tape['dec'])
        g = {key: (go if go is not None else None) for key, go in zip(tape['layout'], gouts)}
        grads = {}                                                     # parameter name -> gradient

        def add(name, val):
            grads[name] = grads[name] + val if name in grads else val

        trunk = tape['trunk']
        gy = [torch.zeros_like(trunk[d]) for d in range(D)]
        for d in range(D):
            y = trunk[d]
            for kind, heads in (('basis', m.basis_heads), ('structure', m.structure_heads), ('udf', m.udf_heads)):
                go = g.get((kind, d))
                if go is None:
                    continue
                gx, gW, gb = linear_backward(y, heads[d].weight, go.to(torch.float32))
                gy[d] += gx
                add('%s_heads.%d.weight' % (kind, d), gW)
                add('%s_heads.%d.bias' % (kind, d), gb)
            go = g.get(('normal', d))
            if go is not None:
                nv, den, by_norm = tape['normal'][d]
                out = nv / den[:, None]
                go = go.to(torch.float32)
                # out = nv / den, den = |nv| where the norm is the largest of the three floors, a constant elsewhere
                gnv = go / den[:, None] - by_norm[:, None].to(torch.float32) * out * (out * go).sum(1, keepdim=True) / den[:, None]
                gx, gW, gb = linear_backward(y, m.normal_heads[d].weight, gnv)
                gy[d] += gx
                add('normal_heads.%d.weight' % d, gW)
                add('normal_heads.%d.bias' % d, gb)
        # decoder, finest level first: level d hands its parents' share up to level d + 1
        gx = [torch.zeros_like(tape['x'][d]) for d in range(D)]
        for d in range(D):
            rec = tape['dec'][d]
            if rec['exist'] is not None:
                gpre = torch.zeros_like(rec['y_pre'])
                gpre[rec['exist']] = gy[d]
            else:
                gpre = gy[d]
            gt, gW, gb = conv3_backward(rec['t'], rec['nbr'], m.up[d].weight, rec['y_pre'], gpre)
            add('up.%d.weight' % d, gW)
            add('up.%d.bias' % d, gb)
            je = rec['je'].long()
            ok = je >= 0
            gx[d].index_add_(0, je[ok], gt[ok])
            if rec['par'] is not None:
                gy[d + 1].index_add_(0, rec['par'].long(), gt)
        # down path, coarsest level first
        g_voxel = None
        for d in range(D - 1, -1, -1):
            gp, gW, gb = conv3_backward(tape['pool'][d], tape['enc_nbr'][d], m.down[d].weight, tape['x'][d], gx[d])
            add('down.%d.weight' % d, gW)
            add('down.%d.bias' % d, gb)
            if d == 0:
                g_voxel = gp
            else:
                st, en = tape['ranges'][d]
                nchild = tape['x'][d - 1].shape[0]
                kid = torch.arange(nchild, device=gp.device)
                par = torch.searchsorted(en.long(), kid, right=True).clamp_max(max(en.numel() - 1, 0))
                inside = (st.long()[par] <= kid) & (kid < en.long()[par])
                cnt = (en - st).to(torch.float32).clamp_min(1.0)
                gx[d - 1] += (gp[par] / cnt[par, None]) * inside[:, None].to(torch.float32)
        out = []
        for name, p in m.named_parameters():
            gr = grads.get(name)
            out.append(gr.reshape(p.shape).to(p.dtype) if gr is not None else torch.zeros_like(p))
        return (None, None, None, None, None, None, g_voxel) + tuple(out)
