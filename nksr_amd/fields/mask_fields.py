"""Mask / texture fields -- host-side mirror of ``nksr.fields.{LayerField, NeuralField,
PCNNField}``.

Reference interface (call sites): ``LayerField(dec_svh, adaptive_depth)`` models/nksr_net.py:132;
``NeuralField(svh=, decoder=, features=)`` + ``set_level_set(2 * voxel_size)`` :115-130;
``PCNNField(xyz, color)`` examples/recons_colored_mesh.py:28.
"""
import torch

from .base_field import BaseField


class LayerField(BaseField):
    """Keeps geometry that lies inside the voxels of the first ``adaptive_depth`` levels."""

    def __init__(self, svh, adaptive_depth):
        super().__init__(svh)
        self.adaptive_depth = int(adaptive_depth)

    def evaluate_mask(self, xyz_model):
        keep = torch.zeros(xyz_model.shape[0], dtype=torch.bool, device=xyz_model.device)
        for d in range(min(self.adaptive_depth, self.svh.depth)):
            g = self.svh.level(d)
            if g.num_voxels == 0:
                continue
            ijk = torch.floor(xyz_model / g.voxel_size).to(torch.int32)
            keep |= g.ijk_to_index(ijk) >= 0
        return keep

    def to_(self, device):
        return self


class NeuralField(BaseField):
    """Scalar field decoded from per-voxel features (the UDF mask branch).  Keeps vertices
    whose decoded value is below ``level_set``."""

    def __init__(self, svh, decoder, features):
        super().__init__(svh)
        self.decoder = decoder
        self.features = features
        self.level_set = 0.0

    def set_level_set(self, level_set):
        self.level_set = float(level_set)

    def _evaluate_f_model(self, xyz, grad):
        from .base_field import EvaluationResult
        return EvaluationResult(self.decoder(xyz, self.svh, self.features), None)

    def evaluate_mask(self, xyz_model):
        return self._evaluate_f_model(xyz_model, False).value < self.level_set

    def to_(self, device):
        device = torch.device(device)
        if isinstance(self.features, (list, tuple)):
            self.features = [None if f is None else f.to(device) for f in self.features]
        elif torch.is_tensor(self.features):
            self.features = self.features.to(device)
        if self.svh.device != device:     # usually shared with (and already moved by) the output field
            self.svh.to_(device)
        return self


class PCNNField(BaseField):
    """Nearest-neighbour colour field over the input cloud (texture for ``mesh.c``)."""

    def __init__(self, xyz, color):
        self.svh = None
        self.scale = 1.0
        self.mask_field = None
        self.texture_field = None
        self.xyz = xyz.to(torch.float32).contiguous()
        self.color = color.to(torch.float32).contiguous()

    @property
    def device(self):
        return self.xyz.device

    def evaluate_color(self, xyz_world):
        """Colour of the nearest input point (grid-hash nearest neighbour, csrc/knn.hip)."""
        from ..normals import PointGrid, choose_cell_size
        if getattr(self, '_grid', None) is None or self._grid.xyz.device != xyz_world.device:
            ref = self.xyz.to(xyz_world.device)
            self._grid = PointGrid(ref, choose_cell_size(ref, 8))
        idx = self._grid.nearest(xyz_world, max_ring=64)
        col = self.color.to(xyz_world.device)
        out = col[idx.clamp_min(0)]
        out[idx < 0] = 0.0
        return out

    def to_(self, device):
        self.xyz, self.color = self.xyz.to(device), self.color.to(device)
        return self
