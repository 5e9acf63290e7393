"""Field base class -- host-side mirror of ``nksr.fields`` (base part).

Reference interface (call sites): ``field.extract_dual_mesh(mise_iter=, grid_upsample=,
max_points=)`` -> object with ``.v .f .c`` (examples/recons_simple.py:27,
recons_colored_mesh.py:30, models/nksr_net.py:214,284; NKSR-USAGE.md:52,79);
``field.evaluate_f(xyz, grad=)`` -> ``.value`` / ``.gradient`` (models/loss.py:189-198);
``set_mask_field`` / ``.mask_field`` / ``.svh`` (models/nksr_net.py:133, models/loss.py:132-133);
``set_texture_field`` (recons_colored_mesh.py:28); ``to_`` (NKSR-USAGE.md:163).
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class EvaluationResult:
    value: torch.Tensor
    gradient: Optional[torch.Tensor] = None


@dataclass
class MeshingResult:
    v: torch.Tensor
    f: torch.Tensor
    c: Optional[torch.Tensor] = None


class BaseField:
    def __init__(self, svh):
        self.svh = svh
        self.scale = 1.0          # world -> model units (Reconstructor's global scale)
        self.mask_field = None
        self.texture_field = None
        self.meshing_depth = 1     # levels whose dual cells are meshed (Reconstructor sets hparams.adaptive_depth)
        self.dual_graph = 'lattice'     # or 'adaptive': marching cubes on the dual graph of the flattened levels (meshing._extract_adaptive)

    @property
    def device(self):
        return self.svh.device

    def set_scale(self, scale):
        self.scale = float(scale)

    def set_mask_field(self, mask_field):
        self.mask_field = mask_field

    def set_texture_field(self, texture_field):
        self.texture_field = texture_field

    # model-unit evaluation, implemented by subclasses
    def _evaluate_f_model(self, xyz, grad):
        raise NotImplementedError

    def evaluate_f(self, xyz, grad=False):
        """f (and optionally its gradient) at world-space positions."""
        xyz = xyz.to(torch.float32)
        if self.scale != 1.0:
            xyz = xyz * self.scale
        res = self._evaluate_f_model(xyz.contiguous(), grad)
        if grad and res.gradient is not None and self.scale != 1.0:
            res.gradient = res.gradient * self.scale
        return res

    def evaluate_f_bar(self, xyz):
        """Value used for occupancy tests: f > 0 <=> inside (models/loss.py:99-100)."""
        return self.evaluate_f(xyz, grad=False).value

    def mask_vertices(self, xyz_model):
        """True where a mesh vertex (model units) survives trimming."""
        if self.mask_field is None:
            return None
        return self.mask_field.evaluate_mask(xyz_model)

    @torch.no_grad()       # a mesh is not differentiated: plain evaluation of the lattice values
    def extract_dual_mesh(self, mise_iter=0, grid_upsample=1, max_points=-1):
        from .. import meshing
        return meshing.extract_dual_mesh(self, mise_iter=mise_iter, grid_upsample=grid_upsample, max_points=max_points)

    def to_(self, device):
        raise NotImplementedError
