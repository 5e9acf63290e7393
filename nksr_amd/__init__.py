"""nksr_amd -- MI355X-native implementation of the NKSR solve-time hot path.

Exports the surface of the reference's ``nksr`` package that its examples and training glue
use: ``Reconstructor``, ``NKSRNetwork``, ``SparseFeatureHierarchy``,
``get_estimate_normal_preprocess_fn`` and the sub-modules ``fields``, ``svh``, ``configs``,
``utils``.  ``import nksr`` resolves to this package through the top-level ``nksr`` shim.
"""
from . import configs, fields, svh, utils
from .nn.network import NKSRNetwork
from .preprocess import get_estimate_normal_preprocess_fn
from .reconstructor import Reconstructor
from .svh import SparseFeatureHierarchy

__all__ = ['Reconstructor', 'NKSRNetwork', 'SparseFeatureHierarchy', 'get_estimate_normal_preprocess_fn',
           'fields', 'svh', 'configs', 'utils']
__version__ = '0.1.0'
