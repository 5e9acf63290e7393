"""Drop-in alias: ``import nksr`` -> nksr_amd (the reference's examples import ``nksr``,
e.g. examples/recons_simple.py:11, ``from nksr import Reconstructor, utils, fields``
recons_colored_mesh.py:12, ``from nksr.fields import KernelField`` models/nksr_net.py:16)."""
import sys

import nksr_amd
from nksr_amd import *  # noqa: F401,F403
from nksr_amd import configs, fields, svh, utils  # noqa: F401

for _name in ('configs', 'fields', 'svh', 'utils'):
    sys.modules['nksr.' + _name] = getattr(nksr_amd, _name)
__version__ = nksr_amd.__version__
