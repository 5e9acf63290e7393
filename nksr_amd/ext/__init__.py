"""MI355X-native counterparts of the reference's ``ext`` package (training ground-truth generators).
``ext.sdfgen.sdf_from_points`` -- ext/sdfgen/sdf_from_points.cu + ext/common/kdtree_cuda.cu in the reference."""
from . import sdfgen  # noqa: F401
