"""Alias of the reference's top-level ``ext`` package (``import ext; ext.sdfgen.sdf_from_points(...)``,
dataset/av_gt_geometry.py:64, models/loss.py:85) onto the MI355X implementation."""
from nksr_amd.ext import sdfgen  # noqa: F401
