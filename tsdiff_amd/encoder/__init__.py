"""Mirrors of the secondary (legacy dual-encoder) encoders of the reference `models/encoder/`."""
from .gin import GINEConv, GINEncoder  # noqa: F401
from .edge import GaussianSmearingEdgeEncoder  # noqa: F401
