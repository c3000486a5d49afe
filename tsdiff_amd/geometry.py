"""Mirror of the hot part of the reference `models/geometry.py:18-30`."""
import torch

from . import engine


def get_distance(pos, edge_index):
    """reference geometry.py:18-19 (plumbing: one gather + norm; the sampling path gets edge_length
    from the geometry kernel instead)."""
    return (pos[edge_index[0]] - pos[edge_index[1]]).norm(dim=-1)


def eq_transform(score_d, pos, edge_index, edge_length):
    """reference geometry.py:22-30 -> HIP kernel tsd_eq_transform."""
    return engine.eq_transform(score_d, pos, edge_index, edge_length)
