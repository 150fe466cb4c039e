"""Name of the reference's compiled extension module (csrc/draw_heatmap.cpp:131-144): the three entry points it exports,
over the C-ABI of libaccv_hip.so.
"""
from .ops import draw_heatmap, draw_heatmap_batched as _batched  # noqa: F401


class _Ragged:
    def __init__(self, tensor, sample_sizes):
        self.tensor, self.sample_sizes = tensor, sample_sizes


def draw_heatmap_batched_impl(heatmaps, centers, radii, nums_targets, diameter_to_sigma_factor=6.0, k_scale=1.0):
    """``draw_heatmap_batched_impl(heatmaps, centers, radii, nums_targets, factor, k)`` of the reference extension."""
    _batched(heatmaps, _Ragged(centers, nums_targets), _Ragged(radii, nums_targets), diameter_to_sigma_factor, k_scale)


def draw_heatmap_batched_classwise_impl(heatmaps, centers, radii, nums_targets, labels, diameter_to_sigma_factor=6.0,
                                        k_scale=1.0):
    """``draw_heatmap_batched_classwise_impl(heatmaps, centers, radii, nums_targets, labels, factor, k)``."""
    _batched(heatmaps, _Ragged(centers, nums_targets), _Ragged(radii, nums_targets), diameter_to_sigma_factor, k_scale,
             _Ragged(labels, nums_targets))
