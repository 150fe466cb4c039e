"""Drop-in at the MODULE level: code written against the reference imports from its sub-modules
(`from accvlab.batching_helpers.batched_processing_py import RaggedBatch`, as the reference's own tests do) — every
module path of the reference packages must resolve here and export the same public names."""
import importlib

import pytest
import torch

# module path in the reference (packages/<pkg>/accvlab/...) -> public names it defines
MODULES = {
    "accvlab.batching_helpers": ["RaggedBatch", "combine_data", "batched_bool_indexing", "batched_indexing_access",
                                 "batched_index_mapping", "get_mask_from_indices", "sum_over_targets"],
    "accvlab.batching_helpers.batched_bool_indexing": None,      # shadowed by the function of the same name (as upstream)
    "accvlab.batching_helpers.batched_index_mapping_op": ["BatchedIndexMapping", "batched_index_mapping", "RaggedBatch"],
    "accvlab.batching_helpers.batched_indexing_ops": ["BatchedIndexingAccess", "BatchedInverseIndexingAccessNewTensor",
                                                      "BatchedInverseIndexingAccessInsert", "batched_indexing_access",
                                                      "batched_inverse_indexing_access", "batched_indexing_write"],
    "accvlab.batching_helpers.batched_mask_from_indices": ["get_mask_from_indices"],
    "accvlab.batching_helpers.batched_processing_py": ["RaggedBatch", "average_over_targets", "sum_over_targets",
                                                       "apply_mask_to_tensor", "squeeze_except_batch_and_sample",
                                                       "get_compact_from_named_tuple", "get_compact_lists", "combine_data",
                                                       "get_indices_from_mask", "batched_bool_indexing"],
    "accvlab.batching_helpers.data_format": ["RaggedBatch"],
    "accvlab.batching_helpers.data_format.ragged_batch": ["RaggedBatch"],
    "accvlab.batching_helpers.data_format.set_padded_to": ["SetPaddedTo"],
    "accvlab.batching_helpers.batched_indexing_access_cuda": ["forward", "backward_new_tensor", "backward_insert",
                                                              "backward_insert_const", "map_values_by_index_pairs",
                                                              "get_mask_from_indices",
                                                              "set_ragged_batch_padded_to_filler_value_in_place"],
    "accvlab.batching_helpers.batched_indexing_access_cpu": ["set_ragged_batch_padded_to_filler_value_in_place"],
    "accvlab.draw_heatmap": ["draw_heatmap", "draw_heatmap_batched"],
    "accvlab.draw_heatmap.funtions": ["draw_heatmap", "draw_heatmap_batched"],
    "accvlab.draw_heatmap.draw_heatmap_ext": ["draw_heatmap", "draw_heatmap_batched_impl",
                                              "draw_heatmap_batched_classwise_impl"],
    "accvlab.multi_tensor_copier": ["start_copy", "AsyncCopyHandle"],
    "accvlab.multi_tensor_copier.async_copy": ["start_copy", "AsyncCopyHandle"],
    "accvlab.lane_helpers": ["polyline"],
    "accvlab.lane_helpers.polyline": ["interpolate", "lengths", "interpolate_var_size_batch", "lengths_var_size_batch"],
    "accvlab.lane_helpers.polyline.functions": ["interpolate", "lengths", "interpolate_var_size_batch",
                                                "lengths_var_size_batch"],
}


@pytest.mark.parametrize("name", sorted(MODULES))
def test_module_path_resolves_and_exports(name):
    mod = importlib.import_module(name)
    for attr in MODULES[name] or []:
        assert hasattr(mod, attr), f"{name} lacks {attr}"


def test_package_attribute_is_the_function_not_the_module():
    import accvlab.batching_helpers as bh
    import accvlab.batching_helpers.batched_bool_indexing  # noqa: F401  (the sub-module import must not rebind it)
    import accvlab.draw_heatmap as dh
    import accvlab.draw_heatmap.funtions.draw_heatmap_batched  # noqa: F401

    assert callable(bh.batched_bool_indexing) and callable(dh.draw_heatmap_batched)
    from accvlab.batching_helpers.batched_processing_py import RaggedBatch
    assert RaggedBatch is bh.RaggedBatch


def test_lane_helpers_validation_messages_match_the_reference():
    # check_type / check_sample_sizes of ext_impl/polyline/src/polyline.cpp:54-81
    from accvlab.batching_helpers import RaggedBatch
    from accvlab.lane_helpers import polyline

    pts = torch.randn(1, 3, 2)
    dist = torch.randn(1, 4)
    good_d = RaggedBatch(dist, sample_sizes=torch.tensor([4]))
    with pytest.raises(RuntimeError, match="points.sample_sizes values must be in"):
        polyline.interpolate_var_size_batch(RaggedBatch(pts, sample_sizes=torch.tensor([-1])), good_d)
    with pytest.raises(RuntimeError, match="distances.sample_sizes values must be in"):
        polyline.interpolate_var_size_batch(RaggedBatch(pts, sample_sizes=torch.tensor([3])),
                                            RaggedBatch(dist, sample_sizes=torch.tensor([5])))
    with pytest.raises(RuntimeError, match="points.sample_sizes"):
        polyline.lengths_var_size_batch(RaggedBatch(pts, sample_sizes=torch.tensor([4])))
    for dtype in (torch.float16, torch.bfloat16):
        with pytest.raises(RuntimeError, match="float32 or float64 on CPU"):
            polyline.lengths(pts.to(dtype))
        with pytest.raises(RuntimeError, match="float32 or float64 on CPU"):
            polyline.interpolate(pts.to(dtype), dist.to(dtype))
    with pytest.raises(RuntimeError, match="same dtype"):
        polyline.interpolate_var_size_batch(RaggedBatch(pts, sample_sizes=torch.tensor([3])), good_d.double())
