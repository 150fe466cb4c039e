"""Seeded random sweeps of the HIP path against the CPU oracle over geometry that the hand-written cases do not enumerate:
odd map sizes, objects far outside the frame, zero / huge / negative radii, negative k, sparse and dense counts, int32 and
int64 counts, every public dispatch hint; ragged gathers / scatters / compactions of random shapes and dtypes.
Tolerance: 1e-5 absolute on heat-map values (north_star), bit-exact on everything that moves bytes or produces indices."""
from types import SimpleNamespace

import numpy as np
import pytest


def _seeds(n):
    """seed range of a sweep; ACCV_FUZZ_SCALE=k runs k times as many seeds (soak runs: profiles/r03_fuzz_soak.log)"""
    import os
    return range(n * max(1, int(os.environ.get("ACCV_FUZZ_SCALE", "1"))))

import torch

from oracle import h1 as oracle
from oracle import h2 as oracle_h2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def rb(t, sizes):
    return SimpleNamespace(tensor=t, sample_sizes=sizes)


def _random_h1_case(rng):
    h = int(rng.choice([1, 2, 7, 15, 16, 17, 33, 64, 100, 129, 250]))
    w = int(rng.choice([1, 3, 4, 5, 8, 30, 64, 124, 128, 132, 260, 516]))
    b = int(rng.integers(1, 5))
    n_max = int(rng.choice([0, 1, 2, 5, 63, 64, 65, 130]))
    counts = rng.integers(-2, n_max + 3, size=b)
    span = max(h, w)
    cx = rng.integers(-span, w + span, size=(b, n_max))
    cy = rng.integers(-span, h + span, size=(b, n_max))
    kind = rng.integers(0, 10, size=(b, n_max))
    radii = rng.integers(0, 12, size=(b, n_max))
    radii = np.where(kind == 0, rng.integers(0, 4 * span + 2, size=(b, n_max)), radii)      # huge
    radii = np.where(kind == 1, -rng.integers(1, 5, size=(b, n_max)), radii)                # negative: never drawn
    radii = np.where(kind == 2, 0, radii)
    centers = np.stack([cx, cy], -1).astype(np.int32)
    return h, w, b, n_max, counts, centers, radii.astype(np.int32)


@pytest.mark.parametrize("seed", _seeds(12))
def test_h1_random_geometry_against_the_oracle(seed):
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_heatmap_batched, ops

    rng = np.random.default_rng(1000 + seed)
    hints = [0, nat.HM_TILE_ROWS_16, nat.HM_SMALL_RADII, nat.HM_WRITE_THROUGH, nat.HM_PLAIN_STORES,
             nat.HM_TILE_ROWS_16 | nat.HM_WRITE_THROUGH]
    for case in range(10):
        h, w, b, n_max, counts, centers, radii = _random_h1_case(rng)
        k = float(rng.choice([1.0, 0.8, -0.5, 2.0]))
        factor = float(rng.choice([3.0, 6.0, 12.0]))
        clear = bool(rng.integers(0, 2))
        classwise = bool(rng.integers(0, 3) == 0)
        ncls = int(rng.integers(1, 4)) if classwise else 0
        labels = rng.integers(-1, ncls + 1, size=(b, n_max)).astype(np.int32) if classwise else None
        counts_t = torch.from_numpy(counts.astype(np.int32 if case % 2 else np.int64))
        shape = (b, ncls, h, w) if classwise else (b, h, w)
        base = (rng.random(shape, dtype=np.float32) - 0.3).astype(np.float32)
        want = base.copy()
        oracle.draw_heatmap_batched(want, centers, radii, np.clip(counts, 0, n_max), labels=labels, factor=factor, k=k, clear=clear)
        got = torch.from_numpy(base.copy()).to(DEV)
        c_t, r_t, n_t = torch.from_numpy(centers).to(DEV), torch.from_numpy(radii).to(DEV), counts_t.to(DEV)
        ops._FORCED_FLAGS = hints[(seed + case) % len(hints)]
        try:
            draw_heatmap_batched(got, rb(c_t, n_t), rb(r_t, n_t), factor, k,
                                 labels=rb(torch.from_numpy(labels).to(DEV), n_t) if classwise else None, clear=clear)
        finally:
            ops._FORCED_FLAGS = 0
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
        assert err <= 1e-5, f"seed {seed} case {case}: {shape}, n_max {n_max}, clear {clear}, k {k}: max abs err {err}"


# the dtype set of the reference's gather / scatter dispatch (batched_indexing_access_helpers.h:60-100); compaction takes more
_DTYPES = [torch.float32, torch.float16, torch.int64, torch.int32, torch.float64, torch.bfloat16]
_MASK_DTYPES = _DTYPES + [torch.uint8, torch.bool]


@pytest.mark.parametrize("seed", _seeds(8))
def test_h2_random_gather_scatter_compaction_against_the_oracle(seed):
    from accvlab.batching_helpers import RaggedBatch, batched_bool_indexing, batched_indexing_access, \
        batched_inverse_indexing_access, get_indices_from_mask

    rng = np.random.default_rng(2000 + seed)
    for case in range(6):
        b = int(rng.integers(1, 6))
        n_src = int(rng.choice([1, 2, 7, 64, 65, 200]))
        n_idx = int(rng.choice([0, 1, 3, 33, 64, 130]))
        inner = tuple(int(x) for x in rng.choice([1, 2, 3, 4, 8, 17], size=int(rng.integers(0, 3))))
        dt = _DTYPES[(seed + case) % len(_DTYPES)]
        data = (torch.rand((b, n_src) + inner) * 100).to(dt)
        idx = torch.from_numpy(rng.integers(-n_src, n_src, size=(b, n_idx)))
        counts = torch.from_numpy(rng.integers(0, n_idx + 1, size=b))
        fill = 7
        # gather
        got = batched_indexing_access(data.to(DEV), RaggedBatch(idx.to(DEV), sample_sizes=counts.to(DEV)), fill)
        want = oracle_h2.gather(data.view(torch.int16).numpy() if dt == torch.bfloat16 else data.numpy(), idx.numpy(), counts.numpy(),
                                torch.tensor(fill, dtype=dt).view(torch.int16).item() if dt == torch.bfloat16 else fill)
        g = got.tensor.cpu()
        assert np.array_equal(g.view(torch.int16).numpy() if dt == torch.bfloat16 else g.numpy(), want), f"gather seed {seed} case {case}"
        # inverse (scatter into a fresh tensor): unique indices so that the result does not depend on write order
        n_tgt = n_src + 3
        perm = np.stack([rng.permutation(n_tgt)[:n_idx] if n_idx <= n_tgt else np.resize(rng.permutation(n_tgt), n_idx) for _ in range(b)])
        if n_idx <= n_tgt:
            src = (torch.rand((b, n_idx) + inner) * 50).to(dt)
            got = batched_inverse_indexing_access(RaggedBatch(src.to(DEV), sample_sizes=counts.to(DEV)),
                                                  RaggedBatch(torch.from_numpy(perm).to(DEV), sample_sizes=counts.to(DEV)), n_tgt, fill)
            want = oracle_h2.scatter_new(src.view(torch.int16).numpy() if dt == torch.bfloat16 else src.numpy(), perm, counts.numpy(), n_tgt,
                                         torch.tensor(fill, dtype=dt).view(torch.int16).item() if dt == torch.bfloat16 else fill, False)
            g = (got.tensor if hasattr(got, "tensor") else got).cpu()
            assert np.array_equal(g.view(torch.int16).numpy() if dt == torch.bfloat16 else g.numpy(), want), f"inverse seed {seed} case {case}"
        # compaction by mask + indices from mask
        mask = torch.from_numpy(rng.random((b, n_src)) < rng.random())
        dt = _MASK_DTYPES[(seed + case) % len(_MASK_DTYPES)]
        data = (torch.rand((b, n_src) + inner) * 100).to(dt) if dt != torch.bool else torch.rand((b, n_src) + inner) > 0.5
        comp = batched_bool_indexing(data.to(DEV), mask.to(DEV))
        want_rows, want_sizes = oracle_h2.bool_compact(data.view(torch.int16).numpy() if dt == torch.bfloat16 else data.numpy(), mask.numpy())
        assert np.array_equal(comp.sample_sizes.cpu().numpy(), want_sizes)
        c = comp.tensor.cpu()
        c = c.view(torch.int16).numpy() if dt == torch.bfloat16 else c.numpy()
        for i in range(b):
            assert np.array_equal(c[i, : want_sizes[i]], want_rows[i][: want_sizes[i]]), f"compaction seed {seed} case {case} sample {i}"
        ind = get_indices_from_mask(mask.to(DEV))
        want_idx, want_cnt = oracle_h2.indices_from_mask(mask.numpy())
        assert np.array_equal(ind.sample_sizes.cpu().numpy(), want_cnt)
        for i in range(b):
            assert np.array_equal(ind.tensor[i, : want_cnt[i]].cpu().numpy(), want_idx[i][: want_cnt[i]])


@pytest.mark.parametrize("seed", _seeds(6))
def test_h1_flat_api_random_planes_against_the_oracle(seed):
    """draw_heatmap (flat input): random plane indices incl. out-of-range ones (ignored), empty planes, N below and above the
    single-launch binning limit"""
    from accvlab.draw_heatmap import draw_heatmap

    rng = np.random.default_rng(3000 + seed)
    for case in range(6):
        p = int(rng.choice([1, 2, 5, 33, 300]))
        h = int(rng.choice([1, 9, 32, 100]))
        w = int(rng.choice([4, 30, 64, 260]))
        n = int(rng.choice([0, 1, 7, 200, 3000]))
        centers = np.stack([rng.integers(-20, w + 20, size=n), rng.integers(-20, h + 20, size=n)], -1).astype(np.int32)
        radii = rng.integers(-1, 15, size=n).astype(np.int32)
        idx = rng.integers(-2, p + 2, size=n).astype(np.int32)
        k = float(rng.choice([1.0, 0.7]))
        base = (rng.random((p, h, w), dtype=np.float32) * 0.4).astype(np.float32)
        want = base.copy()
        oracle.draw_heatmap_flat(want, centers, radii, idx, 6.0, k)
        got = torch.from_numpy(base.copy()).to(DEV)
        draw_heatmap(got, torch.from_numpy(centers).to(DEV), torch.from_numpy(radii).to(DEV), torch.from_numpy(idx).to(DEV), 6.0, k)
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
        assert err <= 1e-5, f"seed {seed} case {case}: P {p} {h}x{w} N {n}: max abs err {err}"


@pytest.mark.parametrize("seed", _seeds(6))
def test_h3_random_trees_round_trip_bit_exact(seed):
    """start_copy over random nestings / dtypes / empty, non-contiguous and numpy leaves, to the GPU and back, both modes, several
    packing parameters, also through the DataLoader hook (pack_batch): structure, dtypes, shapes and bytes are preserved"""
    import test_fuzz_cpu as m

    from accvlab.multi_tensor_copier import pack_batch, start_copy

    rng = np.random.default_rng(4000 + seed)
    for case in range(5):
        tree = [m._random_tree(rng) for _ in range(int(rng.integers(1, 12)))]
        kw = dict(use_background_thread=bool(rng.integers(0, 2)), pack_cpu_tensors=bool(rng.integers(0, 4) > 0),
                  min_packed_alignment_bytes=int(rng.choice([1, 6, 16, 64])), max_packed_chunk_bytes=int(rng.choice([256, 4096, 32 << 20])),
                  use_pinned_staging=bool(rng.integers(0, 4) > 0))
        on_gpu = start_copy(tree, DEV, **kw).get()
        assert m._same_tree(tree, on_gpu), f"seed {seed} case {case} {kw}: host -> GPU"
        back = start_copy(on_gpu, "cpu", **kw).get()
        assert m._same_tree(tree, back), f"seed {seed} case {case} {kw}: GPU -> host"
        via_hook = start_copy(pack_batch(tree, min_packed_alignment_bytes=kw["min_packed_alignment_bytes"]), DEV,
                              use_background_thread=kw["use_background_thread"]).get()
        assert m._same_tree(tree, via_hook), f"seed {seed} case {case}: pack_batch -> GPU"


def _np(t):
    t = t.cpu()
    return t.view(torch.int16).numpy() if t.dtype == torch.bfloat16 else t.numpy()


@pytest.mark.parametrize("seed", _seeds(8))
def test_h2_random_write_mapping_mask_padfill_boolwrite_against_the_oracle(seed):
    """the writing half of batching_helpers: indexing write, index mapping, mask from indices, pad fill, boolean write-back —
    random shapes, dtypes, negative (wrapping) indices, ragged masks"""
    from accvlab.batching_helpers import RaggedBatch, batched_bool_indexing_write, batched_index_mapping, \
        batched_indexing_write, get_mask_from_indices

    rng = np.random.default_rng(5000 + seed)
    for case in range(6):
        b = int(rng.integers(1, 6))
        n_tgt = int(rng.choice([1, 3, 64, 65, 150]))
        n_src = int(rng.choice([1, 5, 70]))
        n_idx = int(rng.choice([0, 1, 4, 40])) if n_tgt >= 40 else int(rng.integers(0, n_tgt + 1))
        n_idx = min(n_idx, n_tgt)
        inner = tuple(int(x) for x in rng.choice([1, 2, 5, 16], size=int(rng.integers(0, 3))))
        dt = _DTYPES[(seed + case) % len(_DTYPES)]
        counts = torch.from_numpy(rng.integers(0, n_idx + 1, size=b))
        # unique target indices per sample, half of them written as negative (wrap once, cu:75-77)
        tgt = np.stack([rng.permutation(n_tgt)[:n_idx] for _ in range(b)]).reshape(b, n_idx).astype(np.int64)
        tgt = np.where(rng.random(tgt.shape) < 0.5, tgt - n_tgt, tgt)
        src_idx = rng.integers(-n_src, n_src, size=(b, n_idx)).astype(np.int64)
        into = (torch.rand((b, n_tgt) + inner) * 90).to(dt)
        vals = (torch.rand((b, n_idx) + inner) * 90).to(dt)
        src = (torch.rand((b, n_src) + inner) * 90).to(dt)
        tgt_rb = RaggedBatch(torch.from_numpy(tgt).to(DEV), sample_sizes=counts.to(DEV))
        got = batched_indexing_write(RaggedBatch(vals.to(DEV), sample_sizes=counts.to(DEV)), tgt_rb, into.to(DEV))
        assert np.array_equal(_np(got), oracle_h2.scatter_insert(_np(vals), tgt, counts.numpy(), _np(into))), f"write {seed}/{case}"
        got = batched_index_mapping(src.to(DEV), RaggedBatch(torch.from_numpy(src_idx).to(DEV), sample_sizes=counts.to(DEV)), tgt_rb,
                                    into.to(DEV))
        assert np.array_equal(_np(got), oracle_h2.map_pairs(_np(src), src_idx, tgt, counts.numpy(), _np(into))), f"mapping {seed}/{case}"
        got = get_mask_from_indices(n_tgt, tgt_rb)
        assert np.array_equal(got.cpu().numpy(), oracle_h2.mask_from_indices(tgt, counts.numpy(), n_tgt)), f"mask {seed}/{case}"
        # pad fill through the RaggedBatch method, bool write-back through a ragged mask
        sizes = torch.from_numpy(rng.integers(0, n_tgt + 1, size=b))
        filled = RaggedBatch(into.clone().to(DEV), sample_sizes=sizes.to(DEV)).with_padded_set_to(3)
        want = oracle_h2.pad_fill(_np(into), sizes.numpy(), torch.tensor(3, dtype=dt).view(torch.int16).item() if dt == torch.bfloat16 else 3)
        assert np.array_equal(_np(filled.tensor), want), f"pad fill {seed}/{case}"
        mask = torch.from_numpy(rng.random((b, n_tgt)) < 0.4)
        valid = sizes
        n_true = np.array([int(mask[i, : int(valid[i])].sum()) for i in range(b)])
        wsz = torch.from_numpy(np.array([int(rng.integers(0, n_true[i] + 1)) if rng.integers(0, 2) else n_true[i] for i in range(b)]))
        width = int(max(int(n_true.max()) if b else 0, 1))
        to_write = (torch.rand((b, width) + inner) * 90).to(dt)
        got = batched_bool_indexing_write(RaggedBatch(to_write.to(DEV), sample_sizes=wsz.to(DEV)),
                                          RaggedBatch(mask.to(DEV), sample_sizes=valid.to(DEV)), into.to(DEV))
        want = oracle_h2.bool_write(_np(to_write), wsz.numpy(), mask.numpy(), _np(into), valid.numpy())
        assert np.array_equal(_np(got.tensor if hasattr(got, "tensor") else got), want), f"bool write {seed}/{case}"


@pytest.mark.parametrize("seed", _seeds(6))
def test_lane_sampler_random_polylines_against_the_oracle(seed):
    """interpolate / lengths (fixed and ragged): random point counts incl. 0 and 1, repeated points (zero-length segments),
    queries before the start and beyond the end, relative mode, 2-D and 3-D points.  atol 1e-5 relative to the polyline scale
    (packages/lane_helpers/tests/polyline_test_utils.py:78-115 uses atol=1e-5 on unit-scale data)."""
    from oracle import lane as oracle_lane

    from accvlab.batching_helpers import RaggedBatch
    from accvlab.lane_helpers.polyline import interpolate_var_size_batch, lengths_var_size_batch

    rng = np.random.default_rng(6000 + seed)
    for case in range(5):
        b = int(rng.integers(1, 7))
        p_max = int(rng.choice([1, 2, 5, 24, 70]))
        q_max = int(rng.choice([1, 3, 64, 130]))
        dims = int(rng.choice([2, 3]))
        relative = bool(rng.integers(0, 2))
        pts = rng.random((b, p_max, dims)).astype(np.float32) * 4.0
        dup = rng.random((b, p_max)) < 0.15                     # repeated points: zero-length segments
        for i in range(b):
            for j in range(1, p_max):
                if dup[i, j]:
                    pts[i, j] = pts[i, j - 1]
        n_pts = rng.integers(0, p_max + 1, size=b)
        n_q = rng.integers(0, q_max + 1, size=b)
        dist = (rng.random((b, q_max)).astype(np.float32) * 1.4 - 0.2) * (1.0 if relative else 6.0)
        got = interpolate_var_size_batch(RaggedBatch(torch.from_numpy(pts).to(DEV), sample_sizes=torch.from_numpy(n_pts).to(DEV)),
                                         RaggedBatch(torch.from_numpy(dist).to(DEV), sample_sizes=torch.from_numpy(n_q).to(DEV)),
                                         relative=relative)
        g = got.tensor.cpu().numpy()
        lens = lengths_var_size_batch(RaggedBatch(torch.from_numpy(pts).to(DEV), sample_sizes=torch.from_numpy(n_pts).to(DEV))).cpu().numpy()
        for i in range(b):
            want = oracle_lane.sample(pts[i, : n_pts[i]], dist[i, : n_q[i]], relative)
            have = g[i, : n_q[i]]
            assert np.array_equal(np.isnan(have), np.isnan(want)), f"lane {seed}/{case} sample {i}: NaN pattern"
            assert np.allclose(have, want, atol=4e-5, rtol=0, equal_nan=True), f"lane {seed}/{case} sample {i}"
            wl_ = oracle_lane.length(pts[i, : n_pts[i]])
            assert (np.isnan(wl_) and np.isnan(lens[i])) or abs(lens[i] - wl_) <= 4e-5, f"lane length {seed}/{case} sample {i}"


@pytest.mark.parametrize("seed", _seeds(5))
def test_wide_mask_compaction_random_against_the_oracle(seed):
    """mask -> indices on the one-wave, the multi-wave and the segmented (two-pass, >= 8192 columns) paths: random widths that
    are no multiples of the segment size, random densities incl. all-False / all-True rows, ragged validity"""
    from accvlab.batching_helpers import RaggedBatch, get_indices_from_mask

    rng = np.random.default_rng(7000 + seed)
    for case in range(5):
        b = int(rng.integers(1, 10))
        w = int(rng.choice([1, 63, 64, 65, 511, 513, 4095, 4097, 8191, 8192, 8193, 12289, 40001, 70000]))
        dens = rng.random(b) * rng.choice([0.0, 0.02, 0.5, 1.0])
        mask = rng.random((b, w)) < dens[:, None]
        if b > 1:
            mask[0] = True
            mask[-1] = False
        valid = rng.integers(0, w + 1, size=b)
        ragged = bool(rng.integers(0, 2))
        m = torch.from_numpy(mask).to(DEV)
        got = get_indices_from_mask(RaggedBatch(m, sample_sizes=torch.from_numpy(valid).to(DEV)) if ragged else m)
        want_idx, want_cnt = oracle_h2.indices_from_mask(mask, valid if ragged else None)
        assert np.array_equal(got.sample_sizes.cpu().numpy(), want_cnt), f"counts {seed}/{case} w {w}"
        g = got.tensor.cpu().numpy()
        assert g.shape[1] == (int(want_cnt.max()) if b else 0)
        for i in range(b):
            assert np.array_equal(g[i, : want_cnt[i]], want_idx[i][: want_cnt[i]]), f"indices {seed}/{case} w {w} row {i}"


@pytest.mark.parametrize("seed", _seeds(6))
def test_ragged_batch_shape_operations_on_the_gpu_against_the_list_model(seed):
    """the model-based sweep of tests/test_fuzz_cpu.py on GPU tensors (mask and pad-fill kernels, split without read-back)"""
    import test_fuzz_cpu as m

    rng = np.random.default_rng(9900 + seed)
    for case in range(6):
        cpu_batch, nb, nu_pos = m._random_ragged(rng)
        leaves = m._model(cpu_batch)
        rbatch = cpu_batch.to_device(DEV)
        shape = tuple(rbatch.batch_shape)

        def check(rb_, want, want_shape, what):
            assert tuple(rb_.batch_shape) == tuple(want_shape), what
            got = m._leaves(rb_.split(), rb_.num_batch_dims)
            assert len(got) == len(want), what
            for g, w in zip(got, want):
                assert g.is_cuda and tuple(g.shape) == tuple(w.shape) and torch.equal(g.cpu(), w), what
            cols = torch.arange(rb_.tensor.shape[rb_.non_uniform_dim], device=DEV)
            assert torch.equal(rb_.mask, cols < rb_.sample_sizes.unsqueeze(-1)), what + ": mask"

        check(rbatch, leaves, shape, "identity")
        check(rbatch.flatten_batch_dims(), leaves, (len(leaves),), "flatten_batch_dims")
        n_data = rbatch.tensor.dim() - nb
        new_pos = int(rng.integers(0, n_data))
        check(rbatch.get_non_uniform_dimension_transposed_to(nb + new_pos), [l.transpose(nu_pos, new_pos) for l in leaves], shape,
              "get_non_uniform_dimension_transposed_to")
        bd, k = int(rng.integers(0, nb)), int(rng.integers(1, 4))
        idx = np.tile(np.arange(len(leaves)).reshape(shape), [k if a == bd else 1 for a in range(nb)])
        check(rbatch.repeat_samples(k, bd), [leaves[j] for j in idx.reshape(-1)], idx.shape, "repeat_samples")
        filled = rbatch.with_padded_set_to(-7.0)
        check(filled, leaves, shape, "with_padded_set_to")
        t = filled.tensor.cpu()
        pos = torch.arange(t.shape[filled.non_uniform_dim]).reshape([-1 if a == filled.non_uniform_dim else 1 for a in range(t.dim())])
        pad = (pos >= cpu_batch.sample_sizes.reshape(shape + (1,) * (t.dim() - nb))).expand(t.shape)
        assert bool((t[pad] == -7.0).all())
        check(rbatch.cpu().to_device(DEV), leaves, shape, "cpu -> device round trip")


@pytest.mark.parametrize("seed", _seeds(5))
def test_matched_pair_loss_random_against_the_composition(seed):
    """fused gather + loss + masked sum against the same thing composed from the package's operators (float64 on the host)"""
    from accvlab.batching_helpers import RaggedBatch, matched_pair_loss_sum

    rng = np.random.default_rng(9950 + seed)
    for case in range(5):
        b, na, nbb, d = int(rng.integers(1, 7)), int(rng.choice([1, 9, 100, 300])), int(rng.choice([1, 7, 64])), int(rng.choice([1, 4, 7, 10]))
        k = int(rng.integers(0, min(na, nbb) + 1))
        a = torch.from_numpy(rng.standard_normal((b, na, d)).astype(np.float32))
        bb = torch.from_numpy(rng.standard_normal((b, nbb, d)).astype(np.float32))
        ia = np.stack([rng.permutation(na)[:k] for _ in range(b)]).reshape(b, k)
        ib = np.stack([rng.permutation(nbb)[:k] for _ in range(b)]).reshape(b, k)
        counts = rng.integers(0, k + 1, size=b)
        kind = ["l1", "l2", "smooth_l1"][(seed + case) % 3]
        got = matched_pair_loss_sum(a.to(DEV), bb.to(DEV),
                                    RaggedBatch(torch.from_numpy(ia).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV)),
                                    RaggedBatch(torch.from_numpy(ib).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV)), kind=kind)
        want = np.zeros(b)
        for i in range(b):
            for j in range(counts[i]):
                diff = a[i, ia[i, j]].double().numpy() - bb[i, ib[i, j]].double().numpy()
                if kind == "l1":
                    want[i] += np.abs(diff).sum()
                elif kind == "l2":
                    want[i] += (diff * diff).sum()
                else:
                    ad = np.abs(diff)
                    want[i] += np.where(ad < 1.0, 0.5 * diff * diff, ad - 0.5).sum()
        assert np.allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-5), f"matched loss {seed}/{case} {kind}"


@pytest.mark.parametrize("seed", _seeds(6))
def test_multiscale_and_lane_raster_random_against_the_per_scale_operators(seed):
    """one-launch multi-scale box maps and lane rasters == the per-scale operators, bit for bit, over random strides, map shapes
    (incl. ones the fused kernels do not take), counts incl. zero, lanes with few / no points, fused clear and in-place"""
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import (draw_heatmap_batched, draw_heatmap_multiscale, draw_polylines_batched, draw_polylines_multiscale,
                                      get_centers_and_radii)
    from accvlab.draw_heatmap.lanes import _draw_polylines_via_targets as _via_targets

    rng = np.random.default_rng(9970 + seed)
    for case in range(4):
        b = int(rng.integers(1, 5))
        sw, sh = int(rng.choice([64, 200, 512, 1000])), int(rng.choice([32, 96, 300]))
        n_scales = int(rng.integers(1, 5))
        strides = [float(x) for x in rng.choice([1.0, 2.0, 3.0, 4.0, 5.0, 8.0, 16.0], size=n_scales, replace=False)]
        clear = bool(rng.integers(0, 2))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        cs, bs = [], []
        for _ in range(b):
            n = int(rng.integers(0, 20))
            c = torch.rand(n, 2, generator=g) * torch.tensor([sw, sh])
            half = torch.rand(n, 4, generator=g) * min(sw, sh) * 0.3
            cs.append(c)
            bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
        crb = combine_data(cs, device=DEV)
        brb = combine_data(bs, device=DEV, other_with_same_sample_sizes=crb)
        shapes = [(b, max(1, int(sh / s)), max(1, int(sw / s))) for s in strides]
        base = [(torch.rand(s_, generator=g) * 0.3).to(DEV) for s_ in shapes]
        fused = [t.clone() for t in base]
        draw_heatmap_multiscale(fused, crb, brb, strides, 6.0, 0.9, clear=clear)
        for i, s in enumerate(strides):
            ci, ri = get_centers_and_radii(crb, brb, s)
            ref = base[i].clone()
            draw_heatmap_batched(ref, ci, ri, 6.0, 0.9, clear=clear)
            assert torch.equal(fused[i], ref), f"box maps {seed}/{case} stride {s} shape {shapes[i]}"
        lanes_n, pts = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        lanes = (torch.rand(b, lanes_n, pts, 2, generator=g) * torch.tensor([sw, sh])).to(DEV)
        num_points = torch.from_numpy(rng.integers(0, pts + 1, size=(b, lanes_n))).to(DEV)
        num_lanes = torch.from_numpy(rng.integers(0, lanes_n + 1, size=b)).to(DEV)
        q, radius = int(rng.choice([1, 17, 64, 128])), int(rng.integers(0, 4))
        fused = [t.clone() for t in base]
        draw_polylines_multiscale(fused, lanes, q, radius, strides, 6.0, 0.9, num_points=num_points, num_lanes=num_lanes, clear=clear)
        for i, s in enumerate(strides):
            ref = base[i].clone()
            _via_targets(ref, lanes, q, radius, s, 6.0, 0.9, num_points=num_points, num_lanes=num_lanes, clear=clear)    # three launches
            assert torch.equal(fused[i], ref), f"lane raster {seed}/{case} stride {s} shape {shapes[i]} q {q} r {radius}"


@pytest.mark.parametrize("seed", _seeds(8))
def test_lane_raster_paths_agree_on_random_shapes(seed):
    """the three ways to the lane maps — the one-launch kernel (tile waves sample the polylines), sampler + point splat, and the
    step operator draw_targets_multiscale (sampler riding in the box-map launch) — bit for bit on random shapes: 1..64 points,
    1..16 polylines per frame, 1..400 samples, ragged counts, non-finite vertices, clear and in place"""
    from accvlab import _amd_native as nat
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale, draw_targets_multiscale, lanes as lanes_mod

    rng = np.random.default_rng(424200 + seed)
    seen = set()
    for case in range(6):
        b = int(rng.integers(1, 4))
        sw, sh = float(rng.choice([512, 1024, 2048, 4000])), float(rng.choice([256, 512, 1080]))
        n_scales = int(rng.integers(1, 4))
        strides = [float(x) for x in rng.choice([1.0, 2.0, 4.0, 8.0, 16.0], size=n_scales, replace=False)]
        strides = [s_ for s_ in strides if sw / s_ * sh / s_ * 4 * b < 3e8] or [8.0]
        clear = bool(rng.integers(0, 2))
        nl, npnt = int(rng.choice([1, 2, 3, 5, 8, 16])), int(rng.choice([1, 2, 3, 8, 17, 24, 33, 64]))
        q, radius = int(rng.choice([1, 2, 17, 64, 100, 128, 256, 400])), int(rng.integers(0, 6))
        start = rng.uniform([0, 0], [sw, sh], size=(b, nl, 1, 2))
        steps = rng.normal(0, 1, size=(b, nl, npnt, 2)) * [sw / npnt / 2, sh / npnt / 2] + [sw / npnt / 3, -sh / npnt / 4]
        pts = (start + np.cumsum(steps, axis=2)).astype(np.float32)
        for _ in range(int(rng.integers(0, 3))):      # a few non-finite / huge vertices
            pts[rng.integers(0, b), rng.integers(0, nl), rng.integers(0, npnt), rng.integers(0, 2)] = \
                rng.choice([np.nan, np.inf, -np.inf, 3.0e7, -1.0e12])
        pts_d = torch.from_numpy(pts).to(DEV)
        ragged = bool(rng.integers(0, 2))
        npts = torch.from_numpy(rng.integers(0, npnt + 1, size=(b, nl)).astype(rng.choice([np.int32, np.int64]))).to(DEV) if ragged else None
        nlanes = torch.from_numpy(rng.integers(0, nl + 1, size=(b,)).astype(rng.choice([np.int32, np.int64]))).to(DEV) if ragged else None
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        shapes = [(b, max(1, int(sh / s_)) , max(4, int(sw / s_) // 4 * 4)) for s_ in strides]
        base = [(torch.rand(s_, generator=g) * 0.3).to(DEV) for s_ in shapes]
        kw = dict(num_points=npts, num_lanes=nlanes, clear=clear)
        one = [t.clone() for t in base]
        draw_polylines_multiscale(one, pts_d, q, radius, strides, 6.0, 0.9, **kw)
        seen.add(nat.last_dispatch().split("<")[0])
        two = [t.clone() for t in base]
        lanes_mod.FUSED_SAMPLER = False
        try:
            draw_polylines_multiscale(two, pts_d, q, radius, strides, 6.0, 0.9, **kw)
        finally:
            lanes_mod.FUSED_SAMPLER = True
        cs = [torch.rand(int(rng.integers(0, 12)), 2, generator=g) * torch.tensor([sw, sh]) for _ in range(b)]
        crb = combine_data(cs, device=DEV)
        brb = combine_data([torch.cat([c - 20.0, c + 30.0], 1) for c in cs], device=DEV, other_with_same_sample_sizes=crb)
        box_a, lane_a = [t.clone() for t in base], [t.clone() for t in base]
        draw_targets_multiscale(box_a, crb, brb, strides, lane_a, pts_d, q, radius, None, 6.0, 0.9, **kw)
        box_b = [t.clone() for t in base]
        draw_heatmap_multiscale(box_b, crb, brb, strides, 6.0, 0.9, clear=clear)
        what = f"{seed}/{case}: b {b} lanes {nl} x {npnt} points, {q} samples, r {radius}, strides {strides}, clear {clear}, ragged {ragged}"
        for i in range(len(strides)):
            assert torch.equal(one[i], two[i]), f"default path vs sampler + point splat, scale {i}, {what}"
            assert torch.equal(lane_a[i], two[i]), f"draw_targets_multiscale lane map, scale {i}, {what}"
            assert torch.equal(box_a[i], box_b[i]), f"draw_targets_multiscale box map, scale {i}, {what}"


@pytest.mark.parametrize("seed", _seeds(6))
def test_h2_random_gradients_against_torch_autograd(seed):
    """backward of gather (repeated indices accumulate), inverse, write and mapping against the same expressions written with torch
    advanced indexing on the GPU (float64: the atomics' summation order is then irrelevant at 1e-10)"""
    from accvlab.batching_helpers import RaggedBatch, batched_index_mapping, batched_indexing_access, batched_indexing_write, \
        batched_inverse_indexing_access

    rng = np.random.default_rng(9985 + seed)
    for case in range(5):
        b, n_src, n_tgt = int(rng.integers(1, 5)), int(rng.choice([1, 6, 70])), int(rng.choice([3, 40, 90]))
        n_idx = int(rng.integers(0, min(n_tgt, 30) + 1))
        inner = tuple(int(x) for x in rng.choice([1, 2, 5], size=int(rng.integers(0, 3))))
        counts = rng.integers(0, n_idx + 1, size=b)
        valid = torch.from_numpy(np.arange(n_idx)[None, :] < counts[:, None]).to(DEV)
        src_idx = torch.from_numpy(rng.integers(0, n_src, size=(b, n_idx))).to(DEV)                       # repeats allowed
        tgt_idx = torch.from_numpy(np.stack([rng.permutation(n_tgt)[:n_idx] for _ in range(b)]).reshape(b, n_idx)).to(DEV)
        cnt = torch.from_numpy(counts).to(DEV)
        rows = torch.arange(b, device=DEV).unsqueeze(1).expand(b, n_idx)

        def leaf(shape):
            return torch.from_numpy(rng.standard_normal(shape)).to(DEV).requires_grad_(True)

        # gather
        data, w = leaf((b, n_src) + inner), torch.from_numpy(rng.standard_normal((b, n_idx) + inner)).to(DEV)
        out = batched_indexing_access(data, RaggedBatch(src_idx, sample_sizes=cnt), 0.0)
        (out.tensor * w).sum().backward()
        ref = data.detach().clone().requires_grad_(True)
        m = valid.reshape(valid.shape + (1,) * len(inner))
        ((ref[rows, src_idx] * m) * w).sum().backward()
        assert torch.allclose(data.grad, ref.grad, rtol=1e-10, atol=1e-10), f"gather grad {seed}/{case}"
        # inverse and write: scatter of `vals` into a fresh / an existing tensor
        vals, into = leaf((b, n_idx) + inner), leaf((b, n_tgt) + inner)
        w2 = torch.from_numpy(rng.standard_normal((b, n_tgt) + inner)).to(DEV)
        tgt_rb = RaggedBatch(tgt_idx, sample_sizes=cnt)
        o1 = batched_inverse_indexing_access(RaggedBatch(vals, sample_sizes=cnt), tgt_rb, n_tgt, 0.0)
        o2 = batched_indexing_write(RaggedBatch(vals, sample_sizes=cnt), tgt_rb, into)
        ((o1 + 2.0 * o2) * w2).sum().backward()
        v_ref, i_ref = vals.detach().clone().requires_grad_(True), into.detach().clone().requires_grad_(True)
        r1 = torch.zeros_like(i_ref).index_put((rows[valid], tgt_idx[valid]), v_ref[valid])
        r2 = i_ref.index_put((rows[valid], tgt_idx[valid]), v_ref[valid])
        ((r1 + 2.0 * r2) * w2).sum().backward()
        assert torch.allclose(vals.grad, v_ref.grad, rtol=1e-10, atol=1e-10), f"scatter source grad {seed}/{case}"
        assert torch.allclose(into.grad, i_ref.grad, rtol=1e-10, atol=1e-10), f"write destination grad {seed}/{case}"
        # mapping: out[i, tgt] = src[i, src_idx] on a copy of `into2`
        src, into2 = leaf((b, n_src) + inner), leaf((b, n_tgt) + inner)
        o3 = batched_index_mapping(src, RaggedBatch(src_idx, sample_sizes=cnt), tgt_rb, into2)
        (o3 * w2).sum().backward()
        s_ref, i2_ref = src.detach().clone().requires_grad_(True), into2.detach().clone().requires_grad_(True)
        r3 = i2_ref.index_put((rows[valid], tgt_idx[valid]), s_ref[rows[valid], src_idx[valid]])
        (r3 * w2).sum().backward()
        assert torch.allclose(src.grad, s_ref.grad, rtol=1e-10, atol=1e-10), f"mapping source grad {seed}/{case}"
        assert torch.allclose(into2.grad, i2_ref.grad, rtol=1e-10, atol=1e-10), f"mapping destination grad {seed}/{case}"


@pytest.mark.parametrize("seed", _seeds(6))
def test_h2_multi_batch_dims_and_moved_index_dim_equal_the_flat_case(seed):
    """several batch dimensions and an index dimension that is not the first data dimension: gather, write and boolean
    compaction must equal the same call on the flattened batch with the indexed dimension moved to the front"""
    from accvlab.batching_helpers import RaggedBatch, batched_bool_indexing, batched_indexing_access, batched_indexing_write

    rng = np.random.default_rng(9960 + seed)
    for case in range(5):
        bshape = tuple(int(x) for x in rng.integers(1, 4, size=int(rng.integers(2, 4))))
        nb, total = len(bshape), int(np.prod(bshape))
        n_src, n_idx = int(rng.choice([2, 9, 40])), int(rng.integers(0, 12))
        extra = int(rng.integers(1, 4))            # one data dimension in front of the indexed one
        tail = tuple(int(x) for x in rng.integers(1, 4, size=int(rng.integers(0, 2))))
        data = torch.from_numpy(rng.standard_normal(bshape + (extra, n_src) + tail).astype(np.float32)).to(DEV)
        idx = torch.from_numpy(rng.integers(0, n_src, size=bshape + (n_idx,))).to(DEV)
        cnt = torch.from_numpy(rng.integers(0, n_idx + 1, size=bshape)).to(DEV)
        dim = nb + 1                               # the indexed dimension of `data`
        got = batched_indexing_access(data, RaggedBatch(idx, sample_sizes=cnt), 2.0, dim_to_index_in=dim)
        flat_data = data.reshape((total, extra, n_src) + tail).transpose(1, 2).contiguous()
        flat = batched_indexing_access(flat_data, RaggedBatch(idx.reshape(total, n_idx), sample_sizes=cnt.reshape(total)), 2.0)
        want = flat.tensor.transpose(1, 2).reshape(bshape + (extra, n_idx) + tail)
        assert tuple(got.batch_shape) == bshape and got.non_uniform_dim == dim and torch.equal(got.tensor, want), f"gather {seed}/{case}"
        # write (unique targets)
        tgt = torch.from_numpy(np.stack([rng.permutation(n_src)[: min(n_idx, n_src)] for _ in range(total)])
                               .reshape(bshape + (min(n_idx, n_src),))).to(DEV)
        k = tgt.shape[-1]
        cnt2 = torch.from_numpy(rng.integers(0, k + 1, size=bshape)).to(DEV)
        vals = torch.from_numpy(rng.standard_normal(bshape + (extra, k) + tail).astype(np.float32)).to(DEV)
        got = batched_indexing_write(RaggedBatch(vals, sample_sizes=cnt2, non_uniform_dim=dim), RaggedBatch(tgt, sample_sizes=cnt2), data,
                                     dim_to_index_in=dim)
        flat_vals = vals.reshape((total, extra, k) + tail).transpose(1, 2).contiguous()
        flat = batched_indexing_write(RaggedBatch(flat_vals, sample_sizes=cnt2.reshape(total)),
                                      RaggedBatch(tgt.reshape(total, k), sample_sizes=cnt2.reshape(total)), flat_data)
        want = flat.transpose(1, 2).reshape(data.shape)
        assert torch.equal(got, want), f"write {seed}/{case}"
        # boolean compaction of a ragged batch with several batch dimensions
        sizes = torch.from_numpy(rng.integers(0, n_src + 1, size=bshape)).to(DEV)
        rdata = RaggedBatch(data.transpose(nb, nb + 1).contiguous(), sample_sizes=sizes)            # non-uniform dim right behind the batch
        mask = torch.from_numpy(rng.random(bshape + (n_src,)) < 0.5).to(DEV)
        got = batched_bool_indexing(rdata, RaggedBatch(mask, sample_sizes=sizes))
        flat = batched_bool_indexing(RaggedBatch(rdata.tensor.reshape((total, n_src, extra) + tail), sample_sizes=sizes.reshape(total)),
                                     RaggedBatch(mask.reshape(total, n_src), sample_sizes=sizes.reshape(total)))
        assert tuple(got.batch_shape) == bshape and torch.equal(got.sample_sizes.reshape(total), flat.sample_sizes)
        assert torch.equal(got.tensor.reshape((total,) + tuple(got.tensor.shape[nb:])), flat.tensor), f"bool indexing {seed}/{case}"


@pytest.mark.parametrize("seed", _seeds(4))
def test_h1_extreme_coordinates_and_radii(seed):
    """centres anywhere in int32 and radii up to 2^30 - 1 (beyond that 2r+1 overflows int32 in the reference as well,
    cuh:61-62): the 32-bit cull must stay conservative, the exact box must be computed in 64 bits, nothing may be written outside
    the map, and the values must match the oracle"""
    from accvlab.draw_heatmap import draw_heatmap_batched

    rng = np.random.default_rng(9940 + seed)
    big = [0, 1, -1, 2 ** 31 - 1, -2 ** 31, 2 ** 30, -2 ** 30, 2 ** 29 + 3, -2 ** 29 - 3, 65536, -65536]
    for case in range(6):
        h, w, b, n = int(rng.choice([5, 40, 130])), int(rng.choice([8, 64, 260])), int(rng.integers(1, 4)), 12
        cx = np.where(rng.random((b, n)) < 0.5, rng.choice(big, size=(b, n)), rng.integers(-5, w + 5, size=(b, n)))
        cy = np.where(rng.random((b, n)) < 0.5, rng.choice(big, size=(b, n)), rng.integers(-5, h + 5, size=(b, n)))
        r = np.where(rng.random((b, n)) < 0.5, rng.choice([2 ** 30 - 1, 2 ** 29, 2 ** 24 + 1, 70000], size=(b, n)), rng.integers(0, 9, size=(b, n)))
        centers = np.stack([cx, cy], -1).astype(np.int32)
        radii = r.astype(np.int32)
        counts = rng.integers(0, n + 1, size=b)
        clear = bool(rng.integers(0, 2))
        base = (rng.random((b, h, w), dtype=np.float32) * 0.2).astype(np.float32)
        want = base.copy()
        oracle.draw_heatmap_batched(want, centers, radii, counts, clear=clear)
        buf = torch.full((b * h * w + 512,), -3.0, device=DEV)
        got = buf[256:256 + b * h * w].view(b, h, w)
        got.copy_(torch.from_numpy(base))
        n_t = torch.from_numpy(counts).to(DEV)
        from accvlab import _amd_native as nat
        from accvlab.draw_heatmap import ops
        ops._FORCED_FLAGS = [0, nat.HM_SMALL_RADII, nat.HM_TILE_ROWS_16, nat.HM_WRITE_THROUGH][(seed + case) % 4]
        try:
            draw_heatmap_batched(got, rb(torch.from_numpy(centers).to(DEV), n_t), rb(torch.from_numpy(radii).to(DEV), n_t), clear=clear)
        finally:
            ops._FORCED_FLAGS = 0
        assert bool((buf[:256] == -3.0).all()) and bool((buf[256 + b * h * w:] == -3.0).all()), f"guard band {seed}/{case}"
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - want.astype(np.float64)).max())
        assert err <= 1e-5, f"extreme {seed}/{case}: max abs err {err}"


@pytest.mark.parametrize("seed", _seeds(4))
def test_target_front_end_random_against_float32_numpy(seed):
    """get_centers_and_radii: c = int(c / stride), r = max(1, ceil(min edge distance / stride)) in IEEE float32
    (packages/draw_heatmap/tests/_test_helpers.py:20-28) — bit-exact against numpy float32 for random boxes incl. degenerate and
    inverted ones, centres outside their box, large coordinates, odd strides"""
    from accvlab.draw_heatmap import get_centers_and_radii

    rng = np.random.default_rng(9920 + seed)
    for case in range(6):
        shape = tuple(int(x) for x in rng.integers(1, 40, size=int(rng.integers(1, 3))))
        scale = float(rng.choice([1.0, 100.0, 5000.0, 1e6]))
        c = ((rng.random(shape + (2,)) - 0.2) * scale).astype(np.float32)
        half = ((rng.random(shape + (4,)) - 0.1) * scale * 0.3).astype(np.float32)
        boxes = np.concatenate([c - half[..., :2], c + half[..., 2:]], -1).astype(np.float32)
        stride = np.float32(rng.choice([1.0, 2.0, 3.0, 4.0, 7.5, 16.0, 0.37]))
        ci, ri = get_centers_and_radii(torch.from_numpy(c).to(DEV), torch.from_numpy(boxes).to(DEV), float(stride))
        m = np.minimum(np.minimum(c[..., 0] - boxes[..., 0], c[..., 1] - boxes[..., 1]),
                       np.minimum(boxes[..., 2] - c[..., 0], boxes[..., 3] - c[..., 1])).astype(np.float32)
        want_r = np.maximum(1, np.ceil((m / stride).astype(np.float32))).astype(np.int64)
        want_c = np.trunc((c / stride).astype(np.float32)).astype(np.int64)
        assert ci.dtype == torch.int32 and ri.dtype == torch.int32
        assert np.array_equal(ci.cpu().numpy().astype(np.int64), want_c), f"centres {seed}/{case}"
        assert np.array_equal(ri.cpu().numpy().astype(np.int64), want_r), f"radii {seed}/{case}"


def test_degenerate_extents_do_not_trip_any_operator():
    """zero batches, zero-sized maps, zero objects / indices / points: every public operator returns (correctly shaped) results
    instead of tripping over an empty tensor's null data pointer or an ambiguous reshape"""
    from accvlab.batching_helpers import (RaggedBatch, batched_bool_indexing, batched_index_mapping, batched_indexing_access,
                                          batched_indexing_write, batched_inverse_indexing_access, combine_data, get_compact_lists,
                                          get_indices_from_mask, get_mask_from_indices, matched_pair_loss_sum)
    from accvlab.draw_heatmap import (draw_heatmap, draw_heatmap_batched, draw_heatmap_multiscale, draw_polylines_batched,
                                      draw_polylines_multiscale, get_centers_and_radii)
    from accvlab.lane_helpers.polyline import interpolate, lengths
    from accvlab.multi_tensor_copier import start_copy

    i32, i64, f32 = torch.int32, torch.int64, torch.float32

    def z(*shape, dtype=f32):
        return torch.zeros(shape, dtype=dtype, device=DEV)

    for b, h, w, n in ((0, 8, 8, 3), (2, 0, 8, 3), (2, 8, 0, 3), (2, 8, 8, 0), (0, 0, 0, 0)):
        for clear in (True, False):
            draw_heatmap_batched(z(b, h, w), rb(z(b, n, 2, dtype=i32), z(b, dtype=i64)), rb(z(b, n, dtype=i32), z(b, dtype=i64)), clear=clear)
            draw_heatmap_batched(z(b, 3, h, w), rb(z(b, n, 2, dtype=i32), z(b, dtype=i64)), rb(z(b, n, dtype=i32), z(b, dtype=i64)),
                                 labels=rb(z(b, n, dtype=i32), z(b, dtype=i64)), clear=clear)
            draw_heatmap(z(b, h, w), z(n, 2, dtype=i32), z(n, dtype=i32), z(n, dtype=i32), clear=clear)
            crb = RaggedBatch(z(b, n, 2), sample_sizes=z(b, dtype=i64))
            draw_heatmap_multiscale([z(b, h, w), z(b, h // 2, w // 2)], crb, z(b, n, 4), (1.0, 2.0), clear=clear)
            draw_polylines_batched(z(b, h, w), z(b, 2, n, 2), 4, 1, 1.0, clear=clear)
            draw_polylines_multiscale([z(b, h, w)], z(b, 0, 3, 2), 4, 1, (1.0,), clear=clear)
    c, r = get_centers_and_radii(z(0, 2), z(0, 4), 4.0)
    assert c.shape == (0, 2) and r.shape == (0,)
    assert interpolate(z(0, 5, 2), z(0, 3)).shape == (0, 3, 2) and lengths(z(0, 5, 2)).shape == (0,)
    assert interpolate(z(2, 5, 2), z(2, 0)).shape == (2, 0, 2)
    assert bool(torch.isnan(interpolate(z(2, 0, 2), z(2, 3))).all())
    for b, n_src, n_idx in ((0, 4, 3), (2, 4, 0), (2, 0, 0), (0, 0, 0)):
        idx = RaggedBatch(z(b, n_idx, dtype=i64), sample_sizes=z(b, dtype=i64))
        out = batched_indexing_access(z(b, n_src, 3), idx, 1.0)
        assert tuple(out.tensor.shape) == (b, n_idx, 3)
        assert tuple(batched_inverse_indexing_access(RaggedBatch(z(b, n_idx, 3), sample_sizes=z(b, dtype=i64)), idx, 5, 1.0).shape) == (b, 5, 3)
        assert tuple(batched_indexing_write(RaggedBatch(z(b, n_idx, 3), sample_sizes=z(b, dtype=i64)), idx, z(b, 5, 3)).shape) == (b, 5, 3)
        assert tuple(batched_index_mapping(z(b, n_src, 3), idx, idx, z(b, 5, 3)).shape) == (b, 5, 3)
        assert tuple(get_mask_from_indices(5, idx).shape) == (b, 5)
        assert tuple(matched_pair_loss_sum(z(b, n_src, 3), z(b, n_src, 3), idx, idx).shape) == (b,)
        mask = torch.zeros((b, n_src), dtype=torch.bool, device=DEV)
        comp = batched_bool_indexing(z(b, n_src, 3), mask)
        assert tuple(comp.tensor.shape) == (b, 0, 3) and tuple(get_indices_from_mask(mask).tensor.shape) == (b, 0)
        assert tuple(get_compact_lists(mask, [z(b, n_src)])[0].tensor.shape) == (b, 0)
    empty = combine_data([torch.zeros(0, 4), torch.zeros(0, 4)], device=DEV)
    assert tuple(empty.tensor.shape) == (2, 0, 4) and [tuple(t.shape) for t in empty.split()] == [(0, 4), (0, 4)]
    for tree in ([], {}, (), [torch.zeros(0)], {"a": [torch.zeros(0, 3), ()], "b": None}):
        got = start_copy(tree, DEV).get()
        assert type(got) is type(tree) and len(got) == len(tree)
        assert start_copy(got, "cpu", use_background_thread=False).get() is not None


@pytest.mark.parametrize("background", [True, False])
def test_copier_many_outstanding_handles_in_random_order(background):
    """40 copies in flight at once (pinned arena blocks, side-stream DMAs and — in background mode — tickets of the native
    orchestrator all outstanding), results fetched in random order, some handles dropped without get(): every result is exact and
    later copies still work"""
    import gc

    import test_fuzz_cpu as m

    from accvlab.multi_tensor_copier import start_copy

    rng = np.random.default_rng(777 + int(background))
    trees = [[m._random_tree(rng) for _ in range(int(rng.integers(1, 8)))] + [torch.randn(int(rng.integers(1, 5000)))] for _ in range(40)]
    handles = [start_copy(t, DEV, use_background_thread=background, max_packed_chunk_bytes=int(rng.choice([512, 8192, 32 << 20])))
               for t in trees]
    order = rng.permutation(len(handles))
    dropped = set(int(i) for i in order[:6])
    for i in order:
        if int(i) in dropped:
            handles[int(i)] = None            # the destructor must wait for the transfer and release the staging
            continue
        h = handles[int(i)]
        while not h.ready():
            pass
        assert m._same_tree(trees[int(i)], h.get()), f"job {i}"
    gc.collect()
    again = start_copy(trees[0], DEV, use_background_thread=background).get()
    assert m._same_tree(trees[0], again)
    # and back to the host with everything in flight at once
    gpu_trees = [start_copy(t, DEV).get() for t in trees[:15]]
    back = [start_copy(g, "cpu", use_background_thread=background) for g in gpu_trees]
    for t, h in zip(trees[:15], back):
        assert m._same_tree(t, h.get())


def test_operators_on_two_streams_and_two_threads():
    """the library keeps no per-call global state: draws and ragged gathers issued from two python threads, each on its own
    stream, give the single-threaded results"""
    import threading

    import bench_workloads as wl
    from accvlab.batching_helpers import RaggedBatch, batched_indexing_access, combine_data
    from accvlab.draw_heatmap import draw_heatmap_batched

    cl, rl = wl.heatmap_objects(6, 270, 480, 1, 40, "A", seed=5)
    c = combine_data(cl, device=DEV)
    r = combine_data(rl, device=DEV, other_with_same_sample_sizes=c)
    want = torch.empty(6, 270, 480, device=DEV)
    draw_heatmap_batched(want, c, r, clear=True)
    data = torch.randn(8, 300, 16, device=DEV)
    idx = RaggedBatch(torch.randint(0, 300, (8, 50), device=DEV), sample_sizes=torch.randint(0, 51, (8,), device=DEV))
    want_g = batched_indexing_access(data, idx, 0.5).tensor.clone()
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(40):
                    hm = torch.empty(6, 270, 480, device=DEV)
                    draw_heatmap_batched(hm, c, r, clear=True)
                    g = batched_indexing_access(data, idx, 0.5).tensor
                    if not (torch.equal(hm, want) and torch.equal(g, want_g)):
                        errors.append(f"thread {k}: mismatch")
                        return
            s.synchronize()
        except Exception as exc:      # noqa: BLE001
            errors.append(f"thread {k}: {exc!r}")

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_whole_target_prep_step_replays_from_one_hip_graph():
    """box maps + lane raster (all strides) + ragged gather / write / mask + fused matched loss + pad fill captured into ONE
    hipGraph and replayed after the inputs were overwritten in place: every output equals the eager result for the new inputs
    (nothing in the hot path synchronises, allocates outside the stream-ordered allocator or reads host state at launch time)"""
    from accvlab.batching_helpers import (RaggedBatch, batched_index_mapping, batched_indexing_access, batched_indexing_write,
                                          get_mask_from_indices, matched_pair_loss_sum)
    from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale

    g = torch.Generator().manual_seed(3)
    b, n, sw, sh = 4, 16, 640, 384
    strides = (4.0, 8.0, 16.0)

    def inputs(seed):
        gg = torch.Generator().manual_seed(seed)
        c = torch.rand(b, n, 2, generator=gg) * torch.tensor([sw, sh])
        half = torch.rand(b, n, 4, generator=gg) * 60
        boxes = torch.cat([c - half[..., :2], c + half[..., 2:]], -1)
        cnt = torch.randint(0, n + 1, (b,), generator=gg)
        lanes = torch.rand(b, 3, 10, 2, generator=gg) * torch.tensor([sw, sh])
        data = torch.randn(b, 50, 8, generator=gg)
        idx = torch.randint(0, 50, (b, 12), generator=gg)
        tgt = torch.stack([torch.randperm(50, generator=gg)[:12] for _ in range(b)])
        icnt = torch.randint(0, 13, (b,), generator=gg)
        return [t.to(DEV) for t in (c, boxes, cnt, lanes, data, idx, tgt, icnt)]

    static = inputs(1)

    def step(c, boxes, cnt, lanes, data, idx, tgt, icnt):
        box_maps = [torch.empty(b, int(sh / s), int(sw / s), device=DEV) for s in strides]
        lane_maps = [torch.empty_like(m_) for m_ in box_maps]
        draw_heatmap_multiscale(box_maps, RaggedBatch(c, sample_sizes=cnt), boxes, strides, clear=True)
        draw_polylines_multiscale(lane_maps, lanes, 64, 2, strides, clear=True)
        irb, trb = RaggedBatch(idx, sample_sizes=icnt), RaggedBatch(tgt, sample_sizes=icnt)
        gathered = batched_indexing_access(data, irb, 0.0)
        written = batched_indexing_write(gathered, trb, data)
        mapped = batched_index_mapping(data, irb, trb, data)
        mask = get_mask_from_indices(50, trb)
        loss = matched_pair_loss_sum(data, written, irb, trb, kind="smooth_l1")
        padded = gathered.with_padded_set_to(-1.0).tensor
        return box_maps + lane_maps + [gathered.tensor, written, mapped, mask, loss, padded]

    for _ in range(2):                                   # warm-up on a side stream, as torch's capture recipe asks
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(*static)
        torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step(*static)
    for seed in (2, 3):
        fresh = inputs(seed)
        for dst, src in zip(static, fresh):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        want = step(*[t.clone() for t in fresh])
        for k, (a, w) in enumerate(zip(outs, want)):
            assert torch.equal(a, w), f"replay with inputs {seed}: output {k} differs from the eager result"


@pytest.mark.parametrize("seed", _seeds(6))
def test_combine_data_to_the_gpu_random(seed):
    """combine_data with a GPU target: CPU samples (pinned padded pack or flat + pack kernel, depending on the size), GPU samples,
    nested lists with and without flattening, shared sample sizes, gradients — equal to the CPU result of the same call"""
    from accvlab.batching_helpers import combine_data

    rng = np.random.default_rng(9910 + seed)
    dts = [torch.float32, torch.float64, torch.int64, torch.int32, torch.float16, torch.uint8, torch.bool]
    for case in range(6):
        outer, inner_b = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        feat = tuple(int(x) for x in rng.integers(1, 6, size=int(rng.integers(0, 3))))
        dt = dts[(seed + case) % len(dts)]
        big = bool(rng.integers(0, 4) == 0)                      # > 1 MB: the flat + pack-kernel route
        max_n = 3000 if big else 6

        def sample():
            n = int(rng.integers(0, max_n))
            a = rng.random((n,) + feat) * 60
            return torch.from_numpy(a > 30) if dt == torch.bool else torch.from_numpy(a).to(dt)

        nested = [[sample() for _ in range(inner_b)] for _ in range(outer)]
        for flatten in (True, False):
            if not flatten and all(t.shape[0] == 0 for row in nested for t in row):
                continue      # the reference's own corner: float32 [*batch, 0] on the CPU unless `device` is given (:527-535)
            want = combine_data(nested, flatten_batch_dims=flatten)
            from_cpu = combine_data(nested, device=DEV, flatten_batch_dims=flatten)
            from_gpu = combine_data([[t.to(DEV) for t in row] for row in nested], flatten_batch_dims=flatten)
            for got in (from_cpu, from_gpu):
                assert got.tensor.is_cuda and got.tensor.dtype == dt and tuple(got.batch_shape) == tuple(want.batch_shape)
                assert torch.equal(got.sample_sizes.cpu(), want.sample_sizes) and torch.equal(got.tensor.cpu(), want.tensor)
                assert torch.equal(got.mask.cpu(), want.mask)
            shared = combine_data(nested, device=DEV, other_with_same_sample_sizes=from_cpu, flatten_batch_dims=flatten)
            assert shared.sample_sizes is from_cpu.sample_sizes and torch.equal(shared.tensor, from_cpu.tensor)
        if dt in (torch.float32, torch.float64) and any(t.shape[0] for row in nested for t in row):
            leaves = [t.clone().to(DEV).requires_grad_(True) for row in nested for t in row]
            rb_ = combine_data(leaves)
            w = torch.rand(rb_.tensor.shape, device=DEV, dtype=dt)
            (rb_.tensor * w).sum().backward()
            for i, leaf in enumerate(leaves):
                if leaf.shape[0] == 0:      # the reference skips empty samples (`if size_elem > 0`, :421-423): no gradient at all
                    assert leaf.grad is None
                else:
                    assert torch.equal(leaf.grad, w[i, : leaf.shape[0]]), f"grad {seed}/{case}/{i}"
