"""DataLoader hook of the multi-tensor copier: pack_batch / packing_collate / PackedBatch.

CPU part: the packed buffer has exactly the byte layout of the copier's pack plan (the offsets the reference's
compute_pack_plan produces for its own test leaves, SURVEY §8c / multi_tensor_copier.cpp:419-590), unpack() restores
the structure bit-exactly, the batch survives pickling and a real multi-process DataLoader.
GPU part: start_copy(PackedBatch) == start_copy(structure), pinned and pageable, through the C-ABI.
"""
import pickle

import numpy as np
import pytest
import torch

from accvlab.multi_tensor_copier import PackedBatch, pack_batch, packing_collate, start_copy


def _same(a, b, dev=None):
    if isinstance(b, np.ndarray):                  # numpy leaves come back as tensors (as in start_copy)
        b = torch.from_numpy(b)
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(a)
    assert type(a) is type(b), (type(a), type(b))
    if isinstance(a, torch.Tensor):
        assert a.dtype == b.dtype and a.shape == b.shape
        if dev is not None:
            assert a.device == dev, (a.device, dev)
        raw = lambda t: t.cpu().contiguous().reshape(-1).view(torch.uint8)  # noqa: E731  bit-exact, any dtype
        assert torch.equal(raw(a), raw(b))
    elif isinstance(a, dict):
        assert list(a.keys()) == list(b.keys())
        for k in a:
            _same(a[k], b[k], dev)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y, dev)
    else:
        assert a is b or a == b


def _reference_test_leaves():
    # the five packable leaves of the reference's own copier test (test_multi_tensor_copier.py), in traversal order
    g = torch.Generator().manual_seed(0)
    return [torch.randn(32, generator=g),                                   # f32 128 B
            torch.arange(17, dtype=torch.int64),                            # i64 136 B
            torch.randn(11, generator=g).to(torch.float16),                 # f16 22 B
            torch.view_as_complex(torch.randn(9, 2, generator=g)),          # c64 72 B
            torch.view_as_complex(torch.randn(5, 2, generator=g, dtype=torch.float64))]   # c128 80 B


@pytest.mark.parametrize("min_align,offsets,total", [(16, [0, 128, 272, 304, 384], 464),
                                                     (1, [288, 80, 416, 216, 0], 438),
                                                     (6, [80, 208, 420, 344, 0], 442)])
def test_buffer_layout_is_the_pack_plan(min_align, offsets, total):
    leaves = _reference_test_leaves()
    pk = pack_batch(leaves, min_packed_alignment_bytes=min_align)
    assert pk.num_tensors == 5 and pk.num_packed == 5
    assert pk.buffer.numel() == total and pk.buffer.dtype == torch.uint8
    assert pk._offsets.tolist() == offsets
    for t, off in zip(leaves, offsets):
        nb = t.numel() * t.element_size()
        assert torch.equal(pk.buffer[off:off + nb], t.contiguous().view(-1).view(torch.uint8))
    _same(pk.unpack(), leaves)


def _mixed_batch(seed=0, n=40):
    g = torch.Generator().manual_seed(seed)
    marker = object()
    samples = []
    for i in range(n):
        k = int(torch.randint(1, 20, (1,), generator=g))
        samples.append({
            "boxes": torch.randn(k, 4, generator=g),
            "labels": torch.randint(0, 10, (k,), generator=g),
            "meta": {"id": i, "name": f"frame{i}", "tag": marker, "scale": (1.5, torch.tensor(2.0))},
            "np": np.arange(k, dtype=np.float64) * 0.5,
            "flags": (torch.zeros(k, dtype=torch.bool), [torch.full((3,), i, dtype=torch.int16)]),
        })
    extra = {
        "big": torch.randn(300, 300, generator=g),                 # > 256 KiB: rides along unpacked
        "strided": torch.randn(8, 8, generator=g)[:, ::2],         # non-contiguous: unpacked
        "empty": torch.empty(0, 4),                                # 0 bytes: unpacked
        "none": None,
    }
    return [samples, extra]


def test_roundtrip_mixed_structure_and_pickle():
    data = _mixed_batch()
    pk = pack_batch(data)
    assert isinstance(pk, PackedBatch)
    assert pk.num_packed == 40 * 6 and pk.num_tensors == 40 * 6 + 3
    out = pk.unpack()
    _same(out, [data[0], {**data[1]}])
    assert out[0][0]["meta"]["tag"] is data[0][0]["meta"]["tag"]           # passthrough by identity
    assert out[1]["strided"].stride() == data[1]["strided"].stride()
    # packed leaves alias the buffer (zero copy)
    lo, hi = pk.buffer.data_ptr(), pk.buffer.data_ptr() + pk.buffer.numel()
    assert lo <= out[0][0]["boxes"].data_ptr() < hi
    # pickling keeps everything except object identity of passthrough leaves
    pk2 = pickle.loads(pickle.dumps(pk))
    out2 = pk2.unpack()
    assert torch.equal(pk2.buffer, pk.buffer)
    assert torch.equal(out2[0][7]["labels"], data[0][7]["labels"]) and out2[0][7]["meta"]["name"] == "frame7"


@pytest.mark.parametrize("data", [
    [],                                            # nothing
    [torch.arange(5)],                             # a single packable leaf: the planner packs only >= 2
    {"a": "text", "b": 3},                         # no tensors
    torch.arange(6).reshape(2, 3),                 # bare tensor
    (torch.ones(2), torch.ones(3, dtype=torch.float64)),
])
def test_degenerate_inputs(data):
    pk = pack_batch(data)
    out = pk.unpack()
    _same(out, data)
    _same(start_copy(pk, "cpu").get(), data)


class _ManySmall(torch.utils.data.Dataset):
    def __len__(self):
        return 12

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        return {"idx": i, "parts": [torch.randn(int(torch.randint(1, 9, (1,), generator=g)), 3, generator=g)
                                    for _ in range(25)], "v": torch.full((4,), float(i))}


@pytest.mark.parametrize("workers", [0, 2])
def test_through_a_dataloader(workers):
    ds = _ManySmall()
    loader = torch.utils.data.DataLoader(ds, batch_size=4, num_workers=workers, collate_fn=packing_collate())
    seen = 0
    for b, pk in enumerate(loader):
        assert isinstance(pk, PackedBatch) and pk.num_packed == 4 * 26
        out = start_copy(pk, "cpu").get()
        assert [s["idx"] for s in out] == list(range(4 * b, 4 * b + 4))
        for s in out:
            _same(s, ds[s["idx"]])
        seen += len(out)
    assert seen == 12


def test_inner_collate_fn_runs_first():
    ds = _ManySmall()
    inner = lambda samples: {"v": torch.stack([s["v"] for s in samples]), "parts": [s["parts"] for s in samples]}  # noqa: E731
    pk = packing_collate(inner)([ds[0], ds[1]])
    out = pk.unpack()
    assert out["v"].shape == (2, 4) and len(out["parts"]) == 2 and len(out["parts"][1]) == 25


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("pinned_buffer", [False, True])
@pytest.mark.parametrize("use_pinned_staging", [True, False])
@pytest.mark.parametrize("background", [True, False])
def test_start_copy_of_a_packed_batch_equals_plain_start_copy(pinned_buffer, use_pinned_staging, background):
    dev = torch.device("cuda", 0)
    data = _mixed_batch(seed=3)
    pk = pack_batch(data)
    if pinned_buffer:
        pk = pk.pin_memory()
        assert pk.is_pinned()
    got = start_copy(pk, dev, use_pinned_staging=use_pinned_staging, use_background_thread=background).get()
    want = start_copy(data, dev).get()
    _same(got, want, None)
    _same(got, [data[0], {**data[1]}])
    assert got[0][0]["boxes"].device == dev and got[1]["big"].device == dev
    # packed outputs share ONE device storage
    ptrs = {got[0][i]["boxes"].untyped_storage().data_ptr() for i in range(40)}
    assert len(ptrs) == 1
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_packed_batch_larger_than_one_staging_slice_and_alignment():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    data = [torch.randn(60_000, generator=g) for _ in range(40)]            # 240 KB each -> 9.6 MB buffer
    pk = pack_batch(data, min_packed_alignment_bytes=64)
    assert pk.alignment == 64 and pk.buffer.numel() > 8 * (1 << 20)
    got = start_copy(pk, dev).get()
    for a, b in zip(got, data):
        assert a.data_ptr() % 64 == 0 and torch.equal(a.cpu(), b)


@pytest.mark.gpu
def test_dataloader_pin_memory_to_gpu():
    dev = torch.device("cuda", 0)
    ds = _ManySmall()
    loader = torch.utils.data.DataLoader(ds, batch_size=6, num_workers=2, pin_memory=True, collate_fn=packing_collate())
    n = 0
    for pk in loader:
        assert pk.is_pinned()                      # DataLoader called PackedBatch.pin_memory()
        out = start_copy(pk, dev).get()
        for s in out:
            _same(s, ds[s["idx"]])
            assert s["v"].device == dev
        n += len(out)
    assert n == 12
