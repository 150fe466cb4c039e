"""Output recycling of the copier's host path, driven directly on CPU storage (no GPU): `_mtc_host.Tree.make_packed_views` +
`rebuild` on a "chunk" that is an ordinary CPU byte tensor.  The same scenarios as tests/test_copier_gpu.py, so that the reference-
count logic (tensor objects re-pointed in place, whole result trees handed out again) also runs under the CPU sanitizers
(scripts/sanitize_cpu.sh)."""
import gc

import numpy as np
import pytest
import torch

from accvlab.multi_tensor_copier import copier

host = copier._host
pytestmark = pytest.mark.skipif(host is None or not hasattr(host, "recycled_tree_count"), reason="host extension not built")


def _copy(data):
    """what start_copy(...).get() does with a fully packable structure, with a CPU chunk standing in for the device buffer"""
    t = host.Tree(data)
    n = t.num_leaves()
    leaves = [t.leaf(i) for i in range(n)]
    offs, total = [], 0
    for leaf in leaves:
        offs.append(total)
        total += (leaf.numel() * leaf.element_size() + 15) // 16 * 16
    chunk = torch.zeros(total + 16, dtype=torch.uint8)
    for leaf, o in zip(leaves, offs):
        nb = leaf.numel() * leaf.element_size()
        chunk[o:o + nb] = leaf.contiguous().view(-1).view(torch.uint8)
    del leaves
    t.make_packed_views(np.arange(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.asarray(offs, dtype=np.int64), [chunk],
                        np.zeros(1, dtype=np.int64))
    return t.rebuild()


TAG = object()


def _batch(k, n=5):
    return {"gt": [torch.full((2 + i, 3), float(10 * k + i)) for i in range(n)],
            "ids": (torch.full((4,), 100 * k, dtype=torch.int64), torch.full((2, 2), k, dtype=torch.int32)), "tag": TAG,
            "nested": [{"a": torch.full((2,), float(k))}, [torch.full((1,), float(-k))]]}


def _check(res, k, n=5):
    assert res["tag"] is TAG and len(res["gt"]) == n
    for i in range(n):
        assert torch.equal(res["gt"][i], torch.full((2 + i, 3), float(10 * k + i)))
    assert torch.equal(res["ids"][0], torch.full((4,), 100 * k, dtype=torch.int64))
    assert torch.equal(res["ids"][1], torch.full((2, 2), k, dtype=torch.int32))
    assert torch.equal(res["nested"][0]["a"], torch.full((2,), float(k)))
    assert torch.equal(res["nested"][1][0], torch.full((1,), float(-k)))


@pytest.fixture(autouse=True)
def _clean_pool():
    host.release_recycled_outputs()
    yield
    host.release_recycled_outputs()
    host.set_output_recycling(True)


def test_loop_hands_the_tree_of_two_steps_ago_out_again():
    ids, res = [], None
    for k in range(7):
        res = _copy(_batch(k))
        _check(res, k)
        ids.append(id(res))
    assert ids[4] == ids[2] == ids[6] and ids[5] == ids[3] and ids[4] != ids[5]
    assert host.recycled_tree_count() == 2 and host.recycled_output_count() > 0
    del res
    gc.collect()
    _check(_copy(_batch(9)), 9)


def test_discarded_results_are_reused_at_once():
    ids = []
    for k in range(4):
        r = _copy(_batch(k))
        _check(r, k)
        ids.append(id(r))
        del r
    assert len(set(ids[1:])) <= 2 and ids[3] in ids[:3]


def test_held_trees_containers_and_leaves_are_never_touched():
    kept = [_copy(_batch(k)) for k in range(10, 14)]
    assert len({id(r) for r in kept}) == 4
    for r, k in zip(kept, range(10, 14)):
        _check(r, k)
    r = _copy(_batch(20))
    inner, leaf = r["nested"], r["gt"][3]
    del r
    more = [_copy(_batch(k)) for k in range(21, 25)]
    assert torch.equal(inner[0]["a"], torch.full((2,), 20.0)) and torch.equal(leaf, torch.full((5, 3), 203.0))
    for m, k in zip(more, range(21, 25)):
        _check(m, k)
        assert m["nested"] is not inner and all(t is not leaf for t in m["gt"])
    view = more[0]["gt"][1][1:]                    # a view keeps its base tensor (and only that one) out of the pool
    want = view.clone()
    del more
    for k in range(30, 34):
        _check(_copy(_batch(k)), k)
    assert torch.equal(view, want)


@pytest.mark.parametrize("change", ["replace", "append", "attribute", "requires_grad", "dict entry"])
def test_a_tree_the_caller_changed_is_not_handed_out_again(change):
    a = _copy(_batch(30))
    if change == "replace":
        a["gt"][0] = torch.zeros(1)
    elif change == "append":
        a["gt"].append(None)
    elif change == "attribute":
        a["gt"][1].note = "mine"
    elif change == "requires_grad":
        a["gt"][2].requires_grad_()
    else:
        a["extra"] = 1
    del a
    b = _copy(_batch(31))
    c = _copy(_batch(32))
    d = _copy(_batch(33))
    for r, k in ((b, 31), (c, 32), (d, 33)):
        _check(r, k)
        assert "extra" not in r and not hasattr(r["gt"][1], "note") and not r["gt"][2].requires_grad


def test_other_structures_and_pass_through_objects_get_fresh_trees():
    x = _copy(_batch(40)); del x
    y = _copy(_batch(41)); del y
    other = _batch(42)
    other["tag"] = object()
    z = _copy(other)
    assert z["tag"] is other["tag"] and z["tag"] is not TAG
    del z
    w = _copy(_batch(43, n=3))
    _check(w, 43, n=3)
    del w
    _check(_copy(_batch(44)), 44)
    changed_dtype = _batch(45)
    changed_dtype["ids"] = (changed_dtype["ids"][0].to(torch.int32), changed_dtype["ids"][1])
    r = _copy(changed_dtype)
    assert r["ids"][0].dtype == torch.int32 and torch.equal(r["ids"][0], torch.full((4,), 4500, dtype=torch.int32))
    del r
    _check(_copy(_batch(46)), 46)


def test_switches():
    r = _copy(_batch(1)); del r
    assert host.recycled_tree_count() == 1
    host.set_output_recycling(False)
    assert host.recycled_tree_count() == 0 and host.recycled_output_count() == 0
    p, q = _copy(_batch(2)), _copy(_batch(3))
    _check(p, 2); _check(q, 3)
    assert host.recycled_tree_count() == 0
    host.set_output_recycling(True)
    del p, q
    r = _copy(_batch(4)); del r
    assert host.recycled_tree_count() == 1
    host.release_recycled_outputs()
    assert host.recycled_tree_count() == 0 and host.recycled_output_count() == 0


def test_metadata_that_changes_every_step_rides_in_the_reused_tree():
    """ids, names, time stamps: pass-through leaves differ every step and dict keys built per step are equal, not identical — the
    kept tree is still handed out again, with this step's objects in its pass-through slots; a pass-through element of a TUPLE
    cannot be replaced, so such a tree is rebuilt"""
    def batch(k, in_tuple=False):
        meta = {"frame " + str(1): 1000 + k, "name": f"sample_{k}", "stamp": 0.5 * k, "flags": [k, None, ("x",)]}
        aux = (torch.full((3,), float(k)), 1000 + k) if in_tuple else [torch.full((3,), float(k)), 1000 + k]
        return {"gt": [torch.full((2, 2), float(k)), torch.full((1,), float(-k))], "meta": meta, "aux": aux}

    def check(res, k, src):
        assert torch.equal(res["gt"][0], torch.full((2, 2), float(k))) and torch.equal(res["gt"][1], torch.full((1,), float(-k)))
        assert torch.equal(res["aux"][0], torch.full((3,), float(k))) and res["aux"][1] == 1000 + k
        assert res["meta"]["frame 1"] == 1000 + k and res["meta"]["name"] is src["meta"]["name"] and res["meta"]["stamp"] == 0.5 * k
        assert res["meta"]["flags"][0] == k and res["meta"]["flags"][1] is None and res["meta"]["flags"][2] == ("x",)
        assert list(res["meta"]) == ["frame 1", "name", "stamp", "flags"]

    ids, res = [], None
    for k in range(300, 307):          # (integers above 256 and fresh strings: new objects every step)
        src = batch(k)
        res = _copy(src)
        check(res, k, src)
        ids.append(id(res))
    assert ids[4] == ids[2] == ids[6] and ids[5] == ids[3], ids
    del res
    ids = []
    for k in range(400, 406):
        src = batch(k, in_tuple=True)
        res = _copy(src)
        check(res, k, src)
        assert isinstance(res["aux"], tuple)
        ids.append(id(res))
        del res
    assert host.recycled_output_count() > 0
