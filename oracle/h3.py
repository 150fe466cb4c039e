"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path under accv-lab_amd/).

Pure-python restatement of the reference's pack planner for small host tensors
(packages/multi_tensor_copier/accvlab/multi_tensor_copier/csrc/multi_tensor_copier.cpp):
  :481-507  make_pack_candidate   (host tensor, GPU target, contiguous, 0 < bytes <= 256 KiB;
                                   required_align = round_up(max(min_align, elem), elem))
  :419-433  pack_bucket_key       (required_align rounded DOWN to 16/8/4/2/1)
  :513-549  layout_packed_offsets (buckets 16 -> 1, insertion order inside, offset = round_up(cursor, align),
                                   new chunk when offset + bytes > max_chunk and the chunk is not empty)
  :553-590  compute_pack_plan     (enabled only if >= 2 tensors were packed)

Parity status: PINNED to the offsets derived from those lines for the five packable leaves of the reference's own
test (tests/test_multi_tensor_copier.py:180-230: f32 128 B, i64 136 B, f16 22 B, c64 72 B, c128 80 B) — SURVEY.md §8c;
the reference test itself only asserts `byte_off % required_align == 0`, which tests/ also check.
"""
from __future__ import annotations

PACK_MAX = 256 * 1024


def round_up(x: int, a: int) -> int:
    if a <= 1:
        return x
    r = x % a
    return x if r == 0 else x + (a - r)


def required_align(min_align: int, elem: int) -> int:
    return round_up(max(max(1, min_align), elem), elem)


def plan(nbytes, elem_sizes, candidate, min_align=16, max_chunk=32 * 1024 * 1024):
    """-> (offset[i] or -1, chunk[i] or -1, chunk_sizes)"""
    n = len(nbytes)
    off, chk = [-1] * n, [-1] * n
    buckets = {16: [], 8: [], 4: [], 2: [], 1: []}
    for i in range(n):
        if not candidate[i]:
            continue
        ra = required_align(min_align, elem_sizes[i])
        key = 16 if ra >= 16 else 8 if ra >= 8 else 4 if ra >= 4 else 2 if ra >= 2 else 1
        buckets[key].append((i, ra))
    cursor, chunk, packed, sizes = 0, 0, 0, []
    for key in (16, 8, 4, 2, 1):
        for i, ra in buckets[key]:
            at = round_up(cursor, ra)
            if at + nbytes[i] > max_chunk and cursor > 0:
                sizes.append(cursor)
                cursor, chunk = 0, chunk + 1
                at = round_up(cursor, ra)
            off[i], chk[i] = at, chunk
            cursor = at + nbytes[i]
            packed += 1
    if cursor > 0:
        sizes.append(cursor)
    if packed < 2 or not sizes:
        return [-1] * n, [-1] * n, []
    return off, chk, sizes
