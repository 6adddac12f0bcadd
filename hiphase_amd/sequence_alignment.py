"""Host-side mirror of reference src/sequence_alignment.rs:7-38 (`edit_distance`) and of
`Variant::closest_allele_clip` (reference src/data_types/variants.rs:624-641) over hp_edit_distance_batch."""
import ctypes as C

import numpy as np

from . import _ffi


def edit_distance_batch(pairs, device_id=0):
    """pairs: list of (bytes, bytes) -> list[int] Levenshtein distances (unit costs)."""
    dll = _ffi.lib()
    n = len(pairs)
    keep = []
    arr = (_ffi.EdPair * max(n, 1))()
    for i, (a, b) in enumerate(pairs):
        na = np.frombuffer(bytes(a), np.uint8) if len(a) else np.zeros(1, np.uint8)
        nb = np.frombuffer(bytes(b), np.uint8) if len(b) else np.zeros(1, np.uint8)
        keep += [na, nb]
        arr[i].a = na.ctypes.data_as(C.POINTER(C.c_uint8))
        arr[i].b = nb.ctypes.data_as(C.POINTER(C.c_uint8))
        arr[i].a_len, arr[i].b_len = len(a), len(b)
    out = np.zeros(max(n, 1), np.uint64)
    _ffi.check(dll.hp_edit_distance_batch(arr, n, out.ctypes.data_as(C.POINTER(C.c_uint64)), device_id))
    return [int(x) for x in out[:n]]


def edit_distance(v1, v2, device_id=0):
    return edit_distance_batch([(v1, v2)], device_id)[0]


def closest_allele_clip(allele, allele0, allele1, head_clip=0, tail_clip=0, device_id=0):
    """variants.rs:624-641 -> (AlleleType, min distance, other distance)."""
    a0 = allele0[head_clip:len(allele0) - tail_clip]
    a1 = allele1[head_clip:len(allele1) - tail_clip]
    d0, d1 = edit_distance_batch([(allele, a0), (allele, a1)], device_id)
    if d0 < d1:
        return 0, d0, d1
    if d0 > d1:
        return 1, d1, d0
    return 2, d0, d1
