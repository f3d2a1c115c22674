"""alfa_ksw_global_noalloc (the entry point that allocates nothing) vs alfa_ksw_global_batch AND vs the oracle (a regression that
hit both entry points alike would otherwise pass), through the C ABI, on the GPU.

Two calls in a row on the same device array, the second with a smaller input; bit-exact scores and CIGARs.
The caller's device buffers come from the HIP runtime through ctypes (tests/hip_rt.py).
"""
import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import dpgen
import hip_rt

pytestmark = pytest.mark.gpu


def _run_noalloc(P, pairs, ws, cig_stride, max_tlen):
    n = len(pairs)
    tasks = np.zeros(n, dtype=ffi.DEV_TASK_DTYPE)
    chunks, off = [], 0
    for i, (q, t) in enumerate(pairs):
        tasks[i] = (off, off + len(q), len(q), len(t), ws[i], 0)
        chunks += [np.asarray(q, np.uint8), np.asarray(t, np.uint8)]
        off += len(q) + len(t)
    seq = np.concatenate(chunks + [np.zeros(64, np.uint8)])
    d_tasks = hip_rt.DevArray(tasks.view(np.uint8))
    d_seq = hip_rt.DevArray(seq)
    d_score = hip_rt.DevArray(np.full(n, 12345, np.int32))
    d_ncig = hip_rt.DevArray(np.full(n, 777, np.int32))
    d_cig = hip_rt.DevArray(nbytes=4 * n * cig_stride, fill=0)
    hip_rt.sync()
    ffi.ksw_global_noalloc(P, d_tasks.ptr, n, max_tlen, d_seq.ptr, d_score.ptr, d_cig.ptr, d_ncig.ptr, cig_stride, 0, 0)   # NULL stream
    hip_rt.sync()
    return d_score.get(np.int32), d_cig.get(np.uint32, (n, cig_stride)), d_ncig.get(np.int32)


def _check(P, pairs, ws):
    max_tlen = max(len(t) for _, t in pairs)
    cig_stride = max(len(q) + len(t) for q, t in pairs) + 2
    score, cig, ncig = _run_noalloc(P, pairs, ws, cig_stride, max_tlen)
    tasks, keep = ffi.make_dp_tasks(pairs, ws, 0)
    s0, c0, n0, o0 = ffi.ksw_global_batch(P, tasks)
    assert np.array_equal(score, s0)
    assert np.array_equal(ncig, n0)
    so, co, no, oo = O.orc_ksw_global_batch(P, tasks)
    assert np.array_equal(score, so)
    for i in range(len(pairs)):
        if n0[i] > 0:
            assert np.array_equal(cig[i, cig_stride - n0[i]:], c0[o0[i]:o0[i] + n0[i]]), i
            assert ncig[i] == no[i] and np.array_equal(cig[i, cig_stride - n0[i]:], co[oo[i]:oo[i] + no[i]]), i
        else:
            assert len(pairs[i][1]) > len(pairs[i][0]) + ws[i]        # only where the reference indexes outside its matrix
    return int((n0 > 0).sum())


def test_noalloc_matches_allocating_entry_point_twice(gpu):
    P = ffi.default_params()
    # first call: mapper-like long-read gaps and windows; second call: fewer, smaller tasks on the same scratch array
    pairs, ws, _ = dpgen.mapper_like_tasks(11, 500, 700, 0.05, 60)
    keep = [i for i, (q, t) in enumerate(pairs) if len(q) >= 1 and 1 <= len(t) <= 300 and 2 * min(ws[i], max(len(q), len(t))) + 1 <= 256]
    keep = keep[:(ffi.NOALLOC_SCRATCH_BYTES - 4096) // (64 + 256 * 300)]      # the documented limit of one call
    pairs1, ws1 = [pairs[i] for i in keep], [ws[i] for i in keep]
    assert _check(P, pairs1, ws1) > 50
    pairs, ws, _ = dpgen.random_tasks(3, 300)
    keep = [i for i, (q, t) in enumerate(pairs) if 1 <= len(q) and 1 <= len(t) <= 120 and 2 * min(ws[i], max(len(q), len(t))) + 1 <= 256]
    pairs2, ws2 = [pairs[i] for i in keep][:150], [ws[i] for i in keep][:150]
    assert len(pairs2) < len(pairs1)
    assert _check(P, pairs2, ws2) > 20


def test_noalloc_rejects_oversize_and_flags_bad_tasks(gpu):
    P = ffi.default_params()
    with pytest.raises(ffi.AlfaError):
        # 4096 tasks x 4096 rows x 256 B of direction bytes is far beyond the fixed array
        ffi.ksw_global_noalloc(P, 1, 4096, 4096, 1, 1, 1, 1, 10000, 0, 0)
    assert "scratch" in ffi.lib().alfa_last_error().decode()
    rng = np.random.default_rng(0)
    q = rng.integers(0, 4, 40).astype(np.uint8)
    pairs = [(q, q.copy()), (q, q[:10].copy()), (q, q.copy())]
    ws = [5, 400, 5]                      # task 1: tlen 10 < ... fine band clamps to 40; make task 2 exceed max_tlen instead
    score, cig, ncig = _run_noalloc(P, pairs, ws, 100, 39)      # tasks 0 and 2 have tlen 40 > max_tlen 39 -> flagged
    assert ncig[0] == -2 and ncig[2] == -2 and score[0] == np.int32(-2**31)
    assert ncig[1] != -2
