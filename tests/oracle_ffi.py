"""ctypes loaders for the parity checkers (TEST INFRASTRUCTURE).

  oracle()  -> oracle/liboracle.so   : our scalar CPU restatement (always available once built)
  reflib()  -> oracle/_ref/libalfalfa_ref.so : the REAL reference search/DP code compiled from
               /root/reference/src in place (only in the build container; None elsewhere)

Nothing in the product package imports this module.
"""
import ctypes as C
import os

import numpy as np

from alfalfa_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libalfalfa_ref.so")
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "alfalfa")

_oracle = None
_ref = None


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("kmr_reads", "child_reads", "lcp_reads", "sa_reads", "s_bytes", "p_bytes", "matches", "queries",
                 "cells_extend", "cells_nw", "cells_global", "cells_dpband")]


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = C.CDLL(ORACLE_SO)
        _oracle.orc_seed.restype = C.c_int64
        _oracle.orc_sa_bytes.restype = C.c_int64
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def reflib():
    global _ref
    if _ref is None:
        if not have_ref():
            return None
        _ref = C.CDLL(REF_SO)
        for f in ("ref_sa_new", "ref_sa_S", "ref_sa_SA", "ref_sa_LCPvec", "ref_sa_CHILD", "ref_sa_ISA", "ref_sa_KMR"):
            getattr(_ref, f).restype = C.c_void_p
        for f in ("ref_sa_N", "ref_sa_K", "ref_sa_logN", "ref_sa_NKm1", "ref_sa_n", "ref_sa_LCP_M_size", "ref_sa_CHILD_size",
                  "ref_sa_ISA_size", "ref_sa_KMR_size", "ref_sa_nseq", "ref_sa_startpos", "ref_seed"):
            getattr(_ref, f).restype = C.c_long
        _ref.ref_sa_descr.restype = C.c_char_p
    return _ref


# ---------------------------------------------------------------- oracle wrappers
def orc_counters_reset():
    oracle().orc_counters_reset()


def orc_counters():
    c = Counters()
    oracle().orc_counters_get(C.byref(c))
    d = {n: getattr(c, n) for n, _ in Counters._fields_}
    d["sa_bytes"] = oracle().orc_sa_bytes(C.byref(c))
    return d


def orc_seed_batch(view, mem_type, queries):
    def fn(v, t, q, nq, m, o):
        return oracle().orc_seed_batch(C.byref(v), t, q, nq, m, o)
    return ffi._seed_call(fn, view, mem_type, queries, oracle().orc_free)


def orc_ksw_extend_batch(params, tasks):
    n = len(tasks)
    out = np.zeros(n, dtype=ffi.EXTEND_DTYPE)
    oracle().orc_ksw_extend_batch(C.byref(params), tasks, n, out.ctypes.data_as(C.c_void_p))
    return out


def orc_ksw_nw_batch(params, tasks):
    n = len(tasks)
    out = np.zeros(n, dtype=np.int32)
    oracle().orc_ksw_nw_batch(C.byref(params), tasks, n, out.ctypes.data_as(C.c_void_p))
    return out


def orc_ksw_global_batch(params, tasks):
    return ffi._global_call(oracle().orc_ksw_global_batch, params, tasks, oracle().orc_free)


# ---------------------------------------------------------------- reference wrappers
class RefIndex:
    """The reference's own sparseSA object (via oracle/ref_shim.cpp)."""

    def __init__(self):
        self.lib = reflib()
        self.h = C.c_void_p(self.lib.ref_sa_new())

    @classmethod
    def build_fasta(cls, path, K=12, suflink=False, child=True, kmer=True):
        r = cls()
        r.lib.ref_sa_build(r.h, path.encode(), C.c_long(K), int(suflink), int(child), int(kmer))
        return r

    @classmethod
    def build_mem(cls, names, seqs, K=12, suflink=False, child=True, kmer=True):
        r = cls()
        n = len(seqs)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        sq = (C.c_char_p * n)(*[s if isinstance(s, bytes) else s.encode() for s in seqs])
        r.lib.ref_sa_build_mem(r.h, n, nm, sq, C.c_long(K), int(suflink), int(child), int(kmer))
        return r

    def save(self, prefix):
        self.lib.ref_sa_save(self.h, prefix.encode())

    def arrays(self):
        L, h = self.lib, self.h
        n = L.ref_sa_n(h)
        N = L.ref_sa_N(h)

        def arr(ptr, cnt, dt):
            if not ptr or cnt == 0:
                return np.zeros(0, dtype=dt)
            nbytes = cnt * np.dtype(dt).itemsize
            return np.frombuffer(C.string_at(ptr, nbytes), dtype=dt).copy()
        nM = L.ref_sa_LCP_M_size(h)
        midx = np.zeros(nM, dtype=np.int64)
        mval = np.zeros(nM, dtype=np.int32)
        if nM:
            L.ref_sa_LCP_M(h, midx.ctypes.data_as(C.c_void_p), mval.ctypes.data_as(C.c_void_p))
        nseq = L.ref_sa_nseq(h)
        return dict(
            N=N, K=L.ref_sa_K(h), logN=L.ref_sa_logN(h), NKm1=L.ref_sa_NKm1(h), n=n,
            S=arr(L.ref_sa_S(h), N, np.uint8), SA=arr(L.ref_sa_SA(h), n, np.uint32),
            LCP=arr(L.ref_sa_LCPvec(h), n, np.uint8), LCP_M_idx=midx, LCP_M_val=mval,
            CHILD=arr(L.ref_sa_CHILD(h), L.ref_sa_CHILD_size(h), np.int32),
            ISA=arr(L.ref_sa_ISA(h), L.ref_sa_ISA_size(h), np.int32),
            KMR=arr(L.ref_sa_KMR(h), 2 * L.ref_sa_KMR_size(h), np.uint32),
            startpos=np.array([L.ref_sa_startpos(h, C.c_long(i)) for i in range(nseq)], dtype=np.int64),
            descr=[L.ref_sa_descr(h, C.c_long(i)).decode() for i in range(nseq)],
        )

    def seed(self, mem_type, P, min_len, max_count, sparse_q, max_smem_cont=0):
        out = C.c_void_p()
        n = self.lib.ref_seed(self.h, int(mem_type), P, len(P), int(min_len), int(max_count), int(sparse_q), int(max_smem_cont),
                              C.byref(out))
        m = np.frombuffer(C.string_at(out.value, n * 16), dtype=ffi.MATCH_DTYPE).copy() if n else np.zeros(0, ffi.MATCH_DTYPE)
        self.lib.ref_free(out)
        return m

    def __del__(self):
        try:
            self.lib.ref_sa_free(self.h)
        except Exception:
            pass


def _mat_ptr(params):
    return C.cast(C.byref(params, ffi.DpParams.mat.offset), C.c_void_p)


def ref_ksw_extend(params, q, t, w, h0):
    L = reflib()
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    o = [C.c_int() for _ in range(5)]
    s = L.ksw_extend(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), params.m, _mat_ptr(params),
                     params.gapo, params.gape, int(w), params.zdrop, int(h0), *[C.byref(x) for x in o])
    return (s,) + tuple(x.value for x in o)


def ref_ksw_nw(params, q, t, w):
    L = reflib()
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    return L.ksw_nw(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), params.m, _mat_ptr(params),
                    params.gapo, params.gape, int(w))


def ref_ksw_global(params, q, t, w):
    L = reflib()
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nc = C.c_int()
    cg = C.c_void_p()
    s = L.ksw_global(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), params.m, _mat_ptr(params),
                     params.gapo, params.gape, int(w), C.byref(nc), C.byref(cg))
    cig = np.frombuffer(C.string_at(cg.value, nc.value * 4), dtype=np.uint32).copy() if nc.value else np.zeros(0, np.uint32)
    if cg.value:
        L.ref_free(cg)
    return s, cig
