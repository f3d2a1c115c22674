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


class _Kswr(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]


def orc_ksw_align_batch(params, tasks):
    n = len(tasks)
    out = np.zeros(n, dtype=ffi.KSWR_DTYPE)
    oracle().orc_ksw_align_batch(C.byref(params), tasks, n, out.ctypes.data_as(C.c_void_p))
    return out


def ref_ksw_align(params, q, t, xtra):
    """the REAL ksw_align (src/ksw.c:330), qry = NULL; gapo/gape passed as params holds them"""
    L = reflib()
    L.ksw_align.restype = _Kswr
    q = np.array(q, dtype=np.uint8)          # copies: the reference reverses its arguments in place
    t = np.array(t, dtype=np.uint8)
    r = L.ksw_align(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), params.m, _mat_ptr(params),
                    params.gapo, params.gape, int(xtra), None)
    return (r.score, r.te, r.qe, r.score2, r.te2, r.tb, r.qb)


# ---------------------------------------------------------------- dynProg (dp.cpp)
def orc_dp_band_batch(variant, scores, tasks):
    L = oracle()

    def chk(rc):
        if rc:
            raise ffi.AlfaError("oracle dp_band error %d" % rc)
    old, ffi_check = ffi.check, None
    try:
        ffi.check = chk
        return ffi.dpband_call(L.orc_dp_band_batch, variant, scores, tasks, L.orc_free)
    finally:
        ffi.check = old


def ref_dynprog(variant, scores, spec):
    """one call of the REAL reference class (dynProg, via oracle/ref_shim.cpp); same tuple as ffi.dpband_call.
    The reference calls exit(1) on a failed traceback: run it in a forked child when that can happen."""
    L = reflib()
    r, q = spec["ref"], spec["query"]
    f = [int(bool(x)) for x in spec.get("flags", (0, 0, 0, 0, 0))]
    out = (C.c_long * 8)()
    cap = len(r) + len(q) + 8
    cc = C.create_string_buffer(cap)
    cl = (C.c_int * cap)()
    L.ref_dynprog(int(variant), int(scores.affine), scores.match, scores.mismatch, scores.openGap, scores.extendGap,
                  r, C.c_long(len(r)), q, C.c_long(len(q)), C.c_long(spec.get("refB", 0)), C.c_long(spec.get("refE", len(r) - 1)),
                  spec.get("queryB", 0), spec.get("queryE", len(q) - 1), f[0], f[1], f[2], f[3], f[4],
                  spec.get("out_type", 4), spec.get("band1", 10), spec.get("band2", 10), out, cc, cl, cap)
    n = int(out[7])
    cig = "".join("%d%s" % (cl[i], chr(cc.raw[i])) for i in range(n))
    return (int(out[0]), int(out[1]), int(out[2]), int(out[3]), int(out[4]), int(out[5]), int(out[6]), cig)


def ref_dynprog_forked(variant, scores, spec):
    """ref_dynprog in a child process; returns the tuple, or ("exit", status) when the reference ended the process"""
    import pickle
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(rd)
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 2)
            res = ref_dynprog(variant, scores, spec)
            os.write(wr, pickle.dumps(res))
        finally:
            os._exit(0)
    os.close(wr)
    data = b""
    while True:
        chunk = os.read(rd, 65536)
        if not chunk:
            break
        data += chunk
    os.close(rd)
    _, status = os.waitpid(pid, 0)
    if not data:
        return ("exit", status)
    return pickle.loads(data)
