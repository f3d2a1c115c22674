"""ctypes binding of libalfa_hot (include/alfa_hot.h, include/alfa_index.h).

This is the thin Python face used by tests/, bench.py and __graft_entry__.py.  It is
plumbing only: every compute call goes through the C ABI into the HIP kernels.  If the
shared library has not been built the import of the library fails loudly -- there is
no Python or CPU fallback.
"""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libalfa_hot.so")

SEED_SMAM, SEED_MEM, SEED_SMEM, SEED_SMEMP = 0, 1, 2, 3


class IndexView(C.Structure):
    _fields_ = [
        ("S", C.c_void_p), ("N", C.c_int64), ("K", C.c_int64), ("logN", C.c_int64), ("NKm1", C.c_int64),
        ("SA", C.c_void_p), ("n", C.c_int64),
        ("LCP", C.c_void_p), ("LCP_M_idx", C.c_void_p), ("LCP_M_val", C.c_void_p), ("nM", C.c_int64),
        ("CHILD", C.c_void_p), ("ISA", C.c_void_p), ("KMR", C.c_void_p),
        ("has_child", C.c_int32), ("has_suflink", C.c_int32), ("has_kmer", C.c_int32),
        ("n_seq", C.c_int32), ("startpos", C.c_void_p),
    ]


class Match(C.Structure):
    _fields_ = [("ref", C.c_int64), ("query", C.c_int32), ("len", C.c_int32)]


MATCH_DTYPE = np.dtype([("ref", "<i8"), ("query", "<i4"), ("len", "<i4")])


REGION_DTYPE = np.dtype([("ref_begin", "<i8"), ("ref_end", "<i8"), ("score", "<i4"), ("first", "<i4"), ("count", "<i4"), ("chain_count", "<i4"),
                         ("chain_anchor", "<i4"), ("chain_edit", "<i4"), ("chain_coverage", "<i4"), ("reserved", "<i4")])


class RegionParams(C.Structure):
    _fields_ = [("max_seed_candidates", C.c_int32), ("max_region_mems", C.c_int32), ("min_mem_reg_length", C.c_int32),
                ("reserved", C.c_int32), ("error_percent", C.c_double)]


class SeedQuery(C.Structure):
    _fields_ = [("P", C.c_char_p), ("plen", C.c_int32), ("min_len", C.c_int32), ("max_count", C.c_int32),
                ("sparse_q", C.c_int32), ("max_smem_cont", C.c_int32)]


class DpTask(C.Structure):
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("qlen", C.c_int32), ("tlen", C.c_int32),
                ("w", C.c_int32), ("h0", C.c_int32)]


class DpParams(C.Structure):
    _fields_ = [("m", C.c_int32), ("mat", C.c_int8 * 64), ("gapo", C.c_int32), ("gape", C.c_int32), ("zdrop", C.c_int32)]


class ExtendResult(C.Structure):
    _fields_ = [("score", C.c_int32), ("qle", C.c_int32), ("tle", C.c_int32), ("gtle", C.c_int32),
                ("gscore", C.c_int32), ("max_off", C.c_int32)]


EXTEND_DTYPE = np.dtype([("score", "<i4"), ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"), ("gscore", "<i4"), ("max_off", "<i4")])


class AlignTask(C.Structure):       # alfa_align_task
    _fields_ = [("query", C.c_void_p), ("target", C.c_void_p), ("qlen", C.c_int32), ("tlen", C.c_int32), ("xtra", C.c_int32)]


KSWR_DTYPE = np.dtype([(n, "<i4") for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")])   # alfa_kswr = kswr_t
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


class DpbandTask(C.Structure):      # alfa_dpband_task
    _fields_ = [("ref", C.c_char_p), ("ref_len", C.c_int64), ("query", C.c_char_p), ("query_len", C.c_int64),
                ("refB", C.c_int64), ("refE", C.c_int64), ("queryB", C.c_int32), ("queryE", C.c_int32),
                ("freeRefB", C.c_uint8), ("freeRefE", C.c_uint8), ("freeQueryB", C.c_uint8), ("freeQueryE", C.c_uint8),
                ("local", C.c_uint8), ("out_type", C.c_int32), ("band1", C.c_int32), ("band2", C.c_int32)]


class DpbandScores(C.Structure):    # alfa_dpband_scores
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("openGap", C.c_int32), ("extendGap", C.c_int32),
                ("affine", C.c_int32)]


class DpbandResult(C.Structure):    # alfa_dpband_result
    _fields_ = [("ret", C.c_int32), ("dpScore", C.c_int32), ("editDist", C.c_int32), ("refB", C.c_int64), ("refE", C.c_int64),
                ("queryB", C.c_int32), ("queryE", C.c_int32), ("n_cigar", C.c_int32), ("cig_off", C.c_int64)]


DPBAND_STATIC, DPBAND_FULL, DPBAND_SCORE, DPBAND_BASIC = 0, 1, 2, 3


class Timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("work_units", C.c_int64), ("launches", C.c_int32)]


class AlfaError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libalfa_hot.so (built by __graft_entry__.build()).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AlfaError("libalfa_hot.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`. "
                            "There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.alfa_last_error.restype = C.c_char_p
        _lib.alfa_host_index_view.restype = C.POINTER(IndexView)
        _lib.alfa_host_index_descr.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise AlfaError("libalfa_hot error %d: %s" % (rc, lib().alfa_last_error().decode()))


def default_params(match=1, mismatch=-4, open_gap=-6, extend_gap=-1, zdrop=100):
    """align_opt::fill_m (reference src/options.cpp:33-41) + the sign convention of the ksw callers."""
    p = DpParams()
    p.m = 5
    k = 0
    for i in range(4):
        for j in range(4):
            p.mat[k] = match if i == j else mismatch
            k += 1
        p.mat[k] = -1
        k += 1
    for j in range(5):
        p.mat[k] = -1
        k += 1
    p.gapo = -open_gap
    p.gape = -extend_gap
    p.zdrop = zdrop
    return p


def view_arrays(v):
    """numpy views (zero-copy) of the arrays behind an IndexView."""
    def arr(ptr, n, dt):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt)
    n = v.n
    return dict(
        S=arr(v.S, v.N, np.uint8), SA=arr(v.SA, n, np.uint32), LCP=arr(v.LCP, n, np.uint8),
        LCP_M_idx=arr(v.LCP_M_idx, v.nM, np.int64), LCP_M_val=arr(v.LCP_M_val, v.nM, np.int32),
        CHILD=arr(v.CHILD, n if v.has_child else 0, np.int32), ISA=arr(v.ISA, n if v.has_suflink else 0, np.int32),
        KMR=arr(v.KMR, 2 * (1 << 20) if v.has_kmer else 0, np.uint32), startpos=arr(v.startpos, v.n_seq, np.int64),
    )


class HostIndex:
    """Host-resident enhanced sparse suffix array (alfa_index.h)."""

    def __init__(self, handle):
        self._h = handle
        self.view = lib().alfa_host_index_view(handle).contents

    @classmethod
    def build_fasta(cls, path, K=12, suflink=False, child=True, kmer=True):
        h = C.c_void_p()
        check(lib().alfa_host_index_build_fasta(path.encode(), C.c_int64(K), int(suflink), int(child), int(kmer), C.byref(h)))
        return cls(h)

    @classmethod
    def build_mem(cls, names, seqs, K=12, suflink=False, child=True, kmer=True):
        n = len(seqs)
        bs = [s if isinstance(s, bytes) else s.encode() for s in seqs]
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        sq = (C.c_char_p * n)(*bs)
        ln = (C.c_int64 * n)(*[len(b) for b in bs])
        h = C.c_void_p()
        check(lib().alfa_host_index_build_mem(n, nm, sq, ln, C.c_int64(K), int(suflink), int(child), int(kmer), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, prefix):
        h = C.c_void_p()
        check(lib().alfa_host_index_load(prefix.encode(), C.byref(h)))
        return cls(h)

    def save(self, prefix):
        check(lib().alfa_host_index_save(self._h, prefix.encode()))

    def arrays(self):
        return view_arrays(self.view)

    def descr(self, i):
        return lib().alfa_host_index_descr(self._h, i).decode()

    def chrom_bounds(self, hit):
        a, b = C.c_int64(), C.c_int64()
        lib().alfa_host_index_chrom_bounds(self._h, C.c_int64(hit), C.byref(a), C.byref(b))
        return a.value, b.value

    def __del__(self):
        try:
            if self._h:
                lib().alfa_host_index_free(self._h)
                self._h = None
        except Exception:
            pass


def index_build_on_device(device):
    """device >= 0: HostIndex.build_* construct their arrays on that GPU; -1: on the host again"""
    check(lib().alfa_index_build_on_device(int(device)))


def make_seed_queries(reads, min_len, max_count, sparse_q, max_smem_cont=0):
    """reads: list of bytes; scalar or per-read parameter lists."""
    n = len(reads)

    def per(x):
        return x if isinstance(x, (list, tuple, np.ndarray)) else [x] * n
    ml, mc, sq, ms = per(min_len), per(max_count), per(sparse_q), per(max_smem_cont)
    q = (SeedQuery * n)()
    for i, r in enumerate(reads):
        q[i].P = r
        q[i].plen = len(r)
        q[i].min_len = int(ml[i])
        q[i].max_count = int(mc[i])
        q[i].sparse_q = int(sq[i])
        q[i].max_smem_cont = int(ms[i])
    return q


def _seed_call(fn, handle_or_view, mem_type, queries, free_fn):
    nq = len(queries)
    mptr = C.c_void_p()
    offs = np.zeros(nq + 1, dtype=np.int64)
    rc = fn(handle_or_view, int(mem_type), queries, nq, C.byref(mptr), offs.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise AlfaError("seed batch failed rc=%d" % rc)
    total = int(offs[-1])
    if total:
        buf = C.string_at(mptr.value, total * MATCH_DTYPE.itemsize)
        matches = np.frombuffer(buf, dtype=MATCH_DTYPE).copy()
    else:
        matches = np.zeros(0, dtype=MATCH_DTYPE)
    free_fn(mptr)
    return matches, offs


def make_dp_tasks(pairs, w, h0=0):
    """pairs: list of (query uint8 ndarray, target uint8 ndarray); returns (ctypes task array, keepalive)."""
    n = len(pairs)

    def per(x):
        return x if isinstance(x, (list, tuple, np.ndarray)) else [x] * n
    ws, h0s = per(w), per(h0)
    t = (DpTask * n)()
    keep = []
    for i, (q, tg) in enumerate(pairs):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        tg = np.ascontiguousarray(tg, dtype=np.uint8)
        keep.append((q, tg))
        t[i].query = q.ctypes.data
        t[i].target = tg.ctypes.data
        t[i].qlen = len(q)
        t[i].tlen = len(tg)
        t[i].w = int(ws[i])
        t[i].h0 = int(h0s[i])
    return t, keep


def _global_call(fn, params, tasks, free_fn):
    n = len(tasks)
    score = np.zeros(n, dtype=np.int32)
    ncig = np.zeros(n, dtype=np.int32)
    coff = np.zeros(n + 1, dtype=np.int64)
    cptr = C.c_void_p()
    rc = fn(C.byref(params), tasks, n, score.ctypes.data_as(C.c_void_p), C.byref(cptr), ncig.ctypes.data_as(C.c_void_p),
            coff.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise AlfaError("ksw_global batch failed rc=%d" % rc)
    total = int(coff[-1])
    cig = np.frombuffer(C.string_at(cptr.value, total * 4), dtype=np.uint32).copy() if total else np.zeros(0, np.uint32)
    free_fn(cptr)
    return score, cig, ncig, coff


class DeviceIndex:
    """HBM-resident index (alfa_index_upload)."""

    def __init__(self, host_index, device=0):
        self.host = host_index
        h = C.c_void_p()
        check(lib().alfa_index_upload(C.byref(host_index.view), int(device), C.byref(h)))
        self._h = h

    @classmethod
    def load(cls, prefix, device=0, host_index=None):
        """index files straight to device memory (alfa_index_load_to_device); the host's share (text, names, offsets) comes
        back as self.host -- or is taken from `host_index` (a second GPU)"""
        hh = C.c_void_p(host_index._h.value if host_index is not None else None)
        h = C.c_void_p()
        check(lib().alfa_index_load_to_device(prefix.encode(), int(device), C.byref(hh), C.byref(h)))
        self = cls.__new__(cls)
        self.host = host_index if host_index is not None else HostIndex(hh)
        self._h = h
        return self

    def seed_batch(self, mem_type, queries):
        def fn(h, t, q, nq, m, o):
            rc = lib().alfa_seed_batch(h, t, q, nq, m, o)
            check(rc)
            return rc
        return _seed_call(fn, self._h, mem_type, queries, lib().alfa_free)

    def seed_regions_batch(self, mem_type, queries, edit_dist, max_seed_candidates=10000, max_region_mems=20, min_mem_reg_length=50,
                           error_percent=0.08):
        """alfa_seed_regions_batch: (matches, match_off, regions, region_off, members, member_off, status, chains)"""
        nq = len(queries)
        ed = np.ascontiguousarray(edit_dist, dtype=np.int32)
        prm = RegionParams(int(max_seed_candidates), int(max_region_mems), int(min_mem_reg_length), 0, float(error_percent))
        m, r, mb, ch = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        moff, roff, boff = (np.zeros(nq + 1, dtype=np.int64) for _ in range(3))
        status = np.zeros(max(nq, 1), dtype=np.int32)
        L = lib()
        L.alfa_seed_regions_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 8
        check(L.alfa_seed_regions_batch(self._h, int(mem_type), C.cast(queries, C.c_void_p), ed.ctypes.data_as(C.c_void_p), nq, C.byref(prm),
                                        C.byref(m), moff.ctypes.data_as(C.c_void_p), C.byref(r), roff.ctypes.data_as(C.c_void_p),
                                        C.byref(mb), boff.ctypes.data_as(C.c_void_p), C.byref(ch), status.ctypes.data_as(C.c_void_p)))

        def take(ptr, count, dt):
            out = np.frombuffer(C.string_at(ptr.value, int(count) * dt.itemsize), dtype=dt).copy() if count else np.zeros(0, dtype=dt)
            L.alfa_free(ptr)
            return out
        return (take(m, moff[nq], MATCH_DTYPE), moff, take(r, roff[nq], REGION_DTYPE), roff, take(mb, boff[nq], MATCH_DTYPE), boff, status[:nq],
                take(ch, boff[nq], np.dtype("<i4")))

    def __del__(self):
        try:
            if self._h:
                lib().alfa_index_free(self._h)
                self._h = None
        except Exception:
            pass


def ksw_extend_batch(params, tasks):
    n = len(tasks)
    out = np.zeros(n, dtype=EXTEND_DTYPE)
    check(lib().alfa_ksw_extend_batch(C.byref(params), tasks, n, out.ctypes.data_as(C.c_void_p)))
    return out


def ksw_nw_batch(params, tasks):
    n = len(tasks)
    out = np.zeros(n, dtype=np.int32)
    check(lib().alfa_ksw_nw_batch(C.byref(params), tasks, n, out.ctypes.data_as(C.c_void_p)))
    return out


def ksw_global_batch(params, tasks):
    def fn(*a):
        rc = lib().alfa_ksw_global_batch(*a)
        check(rc)
        return rc
    return _global_call(fn, params, tasks, lib().alfa_free)


def make_align_tasks(pairs, xtra):
    """pairs: list of (query uint8 ndarray, target uint8 ndarray); xtra: int or per-task list (KSW_X* flags | threshold)."""
    n = len(pairs)
    xs = xtra if isinstance(xtra, (list, tuple, np.ndarray)) else [xtra] * n
    t = (AlignTask * n)()
    keep = []
    for i, (q, tg) in enumerate(pairs):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        tg = np.ascontiguousarray(tg, dtype=np.uint8)
        keep.append((q, tg))
        t[i].query = q.ctypes.data
        t[i].target = tg.ctypes.data
        t[i].qlen = len(q)
        t[i].tlen = len(tg)
        t[i].xtra = int(xs[i])
    return t, keep


def align_call(fn, params, tasks):
    n = len(tasks)
    out = np.zeros(n, dtype=KSWR_DTYPE)
    check(fn(C.byref(params), tasks, n, out.ctypes.data_as(C.c_void_p)))
    return out


def ksw_align_batch(params, tasks):
    """alfa_ksw_align_batch: params.gapo / gape are used as given (the reference's callers pass negative numbers)."""
    return align_call(lib().alfa_ksw_align_batch, params, tasks)


def make_dpband_tasks(specs):
    """specs: dicts with ref, query (bytes) and optional refB, refE, queryB, queryE, flags (5 bools: freeRefB, freeRefE, freeQueryB,
    freeQueryE, local), out_type, band1, band2"""
    t = (DpbandTask * len(specs))()
    for i, sp in enumerate(specs):
        r, q = sp["ref"], sp["query"]
        t[i].ref, t[i].ref_len, t[i].query, t[i].query_len = r, len(r), q, len(q)
        t[i].refB, t[i].refE = sp.get("refB", 0), sp.get("refE", len(r) - 1)
        t[i].queryB, t[i].queryE = sp.get("queryB", 0), sp.get("queryE", len(q) - 1)
        f = sp.get("flags", (0, 0, 0, 0, 0))
        t[i].freeRefB, t[i].freeRefE, t[i].freeQueryB, t[i].freeQueryE, t[i].local = [int(bool(x)) for x in f]
        t[i].out_type, t[i].band1, t[i].band2 = sp.get("out_type", 4), sp.get("band1", 10), sp.get("band2", 10)
    t._keep = specs
    return t


def dpband_call(fn, variant, scores, tasks, free_fn):
    """-> list of (ret, dpScore, editDist, refB, refE, queryB, queryE, cigar) with cigar like '12=1X3D'"""
    n = len(tasks)
    res = (DpbandResult * max(n, 1))()
    cc, cl = C.c_void_p(), C.c_void_p()
    check(fn(int(variant), C.byref(scores), tasks, n, res, C.byref(cc), C.byref(cl)))
    out = []
    for i in range(n):
        r = res[i]
        cig = ""
        if r.n_cigar > 0:
            ch = C.string_at(cc.value + r.cig_off, r.n_cigar)
            ln = np.frombuffer(C.string_at(cl.value + 4 * r.cig_off, 4 * r.n_cigar), dtype=np.int32)
            cig = "".join("%d%s" % (int(l), chr(c)) for l, c in zip(ln, ch))
        out.append((r.ret, r.dpScore, r.editDist, r.refB, r.refE, r.queryB, r.queryE, cig))
    free_fn(cc)
    free_fn(cl)
    return out


def dp_band_batch(variant, scores, tasks):
    L = lib()
    return dpband_call(L.alfa_dp_band_batch, variant, scores, tasks, L.alfa_free)


def last_timing():
    t = Timing()
    check(lib().alfa_last_timing(C.byref(t)))
    return t


# ------------------------------------------------------------------ resident batches (create / launch / fetch)
DP_NW, DP_GLOBAL, DP_EXTEND = 0, 1, 2


class DpBatch:
    def __init__(self, mode, params, tasks, device=0):
        self.mode, self.n = mode, len(tasks)
        self._tasks = tasks
        h = C.c_void_p()
        check(lib().alfa_dp_batch_create(int(mode), C.byref(params), tasks, self.n, int(device), C.byref(h)))
        self._h = h
        L = lib()
        L.alfa_dp_batch_cells.restype = C.c_int64
        L.alfa_dp_batch_kernel_ms.restype = C.c_double

    def launch(self):
        check(lib().alfa_dp_batch_launch(self._h))

    def fetch(self):
        n = self.n
        if self.mode == DP_EXTEND:
            out = np.zeros(n, dtype=EXTEND_DTYPE)
            check(lib().alfa_dp_batch_fetch(self._h, out.ctypes.data_as(C.c_void_p), None, None, None))
            return out
        out = np.zeros(n, dtype=np.int32)
        if self.mode == DP_NW:
            check(lib().alfa_dp_batch_fetch(self._h, out.ctypes.data_as(C.c_void_p), None, None, None))
            return out
        ncig = np.zeros(n, dtype=np.int32)
        coff = np.zeros(n + 1, dtype=np.int64)
        cptr = C.c_void_p()
        check(lib().alfa_dp_batch_fetch(self._h, out.ctypes.data_as(C.c_void_p), C.byref(cptr), ncig.ctypes.data_as(C.c_void_p),
                                        coff.ctypes.data_as(C.c_void_p)))
        total = int(coff[-1])
        cig = np.frombuffer(C.string_at(cptr.value, total * 4), dtype=np.uint32).copy() if total else np.zeros(0, np.uint32)
        lib().alfa_free(cptr)
        return out, cig, ncig, coff

    @property
    def cells(self):
        return lib().alfa_dp_batch_cells(self._h)

    @property
    def kernel_ms(self):
        return lib().alfa_dp_batch_kernel_ms(self._h)

    def giveups(self):
        """(tasks in the band-in-registers kernel, ksw_extend tasks that left it in the last launch)"""
        L = lib()
        L.alfa_dp_batch_giveups.restype = C.c_int64
        nf = C.c_int64(0)
        g = L.alfa_dp_batch_giveups(self._h, C.byref(nf))
        return int(nf.value), int(g)

    def __del__(self):
        try:
            if self._h:
                lib().alfa_dp_batch_free(self._h)
                self._h = None
        except Exception:
            pass


class SeedPlan:
    def __init__(self, dev_index, mem_type, queries):
        self.index = dev_index
        self.nq = len(queries)
        self._keep = queries
        h = C.c_void_p()
        check(lib().alfa_seed_plan_create(dev_index._h, int(mem_type), queries, self.nq, C.byref(h)))
        self._h = h
        L = lib()
        L.alfa_seed_plan_queries.restype = C.c_int64
        L.alfa_seed_plan_kernel_ms.restype = C.c_double

    def launch(self):
        check(lib().alfa_seed_plan_launch(self._h))

    def fetch(self):
        def fn(h, t, q, nq, m, o):
            rc = lib().alfa_seed_plan_fetch(h, m, o)
            check(rc)
            return rc
        return _seed_call(fn, self._h, 0, self._keep, lib().alfa_free)

    @property
    def queries(self):
        return lib().alfa_seed_plan_queries(self._h)

    @property
    def kernel_ms(self):
        return lib().alfa_seed_plan_kernel_ms(self._h)

    @property
    def sorted(self):
        """queries evaluated in 10-mer order (the path for indexes beyond the caches)"""
        return bool(lib().alfa_seed_plan_sorted(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().alfa_seed_plan_free(self._h)
                self._h = None
        except Exception:
            pass


def sync(device=0):
    check(lib().alfa_sync(int(device)))


def set_lazy_join(device=0, on=True):
    check(lib().alfa_set_lazy_join(int(device), int(bool(on))))


def timer_start(device=0):
    check(lib().alfa_timer_start(int(device)))


def timer_stop(device=0):
    ms = C.c_double()
    check(lib().alfa_timer_stop(int(device), C.byref(ms)))
    return ms.value


# ---- ksw_global for callers that cannot allocate (alfa_ksw_global_noalloc): every array is a device pointer
DEV_TASK_DTYPE = np.dtype([("qoff", "<i8"), ("toff", "<i8"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"), ("pad", "<i4")])
NOALLOC_SCRATCH_BYTES = 32 << 20


def ksw_global_noalloc(params, d_tasks_ptr, n, max_tlen, d_seq_ptr, d_score_ptr, d_cigar_ptr, d_ncig_ptr, cig_stride, device=0, stream=0):
    """Thin binding: all *_ptr are integer device addresses (e.g. torch tensor .data_ptr()); enqueues only."""
    L = lib()
    L.alfa_ksw_global_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_void_p]
    rc = L.alfa_ksw_global_noalloc(C.byref(params), d_tasks_ptr, int(n), int(max_tlen), d_seq_ptr, d_score_ptr, d_cigar_ptr, d_ncig_ptr,
                                   int(cig_stride), int(device), stream)
    check(rc)


# ---- whole reads (alfa_map_reads_batch)
class MapOptions(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("open_gap", C.c_int32), ("extend_gap", C.c_int32),
                ("global_", C.c_int32), ("fixed_band", C.c_int32), ("max_trial", C.c_int32), ("max_trial_best", C.c_int32),
                ("max_aln_count", C.c_int32), ("rescue", C.c_int32),
                ("mem_type", C.c_int32), ("max_seed_candidates", C.c_int32), ("max_smem_start", C.c_int32), ("max_region_mems", C.c_int32),
                ("min_mem_reg_length", C.c_int32), ("cigar_alignments", C.c_int32), ("rescue_only", C.c_int32), ("reserved", C.c_int32),
                ("error_percent", C.c_double)]


class MapRead(C.Structure):
    _fields_ = [("P", C.c_char_p), ("Prc", C.c_char_p), ("plen", C.c_int32), ("min_len", C.c_int32), ("sparse_q", C.c_int32),
                ("edit_dist", C.c_int32), ("min_score", C.c_int32), ("min_coverage", C.c_int32), ("min_score2", C.c_int32),
                ("min_coverage2", C.c_int32), ("region_edit_dist", C.c_int32), ("reserved", C.c_int32)]


MAP_ALN_DTYPE = np.dtype([("ref_begin", "<i8"), ("ref_end", "<i8"), ("q_begin", "<i4"), ("q_end", "<i4"), ("score", "<i4"),
                          ("forward", "<i4"), ("chain_first", "<i4"), ("chain_count", "<i4"), ("op_first", "<i4"), ("op_count", "<i4"),
                          ("edit_dist", "<i4"), ("reserved", "<i4")])
MAP_OP_DTYPE = np.dtype([("len", "<i4"), ("code", "<i4")])


def default_map_options(error_percent=0.08):
    """mapOptions_t::initOptions (src/options.cpp:43-92) as alfa_map_options"""
    o = MapOptions()
    o.match, o.mismatch, o.open_gap, o.extend_gap = 1, -4, -6, -1
    o.global_, o.fixed_band, o.max_trial, o.max_trial_best, o.max_aln_count, o.rescue = 1, 100, 10, 1, 5000, 1
    o.mem_type, o.max_seed_candidates, o.max_smem_start, o.max_region_mems, o.min_mem_reg_length = SEED_SMAM, 10000, 10, 20, 50
    o.cigar_alignments = 1
    o.error_percent = error_percent
    return o


def make_map_reads(fw, rc, K, opt, min_mem=40, min_coverage=0.25):
    """per-read records with the reference's length-dependent arithmetic (src/mapper.cpp:1224-1231, 1280-1287, 533-536)"""
    n = len(fw)
    arr = (MapRead * n)()
    e = opt.error_percent
    for i in range(n):
        pl = len(fw[i])
        ed = int(e * pl) + 1
        min_len = min_mem
        if pl >= 1000:
            min_len = int(min_len + (20 * (pl // 500)) / (100 * e))
        sq = max((min_len - 10) // K, 1) if K >= 4 else max((min_len - 12) // K, 1)
        a = arr[i]
        a.P, a.Prc, a.plen, a.min_len, a.sparse_q, a.edit_dist = fw[i], rc[i], pl, min_len, sq, ed
        a.min_score = int(pl * opt.match + (opt.mismatch - opt.match) * ed)
        a.min_coverage = int(pl * min_coverage)
        a.min_score2 = max(0, int(pl * opt.match + (opt.mismatch - opt.match) * 0.2 * pl))
        a.min_coverage2 = int(pl * 0.01)
    return arr


last_map_call_seconds = None


def map_reads_batch(dev_index, opt, params, reads):
    """-> (status[n], aln_off[n+1], alignments, seed_off[n+1], chain seeds, op_off[n+1], CIGAR operations)"""
    n = len(reads)
    st = np.zeros(n, np.int32)
    aoff = np.zeros(n + 1, np.int64)
    soff = np.zeros(n + 1, np.int64)
    ooff = np.zeros(n + 1, np.int64)
    ap, sp, op = C.c_void_p(), C.c_void_p(), C.c_void_p()
    global last_map_call_seconds
    t0 = time.perf_counter()
    rc = lib().alfa_map_reads_batch(dev_index._h, C.byref(opt), C.byref(params), reads, n, st.ctypes.data_as(C.c_void_p),
                                    aoff.ctypes.data_as(C.c_void_p), C.byref(ap), soff.ctypes.data_as(C.c_void_p), C.byref(sp),
                                    ooff.ctypes.data_as(C.c_void_p), C.byref(op))
    last_map_call_seconds = time.perf_counter() - t0        # the library call alone (numpy copies of its results come after)
    check(rc)

    def take(ptr, count, dtype):
        if not count:
            return np.zeros(0, dtype)
        buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).copy()
    al = take(ap, int(aoff[-1]), MAP_ALN_DTYPE)
    sd = take(sp, int(soff[-1]), MATCH_DTYPE)
    ops = take(op, int(ooff[-1]), MAP_OP_DTYPE)
    lib().alfa_free(op)
    lib().alfa_free(ap)
    lib().alfa_free(sp)
    return st, aoff, al, soff, sd, ooff, ops
