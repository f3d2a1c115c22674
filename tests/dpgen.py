"""Deterministic generators of DP tasks shared by the oracle and GPU parity tests."""
import numpy as np


def mutate(rng, s, p):
    out = []
    for ch in s:
        r = rng.random()
        if r < p * 0.6:
            out.append(int(rng.integers(0, 4)))
        elif r < p * 0.8:
            out.append(int(ch))
            out.append(int(rng.integers(0, 4)))
        elif r < p:
            pass
        else:
            out.append(int(ch))
    return np.array(out, dtype=np.uint8)


def random_tasks(seed, n, qlens=(0, 1, 2, 3, 5, 10, 30, 60, 120, 250, 600), bands=(0, 1, 2, 5, 10, 21, 50, 100, 1000),
                 h0s=(0, 1, 5, 20, 40, 100, 250, -3), with_n=True):
    """Mixed bag: related pairs at several error rates, unrelated pairs, N bases, degenerate lengths."""
    rng = np.random.default_rng(seed)
    pairs, ws, hs = [], [], []
    for _ in range(n):
        ql = int(rng.choice(qlens))
        if rng.random() < 0.5:
            ql = int(rng.integers(0, ql + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        if rng.random() < 0.6:
            t = mutate(rng, q, float(rng.choice([0, 0.02, 0.1, 0.3])))
            if rng.random() < 0.7:
                t = np.concatenate([t, rng.integers(0, 4, int(rng.integers(0, 40))).astype(np.uint8)])
        else:
            t = rng.integers(0, 4, int(rng.integers(0, 300))).astype(np.uint8)
        if with_n and rng.random() < 0.1 and len(q) > 0:
            q[rng.integers(0, len(q))] = 4
        if with_n and rng.random() < 0.1 and len(t) > 0:
            t[rng.integers(0, len(t))] = 4
        pairs.append((q, t))
        ws.append(int(rng.choice(bands)))
        hs.append(int(rng.choice(h0s)))
    return pairs, ws, hs


def mapper_like_tasks(seed, n, read_len=250, err=0.02, band=21):
    """Extension-shaped tasks: query = read overhang, target = query + editDist extra reference bases."""
    rng = np.random.default_rng(seed)
    pairs, ws, hs = [], [], []
    for _ in range(n):
        ql = int(rng.integers(1, read_len))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        t = mutate(rng, q, err)
        t = np.concatenate([t, rng.integers(0, 4, band).astype(np.uint8)])
        pairs.append((q, t))
        ws.append(min(band, ql))
        hs.append(int(rng.integers(20, read_len)))
    return pairs, ws, hs


# ---------------------------------------------------------------- dynProg (dp.cpp) cases, SURVEY 8c item 2(iii)
def dpband_valid_flags(variant):
    """dp_type flag tuples (freeRefB, freeRefE, freeQueryB, freeQueryE, local) the variant accepts"""
    out = []
    for m in range(32):
        f = tuple((m >> k) & 1 for k in range(5))
        if variant == 0:        # dpBandStatic asserts: not local, no free begin together with a free end (src/dp.cpp:663-664)
            if f[4] or ((f[0] or f[2]) and (f[1] or f[3])):
                continue
        out.append(f)
    return out


def dpband_cases(rng, n, variant, max_len=200, alphabet=b"acgt"):
    """random (ref, query) pairs related by substitutions and indels, embedded in longer strings with boundaries"""
    import numpy as np
    flags = dpband_valid_flags(variant)
    specs = []
    for _ in range(n):
        L = int(rng.integers(0, max_len + 1)) if rng.random() < 0.9 else int(rng.integers(0, 4))
        a = bytes(rng.choice(np.frombuffer(alphabet, np.uint8), L))
        b = bytearray()
        err = rng.choice([0.0, 0.03, 0.1, 0.3])
        for ch in a:
            u = rng.random()
            if u < err * 0.5:
                b.append(alphabet[int(rng.integers(0, len(alphabet)))])
            elif u < err * 0.75:
                b.append(ch); b.append(alphabet[int(rng.integers(0, len(alphabet)))])
            elif u < err:
                pass
            else:
                b.append(ch)
        if rng.random() < 0.15:       # unrelated tail / head so that free ends matter
            b = bytearray(rng.choice(np.frombuffer(alphabet, np.uint8), int(rng.integers(0, 15)))) + b
        if rng.random() < 0.15:
            a = a + bytes(rng.choice(np.frombuffer(alphabet, np.uint8), int(rng.integers(0, 15))))
        if rng.random() < 0.1:
            b = bytearray(bytes(b).upper()) if rng.random() < 0.5 else b + b"n"
        pr, pq = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        ref = b"t" * pr + a + b"g" * int(rng.integers(0, 4))
        qry = b"c" * pq + bytes(b) + b"a" * int(rng.integers(0, 4))
        sp = dict(ref=ref, query=qry, refB=pr, refE=pr + len(a) - 1, queryB=pq, queryE=pq + len(b) - 1,
                  flags=flags[int(rng.integers(0, len(flags)))], out_type=int(rng.choice([0, 1, 3, 4])))
        if variant == 0:
            sp["band1"] = int(rng.choice([1, 2, 5, 10, 30]))
            sp["band2"] = int(rng.choice([1, 10, 40, 300]))
        else:
            if rng.random() < 0.7:
                sp["band1"] = sp["band2"] = int(rng.choice([1, 3, 10, 30]))
            else:
                sp["band1"], sp["band2"] = int(rng.choice([1, 3, 10, 30])), int(rng.choice([1, 3, 10, 30]))
        specs.append(sp)
    return specs


DPBAND_SCORES = [(1, -4, -6, -1), (2, -2, -2, -2), (1, -1, 0, -1), (0, -2, 0, -2), (3, -5, -8, -3), (1, -3, 0, -2)]


# ---------------------------------------------------------------- ksw_align (src/ksw.c:330) cases
# (match, mismatch, gapo, gape) exactly as handed to ksw_align: the reference's call sites pass NEGATIVE gap numbers
ALIGN_SCORES = [(1, -4, -6, -1), (1, -4, 6, 1), (2, -3, 5, 2), (1, -1, 0, 1), (1, -2, 0, 0), (5, -4, 10, 1), (2, -2, -2, -2),
                (1, -4, 1, 6), (3, -1, 2, 0)]
XBYTE, XSTOP, XSUBO, XSTART = 0x10000, 0x20000, 0x40000, 0x80000


def align_cases(seed, n, max_q=300, max_t=700, match=1):
    """mate-rescue shaped local alignments: a window with zero, one or two (mutated) copies of the query.
    Returns (pairs, xtras).  The xtra of most cases is the mapper's own (KSW_XSTART | KSW_XBYTE when qlen*match < 250)."""
    rng = np.random.default_rng(seed)
    pairs, xs = [], []
    qchoices = (0, 1, 2, 7, 8, 9, 15, 16, 17, 31, 33, 50, 100, 101, 150, 249, 250, max_q)
    for _ in range(n):
        ql = int(rng.choice(qchoices)) if rng.random() < 0.5 else int(rng.integers(0, max_q + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        parts = [rng.integers(0, 4, int(rng.integers(0, max_t // 3 + 1))).astype(np.uint8)]
        copies = int(rng.choice([0, 1, 1, 1, 2]))
        for _c in range(copies):
            a, b = sorted(int(x) for x in rng.integers(0, ql + 1, 2)) if rng.random() < 0.3 else (0, ql)
            parts.append(mutate(rng, q[a:b], float(rng.choice([0, 0.02, 0.05, 0.15]))))
            parts.append(rng.integers(0, 4, int(rng.integers(0, max_t // 4 + 1))).astype(np.uint8))
        t = np.concatenate(parts)[:max_t]
        if rng.random() < 0.03:
            t = t[:int(rng.integers(0, 3))]
        if rng.random() < 0.1 and len(q):
            q[rng.integers(0, len(q))] = 4
        if rng.random() < 0.1 and len(t):
            t[rng.integers(0, len(t))] = 4
        u = rng.random()
        x = XSTART | (XBYTE if ql * match < 250 else 0)
        if u < 0.15:
            x = XSTART | XSUBO | int(rng.integers(0, 60)) | (XBYTE if rng.random() < 0.5 else 0)
        elif u < 0.25:
            x = XSTOP | int(rng.integers(1, 80)) | (XSTART if rng.random() < 0.5 else 0) | (XBYTE if rng.random() < 0.5 else 0)
        elif u < 0.32:
            x = int(rng.choice([0, XBYTE, XSUBO | 10, XSUBO | XBYTE | 5, XSTART, XSTART | XBYTE]))
        pairs.append((q, t))
        xs.append(int(x))
    return pairs, xs


def align_params(sc):
    """DpParams for one ALIGN_SCORES entry (gap numbers untouched)"""
    from alfalfa_amd import ffi
    p = ffi.default_params(match=sc[0], mismatch=sc[1])
    p.gapo, p.gape = sc[2], sc[3]
    return p
