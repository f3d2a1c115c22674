"""Deterministic synthetic references / reads (iid ACGT reference, 60/20/20 sub/ins/del errors)."""
import numpy as np

ACGT = np.frombuffer(b"acgt", dtype=np.uint8)
_RC = bytes.maketrans(b"acgtn", b"tgcan")


def rand_ref(rng, n):
    return ACGT[rng.integers(0, 4, n)].tobytes()


def revcomp(s):
    return s[::-1].translate(_RC)


def mutate(rng, s, p):
    """per-base error p: 60 % substitution, 20 % insertion, 20 % deletion (vectorised)."""
    a = np.frombuffer(s, dtype=np.uint8)
    r = rng.random(len(a))
    sub = r < 0.6 * p
    ins = (r >= 0.6 * p) & (r < 0.8 * p)
    dele = (r >= 0.8 * p) & (r < p)
    a = a.copy()
    a[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    keep = ~dele
    reps = keep.astype(np.int64) + ins.astype(np.int64)
    out = np.repeat(a, reps)
    # inserted base = second copy of an 'ins' position: randomise it
    pos = np.cumsum(reps)[ins] - 1
    out[pos] = ACGT[rng.integers(0, 4, len(pos))]
    return out.tobytes()


def sim_reads(rng, ref, n, lo, hi, err, both_strands=True):
    """n reads of length U[lo,hi] drawn uniformly over `ref` (bytes), 50/50 strand, per-base error err."""
    reads = []
    L = len(ref)
    for _ in range(n):
        ln = int(rng.integers(lo, hi + 1))
        p = int(rng.integers(0, L - ln))
        r = mutate(rng, ref[p:p + ln], err)
        if both_strands and rng.random() < 0.5:
            r = revcomp(r)
        reads.append(r)
    return reads


def seed_params(K, plen, err_percent=0.08, min_mem=40, fixed_min=False):
    """min_len and sparseSkip exactly as the reference mapper derives them
    (src/mapper.cpp:1224-1231 and :533-536)."""
    edit = int(err_percent * plen) + 1
    min_len = min_mem
    if not fixed_min and plen >= 1000:
        min_len += int((20 * (plen // 500)) / (int(100) * err_percent))
    if fixed_min and plen // edit > min_len:
        min_len = plen // edit
    if K >= 4:
        sq = max((min_len - 10) // K, 1)
    else:
        sq = max((min_len - 12) // K, 1)
    return min_len, sq


def sim_pairs(rng, ref, n, rlen, ins_lo, ins_hi, err1, err2, margin):
    """n FR read pairs: fragment of length U[ins_lo,ins_hi] at least `margin` away from both ends of `ref`; mate 1 is the
    fragment's left end (forward), mate 2 its right end (reverse complement); the pair as a whole is flipped half of the
    time.  err1/err2: per-base error of the two mates (a callable gives a per-pair rate)."""
    m1, m2 = [], []
    L = len(ref)
    for _ in range(n):
        ins = int(rng.integers(ins_lo, ins_hi + 1))
        p = int(rng.integers(margin, L - margin - ins))
        frag = ref[p:p + ins]
        e1 = err1(rng) if callable(err1) else err1
        e2 = err2(rng) if callable(err2) else err2
        a = mutate(rng, frag[:rlen], e1)
        if e2 == "periodic":        # a substitution every 14..18 bases and no indel: no seed of 20+, yet a high ungapped score
            x = np.frombuffer(frag[-rlen:], dtype=np.uint8).copy()
            step = int(rng.integers(14, 19))
            for q in range(int(rng.integers(0, step)), len(x), step):
                x[q] = ACGT[(int(np.nonzero(ACGT == x[q])[0][0]) + int(rng.integers(1, 4))) % 4]
            b = revcomp(x.tobytes())
        else:
            b = revcomp(mutate(rng, frag[-rlen:], e2))
        if rng.random() < 0.5:
            a, b = b, a
        m1.append(a)
        m2.append(b)
    return m1, m2
