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
