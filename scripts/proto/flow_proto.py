"""Prototype of the exact parallel evaluation of a sequential rejection chain ("coalescing flows").

Chain: rows t = 0..n-1 are served in order from a stream of quads j = 0, 1, ...: row t takes the first
quad j >= (quad after the one row t-1 took) with A(t, j) true.  State (t, j); step: t += A(t, j); j += 1.
Chunks of Lq quads; for chunk c every candidate entering row in a window is walked, walkers that reach
the same row merge (paths are monotone, never cross).  Then the chunks' maps are composed serially.
"""
import numpy as np, sys, time

def accept(t, j, p):
    # deterministic pseudo-random predicate with P(accept) = p[t]
    h = (t.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) ^ (j.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)))
    h ^= h >> np.uint64(29); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(32)
    h *= np.uint64(0x94D049BB133111EB); h ^= h >> np.uint64(29)
    u = (h >> np.uint64(11)).astype(np.float64) / 2.0**53
    return u < p[np.minimum(t, len(p) - 1)]

def sequential(p):
    n = len(p); t = 0; j = 0; took = np.empty(n, np.int64)
    while t < n:
        # vector burst: try 64 quads for this row
        js = np.arange(j, j + 64)
        a = accept(np.full(64, t), js, p)
        k = np.argmax(a) if a.any() else -1
        while k < 0:
            j += 64; js = np.arange(j, j + 64); a = accept(np.full(64, t), js, p); k = np.argmax(a) if a.any() else -1
        took[t] = j + k; j = j + k + 1; t += 1
    return took, j

def flows(p, Lq, ksig=5.0, subq=256):
    n = len(p)
    m = 1 / p; v = (1 - p) / p**2
    M = np.concatenate([[0], np.cumsum(m)]); V = np.concatenate([[0], np.cumsum(v)])
    total_q = int(M[-1] + ksig * np.sqrt(V[-1])) + 1
    C = (total_q + Lq - 1) // Lq
    evals = 0; finals = []; snaps = []; maxlive_end = 0
    for c in range(C):
        J = c * Lq
        if c == 0:
            lo = hi = 0
        else:
            tstar = int(np.searchsorted(M, J))  # expected row at quad J
            tstar = min(tstar, n)
            h = int(np.ceil(ksig * np.sqrt(V[tstar]))) + 2
            lo = max(0, tstar - h); hi = min(n, tstar + h)
        first_in = np.arange(lo, hi + 1); cur = first_in.copy()
        csn = []
        for s in range(Lq):
            a = accept(cur, np.full(len(cur), J + s), p) & (cur < n)
            evals += len(cur)
            cur = cur + a
            if (s + 1) % 16 == 0 or s == Lq - 1:
                keep = np.concatenate([[True], cur[1:] != cur[:-1]])
                cur = cur[keep]; first_in = first_in[keep]
            if (s + 1) % subq == 0 and s != Lq - 1:
                csn.append((first_in.copy(), cur.copy()))
        finals.append((lo, hi, first_in, cur)); snaps.append(csn)
        maxlive_end = max(maxlive_end, len(cur))
    # resolve
    T = 0; entering = []; ok = True
    for c in range(C):
        lo, hi, fi, cur = finals[c]
        entering.append(T)
        if T < lo or T > hi:
            ok = False; break
        k = np.searchsorted(fi, T, side='right') - 1
        T = int(cur[k])
    return dict(C=C, evals=evals, ok=ok, entering=entering, finals=finals, snaps=snaps, maxlive_end=maxlive_end)

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    Lq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    rng = np.random.default_rng(0)
    p = rng.uniform(0.45, 0.95, n)
    t0 = time.time(); took, jend = sequential(p); print('sequential', time.time() - t0, 'quads', jend, 'per row', jend / n)
    t0 = time.time(); R = flows(p, Lq); print('flows', time.time() - t0)
    print('chunks', R['C'], 'ok', R['ok'], 'evals/quad', R['evals'] / jend, 'max live at chunk end', R['maxlive_end'])
    # check entering rows against the sequential path: row at quad J = number of rows whose quad < J
    for c, T in enumerate(R['entering']):
        J = c * Lq
        want = int(np.searchsorted(took, J))  # rows served before quad J
        assert T == want, (c, T, want)
    print('entering rows of all chunks exact')

def survivors(n=400000, W=20000, steps=40000, seed=1):
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.45, 0.95, n)
    cur = np.arange(1000, 1000 + W)
    out = []
    for s in range(steps):
        a = accept(cur, np.full(len(cur), 5000 + s), p)
        cur = cur + a
        if (s + 1) % 8 == 0:
            cur = cur[np.concatenate([[True], cur[1:] != cur[:-1]])]
        if (s + 1) in (64, 256, 1024, 4096, 16384, 40000):
            out.append((s + 1, len(cur), len(cur) * np.sqrt(s + 1) / W))
    return out
