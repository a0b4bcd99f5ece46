"""Survivor statistics of the coalescing flow with the REAL acceptance functions (util.hpp:15-60) on shared quads."""
import numpy as np, sys
from scipy.special import erfc, erf

def make_rows(n, task, rng, spread=1.5):
    pred = rng.normal(0, spread, n)
    lat = pred + rng.normal(0, 1, n)
    if task == 'ordered':
        gam = np.quantile(lat, [0.2, 0.4, 0.6, 0.8])
        cls = np.digitize(lat, gam)
        A = np.empty(n); B = np.empty(n)
        for c in range(5):
            m = cls == c
            if c == 0:
                mu = -(gam[0] - pred[m]); A[m] = mu; B[m] = np.nan
            elif c == 4:
                mu = gam[3] - pred[m]; A[m] = mu; B[m] = np.nan
            else:
                A[m] = gam[c-1] - pred[m]; B[m] = gam[c] - pred[m]
    else:
        yy = lat > 0
        mu = np.where(yy, -pred, pred); A = mu; B = np.full(n, np.nan)
    one = np.isnan(B)
    E = one & (A >= 0)
    alpha = (A + np.sqrt(A*A + 4)) / 2
    A2 = np.where(E, alpha, A); B2 = np.where(E, A, B)
    return A2, B2

def quads(nq, rng):
    u1 = rng.random(nq); u2 = rng.random(nq)
    x = 2*u1-1; y = 2*u2-1; r2 = x*x+y*y
    ok = (r2 <= 1) & (r2 > 0)
    mult = np.sqrt(-2*np.log(np.where(ok, r2, 0.5))/np.where(ok, r2, 0.5))
    n1 = np.where(ok, y*mult, np.nan); n2 = np.where(ok, x*mult, np.nan)
    return u1, np.log(u2), -np.log(u1), n1, n2

def accept(A, B, t, q, j):
    u1, l2, nl1, n1, n2 = q
    a = A[t]; b = B[t]
    isN = np.isnan(b); isE = (~isN) & (a > b)
    with np.errstate(invalid='ignore'):
        accN = (n1[j] > a) | (n2[j] > a)
        zE = nl1[j]/a + b; argE = -(zE-a)**2/2
        zT = u1[j]*(b-a) + a
        c = np.where((a <= 0) & (b >= 0), 0.0, np.where(b < 0, b*b, a*a))
        argT = (c - zT*zT)/2
        acc = np.where(isN, accN, np.where(isE, l2[j] < argE, l2[j] < argT))
    return acc

def probs(A, B):
    isN = np.isnan(B); isE = (~isN) & (A > B)
    p = np.empty(len(A))
    Phi = 0.5*erfc(-A[isN]/np.sqrt(2)); p[isN] = np.pi/4*(1-Phi**2)
    a = A[isE]; m = B[isE]
    p[isE] = a*np.sqrt(np.pi/2)*np.exp(-0.5*(a-m)**2)*np.exp(m*m/2)*erfc(m/np.sqrt(2))
    T = ~(isN | isE); a = A[T]; b = B[T]
    base = np.sqrt(2*np.pi)*0.5*(erf(b/np.sqrt(2)) - erf(a/np.sqrt(2)))/(b-a)
    f = np.where((a <= 0) & (b >= 0), 1.0, np.where(b < 0, np.exp(b*b/2), np.exp(a*a/2)))
    p[T] = base*f
    return p

if __name__ == '__main__':
    task = sys.argv[1]; spread = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
    rng = np.random.default_rng(0)
    n = 600000; A, B = make_rows(n, task, rng, spread)
    p = probs(A, B); m = 1/p; v = (1-p)/p**2
    print(task, 'mean quads/row', m.mean(), 'var/row', v.mean(), 'types N/E/T', np.isnan(B).mean(), ((~np.isnan(B)) & (A > B)).mean())
    nq = 400000; q = quads(nq, rng)
    # check the closed-form p against the simulation: true path from row 0
    t = 0; ts = np.zeros(nq, np.int64)
    # flow of W walkers from a window in the middle
    W = 20000; cur = np.arange(100000, 100000 + W)
    for s in range(60000):
        a = accept(A, B, cur, q, np.full(len(cur), s))
        cur = cur + a
        if (s+1) % 8 == 0:
            cur = cur[np.concatenate([[True], cur[1:] != cur[:-1]])]
        if (s+1) in (64, 256, 1024, 4096, 16384, 60000):
            print(s+1, len(cur), len(cur)*np.sqrt(s+1)/W, 'spread', cur[-1]-cur[0])
    # bias check: one walker over 60000 quads vs expected rows advanced
    M = np.concatenate([[0], np.cumsum(m)])
    adv = cur[0] - 100000
    print('walker advanced', adv, 'rows; expected', np.searchsorted(M, M[100000] + 60000) - 100000, 'sd', np.sqrt(v[100000:100000+adv].sum())/m.mean())
