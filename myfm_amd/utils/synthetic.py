"""Seeded synthetic workloads: the shapes BASELINE.json names (bench.py, smoke()) and the small designs of the parity tests
(tests/datasets.py re-exports this module).

Own numpy code. The small designs mirror the *shapes and seeds* of the reference's test
fixtures so that the same identities can be asserted:
  middle_data  <- tests/conftest.py:13-45      (N=1000, 3 features, values in {-2,-1,1,2})
  block_design <- tests/regression/test_block.py:80-113 (N=100, 1+3+2 features, 2 blocks)
  toy          <- README.md:46-59 / examples/toy.py
"""
import numpy as np
import scipy.sparse as sps

STUB_W0 = -3.0
STUB_W = np.array([1.0, 2.0, -1.0])
STUB_V = np.array([[1.0, -1.0, 0], [0.0, 1.0, 1.0], [1.0, 1.0, 1.0], [-1.0, 0, -1.0]])  # (latent, feature)


def fm_score(X, w0, w, V):
    """closed form  w0 + Xw + 1/2 [ (XV)^2 . 1 - X^2 . sum V^2 ]   (V is (D, K))."""
    X = sps.csr_matrix(X)
    X2 = X.copy()
    X2.data = X2.data ** 2
    XV = X.dot(V)
    return w0 + X.dot(w) + 0.5 * ((XV ** 2).sum(axis=1) - X2.dot((V ** 2).sum(axis=1)))


def toy():
    X = np.array(
        [
            [19.0, 0, 0, 0, 1, 1, 0, 0, 0],
            [33.0, 0, 0, 1, 0, 0, 1, 0, 0],
            [55.0, 0, 1, 0, 0, 0, 0, 1, 0],
            [20.0, 1, 0, 0, 0, 0, 0, 0, 1],
        ]
    )
    return sps.csr_matrix(X), np.array([0.0, 1.0, 1.0, 0.0])


def middle_data(n_train=1000):
    rns = np.random.RandomState(0)
    rows, cols, data = [], [], []
    for row in range(n_train):
        indices = np.where(rns.random(3) > 0.5)[0]
        for ind in indices:
            rows.append(row)
            cols.append(ind)
            data.append(float(rns.choice([-2, -1, 1, 2])))
    X = sps.csr_matrix((data, (rows, cols)), shape=(n_train, 3))
    return X, fm_score(X, STUB_W0, STUB_W, STUB_V.T)


def block_design(n_train=100, seed=0):
    rns = np.random.RandomState(seed)
    user_block = sps.csr_matrix(np.eye(3))
    user_indices = rns.randint(0, 3, size=n_train)
    item_block = sps.csr_matrix(np.eye(2))
    group_shapes = [1, 3, 2]
    item_indices = rns.randint(0, 2, size=n_train)
    tm_column = rns.randn(n_train, 1)
    X_flat = sps.hstack([tm_column, user_block[user_indices], item_block[item_indices]]).tocsr()
    weights = rns.randn(3, X_flat.shape[1])
    y = fm_score(X_flat, 0.0, np.zeros(X_flat.shape[1]), weights.T) + rns.randn(n_train)
    blocks = [(user_indices.astype(np.int64), user_block), (item_indices.astype(np.int64), item_block)]
    return sps.csr_matrix(tm_column), X_flat, blocks, y, group_shapes


def multihot_block_design(n_train=400, seed=3):
    """Blocks with multi-hot, non-unit values so that block columns conflict (several levels)."""
    rng = np.random.default_rng(seed)
    ub = sps.random(12, 7, density=0.4, random_state=np.random.RandomState(seed), format="csr")
    ub.data = np.round(rng.uniform(-1.5, 1.5, size=ub.nnz), 2)
    ib = sps.random(9, 5, density=0.5, random_state=np.random.RandomState(seed + 1), format="csr")
    ib.data = np.round(rng.uniform(0.2, 1.0, size=ib.nnz), 2)
    ui = rng.integers(0, 12, size=n_train)
    ii = rng.integers(0, 9, size=n_train)
    main = sps.random(n_train, 4, density=0.5, random_state=np.random.RandomState(seed + 2), format="csr")
    main.data = np.round(rng.normal(size=main.nnz), 2)
    X_flat = sps.hstack([main, ub[ui], ib[ii]]).tocsr()
    D = X_flat.shape[1]
    w = rng.normal(size=D) * 0.3
    V = rng.normal(size=(D, 3)) * 0.4
    y = fm_score(X_flat, 0.5, w, V) + rng.normal(size=n_train) * 0.5
    blocks = [(ui.astype(np.int64), ub), (ii.astype(np.int64), ib)]
    return main, X_flat, blocks, y, [4, 7, 5]


def onehot_mf(n_rows, n_users, n_items, rank_true=8, seed=0, sort_by_user=True, noise=0.3, zipf=1.0):
    """MovieLens-shaped two-field one-hot design: returns (X csr, y, group_shapes)."""
    rng = np.random.default_rng(seed)
    pu = 1.0 / np.arange(1, n_users + 1) ** (0.5 * zipf)
    pi = 1.0 / np.arange(1, n_items + 1) ** zipf
    u = rng.choice(n_users, size=n_rows, p=pu / pu.sum())
    i = rng.choice(n_items, size=n_rows, p=pi / pi.sum())
    # shuffle ids so that popularity is not monotone in the index
    u = rng.permutation(n_users)[u]
    i = rng.permutation(n_items)[i]
    if sort_by_user:
        order = np.argsort(u, kind="stable")
        u, i = u[order], i[order]
    bu = rng.normal(size=n_users) * 0.3
    bi = rng.normal(size=n_items) * 0.3
    U = rng.normal(size=(n_users, rank_true)) * 0.3
    It = rng.normal(size=(n_items, rank_true)) * 0.3
    y = 3.5 + bu[u] + bi[i] + (U[u] * It[i]).sum(axis=1) + rng.normal(size=n_rows) * noise
    y = np.clip(np.round(y * 2) / 2, 0.5, 5.0)
    indptr = np.arange(0, 2 * n_rows + 1, 2, dtype=np.int64)
    indices = np.empty(2 * n_rows, dtype=np.int32)
    indices[0::2] = u
    indices[1::2] = n_users + i
    data = np.ones(2 * n_rows)
    X = sps.csr_matrix((data, indices, indptr), shape=(n_rows, n_users + n_items))
    return X, y, [n_users, n_items]


def group_index_from_shapes(shapes):
    return np.concatenate([np.full(s, g, dtype=np.int32) for g, s in enumerate(shapes)])


def movielens_like(n_rows, n_users, n_items, rank_true=32, seed=1, noise=0.8, user_offset=700.0, item_offset=40.0):
    """ML-10M-shaped two-field one-hot design (BASELINE config 3 / SURVEY 8d).

    Popularity follows p(rank r) ~ 1 / (r + offset): with the defaults and 10 M rows the most active
    user has ~3 k ratings and the least ~30 (ML-10M: 20 .. 7 359, mean 143), the most popular item
    ~45 k and the median item ~330 (ML-10M: max 34 864). Rows are sorted by user like the MovieLens
    files. Targets: rank-`rank_true` truth + noise on the half-star grid.
    """
    rng = np.random.default_rng(seed)
    pu = 1.0 / (np.arange(1, n_users + 1) + user_offset)
    pi = 1.0 / (np.arange(1, n_items + 1) + item_offset)
    u = rng.choice(n_users, size=n_rows, p=pu / pu.sum()).astype(np.int32)
    i = rng.choice(n_items, size=n_rows, p=pi / pi.sum()).astype(np.int32)
    u = rng.permutation(n_users).astype(np.int32)[u]
    i = rng.permutation(n_items).astype(np.int32)[i]
    order = np.argsort(u, kind="stable")
    u, i = u[order], i[order]
    bu = rng.normal(size=n_users) * 0.4
    bi = rng.normal(size=n_items) * 0.4
    U = rng.normal(size=(n_users, rank_true)) * (0.6 / np.sqrt(rank_true))
    It = rng.normal(size=(n_items, rank_true)) * (0.6 / np.sqrt(rank_true)) * np.sqrt(rank_true)
    y = np.empty(n_rows)
    step = 2_000_000
    for s in range(0, n_rows, step):
        e = min(n_rows, s + step)
        y[s:e] = 3.5 + bu[u[s:e]] + bi[i[s:e]] + np.einsum("ij,ij->i", U[u[s:e]], It[i[s:e]])
    y += rng.normal(size=n_rows) * noise
    y = np.clip(np.round(y * 2) / 2, 0.5, 5.0)
    indptr = np.arange(0, 2 * n_rows + 1, 2, dtype=np.int64)
    indices = np.empty(2 * n_rows, dtype=np.int32)
    indices[0::2] = u
    indices[1::2] = n_users + i
    X = sps.csr_matrix((np.ones(2 * n_rows), indices, indptr), shape=(n_rows, n_users + n_items))
    return X, y, [n_users, n_items]


def movielens_like_shard(rows_per_rank, rank, world, n_users, n_items, rank_true=32, seed=1, noise=0.8, user_offset=700.0,
                         item_offset=40.0):
    """This rank's contiguous slice (about rows_per_rank rows, cut at user boundaries) of ONE user-sorted table
    of world * rows_per_rank rows with the popularity profile of movielens_like -- the layout a row-sharded
    multi-GPU run sees: every rank holds a contiguous range of users. Every rank derives the same global user
    boundaries and truth parameters from `seed`; items and noise are per rank.
    Returns (X csr local rows x all features, y, group_shapes, first global row, total rows)."""
    total = world * rows_per_rank
    g = np.random.default_rng(seed)
    pu = 1.0 / (np.arange(1, n_users + 1) + user_offset)
    pu = pu[g.permutation(n_users)]
    bounds = np.floor(np.cumsum(pu / pu.sum()) * total + 0.5).astype(np.int64)  # user u owns rows [bounds[u-1], bounds[u])
    bounds[-1] = total
    iperm = g.permutation(n_items).astype(np.int32)
    bu = g.normal(size=n_users) * 0.4
    bi = g.normal(size=n_items) * 0.4
    U = g.normal(size=(n_users, rank_true)) * (0.6 / np.sqrt(rank_true))
    It = g.normal(size=(n_items, rank_true)) * 0.6
    # shard boundaries snapped to user boundaries (a loader sharding a user-sorted table by users): the ranks'
    # row counts differ by at most one user's rows and no user is split between two ranks
    edges = np.concatenate([[0], bounds])
    cut = [int(edges[np.argmin(np.abs(edges - r * rows_per_rank))]) for r in range(world + 1)]
    cut[0], cut[-1] = 0, total
    lo, rows_per_rank = cut[rank], cut[rank + 1] - cut[rank]
    rows = np.arange(lo, lo + rows_per_rank, dtype=np.int64)
    u = np.searchsorted(bounds, rows, side="right").astype(np.int32)
    r = np.random.default_rng([seed, 1000 + rank])
    pi = 1.0 / (np.arange(1, n_items + 1) + item_offset)
    i = iperm[r.choice(n_items, size=rows_per_rank, p=pi / pi.sum())]
    y = np.empty(rows_per_rank)
    step = 2_000_000
    for s0 in range(0, rows_per_rank, step):
        e = min(rows_per_rank, s0 + step)
        y[s0:e] = 3.5 + bu[u[s0:e]] + bi[i[s0:e]] + np.einsum("ij,ij->i", U[u[s0:e]], It[i[s0:e]])
    y += r.normal(size=rows_per_rank) * noise
    y = np.clip(np.round(y * 2) / 2, 0.5, 5.0)
    indptr = np.arange(0, 2 * rows_per_rank + 1, 2, dtype=np.int64)
    indices = np.empty(2 * rows_per_rank, dtype=np.int32)
    indices[0::2] = u
    indices[1::2] = n_users + i
    X = sps.csr_matrix((np.ones(2 * rows_per_rank), indices, indptr), shape=(rows_per_rank, n_users + n_items))
    return X, y, [n_users, n_items], lo, total


def _onehot(idx, n):
    idx = np.asarray(idx)
    return sps.csr_matrix((np.ones(len(idx)), (np.arange(len(idx)), idx)), shape=(len(idx), n))


def _multihot(rng, n_rows, n_cols, mean):
    """multi-hot rows with Poisson(mean) distinct columns, values 1 / sqrt(n) (utils/encoders/multi_value.py:51-82)"""
    rows, cols, vals = [], [], []
    for r in range(n_rows):
        k = min(max(1, rng.poisson(mean)), n_cols)
        c = np.sort(rng.choice(n_cols, size=k, replace=False))
        rows += [r] * k
        cols += list(c)
        vals += [1.0 / np.sqrt(k)] * k
    return sps.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols))


def ml100k_extended_like(n_rows=80000, n_users=943, n_items=1682, seed=0, implicit_user=85, implicit_item=48):
    """BASELINE configs[3] / SURVEY 8d config 4: ML-100k-extended-shaped relation blocks (examples/ml-100k-extended.ipynb
    cells 2-10): main table = one-hot date (212 columns); user block = [id 944 | age bin 10 | occupation 21 | zip 10 |
    implicit movies 1683 multi-hot]; movie block = [id 1683 | year bin 10 | genres 19 multi-hot | implicit users 944
    multi-hot]; maps drawn with the popularity profile of config 2.
    Returns (main csr, [(map, block csr), ...], y, group_shapes)."""
    rng = np.random.default_rng(seed)
    pu = 1.0 / (np.arange(1, n_users + 1) + 30.0)
    pi = 1.0 / (np.arange(1, n_items + 1) + 20.0)
    u = rng.choice(n_users, size=n_rows, p=pu / pu.sum())
    it = rng.choice(n_items, size=n_rows, p=pi / pi.sum())
    date = rng.integers(0, 212, size=n_rows)
    main = _onehot(date, 212)
    ub = sps.hstack([_onehot(np.arange(n_users), n_users + 1), _onehot(rng.integers(0, 10, n_users), 10),
                     _onehot(rng.integers(0, 21, n_users), 21), _onehot(rng.integers(0, 10, n_users), 10),
                     _multihot(rng, n_users, n_items + 1, implicit_user)]).tocsr()
    ib = sps.hstack([_onehot(np.arange(n_items), n_items + 1), _onehot(rng.integers(0, 10, n_items), 10),
                     _multihot(rng, n_items, 19, 2), _multihot(rng, n_items, n_users + 1, implicit_item)]).tocsr()
    shapes = [212, n_users + 1, 10, 21, 10, n_items + 1, n_items + 1, 10, 19, n_users + 1]
    bu, bi = rng.normal(size=n_users) * 0.4, rng.normal(size=n_items) * 0.4
    y = np.clip(np.round(3.5 + bu[u] + bi[it] + rng.normal(size=n_rows)), 1, 5)
    return main, [(u.astype(np.int64), ub), (it.astype(np.int64), ib)], y, shapes


def config5_like(scale=0.01, seed=2, ordered=True):
    """BASELINE configs[4] / SURVEY 8d config 5 at `scale` (1.0: N = 50 M rows, main table = two one-hot fields
    500 000 + 50 000, four relation blocks: user-side 500 000 x 2000 (10 nnz / row), item-side 50 000 x 1000 (10),
    two context blocks 1000 x 200 (5)); targets 0..4 cut at the 20/40/60/80 % quantiles of a latent score (ordered)
    or the latent score itself. Returns (main csr sorted by user, blocks, y, group_shapes)."""
    rng = np.random.default_rng(seed)
    N = int(50_000_000 * scale)
    nu, ni = max(1000, int(500_000 * scale)), max(200, int(50_000 * scale))
    u = np.sort(rng.integers(0, nu, size=N)).astype(np.int32)
    it = rng.integers(0, ni, size=N).astype(np.int32)
    indices = np.empty(2 * N, dtype=np.int32)
    indices[0::2] = u
    indices[1::2] = nu + it
    main = sps.csr_matrix((np.ones(2 * N), indices, np.arange(0, 2 * N + 1, 2, dtype=np.int64)), shape=(N, nu + ni))

    def block(n_rows, n_cols, per_row):
        cols = rng.integers(0, n_cols, size=(n_rows, per_row))
        cols.sort(axis=1)
        keep = np.ones_like(cols, dtype=bool)
        keep[:, 1:] = cols[:, 1:] != cols[:, :-1]  # drop duplicate columns inside a row
        rows = np.repeat(np.arange(n_rows), per_row).reshape(n_rows, per_row)
        return sps.csr_matrix((np.full(keep.sum(), 1.0 / np.sqrt(per_row)), (rows[keep], cols[keep])), shape=(n_rows, n_cols))

    blocks = [(u.astype(np.int64), block(nu, 2000, 10)), (it.astype(np.int64), block(ni, 1000, 10)),
              (rng.integers(0, 1000, size=N), block(1000, 200, 5)), (rng.integers(0, 1000, size=N), block(1000, 200, 5))]
    score = rng.normal(size=N) + 0.5 * np.sin(u * 0.01) + 0.3 * np.cos(it * 0.1)
    if ordered:
        y = np.digitize(score, np.quantile(score[: min(N, 1_000_000)], [0.2, 0.4, 0.6, 0.8])).astype(np.float64)
    else:
        y = score
    shapes = [nu, ni] + [b.shape[1] for _, b in blocks]
    return main, blocks, y, shapes


def tuple_design(n_rows=60000, n_users=3000, n_items=5000, ctx=(50, 37), user_cols=40, item_cols=30, seed=4, third_field=0,
                 with_item_field=True, with_item_block=True, user_block_rows=None, user_max=None, with_user_block=True):
    """A design whose rows are index tuples (the shape of BASELINE configs[4], any size): main table = one-hot user field
    (rows sorted by user) [+ one-hot item field] [+ a small third one-hot field]; relation blocks: user side (mapped by the user
    column, multi-hot), item side (mapped by the item index), one context block per entry of `ctx` (own random maps).
    Returns (main csr, blocks [(map, csr)], y, group_shapes)."""
    rng = np.random.default_rng(seed)
    u = np.sort(rng.integers(0, user_max or n_users, size=n_rows)).astype(np.int32)
    it = rng.integers(0, n_items, size=n_rows).astype(np.int32)
    cols, width, shapes = [u], n_users, [n_users]
    if with_item_field:
        cols.append(width + it)
        width += n_items
        shapes.append(n_items)
    if third_field:
        cols.append(width + rng.integers(0, third_field, size=n_rows).astype(np.int32))
        width += third_field
        shapes.append(third_field)
    W = len(cols)
    indices = np.empty(W * n_rows, dtype=np.int32)
    for p, c in enumerate(cols):
        indices[p::W] = c
    main = sps.csr_matrix((np.ones(W * n_rows), indices, np.arange(0, W * n_rows + 1, W, dtype=np.int64)), shape=(n_rows, width))

    def block(n, n_cols, per_row):
        c = rng.integers(0, n_cols, size=(n, per_row))
        c.sort(axis=1)
        keep = np.ones_like(c, dtype=bool)
        keep[:, 1:] = c[:, 1:] != c[:, :-1]
        rows = np.repeat(np.arange(n), per_row).reshape(n, per_row)
        vals = rng.uniform(0.3, 1.0, size=keep.sum())
        return sps.csr_matrix((vals, (rows[keep], c[keep])), shape=(n, n_cols))

    blocks = [(u.astype(np.int64), block(user_block_rows or n_users, user_cols, 3))] if with_user_block else []
    if with_item_block:
        blocks.append((it.astype(np.int64), block(n_items, item_cols, 3)))
    for n_c in ctx:
        blocks.append((rng.integers(0, n_c, size=n_rows).astype(np.int64), block(n_c, 8, 2)))
    y = rng.normal(size=n_rows) + 0.5 * np.sin(u * 0.01) + 0.3 * np.cos(it * 0.1)
    shapes = shapes + [b.shape[1] for _, b in blocks]
    return main, blocks, y, shapes
