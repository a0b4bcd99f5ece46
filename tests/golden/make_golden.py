#!/usr/bin/env python
"""Generate the committed golden vectors under tests/golden/ (run from the repo root, in the build container).

What the fixtures are anchored on:
  * faddeeva_erfcx.npz      -- OUTPUTS OF THE REFERENCE ITSELF: /root/reference/cpp_source/Faddeeva.cc compiled in
                               place into oracle/_ref/libfaddeeva_ref.so (oracle/Makefile), called on a fixed grid.
                               erfcx is what the truncated-normal samplers and the ordered-probit likelihood of
                               the reference rest on (util.hpp:15-78, OProbitSampler.hpp:29-60).
  * faddeeva_erfcx_wide.npz -- same source, the large-|x| ranges (continued-fraction branch, overflow threshold).
  * libstdcxx_stream.npz    -- std::mt19937 + libstdc++ normal_distribution / gamma_distribution (the engine and
                               distributions the reference draws from, FMTrainer.hpp:122-125,142-143,164-165),
                               produced by the oracle's extern "C" wrappers around the real libstdc++ classes.
  * chain_*.npz             -- seeded Gibbs chains of the CPU oracle (oracle/myfm_oracle.cpp, "parity unpinned":
                               the reference core needs Eigen, absent from this image, so these vectors pin the
                               HIP path and the oracle against regressions, not against a reference binary).
The reference's own tests hold no numeric golden vectors for this path (tests/ are statistical recovery tests,
SURVEY section 4); their data generators are restated in tests/datasets.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import datasets as ds  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def csr_parts(X):
    X = X.tocsr()
    X.sort_indices()
    return dict(indptr=X.indptr.astype(np.int64), indices=X.indices.astype(np.int32), data=X.data.astype(np.float64),
                shape=np.asarray(X.shape, dtype=np.int64))


def faddeeva():
    R = O.ref_faddeeva()
    if R is None:
        raise SystemExit("oracle/_ref/libfaddeeva_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    x = np.concatenate([np.linspace(-6, 6, 241), np.array([-30.0, -10.0, 8.0, 12.5, 27.0, 50.0, 1e3, 1e6]),
                        np.array([1e-12, -1e-12, 1e-3, 0.49999, 0.5, 2.99999, 3.0, 3.00001])])
    np.savez(os.path.join(OUT, "faddeeva_erfcx.npz"), x=x, erfcx=np.array([R.ref_erfcx(float(v)) for v in x]),
             erfc=np.array([R.ref_erfc(float(v)) for v in x]), erf=np.array([R.ref_erf(float(v)) for v in x]))


def faddeeva_wide():
    """the ranges the first grid is thin on: the continued-fraction branch of the device / oracle erfcx (x >= 3,
    three depth classes, the asymptotic tail) and large negative arguments up to the overflow threshold."""
    R = O.ref_faddeeva()
    if R is None:
        raise SystemExit("oracle/_ref/libfaddeeva_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    x = np.concatenate([np.geomspace(3.0, 1e8, 97), np.array([4.99999, 5.0, 5.00001, 9.99999, 10.0, 10.00001, 4.9e7, 5.1e7]),
                        -np.geomspace(1e-6, 26.6, 64), np.linspace(-26.7, -6.0, 32)])
    np.savez(os.path.join(OUT, "faddeeva_erfcx_wide.npz"), x=x, erfcx=np.array([R.ref_erfcx(float(v)) for v in x]))


def stream():
    X, y = ds.toy()
    t = O.OracleTrainer(X, y, rank=2, seed=1234)
    st, pos = t.rng_state()
    normals = t.rng_sample_normals(4096)
    gam = np.array([t.rng_gamma(s, 1.0) for s in (0.5, 1.0, 2.5, 50.0, 5e6)])
    more = t.rng_sample_normals(16)
    np.savez(os.path.join(OUT, "libstdcxx_stream.npz"), state=np.asarray(st, dtype=np.uint32), position=np.int64(pos),
             normals=normals, gamma_shapes=np.array([0.5, 1.0, 2.5, 50.0, 5e6]), gammas=gam, normals_after=more)


def chain(name, X, y, gi, rank, blocks=(), iters=(1, 2, 5), seed=42, **kw):
    t = O.OracleTrainer(X, y, blocks, rank=rank, group_index=gi, seed=seed, **kw)
    n = X.shape[0]
    out = {"y": np.asarray(y, dtype=np.float64), "group_index": np.asarray(gi, dtype=np.int32), "rank": np.int64(rank),
           "seed": np.int64(seed), "iters": np.asarray(iters, dtype=np.int64)}
    out.update({"X_" + k: v for k, v in csr_parts(X).items()})
    for b, (mp, B) in enumerate(blocks):
        out["map%d" % b] = np.asarray(mp, dtype=np.int64)
        out.update({"B%d_%s" % (b, k): v for k, v in csr_parts(B).items()})
    out["n_blocks"] = np.int64(len(blocks))
    w0, w, V = t.fm()
    out.update(init_w0=np.float64(w0), init_w=w, init_V=V, init_e=t.e(n))
    st, pos = t.rng_state()
    out.update(rng_state=np.asarray(st, dtype=np.uint32), rng_position=np.int64(pos))
    done = 0
    for it in iters:
        while done < it:
            t.step()
            done += 1
        w0, w, V = t.fm()
        h = t.hyper()
        out.update({"it%d_w0" % it: np.float64(w0), "it%d_w" % it: w, "it%d_V" % it: V, "it%d_e" % it: t.e(n),
                    "it%d_alpha" % it: np.float64(h["alpha"]), "it%d_lambda_w" % it: h["lambda_w"], "it%d_mu_w" % it: h["mu_w"],
                    "it%d_lambda_V" % it: h["lambda_V"], "it%d_mu_V" % it: h["mu_V"]})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def main():
    O.build()
    faddeeva()
    stream()
    X, y, shapes = ds.onehot_mf(3000, 40, 25, seed=3, sort_by_user=True)
    chain("chain_onehot_sorted", X, y, ds.group_index_from_shapes(shapes), 4)
    X, y, shapes = ds.onehot_mf(3000, 40, 25, seed=4, sort_by_user=False)
    X = X.copy()
    X.data = np.where(np.arange(X.nnz) % 3 == 0, 0.5, 1.5)
    chain("chain_onehot_values", X, y, ds.group_index_from_shapes(shapes), 3)
    X, y = ds.toy()  # README.md:46-59 (BASELINE configs[0])
    chain("chain_toy", X, y, np.zeros(X.shape[1], dtype=np.int32), 4, iters=(1, 2, 10))
    X, y = ds.middle_data()[:2]
    chain("chain_middle", X, y, np.zeros(X.shape[1], dtype=np.int32), 3, iters=(1, 10, 30))
    main_X, _, blocks, y, shapes = ds.block_design()
    chain("chain_blocks", main_X, y, ds.group_index_from_shapes(shapes), 2, blocks=blocks, fit_w0=False)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    if "--only-wide" in sys.argv:  # added later: leaves the other (already committed) fixtures untouched
        O.build()
        faddeeva_wide()
    else:
        main()
        faddeeva_wide()
