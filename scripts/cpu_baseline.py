"""CPU leg of bench.py (SURVEY 8d "CPU baseline timing"): the oracle -- the Eigen-free restatement of the reference's
sampler, oracle/myfm_oracle.cpp -- built HERE with -O3 -march=native, ONE thread pinned to core 0 (the reference's
training loop is single-threaded), full update_all iterations of the pickled design for at least `min_seconds`.
Prints one JSON object. usage: cpu_baseline.py <workload.pkl> <min_seconds> <max_iters>"""
import json
import os
import pickle
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    path, min_seconds, max_iters = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    pinned = False
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
        pinned = True
    except (AttributeError, OSError):
        pass
    flags = "-O3 -march=native -std=c++17 -fPIC -ffp-contract=off"
    nat_dir = os.path.join(ROOT, "oracle", "_native")
    nat = os.path.join(nat_dir, "libmyfm_oracle_native.so")
    src = os.path.join(ROOT, "oracle", "myfm_oracle.cpp")
    try:
        os.makedirs(nat_dir, exist_ok=True)
        if not os.path.exists(nat) or os.path.getmtime(nat) < os.path.getmtime(src):
            subprocess.check_call(["g++"] + flags.split() + ["-shared", src, "-o", nat])
        os.environ["MYFM_ORACLE_LIB"] = nat
    except (OSError, subprocess.CalledProcessError):
        flags = "-O3 -std=c++17 -fPIC -ffp-contract=off (prebuilt: no compiler on this box)"
    from oracle import oracle as O

    with open(path, "rb") as fh:
        W = pickle.load(fh)
    task = {"regression": O.REGRESSION, "classification": O.CLASSIFICATION, "ordered": O.ORDERED}[W["task"]]
    t0 = time.perf_counter()
    t = O.OracleTrainer(W["X"], W["y"], W["blocks"], rank=W["rank"], group_index=W["gi"], seed=42, task=task)
    t_setup = time.perf_counter() - t0
    n, c0 = 0, time.perf_counter()
    while n < max_iters and (n < 1 or time.perf_counter() - c0 < min_seconds):
        t.step()
        n += 1
    el = time.perf_counter() - c0
    X = W["X"]
    print(json.dumps({
        "value": round(n / el, 5), "unit": "Gibbs iterations/sec", "cores": 1, "kind": "port",
        "sample": "%d full update_all iterations of the same design (N=%d, nnz=%d, %d relation blocks, rank %d, %s) by the CPU "
                  "oracle (oracle/myfm_oracle.cpp: loop-for-loop restatement of the reference; the reference core itself needs "
                  "Eigen, absent here), 1 thread" % (n, X.shape[0], X.nnz, len(W["blocks"]), W["rank"], W["task"]),
        "seconds": round(el, 2), "iterations": n, "pinned_to_one_core": pinned, "build_flags": flags,
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "oracle_setup_s": round(t_setup, 2)}))


if __name__ == "__main__":
    main()
