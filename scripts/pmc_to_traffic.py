"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) of a bench.py run into
profiles/<name>_pmc_traffic.{txt,json}: HBM bytes per launch and per kernel class.

Correction (MI355X_MICROARCH.md "HBM"): on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes ->
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024. Calibrated on this box with two kernels of known traffic:
k_shift_e (reads 160 MB of e/q lines, 16 B/lane) and k_reduce_e_partial (8 B/lane of every 16): both report
FETCH_SIZE = 80 MB = 1/2. WRITE_SIZE of k_shift_e (8 of every 16 bytes written) = 160 MB: partial-line stores
cost whole lines, no correction."""
import json
import re
import sys

import pandas as pd

fetch_dir, write_dir, out_prefix = sys.argv[1], sys.argv[2], sys.argv[3]

CLASS = [  # kernel-name pattern -> bench.py kernel class
    (r"k_mf_resident", "sweep_V_resident"),
    (r"k_cs_stream", "block_chain_stream"),
    (r"k_mf_pass", "sweep_V_fused_next"),
    (r"k_mf_draw", "sweep_V_scattered"),
    (r"k_mf_score", "update_e_score"),
    (r"k_tile_apply_next", "sweep_V_fused_next"),
    (r"k_tile_(stats|draw|apply|old)\w*<mfm::PMainV|k_tile_old", "sweep_V_scattered"),
    (r"k_tile_(stats|draw|apply)\w*<mfm::PMainW", "sweep_w_scattered"),
    (r"k_scat_(stats|draw|apply)\w*<mfm::PMainV", "sweep_V_scattered"),
    (r"k_scat_(stats|draw|apply)\w*<mfm::PMainW", "sweep_w_scattered"),
    (r"k_level_heavy<mfm::PMainVq|k_level_heavy<mfm::PMainV", "sweep_V_heavy"),
    (r"k_level_light<mfm::PMainVq|k_level_light<mfm::PMainV", "sweep_V_light"),
    (r"k_long_coop<mfm::PMainV", "sweep_V_coop"),
    (r"k_level_heavy<mfm::PMainW", "sweep_w_heavy"),
    (r"k_level_light<mfm::PMainW", "sweep_w_light"),
    (r"k_qbuild", "qbuild_spmv"),
    (r"k_score", "update_e_score"),
]


def short(n):
    m = re.search(r"k_\w+(<[^>]*>)?", n)
    return m.group(0) if m else n[:40]


def load(d, counter):
    df = pd.read_csv(f"{d}/p_counter_collection.csv")
    df = df[df.Counter_Name == counter].copy()
    df["k"] = df.Kernel_Name.map(short)
    df["us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    return df


f, w = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
gf = f.groupby("k").agg(calls=("Counter_Value", "count"), fetch_kb=("Counter_Value", "mean"), us=("us", "mean"))
gw = w.groupby("k").agg(write_kb=("Counter_Value", "mean"))
g = gf.join(gw, how="outer").fillna(0)
g["hbm_MB_per_launch"] = (2 * g.fetch_kb + g.write_kb) * 1024 / 1e6
g = g[g.k if False else g.index.str.startswith("k_")].sort_values("hbm_MB_per_launch", ascending=False)
classes = {}
for k, row in g.iterrows():
    for pat, cls in CLASS:
        if re.search(pat, k):
            c = classes.setdefault(cls, {"bytes_per_level_launch": 0.0, "kernels": []})
            c["bytes_per_level_launch"] += float(row.hbm_MB_per_launch) * 1e6
            c["kernels"].append(k)
            break
with open(out_prefix + ".txt", "w") as fh:
    fh.write("# HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py config 3\n")
    fh.write("# hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (see scripts/pmc_to_traffic.py for the calibration)\n")
    fh.write(g.to_string() + "\n\n# per kernel class (one 'launch' of a class = one level of one sweep):\n")
    for cls, c in classes.items():
        fh.write("%-20s %10.1f MB  %s\n" % (cls, c["bytes_per_level_launch"] / 1e6, ", ".join(c["kernels"])))
json.dump(classes, open(out_prefix + ".json", "w"), indent=1)
print(open(out_prefix + ".txt").read())
