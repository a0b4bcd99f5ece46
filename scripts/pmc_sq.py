"""Summarise a rocprofv3 --pmc run (csv) per kernel: mean counter values per launch.
usage: python scripts/pmc_sq.py <dir with *_counter_collection.csv> [kernel-name substring]"""
import glob
import sys

import pandas as pd

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df["k"] = df.Kernel_Name.str.extract(r"(k_\w+)")[0]
if pat:
    df = df[df.Kernel_Name.str.contains(pat, regex=False)]
t = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
t["launches"] = df.groupby("k").Dispatch_Id.nunique()
pd.set_option("display.width", 250)
pd.set_option("display.max_columns", 50)
import os
keep = os.environ.get("SQ_KERNEL")
if keep:
    t = t[[keep in str(i) for i in t.index]]
print(t.sort_values(t.columns[0], ascending=False).head(12).to_string())
