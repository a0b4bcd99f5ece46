"""Three rocprofv3 --pmc passes of scripts/prof_sq.sh (dirs <tag>_a, _b, _c) -> profiles-style JSON for bench.py's
`roofline.second_bound`: {class: {kernel, insts_*, wave_wait_any_frac, valu_active_frac, lds_active_frac, ...}}.
usage: python scripts/sq_to_json.py <tag prefix (dir without _a)> <kernel substring> <bench kernel class> <out.json> <source note>"""
import glob
import json
import sys

import pandas as pd

prefix, kern, cls, out, note = sys.argv[1:6]


def mean(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains(kern, regex=False)]
    return df.pivot_table(index="Counter_Name", values="Counter_Value", aggfunc="mean")["Counter_Value"].to_dict()


a, b, c = mean(prefix + "_a"), mean(prefix + "_b"), mean(prefix + "_c")
rec = {
    "kernel": kern,
    "insts_valu": b["SQ_INSTS_VALU"], "insts_salu": b["SQ_INSTS_SALU"], "insts_lds": b["SQ_INSTS_LDS"],
    "insts_vmem_rd": b["SQ_INSTS_VMEM_RD"], "insts_vmem_wr": b["SQ_INSTS_VMEM_WR"],
    "wave_cycles": a["SQ_WAVE_CYCLES"], "busy_cycles": a["SQ_BUSY_CYCLES"],
    "wave_wait_any_frac": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
    "valu_active_frac": round(a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], 3),
    "lds_active_frac": round(b["SQ_LDS_IDX_ACTIVE"] / a["SQ_WAVE_CYCLES"], 3) if "SQ_LDS_IDX_ACTIVE" in b else None,
    "lds_bank_conflict_frac_of_active": round(b["SQ_LDS_BANK_CONFLICT"] / max(b.get("SQ_LDS_IDX_ACTIVE", 0), 1), 3),
    "waves": c.get("SQ_WAVES"),
    "source": note,
}
json.dump({cls: rec}, open(out, "w"), indent=1)
print(json.dumps({cls: rec}, indent=1))
