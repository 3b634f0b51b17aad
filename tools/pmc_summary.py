#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel mean of each PMC counter / kernel-trace durations."""
import glob
import sys

import pandas as pd

d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    df = pd.read_csv(f)
    df = df[~df["Kernel_Name"].str.contains("rocclr|at::native|Cijk", regex=True)]
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 40)
    t = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    t["n"] = df.groupby("k")["Dispatch_Id"].nunique()
    pd.set_option("display.width", 250, "display.max_columns", 30, "display.float_format", lambda v: f"{v:,.0f}")
    print(f)
    print(t)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    df = pd.read_csv(f)
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 40)
    df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
    cols = [c for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size") if c in df.columns]
    g = df.groupby("k").agg(n=("us", "size"), mean_us=("us", "mean"), total_ms=("us", lambda x: x.sum() / 1e3), **{c: (c, "first") for c in cols})
    pd.set_option("display.width", 250, "display.max_columns", 30, "display.float_format", lambda v: f"{v:,.2f}")
    print(f)
    print(g.sort_values("total_ms", ascending=False))
