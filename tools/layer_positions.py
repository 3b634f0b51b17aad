import glob, sys
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
df = pd.read_csv(f).sort_values("Start_Timestamp")
df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
k = df[df["Kernel_Name"].str.contains("k_layer")].reset_index(drop=True)
k["pos"] = k.index % 8
print(k.groupby("pos")["us"].agg(["mean", "min", "count"]).round(2))
e = df[df["Kernel_Name"].str.contains("k_embed_front")]
print("embed_front mean", round(e["us"].mean(), 2))
