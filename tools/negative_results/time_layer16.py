"""Per-block time of the layer kernels through the stop-after test hook: run under
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl16 -- python3 tools/time_layer16.py [16|32]
then tools/time_layer16.py --report gpurun_out/tl16 prints the mean duration per stop point (bs=32, T=1800, layer 0)."""
import os, sys, glob, csv
if len(sys.argv) > 1 and sys.argv[1] == "--report":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "k_layer" in r["Kernel_Name"][:16]:
                    rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:24]))
    rows.sort()
    REP = 10
    names = ["after sa", "after ca", "after ffn", "whole layer"]
    prev = 0.0
    for i, n in enumerate(names):
        d = sorted(x[1] for x in rows[i * REP:(i + 1) * REP])
        med = d[len(d) // 2]
        print(f"{n:12s} {med:8.1f} us   (+{med - prev:6.1f})   {rows[i * REP][2]}")
        prev = med
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from helpers import xf_pair, batch_noise, make_model
if len(sys.argv) > 1 and sys.argv[1] == "16":
    os.environ["DC_LAYER16"] = "1"
B, T = 32, 1800
m = make_model("fp16")
xfp, xfo = xf_pair(B, T)
x = torch.from_numpy(batch_noise(B, T)).cuda()
t = np.array([(7 * b + 3) % 50 for b in range(B)])
nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
for stage in (1, 2, 3, 0):
    for _ in range(10):
        nat.debug_denoise(x, t, 1, stage)
    torch.cuda.synchronize()
