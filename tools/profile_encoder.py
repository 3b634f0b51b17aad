#!/usr/bin/env python3
"""encode_music + set_conditioning for bs=32 x 60 s, three times (for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_encoder.py`)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_model, batch_mel
m = make_model("fp16")
mel = torch.from_numpy(batch_mel(32, 5400)).cuda()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    xp, x = m.encode_music(mel, "cuda:0")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m.set_conditioning(xp, x, [1800] * 32)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"encode_music {1e3*(t1-t0):.2f} ms, set_conditioning {1e3*(t2-t1):.2f} ms")
