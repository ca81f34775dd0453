"""Profiling driver: N steady-state 1080p P-frames after the packing / warm-up frames (run as `python3 tools/steady_frame.py`
under rocprofv3 --kernel-trace; tools/last_frame.py cuts the LAST frame out of the trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
runner = bench.GopRunner(model, bench.make_inputs(1234, dev), enabled_amp=os.environ.get("TDVC_FP32_ISLANDS", "0") != "1")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for _ in range(n):
    runner.step()
torch.cuda.synchronize()
print("done", n)
