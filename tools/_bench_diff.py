import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dt, log = bench.train_measure(4, 30, 5, 0, 1, dev, None)
print("train_measure before rank_identity:", 1e3 * dt / 30, flush=True)
print(bench.rank_identity(0, 0, dev), flush=True)
dt, log = bench.train_measure(4, 30, 5, 0, 1, dev, None)
print("train_measure after rank_identity:", 1e3 * dt / 30, flush=True)
