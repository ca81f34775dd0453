"""diagnostic: achievable HBM rates of plain torch kernels at feature-map sizes (fill / copy / add), cycling over
several buffers so that the 256 MB Infinity Cache does not absorb the traffic"""
import torch

def t(fn, n=20):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

nb = 1088 * 1920 * 64 * 2
for nbuf in (1, 8):
    xs = [torch.randn(1088, 1920, 64, device="cuda").half() for _ in range(nbuf)]
    ys = [torch.empty_like(x) for x in xs]
    ms = t(lambda i: ys[i % nbuf].fill_(1.0))
    print(f"buffers {nbuf}: fill  {ms*1e3:7.1f} us  {nb/ms/1e6:7.1f} GB/s written")
    ms = t(lambda i: ys[i % nbuf].copy_(xs[i % nbuf]))
    print(f"buffers {nbuf}: copy  {ms*1e3:7.1f} us  {2*nb/ms/1e6:7.1f} GB/s moved")
    ms = t(lambda i: torch.add(xs[i % nbuf], xs[(i + 1) % nbuf], out=ys[i % nbuf]))
    print(f"buffers {nbuf}: add   {ms*1e3:7.1f} us  {3*nb/ms/1e6:7.1f} GB/s moved")
    ms = t(lambda i: xs[i % nbuf].sum())
    print(f"buffers {nbuf}: sum   {ms*1e3:7.1f} us  {nb/ms/1e6:7.1f} GB/s read")
