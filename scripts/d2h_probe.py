"""pageable vs pinned D2H / H2D rates on the box (decides whether the host compressor should pin its staging buffers)"""
import time
import torch
d = torch.device("cuda:0")
for mb in (1, 22, 256):
    n = mb << 20
    g = torch.empty(n, dtype=torch.uint8, device=d)
    hp = torch.empty(n, dtype=torch.uint8)
    hq = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, h in (("pageable", hp), ("pinned", hq)):
        for dirn in ("d2h", "h2d"):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                if dirn == "d2h":
                    h.copy_(g)
                else:
                    g.copy_(h)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 5
            print(f"{mb:4d} MB {name:8s} {dirn}: {dt * 1e3:8.3f} ms  {n / dt / 1e9:6.1f} GB/s")
