"""profiles/stall_trigger.py — which set-up action is answered by the 65-80 ms hold of the queues some 10-50 ms later
(profiles/r03_stall_*.txt)?  After each candidate action: 25 ms of small back-to-back kernels with the host running ahead,
wall time taken; a trial that takes more than twice the median counts as a stall.  Measurement tooling (torch only)."""
import time

import numpy as np
import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1 << 22, device=dev)
torch.cuda.synchronize()


def burst():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        x.add_(1.0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def big_alloc_free():
    a = torch.empty(1_400_000_000 // 4, device=dev)
    a.zero_()
    torch.cuda.synchronize()
    del a
    torch.cuda.empty_cache()


def pinned_alloc_free():
    a = torch.empty(1 << 18, pin_memory=True)
    a.zero_()
    del a
    torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None


_held = []


def pinned_alloc_only():
    a = torch.empty(1 << 18, pin_memory=True)
    a.zero_()
    _held.append(a)


def pinned_free_only():
    if _held:
        _held.pop()
    torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None


def pageable_h2d():
    a = np.random.rand(1 << 18).astype(np.float32)
    torch.from_numpy(a).to(dev)
    torch.cuda.synchronize()
    del a


def pageable_h2d_big():
    a = np.random.rand(1 << 24).astype(np.float32)
    torch.from_numpy(a).to(dev)
    torch.cuda.synchronize()
    del a


def nothing():
    pass


for _ in range(5):
    burst()
for name, act in (("nothing", nothing), ("hipMalloc + hipFree of 1.4 GB", big_alloc_free), ("pinned host alloc + free", pinned_alloc_free),
                  ("pinned host alloc only", pinned_alloc_only), ("pinned host free only", pinned_free_only),
                  ("pageable H2D copy, 1 MB", pageable_h2d), ("pageable H2D copy, 64 MB", pageable_h2d_big), ("nothing (again)", nothing)):
    t = []
    for _ in range(30):
        act()
        t.append(burst())
    med = float(np.median(t))
    print(f"{name:34s} median {med:7.2f} ms, stalls (> 2 x median): {sum(v > 2 * med for v in t):2d} of {len(t)}; worst {max(t):7.2f} ms", flush=True)
