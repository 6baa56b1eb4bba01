"""PCIe host-to-device rates on the box: pageable vs pinned source, one big copy vs chunks, and the CPU-side cost of filling a pinned
staging buffer from a pageable array (numpy memcpy, N threads)."""
import time, torch, numpy as np, concurrent.futures as cf
n = 4_000_000_000 // 8
a = np.ones(n)                       # pageable
d = torch.empty(n, dtype=torch.float64, device="cuda")
t = torch.from_numpy(a)
def tm(f, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
print("pageable -> device: %.1f GB/s" % (4.0 / tm(lambda: d.copy_(t))))
p = torch.empty(n, dtype=torch.float64).pin_memory()
print("pinned   -> device: %.1f GB/s" % (4.0 / tm(lambda: d.copy_(p, non_blocking=True))))
pn = p.numpy()
for nt in (1, 4, 8, 16, 32):
    ex = cf.ThreadPoolExecutor(nt)
    edges = np.linspace(0, n, nt + 1).astype(np.int64)
    def fill():
        list(ex.map(lambda i: np.copyto(pn[edges[i]:edges[i + 1]], a[edges[i]:edges[i + 1]]), range(nt)))
    t0 = time.perf_counter(); fill(); t1 = time.perf_counter(); fill(); t2 = time.perf_counter()
    print("host memcpy pageable -> pinned, %2d threads: %.1f GB/s" % (nt, 4.0 / min(t1 - t0, t2 - t1)))
# device -> host
h = torch.empty(25_000_000, dtype=torch.float64).pin_memory(); dd = torch.empty(25_000_000, dtype=torch.float64, device="cuda")
print("device -> pinned (0.2 GB): %.1f GB/s" % (0.2 / tm(lambda: h.copy_(dd, non_blocking=True))))
