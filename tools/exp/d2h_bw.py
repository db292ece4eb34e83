# Experiment: what D2H bandwidth does this box give for the 82.5 MB score matrix?  (pinned host memory)
#   one copy / 4 row blocks on one stream / 4 blocks on 2 or 4 streams; HSA_ENABLE_SDMA=0 (blit kernels) via the environment
import os, sys, time, torch
n = 4541
dev = torch.device("cuda")
src = torch.rand(n, n, device=dev)
dst = torch.empty(n, n).pin_memory()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
mb = n * n * 4 / 1e6
def one(): dst.copy_(src, non_blocking=True)
def blocks(k, ns):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def f():
        for q in range(k):
            r0, r1 = n * q // k, n * (q + 1) // k
            with torch.cuda.stream(streams[q % ns]):
                dst[r0:r1].copy_(src[r0:r1], non_blocking=True)
    return f
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
t = timeit(one); print("one copy            %.3f ms  %.1f GB/s" % (t * 1e3, mb / t / 1e3))
for k, ns in ((4, 1), (4, 2), (4, 4), (8, 4), (16, 4)):
    t = timeit(blocks(k, ns)); print("%2d blocks %d streams  %.3f ms  %.1f GB/s" % (k, ns, t * 1e3, mb / t / 1e3))
