import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from combat_amd.data import ArrayLoader, synthetic_cifar10
x, y = synthetic_cifar10(25600, 1234, 32, 10)
ld = ArrayLoader(x, y, 128, True)
def t(f, n=1):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("threads", torch.get_num_threads())
print("epoch iterate: %.2f ms per batch" % (t(lambda: [0 for _ in ld]) / len(ld)))
ref = ((ld.x[ld.epoch_order(7)[:128]].float() / 255.0) - 0.5) / 0.5
ld.epoch = 7
got = next(iter(ld))[0]
print("bit-identical to the torch chain:", bool(torch.equal(got, ref)))
idx = torch.randperm(25600)[:128]
X = ld.x
print("x[idx] %.3f ms" % t(lambda: X[idx], 50))
u = X[idx]
print("float/255 chain %.3f ms" % t(lambda: ((u.float() / 255.0) - 0.5) / 0.5, 50))
print("order slice %.3f ms" % t(lambda: idx[0:128], 50))
torch.set_num_threads(1)
print("1 thread: x[idx] %.3f ms, chain %.3f ms" % (t(lambda: X[idx], 50), t(lambda: ((u.float() / 255.0) - 0.5) / 0.5, 50)))
print("epoch iterate (1 thread): %.2f ms per batch" % (t(lambda: [0 for _ in ld]) / len(ld)))
