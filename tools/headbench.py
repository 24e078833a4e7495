import os, sys, torch
sys.path.insert(0, os.getcwd())
from combat_amd import ops
from combat_amd._lib import lib
bf16 = torch.bfloat16
n, hw, c, classes = 128, 4, 512, 10
feat = torch.randn(n, hw, hw, c, device="cuda").to(bf16)
W = torch.randn(classes, c, device="cuda"); b = torch.randn(classes, device="cuda")
t = torch.randint(0, classes, (n,), device="cuda")
pooled = torch.empty(n, c, device="cuda"); logits = torch.empty(n, classes, device="cuda")
loss = torch.zeros(1, device="cuda"); cor = torch.zeros(2, dtype=torch.int32, device="cuda")
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
full = lambda: ops.head_fwd(feat, W, b, logits, targets=t, pooled=pooled, loss_sum=loss, correct=cor)
noatom = lambda: ops.head_fwd(feat, W, b, logits, targets=t, pooled=pooled)
nopool = lambda: ops.head_fwd(feat, W, b, logits, targets=t, loss_sum=loss, correct=cor)
bare = lambda: ops.head_fwd(feat, W, b, logits)
print("head_fwd b2b us: full %.1f | no atomics %.1f | no pooled store %.1f | logits only %.1f" % (timeit(full), timeit(noatom), timeit(nopool), timeit(bare)))
dl = torch.empty(n, classes, device="cuda"); dfeat = torch.empty_like(feat); dW = torch.zeros_like(W); db = torch.zeros_like(b)
import inspect
print(inspect.signature(ops.head_bwd))
