"""Where the microseconds of `compute_loss(...).backward()` go, B = 64, K = 64, D = 128, TransE / margin:
  standalone   leaves -> fused loss -> backward(), the engine's default threading and on the calling thread
               (torch.autograd.set_multithreading_enabled(False): a THREAD-LOCAL switch, checked below)
  floor        the same with a node that launches nothing (what PyTorch's engine costs per backward() on this host)
  in a graph   e = leaf * 1 (one stock node upstream, as an encoder would be): the engine already runs on the device's worker
               thread for the stock node; what our node ADDS is `fused - floor` there
  DataParallel two replicas on device 0 (the reference's training wrapper, train.py:329-330,344): per step, fused loss against
               the reference's expressions through stock PyTorch-ROCm
    python tools/loss_step_probe.py"""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blp_amd import models, ops
from oracle import ref_port

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
B, K, D = 64, 64, 128
ent = torch.randn(B, 2, D, device=dev, generator=g) * 0.4
rel = torch.randn(B, 1, D, device=dev, generator=g) * 0.3
neg_idx = torch.randint(0, 2 * B, (B, K, 2), device=dev, generator=g)
e_leaf, r_leaf = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
glue = ops.torch_glue()


def wall(fn, n=400):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


def step(loss_of, upstream):
    def fn():
        e_leaf.grad = r_leaf.grad = None
        e = e_leaf * 1.0 if upstream else e_leaf
        loss_of(e).backward()
    return fn


fused = lambda e: ops.inbatch_loss("transe", "margin", e, r_leaf, neg_idx, 0.0)
floor = lambda e: glue.autograd_floor(e, r_leaf, neg_idx)
stock = lambda e: ref_port.compute_loss("transe", "margin", e, r_leaf, neg_idx, 0.0)
trivial = lambda e: e.sum()

print(f"{'':34s}{'default threading':>20s}{'calling thread':>18s}")
for label, upstream in (("standalone (leaves)", False), ("in a graph (leaf * 1 upstream)", True)):
    rows = {}
    for name, f in (("fused loss", fused), ("floor (node without kernels)", floor), ("stock expressions", stock), ("e.sum() (one stock node)", trivial)):
        a = wall(step(f, upstream))
        with torch.autograd.set_multithreading_enabled(False):
            b = wall(step(f, upstream))
        rows[name] = (a, b)
        print(f"{label:34s}{name:30s}{a:10.1f} us{b:14.1f} us")
    print(f"{label:34s}{'fused - floor (what our node adds)':30s}{rows['fused loss'][0] - rows['floor (node without kernels)'][0]:10.1f} us"
          f"{rows['fused loss'][1] - rows['floor (node without kernels)'][1]:14.1f} us")

# the switch is thread-local: another thread still sees the default
seen = []
with torch.autograd.set_multithreading_enabled(False):
    t = threading.Thread(target=lambda: seen.append(torch.autograd.is_multithreading_enabled()))
    t.start(); t.join()
    here = torch.autograd.is_multithreading_enabled()
print(f"set_multithreading_enabled(False) inside this thread: {here}; seen from another thread meanwhile: {seen[0]} (thread-local)")

# nn.DataParallel, two replicas on device 0
E, R = 2000, 237


class StockLP(models.TransductiveLinkPrediction):
    def compute_loss(self, ent_embs, rels, neg_idx):
        return ref_port.compute_loss(self.rel_model, "margin", ent_embs, self.rel_emb(rels), neg_idx, self.regularizer)


pairs = torch.randint(0, E, (2 * B, 2), device=dev, generator=g)
rels = torch.randint(0, R, (2 * B, 1), device=dev, generator=g)
negs = torch.cat([torch.randint(0, 2 * B, (B, K, 2), device=dev, generator=g) for _ in range(2)])
for label, cls in (("fused loss", models.TransductiveLinkPrediction), ("stock expressions", StockLP)):
    net = cls(D, "transe", "margin", E, R, 0).to(dev)
    dp = torch.nn.DataParallel(net, device_ids=[0, 0])

    def dp_step():
        net.zero_grad(set_to_none=True)
        dp(pairs, rels, negs).mean().backward()

    def one_step():
        net.zero_grad(set_to_none=True)
        net(pairs[:B], rels[:B], negs[:B]).backward()

    a, b = wall(dp_step, 100), wall(one_step, 200)
    with torch.autograd.set_multithreading_enabled(False):
        c = wall(one_step, 200)
    print(f"TransductiveLinkPrediction, {label:18s}: DataParallel x2 on device 0 {a:8.1f} us per step (2 x {B} triples); "
          f"one replica alone {b:7.1f} us, on the calling thread {c:7.1f} us")
