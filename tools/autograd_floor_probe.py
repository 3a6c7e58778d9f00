"""Why does a custom autograd node over LEAVES cost ~30 us more per backward() with the engine's worker threads than on the
calling thread, when a stock node (e.sum()) and the same custom node behind one stock node do not?  Variants of a node that
launches nothing; default threading vs calling thread."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blp_amd import ops
dev = torch.device("cuda", 0)
B, K, D = 64, 64, 128
e_leaf = (torch.randn(B, 2, D, device=dev) * 0.4).requires_grad_(True)
r_leaf = (torch.randn(B, 1, D, device=dev) * 0.3).requires_grad_(True)
neg = torch.randint(0, 2 * B, (B, K, 2), device=dev)
glue = ops.torch_glue()
pre_e, pre_r = torch.zeros_like(e_leaf), torch.zeros_like(r_leaf)


class PyFloor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, r):
        ctx.save_for_backward(e, r)
        return torch.empty((), device=e.device)

    @staticmethod
    def backward(ctx, g):
        e, r = ctx.saved_tensors
        return torch.empty_like(e), torch.empty_like(r)


class PyFloorNoSave(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, r):
        ctx.shape = (e.shape, r.shape)
        return torch.empty((), device=e.device)

    @staticmethod
    def backward(ctx, g):
        return torch.empty(ctx.shape[0], device=g.device), torch.empty(ctx.shape[1], device=g.device)


class PyFloorPrealloc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, r):
        return torch.empty((), device=e.device)

    @staticmethod
    def backward(ctx, g):
        return pre_e, pre_r


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


def step(f, keep_grads=False):
    def fn():
        if not keep_grads:
            e_leaf.grad = r_leaf.grad = None
        f().backward()
    return fn


variants = {
    "C++ floor(e, r, neg)": lambda: glue.autograd_floor(e_leaf, r_leaf, neg),
    "C++ floor, e only a leaf (r detached)": lambda: glue.autograd_floor(e_leaf, r_leaf.detach(), neg),
    **{f"C++ floor variant flags={fl}": (lambda fl=fl: glue.autograd_floor_variant(e_leaf, r_leaf, neg, fl)) for fl in (0, 1, 2, 3, 4, 8, 16, 19)},
    "C++ floor, r = r_leaf.view(B, D) (non-leaf)": lambda: glue.autograd_floor(e_leaf, r_leaf.view(B, D), neg),
    "Python floor, saves e, r": lambda: PyFloor.apply(e_leaf, r_leaf),
    "Python floor, saves nothing": lambda: PyFloorNoSave.apply(e_leaf, r_leaf),
    "Python floor, preallocated grads": lambda: PyFloorPrealloc.apply(e_leaf, r_leaf),
    "stock e.sum()": lambda: e_leaf.sum(),
    "stock e.sum() + r.sum()": lambda: e_leaf.sum() + r_leaf.sum(),
    "stock (e * r).sum()": lambda: (e_leaf * r_leaf).sum(),
}
print(f"{'':44s}{'default':>10s}{'calling thr':>13s}{'default, .grad kept':>22s}")
for name, f in variants.items():
    a = wall(step(f))
    with torch.autograd.set_multithreading_enabled(False):
        b = wall(step(f))
    c = wall(step(f, keep_grads=True))
    print(f"{name:44s}{a:8.1f} us{b:10.1f} us{c:18.1f} us")
