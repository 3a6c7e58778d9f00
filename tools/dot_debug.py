"""Debug: the streaming kernels' variants against each other on a random problem."""
import sys, torch
sys.path.insert(0, "/root/repo")
from blp_amd import ops, _lib
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
for model, D in (("distmult", 64), ("distmult", 128), ("complex", 128)):
    for N in (64, 70001, 300000):
        for qh, qt in ((2, 2), (4, 3), (1, 0)):
            Q = qh + qt
            table = (torch.randn(N, D, generator=g) * 0.1).to(dev)
            rel = ((torch.rand(9, D, generator=g) - 0.5) * 0.25).to(dev)
            fixed = torch.randint(0, N, (Q,), generator=g).to(dev)
            true = torch.randint(0, N, (Q,), generator=g).to(dev)
            r = torch.randint(0, 9, (Q,), generator=g).to(dev)
            qf, qr = table[fixed].contiguous(), rel[r].contiguous()
            res = {}
            for knob in (2, 5, 0, 3, 4):
                _lib.reset_knobs(); _lib.set_knob("small_kernel", 2)
                _lib.set_knob("stream_kernel", knob)
                res[knob] = ops.rank_all(model, table, qf, qr, qh, true_row=true).cpu()
            ok = {k: bool(torch.equal(v, res[2])) for k, v in res.items()}
            print(model, D, N, qh, qt, ok)
            for k, v in res.items():
                if not ok[k]:
                    print("   knob", k, (v - res[2])[:, 0].tolist(), " ref gt", res[2][:, 0].tolist())
