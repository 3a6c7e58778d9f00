import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity as fz
from blp_amd import _lib, ops
from oracle import oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
model, D, N, q_head, q_tail, kind, table, q_fixed, q_rel, true_row, csr, by_vector, env, rel_ids = fz.make_case(rng)
print(model, D, N, q_head, q_tail, kind, env, "csr nnz", None if csr is None else len(csr[1]), "table absmax", table.abs().max().item(), flush=True)
for k, v in env.items(): _lib.set_knob(k, v)
for variant in sys.argv[2:] or ["full"]:
    kw = {}
    if variant in ("full", "nocsr"):
        if by_vector: kw["q_true"] = table[true_row].cuda()
        else: kw["true_row"] = true_row.cuda()
    if variant == "full" and csr is not None:
        kw.update(filt_rowptr=torch.from_numpy(csr[0]).cuda(), filt_col=torch.from_numpy(csr[1]).cuda())
    print("variant", variant, flush=True)
    got = ops.rank_all(model, table.cuda(), q_fixed.cuda(), q_rel.cuda(), q_head, **kw)
    torch.cuda.synchronize()
    print("ok", got.sum().item(), flush=True)
