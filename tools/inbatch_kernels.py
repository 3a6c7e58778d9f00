"""The in-batch loss step's three kernels issued through the raw C-ABI, 300 steps per shape -- meant to run under
    rocprofv3 --kernel-trace --stats -- python tools/inbatch_kernels.py [shape name ...]
so that kernel_stats.csv gives each kernel's average duration (bench.py reports the back-to-back step time)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from blp_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
for name in (sys.argv[1:] or list(bench.INBATCH_SHAPES)):
    c = bench.INBATCH_SHAPES[name]
    B, K, D = c["B"], c["K"], c["D"]
    g = torch.Generator(device=dev).manual_seed(7)
    ent = (torch.randn(B, 2, D, device=dev, generator=g) * 0.4).to(getattr(torch, c["dtype"]))
    rel = (torch.randn(B, D, device=dev, generator=g) * 0.3).contiguous()
    neg_idx = torch.randint(0, 2 * B, (B, K, 2), device=dev, generator=g)
    loss = torch.empty((), device=dev)
    pos, neg = torch.empty(_lib.inbatch_save_floats(_lib.MODEL_IDS[c["model"]], B, K, D), device=dev), torch.empty(B, K, device=dev)
    ticket = torch.zeros(_lib.INBATCH_TICKET_INTS, dtype=torch.int32, device=dev)
    g_ent, g_rel, one = torch.empty_like(ent), torch.empty(B, D, device=dev), torch.ones((), device=dev)
    args = (_lib.MODEL_IDS[c["model"]], _lib.LOSS_IDS[c["loss"]], _lib.DTYPE_NAMES.index(c["dtype"]), 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(300):
        _lib.check(L.blp_inbatch_loss_fwd(*args, ent.data_ptr(), rel.data_ptr(), neg_idx.data_ptr(), B, K, D, c["reg"],
                                            loss.data_ptr(), pos.data_ptr(), neg.data_ptr(), ticket.data_ptr(), 0, stream), "fwd")
        _lib.check(L.blp_inbatch_loss_bwd(*args, ent.data_ptr(), rel.data_ptr(), neg_idx.data_ptr(), B, K, D, c["reg"],
                                            one.data_ptr(), pos.data_ptr(), neg.data_ptr(), g_ent.data_ptr(), g_rel.data_ptr(),
                                            0, stream), "bwd")
    torch.cuda.synchronize()
    print(name, "done", float(loss))
