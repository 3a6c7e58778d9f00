"""What the forward launch of the in-batch loss spends where: the same call with parts switched off (hooks build, knob
`inbatch_probe`; the results are then WRONG -- timing only), back-to-back launches timed with device events.
    python tools/inbatch_probe.py [shape ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from blp_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or list(bench.INBATCH_SHAPES)):
    c = bench.INBATCH_SHAPES[name]
    B, K, D = c["B"], c["K"], c["D"]
    g = torch.Generator(device=dev).manual_seed(7)
    ent = (torch.randn(B, 2, D, device=dev, generator=g) * 0.4).to(getattr(torch, c["dtype"]))
    rel = (torch.randn(B, D, device=dev, generator=g) * 0.3).contiguous()
    neg_idx = torch.randint(0, 2 * B, (B, K, 2), device=dev, generator=g)
    loss = torch.empty((), device=dev)
    pos = torch.empty(_lib.inbatch_save_floats(_lib.MODEL_IDS[c["model"]], B, K, D), device=dev)
    neg = torch.empty(B, K, device=dev)
    ticket = torch.zeros(_lib.INBATCH_TICKET_INTS, dtype=torch.int32, device=dev)
    g_ent, g_rel, one = torch.empty_like(ent), torch.empty(B, D, device=dev), torch.ones((), device=dev)
    args = (_lib.MODEL_IDS[c["model"]], _lib.LOSS_IDS[c["loss"]], _lib.DTYPE_NAMES.index(c["dtype"]), 0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for probe in (0, 1, 2, 4, 3, 7, 8):
        if probe != 8:
            _lib.set_knob("inbatch_probe", probe)
        L = _lib.lib()

        def fwd():
            _lib.check(L.blp_inbatch_loss_fwd(*args, ent.data_ptr(), rel.data_ptr(), neg_idx.data_ptr(), B, K, D, c["reg"],
                                                loss.data_ptr(), pos.data_ptr(), neg.data_ptr(), ticket.data_ptr(), 0, stream), "fwd")

        def bwd():
            _lib.check(L.blp_inbatch_loss_bwd(*args, ent.data_ptr(), rel.data_ptr(), neg_idx.data_ptr(), B, K, D, c["reg"],
                                                one.data_ptr(), pos.data_ptr(), neg.data_ptr(), g_ent.data_ptr(), g_rel.data_ptr(), 0, stream), "bwd")

        fn = bwd if probe == 8 else fwd
        if probe == 8:
            fwd()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 300
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ticket.zero_()
        out[{0: "full", 1: "no index", 2: "no tickets", 4: "no positives", 3: "no index, no tickets", 7: "scores only", 8: "backward"}[probe]] = round(a.elapsed_time(b) / n * 1e3, 2)
    _lib.reset_knobs()
    for shares in (1, 2, 4, 8, 16):  # the backward with S waves per entity row
        _lib.set_knob("inbatch_shares", shares)
        L = _lib.lib()
        fwd()
        for _ in range(20):
            bwd()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(300):
            bwd()
        b.record()
        torch.cuda.synchronize()
        out[f"backward S={shares}"] = round(a.elapsed_time(b) / 300 * 1e3, 2)
    _lib.reset_knobs()
    print(name, out, flush=True)
