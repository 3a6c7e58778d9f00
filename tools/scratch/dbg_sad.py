import numpy as np, torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from blp_amd import ops
from test_gpu_parity import random_problem
captured = {}
real_empty = torch.empty
def spy(*a, **k):
    t = real_empty(*a, **k)
    if k.get("dtype") == torch.uint8: captured["ws"] = t
    return t
ops.torch.empty = spy
D=128; N=64; qh=256; qt=0
table, q_fixed, q_rel, true_row = random_problem("transe", N, D, qh, qt, seed=31)
got = ops.rank_all("transe", table.cuda(), q_fixed.cuda(), q_rel.cuda(), qh, true_row=true_row.cuda()).cpu().numpy()
torch.cuda.synchronize()
ws = captured["ws"].cpu().numpy()
def al(x): return (x + 255)//256*256
Q=qh+qt
off=0
coef_head=off; off=al(off+qh*2*D*4)
coef_tail=off; off=al(off+qt*2*D*4)
key_true=off; off=al(off+Q*4)
acc=off; off=al(off+Q*8)
acc_f=off; off=al(off+Q*8)
params=off; off=al(off+32)
thr=off; off=al(off+Q*8)
qimg=off; off=al(off+Q*(D//2)*4+64)
cimg=off; off=al(off+((N+63)//64)*64*(D//2)*4)
P = ws[params:params+32].view(np.int32)
print("params lo_ord %d hi_ord %d maxabs %g nonfinite %d n_pairs %d" % (P[0], P[1], ws[params+8:params+12].view(np.float32)[0], P[3], P[4]))
def ord2f(o): 
    o = np.int32(o); b = o if o >= 0 else np.int32(o ^ 0x7fffffff)
    return np.array([b], np.int32).view(np.float32)[0]
lo = ord2f(P[0]); hi = ord2f(P[1]); print("lo", lo, "hi", hi)
t=table.numpy(); f=q_fixed.numpy(); r=q_rel.numpy()
head=(np.arange(Q)<qh)[:,None]
c=np.where(head, f-r, f+r).astype(np.float32)
print("expect lo", min(t.min(), c.min()), "hi", max(t.max(), c.max()))
s=np.float32(np.float32(65535)/np.float32(hi-lo))
def quant(x): return np.clip(np.rint(((x-lo).astype(np.float32))*s), 0, 65535).astype(np.int64)
tq=quant(t); cq=quant(c)
QI = ws[qimg:qimg+Q*(D//2)*4].view(np.uint32).reshape(Q, D//2)
qd = np.stack([QI & 0xffff, QI >> 16], -1).reshape(Q, D).astype(np.int64)
print("qimg matches:", np.array_equal(qd, cq), np.abs(qd-cq).max())
CI = ws[cimg:cimg+64*(D//2)*4].view(np.uint32).reshape(D//8, 64, 4)   # [j4][lane][4]
cd = CI.transpose(1,0,2).reshape(64, D//2)
cdd = np.stack([cd & 0xffff, cd >> 16], -1).reshape(64, D).astype(np.int64)
print("cimg matches:", np.array_equal(cdd[:N], tq), np.abs(cdd[:N]-tq).max())
TH = ws[thr:thr+Q*8].view(np.uint32).reshape(Q,2)
KT = ws[key_true:key_true+Q*4].view(np.float32)
sad = np.abs(cq[:,None,:]-tq[None,:,:]).sum(-1)
print("thr[0:4]", TH[:4].tolist(), "s*dt", (s*-KT[:4]).tolist(), "sad true", sad[np.arange(4), true_row.numpy()[:4]].tolist())
above = (sad < TH[:,0:1]).sum(1); print("expected decided-above", above[:8].tolist(), "got", got[:8,0].tolist())
