import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ops
from oracle import ref_port
seed = int(sys.argv[1]); rng = np.random.default_rng(seed); torch.manual_seed(seed)
model = str(rng.choice(["transe", "distmult", "complex", "simple"])); loss_fn = str(rng.choice(["margin", "nll"]))
D = int(rng.choice([64, 128, 256, 320])) if model != "transe" else int(rng.choice([64, 128, 300, 768, 36]))
B, K = int(rng.integers(2, 160)), int(rng.integers(2, 80))
dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
rel_f32 = dtype == torch.float32 or rng.random() < 0.5
reg = float(rng.choice([0.0, 1e-3, 1e-2]))
ent = (torch.randn(B, 2, D) * float(rng.choice([0.1, 0.4, 1.0]))).to(dtype)
rel = torch.randn(B, 1, D) * 0.3
rel = rel if rel_f32 else rel.to(dtype)
neg_idx = torch.randint(0, 2 * B, (B, K, 2))
print(model, loss_fn, B, K, D, dtype, reg)
e_ref, r_ref = ent.float().clone().requires_grad_(True), rel.float().clone().requires_grad_(True)
ref = ref_port.compute_loss(model, loss_fn, e_ref, r_ref, neg_idx, reg); ref.backward()
e, r = ent.cuda().requires_grad_(True), rel.cuda().requires_grad_(True)
loss = ops.inbatch_loss(model, loss_fn, e, r, neg_idx.cuda(), reg); loss.backward()
ge, gr = e.grad.float().cpu(), r.grad.float().cpu()
print("loss", loss.item(), ref.item())
for name, a, b in (("ent", ge, e_ref.grad), ("rel", gr, r_ref.grad)):
    d = (a - b).abs(); i = d.argmax()
    print(name, "max abs diff", d.max().item(), "at", np.unravel_index(i.item(), d.shape), "got", a.flatten()[i].item(), "want", b.flatten()[i].item(), "max |grad|", b.abs().max().item())
h, t = ent[:, :1].float(), torch.randn(B, K, D)
want = ref_port.SCORE_FNS[model](h, t, rel.float()); got = ops.score(model, h.cuda(), t.cuda(), rel.float().cuda()).cpu()
print("score equal", torch.equal(got, want), (got - want).abs().max().item())
