import torch, sys
sys.path.insert(0, "/root/repo")
from blp_amd import ops
B, K, D = 8, 4, 128
ent = (torch.randn(B, 2, D) * 0.4).half().cuda().requires_grad_(True)
rel = (torch.randn(B, 1, D) * 0.3).cuda().requires_grad_(True)
neg = torch.randint(0, 2 * B, (B, K, 2)).cuda()
loss = ops.inbatch_loss("transe", "margin", ent, rel, neg, 0.0)
print(loss, loss.requires_grad, loss.grad_fn)
loss.backward()
print(ent.grad is None, rel.grad is None, ent.is_leaf, rel.is_leaf)
