"""Torch-CPU restatement ("port") of the reference's scoring / loss / metric expressions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Two uses:
  * oracle for the floating-point in-batch loss and its gradients (autograd of these expressions
    is what the reference differentiates), compared with a tolerance;
  * bench.py's ``cpu_baseline`` (kind "port"): the same torch CPU kernels the reference runs.

It is written expression-for-expression so the torch CPU kernels -- and therefore the reduction
order -- are the ones the reference hits; tests/test_oracle_golden.py checks it bit-for-bit against
vectors produced by the imported reference.

Reference lines restated:
  score functions          /root/reference/models.py:222-248
  margin / nll / l2 reg    /root/reference/models.py:251-266
  compute_loss             /root/reference/models.py:51-70
  get_metrics              /root/reference/utils.py:86-111
  eval scoring block       /root/reference/train.py:141-171
"""
import torch
import torch.nn.functional as F


def _halves(x):
    half = x.shape[-1] // 2
    return x[..., :half], x[..., half:]


def transe(h, t, r):
    # vector_norm(ord=1) is the kernel torch.norm(p=1) dispatches to: sequential f32 sum on CPU.
    return torch.linalg.vector_norm(h + r - t, ord=1, dim=-1).neg()


def distmult(h, t, r):
    return (h * r * t).sum(dim=-1)


def complex_(h, t, r):
    h_re, h_im = _halves(h)
    t_re, t_im = _halves(t)
    r_re, r_im = _halves(r)
    terms = r_re * h_re * t_re + r_re * h_im * t_im + r_im * h_re * t_im - r_im * h_im * t_re
    return terms.sum(dim=-1)


def simple(h, t, r):
    h_head, h_tail = _halves(h)
    t_head, t_tail = _halves(t)
    r_fwd, r_inv = _halves(r)
    return (h_head * r_fwd * t_tail + t_head * r_inv * h_tail).sum(dim=-1) / 2


SCORE_FNS = {"transe": transe, "distmult": distmult, "complex": complex_, "simple": simple}


def margin(pos, neg):
    # In-place masking, not relu: the gradient still flows where the hinge is exactly 0.
    hinge = 1 - pos + neg
    hinge[hinge < 0] = 0
    return hinge.mean()


def nll(pos, neg):
    return (F.softplus(-pos).mean() + F.softplus(neg).mean()) / 2


LOSS_FNS = {"margin": margin, "nll": nll}


def l2_reg(*tensors):
    total = 0.0
    for x in tensors:
        total = total + (x ** 2).mean()
    return total / 3.0


def compute_loss(rel_model, loss_fn, ent_embs, rel_vecs, neg_idx, regularizer=0.0):
    """ent_embs (B, 2, D); rel_vecs (B, 1, D) already gathered from rel_emb; neg_idx (B, K, 2)."""
    score = SCORE_FNS[rel_model]
    batch = ent_embs.shape[0]
    pos_h, pos_t = ent_embs[:, 0:1], ent_embs[:, 1:2]
    pos = score(pos_h, pos_t, rel_vecs)
    reg = regularizer * l2_reg(pos_h, pos_t, rel_vecs) if regularizer > 0 else 0
    gathered = ent_embs.reshape(batch * 2, -1)[neg_idx]            # (B, K, 2, D)
    neg = score(gathered[:, :, 0], gathered[:, :, 1], rel_vecs)    # (B, K)
    return LOSS_FNS[loss_fn](pos, neg) + reg


def rank_metrics(pred, true_idx, k_values=(1, 3, 10)):
    """pred (Q, N), true_idx (Q, 1) -> gt, ge (int64 (Q,)), rr (Q,) f32, hits (Q, nk) bool."""
    true = pred.gather(1, true_idx)
    gt = (pred > true).sum(dim=1)
    ge = (pred >= true).sum(dim=1)
    avg = (gt + 1 + ge).float() * 0.5
    k = torch.tensor([list(k_values)], dtype=torch.float32, device=pred.device)
    return gt, ge, avg.reciprocal(), avg.unsqueeze(1) <= k


def eval_batch(rel_model, table, heads, tails, rel_vecs, filter_mask=None):
    """One reference eval batch (train.py:141-171) on CPU.

    table (N, D); heads, tails (B,) row indices; rel_vecs (B, D); filter_mask (2B, N) bool or None.
    Returns dict with raw and (if mask given) filtered gt / ge / rr / hits, head queries first.
    """
    score = SCORE_FNS[rel_model]
    ent = table.unsqueeze(0)
    h = table[heads].unsqueeze(1)
    t = table[tails].unsqueeze(1)
    r = rel_vecs.unsqueeze(1)
    pred = torch.cat((score(ent, t, r), score(h, ent, r)))
    true_idx = torch.cat((heads, tails)).unsqueeze(1)
    out = {}
    out["gt"], out["ge"], out["rr"], out["hits"] = rank_metrics(pred, true_idx)
    if filter_mask is not None:
        pred = pred.clone()
        pred[filter_mask] = pred.min() - 1.0
        out["gt_filt"], out["ge_filt"], out["rr_filt"], out["hits_filt"] = rank_metrics(pred, true_idx)
    return out


def neg_idx_from_draws(draw, which):
    """The reference sampler's index construction (data.py:35-81) as plain loops over integer draws -- the checker of
    blp_amd.data.negative_indices_from_draws.  Slots are numbered row-wise ([[0, 1], [2, 3], ...], data.py:52-53); the candidates
    of row b are every slot except its own two (the (B, 2B) weight matrix with the pair zeroed, data.py:57-60), so draw d
    names the d-th slot of [0 .. 2b - 1, 2b + 2 .. 2B - 1]; column `which` of the pair is replaced by it (data.py:63-67)."""
    b, k = len(draw), len(draw[0])
    out = []
    for row in range(b):
        candidates = [s for s in range(2 * b) if s // 2 != row]
        rows = []
        for j in range(k):
            pair = [2 * row, 2 * row + 1]
            pair[int(which[row][j])] = candidates[int(draw[row][j])]
            rows.append(pair)
        out.append(rows)
    return torch.tensor(out, dtype=torch.long).reshape(b, k, 2)
