"""Link-prediction models with the reference's interface (/root/reference/models.py), MI355X-native
underneath: on a HIP device the relational scoring, the in-batch-negatives loss and the all-entities
ranking run in hand-written gfx950 kernels through libblp_hip.so (blp_amd.ops); the text encoders
stay stock PyTorch-ROCm modules.

Same constructor signatures, attribute names and state_dict keys as the reference, so checkpoints and
calling code carry over:
    LinkPrediction(dim, rel_model, loss_fn, num_relations, regularizer)          models.py:7-70
    InductiveLinkPrediction.forward(text_tok, text_mask, rels=None, neg_idx=None) models.py:73-93
    BertEmbeddingsLP / WordEmbeddingsLP / BOW / DKRL / TransductiveLinkPrediction models.py:96-219
    transe_score, distmult_score, complex_score, simple_score                    models.py:222-248
    margin_loss, nll_loss, l2_regularization                                     models.py:251-266

Device dispatch is explicit, never a silent fallback: tensors on a HIP device always go through the
HIP library (and raise if it is missing); CPU tensors (the reference's CPU-runnable UMLS smoke
configuration) use plain torch expressions with the reference's evaluation order.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

REL_MODELS = ("transe", "distmult", "complex", "simple")


def _halves(x):
    half = x.shape[-1] // 2
    return x[..., :half], x[..., half:]


def _cpu_transe(h, t, r):
    return -torch.norm(h + r - t, dim=-1, p=1)


def _cpu_distmult(h, t, r):
    return torch.sum(h * r * t, dim=-1)


def _cpu_complex(h, t, r):
    h_re, h_im = _halves(h)
    t_re, t_im = _halves(t)
    r_re, r_im = _halves(r)
    return torch.sum(r_re * h_re * t_re + r_re * h_im * t_im + r_im * h_re * t_im - r_im * h_im * t_re, dim=-1)


def _cpu_simple(h, t, r):
    h_head, h_tail = _halves(h)
    t_head, t_tail = _halves(t)
    r_fwd, r_inv = _halves(r)
    return torch.sum(h_head * r_fwd * t_tail + t_head * r_inv * h_tail, dim=-1) / 2


_CPU_SCORE = {"transe": _cpu_transe, "distmult": _cpu_distmult, "complex": _cpu_complex, "simple": _cpu_simple}


class ScoreFn:
    """score_fn(heads, tails, rels): broadcasts over leading dims, reduces the last (models.py:222-248).

    Callable exactly like the reference's module-level functions; ``rel_model`` names the relational
    model so the fused paths (blp_amd.ranking, LinkPrediction.compute_loss) can pick their kernels.
    """

    def __init__(self, rel_model):
        if rel_model not in REL_MODELS:
            raise ValueError(f"Unknown relational model {rel_model}.")
        self.rel_model = rel_model
        self.__name__ = f"{rel_model}_score"

    def __call__(self, heads, tails, rels):
        if heads.is_cuda or tails.is_cuda or rels.is_cuda:
            return ops.score(self.rel_model, heads, tails, rels)
        return _CPU_SCORE[self.rel_model](heads, tails, rels)

    def __repr__(self):
        return f"<blp_amd {self.__name__}>"


transe_score = ScoreFn("transe")
distmult_score = ScoreFn("distmult")
complex_score = ScoreFn("complex")
simple_score = ScoreFn("simple")


def margin_loss(pos_scores, neg_scores):
    # masked in place like the reference: the gradient still passes where the hinge is exactly 0
    loss = 1 - pos_scores + neg_scores
    loss[loss < 0] = 0
    return loss.mean()


def nll_loss(pos_scores, neg_scores):
    return (F.softplus(-pos_scores).mean() + F.softplus(neg_scores).mean()) / 2


margin_loss.loss_name = "margin"
nll_loss.loss_name = "nll"


def l2_regularization(heads, tails, rels):
    reg_loss = 0.0
    for tensor in (heads, tails, rels):
        reg_loss = reg_loss + torch.mean(tensor ** 2)
    return reg_loss / 3.0


class LinkPrediction(nn.Module):
    """Relation lookup table + score function + loss (reference models.py:7-70)."""

    def __init__(self, dim, rel_model, loss_fn, num_relations, regularizer):
        super().__init__()
        self.dim = dim
        self.normalize_embs = False
        self.regularizer = regularizer

        if rel_model == "transe":
            self.score_fn = transe_score
            self.normalize_embs = True
        elif rel_model == "distmult":
            self.score_fn = distmult_score
        elif rel_model == "complex":
            self.score_fn = complex_score
        elif rel_model == "simple":
            self.score_fn = simple_score
        else:
            raise ValueError(f"Unknown relational model {rel_model}.")
        self.rel_model = rel_model

        self.rel_emb = nn.Embedding(num_relations, self.dim)
        nn.init.xavier_uniform_(self.rel_emb.weight.data)

        if loss_fn == "margin":
            self.loss_fn = margin_loss
        elif loss_fn == "nll":
            self.loss_fn = nll_loss
        else:
            raise ValueError(f"Unkown loss function {loss_fn}")
        self.loss_name = loss_fn

    def encode(self, *args, **kwargs):
        return self._finish(self._encode_entity(*args, **kwargs))

    def _finish(self, ent_emb):
        if self.normalize_embs:
            ent_emb = F.normalize(ent_emb, dim=-1)
        return ent_emb

    def encode_into(self, out, *args, **kwargs):
        """Encode straight into rows ``out`` of an entity table (train.py:109-113); subclasses may fuse the last steps."""
        return out.copy_(self.encode(*args, **kwargs))

    def _encode_entity(self, *args, **kwargs):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def compute_loss(self, ent_embs, rels, neg_idx):
        """ent_embs (B, 2, D), rels (B, 1) relation ids, neg_idx (B, K, 2) rows of ent_embs.view(2B, D).

        On a HIP device this is one fused forward (+ one fused, deterministic backward) instead of the
        reference's gather temporary and ~40 small kernels; on CPU it is the reference's expression.
        """
        rel_vecs = self.rel_emb(rels)
        fused_dtype = ent_embs.dtype in (torch.float32, torch.float16, torch.bfloat16)
        if ent_embs.is_cuda and fused_dtype and ent_embs.dim() == 3 and neg_idx.dim() == 3:
            if rel_vecs.dtype not in (ent_embs.dtype, torch.float32):
                rel_vecs = rel_vecs.float()
            return ops.inbatch_loss(self.rel_model, self.loss_name, ent_embs, rel_vecs, neg_idx, self.regularizer)
        return self._compute_loss_torch(ent_embs, rel_vecs, neg_idx)

    def _compute_loss_torch(self, ent_embs, rel_vecs, neg_idx):
        batch_size = ent_embs.shape[0]
        heads, tails = torch.chunk(ent_embs, chunks=2, dim=1)
        pos_scores = self.score_fn(heads, tails, rel_vecs)
        reg_loss = self.regularizer * l2_regularization(heads, tails, rel_vecs) if self.regularizer > 0 else 0
        neg_embs = ent_embs.view(batch_size * 2, -1)[neg_idx]
        heads, tails = torch.chunk(neg_embs, chunks=2, dim=2)
        neg_scores = self.score_fn(heads.squeeze(2), tails.squeeze(2), rel_vecs)
        return self.loss_fn(pos_scores, neg_scores) + reg_loss


class InductiveLinkPrediction(LinkPrediction):
    """Description-based link prediction: entities are encoded from their text (models.py:73-93)."""

    def _encode_entity(self, text_tok, text_mask):
        raise NotImplementedError

    def forward(self, text_tok, text_mask, rels=None, neg_idx=None):
        batch_size, _, num_text_tokens = text_tok.shape
        ent_embs = self.encode(text_tok.view(-1, num_text_tokens), text_mask.view(-1, num_text_tokens))
        if rels is None and neg_idx is None:
            return ent_embs  # embeddings only (entity-table build, train.py:109)
        return self.compute_loss(ent_embs.view(batch_size, 2, -1), rels, neg_idx)


def _load_bert(encoder_name, **kwargs):
    """BertModel.from_pretrained(encoder_name) as the reference does; when the weights cannot be
    fetched (no network) and ``encoder_name`` is a BertConfig or a dict of config fields, build the
    architecture with random weights instead (benchmarks / tests use that path)."""
    from transformers import BertConfig, BertModel
    if isinstance(encoder_name, BertConfig):
        return BertModel(encoder_name)
    if isinstance(encoder_name, dict):
        return BertModel(BertConfig(**encoder_name))
    return BertModel.from_pretrained(encoder_name, **kwargs)


class BertEmbeddingsLP(InductiveLinkPrediction):
    """BERT for Link Prediction (BLP): [CLS] representation -> bias-free linear (models.py:96-111)."""

    def __init__(self, dim, rel_model, loss_fn, num_relations, encoder_name, regularizer):
        super().__init__(dim, rel_model, loss_fn, num_relations, regularizer)
        self.encoder = _load_bert(encoder_name, output_attentions=False, output_hidden_states=False)
        hidden_size = self.encoder.config.hidden_size
        self.enc_linear = nn.Linear(hidden_size, self.dim, bias=False)

    def _encode_entity(self, text_tok, text_mask):
        embs = self.encoder(text_tok, text_mask)[0][:, 0]
        return self.enc_linear(embs)

    def encode_into(self, out, text_tok, text_mask):
        """Table build (train.py:96-121): on a HIP device and outside autograd, enc_linear + F.normalize + the row
        assignment are one kernel that reads the [CLS] rows in place and writes the table rows (ops.project_rows)."""
        if out.is_cuda and not torch.is_grad_enabled() and out.dtype == torch.float32 and \
                self.enc_linear.weight.dtype == torch.float32 and \
                ops.project_rows_supported(self.enc_linear.in_features, self.dim):
            cls = self.encoder(text_tok, text_mask)[0][:, 0]
            if cls.dtype == torch.float32:
                return ops.project_rows(cls, self.enc_linear.weight, out, self.normalize_embs)
            return out.copy_(self._finish(self.enc_linear(cls)))
        return super().encode_into(out, text_tok, text_mask)


class WordEmbeddingsLP(InductiveLinkPrediction):
    """Description encoder over a word-embedding table taken from BERT or a tensor file
    (models.py:114-137).  ``embeddings`` may also be a tensor (in-memory table)."""

    def __init__(self, rel_model, loss_fn, num_relations, regularizer, dim=None, encoder_name=None,
                 embeddings=None):
        if not encoder_name and embeddings is None:
            raise ValueError("Must provided one of encoder_name or embeddings")
        if encoder_name is not None:
            embeddings = _load_bert(encoder_name).embeddings.word_embeddings
        else:
            emb_tensor = embeddings if isinstance(embeddings, torch.Tensor) else torch.load(embeddings)
            num_embeddings, embedding_dim = emb_tensor.shape
            embeddings = nn.Embedding(num_embeddings, embedding_dim)
            embeddings.weight.data = emb_tensor
        if dim is None:
            dim = embeddings.embedding_dim
        super().__init__(dim, rel_model, loss_fn, num_relations, regularizer)
        self.embeddings = embeddings

    def _encode_entity(self, text_tok, text_mask):
        raise NotImplementedError


class BOW(WordEmbeddingsLP):
    """Bag of words: mean of the description's word embeddings (models.py:140-155)."""

    def _encode_entity(self, text_tok, text_mask=None):
        if text_mask is None:
            text_mask = torch.ones_like(text_tok, dtype=torch.float)
        embs = self.embeddings(text_tok)
        lengths = torch.sum(text_mask, dim=-1, keepdim=True)
        embs = torch.sum(text_mask.unsqueeze(dim=-1) * embs, dim=1)
        return embs / lengths

    def encode_into(self, out, text_tok, text_mask=None, defer_check=False):
        """Table build (train.py:96-121): on a HIP device and outside autograd the lookup, the masked mean, F.normalize and
        the row assignment are one kernel that reads every gathered word vector once (ops.bow_rows -> blp_bow_rows); the
        stock modules stream the (B, L, E) gather three times.  A token id outside the embedding table raises IndexError
        before this returns (one host read) -- unless ``defer_check``: the caller then calls check_tokens() itself after
        its last chunk (ranking.build_entity_table: one read per table instead of one per chunk)."""
        weight = self.embeddings.weight
        if out.is_cuda and not torch.is_grad_enabled() and out.dtype == torch.float32 and weight.dtype == torch.float32 \
                and text_tok.dim() == 2 and out.stride(1) == 1 and ops.bow_rows_supported(weight.shape[1]) \
                and type(self.embeddings) is nn.Embedding and self.embeddings.max_norm is None:  # (padding_idx: gradients only)
            flag = self.__dict__.get("_bow_bad_tok")  # (a plain attribute: not a buffer, not in the state_dict)
            if flag is None or flag.device != out.device:
                flag = self.__dict__["_bow_bad_tok"] = torch.zeros((), dtype=torch.int32, device=out.device)
            rows = ops.bow_rows(text_tok, text_mask, weight, out, self.normalize_embs, bad_flag=flag)
            if not defer_check:
                self.check_tokens()
            return rows
        return super().encode_into(out, text_tok, text_mask)

    def check_tokens(self):
        """Raise IndexError if a fused encode_into since the last check saw a token id outside the embedding table (the
        kernel reads row 0 for it and sets a device flag; nn.Embedding's own device-side assert surfaces as late as this).
        ranking.build_entity_table calls it once after its last chunk: one host read per table, none per chunk."""
        flag = self.__dict__.get("_bow_bad_tok")
        if flag is not None and flag.item() < 0:
            flag.zero_()
            raise IndexError("a token id is outside the embedding table")


class DKRL(WordEmbeddingsLP):
    """DKRL two-layer CNN description encoder (models.py:158-204)."""

    def __init__(self, dim, rel_model, loss_fn, num_relations, regularizer, encoder_name=None, embeddings=None):
        super().__init__(rel_model, loss_fn, num_relations, regularizer, dim, encoder_name, embeddings)
        emb_dim = self.embeddings.embedding_dim
        self.conv1 = nn.Conv1d(emb_dim, self.dim, kernel_size=2)
        self.conv2 = nn.Conv1d(self.dim, self.dim, kernel_size=2)

    def _encode_entity(self, text_tok, text_mask):
        if text_mask is None:
            text_mask = torch.ones_like(text_tok, dtype=torch.float)
        embs = self.embeddings(text_tok) * text_mask.unsqueeze(dim=-1)
        embs = embs.transpose(1, 2)            # (N, C, L)
        text_mask = text_mask.unsqueeze(1)
        embs = self.conv1(F.pad(embs, [0, 1])) * text_mask
        length = embs.shape[2]
        kernel_size = 4 if length >= 4 else (1 if length == 1 else 2)
        embs = F.max_pool1d(embs, kernel_size=kernel_size)
        text_mask = F.max_pool1d(text_mask, kernel_size=kernel_size)
        embs = self.conv2(F.pad(torch.tanh(embs), [0, 1]))
        lengths = torch.sum(text_mask, dim=-1)
        return torch.tanh(torch.sum(embs * text_mask, dim=-1) / lengths)

    def encode_into(self, out, text_tok, text_mask=None, defer_check=False):
        """(``defer_check``: as BOW.encode_into.)
        Table build (train.py:96-121): on a HIP device and outside autograd the whole encoder, F.normalize and the row
        assignment are one kernel (ops.dkrl_rows -> blp_dkrl_rows: conv1 on the matrix cores in f32, pooling and tanh in its
        accumulators, conv2 folded into the masked mean); the stock modules run a gather, two transposes, two convolutions, two
        poolings and five elementwise kernels over (B, L, E) / (B, dim, L) temporaries."""
        weight = self.embeddings.weight
        if out.is_cuda and not torch.is_grad_enabled() and out.dtype == torch.float32 and weight.dtype == torch.float32 \
                and self.conv1.weight.dtype == torch.float32 and text_tok.dim() == 2 and out.stride(1) == 1 \
                and type(self.embeddings) is nn.Embedding and self.embeddings.max_norm is None \
                and type(self.conv1) is nn.Conv1d and type(self.conv2) is nn.Conv1d and self.conv1.bias is not None \
                and self.conv2.bias is not None and ops.dkrl_rows_supported(weight.shape[1], self.dim, text_tok.shape[1]):
            flag = self.__dict__.get("_dkrl_bad_tok")  # (a plain attribute: not a buffer, not in the state_dict)
            if flag is None or flag.device != out.device:
                flag = self.__dict__["_dkrl_bad_tok"] = torch.zeros((), dtype=torch.int32, device=out.device)
            rows = ops.dkrl_rows(text_tok, text_mask, weight, self.conv1, self.conv2, out, self.normalize_embs, bad_flag=flag)
            if not defer_check:
                self.check_tokens()
            return rows
        return super().encode_into(out, text_tok, text_mask)

    def check_tokens(self):
        """As BOW.check_tokens: raise IndexError if a fused encode_into since the last check saw a token id outside the
        embedding table (ranking.build_entity_table calls it once after its last chunk)."""
        flag = self.__dict__.get("_dkrl_bad_tok")
        if flag is not None and flag.item() < 0:
            flag.zero_()
            raise IndexError("a token id is outside the embedding table")


class TransductiveLinkPrediction(LinkPrediction):
    """Entity lookup table instead of a text encoder (models.py:207-219)."""

    def __init__(self, dim, rel_model, loss_fn, num_entities, num_relations, regularizer):
        super().__init__(dim, rel_model, loss_fn, num_relations, regularizer)
        self.ent_emb = nn.Embedding(num_entities, dim)
        nn.init.xavier_uniform_(self.ent_emb.weight.data)

    def _encode_entity(self, entities):
        return self.ent_emb(entities)

    def forward(self, pos_pairs, rels=None, neg_idx=None):
        embs = self.encode(pos_pairs)
        if rels is None and neg_idx is None:
            return embs  # lets the evaluation loop build the entity table (the reference's call at
            #              train.py:113 passes one argument and raises TypeError there)
        return self.compute_loss(embs, rels, neg_idx)
