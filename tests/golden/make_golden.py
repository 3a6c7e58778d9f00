"""Generate golden vectors by IMPORTING the reference (dfdazac/blp) in the build container.

Run:  python tests/golden/make_golden.py            (needs /root/reference; CPU only)

The reference is pure Python and cannot travel to the GPU box, so its outputs on seeded inputs are
committed here as small .npz fixtures (data only: inputs and expected outputs).  Nothing in the test
suite, smoke() or bench.py reads /root/reference at run time.

``nltk`` and ``sacred`` are not installed in this image; the two stub modules registered below exist
only inside this script so that ``import data`` / ``import train`` succeed.  The functions whose
outputs we record (models.*_score, LinkPrediction.compute_loss, utils.get_metrics,
utils.get_triple_filters, utils.make_ent2idx, data.get_negative_sampling_indices,
train.eval_link_prediction, models.BOW / models.DKRL `encode`) are the reference's own, unmodified.
"""
import logging
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

REF = os.environ.get("BLP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    nltk = types.ModuleType("nltk")
    nltk.download = lambda *a, **k: True
    nltk.word_tokenize = lambda text: text.split()
    corpus = types.ModuleType("nltk.corpus")
    stopwords = types.SimpleNamespace(words=lambda lang: ["the", "a", "of"])
    corpus.stopwords = stopwords
    nltk.corpus = corpus
    sys.modules["nltk"] = nltk
    sys.modules["nltk.corpus"] = corpus

    sacred = types.ModuleType("sacred")

    class Experiment:
        def __init__(self, *a, **k):
            self.observers = []
            self.logger = None

        def _identity(self, fn):
            return fn

        config = capture = command = automain = main = _identity

        def run_commandline(self, *a, **k):
            return None

    sacred.Experiment = Experiment
    run_mod = types.ModuleType("sacred.run")

    class Run:
        pass

    run_mod.Run = Run
    obs_mod = types.ModuleType("sacred.observers")
    obs_mod.MongoObserver = object
    sacred.run = run_mod
    sacred.observers = obs_mod
    sys.modules["sacred"] = sacred
    sys.modules["sacred.run"] = run_mod
    sys.modules["sacred.observers"] = obs_mod


_install_stubs()
sys.path.insert(0, REF)
_cwd = os.getcwd()
os.chdir(tempfile.mkdtemp())  # the reference writes nothing at import, but keep cwd off the repo
import models  # noqa: E402  (reference)
import utils  # noqa: E402  (reference)
import data  # noqa: E402  (reference)
import train  # noqa: E402  (reference)
import networkx as nx  # noqa: E402

os.chdir(_cwd)

REL_MODELS = ("transe", "distmult", "complex", "simple")
SCORE = {"transe": models.transe_score, "distmult": models.distmult_score,
         "complex": models.complex_score, "simple": models.simple_score}


def _seed(*parts):
    return zlib.crc32(repr(parts).encode()) % (2 ** 31)


def save(name, **arrays):
    arrays["torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def make_table(kind, n, d, gen):
    if kind == "normalized":      # what the TransE encoder emits (models.py:40-41)
        return torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=-1)
    if kind == "gauss":           # un-normalised, bilinear models
        return torch.randn(n, d, generator=gen) * 0.1
    if kind == "dyadic":          # k/64: every partial sum exact -> any summation order agrees
        return torch.randint(-64, 65, (n, d), generator=gen).float() / 64.0
    if kind == "ties":            # duplicated rows: exercises the > / >= average-rank rule
        base = torch.randint(-8, 9, (n // 4, d), generator=gen).float() / 8.0
        return base.repeat(4, 1)[torch.randperm(4 * (n // 4), generator=gen)]
    raise ValueError(kind)


def gen_scores():
    """models.*_score on the eval broadcast shapes (train.py:146-147) + utils.get_metrics."""
    for rel_model in REL_MODELS:
        for kind in ("normalized", "gauss", "dyadic", "ties"):
            for d in (128, 64):
                gen = torch.Generator().manual_seed(_seed(rel_model, kind, d))
                n, b, nrel = 192, 6, 5
                table = make_table(kind, n, d, gen)
                rel_w = (torch.rand(nrel, d, generator=gen) - 0.5) * 0.25
                if kind in ("dyadic", "ties"):
                    rel_w = torch.randint(-16, 17, (nrel, d), generator=gen).float() / 16.0
                heads = torch.randint(0, n, (b, 1), generator=gen)
                tails = torch.randint(0, n, (b, 1), generator=gen)
                rels = torch.randint(0, nrel, (b, 1), generator=gen)
                ent = table.unsqueeze(0)
                h, t, r = table[heads], table[tails], rel_w[rels]
                fn = SCORE[rel_model]
                head_pred = fn(ent, t, r)
                tail_pred = fn(h, ent, r)
                pred = torch.cat((head_pred, tail_pred))
                true = torch.cat((heads, tails))
                k_values = torch.tensor([[1, 3, 10]])
                rr, hits = utils.get_metrics(pred, true, k_values)
                true_scores = pred.gather(1, true)
                gt = (pred > true_scores).sum(1)
                ge = (pred >= true_scores).sum(1)
                save(f"scores_{rel_model}_{kind}_d{d}", table=table, rel_w=rel_w, heads=heads,
                     tails=tails, rels=rels, head_pred=head_pred, tail_pred=tail_pred, rr=rr,
                     hits=hits, gt=gt, ge=ge)


def gen_sum_order():
    """Pin torch.sum(dim=-1) and torch.norm(p=1) order for the widths the encoders can emit."""
    gen = torch.Generator().manual_seed(7)
    out = {}
    for n in (8, 32, 64, 100, 128, 200, 300, 512, 768, 1024, 2080):
        x = torch.randn(64, n, generator=gen)
        out[f"x_{n}"] = x
        out[f"sum_{n}"] = torch.sum(x, dim=-1)
        out[f"l1_{n}"] = torch.norm(x, dim=-1, p=1)
    save("sum_order", **out)


def gen_score_pairs_wide():
    """TransE on BOW widths (300 = GloVe, 768 = BERT embeddings) and training shape (B, K, D)."""
    gen = torch.Generator().manual_seed(11)
    out = {}
    for d in (300, 768):
        h = torch.nn.functional.normalize(torch.randn(40, d, generator=gen), dim=-1)
        t = torch.nn.functional.normalize(torch.randn(40, d, generator=gen), dim=-1)
        r = (torch.rand(40, d, generator=gen) - 0.5) * 0.1
        out[f"h_{d}"], out[f"t_{d}"], out[f"r_{d}"] = h, t, r
        out[f"transe_{d}"] = models.transe_score(h, t, r)
    b, k, d = 8, 16, 128
    for rel_model in REL_MODELS:
        h = torch.randn(b, k, d, generator=gen) * 0.3
        t = torch.randn(b, k, d, generator=gen) * 0.3
        r = torch.randn(b, 1, d, generator=gen) * 0.3
        out[f"train_h_{rel_model}"], out[f"train_t_{rel_model}"], out[f"train_r_{rel_model}"] = h, t, r
        out[f"train_{rel_model}"] = SCORE[rel_model](h, t, r)
    save("score_pairs", **out)


def gen_loss():
    """LinkPrediction.compute_loss (models.py:51-70) value + grads wrt ent_embs and rel_emb.weight."""
    for rel_model in REL_MODELS:
        for loss_fn in ("margin", "nll"):
            for reg in (0.0, 1e-3):
                seed = _seed(rel_model, loss_fn, reg)
                torch.manual_seed(seed)
                b, k, d, nrel = 8, 12, 128, 7
                model = models.LinkPrediction(d, rel_model, loss_fn, nrel, reg)
                ent = torch.randn(b, 2, d) * (0.5 if rel_model != "transe" else 1.0)
                if rel_model == "transe":
                    ent = torch.nn.functional.normalize(ent, dim=-1)
                ent.requires_grad_(True)
                rels = torch.randint(0, nrel, (b, 1))
                neg_idx = data.get_negative_sampling_indices(b, k)
                loss = model.compute_loss(ent, rels, neg_idx)
                loss.backward()
                tag = f"loss_{rel_model}_{loss_fn}_reg{'0' if reg == 0 else '1e-3'}"
                save(tag, ent_embs=ent.detach(), rel_w=model.rel_emb.weight.detach(), rels=rels,
                     neg_idx=neg_idx, regularizer=np.float64(reg), loss=loss.detach(),
                     grad_ent=ent.grad, grad_rel_w=model.rel_emb.weight.grad)


def gen_neg_sampling():
    """data.get_negative_sampling_indices for fixed torch seeds (same torch build on both boxes)."""
    out = {}
    for i, (b, k, rep) in enumerate([(4, 3, 1), (8, 5, 1), (6, 4, 2), (64, 64, 1)]):
        torch.manual_seed(100 + i)
        out[f"case{i}_args"] = np.array([b, k, rep, 100 + i])
        out[f"case{i}_neg_idx"] = data.get_negative_sampling_indices(b, k, repeats=rep)
    out["ent2idx_doc"] = utils.make_ent2idx(torch.tensor([4, 5, 0]), 5)  # utils.py:36-38
    ents = torch.tensor([9, 2, 7, 0, 11])
    out["ent2idx_ents"] = ents
    out["ent2idx_out"] = utils.make_ent2idx(ents, 14)
    save("neg_sampling", **out)


class _ToyTriples(torch.utils.data.Dataset):
    def __init__(self, triples, rel_categories, has_cats):
        self.triples = triples
        self.rel_categories = rel_categories
        self.has_rel_categories = has_cats

    def __getitem__(self, i):
        return self.triples[i]

    def __len__(self):
        return self.triples.shape[0]


class _RecordingRun:
    _id = None

    def __init__(self):
        self.scalars = {}

    def log_scalar(self, name, value, step=None):
        self.scalars[name] = float(value)


def gen_filters_and_eval():
    """utils.get_triple_filters masks and the full train.eval_link_prediction scalar dict on a toy
    graph (raw + filtered MRR / Hits@k, by-new-position split), one fixture per relational model."""
    gen = torch.Generator().manual_seed(2024)
    num_ids, nrel, d, vocab, max_len = 70, 6, 128, 40, 8
    # entity ids are sparse on purpose (ent2idx == -1 for unused ids and for non-candidates)
    all_ids = torch.randperm(num_ids, generator=gen)[:60]
    cand = all_ids[:52]                      # candidate entities of this eval (rows of the table)
    hot = cand[:10]
    def draw(m):
        h = torch.where(torch.rand(m, generator=gen) < 0.5, hot[torch.randint(0, 10, (m,), generator=gen)],
                        cand[torch.randint(0, 52, (m,), generator=gen)])
        t = cand[torch.randint(0, 52, (m,), generator=gen)]
        r = torch.randint(0, nrel, (m,), generator=gen)
        return torch.stack((h, t, r), dim=1)
    train_triples = draw(400)
    # edges touching non-candidate ids: must be skipped by ent2idx == -1 (utils.py:73,80)
    outside = all_ids[52:]
    extra = torch.stack((hot[torch.randint(0, 10, (40,), generator=gen)],
                         outside[torch.randint(0, 8, (40,), generator=gen)],
                         torch.randint(0, nrel, (40,), generator=gen)), dim=1)
    test_triples = draw(44)
    # duplicate a few test triples and add reflexive ones (h == t) to hit the t != tail exclusions
    test_triples[40] = test_triples[3]
    test_triples[41] = torch.tensor([hot[0], hot[0], 1])
    graph = nx.MultiDiGraph()
    all_triples = torch.cat((train_triples, extra, test_triples))
    graph.add_weighted_edges_from(all_triples.tolist())
    new_entities = set(cand[40:].tolist())
    rel_categories = torch.randint(0, 4, (nrel,), generator=gen)

    text_data = torch.zeros((num_ids, max_len + 1), dtype=torch.long)
    lens = torch.randint(1, max_len + 1, (num_ids,), generator=gen)
    for i in range(num_ids):
        text_data[i, :lens[i]] = torch.randint(1, vocab, (int(lens[i]),), generator=gen)
        text_data[i, -1] = lens[i]
    text_ds = types.SimpleNamespace(text_data=text_data)
    text_ds.get_entity_description = types.MethodType(data.TextGraphDataset.get_entity_description, text_ds)

    word_emb = torch.randn(vocab, d, generator=gen) * 0.3
    emb_file = os.path.join(tempfile.mkdtemp(), "emb.pt")
    torch.save(word_emb, emb_file)

    # masks alone, for the CSR builder test
    max_ent_id = max(graph.nodes)
    ent2idx = utils.make_ent2idx(cand, max_ent_id)
    hf, tf = utils.get_triple_filters(test_triples, graph, cand.shape[0], ent2idx)
    save("filters_toy", triples=test_triples, graph_edges=all_triples, entities=cand,
         max_ent_id=np.int64(max_ent_id), ent2idx=ent2idx, heads_filter=hf, tails_filter=tf)

    log = logging.getLogger("golden")
    for rel_model in REL_MODELS:
        torch.manual_seed(5)
        model = models.BOW(rel_model, "margin", nrel, 0.0, embeddings=emb_file)
        loader = torch.utils.data.DataLoader(_ToyTriples(test_triples, rel_categories, True), 8)
        run = _RecordingRun()
        train.device = torch.device("cpu")
        mrr, ent_emb = train.eval_link_prediction(model, loader, text_ds, cand, 3, 16, run, log,
                                                  prefix="test", filtering_graph=graph,
                                                  new_entities=new_entities, return_embeddings=True)
        run_raw = _RecordingRun()
        train.eval_link_prediction(model, loader, text_ds, cand, 3, 16, run_raw, log, prefix="valid",
                                   max_num_batches=5)
        names = sorted(run.scalars)
        names_raw = sorted(run_raw.scalars)
        save(f"eval_toy_{rel_model}", triples=test_triples, graph_edges=all_triples, entities=cand,
             new_entities=np.array(sorted(new_entities)), rel_categories=rel_categories,
             text_data=text_data, word_emb=word_emb, rel_w=model.rel_emb.weight.detach(),
             ent_emb=ent_emb.squeeze(0), returned_mrr=np.float64(mrr),
             scalar_names=np.array(names), scalar_values=np.array([run.scalars[n] for n in names]),
             raw_names=np.array(names_raw),
             raw_values=np.array([run_raw.scalars[n] for n in names_raw]),
             eval_batch_size=np.int64(8), emb_batch_size=np.int64(16))


def gen_encoders():
    """The description encoders whose table build is fused on the GPU (models.BOW: models.py:140-155; models.DKRL:
    models.py:158-204), through the reference's own `encode` (F.normalize for TransE, models.py:38-43): padded descriptions of
    every length, two chunk lengths (one M-tile of 32 positions; two)."""
    for L in (9, 37):
        gen = torch.Generator().manual_seed(_seed("encoders", L))
        vocab, E, n, dim = 40, 20, 14, 128
        word_emb = torch.randn(vocab, E, generator=gen) * 0.3
        emb_file = os.path.join(tempfile.mkdtemp(), "emb.pt")
        torch.save(word_emb, emb_file)
        lengths = torch.randint(1, L + 1, (n,), generator=gen)
        lengths[0] = L
        tok = torch.randint(1, vocab, (n, L), generator=gen)
        mask = (torch.arange(L).unsqueeze(0) < lengths.unsqueeze(1)).float()
        tok = tok * mask.long()
        out = {}
        for rel_model in ("transe", "distmult"):
            torch.manual_seed(_seed("dkrl", rel_model, L))
            dkrl = models.DKRL(dim, rel_model, "margin", 3, 0.0, embeddings=emb_file)
            with torch.no_grad():
                dkrl.conv1.bias.uniform_(-0.2, 0.2)
                dkrl.conv2.bias.uniform_(-0.2, 0.2)
                out[f"dkrl_{rel_model}"] = dkrl.encode(tok, mask)
                if rel_model == "transe":
                    out.update(conv1_w=dkrl.conv1.weight.detach(), conv1_b=dkrl.conv1.bias.detach(),
                               conv2_w=dkrl.conv2.weight.detach(), conv2_b=dkrl.conv2.bias.detach())
                else:  # the same weights: only the normalisation differs
                    dkrl.load_state_dict({**dkrl.state_dict(), "conv1.weight": out["conv1_w"], "conv1.bias": out["conv1_b"],
                                          "conv2.weight": out["conv2_w"], "conv2.bias": out["conv2_b"]})
                    out[f"dkrl_{rel_model}"] = dkrl.encode(tok, mask)
                bow = models.BOW(rel_model, "margin", 3, 0.0, embeddings=emb_file)
                out[f"bow_{rel_model}"] = bow.encode(tok, mask)
        save(f"encoders_L{L}", word_emb=word_emb, tok=tok, mask=mask, **out)


def main():
    torch.set_num_threads(1)
    logging.basicConfig(level=logging.WARNING)
    gen_scores()
    gen_sum_order()
    gen_score_pairs_wide()
    gen_loss()
    gen_neg_sampling()
    gen_filters_and_eval()
    gen_encoders()


if __name__ == "__main__":
    main()
