"""Datasets, collate functions and the in-batch negative sampler with the reference's interface
(/root/reference/data.py).  These are the producers of the hot path's inputs -- (text_tok, text_mask,
rels, neg_idx) for training, (head, tail, rel) triples for evaluation -- and stay host-side Python:
I/O and tokenisation are outside the accelerated path.

File formats are the reference's: ``entities.txt`` / ``relations.txt`` (one name per line, id = line
number), ``*.tsv`` triples "head rel tail" by name, ``entity2text[long].txt`` "name<TAB>description",
optional ``relations-cat.txt``; triples are stored as (head, tail, rel) ids (data.py:128).
``write_synthetic_dataset`` produces a dataset of any shape in that format (there is no network to
download the real ones), including a GloVe-style embedding tensor and vocabulary map.
"""
import logging
import os
import os.path as osp
import re
import string

import torch
from torch.utils.data import Dataset

UNK = "[UNK]"
CATEGORY_IDS = {"1-to-1": 0, "1-to-many": 1, "many-to-1": 2, "many-to-many": 3}

try:  # the reference downloads NLTK corpora at import; NLTK is optional here
    import nltk
    from nltk.corpus import stopwords as _sw
    STOP_WORDS = _sw.words("english")
    _word_tokenize = nltk.word_tokenize
except Exception:  # noqa: BLE001 - any failure (missing package, missing corpus) -> built-in fallback
    STOP_WORDS = ("i me my myself we our ours ourselves you your yours yourself yourselves he him his himself she "
                  "her hers herself it its itself they them their theirs themselves what which who whom this that "
                  "these those am is are was were be been being have has had having do does did doing a an the and "
                  "but if or because as until while of at by for with about against between into through during "
                  "before after above below to from up down in out on off over under again further then once here "
                  "there when where why how all any both each few more most other some such no nor not only own "
                  "same so than too very s t can will just don should now").split()

    def _word_tokenize(text):
        return re.findall(r"\w+|[^\w\s]", text)
DROPPED = list(STOP_WORDS) + list(string.punctuation)


def file_to_ids(file_path):
    """One name per line -> {name: line number}."""
    with open(file_path) as file:
        return {line.strip(): i for i, line in enumerate(file)}


def get_negative_sampling_indices(batch_size, num_negatives, repeats=1):
    """Indices of in-batch negatives, shape (batch_size * repeats, num_negatives, 2).

    The 2 * batch_size entity slots of a batch are numbered row-wise ([[0, 1], [2, 3], ...]).  A
    negative for row b keeps one of its two slots and replaces the other by a slot drawn uniformly
    from the OTHER rows (never its own pair).  With ``repeats`` = number of devices, each device's
    slice of the batch gets indices local to that slice (nn.DataParallel scatters dim 0).
    Consumes the torch RNG exactly like the reference (one multinomial over a (B, 2B) weight matrix,
    then one randint), so a fixed seed reproduces the reference's indices (tests/golden).
    """
    num_slots = batch_size * 2
    pairs = torch.arange(num_slots).reshape(batch_size, 2)
    total = num_negatives * repeats

    # uniform over every slot except the two of the row itself
    weights = torch.ones(batch_size, num_slots, dtype=torch.float)
    weights.scatter_(1, pairs, torch.zeros(batch_size, 2))
    replacement = weights.multinomial(total, replacement=True).t().flatten()

    which_column = torch.randint(0, 2, [batch_size * total])
    neg_idx = pairs.repeat((total, 1))
    neg_idx[torch.arange(batch_size * total), which_column] = replacement
    neg_idx = neg_idx.reshape(-1, batch_size * repeats, 2)
    return neg_idx.transpose_(0, 1)


def negative_indices_from_draws(draw, which):
    """The index construction of the in-batch sampler as a deterministic function of INTEGER draws, on whatever device they
    live -- the checkable half of the device sampler (the other half is two torch.randint calls):

        draw  (B, K) int64 in [0, 2B - 2): which of the 2B - 2 slots of the OTHER rows replaces a slot of row b -- counted in
              slot order with row b's own pair {2b, 2b + 1} skipped (what the reference's multinomial over its (B, 2B) weight
              matrix with the pair zeroed draws, data.py:57-61);
        which (B, K) int64 in {0, 1}: the column of the pair that is replaced (the reference's randint, data.py:63);
        ->    (B, K, 2) int64: row b's pair (2b, 2b + 1) with column which[b, k] replaced by the drawn slot (data.py:64-67).

    oracle/ref_port.py restates it with Python loops; tests hold the two bit-for-bit on the GPU box, and invert the
    reference's own golden indices (tests/golden/neg_sampling.npz) into draws that this function maps back onto them."""
    b = draw.shape[0]
    own = torch.arange(b, device=draw.device).view(b, 1)
    replacement = draw + 2 * (draw >= 2 * own).long()   # skip the own pair
    pairs = torch.stack((2 * own, 2 * own + 1), dim=-1).expand(b, draw.shape[1], 2).clone()
    pairs.scatter_(2, which.unsqueeze(-1), replacement.unsqueeze(-1))
    return pairs


def draws_from_negative_indices(neg_idx):
    """Inverse of negative_indices_from_draws on valid indices: (draw, which) of a (B, K, 2) index tensor."""
    b = neg_idx.shape[0]
    own = torch.arange(b, device=neg_idx.device).view(b, 1)
    which = (neg_idx[..., 0] == 2 * own).long()  # column 0 intact -> column 1 is the replaced one (a drawn slot is never the row's own)
    replaced = torch.where(which == 0, neg_idx[..., 0], neg_idx[..., 1])
    return replaced - 2 * (replaced > 2 * own + 1).long(), which


def get_negative_sampling_indices_on_device(batch_size, num_negatives, device, generator=None):
    """Same distribution as get_negative_sampling_indices (repeats = 1), drawn directly on ``device``: two torch.randint calls
    (the draws) + negative_indices_from_draws (the indices) -- no (B, 2B) weight matrix, no host-to-device copy per step
    (SURVEY.md 8f, next-row 4).  The torch CUDA/HIP generator is a different stream than the CPU one, so the values
    differ from the reference's for a given seed (why train.py keeps the reference's CPU sampler by default: a seeded run
    reproduces the reference's indices); the law is identical: for row b one of the two slots is kept and the other is
    replaced by a slot drawn uniformly from the 2B - 2 slots of the other rows."""
    b, k = batch_size, num_negatives
    draw = torch.randint(0, 2 * b - 2, (b, k), device=device, generator=generator)
    which = torch.randint(0, 2, (b, k), device=device, generator=generator)
    return negative_indices_from_draws(draw, which)


class GraphDataset(Dataset):
    """Triples of a knowledge graph as a (num_triples, 3) tensor of (head, tail, rel) ids."""

    def __init__(self, triples_file, neg_samples=None, write_maps_file=False, num_devices=1):
        directory = osp.dirname(triples_file)
        maps_path = osp.join(directory, "maps.pt")
        if not write_maps_file:
            if not osp.exists(maps_path):
                raise ValueError("Maps file not found.")
            maps = torch.load(maps_path)
            ent_ids, rel_ids = maps["ent_ids"], maps["rel_ids"]
        else:
            ent_ids = file_to_ids(osp.join(directory, "entities.txt"))
            rel_ids = file_to_ids(osp.join(directory, "relations.txt"))

        entities, relations, triples = set(), set(), []
        with open(triples_file) as file:
            for line in file:
                values = line.split()
                if len(values) > 3 and values[3] == "-1":  # FB13 / WN11 negative examples
                    continue
                head, rel, tail = values[:3]
                entities.update([head, tail])
                relations.add(rel)
                triples.append([ent_ids[head], ent_ids[tail], rel_ids[rel]])
        self.triples = torch.tensor(triples, dtype=torch.long).reshape(-1, 3)

        self.rel_categories = torch.zeros(len(rel_ids), dtype=torch.long)
        self.has_rel_categories = False
        categories_file = osp.join(directory, "relations-cat.txt")
        if osp.exists(categories_file):
            with open(categories_file) as f:
                for line in f:
                    rel, cat = line.strip().split()
                    self.rel_categories[rel_ids[rel]] = CATEGORY_IDS[cat]
            self.has_rel_categories = True

        if write_maps_file:
            torch.save({"ent_ids": ent_ids, "rel_ids": rel_ids}, maps_path)

        self.num_ents = len(entities)
        self.num_rels = len(relations)
        self.entities = torch.tensor([ent_ids[ent] for ent in entities], dtype=torch.long)
        self.num_triples = self.triples.shape[0]
        self.directory = directory
        self.maps_path = maps_path
        self.neg_samples = neg_samples
        self.num_devices = num_devices
        self.sampler_device = None  # a HIP device: draw neg_idx there (train.py `device_sampler=True`)

    def _neg_idx(self, batch_size, repeats):
        """In-batch negative indices (B * repeats, K, 2): the reference's CPU sampler, or -- with sampler_device
        set -- the same law drawn on the GPU, one independent draw per device slice (indices local to the slice)."""
        if self.sampler_device is None:
            return get_negative_sampling_indices(batch_size, self.neg_samples, repeats=repeats)
        return torch.cat([get_negative_sampling_indices_on_device(batch_size, self.neg_samples, self.sampler_device)
                          for _ in range(repeats)])

    def __getitem__(self, index):
        return self.triples[index]

    def __len__(self):
        return self.num_triples

    def collate_fn(self, data_list):
        """Batch of triples -> (pos_pairs (B, 2), rels (B, 1), neg_idx (B, K, 2))."""
        pos_pairs, rels = torch.stack(data_list).split(2, dim=1)
        return pos_pairs, rels, self._neg_idx(len(data_list), 1)


class TextGraphDataset(GraphDataset):
    """A graph plus tokenised entity descriptions: ``text_data`` is (num_entities, max_len + 1) int64,
    token ids padded with 0, last column = description length (data.py:216-253)."""

    def __init__(self, triples_file, neg_samples, max_len, tokenizer, drop_stopwords, write_maps_file=False,
                 use_cached_text=False, num_devices=1):
        super().__init__(triples_file, neg_samples, write_maps_file, num_devices)
        ent_ids = torch.load(self.maps_path)["ent_ids"]
        if max_len is None:
            max_len = getattr(tokenizer, "model_max_length", None) or tokenizer.max_len

        cached_text_path = osp.join(self.directory, "text_data.pt")
        logger = logging.getLogger()
        if use_cached_text and osp.exists(cached_text_path):
            self.text_data = torch.load(cached_text_path)
            logger.info(f"Loaded cached text data for {self.text_data.shape[0]} entities, "
                        f"and maximum length {self.text_data.shape[1]}.")
            return
        if use_cached_text:
            logger.info("Cached text data not found.")

        self.text_data = torch.zeros((len(ent_ids), max_len + 1), dtype=torch.long)
        read_entities = set()
        for text_file in ("entity2textlong.txt", "entity2text.txt"):
            file_path = osp.join(self.directory, text_file)
            if not osp.exists(file_path):
                continue
            with open(file_path) as f:
                for line in f:
                    values = line.strip().split("\t")
                    entity = values[0]
                    if entity not in ent_ids or entity in read_entities:
                        continue
                    read_entities.add(entity)
                    text = " ".join(values[1:])
                    if drop_stopwords:
                        text = " ".join(t for t in _word_tokenize(text) if t.lower() not in DROPPED)
                    tokens = _encode(tokenizer, text, max_len)
                    row = ent_ids[entity]
                    self.text_data[row, : tokens.shape[0]] = tokens
                    self.text_data[row, -1] = tokens.shape[0]
        if len(read_entities) != len(ent_ids):
            raise ValueError(f"Read {len(read_entities):,} descriptions, but {len(ent_ids):,} were expected.")
        if self.text_data[:, -1].min().item() < 1:
            raise ValueError("Some entries in text_data contain length-0 descriptions.")
        torch.save(self.text_data, cached_text_path)

    def get_entity_description(self, ent_ids):
        """(tokens truncated to the longest description of the batch, float mask, lengths)."""
        text_data = self.text_data[ent_ids]
        text_tok, text_len = text_data.split(text_data.shape[-1] - 1, dim=-1)
        text_tok = text_tok[..., : text_len.max()]
        return text_tok, (text_tok > 0).float(), text_len

    def collate_fn(self, data_list):
        """Batch of triples -> (text_tok (B, 2, L), text_mask (B, 2, L), rels (B, 1), neg_idx)."""
        batch_size = len(data_list) // self.num_devices
        if batch_size <= 1:
            raise ValueError("collate_text can only work with batch sizes larger than 1.")
        pos_pairs, rels = torch.stack(data_list).split(2, dim=1)
        text_tok, text_mask, _ = self.get_entity_description(pos_pairs)
        neg_idx = self._neg_idx(batch_size, self.num_devices)
        return text_tok, text_mask, rels, neg_idx


def _encode(tokenizer, text, max_len):
    """1-D LongTensor of at most max_len token ids, for HF tokenizers (old and new API) and GloVeTokenizer."""
    try:
        tokens = tokenizer.encode(text, max_length=max_len, truncation=True, return_tensors="pt")
    except TypeError:
        tokens = tokenizer.encode(text, max_length=max_len, return_tensors="pt")
    return torch.as_tensor(tokens).reshape(-1)[:max_len]


class GloVeTokenizer:
    """Word-level tokenizer over a {word: id} map saved with torch.save (data.py:303-334)."""

    def __init__(self, vocab_dict_file, uncased=True):
        self.word2idx = vocab_dict_file if isinstance(vocab_dict_file, dict) else torch.load(vocab_dict_file)
        self.uncased = uncased

    def encode(self, text, max_length, return_tensors=None, **kwargs):
        if self.uncased:
            text = text.lower()
        unk = self.word2idx[UNK]
        encoded = [[self.word2idx.get(t, unk) for t in _word_tokenize(text)][:max_length]]
        return torch.tensor(encoded) if return_tensors else encoded

    def batch_encode_plus(self, batch, max_length, **kwargs):
        rows = []
        for text in batch:
            tokens = self.encode(text, max_length, return_tensors=False)[0]
            rows.append(tokens + [0] * (max_length - len(tokens)))
        input_ids = torch.tensor(rows, dtype=torch.long)
        return {"input_ids": input_ids, "attention_mask": (input_ids > 0).float()}


# --------------------------------------------------------------------------------------------------
def write_synthetic_dataset(root, name, num_entities, num_relations, num_train, num_valid, num_test,
                            vocab_size=2000, emb_dim=300, max_words=24, inductive=False, seed=0):
    """Write a random dataset in the reference's on-disk format under ``root/name`` plus a GloVe-style
    embedding table under ``root/glove`` (so model='glove-bow' / 'glove-dkrl' run offline).

    inductive=True writes ind-train/ind-dev/ind-test.tsv where dev and test triples each contain at
    least one entity unseen in training, like the reference's inductive splits (data/utils.py:80-199).
    """
    gen = torch.Generator().manual_seed(seed)
    directory = osp.join(root, name)
    os.makedirs(directory, exist_ok=True)
    os.makedirs(osp.join(root, "glove"), exist_ok=True)
    ents = [f"e{i}" for i in range(num_entities)]
    rels = [f"r{i}" for i in range(num_relations)]
    words = [f"w{i}" for i in range(vocab_size)]
    with open(osp.join(directory, "entities.txt"), "w") as f:
        f.write("\n".join(ents) + "\n")
    with open(osp.join(directory, "relations.txt"), "w") as f:
        f.write("\n".join(rels) + "\n")
    with open(osp.join(directory, "entity2text.txt"), "w") as f:
        for e in ents:
            n = int(torch.randint(4, max_words + 1, (1,), generator=gen))
            ids = torch.randint(0, vocab_size, (n,), generator=gen).tolist()
            f.write(e + "\t" + " ".join(words[i] for i in ids) + "\n")

    def triples(n, pool_h, pool_t):
        h = pool_h[torch.randint(0, len(pool_h), (n,), generator=gen)]
        t = pool_t[torch.randint(0, len(pool_t), (n,), generator=gen)]
        r = torch.randint(0, num_relations, (n,), generator=gen)
        return [(ents[a], rels[c], ents[b]) for a, b, c in zip(h.tolist(), t.tolist(), r.tolist())]

    perm = torch.randperm(num_entities, generator=gen)
    if inductive:
        n_new = max(2, num_entities // 10)
        train_pool, dev_new, test_new = perm[: num_entities - 2 * n_new], perm[-2 * n_new: -n_new], perm[-n_new:]
        splits = {"ind-train": triples(num_train, train_pool, train_pool),
                  "ind-dev": triples(num_valid, dev_new, train_pool),
                  "ind-test": triples(num_test, test_new, torch.cat((train_pool, dev_new)))}
        # every entity of a split must occur in it (the reference derives entity sets from the triples)
        splits["ind-train"] += [(ents[int(a)], rels[0], ents[int(b)]) for a, b in zip(train_pool, train_pool.roll(1))]
    else:
        splits = {"train": triples(num_train, perm, perm), "dev": triples(num_valid, perm, perm),
                  "test": triples(num_test, perm, perm)}
        splits["train"] += [(ents[int(a)], rels[i % num_relations], ents[int(b)])
                            for i, (a, b) in enumerate(zip(perm, perm.roll(1)))]
    for split, rows in splits.items():
        with open(osp.join(directory, split + ".tsv"), "w") as f:
            for h, r, t in rows:
                f.write(f"{h}\t{r}\t{t}\n")

    emb = torch.randn(vocab_size + 2, emb_dim, generator=gen) * 0.3
    emb[0] = 0  # id 0 is padding
    word2idx = {w: i + 1 for i, w in enumerate(words)}
    word2idx[UNK] = vocab_size + 1
    torch.save(emb, osp.join(root, "glove", "glove.6B.300d.pt"))
    torch.save(word2idx, osp.join(root, "glove", "glove.6B.300d-maps.pt"))
    return directory
