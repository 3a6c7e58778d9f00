"""python train.py link_prediction with ... on a synthetic dataset in the reference's on-disk format
(BASELINE config 0 shape: UMLS-sized graph, bag-of-words encoder, TransE, margin loss, CPU only)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_link_prediction_cli_end_to_end(tmp_path):
    from blp_amd.data import write_synthetic_dataset
    write_synthetic_dataset(str(tmp_path / "data"), "umls-synth", num_entities=135, num_relations=46,
                            num_train=1280, num_valid=160, num_test=160, vocab_size=500, emb_dim=48, seed=0)
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "link_prediction", "with", "dataset=umls-synth",
           "inductive=False", "model=glove-bow", "rel_model=transe", "loss_fn=margin", "regularizer=1e-2",
           "max_len=32", "num_negatives=16", "lr=1e-3", "use_scheduler=False", "batch_size=64",
           "emb_batch_size=512", "eval_batch_size=64", "max_epochs=1", f"data_root={tmp_path / 'data'}", "seed=1"]
    proc = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    log = proc.stderr + proc.stdout
    for needle in ("Training on CPU", "valid mrr:", "test mrr:", "mrr_filt:", "hits@10_filt:"):
        assert needle in log, needle
    out = tmp_path / "output"
    ent_emb = torch.load(out / "ent_emb-None.pt")
    ents = torch.load(out / "ents-None.pt")
    assert ent_emb.shape == (1, 135, 48) and ents.shape == (135,)
    state = torch.load(out / "model-None.pt")
    assert sorted(state) == ["embeddings.weight", "rel_emb.weight"]
    # TransE rows are L2-normalised (models.py:40-41)
    assert torch.allclose(ent_emb[0].norm(dim=-1), torch.ones(135), atol=1e-5)


def test_node_classification_cli_on_saved_embeddings(tmp_path):
    """train.py node_classification (train.py:408-481 of the reference): embeddings + ids in the files link_prediction
    writes, class files in the dataset directory; classes that are linearly separable in the embeddings are learnt, the
    classifier is saved in the reference's format."""
    import joblib
    import numpy as np
    n, dim = 120, 16
    g = torch.Generator().manual_seed(0)
    labels = torch.arange(n) % 3
    emb = torch.randn(n, dim, generator=g) * 0.1
    emb[torch.arange(n), labels] += 1.0                      # class k stands out along axis k
    ids = torch.randperm(n, generator=g) + 5                 # entity ids are not rows
    (tmp_path / "output").mkdir()
    torch.save(emb.unsqueeze(0), tmp_path / "output" / "ent_emb-7.pt")   # (1, N, D), as eval_link_prediction returns it
    torch.save(ids, tmp_path / "output" / "ents-7.pt")
    data = tmp_path / "data" / "toy"
    data.mkdir(parents=True)
    names = {f"ent{int(i)}": int(i) for i in ids}
    torch.save({"ent_ids": names, "rel_ids": {}}, data / "maps.pt")
    for split, rows in (("train", range(0, 60)), ("dev", range(60, 90)), ("test", range(90, 120))):
        with open(data / f"{split}-ents-class.txt", "w") as f:
            for r in rows:
                f.write(f"ent{int(ids[r])} class{int(labels[r])}\n")
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "node_classification", "with", "dataset=toy", "checkpoint=7",
           f"data_root={tmp_path / 'data'}"]
    proc = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    log = proc.stderr + proc.stdout
    for needle in ("Loaded 120 embeddings with dim=16", "Best regularization coefficient:", "Test accuracy_score: 1.000",
                   "Test balanced_accuracy_score: 1.000"):
        assert needle in log, (needle, log[-1500:])
    saved = joblib.load(tmp_path / "output" / "classifier-7.joblib")
    assert sorted(saved["id_to_class"].values()) == ["class0", "class1", "class2"]
    pred = saved["model"].predict(emb[90:].numpy())
    assert np.array_equal([saved["id_to_class"][p] for p in pred], [f"class{int(l)}" for l in labels[90:]])


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CLI_ARGS = ["link_prediction", "with", "dataset=umls-synth", "inductive=False", "model=glove-bow", "rel_model=transe", "loss_fn=margin",
            "regularizer=1e-2", "max_len=32", "num_negatives=16", "lr=1e-3", "use_scheduler=False", "batch_size=64", "emb_batch_size=512",
            "eval_batch_size=64", "seed=1"]


def test_link_prediction_cli_under_torchrun_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 train.py link_prediction ...` (gloo here, RCCL on GPUs): train.py
    honours RANK / LOCAL_RANK / WORLD_SIZE -- process group, DistributedDataParallel over DataParallel's slices of the same
    global batch, evaluations sharded over the ranks (each encodes its rows; one all-gather of the counts), rank 0 writes
    the files.  Its final scalars equal, to the last bit, those of a SINGLE process evaluating the checkpoint it saved."""
    import json
    from blp_amd.data import write_synthetic_dataset
    write_synthetic_dataset(str(tmp_path / "data"), "umls-synth", num_entities=135, num_relations=46,
                            num_train=1280, num_valid=160, num_test=160, vocab_size=500, emb_dim=48, seed=0)
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    two = tmp_path / "two"
    two.mkdir()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "train.py"), *CLI_ARGS, "max_epochs=1", f"data_root={tmp_path / 'data'}"]
    proc = subprocess.run(cmd, cwd=two, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    log = proc.stderr + proc.stdout
    assert "2 processes, backend gloo" in log and "Processes used: 2" in log
    assert log.count("test mrr:") == 1, "one voice: only rank 0 logs"
    sharded = json.load(open(two / "output" / "scalars-None.json"))
    state = torch.load(two / "output" / "model-None.pt")
    assert sorted(state) == ["module.embeddings.weight", "module.rel_emb.weight"]  # the wrapper's prefix, as under nn.DataParallel
    assert torch.load(two / "output" / "ent_emb-None.pt").shape == (1, 135, 48)
    assert 0.0 < sharded["train_loss"] < 2.0

    one = tmp_path / "one"
    one.mkdir()
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), *CLI_ARGS, "max_epochs=0", f"data_root={tmp_path / 'data'}",
           f"checkpoint={two / 'output' / 'model-None.pt'}"]
    proc = subprocess.run(cmd, cwd=one, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    single = json.load(open(one / "output" / "scalars-None.json"))
    compared = [k for k in single if k.startswith(("valid_", "test_"))]
    assert len(compared) == 22
    for name in compared:
        assert sharded[name] == single[name], (name, sharded[name], single[name])
    assert torch.equal(torch.load(one / "output" / "ent_emb-None.pt"), torch.load(two / "output" / "ent_emb-None.pt"))
