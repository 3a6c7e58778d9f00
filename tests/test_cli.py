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
