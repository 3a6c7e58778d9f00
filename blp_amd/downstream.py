"""Node classification on saved entity embeddings -- the reference's downstream command (train.py:408-481): a multinomial
logistic regression on the ``ent_emb-{id}.pt`` / ``ents-{id}.pt`` files a link-prediction run writes, regularisation
strength picked on the dev split, accuracy and balanced accuracy logged, the classifier saved with joblib.  CPU only
(scikit-learn); it never touches the scoring / ranking path, and it is kept so that launch scripts which call
``train.py node_classification with dataset=... checkpoint=...`` keep working (CLI to keep)."""
import os.path as osp

import numpy as np
import torch

from . import utils

SPLITS = ("train", "dev", "test")
C_GRID = tuple(10.0 ** k for k in range(4, -2, -1))  # 1e4 ... 1e-1, the reference's order (ties keep the earlier one)


def load_embeddings(output_dir, checkpoint):
    """(embeddings (n, dim) float array, id -> row map) from the two files of a link-prediction run."""
    emb = torch.load(osp.join(output_dir, f"ent_emb-{checkpoint}.pt"), map_location="cpu")
    emb = emb[0] if isinstance(emb, tuple) else emb
    emb = emb.squeeze().numpy()
    ids = torch.load(osp.join(output_dir, f"ents-{checkpoint}.pt"), map_location="cpu")
    return emb, utils.make_ent2idx(ids, max_ent_id=ids.max()).numpy()


def load_labelled_split(path, ent_ids, ent2idx, classes):
    """``<entity name> <class name>`` lines -> (rows of the embedding table, integer labels); ``classes`` (name -> label)
    grows in order of first appearance across the splits."""
    rows, labels = [], []
    with open(path) as f:
        for line in f:
            name, cls = line.strip().split()
            rows.append(ent2idx[ent_ids[name]])
            labels.append(classes.setdefault(cls, len(classes)))
    return np.asarray(rows, dtype=np.int64), np.asarray(labels)


def _classifier(c):
    import sklearn
    from sklearn.linear_model import LogisticRegression
    major, minor = (int(x) for x in sklearn.__version__.split(".")[:2])
    if (major, minor) >= (1, 5):  # multinomial is the only behaviour for multi-class problems (the keyword warns, then goes)
        return LogisticRegression(C=c, max_iter=1000)
    return LogisticRegression(C=c, multi_class="multinomial", max_iter=1000)


def classify_nodes(dataset, checkpoint, log, data_root="data", output_dir="output"):
    """Fit, select, evaluate, save.  Returns {'best_c', 'dev_accuracy', 'test_accuracy', 'test_balanced_accuracy'}."""
    import joblib
    from sklearn.metrics import accuracy_score, balanced_accuracy_score
    emb, ent2idx = load_embeddings(output_dir, checkpoint)
    log.info(f"Loaded {emb.shape[0]} embeddings with dim={emb.shape[1]}")
    ent_ids = torch.load(osp.join(data_root, dataset, "maps.pt"))["ent_ids"]
    classes, data = {}, {}
    for split in SPLITS:
        rows, labels = load_labelled_split(osp.join(data_root, dataset, f"{split}-ents-class.txt"), ent_ids, ent2idx, classes)
        data[split] = (emb[rows], labels)
    (x_train, y_train), (x_dev, y_dev), (x_test, y_test) = (data[s] for s in SPLITS)

    best_c, best_dev = 0, 0.0
    for c in C_GRID:
        dev_acc = accuracy_score(y_dev, _classifier(c).fit(x_train, y_train).predict(x_dev))
        log.info(f"{c:.3f} - {dev_acc:.3f}")
        if dev_acc > best_dev:
            best_c, best_dev = c, dev_acc
    if best_c == 0:  # (the reference would go on to fit with C = 0 and fail inside scikit-learn, train.py:452-460)
        raise ValueError("node_classification: no regularisation coefficient reached a dev accuracy above 0 -- "
                         "check the class files and the embeddings")
    log.info(f"Best regularization coefficient: {best_c:.4f}")

    x_fit, y_fit = np.concatenate((x_train, x_dev)), np.concatenate((y_train, y_dev))
    model = _classifier(best_c).fit(x_fit, y_fit)
    fit_pred, test_pred = model.predict(x_fit), model.predict(x_test)
    scores = {}
    for metric in (accuracy_score, balanced_accuracy_score):
        log.info(f"Train {metric.__name__}: {metric(y_fit, fit_pred):.3f}")
        scores[metric.__name__] = metric(y_test, test_pred)
        log.info(f"Test {metric.__name__}: {scores[metric.__name__]:.3f}")
    joblib.dump({"model": model, "id_to_class": {label: name for name, label in classes.items()}},
                osp.join(output_dir, f"classifier-{checkpoint}.joblib"))
    return {"best_c": best_c, "dev_accuracy": best_dev, "test_accuracy": scores["accuracy_score"],
            "test_balanced_accuracy": scores["balanced_accuracy_score"]}
