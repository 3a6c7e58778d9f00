"""python train.py [link_prediction|node_classification] with key=value ...

The reference's Sacred CLI (/root/reference/train.py) on the MI355X-native hot path: same commands,
config keys and defaults (train.py:35-54), same outputs (output/model-{id}.pt, ent_emb-{id}.pt,
ents-{id}.pt) and log-scalar names, with
  * models from blp_amd.models (fused HIP in-batch loss on a GPU),
  * evaluation from blp_amd.ranking (fused HIP all-entities ranking, filters as segments of a sorted index),
    sharded over every visible GPU (each device encodes and keeps its rows of the entity table, one
    all-gather of the int32 rank counts):
      - `python train.py ...` on a box with several GPUs -- the reference's launch (one process, nn.DataParallel for
        training, train.py:329-330): the evaluation runs one thread per device (blp_amd.multidevice), no launcher needed;
        config key eval_devices picks the devices (default None = all visible);
      - `python -m torch.distributed.run --nproc-per-node G train.py ...`: one process per device (RANK / LOCAL_RANK /
        WORLD_SIZE are honoured: process group nccl on GPUs, gloo on CPU; DistributedDataParallel in place of
        DataParallel, every rank taking DataParallel's slice of the SAME global batch, so the step is the reference's;
        rank 0 writes the output files),
  * Sacred itself if installed, otherwise blp_amd.sacred_shim.
Extra, optional config keys: data_root (default 'data'), seed (default None = unseeded, like the
reference), amp, device_sampler, eval_dropout (default True: the reference's train-mode table build), eval_devices,
rank_table_dtype.  Extra output: output/scalars-{id}.json (the last value of every logged scalar).
"""
import json
import os
import os.path as osp

import numpy as np
import torch
import torch.distributed as dist
from torch.optim import Adam
from torch.utils.data import DataLoader

try:
    from sacred import Experiment
    from sacred.observers import MongoObserver
except ImportError:  # Sacred is not in this image
    from blp_amd.sacred_shim import Experiment
    MongoObserver = None

from blp_amd import utils
from blp_amd.data import GraphDataset, TextGraphDataset, GloVeTokenizer
from blp_amd.ranking import eval_link_prediction

OUT_PATH = "output/"
# torch.distributed.run's environment (absent: the reference's single process)
WORLD_SIZE = int(os.environ.get("WORLD_SIZE", "1"))
RANK = int(os.environ.get("RANK", "0"))
LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
if torch.cuda.is_available():
    # one device per process under a launcher (more ranks than devices -- a functional run on a small box -- wrap around and
    # exchange through gloo: RCCL refuses two ranks on one device)
    device = torch.device("cuda", LOCAL_RANK % torch.cuda.device_count() if WORLD_SIZE > 1 else 0)
else:
    device = torch.device("cpu")

ex = Experiment()
ex.logger = utils.get_logger()
if MongoObserver is not None and all([os.environ.get("DB_URI"), os.environ.get("DB_NAME")]):
    ex.observers.append(MongoObserver(os.environ["DB_URI"], os.environ["DB_NAME"]))


@ex.config
def config():
    dataset = 'umls'
    inductive = True
    dim = 128
    model = 'blp'
    rel_model = 'transe'
    loss_fn = 'margin'
    encoder_name = 'bert-base-cased'
    regularizer = 0
    max_len = 32
    num_negatives = 64
    lr = 2e-5
    use_scheduler = True
    batch_size = 64
    emb_batch_size = 512
    eval_batch_size = 64
    max_epochs = 40
    checkpoint = None
    use_cached_text = False
    data_root = 'data'
    seed = None
    amp = None  # None | 'fp16' | 'bf16': autocast the training step (BASELINE config 5; not in the reference)
    device_sampler = False  # draw the in-batch negative indices on the GPU (same law as data.py:35-81, other RNG stream)
    rank_table_dtype = None  # None | 'float16' | 'bfloat16': rank the evaluations against a 16-bit copy of the entity table, one
    #                          table pass per eval batch as the reference does (only where the passes then read the 16-bit rows
    #                          themselves: eval_batch_size <= 4, dim 128 / 256 -- the Wikidata5M scripts; elsewhere a log line says
    #                          it is not used).  Changes the INPUT (rounded rows), not the arithmetic; not in the reference
    eval_devices = None  # one process, several GPUs: the devices the evaluations are sharded over, one thread each (None = every
    #                      visible GPU; a list such as [0, 1] or -- two shards on one GPU -- [0, 0]).  Ignored under a launcher
    eval_dropout = True  # the reference builds its entity tables in train mode (train.py:57-121 never calls model.eval()):
    #                      True reproduces that; False puts the encoder in eval mode for the table build (deterministic metrics)


def _linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps):
    """transformers.get_linear_schedule_with_warmup (train.py:337), restated to avoid the import cost."""
    def factor(step):
        if step < num_warmup_steps:
            return step / max(1, num_warmup_steps)
        return max(0.0, (num_training_steps - step) / max(1, num_training_steps - num_warmup_steps))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor)


def _tokenizer(model, encoder_name, data_root):
    if model.startswith('bert') or model == 'blp':
        from transformers import BertTokenizer
        return BertTokenizer.from_pretrained(encoder_name)
    return GloVeTokenizer(osp.join(data_root, 'glove', 'glove.6B.300d-maps.pt'))


def _get_model(model, dim, rel_model, loss_fn, num_entities, num_relations, encoder_name, regularizer, data_root):
    if model in ('glove-bow', 'glove-dkrl') and data_root != 'data':
        from blp_amd import models as m
        emb = osp.join(data_root, 'glove', 'glove.6B.300d.pt')
        if model == 'glove-bow':
            return m.BOW(rel_model, loss_fn, num_relations, regularizer, embeddings=emb)
        return m.DKRL(dim, rel_model, loss_fn, num_relations, regularizer, embeddings=emb)
    return utils.get_model(model, dim, rel_model, loss_fn, num_entities, num_relations, encoder_name, regularizer)


@ex.command
def link_prediction(dataset, inductive, dim, model, rel_model, loss_fn, encoder_name, regularizer, max_len,
                    num_negatives, lr, use_scheduler, batch_size, emb_batch_size, eval_batch_size, max_epochs,
                    checkpoint, use_cached_text, data_root, seed, amp, device_sampler, eval_dropout, rank_table_dtype, eval_devices,
                    _run, _log):
    distributed = WORLD_SIZE > 1
    if distributed:
        _init_process_group(_log)
        if seed is None:  # every rank must see the same shuffle and draw the same negatives: rank 0's seed for all
            drawn = torch.randint(0, 2 ** 31 - 1, (1,), device=device if dist.get_backend() == 'nccl' else 'cpu')
            dist.broadcast(drawn, src=0)
            seed = int(drawn)
    try:
        return _link_prediction(dataset, inductive, dim, model, rel_model, loss_fn, encoder_name, regularizer, max_len,
                                num_negatives, lr, use_scheduler, batch_size, emb_batch_size, eval_batch_size, max_epochs,
                                checkpoint, use_cached_text, data_root, seed, amp, device_sampler, eval_dropout,
                                rank_table_dtype, eval_devices, distributed, _run, _log)
    finally:
        if distributed:
            dist.destroy_process_group()


def _init_process_group(_log):
    """torch.distributed.run started us: nccl (= RCCL over xGMI) when every rank has a GPU of its own, else gloo."""
    own_gpu = device.type == 'cuda' and WORLD_SIZE <= torch.cuda.device_count()
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    dist.init_process_group('nccl' if own_gpu else 'gloo')
    if RANK != 0:
        _log.setLevel('WARNING')  # one voice
    _log.info(f'{WORLD_SIZE} processes, backend {dist.get_backend()}, this one on {device}')


def _load_weights(net, state):
    """load_state_dict that takes a checkpoint saved with or without the wrapper's "module." prefix (the reference saves
    whatever `model.state_dict()` is at the time: prefixed under nn.DataParallel on GPUs, bare on the CPU -- train.py:326-330,340)."""
    wrapped = isinstance(net, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel))
    has_prefix = all(k.startswith('module.') for k in state)
    if wrapped and not has_prefix:
        state = {'module.' + k: v for k, v in state.items()}
    elif not wrapped and has_prefix and state:
        state = {k[len('module.'):]: v for k, v in state.items()}
    net.load_state_dict(state)


def _link_prediction(dataset, inductive, dim, model, rel_model, loss_fn, encoder_name, regularizer, max_len,
                     num_negatives, lr, use_scheduler, batch_size, emb_batch_size, eval_batch_size, max_epochs,
                     checkpoint, use_cached_text, data_root, seed, amp, device_sampler, eval_dropout, rank_table_dtype,
                     eval_devices, distributed, _run, _log):
    if amp not in (None, 'fp16', 'bf16'):
        raise ValueError(f'Unknown amp mode {amp}')
    if rank_table_dtype not in (None, 'float16', 'bfloat16'):
        raise ValueError(f'Unknown rank_table_dtype {rank_table_dtype}')
    # a 16-bit ranking table goes with the reference's pass structure (a table pass per eval batch); everything else is ranked in
    # blocks of 65 536 triples
    table16 = dict(rank_table_dtype=getattr(torch, rank_table_dtype), block_size=eval_batch_size) if rank_table_dtype else {}
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    drop_stopwords = model in {'bert-bow', 'bert-dkrl', 'glove-bow', 'glove-dkrl'}
    prefix = 'ind-' if inductive and model != 'transductive' else ''
    triples_file = osp.join(data_root, dataset, f'{prefix}train.tsv')

    if distributed:  # one process per device: the batch is split over the ranks as DataParallel splits it over its devices
        num_devices = WORLD_SIZE
        if batch_size % num_devices != 0:
            raise ValueError(f'Batch size ({batch_size}) must be a multiple of the number of processes ({num_devices})')
        _log.info(f'Processes used: {num_devices} ({"CUDA devices" if device.type == "cuda" else "CPU"})')
    elif device != torch.device('cpu'):
        num_devices = torch.cuda.device_count()
        if batch_size % num_devices != 0:
            raise ValueError(f'Batch size ({batch_size}) must be a multiple of the number of CUDA devices '
                             f'({num_devices})')
        _log.info(f'CUDA devices used: {num_devices}')
    else:
        num_devices = 1
        _log.info('Training on CPU')
    # the evaluations' devices: under a launcher the process group shards them; in one process, one thread per visible GPU
    eval_on = {}
    if not distributed and device.type == 'cuda':
        if eval_devices is not None:
            eval_on = dict(devices=[eval_devices] if isinstance(eval_devices, int) else list(eval_devices))
        elif torch.cuda.device_count() > 1:
            eval_on = dict(devices=list(range(torch.cuda.device_count())))

    # Under a launcher only rank 0 WRITES the dataset's side files (maps.pt, the tokenised-text cache): the others wait for it and
    # read them (two ranks writing and reading one file at once is a torn read).
    first = RANK == 0
    if distributed and not first:
        dist.barrier()
    if model == 'transductive':
        train_data = GraphDataset(triples_file, num_negatives, write_maps_file=first, num_devices=num_devices)
    else:
        train_data = TextGraphDataset(triples_file, num_negatives, max_len, _tokenizer(model, encoder_name, data_root),
                                      drop_stopwords, write_maps_file=first, use_cached_text=use_cached_text or not first,
                                      num_devices=num_devices)
    if distributed and first:
        dist.barrier()
    if device_sampler and device != torch.device('cpu'):
        train_data.sampler_device = device
    train_loader = DataLoader(train_data, batch_size, shuffle=True, collate_fn=train_data.collate_fn,
                              num_workers=0, drop_last=True)
    train_eval_loader = DataLoader(train_data, eval_batch_size)
    valid_data = GraphDataset(osp.join(data_root, dataset, f'{prefix}dev.tsv'))
    valid_loader = DataLoader(valid_data, eval_batch_size)
    test_data = GraphDataset(osp.join(data_root, dataset, f'{prefix}test.tsv'))
    test_loader = DataLoader(test_data, eval_batch_size)

    # filtering graph over every known triple (train.py:298-302); a sorted index instead of networkx
    train_ent = set(train_data.entities.tolist())
    if dataset != 'Wikidata5M':
        graph = utils.FilterIndex(torch.cat((train_data.triples, valid_data.triples, test_data.triples)),
                                  num_relations=train_data.rel_categories.shape[0], device=device)
        train_val_ent = set(valid_data.entities.tolist()).union(train_ent)
        train_val_test_ent = set(test_data.entities.tolist()).union(train_val_ent)
        val_new_ents = train_val_ent.difference(train_ent)
        test_new_ents = train_val_test_ent.difference(train_val_ent)
    else:
        graph = None
        train_val_ent = set(valid_data.entities.tolist())
        train_val_test_ent = set(test_data.entities.tolist())
        val_new_ents = test_new_ents = None
    _run.log_scalar('num_train_entities', len(train_ent))
    train_ent = torch.tensor(list(train_ent))
    train_val_ent = torch.tensor(list(train_val_ent))
    train_val_test_ent = torch.tensor(list(train_val_test_ent))

    net = _get_model(model, dim, rel_model, loss_fn, len(train_val_test_ent), train_data.num_rels, encoder_name,
                     regularizer, data_root)
    if checkpoint is not None:
        _load_weights(net, torch.load(checkpoint, map_location='cpu'))
    if distributed:
        # (DistributedDataParallel averages the ranks' gradients = the gradient of DataParallel's `net(*data).mean()` over its
        #  replicas' losses; its constructor broadcasts rank 0's parameters)
        net = torch.nn.parallel.DistributedDataParallel(net.to(device), device_ids=[device.index] if device.type == 'cuda' else None)
    elif device != torch.device('cpu'):
        net = torch.nn.DataParallel(net).to(device)

    optimizer = Adam(net.parameters(), lr=lr)
    total_steps = len(train_loader) * max_epochs
    scheduler = _linear_schedule_with_warmup(optimizer, int(0.2 * total_steps), total_steps) if use_scheduler else None
    best_valid_mrr = 0.0
    saved_checkpoint = False
    os.makedirs(OUT_PATH, exist_ok=True)
    checkpoint_file = osp.join(OUT_PATH, f'model-{_run._id}.pt')
    log_every = max(1, int(0.05 * len(train_loader)))
    # amp: the encoder runs under autocast and hands half-precision embeddings to the fused loss, which
    # widens them exactly and accumulates in f32 (blp_inbatch_loss_*_t); fp16 needs loss scaling
    use_amp = amp is not None and device != torch.device('cpu')
    amp_dtype = torch.float16 if amp == 'fp16' else torch.bfloat16
    scaler = torch.amp.GradScaler('cuda', enabled=use_amp and amp == 'fp16')
    single_device = not distributed and (device == torch.device('cpu') or torch.cuda.device_count() <= 1)
    for epoch in range(1, max_epochs + 1):
        train_loss = 0
        for step, data in enumerate(train_loader):
            if distributed:  # DataParallel's scatter, by process: this rank's slice of the global batch (neg_idx is local to it)
                data = tuple(t.chunk(WORLD_SIZE)[RANK].to(device) for t in data)
            with torch.autocast('cuda', dtype=amp_dtype, enabled=use_amp):
                loss = net(*data).mean()
            optimizer.zero_grad()
            # one device: backward() runs its nodes on THIS thread (no hand-over to the engine's per-device worker and back:
            # [measured] compute_loss(...).backward() 82 -> 30 us at the FB15k-237 batch, 62 -> 19 us of it the engine's own
            # cost; bench.py: inbatch_loss.us_per_step_autograd_engine_single_threaded); nn.DataParallel over several devices
            # needs the workers
            with torch.autograd.set_multithreading_enabled(not single_device):
                scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
            if scheduler is not None:
                scheduler.step()
            if distributed:  # the logged loss is the mean over the replicas, as `net(*data).mean()` is in one process
                loss = loss.detach().clone()
                dist.all_reduce(loss)
                loss /= WORLD_SIZE
            train_loss += loss.item()
            if step % log_every == 0:
                _log.info(f'Epoch {epoch}/{max_epochs} [{step}/{len(train_loader)}]: {loss.item():.6f}')
                _run.log_scalar('batch_loss', loss.item())
        _run.log_scalar('train_loss', train_loss / len(train_loader), epoch)

        if dataset != 'Wikidata5M':
            _log.info('Evaluating on sample of training set')
            eval_link_prediction(net, train_eval_loader, train_data, train_ent, epoch, emb_batch_size, _run, _log,
                                 prefix='train', max_num_batches=len(valid_loader), device=device, eval_mode=not eval_dropout, **table16, **eval_on)
        _log.info('Evaluating on validation set')
        val_mrr, _ = eval_link_prediction(net, valid_loader, train_data, train_val_ent, epoch, emb_batch_size,
                                          _run, _log, prefix='valid', device=device, eval_mode=not eval_dropout, **table16, **eval_on)
        if val_mrr > best_valid_mrr:  # best checkpoint by raw validation MRR (the same value on every rank)
            best_valid_mrr = val_mrr
            if RANK == 0:
                torch.save(net.state_dict(), checkpoint_file)
            saved_checkpoint = True

    # the reference reloads model-{id}.pt unconditionally (train.py:378-379) and crashes if no epoch improved on
    # 0.0; without an observer the id is None, so a file of that name may also be a stale one from an earlier
    # run -- only reload what THIS run saved
    if saved_checkpoint:
        if distributed:
            dist.barrier()  # rank 0 has written it
        _load_weights(net, torch.load(checkpoint_file, map_location=device))

    if dataset == 'Wikidata5M':
        graph = utils.FilterIndex(valid_data.triples, num_relations=train_data.rel_categories.shape[0], device=device)
    _log.info('Evaluating on validation set (with filtering)')
    eval_link_prediction(net, valid_loader, train_data, train_val_ent, max_epochs + 1, emb_batch_size, _run, _log,
                         prefix='valid', filtering_graph=graph, new_entities=val_new_ents, device=device,
                         eval_mode=not eval_dropout, **table16, **eval_on)
    if dataset == 'Wikidata5M':
        graph = utils.FilterIndex(test_data.triples, num_relations=train_data.rel_categories.shape[0], device=device)
    _log.info('Evaluating on test set')
    _, ent_emb = eval_link_prediction(net, test_loader, train_data, train_val_test_ent, max_epochs + 1,
                                      emb_batch_size, _run, _log, prefix='test', filtering_graph=graph,
                                      new_entities=test_new_ents, return_embeddings=True, device=device,
                                      eval_mode=not eval_dropout, **table16, **eval_on)

    scalars = {name: values[-1][1] for name, values in getattr(_run, 'scalars', {}).items()}
    if RANK == 0:
        torch.save(ent_emb, osp.join(OUT_PATH, f'ent_emb-{_run._id}.pt'))
        torch.save(train_val_test_ent, osp.join(OUT_PATH, f'ents-{_run._id}.pt'))
        if scalars:
            with open(osp.join(OUT_PATH, f'scalars-{_run._id}.json'), 'w') as f:
                json.dump(scalars, f, indent=1, sort_keys=True)
    return scalars


@ex.command
def node_classification(dataset, checkpoint, data_root, _run, _log):
    """The reference's downstream task on saved embeddings (train.py:408-481): logistic regression on
    output/ent_emb-{checkpoint}.pt; CPU / scikit-learn, outside the HIP hot path (blp_amd.downstream)."""
    from blp_amd.downstream import classify_nodes
    return classify_nodes(dataset, checkpoint, _log, data_root=data_root, output_dir=OUT_PATH)


if __name__ == '__main__':
    ex.run_commandline()
