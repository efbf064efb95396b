"""The graph-regularised SMILES-VAE training step — mirror of
`experiments/chem_reactions/reactions_vae.py::train` (:28-116).

`train(...)` keeps the reference signature and per-bundle semantics (one bundle
of <= 21 SMILES per optimizer step).  `train_step_bundles(...)` is the batched
scale-up (BASELINE configs 2/5): K bundles in one step, loss =
decoder_loss + gamma * sum_b graph_loss(z_b, G_b); with one bundle it is exactly
the reference step.  As in the reference the KL term is computed but not added
(`kl_scale` unused at :85-93) unless `use_kl=True`.
"""
import logging
from time import time
from typing import Optional, Sequence

import numpy as np
import torch

from .data_prep import (hop_distance_matrices_device, hop_distance_matrix, hop_distances_launch, hop_inputs_host,
                        packed_preparation_host, packed_preparation_to_device, packed_sequential_data_preparation)
from .graph_constrained_loss import graph_loss, graph_loss_segmented
from .paccmann_encoder import get_device

logger = logging.getLogger(__name__)


def kl_schedule(epoch, _iter, n_iter, growth_rate=0.0015):
    """reactions_vae.py:85-88."""
    return 1 / (1 + np.exp((6 - growth_rate * epoch + (_iter / n_iter))))


def train_step_bundles(model, optimizer, enc_seq, dec_seq, tgt_seq, g_dists, seg_sizes, graph_gamma,
                       eps=None, use_kl=False, kl_scale=0.0, grad_sync=None):
    """One optimizer step over K bundles resident on the device.
    `g_dists`: flat concatenation (or list) of the bundles' hop matrices; `seg_sizes` their node counts.
    `grad_sync`: optional callable run between backward and the optimizer step (data-parallel all-reduce).
    Returns (loss, decoder_loss, gr_loss) as 0-dim device tensors (no host sync)."""
    optimizer.zero_grad()
    decoder_loss, mu, logvar, z = model.train_step(enc_seq, dec_seq, tgt_seq, eps)
    z = z.squeeze(0)
    gr_loss = graph_loss_segmented(z, g_dists, seg_sizes)
    loss = decoder_loss + graph_gamma * gr_loss
    if use_kl:
        loss = loss + kl_scale * model.gru_vae.last_kl
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    return loss.detach(), decoder_loss.detach(), gr_loss.detach()


def train(epoch, model, train_loader, optimizer, scheduler, graph_gamma, growth_rate=0.0015, writer=None,
          verbose=False, distance_order='networkx'):
    """Reference reactions_vae.py:28-116: one bundle per step."""
    start_time = time()
    device = get_device()
    model.gru_vae.to(device)
    model.gru_vae.train()
    input_keep, start_index, end_index = 1., 2, 3
    # the reference reads `loss.item()` every step (:105,:108): a host sync per bundle.  Here the three scalars stay
    # on the device and are read once per epoch; the writer receives the same values, later.
    history = []
    kl_scale = 0.0
    n_iter = len(train_loader)
    _iter = 0
    for _iter, data in enumerate(train_loader):
        seqs = data.x
        batch_size = len(seqs)
        model.gru_decoder._update_batch_size(batch_size)
        model.gru_encoder._update_batch_size(batch_size)
        encoder_seq, decoder_seq, target_seq = packed_sequential_data_preparation(
            seqs, input_keep=input_keep, start_index=start_index, end_index=end_index, device=device)
        optimizer.zero_grad()
        decoder_loss, mu, logvar, z = model.train_step(encoder_seq, decoder_seq, target_seq)
        if batch_size <= 64:                                            # one HIP launch (ppk_hop_distances), same integers
            flat, _, _ = hop_distance_matrices_device([data.edge_index], [batch_size], order=distance_order, device=device)
            dists = flat.view(batch_size, batch_size)
        else:
            dists = torch.tensor(hop_distance_matrix(data.edge_index, batch_size, order=distance_order)).to(device)
        kl_scale = kl_schedule(epoch, _iter, n_iter, growth_rate)       # computed, unused (as in the reference)
        z = z.squeeze()
        gr_loss = graph_loss(z, dists)
        loss = decoder_loss + graph_gamma * gr_loss
        loss.backward()
        optimizer.step()
        history.append(torch.stack([decoder_loss.detach().reshape(()), gr_loss.detach().reshape(()), loss.detach().reshape(())]))
    values = torch.stack(history).cpu().tolist() if history else []      # the epoch's only device-to-host read
    if writer is not None:
        for i, (dl, gl, l) in enumerate(values):
            step = i + epoch * n_iter
            writer.add_scalar('recon_loss', dl, step)
            writer.add_scalar('graph_loss', gl, step)
            writer.add_scalar('loss', l, step)
    train_loss = sum(v[2] for v in values)
    if scheduler is not None:
        scheduler.step()
    logger.info(f"Learning rate {optimizer.param_groups[0]['lr']}\tkl_scale {kl_scale}")
    logger.info(f'{epoch}\t{train_loss / max(_iter, 1)}\t{time() - start_time}')
    return train_loss / max(_iter + 1, 1)


def _bundle_groups(train_loader, bundles_per_step):
    group = []
    for data in train_loader:
        group.append(data)
        if len(group) == bundles_per_step:
            yield group
            group = []
    if group:
        yield group


def _prepare_group_host(group):
    """Everything of one optimizer step that is host work (runs on the worker thread): packing index arithmetic of
    all SMILES of the group's bundles (reference utils.py:35-98) and the flat edge arrays of its hop-distance
    matrices (reactions_vae.py:79-83), in page-locked memory so the copies are asynchronous."""
    seqs = [s for data in group for s in data.x]
    sizes = [len(data.x) for data in group]
    if max(sizes) > 64:
        raise ValueError('train_bundled: bundles of at most 64 SMILES (the reference samples <= 21, utils.py:156-196)')
    packed = packed_preparation_host(seqs, pin=True)
    hops = hop_inputs_host([data.edge_index for data in group], sizes, pin=True)
    return packed, hops, len(seqs)


def train_bundled(epoch, model, train_loader, optimizer, scheduler, graph_gamma, bundles_per_step=128, growth_rate=0.0015,
                  writer=None, verbose=False, distance_order='networkx', use_kl=False, grad_sync=None):
    """Epoch driver of the batched scale-up (BASELINE configs[1] / [4]): the reference loader's bundles
    (`Data(x = list of token tensors, edge_index)`, reactions_vae.py:52-59) are taken `bundles_per_step` at a time
    and each group is ONE optimizer step of `train_step_bundles` — decoder loss over all its SMILES + gamma * the sum
    of the per-bundle graph losses, i.e. `bundles_per_step` reference steps with gradient accumulation.  With
    bundles_per_step == 1 it is the reference's loop (`train`).

    Host / device overlap (after the kernels, host Python is the critical path): a worker thread
    packs group t + 1 and builds its hop-matrix inputs into pinned buffers while the GPU runs group t; the main
    thread only issues asynchronous copies and launches.  Nothing inside the epoch reads the device: the three loss
    scalars of every step stay there and are read once at the end (the writer gets the same values, later).
    Returns the mean loss per step like `train`."""
    from concurrent.futures import ThreadPoolExecutor
    start_time = time()
    device = get_device()
    model.gru_vae.to(device)
    model.gru_vae.train()
    history = []
    groups = _bundle_groups(train_loader, int(bundles_per_step))
    n_iter = max(1, -(-len(train_loader) // int(bundles_per_step))) if hasattr(train_loader, '__len__') else 1
    kl_scale = 0.0
    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = next(groups, None)
        pending = pool.submit(_prepare_group_host, nxt) if nxt is not None else None
        _iter = 0
        while pending is not None:
            packed, hops, batch_size = pending.result()
            nxt = next(groups, None)
            pending = pool.submit(_prepare_group_host, nxt) if nxt is not None else None       # overlaps this step's GPU work
            model.gru_decoder._update_batch_size(batch_size)
            model.gru_encoder._update_batch_size(batch_size)
            encoder_seq, decoder_seq, target_seq = packed_preparation_to_device(packed, device, non_blocking=True)
            g_flat, sizes, _ = hop_distances_launch(hops, distance_order, device, non_blocking=True)
            kl_scale = kl_schedule(epoch, _iter, n_iter, growth_rate)
            loss, decoder_loss, gr_loss = train_step_bundles(model, optimizer, encoder_seq, decoder_seq, target_seq, g_flat, sizes,
                                                             graph_gamma, use_kl=use_kl, kl_scale=kl_scale, grad_sync=grad_sync)
            history.append(torch.stack([decoder_loss.reshape(()), gr_loss.reshape(()), loss.reshape(())]))
            _iter += 1
    values = torch.stack(history).cpu().tolist() if history else []          # the epoch's only device-to-host read
    if writer is not None:
        for i, (dl, gl, l) in enumerate(values):
            step = i + epoch * n_iter
            writer.add_scalar('recon_loss', dl, step)
            writer.add_scalar('graph_loss', gl, step)
            writer.add_scalar('loss', l, step)
    train_loss = sum(v[2] for v in values)
    if scheduler is not None:
        scheduler.step()
    logger.info(f"Learning rate {optimizer.param_groups[0]['lr']}\tkl_scale {kl_scale}")
    logger.info(f'{epoch}\t{train_loss / max(len(values), 1)}\t{time() - start_time}')
    return train_loss / max(len(values), 1)


def export_latents(model, pairs: Sequence[Sequence[torch.Tensor]], device=None, pairs_per_call: int = 2048):
    """Downstream latent export (reference reactions_vae.py:229-260: for every item of the downstream dataset,
    `z = model.encode(pack(seqs))` with batch size 2 and `z.view(1, -1)` -> one row of [len(seqs) * Z]).
    The reference encodes one item per call; rows are independent, so `pairs_per_call` items go through the
    sequence kernels together and the rows are cut afterwards — same values, one launch chain per chunk.
    `pairs`: sequence of items, each a sequence of 1-D token tensors (all items the same count).
    Returns a numpy array [n_items, count * Z]."""
    device = get_device() if device is None else device
    if len(pairs) == 0:
        return np.zeros((0, 0), dtype=np.float32)
    count = len(pairs[0])
    if any(len(p) != count for p in pairs):
        raise ValueError()
    rows = []
    for c0 in range(0, len(pairs), pairs_per_call):
        chunk = pairs[c0:c0 + pairs_per_call]
        seqs = [s for item in chunk for s in item]
        model.update_batch_size(len(seqs))
        encoder_seq, _, _ = packed_sequential_data_preparation(seqs, input_keep=1., start_index=2, end_index=3, device=device)
        mu = model.encode(encoder_seq)
        rows.append(mu.reshape(len(chunk), -1))
    return torch.cat(rows).cpu().numpy()
