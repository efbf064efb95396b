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

from .data_prep import hop_distance_matrices_device, hop_distance_matrix, packed_sequential_data_preparation
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
    train_loss = 0
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
        if writer is not None:
            step = _iter + epoch * n_iter
            writer.add_scalar('recon_loss', decoder_loss, step)
            writer.add_scalar('graph_loss', gr_loss, step)
            writer.add_scalar('loss', loss.item(), step)
        train_loss += loss.item()
    if scheduler is not None:
        scheduler.step()
    logger.info(f"Learning rate {optimizer.param_groups[0]['lr']}\tkl_scale {kl_scale}")
    logger.info(f'{epoch}\t{train_loss / max(_iter, 1)}\t{time() - start_time}')
    return train_loss / max(_iter + 1, 1)
