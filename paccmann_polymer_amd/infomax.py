"""The GNN (DeepGraphInfomax) variant of the trainer — mirror of
`experiments/chem_reactions/reactions_infomax.py`:

  :38-83    Encoder(nn.Module): frozen StackGRU-VAE encoder -> GCNConv(256, 512, cached=False) -> PReLU(512)
  :86-146   train(epoch, model, train_loader, optimizer, scheduler, writer=None, verbose=False)
  :251-264  DeepGraphInfomax(hidden_channels=512, encoder, summary=sigmoid(mean_0 z), corruption)

Every bundle's packed token batch goes through the HIP sequence kernels (K1) under
`no_grad` (paccmann_encoder.py:99), the latent means through K6/K7 + the MFMA
GEMM, and `corruption` (utils.py:22-31) re-packs the permuted sequences for the
negative pass exactly as the reference does.
"""
import logging
from time import time

import torch
import torch.nn as nn

from .data_prep import packed_sequential_data_preparation
from .gcn import DeepGraphInfomax, GCNConv, PReLU, corruption, default_summary
from .paccmann_encoder import Encoder as EncoderPaccmann
from .paccmann_encoder import get_device

logger = logging.getLogger(__name__)


class Encoder(nn.Module):
    """reactions_infomax.py:38-83.  `PARAM_FILE` may be a params dict (as `paccmann_encoder.Encoder` accepts) and
    `paccmann_vae` an already built facade (weights shared with a VAE trainer)."""

    def __init__(self, PARAM_FILE=None, LANG_FILE=None, WEIGHT_FILE=None, device=None, in_channels=256,
                 hidden_channels=512, batch_mode='Packed', paccmann_vae=None):
        super().__init__()
        self.paccmann_vae = paccmann_vae if paccmann_vae is not None else EncoderPaccmann(
            PARAM_FILE, LANG_FILE, WEIGHT_FILE, batch_size=1, batch_mode='Packed')
        self.paccmann_vae.gru_vae.encoder.set_batch_mode('packed')
        self.conv = GCNConv(in_channels, hidden_channels, cached=False)
        self.prelu = PReLU(hidden_channels)
        self.to(device if device is not None else get_device())

    def forward(self, x, edge_index):
        x = self.paccmann_vae.encode(x)             # under no_grad: the sequence encoder is frozen (paccmann_encoder.py:99)
        x = self.conv(x, edge_index)
        x = self.prelu(x)
        return x

    def update_batch_size(self, batch_size: int):
        self.paccmann_vae.gru_decoder._update_batch_size(batch_size)
        self.paccmann_vae.gru_encoder._update_batch_size(batch_size)

    def to(self, device):                            # the reference's override: returns None there; returning self is harmless
        self.paccmann_vae.gru_vae.to(device)
        self.conv.to(device)
        self.prelu.to(device)
        return self


def build_infomax(encoder: Encoder, hidden_channels=512):
    """reactions_infomax.py:251-264."""
    return DeepGraphInfomax(hidden_channels=hidden_channels, encoder=encoder, summary=default_summary,
                            corruption=corruption)


def infomax_step(model, optimizer, seqs, edge_index, device=None):
    """One bundle (reactions_infomax.py:116-134); returns the loss as a 0-dim device tensor (no host sync)."""
    device = get_device() if device is None else device
    model.encoder.update_batch_size(len(seqs))
    encoder_seq, _, _ = packed_sequential_data_preparation(seqs, input_keep=1., start_index=2, end_index=3, device=device)
    optimizer.zero_grad()
    pos_z, neg_z, summary = model(encoder_seq, edge_index.to(device))
    loss = model.loss(pos_z, neg_z, summary)
    loss.backward()
    optimizer.step()
    return loss.detach()


def train(epoch, model, train_loader, optimizer, scheduler, writer=None, verbose=False):
    """reactions_infomax.py:86-146.  The per-bundle `loss.item()` of the reference (:137,:140) is a host sync per
    step; the scalars stay on the device and are read once per epoch, the writer gets the same values."""
    start_time = time()
    device = get_device()
    model.to(device)
    model.train()
    history = []
    _iter = 0
    for _iter, data in enumerate(train_loader):
        history.append(infomax_step(model, optimizer, data.x, data.edge_index, device).reshape(()))
    values = torch.stack(history).cpu().tolist() if history else []
    if writer is not None:
        for i, v in enumerate(values):
            writer.add_scalar('loss', v, i + epoch * len(train_loader))
    train_loss = sum(values)
    if scheduler is not None:
        scheduler.step()
    logger.info(f"Learning rate {optimizer.param_groups[0]['lr']}")
    logger.info(f'Epoch: {epoch}\t{train_loss / max(_iter, 1)}\t{time() - start_time}')
    return train_loss / max(_iter + 1, 1)
