"""ORACLE (test infrastructure, NOT product code).

CPU restatement, from torch primitives, of the stack-augmented GRU VAE that the
reference's hot path calls in the third-party package
`paccmann_chemistry@0.0.4` (NOT present in /root/reference nor in this
container — `setup.py:18`, `requirements.txt:5`).  Reference call sites:
  paccmann_polymer/models/paccmann_encoder.py:52-59  (construction)
  paccmann_polymer/models/paccmann_encoder.py:166-168 (encode / reparameterize / decode)
  .../experiments/chem_reactions/reactions_vae.py:62-63,74-76 (train step)
  .../experiments/latent_space_exploration/latent_space_map_fns.py:284-306
        (in-tree re-implementation of the decode loop: attribute names, shapes)

PARITY UNPINNED versus upstream paccmann_chemistry: the package cannot be
imported or fetched offline and the reference holds no fixtures for it; the
published algorithm is restated per SURVEY.md Appendix B.  What IS pinned:
the GRU cell arithmetic is torch's own `nn.GRU` (authoritative in this
container), the loss is torch's `CrossEntropyLoss(ignore_index=0)`, packing is
torch's `pack_sequence`; the product's HIP path is compared against THIS
module in fp32 and fp64 on identical weights, tokens and injected eps.

Documented choices (switchable where upstream behaviour is not remembered):
  * `mu` / `logvar` are returned in ORIGINAL sample order (rows align with the
    relation-graph rows, reactions_vae.py:74-91);
  * the decoder re-orders the latent rows with the PackedSequence's
    `sorted_indices` so that row b of the packed batch decodes sample b's z.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import PackedSequence

DEFAULT_PARAMS = dict(
    vocab_size=64, embedding_size=128, rnn_cell_size=512, n_layers=2, stack_width=50,
    stack_depth=50, latent_dim=256, dropout=0.0, bidirectional=False, use_stack=True,
    batch_mode='packed', padding_index=0, batch_size=1,
)


class StackGRURef(nn.Module):
    """Appendix B: one step = embedding -> stack update from hidden[-1] ->
    nn.GRU (seq_len 1, n_layers deep)."""

    def __init__(self, params):
        super().__init__()
        p = dict(DEFAULT_PARAMS)
        p.update(params)
        self.params = p
        self.vocab_size = p['vocab_size']
        self.embedding_size = p['embedding_size']
        self.rnn_cell_size = p['rnn_cell_size']
        self.n_layers = p['n_layers']
        self.stack_width = p['stack_width']
        self.stack_depth = p['stack_depth']
        self.use_stack = p.get('use_stack', True)
        self.embedding = nn.Embedding(self.vocab_size, self.embedding_size, padding_idx=p['padding_index'])
        in_size = self.embedding_size + (self.stack_width if self.use_stack else 0)
        self.gru = nn.GRU(in_size, self.rnn_cell_size, self.n_layers, dropout=p.get('dropout', 0.0))
        self.stack_controls_layer = nn.Linear(self.rnn_cell_size, 3)
        self.stack_input_layer = nn.Linear(self.rnn_cell_size, self.stack_width)

    def init_state(self, batch, dtype, device):
        h = torch.zeros(self.n_layers, batch, self.rnn_cell_size, dtype=dtype, device=device)
        s = torch.zeros(batch, self.stack_depth, self.stack_width, dtype=dtype, device=device)
        return h, s

    # upstream plumbing read by the reference's in-tree callers (latent_space_map_fns.py:284,289)
    def _update_batch_size(self, batch_size):
        self.batch_size = int(batch_size)

    @property
    def init_stack(self):
        w = self.embedding.weight
        return self.init_state(getattr(self, 'batch_size', 1), w.dtype, w.device)[1]

    @property
    def init_hidden(self):
        w = self.embedding.weight
        return self.init_state(getattr(self, 'batch_size', 1), w.dtype, w.device)[0]

    def stack_augmentation(self, input_val, prev_stack, controls):
        # input_val [B, W], prev_stack [B, D, W], controls [B, 3] = (push, pop, no-op)
        B = prev_stack.shape[0]
        controls = controls.view(B, 3, 1, 1)
        zeros_at_bottom = torch.zeros(B, 1, self.stack_width, dtype=prev_stack.dtype, device=prev_stack.device)
        a_push, a_pop, a_no_op = controls[:, 0], controls[:, 1], controls[:, 2]
        stack_down = torch.cat((prev_stack[:, 1:], zeros_at_bottom), dim=1)
        stack_up = torch.cat((input_val.unsqueeze(1), prev_stack[:, :-1]), dim=1)
        return a_no_op * prev_stack + a_push * stack_up + a_pop * stack_down

    def forward(self, token, hidden, stack, drop=None):
        # token [1, B] int64, hidden [L, B, H], stack [B, D, W]; drop [L-1, B, H] injected dropout factors (None: self.gru as is)
        emb = self.embedding(token)                                       # [1, B, E]
        if self.use_stack:
            top = hidden[-1]
            controls = F.softmax(self.stack_controls_layer(top), dim=-1)
            s_in = torch.tanh(self.stack_input_layer(top))
            stack = self.stack_augmentation(s_in, stack, controls)
            emb = torch.cat((emb, stack[:, 0, :].unsqueeze(0)), dim=2)
        if drop is None:
            out, hidden = self.gru(emb, hidden)
        else:
            out, hidden = self.gru_layers_with_factors(emb, hidden, drop)
        return out, hidden, stack

    def gru_layers_with_factors(self, x, hidden, drop):
        """nn.GRU(seq_len = 1, dropout = p) in training mode with the dropout FACTORS injected (SURVEY landmine 10: the
        device and the CPU cannot share noise): drop [L-1, B, H] holds 0 or 1 / (1 - p); torch applies dropout to the
        output of every layer except the last, on the way to the next layer only.  The cell is torch's own
        (`torch.gru_cell` on the module's weights); with all factors 1 this equals `self.gru(x, hidden)`
        (tests/test_oracle_dropout.py)."""
        inp = x.squeeze(0)
        new_h = []
        for l in range(self.n_layers):
            w_ih, w_hh = getattr(self.gru, f'weight_ih_l{l}'), getattr(self.gru, f'weight_hh_l{l}')
            b_ih, b_hh = getattr(self.gru, f'bias_ih_l{l}'), getattr(self.gru, f'bias_hh_l{l}')
            h = torch.gru_cell(inp, hidden[l], w_ih, w_hh, b_ih, b_hh)
            new_h.append(h)
            inp = h * drop[l] if l < self.n_layers - 1 else h
        hidden = torch.stack(new_h, dim=0)
        return hidden[-1].unsqueeze(0), hidden

    drop_factors = None       # [L-1, NT, H] in packed-row order: injected dropout factors for the next run_packed calls

    def run_packed(self, packed: PackedSequence, hidden, stack):
        """Steps through a PackedSequence.  `hidden`/`stack` are in SORTED order.
        Returns (outputs [sum T, H] packed order, final_hidden [L, B, H] sorted order)."""
        data, bs = packed.data, packed.batch_sizes
        outs = []
        final = torch.zeros_like(hidden)
        off = 0
        prev_b = int(bs[0])
        for t in range(len(bs)):
            b = int(bs[t])
            if b < prev_b:
                final[:, b:prev_b] = hidden[:, b:prev_b]
                hidden = hidden[:, :b]
                stack = stack[:b]
            tok = data[off:off + b].unsqueeze(0)
            drop = None if self.drop_factors is None else self.drop_factors[:, off:off + b]
            out, hidden, stack = self.forward(tok, hidden, stack, drop)
            outs.append(out.squeeze(0))
            off += b
            prev_b = b
        final[:, :prev_b] = hidden
        return torch.cat(outs, dim=0), final


class StackGRUEncoderRef(StackGRURef):

    def __init__(self, params):
        super().__init__(params)
        self.latent_dim = self.params['latent_dim']
        self.hidden_to_mu = nn.Linear(self.rnn_cell_size, self.latent_dim)
        self.hidden_to_logvar = nn.Linear(self.rnn_cell_size, self.latent_dim)

    def encoder_train_step(self, packed: PackedSequence):
        B = int(packed.batch_sizes[0])
        p = next(self.parameters())
        hidden, stack = self.init_state(B, p.dtype, p.device)
        _, final = self.run_packed(packed, hidden, stack)
        top = final[-1]                                                   # sorted order
        if packed.unsorted_indices is not None:
            top = top.index_select(0, packed.unsorted_indices)            # original order
        return self.hidden_to_mu(top), self.hidden_to_logvar(top)


class StackGRUDecoderRef(StackGRURef):

    def __init__(self, params):
        super().__init__(params)
        self.latent_dim = self.params['latent_dim']
        self.latent_to_hidden = nn.Linear(self.latent_dim, self.rnn_cell_size)
        self.output_layer = nn.Linear(self.rnn_cell_size, self.vocab_size)

    def decoder_train_step(self, latent_z, packed_in: PackedSequence, packed_tgt: PackedSequence):
        # latent_z [1 or L, B, Z] in ORIGINAL sample order
        if latent_z.shape[0] == 1:
            latent_z = latent_z.repeat(self.n_layers, 1, 1)
        if packed_in.sorted_indices is not None:
            latent_z = latent_z.index_select(1, packed_in.sorted_indices)
        hidden = self.latent_to_hidden(latent_z)
        B = hidden.shape[1]
        _, stack = self.init_state(B, hidden.dtype, hidden.device)
        outs, _ = self.run_packed(packed_in, hidden, stack)
        logits = self.output_layer(outs)                                  # [sum T, V]
        return F.cross_entropy(logits, packed_tgt.data, ignore_index=0)


class TeacherVAERef(nn.Module):

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def encode(self, input_seq):
        return self.encoder.encoder_train_step(input_seq)

    def reparameterize(self, mu, logvar, eps=None):
        std = torch.exp(0.5 * logvar)
        if eps is None:
            eps = torch.randn_like(std)
        return eps.mul(std).add_(mu)

    def decode(self, latent_z, input_seq, target_seq):
        return self.decoder.decoder_train_step(latent_z, input_seq, target_seq)


def kl_divergence(mu, logvar):
    """cora/vae.py:50 (same in topo_vae.py:115, topo_vae_mnist.py:162)."""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())


def train_step_ref(vae: TeacherVAERef, enc_seq, dec_seq, tgt_seq, eps):
    """paccmann_encoder.py:141-169 with injected eps."""
    mu, logvar = vae.encode(enc_seq)
    latent_z = vae.reparameterize(mu, logvar, eps).unsqueeze(0)
    decoder_loss = vae.decode(latent_z, dec_seq, tgt_seq)
    return decoder_loss, mu, logvar, latent_z
