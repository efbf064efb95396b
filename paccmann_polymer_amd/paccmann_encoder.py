"""Facade over the HIP StackGRU VAE — mirror of the reference class
`paccmann_polymer/models/paccmann_encoder.py::Encoder` (:15-169): same
constructor arguments, attributes (`gru_encoder`, `gru_decoder`, `gru_vae`,
`smiles_language`, `transform`, `device`) and methods.  `train_step` (:141-169)
is the hot path.  `decode` / `reconstruct` (:115-139) run token generation through the single-step
kernels (`TeacherVAE.generate`, build-defined: the upstream body is not in the reference tree); the SMILES
text side uses a minimal character-level `TokenLanguage` in place of pytoda's `SMILESLanguage`.
"""
import json
from typing import Optional, Union

import torch

from .stack_gru_vae import SamplingSearch, StackGRUDecoder, StackGRUEncoder, TeacherVAE


def get_device():
    """paccmann_chemistry.utils.get_device: cuda if available (HIP on ROCm)."""
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


class TokenLanguage:
    """Minimal stand-in for pytoda's SMILESLanguage (absent offline): a
    token -> index table loaded from JSON ({"token_to_index": {...}}), character-level."""

    def __init__(self, token_to_index=None):
        self.token_to_index = dict(token_to_index or {})
        self.padding_index, self.unknown_index, self.start_index, self.stop_index = 0, 1, 2, 3

    @classmethod
    def load(cls, path):
        with open(path, 'r') as f:
            raw = json.load(f)
        return cls(raw.get('token_to_index', raw))

    def smiles_to_token_indexes(self, smiles: str):
        return [self.start_index] + [self.token_to_index.get(c, self.unknown_index) for c in smiles] + [self.stop_index]

    def token_indexes_to_smiles(self, token_indexes):
        index_to_token = {v: k for k, v in self.token_to_index.items()}
        special = (self.padding_index, self.start_index, self.stop_index)
        return ''.join(index_to_token.get(int(t), '?') for t in token_indexes if int(t) not in special)


class Encoder:

    def __init__(
        self,
        params_file: Union[str, dict],
        lang_file: Optional[str] = None,
        weights_file: Optional[str] = None,
        batch_size: int = 1,
        batch_mode: str = 'Padded',
        seed: int = 11254,
    ):
        self.seed = seed
        self.load_pretrained_paccmann(params_file, lang_file, weights_file, batch_size, batch_mode)

    def set_seed(self, seed: int):
        self.seed = seed

    def __call__(self, smiles):
        return self.encode_node_smiles(smiles)

    def load_pretrained_paccmann(self, params_file, lang_file, weights_file, batch_size, batch_mode):
        params = dict()
        if isinstance(params_file, dict):
            params.update(params_file)
        else:
            with open(params_file, 'r') as f:
                params.update(json.load(f))
        params['batch_mode'] = batch_mode
        params['batch_size'] = batch_size
        self.params = params
        self.selfies = params.get('selfies', False)
        self.device = get_device()
        self.smiles_language = TokenLanguage.load(lang_file) if lang_file else TokenLanguage()
        self.gru_encoder = StackGRUEncoder(params).to(self.device)
        self.gru_decoder = StackGRUDecoder(params).to(self.device)
        self.gru_vae = TeacherVAE(self.gru_encoder, self.gru_decoder).to(self.device)
        if weights_file:
            self.update_vae_weights(weights_file)
        self.gru_vae.eval()
        self.transform = lambda s: torch.tensor(
            self.smiles_language.smiles_to_token_indexes(s), dtype=torch.int64, device=self.device)

    def update_vae_weights(self, weights_file: str):
        self.gru_vae.load_state_dict(torch.load(weights_file, map_location=self.device, weights_only=True))

    def update_batch_size(self, batch_size: int):
        self.gru_decoder._update_batch_size(batch_size)
        self.gru_encoder._update_batch_size(batch_size)

    def encode(self, token_seq) -> torch.Tensor:
        with torch.no_grad():
            mu, logvar = self.gru_vae.encode(token_seq)
            return mu

    def encode_node_smiles(self, smiles: str) -> torch.Tensor:
        from torch.nn.utils.rnn import pack_sequence
        tokens = self.transform(smiles).long()
        return self.encode(pack_sequence([tokens.cpu()])).detach().squeeze()

    def decode(self, latent_z, generate_len=100, search=None):
        """Reference :115-131: seeds the torch RNG with `self.seed`, generates from `latent_z [B, Z]`, returns one
        token list per sample starting with the start index."""
        latent_z = latent_z.view(1, latent_z.shape[0], latent_z.shape[1]).float()
        torch.manual_seed(self.seed)
        molecule_iter = self.gru_vae.generate(
            latent_z,
            prime_input=torch.tensor([self.smiles_language.start_index]).to(self.device),
            end_token=torch.tensor([self.smiles_language.stop_index]).to(self.device),
            generate_len=generate_len,
            search=search if search is not None else SamplingSearch())
        return [[self.smiles_language.start_index] + m.cpu().detach().tolist() for m in molecule_iter]

    def reconstruct(self, smiles):
        """Reference :133-139 (`smiles` is a packed token batch, as there)."""
        mu, logvar = self.gru_vae.encode(smiles)
        tokens = self.decode(mu)
        rec_smils = [self.smiles_language.token_indexes_to_smiles(t) for t in tokens]
        return rec_smils, mu, logvar

    def train_step(self, input_seq, decoder_seq, target_seq, eps=None):
        """Reference :141-169.  Returns (decoder_loss, mu, logvar, latent_z[1, B, Z]).
        `eps` optionally injects the reparameterisation noise (parity tests)."""
        mu, logvar = self.gru_vae.encode(input_seq)
        latent_z = self.gru_vae.reparameterize(mu, logvar, eps).unsqueeze(0)
        decoder_loss = self.gru_vae.decode(latent_z, decoder_seq, target_seq)
        return decoder_loss, mu, logvar, latent_z
