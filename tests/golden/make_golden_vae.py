"""Golden vectors for reparameterise + KL from the REAL in-tree reference twin
paccmann_polymer/topologically_regularized_models/experiments/cora/vae.py (VAE.reparameterize :31-34,
the KLD term of VAE.loss :50).  Run only in the build container (needs /root/reference):

    python tests/golden/make_golden_vae.py

Only numbers are stored (inputs, the noise the reference drew, outputs, autograd gradients)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
from paccmann_polymer.topologically_regularized_models.experiments.cora.vae import VAE  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    vae = VAE(number_features=8)
    for B, Z, seed in ((1, 20, 0), (16, 256, 1), (37, 5, 2)):
        rng = np.random.default_rng(seed)
        mu = torch.tensor(rng.standard_normal((B, Z)).astype(np.float32), requires_grad=True)
        logvar = torch.tensor((rng.standard_normal((B, Z)) * 0.7).astype(np.float32), requires_grad=True)
        torch.manual_seed(100 + seed)
        z = vae.reparameterize(mu, logvar)                       # draws eps = randn_like(std) from the seeded generator
        torch.manual_seed(100 + seed)
        eps = torch.randn_like(mu)                               # the same draw, recorded as an input
        x = torch.rand(B, 8)
        recon = torch.rand(B, 8)
        _, (bce, kld) = vae.loss(recon, x, mu, logvar)           # KLD of :50
        w = torch.tensor(rng.standard_normal((B, Z)).astype(np.float32))
        ((z * w).sum() + 0.3 * kld).backward()
        np.savez(os.path.join(HERE, f'reparam_kl_B{B}_Z{Z}.npz'), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
                 eps=eps.numpy(), z=z.detach().numpy(), kld=np.float32(kld.item()), w=w.numpy(),
                 dmu=mu.grad.numpy(), dlogvar=logvar.grad.numpy())
        print(f'reparam_kl_B{B}_Z{Z}', float(kld))


if __name__ == '__main__':
    main()
