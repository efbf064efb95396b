"""Oracle restatement of reparameterise + KL (oracle/stackgru_ref.py) against golden vectors from the REAL
in-tree reference twin (cora/vae.py:31-34,50; tests/golden/make_golden_vae.py)."""
import glob
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FILES = sorted(glob.glob(os.path.join(GOLDEN, 'reparam_kl_*.npz')))


def test_fixtures_present():
    assert len(FILES) == 3


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_matches_reference_twin(path):
    from oracle import stackgru_ref as R
    g = np.load(path)
    mu = torch.tensor(g['mu'], requires_grad=True)
    logvar = torch.tensor(g['logvar'], requires_grad=True)
    vae = R.TeacherVAERef(R.StackGRUEncoderRef(dict(latent_dim=mu.shape[1])), R.StackGRUDecoderRef(dict(latent_dim=mu.shape[1])))
    z = vae.reparameterize(mu, logvar, torch.tensor(g['eps']))
    kld = R.kl_divergence(mu, logvar)
    ((z * torch.tensor(g['w'])).sum() + 0.3 * kld).backward()
    assert np.allclose(z.detach().numpy(), g['z'], rtol=1e-6, atol=1e-7)
    assert abs(float(kld) - float(g['kld'])) <= 1e-6 * abs(float(g['kld']))
    assert np.allclose(mu.grad.numpy(), g['dmu'], rtol=1e-6, atol=1e-7)
    assert np.allclose(logvar.grad.numpy(), g['dlogvar'], rtol=1e-5, atol=1e-6)
