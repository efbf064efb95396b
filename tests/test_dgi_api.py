"""`DeepGraphInfomax.test` (torch_geometric API named in SURVEY §8b): host-side logistic-regression probe on latents."""
import numpy as np
import torch


def test_dgi_test_scores_a_separable_problem():
    from paccmann_polymer_amd.gcn import DeepGraphInfomax
    model = DeepGraphInfomax(4, encoder=torch.nn.Identity())
    rng = np.random.default_rng(0)
    y = rng.integers(0, 3, size=120)
    z = np.eye(3)[y] * 4.0 + rng.standard_normal((120, 3)) * 0.1
    z = np.concatenate([z, rng.standard_normal((120, 1))], axis=1)
    zt, yt = torch.tensor(z, dtype=torch.float32), torch.tensor(y)
    acc = model.test(zt[:80], yt[:80], zt[80:], yt[80:], max_iter=200)
    assert acc == 1.0
    assert 0.0 <= model.test(zt[:80], yt[:80], zt[80:], (yt[80:] + 1) % 3) <= 0.1
