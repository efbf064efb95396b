"""ORACLE (test infrastructure, NOT product code).

CPU restatement of the torch_geometric==1.6.1 pieces the reference uses
(`GCNConv`, `DeepGraphInfomax`; call sites reactions_infomax.py:60,64-68,251-264).
torch_geometric / torch_scatter are NOT installed and not in /root/reference
(requirements.txt:9-13): PARITY UNPINNED versus PyG itself; semantics restated
from SURVEY.md Appendix C.  Integer outputs (CSR) are pinned against numpy's
stable argsort; float outputs use `index_add_` in edge order (what a CPU
scatter_add computes) in fp64.
"""
import numpy as np
import torch


def gcn_norm_ref(edge_index, num_nodes, dtype=torch.float64):
    """remove self loops, append (i, i) at the END, deg = in-degree by target,
    norm = deg^-1/2[src] * deg^-1/2[dst]."""
    ei = edge_index.to(torch.int64)
    keep = ei[0] != ei[1]
    orig_id = torch.arange(ei.shape[1])[keep]
    ei = ei[:, keep]
    loops = torch.arange(num_nodes, dtype=torch.int64)
    src = torch.cat([ei[0], loops])
    dst = torch.cat([ei[1], loops])
    eid = torch.cat([orig_id, edge_index.shape[1] + loops])
    w = torch.ones(src.numel(), dtype=dtype)
    deg = torch.zeros(num_nodes, dtype=dtype).index_add_(0, dst, w)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    return src, dst, eid, dinv[src] * w * dinv[dst]


def csr_ref(edge_index, num_nodes):
    """Stable sort of the augmented edge list by target -> (rowptr, col, eid)."""
    src, dst, eid, norm = gcn_norm_ref(edge_index, num_nodes)
    order = np.argsort(dst.numpy(), kind='stable')
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst.numpy(), minlength=num_nodes), out=rowptr[1:])
    return rowptr, src.numpy()[order], eid.numpy()[order], norm.numpy()[order]


def gcn_conv_ref(x, edge_index, weight, bias):
    """out[i] = sum_{(j -> i)} norm * (x W)[j] + b, summed in edge-list order per target."""
    n = x.shape[0]
    src, dst, _, norm = gcn_norm_ref(edge_index, n, x.dtype)
    xw = x @ weight
    out = torch.zeros(n, weight.shape[1], dtype=x.dtype).index_add_(0, dst, norm.unsqueeze(1) * xw[src])
    return out + bias if bias is not None else out


def dgi_loss_ref(pos_z, neg_z, summary, weight):
    eps = 1e-15
    pos = torch.sigmoid(pos_z @ (weight @ summary))
    neg = torch.sigmoid(neg_z @ (weight @ summary))
    return -torch.log(pos + eps).mean() - torch.log(1 - neg + eps).mean()
