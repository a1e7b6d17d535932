"""Trainer(strategy="ddp"): world-size-2 gloo run on CPU -- replicas stay identical and the averaged gradient step equals
the single-process step on the union of the two shards (pure-torch loss: the FEM kernels need a GPU)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _module():
    from diffnet_amd.base import PDE

    class M(PDE):
        def training_step(self, batch, idx):
            x, y = batch
            return ((self.network(x) - y) ** 2).mean()

        def configure_optimizers(self):
            return [torch.optim.SGD(self.network.parameters(), lr=0.1)], []

    torch.manual_seed(3)
    return M(nn.Sequential(nn.Linear(4, 8), nn.Tanh(), nn.Linear(8, 1)))


def _data():
    g = torch.Generator().manual_seed(9)
    return torch.randn(16, 4, generator=g), torch.randn(16, 1, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffnet_amd.datasets import DeviceLoader
    from diffnet_amd.trainer import Trainer
    x, y = _data()
    ds = torch.utils.data.TensorDataset(x, y)
    loader = DeviceLoader(ds, batch_size=4, device="cpu", rank=rank, world=world)
    m = _module()
    keys_before = sorted(m.state_dict().keys())
    Trainer(max_epochs=2, device="cpu", strategy="ddp").fit(m, loader)
    # DistributedDataParallel adds a "module." level; the module keeps the reference's checkpoint keys and loads them back
    assert sorted(m.state_dict().keys()) == keys_before, (sorted(m.state_dict().keys()), keys_before)
    m.load_state_dict(m.state_dict())
    torch.save([p.detach().clone() for p in m.network.parameters()], os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_ddp_replicas_match_the_large_batch_step(tmp_path):
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = (torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in (0, 1))
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    # single process: batches of 8 made of the two ranks' 4-sample shards (mean of equal-size shard means = batch mean)
    from diffnet_amd.trainer import Trainer
    x, y = _data()
    batches = []
    for b in range(2):
        idx = torch.cat([torch.arange(r, 16, 2)[b * 4:(b + 1) * 4] for r in (0, 1)])
        batches.append((x[idx], y[idx]))
    m = _module()
    Trainer(max_epochs=2, device="cpu").fit(m, batches)
    for a, b in zip(p0, m.network.parameters()):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_multi_optimizer_loop_passes_optimizer_idx():
    """Lightning 1.x semantics the reference's FSDT script relies on (e1_plate_bending_fsdt.py:
    `training_step(self, batch, batch_idx, optimizer_idx)` returning `loss_vals[optimizer_idx]`, one optimizer per field)."""
    from diffnet_amd.base import PDE
    from diffnet_amd.trainer import Trainer
    seen = []

    class M(PDE):
        def training_step(self, batch, batch_idx, optimizer_idx):
            seen.append(optimizer_idx)
            a, b = self.network[0], self.network[1]
            return [(a - 1.0).pow(2).sum(), (b + 2.0).pow(2).sum()][optimizer_idx]

        def configure_optimizers(self):
            return [torch.optim.SGD([self.network[0]], lr=0.25), torch.optim.SGD([self.network[1]], lr=0.25)], []

    m = M(nn.ParameterList([nn.Parameter(torch.zeros(3)), nn.Parameter(torch.zeros(3))]))
    Trainer(max_epochs=3, device="cpu").fit(m, [(torch.zeros(1), torch.zeros(1))])
    assert seen == [0, 1] * 3
    # each optimizer descended its own loss: a -> 1 (factor 1/2 per step), b -> -2
    np.testing.assert_allclose(m.network[0].detach().numpy(), 1.0 - 0.5 ** 3, rtol=1e-6)
    np.testing.assert_allclose(m.network[1].detach().numpy(), -2.0 * (1.0 - 0.5 ** 3), rtol=1e-6)

    class M2(M):                                   # a two-argument training_step keeps working with several optimizers
        def training_step(self, batch, batch_idx):
            return (self.network[0] - 1.0).pow(2).sum() + (self.network[1] + 2.0).pow(2).sum()

    m2 = M2(nn.ParameterList([nn.Parameter(torch.zeros(3)), nn.Parameter(torch.zeros(3))]))
    Trainer(max_epochs=1, device="cpu").fit(m2, [(torch.zeros(1), torch.zeros(1))])
    assert torch.isfinite(m2.network[0]).all()
