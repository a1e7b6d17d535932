"""Minimal fit loop for `PDE` modules where pytorch_lightning is not installed (it is absent on the build and GPU
boxes).  It drives exactly the hooks the reference scripts implement -- `configure_optimizers`, `training_step`
(tensor or {"loss": tensor}), optional `validation_step`, callbacks with `on_train_epoch_end(trainer, module)` -- and
nothing else; checkpointing, loggers and DDP remain Lightning's job when it is available.

`Trainer(graph=True)`: for full-batch problems (one static batch, the shape of the reference's single-instance examples)
the whole iteration -- forward, fused loss kernel, backward, optimizer update -- is captured once into a HIP graph and
replayed, so a step costs one graph launch instead of ~50 kernel launches from Python (the small meshes of
BASELINE configs[0] are launch-bound: tools/bench_graph.py)."""
import inspect

import torch


def _takes_optimizer_idx(module):
    """Lightning 1.x passes `optimizer_idx` to `training_step` when several optimizers are configured and the method accepts
    it (e1_plate_bending_fsdt.py: `training_step(self, batch, batch_idx, optimizer_idx)`)."""
    try:
        params = inspect.signature(module.training_step).parameters
    except (TypeError, ValueError):
        return False
    if "optimizer_idx" in params:
        return True
    positional = [p for p in params.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    return len(positional) >= 3


def _to(batch, device):
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to(b, device) for b in batch)
    return batch


class Trainer:
    def __init__(self, max_epochs=1, device=None, callbacks=(), max_steps=None, graph=False, graph_warmup=3, log_every=1,
                 strategy=None):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.strategy = strategy
        self.graph, self.graph_warmup, self.log_every = graph, graph_warmup, max(1, log_every)
        self.device = torch.device(device) if device is not None else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
        self.callbacks = list(callbacks)
        self.current_epoch = 0
        self.global_step = 0
        self.history = []

    @staticmethod
    def _loss_of(out):
        return out["loss"] if isinstance(out, dict) else out

    def _fit_graph(self, module, batch, opts):
        """max_epochs iterations on one static batch; iterations after the warm-up are replays of one captured graph."""
        if self.device.type != "cuda":
            raise RuntimeError("Trainer(graph=True) needs a GPU")
        if any(isinstance(o, torch.optim.LBFGS) for o in opts):
            raise ValueError("Trainer(graph=True): closure-driven optimizers (LBFGS) re-evaluate a data-dependent number of times "
                             "and cannot be captured; use Adam/SGD or graph=False")
        batch = _to(batch, self.device)
        for o in opts:
            for grp in o.param_groups:
                if "capturable" in grp:
                    grp["capturable"] = True       # optimizer state (step counters) lives on the device

        with_idx = len(opts) > 1 and _takes_optimizer_idx(module)

        def iteration():
            loss = None
            for k, o in enumerate(opts):
                o.zero_grad(set_to_none=False)
                loss = self._loss_of(module.training_step(batch, 0, k) if with_idx else module.training_step(batch, 0))
                loss.backward()
                o.step()
            return loss

        total = self.max_epochs if self.max_steps is None else min(self.max_epochs, self.max_steps)
        module.train()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):              # warm-up iterations are ordinary training steps (allocations, autotuning)
            for _ in range(min(self.graph_warmup, total)):
                self.history.append(float(iteration().detach()))
                self.global_step += 1
        torch.cuda.current_stream(self.device).wait_stream(side)
        if self.global_step >= total:
            return self
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                static_loss = iteration()
        except RuntimeError as e:
            raise RuntimeError("Trainer(graph=True): the training iteration could not be captured into a HIP graph -- most often a host "
                               "synchronisation inside training_step (`.item()`, `.cpu()`, printing a tensor).  Log detached tensors "
                               "(`self.log(name, value.detach())`) or use graph=False.  Original error: " + str(e)) from e
        # the capture itself does not execute; every replay is one full iteration
        while self.global_step < total:
            g.replay()
            self.global_step += 1
            if self.global_step % self.log_every == 0 or self.global_step == total:
                self.history.append(float(static_loss.detach()))
        self.current_epoch = module.current_epoch = total - 1
        self._graph = g
        return self

    def _wrap_ddp(self, module):
        """strategy="ddp" (the reference's `pl.Trainer(strategy='ddp')`, IBN_3D.py:193-195): one process per GPU, the
        network replicated, gradients averaged by bucketed all-reduces (RCCL on ROCm) overlapped with backward.  The
        process group must exist (torchrun / torch.distributed.run); every rank feeds its own shard of the data, e.g.
        `DeviceLoader(..., rank=r, world=n)`."""
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('Trainer(strategy="ddp") needs an initialised process group (launch with torch.distributed.run)')
        if dist.get_world_size() > 1 and any(p.requires_grad for p in module.network.parameters()):
            ids = [self.device.index] if self.device.type == "cuda" else None
            module.network = DDP(module.network, device_ids=ids)
            # keep the reference's checkpoint keys ("network.<...>", no "module." level) in state_dict() / load_state_dict()
            def strip(mod, sd, prefix, meta):
                for k in [k for k in sd if k.startswith(prefix + "network.module.")]:
                    sd[prefix + "network." + k[len(prefix + "network.module."):]] = sd.pop(k)
                return sd

            def add(sd, prefix, *unused):
                for k in [k for k in sd if k.startswith(prefix + "network.") and not k.startswith(prefix + "network.module.")]:
                    sd[prefix + "network.module." + k[len(prefix + "network."):]] = sd.pop(k)

            module._register_state_dict_hook(strip)
            module._register_load_state_dict_pre_hook(add)
        return module

    def fit(self, module, train_dataloaders, val_dataloaders=None):
        module.to(self.device)
        if self.strategy == "ddp":
            self._wrap_ddp(module)
        elif self.strategy is not None:
            raise ValueError(f"unknown strategy {self.strategy!r} (None or 'ddp')")
        conf = module.configure_optimizers()
        opts, scheds = conf if isinstance(conf, tuple) else (conf, [])
        opts = list(opts) if isinstance(opts, (list, tuple)) else [opts]
        if self.graph:
            batches = list(train_dataloaders)
            if len(batches) != 1 or scheds or val_dataloaders is not None or self.callbacks:
                raise ValueError("Trainer(graph=True) captures ONE static iteration: a single batch, no schedulers / validation / callbacks")
            return self._fit_graph(module, batches[0], opts)
        with_idx = len(opts) > 1 and _takes_optimizer_idx(module)
        for epoch in range(self.max_epochs):
            self.current_epoch = module.current_epoch = epoch
            module.train()
            for idx, batch in enumerate(train_dataloaders):
                batch = _to(batch, self.device)
                for k, opt in enumerate(opts):
                    def closure():
                        opt.zero_grad(set_to_none=True)
                        out = module.training_step(batch, idx, k) if with_idx else module.training_step(batch, idx)
                        loss = self._loss_of(out)
                        loss.backward()
                        return loss
                    loss = opt.step(closure)
                self.history.append(float(loss.detach()))
                self.global_step += 1
                if self.max_steps is not None and self.global_step >= self.max_steps:
                    return self
            for s in scheds:
                s.step()
            if val_dataloaders is not None and hasattr(module, "validation_step"):
                module.eval()
                with torch.no_grad():
                    for idx, batch in enumerate(val_dataloaders):
                        module.validation_step(_to(batch, self.device), idx)
            for cb in self.callbacks:
                if hasattr(cb, "on_train_epoch_end"):
                    cb.on_train_epoch_end(self, module)
        return self
