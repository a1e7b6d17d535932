"""Minimal fit loop for `PDE` modules where pytorch_lightning is not installed (it is absent on the build and GPU
boxes).  It drives exactly the hooks the reference scripts implement -- `configure_optimizers`, `training_step`
(tensor or {"loss": tensor}), optional `validation_step`, callbacks with `on_train_epoch_end(trainer, module)` -- and
nothing else; checkpointing, loggers and DDP remain Lightning's job when it is available."""
import torch


def _to(batch, device):
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to(b, device) for b in batch)
    return batch


class Trainer:
    def __init__(self, max_epochs=1, device=None, callbacks=(), max_steps=None):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.device = torch.device(device) if device is not None else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
        self.callbacks = list(callbacks)
        self.current_epoch = 0
        self.global_step = 0
        self.history = []

    @staticmethod
    def _loss_of(out):
        return out["loss"] if isinstance(out, dict) else out

    def fit(self, module, train_dataloaders, val_dataloaders=None):
        module.to(self.device)
        conf = module.configure_optimizers()
        opts, scheds = conf if isinstance(conf, tuple) else (conf, [])
        opts = list(opts) if isinstance(opts, (list, tuple)) else [opts]
        for epoch in range(self.max_epochs):
            self.current_epoch = module.current_epoch = epoch
            module.train()
            for idx, batch in enumerate(train_dataloaders):
                batch = _to(batch, self.device)
                for opt in opts:
                    def closure():
                        opt.zero_grad(set_to_none=True)
                        loss = self._loss_of(module.training_step(batch, idx))
                        loss.backward()
                        return loss
                    loss = opt.step(closure)
                self.history.append(float(loss))
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
