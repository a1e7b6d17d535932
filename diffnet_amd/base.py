"""PDE base class -- same constructor kwargs, attributes and default hooks as the reference's
`DiffNet/base.py:6-55`, so user subclasses (IBN_2D.py, examples/*) drop in unchanged.

When pytorch_lightning is importable `PDE` is a real LightningModule; otherwise a minimal
stand-in (`torch.nn.Module` + no-op `log`) keeps everything else usable (tests, bench, the
built-in fit loop in diffnet_amd.trainer) without Lightning.
"""
import torch

try:  # Lightning is optional at run time (absent on the build and GPU boxes)
    from pytorch_lightning.core import LightningModule as _Base
    HAVE_LIGHTNING = True
except Exception:  # pragma: no cover - depends on the environment
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """Subset of the LightningModule surface the reference scripts touch on the module itself."""

        current_epoch = 0
        logger = None

        def __init__(self):
            super().__init__()
            self.logged = {}

        def log(self, name, value, *args, **kwargs):
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        def log_dict(self, d, *args, **kwargs):
            for k, v in d.items():
                self.log(k, v)


class PDE(_Base):
    """kwargs (reference names): nsd, domain_size, domain_sizes (X,Y,Z), domain_length, domain_lengths,
    batch_size, n_workers, learning_rate.  A second positional `dataset` is accepted (and stored) because
    64 legacy scripts of the reference still pass one ("ctor arity drift")."""

    #: scalar keyword arguments and their defaults (DiffNet/base.py:16-22)
    _SCALARS = (('nsd', 2), ('batch_size', 64), ('n_workers', 1), ('learning_rate', 3e-4), ('domain_length', 1.), ('domain_size', 64))

    def __init__(self, network, dataset=None, **kwargs):
        super().__init__()
        self.kwargs, self.network = kwargs, network
        if dataset is not None:
            self.dataset = dataset
        for name, default in self._SCALARS:
            setattr(self, name, kwargs.get(name, default))
        # per-axis extents default to the isotropic value; the X / Y / Z aliases exist up to the problem's dimension
        self.domain_lengths_nd = kwargs.get('domain_lengths', (self.domain_length,) * 3)
        self.domain_sizes_nd = kwargs.get('domain_sizes', (self.domain_size,) * 3)
        if self.nsd >= 2:
            for axis, letter in enumerate('XYZ'[:min(self.nsd, 3)]):
                setattr(self, 'domain_length' + letter, self.domain_lengths_nd[axis])
                setattr(self, 'domain_size' + letter, self.domain_sizes_nd[axis])

    def loss(self, u, inputs_tensor, forcing_tensor):
        raise NotImplementedError

    def forward(self, batch):
        inputs_tensor, forcing_tensor = batch
        return self.network(inputs_tensor), inputs_tensor, forcing_tensor

    def training_step(self, batch, batch_idx):
        value = self.loss(*self.forward(batch)).mean()
        # the two names the reference logs (base.py:45-46).  `.item()` is a host sync, illegal while the iteration is being
        # captured into a HIP graph (Trainer(graph=True)): log the detached tensor then -- it is the graph's static output
        # and reads the value of the latest replay
        capturing = value.is_cuda and torch.cuda.is_current_stream_capturing()
        for key in ('PDE_loss', 'loss'):
            self.log(key, value.detach() if capturing else value.item())
        return value

    def configure_optimizers(self):
        return [torch.optim.Adam(self.network.parameters(), lr=self.learning_rate)], []
