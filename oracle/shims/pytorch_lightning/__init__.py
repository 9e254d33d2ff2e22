"""TEST INFRASTRUCTURE ONLY -- oracle shim for pytorch_lightning==1.9.0
(reference requirements.txt:7).  Just enough for ``import rave`` and for driving
``RAVE.training_step`` by hand (rave/model.py:134,288-424)."""
import torch.nn as nn
from . import callbacks, loggers  # noqa: F401


class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._opts = None
        self._scheds = None
        self.logged = {}
        self.trainer = None
        self.logger = None
        self.automatic_optimization = True
        self.global_step = 0

    def log(self, name, value, *a, **k):
        self.logged[name] = value

    def log_dict(self, d, *a, **k):
        self.logged.update(d)

    def optimizers(self):
        return self._opts

    def lr_schedulers(self):
        return self._scheds

    def on_train_batch_end(self, outputs, batch, batch_idx):
        return None

    def save_hyperparameters(self, *a, **k):
        pass


class Callback:
    pass


class Trainer:  # never run by the oracle
    def __init__(self, *a, **k):
        raise NotImplementedError
