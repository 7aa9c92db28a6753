"""Module / Sequential / ModuleList -- the container contract of neunet/nn/modules.py.

`parameters()` order (walk of __dict__ in insertion order, depth-first, dedup by id; modules.py:23-39)
defines the flat gradient-bucket order of the data-parallel all-reduce and of the multi-tensor AdamW.
"""
from collections import OrderedDict

import numpy as np

from ..autograd import Tensor


class Module:
    def __init__(self):
        self.training = True

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def parameters(self):
        params, seen = [], set()
        for _, item in self.__dict__.items():
            if isinstance(item, Tensor):
                if item.requires_grad and item.__class__.__name__ == "Parameter" and id(item) not in seen:
                    params.append(item)
                    seen.add(id(item))
            if hasattr(item, "parameters"):
                params.extend(item.parameters())
        return params

    def eval(self):
        self.training = False
        for _, item in self.__dict__.items():
            if hasattr(item, "eval"):
                item.eval()

    def train(self, mode: bool = True):
        self.training = mode
        for _, item in self.__dict__.items():
            if hasattr(item, "train"):
                item.train(mode)

    def to(self, device):
        """modules.py:53-68: rebinds every attribute that has .to() (breaks weight tying, as the reference)."""
        self.device = device
        for name, item in list(self.__dict__.items()):
            if hasattr(item, "to") and not isinstance(item, (str, bytes)):
                self.__dict__[name] = item.to(device)
        return self

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def state_dict(self):
        """Host NumPy arrays, so pickles stay interchangeable with the reference (modules.py:76-86)."""
        sd = OrderedDict()
        for name, item in self.__dict__.items():
            if isinstance(item, Tensor) and item.__class__.__name__ == "Parameter":
                sd[name] = item.numpy().copy()
            elif hasattr(item, "state_dict"):
                for k, v in item.state_dict().items():
                    sd[name + "." + k] = v
        return sd

    def load_state_dict(self, state_dict):
        for name, item in self.__dict__.items():
            if isinstance(item, Tensor) and item.__class__.__name__ == "Parameter":
                if name in state_dict:
                    src = np.asarray(state_dict[name], dtype=item.dtype)
                    if item.device == "cpu":
                        item.data = src.copy()
                    else:
                        import torch
                        item.data.copy_(torch.from_numpy(np.ascontiguousarray(src)))
            elif hasattr(item, "load_state_dict"):
                sub = {k.split(".", 1)[1]: v for k, v in state_dict.items() if k.startswith(name + ".")}
                item.load_state_dict(sub)


class Sequential(Module):
    """modules.py:110-170."""

    def __init__(self, *modules):
        super().__init__()
        self.modules = list(modules)

    def forward(self, X, *args, **kwargs):
        for m in self.modules:
            X = m(X)
        return X

    def parameters(self):
        params = []
        for m in self.modules:
            if hasattr(m, "parameters"):
                params.extend(m.parameters())
        return params

    def to(self, device):
        self.device = device
        self.modules = [m.to(device) if hasattr(m, "to") else m for m in self.modules]
        return self

    def eval(self):
        self.training = False
        for m in self.modules:
            if hasattr(m, "eval"):
                m.eval()

    def train(self, mode=True):
        self.training = mode
        for m in self.modules:
            if hasattr(m, "train"):
                m.train(mode)

    def state_dict(self):
        sd = OrderedDict()
        for i, m in enumerate(self.modules):
            if hasattr(m, "state_dict"):
                for k, v in m.state_dict().items():
                    sd[f"{i}.{k}"] = v
        return sd

    def load_state_dict(self, state_dict):
        for i, m in enumerate(self.modules):
            if hasattr(m, "load_state_dict"):
                pre = f"{i}."
                m.load_state_dict({k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)})


class ModuleList(Sequential):
    """modules.py:173-250 (list container; no forward)."""

    def __init__(self, modules=None):
        super().__init__(*(modules or []))

    def forward(self, *a, **k):
        raise NotImplementedError("ModuleList is a container")

    def __getitem__(self, i):
        return self.modules[i]

    def __len__(self):
        return len(self.modules)

    def __iter__(self):
        return iter(self.modules)

    def append(self, m):
        self.modules.append(m)
