"""Parameter(Tensor) -- neunet/nn/parameter.py:16-49.  Identified by CLASS NAME in
Module.parameters (neunet/nn/modules.py:31), so the class must be called `Parameter`."""
from ..autograd import Tensor


class Parameter(Tensor):
    def __init__(self, data: Tensor, requires_grad=True):
        if not isinstance(data, Tensor):
            raise TypeError("Data must be a tensor")
        super().__init__(data=data.data, requires_grad=requires_grad, device=data.device, dtype=data.dtype)

    def to(self, device):
        """Always copies (parameter.py:28-49): build optimizers AFTER .to(device)."""
        if device not in ("cpu", "cuda"):
            raise ValueError("Device must be 'cpu' or 'cuda'")
        return Parameter(Tensor(self.data, dtype=self.dtype, device=device), requires_grad=self.requires_grad)
