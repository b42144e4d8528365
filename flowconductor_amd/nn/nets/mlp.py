"""Plain MLP conditioner (PyTorch-ROCm; boundary only).

Constructor arguments and the ``net.N`` ``state_dict`` layout follow flowcon/nn/nets/mlp.py:13-66.
Used by the hyper-network ("conditional") transforms.
"""
import torch
from torch import nn


class MLP(nn.Module):
    """Linear / activation stack mapping ``in_shape`` tensors to ``out_shape`` tensors."""

    def __init__(self, in_shape, out_shape, hidden_sizes, activation=torch.nn.ReLU(),
                 activate_output=False):
        super().__init__()
        self._in_shape = torch.Size(in_shape)
        self._in_prod = self._in_shape.numel()
        self._out_shape = torch.Size(out_shape)
        self._hidden_sizes = hidden_sizes
        self._activation = activation
        self._activate_output = activate_output
        if len(hidden_sizes) == 0:
            raise ValueError("List of hidden sizes can't be empty.")
        widths = [self._in_prod] + list(hidden_sizes) + [self._out_shape.numel()]
        layers = []
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            if i > 0:
                layers.append(activation)
            layers.append(nn.Linear(fan_in, fan_out))
        if activate_output:
            layers.append(activation)
        self.net = nn.Sequential(*layers)

    def forward(self, inputs):
        return self.net(inputs.view(-1, self._in_prod)).view(-1, *self._out_shape)
