from flowconductor_amd.nn.nets.mlp import MLP  # noqa: F401
from flowconductor_amd.nn.nets.resnet import ResidualBlock, ResidualNet  # noqa: F401
