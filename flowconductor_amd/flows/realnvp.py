"""Canned RealNVP for 1-dim feature vectors (API of flowcon/flows/realnvp.py:17-71)."""
import torch
from torch.nn import functional as F

from flowconductor_amd.distributions.normal import StandardNormal
from flowconductor_amd.flows.base import Flow
from flowconductor_amd.nn import nets
from flowconductor_amd.transforms.base import CompositeTransform
from flowconductor_amd.transforms.coupling import AdditiveCouplingTransform, AffineCouplingTransform


class SimpleRealNVP(Flow):
    """Alternating checkerboard affine (or additive) coupling layers with ResidualNet conditioners."""

    def __init__(self, features, hidden_features, num_layers, num_blocks_per_layer,
                 use_volume_preserving=False, activation=F.relu, dropout_probability=0.0,
                 batch_norm_within_layers=False, batch_norm_between_layers=False):
        coupling = AdditiveCouplingTransform if use_volume_preserving else AffineCouplingTransform
        mask = torch.ones(features)
        mask[::2] = -1

        def create_resnet(in_features, out_features):
            return nets.ResidualNet(in_features, out_features, hidden_features=hidden_features,
                                    num_blocks=num_blocks_per_layer, activation=activation,
                                    dropout_probability=dropout_probability,
                                    use_batch_norm=batch_norm_within_layers)

        layers = []
        for _ in range(num_layers):
            layers.append(coupling(mask=mask, transform_net_create_fn=create_resnet))
            mask *= -1
            if batch_norm_between_layers:
                from flowconductor_amd.transforms.normalization import BatchNorm
                layers.append(BatchNorm(features=features))
        super().__init__(transform=CompositeTransform(layers), distribution=StandardNormal([features]))
