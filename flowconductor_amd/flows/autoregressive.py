"""Canned masked autoregressive flow (API of flowcon/flows/autoregressive.py:13-62)."""
from torch.nn import functional as F

from flowconductor_amd.distributions.normal import StandardNormal
from flowconductor_amd.flows.base import Flow
from flowconductor_amd.transforms.autoregressive import MaskedAffineAutoregressiveTransform
from flowconductor_amd.transforms.base import CompositeTransform
from flowconductor_amd.transforms.permutations import RandomPermutation, ReversePermutation


class MaskedAutoregressiveFlow(Flow):
    """num_layers x [permutation, MAF layer, (BatchNorm)] over a standard normal base."""

    def __init__(self, features, hidden_features, num_layers, num_blocks_per_layer,
                 use_residual_blocks=True, use_random_masks=False, use_random_permutations=False,
                 activation=F.relu, dropout_probability=0.0, batch_norm_within_layers=False,
                 batch_norm_between_layers=False):
        permutation = RandomPermutation if use_random_permutations else ReversePermutation
        layers = []
        for _ in range(num_layers):
            layers.append(permutation(features))
            layers.append(MaskedAffineAutoregressiveTransform(
                features=features, hidden_features=hidden_features, num_blocks=num_blocks_per_layer,
                use_residual_blocks=use_residual_blocks, random_mask=use_random_masks,
                activation=activation, dropout_probability=dropout_probability,
                use_batch_norm=batch_norm_within_layers))
            if batch_norm_between_layers:
                from flowconductor_amd.transforms.normalization import BatchNorm
                layers.append(BatchNorm(features))
        super().__init__(transform=CompositeTransform(layers), distribution=StandardNormal([features]))
