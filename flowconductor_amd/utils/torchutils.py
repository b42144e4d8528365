"""Tensor helpers used around the bijector path (API of flowcon/utils/torchutils.py).

Shape helpers (:25-58), mask builders (:102-144) and the compare-count ``searchsorted``
(:147-149) keep the reference's names and argument meaning.
"""
import torch

from flowconductor_amd.utils import typechecks as check


def tile(x, n):
    """``[a, b] -> [a]*n + [b]*n`` (each element repeated n times, order kept)."""
    if not check.is_positive_int(n):
        raise TypeError("Argument 'n' must be a positive integer.")
    return x.reshape(-1).repeat_interleave(n)


def sum_except_batch(x, num_batch_dims=1):
    """Sums all elements of `x` except for the first `num_batch_dims` dimensions."""
    if not check.is_nonnegative_int(num_batch_dims):
        raise TypeError("Number of batch dimensions must be a non-negative integer.")
    return torch.sum(x, dim=list(range(num_batch_dims, x.ndimension())))


def split_leading_dim(x, shape):
    """Reshapes the leading dim of `x` to have the given shape."""
    return torch.reshape(x, torch.Size(shape) + x.shape[1:])


def merge_leading_dims(x, num_dims):
    """Merges the first `num_dims` dimensions of `x` into one."""
    if not check.is_positive_int(num_dims):
        raise TypeError("Number of leading dims must be a positive integer.")
    if num_dims > x.dim():
        raise ValueError("Number of leading dims can't be greater than total number of dims.")
    return torch.reshape(x, torch.Size([-1]) + x.shape[num_dims:])


def repeat_rows(x, num_reps):
    """Each row of tensor `x` is repeated `num_reps` times along leading dimension."""
    if not check.is_positive_int(num_reps):
        raise TypeError("Number of repetitions must be a positive integer.")
    return x.repeat_interleave(num_reps, dim=0)


def create_alternating_binary_mask(features, even=True):
    """Byte mask 1,0,1,0,... (even=True) or 0,1,0,1,... (even=False)."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[(0 if even else 1)::2] = 1
    return mask


def create_mid_split_binary_mask(features):
    """Byte mask with the first ceil(features/2) entries set."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[: (features + 1) // 2] = 1
    return mask


def create_random_binary_mask(features):
    """Byte mask with ceil(features/2) randomly chosen entries set."""
    mask = torch.zeros(features, dtype=torch.uint8)
    chosen = torch.multinomial(torch.ones(features), (features + 1) // 2, replacement=False)
    mask[chosen] = 1
    return mask


def searchsorted(bin_locations, inputs, eps=1e-6):
    """Compare-count bin index; like the reference it nudges the last edge by `eps` in place."""
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


def cbrt(x):
    """Cube root via sign * exp(log|x| / 3)."""
    return torch.sign(x) * torch.exp(torch.log(torch.abs(x)) / 3.0)


def get_num_parameters(model):
    """Number of elements over all parameters of `model`."""
    return sum(p.numel() for p in model.parameters())


def batch_jacobian(g, x):
    """[B, n_out, n_in] Jacobian of g w.r.t. x by one backward pass per output dim."""
    rows = []
    for d in range(g.shape[1]):
        rows.append(torch.autograd.grad(torch.sum(g[:, d]), x, retain_graph=True, create_graph=True)[0]
                    .view(x.shape[0], 1, x.shape[1]))
    return torch.cat(rows, 1)


def logabsdet(x):
    """log|det x| of a square matrix."""
    return torch.linalg.slogdet(x)[1]


def random_orthogonal(size):
    """Random orthogonal matrix from the QR of a Gaussian matrix."""
    q, _ = torch.linalg.qr(torch.randn(size, size))
    return q
