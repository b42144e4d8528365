from flowconductor_amd.utils.torchutils import (  # noqa: F401
    batch_jacobian,
    cbrt,
    create_alternating_binary_mask,
    create_mid_split_binary_mask,
    create_random_binary_mask,
    get_num_parameters,
    logabsdet,
    merge_leading_dims,
    random_orthogonal,
    repeat_rows,
    searchsorted,
    split_leading_dim,
    sum_except_batch,
    tile,
)
from flowconductor_amd.utils.typechecks import (  # noqa: F401
    is_bool,
    is_int,
    is_nonnegative_int,
    is_positive_int,
    is_power_of_two,
)
