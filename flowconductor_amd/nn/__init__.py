from flowconductor_amd.nn import nets  # noqa: F401
