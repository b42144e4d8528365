"""Small argument predicates (API of flowcon/utils/typechecks.py)."""


def is_bool(x):
    return isinstance(x, bool)


def is_int(x):
    return isinstance(x, int)


def is_positive_int(x):
    return isinstance(x, int) and x > 0


def is_nonnegative_int(x):
    return isinstance(x, int) and x >= 0


def is_power_of_two(n):
    return is_positive_int(n) and (n & (n - 1)) == 0
