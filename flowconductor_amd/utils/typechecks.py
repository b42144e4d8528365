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


# ---- helpers of this package (not in the reference): argument checks that raise, stubs for abstract methods ----

def need_positive_int(value, what):
    """TypeError("<what> must be a positive integer.") unless ``value`` is one."""
    if not is_positive_int(value):
        raise TypeError("%s must be a positive integer." % what)
    return value


def need_bool(value, what):
    if not is_bool(value):
        raise TypeError("%s must be boolean." % what)
    return value


def abstract(name, doc, exception=NotImplementedError):
    """A method body for protocol methods a subclass must supply: raises ``exception`` naming the class."""
    def stub(self, *args, **kwargs):
        raise exception("%s.%s" % (type(self).__name__, name)) if exception is NotImplementedError else exception()
    stub.__name__ = stub.__qualname__ = name
    stub.__doc__ = doc
    return stub
