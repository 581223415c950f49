from . import composite  # noqa: F401
