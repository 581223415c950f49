from . import nn, utils, data, loader  # noqa: F401
