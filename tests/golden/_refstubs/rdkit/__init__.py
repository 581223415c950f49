from . import Chem  # noqa: F401
