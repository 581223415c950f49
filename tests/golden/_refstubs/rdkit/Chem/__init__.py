AllChem = None
SDMolSupplier = None
MolFromMol2File = None
from . import rdMolAlign  # noqa: F401,E402
