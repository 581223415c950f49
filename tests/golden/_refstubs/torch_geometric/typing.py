from typing import Any
Adj = Any
OptTensor = Any
Size = Any
