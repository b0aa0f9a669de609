from typing import Optional, Tuple, Union
from torch import Tensor
Adj = Union[Tensor, object]
OptTensor = Optional[Tensor]
Size = Optional[Tuple[int, int]]
