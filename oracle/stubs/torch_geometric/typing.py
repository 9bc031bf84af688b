from typing import Optional, Union

from torch import Tensor

Adj = Union[Tensor]
OptTensor = Optional[Tensor]
