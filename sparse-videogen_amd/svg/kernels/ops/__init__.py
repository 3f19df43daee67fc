from .attention_ops import *  # noqa: F401,F403  (ref: svg/kernels/ops/__init__.py)
from .attention_ops_wan import *  # noqa: F401,F403
