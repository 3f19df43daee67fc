"""ref: svg/models/cosmos/placement.py (same layout transformation as Wan: text-free, frame-major <-> token-major)."""
from ..wan.placement import *  # noqa: F401,F403
from ..wan.placement import wan_hidden_states_placement as cosmos_hidden_states_placement  # noqa: F401
from ..wan.placement import wan_sparse_head_placement as cosmos_sparse_head_placement  # noqa: F401
from ..wan.placement import ref_wan_hidden_states_placement as ref_cosmos_hidden_states_placement  # noqa: F401,E402
from ..wan.placement import ref_wan_sparse_head_placement as ref_cosmos_sparse_head_placement  # noqa: F401,E402
from ..wan.placement import wan_token_reorder_to_frame_major as cosmos_token_reorder_to_frame_major  # noqa: F401,E402
from ..wan.placement import wan_token_reorder_to_token_major as cosmos_token_reorder_to_token_major  # noqa: F401,E402
