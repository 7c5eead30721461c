from mrfp_amd.network.sync_switchwhiten import *  # noqa: F401,F403
from mrfp_amd.network.sync_switchwhiten import SyncSwitchWhiten2d  # noqa: F401
