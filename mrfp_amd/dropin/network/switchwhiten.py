from mrfp_amd.network.switchwhiten import SwitchWhiten2d  # noqa: F401
