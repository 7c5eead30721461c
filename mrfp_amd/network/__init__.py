"""Drop-in for the reference's `network` package (hot-path subset: Resnet, mynn,
instance_whitening, sync_switchwhiten)."""
