"""Drop-in for the reference's `network` package (hot-path subset)."""
