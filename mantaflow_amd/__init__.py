"""mantaflow_amd -- MI355X-native hot path of mantaflow (advection, GridCg pressure projection, FLIP transfers)
behind the reference's Python scene API.  `from manta import *` (the top-level `manta` package) re-exports `api`."""
__version__ = "0.1.0"
