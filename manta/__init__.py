"""`from manta import *` -- the scene-facing module name of the reference (pwrapper/registry.cpp:21)."""
from mantaflow_amd.api import *  # noqa: F401,F403
from mantaflow_amd.api import args, SCENEFILE  # noqa: F401
