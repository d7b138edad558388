from vidu4d_amd.gs.graphics_utils import *  # noqa: F401,F403
