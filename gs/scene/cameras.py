from vidu4d_amd.gs.cameras import KCamera, MiniCam  # noqa: F401
