from vidu4d_amd.gs.gaussian_renderer import render  # noqa: F401  (reference: gs/gaussian_renderer/__init__.py:21)
