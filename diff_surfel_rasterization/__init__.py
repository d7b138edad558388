"""Top-level alias so that `import diff_surfel_rasterization` (the name the reference's callers use,
gs/gaussian_renderer/__init__.py:14) resolves to the MI355X-native implementation when the repo root
is on sys.path."""
from vidu4d_amd.diff_surfel_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                                  _RasterizeGaussians, rasterize_frames, rasterize_gaussians, AUX_ALPHA)
from vidu4d_amd import _C  # noqa: F401
