from vidu4d_amd.gs.gaussian_model import GaussianModel, build_rotation, inverse_sigmoid  # noqa: F401
