from vidu4d_amd.gs.sh_utils import C0, RGB2SH, SH2RGB  # noqa: F401
