"""SH <-> RGB for the DC coefficient (reference: gs/utils/sh_utils.py:114-118)."""
C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
