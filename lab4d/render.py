"""`python lab4d/render.py --flagfile=... --load_suffix latest --render_res 512` -- the reference's forward-only
rendering entry (/root/reference/lab4d/render.py:279-354), served by the MI355X-native path (vidu4d_amd/lab4d/render.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vidu4d_amd.lab4d.render import main  # noqa: E402

if __name__ == "__main__":
    main()
