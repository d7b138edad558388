"""`python lab4d/train.py --fg_motion gs-bob …` -- the reference's Stage-3 entry point
(/root/reference/lab4d/train.py:20-51), served by the MI355X-native path (vidu4d_amd/lab4d/train.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vidu4d_amd.lab4d.train import main  # noqa: E402

if __name__ == "__main__":
    main()
