"""Alias package: `from simple_knn._C import distCUDA2` (gs/scene/gaussian_model.py) resolves to the
MI355X implementation in vidu4d_amd/simple_knn.py."""
