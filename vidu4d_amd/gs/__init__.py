"""Host-side mirror of the 2DGS glue that sits between Vidu4D's Stage-3 field and the rasterizer
(reference: /root/reference/gs/{gaussian_renderer,scene,utils}).  Same names, argument meaning and
returned keys; PyTorch only moves memory and runs the small per-image epilogue, the hot work is in
the HIP library.  A top-level `gs/` alias package makes `from gs.gaussian_renderer import render`
etc. resolve here."""
