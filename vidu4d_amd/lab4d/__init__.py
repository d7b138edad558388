"""Host-side mirror of the lab4d pieces on the Stage-3 hot path: quaternion / dual-quaternion algebra,
the "bob" (bag-of-bones) linear-blend-skinning warp, the per-frame surfel field and its losses, and
the fitting loop with frame-parallel multi-GPU (reference: /root/reference/lab4d/{utils,nnutils,engine})."""
