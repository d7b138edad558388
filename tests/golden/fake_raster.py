"""A differentiable stand-in for the rasterizer, used ONLY to pin the per-frame render loop against the
imported reference: `make_refpy_golden.py::gen_loop` runs the reference's `DeformableGaussian.query_field` /
`render_view` around it, `tests/test_refpy_host.py::test_render_loop_*` runs `DeformableSurfels.render_frames` around
the very same function.  What it computes has no meaning; it only has to depend smoothly on EVERY argument the loop
hands to the rasterizer (warped centres, activated rotations / scales / opacities, SH rows, the camera's field of view
and image size), so that a wrong override, a missing activation or a frame mix-up changes the images and gradients.
Test infrastructure; loaded by file path (the generator keeps the repository root off sys.path)."""
import torch


def _mix(n_rows, n_cols, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n_rows, n_cols, generator=g)


def fake_raster(settings, means3D, opacities, shs, scales, rotations):
    """-> color (3,H,W), radii (N,) int32, allmap (8,H,W): fixed random linear maps of per-surfel features, then a
    plane-wise nonlinearity (alpha in (0,1), positive depths)."""
    H, W = int(settings.image_height), int(settings.image_width)
    N = means3D.shape[0]
    dev = means3D.device
    feats = torch.cat([means3D, rotations, scales, opacities, shs[:, 0, :], shs[:, 5, :]], dim=1)  # (N,16)
    K = 6
    pooled = (_mix(K, N, 7).to(dev) @ feats) / N ** 0.5                                   # (K,16)
    fov = torch.as_tensor([float(settings.tanfovx), float(settings.tanfovy)], device=dev)
    pooled = torch.cat([pooled.reshape(-1), fov])                                         # (K*16+2,)
    planes = (_mix(11 * H * W, pooled.numel(), 11).to(dev) @ pooled).reshape(11, H, W) / pooled.numel() ** 0.5
    color = torch.sigmoid(planes[:3]) * 1.0  # (upstream's render_view composites the background IN PLACE into this
    # tensor, deformable_gaussian.py:190: it must not be an output autograd saved)
    alpha = torch.sigmoid(2.0 * planes[4:5])
    # (no exactly-empty pixels: upstream's d(depth / alpha) is NaN there -- refpy_render.npz pins that behaviour -- and
    # through this function's pooling one NaN would reach every gradient)
    depth = 2.0 + torch.sigmoid(planes[3:4])
    allmap = torch.cat([depth * alpha, alpha, planes[5:8] * alpha, depth * 1.01,
                        0.1 * torch.sigmoid(planes[8:9]), torch.sigmoid(planes[9:10])], dim=0)
    radii = (opacities[:, 0].detach() * 97.0).to(torch.int32) % 4   # some surfels invisible (sigmoid of the same numbers
    # on both sides: bit-identical, unlike anything derived from the warped centres)
    return color, radii, allmap
