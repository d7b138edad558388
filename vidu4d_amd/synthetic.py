"""Synthetic surfel scenes of BASELINE.md §3 / SURVEY.md §8(d).

No dataset ships with the reference, so tests and bench.py use seeded random camera-space surfel
sets shaped like what `DeformableGaussian.query_field` hands to the rasterizer
(/root/reference/lab4d/nnutils/deformable_gaussian.py:1175-1228): identity view matrix, campos = 0,
centred principal point, SH degree 3 (16 coefficients), scales (N,2), unit quaternions (N,4),
opacities (N,1) in (0,1).

Everything is generated on the CPU with a seeded torch.Generator (so that CPU oracle and GPU runs
see bit-identical inputs) and moved to `device` afterwards.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class SurfelScene:
    means3D: torch.Tensor  # (N,3) camera space
    scales: torch.Tensor  # (N,2) activated (positive)
    rotations: torch.Tensor  # (N,4) unit quaternions, w first
    opacities: torch.Tensor  # (N,1) in (0,1)
    shs: torch.Tensor  # (N,16,3)
    viewmatrix: torch.Tensor  # (4,4) row-vector convention (= W^T), identity here
    projmatrix: torch.Tensor  # (4,4)
    campos: torch.Tensor  # (3,)
    bg: torch.Tensor  # (3,)
    width: int
    height: int
    tanfovx: float
    tanfovy: float
    sh_degree: int

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return SurfelScene(**kw)

    @property
    def num_surfels(self):
        return self.means3D.shape[0]


def projection_matrix(tanfovx: float, tanfovy: float, znear: float = 0.01, zfar: float = 100.0) -> torch.Tensor:
    """Row-vector (transposed) perspective matrix with a centred principal point
    (semantics of getProjectionMatrix, /root/reference/gs/utils/graphics_utils.py:51-71)."""
    top, right = tanfovy * znear, tanfovx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P.t().contiguous()


def make_scene(n: int, width: int, height: int | None = None, seed: int = 1234, tanfov: float = 0.5,
               sigma_px: float = 1.5, big_fraction: float = 0.01, big_factor: float = 8.0,
               opacity_mode: str = "random", sh_degree: int = 3, bg=(0.0, 0.0, 0.0), device="cpu") -> SurfelScene:
    """BASELINE.md §3 scene: z~U(2,4); x,y = U(-1,1)*0.9*z*tanfov; scales = s0*exp(0.35*N(0,1)) with
    s0 = sigma_px*3/focal (3 = mean depth), `big_fraction` of the surfels `big_factor` x larger;
    rotations = normalised N(0,1)^4; opacity = sigmoid(N(0,2^2)) ("random") or 0.1 ("init",
    gaussian_model.py:143); f_dc ~ 0.5 N(0,1), f_rest ~ 0.1 N(0,1)."""
    height = width if height is None else height
    g = torch.Generator().manual_seed(seed)
    tanfovx = tanfov
    tanfovy = tanfov * height / width
    focal = width / (2.0 * tanfovx)
    z = torch.rand(n, generator=g) * 2.0 + 2.0
    u = torch.rand(n, generator=g) * 2.0 - 1.0
    v = torch.rand(n, generator=g) * 2.0 - 1.0
    means = torch.stack([u * 0.9 * z * tanfovx, v * 0.9 * z * tanfovy, z], dim=1)
    s0 = sigma_px * 3.0 / focal
    scales = s0 * torch.exp(0.35 * torch.randn(n, 2, generator=g))
    big = torch.rand(n, generator=g) < big_fraction
    scales = torch.where(big[:, None], scales * big_factor, scales)
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    if opacity_mode == "random":
        opac = torch.sigmoid(2.0 * torch.randn(n, 1, generator=g))
    elif opacity_mode == "init":
        opac = torch.full((n, 1), 0.1)
    else:
        raise ValueError(opacity_mode)
    f_dc = 0.5 * torch.randn(n, 1, 3, generator=g)
    f_rest = 0.1 * torch.randn(n, 15, 3, generator=g)
    shs = torch.cat([f_dc, f_rest], dim=1).contiguous()
    view = torch.eye(4)
    proj = projection_matrix(tanfovx, tanfovy)
    scene = SurfelScene(
        means3D=means.float().contiguous(), scales=scales.float().contiguous(), rotations=rot.float().contiguous(),
        opacities=opac.float().contiguous(), shs=shs.float(), viewmatrix=view, projmatrix=(view @ proj).contiguous(),
        campos=torch.zeros(3), bg=torch.tensor(bg, dtype=torch.float32), width=width, height=height,
        tanfovx=float(tanfovx), tanfovy=float(tanfovy), sh_degree=sh_degree)
    return scene.to(device)


def make_object_scene(n: int, width: int, height: int | None = None, radius: float = 1.0, seed: int = 1234,
                      **kw) -> SurfelScene:
    """Object-centric variant of `make_scene` (what a Stage-3 frame looks like after lab4d's crop around
    the object): same surfel attributes, but the centres fill a ball of `radius` at depth 3 (camera at the
    origin, tan(fov/2) = 0.5, so radius 1 covers about a third of the image).  Tile lists are then very
    uneven -- a few hundred tiles hold all pairs -- which is the regime the longest-first schedule and
    the segment-parallel blend exist for."""
    sc = make_scene(n, width, height, seed=seed, **kw)
    g = torch.Generator().manual_seed(seed + 7)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True) * torch.rand(n, 1, generator=g).pow(1.0 / 3.0)
    sc.means3D = (radius * d + torch.tensor([0.0, 0.0, 3.0])).float().contiguous()
    return sc


def make_upstream_grads(width: int, height: int, seed: int = 4321, device="cpu"):
    """Upstream gradients for op-level runs: N(0,1)/(H*W) on the colour image and on
    all 8 auxiliary planes (worst case: every backward branch live)."""
    g = torch.Generator().manual_seed(seed)
    hw = float(width * height)
    d_color = torch.randn(3, height, width, generator=g) / hw
    d_others = torch.randn(8, height, width, generator=g) / hw
    return d_color.to(device), d_others.to(device)


def frame_motion(scene: SurfelScene, frame: int, num_frames: int, seed: int = 99) -> SurfelScene:
    """A cheap deterministic per-frame rigid-ish motion of the camera-space surfels (rotation about
    the view axis by <= 0.3 rad plus a small in-plane shift), standing in for the bob warp when only
    the rasterizer op is measured.  The fitting-loop driver uses the real LBS warp instead."""
    t = 2.0 * math.pi * frame / max(1, num_frames)
    ang = 0.3 * math.sin(t)
    c, s = math.cos(ang), math.sin(ang)
    Rz = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=scene.means3D.dtype,
                      device=scene.means3D.device)
    shift = torch.tensor([0.05 * math.cos(t), 0.05 * math.sin(2 * t), 0.1 * math.sin(t)],
                         dtype=scene.means3D.dtype, device=scene.means3D.device)
    means = scene.means3D @ Rz.t() + shift
    # quaternion of Rz composed on the left: q' = qz * q
    qz = torch.tensor([math.cos(ang / 2), 0.0, 0.0, math.sin(ang / 2)], dtype=scene.rotations.dtype,
                      device=scene.rotations.device)
    w1, x1, y1, z1 = qz
    w2, x2, y2, z2 = scene.rotations.unbind(-1)
    rot = torch.stack([
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)
    kw = dict(scene.__dict__)
    kw.update(means3D=means.contiguous(), rotations=rot.contiguous())
    return SurfelScene(**kw)
