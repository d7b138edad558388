"""The "bob" (bag-of-bones) linear-blend-skinning warp of Stage-3 (`--fg_motion gs-bob`).

Reference: SkinningWarp.forward (lab4d/nnutils/warping.py:378-444), SkinningField.forward
(lab4d/nnutils/skinning.py:89-142), dual_quaternion_skinning (lab4d/utils/geom_utils.py:48-92),
ArticulationFlatMLP (lab4d/nnutils/pose.py:241-323), apply_qt_to_gaussian / forward_warp
(lab4d/nnutils/deformable_gaussian.py:1032-1046, :1395-1434).

Per frame m and canonical point x:
    se3_b      = t_articulation_b o rest_articulation_b^-1                  (B = 25 bones, dual quats)
    x_bone_b   = (rest_articulation_b^-1 x) / gauss_b                        Gaussian-bone coordinates
    skin_b     = -(|x_bone_b|^2 + 0.1 relu(MLP([x_bone (3B), t_embed, inst_code])_b))
    w          = softmax_b(skin)
    (q, t)     = normalise(sum_b w_b sign_b se3_b)   with sign_b aligning bone b's real part with the
                 arg-max bone's (same hemisphere), then dual quaternion -> (rotation, translation)
    x_t = q x q* + t ,  rot_t = q (x) rot ;  then the same with the field-to-camera (q, t).

Results-equivalent restructurings relative to upstream (which materialises (M,N,B,4) copies of the
bone dual quaternions, geom_utils.py:66-74, and concatenates a 235-wide MLP input per point): the
hemisphere signs are gathered from the BxB table of bone-pair dot products, the blend is a batched
(N,B)x(B,4) product, and the time / instance part of the first MLP layer -- identical for every point
of a frame -- is folded into its bias."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import quat_transform as qt


def fourier_embed(t: torch.Tensor, num_freq: int) -> torch.Tensor:
    """[t, sin(2^k pi t), cos(2^k pi t)] (PosEmbedding, lab4d/nnutils/embedding.py)."""
    out = [t]
    for k in range(num_freq):
        out += [torch.sin((2.0 ** k) * math.pi * t), torch.cos((2.0 ** k) * math.pi * t)]
    return torch.cat(out, -1)


class TimeEmbedding(nn.Module):
    """frame id -> 128-d code (TimeEmbedding, embedding.py:137-227): Fourier features of the
    normalised time followed by a linear map; `mean_embedding` is the code of the sequence mean."""

    def __init__(self, num_frames: int, num_freq: int = 6, out_channels: int = 128):
        super().__init__()
        self.num_frames, self.num_freq = num_frames, num_freq
        self.mapping = nn.Linear(2 * num_freq + 1, out_channels)
        self.out_channels = out_channels

    def forward(self, frame_id: torch.Tensor) -> torch.Tensor:
        t = (frame_id.float() / max(1, self.num_frames - 1)) * 2.0 - 1.0
        return self.mapping(fourier_embed(t[..., None], self.num_freq))

    def get_mean_embedding(self, device) -> torch.Tensor:
        ids = torch.arange(self.num_frames, device=device)
        return self.forward(ids).mean(0, keepdim=True)


class ArticulationFlatMLP(nn.Module):
    """time -> B bone-to-object SE(3) as dual quaternions (pose.py:241-323): a rest pose (bone centres,
    identity rotations) composed with a per-frame delta predicted by an MLP on the time code."""

    def __init__(self, num_frames: int, num_se3: int = 25, W: int = 256, init_radius: float = 0.1, seed: int = 0):
        super().__init__()
        self.num_se3 = num_se3
        g = torch.Generator().manual_seed(seed)
        centres = (torch.rand(num_se3, 3, generator=g) * 2 - 1) * init_radius
        self.rest_trans = nn.Parameter(centres)
        self.time_embedding = TimeEmbedding(num_frames)
        self.mlp = nn.Sequential(nn.Linear(self.time_embedding.out_channels, W), nn.ReLU(True), nn.Linear(W, W),
                                 nn.ReLU(True), nn.Linear(W, 6 * num_se3))
        nn.init.normal_(self.mlp[-1].weight, std=1e-2)
        nn.init.zeros_(self.mlp[-1].bias)

    def rest(self, device):
        q = torch.zeros(self.num_se3, 4, device=device)
        q[:, 0] = 1.0
        return qt.quaternion_translation_to_dual_quaternion(q, self.rest_trans.to(device))

    def get_vals_and_mean(self, frame_id: torch.Tensor):
        """-> (t_articulation, rest_articulation), each ((M,B,4), (M,B,4))."""
        M = frame_id.shape[0]
        dev = frame_id.device
        delta = self.mlp(self.time_embedding(frame_id)).view(M, self.num_se3, 6)
        dq_delta = qt.quaternion_translation_to_dual_quaternion(qt.axis_angle_to_quaternion(delta[..., :3]),
                                                                delta[..., 3:])
        rest = self.rest(dev)
        rest = (rest[0][None].expand(M, -1, -1).contiguous(), rest[1][None].expand(M, -1, -1).contiguous())
        return qt.dual_quaternion_mul(rest, dq_delta), rest


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = (q / q.norm(dim=-1, keepdim=True)).unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(q.shape[:-1] + (3, 3))


class SkinningField(nn.Module):
    """Gaussian bones + delta-skin MLP (skinning.py:14-142; D=2, W=64, 3B + 128 + 32 inputs)."""

    def __init__(self, num_bones: int, num_frames: int, num_inst: int = 1, W: int = 64, inst_channels: int = 32,
                 init_scale: float = 0.03, delta_skin: bool = True):
        super().__init__()
        self.num_bones = num_bones
        self.log_gauss = nn.Parameter(torch.log(init_scale * torch.ones(num_bones, 3)))
        self.delta_skin = delta_skin
        if delta_skin:
            self.time_embedding = TimeEmbedding(num_frames)
            self.inst_code = nn.Embedding(num_inst, inst_channels)
            self.fc_xyz = nn.Linear(3 * num_bones, W, bias=True)
            self.fc_time = nn.Linear(self.time_embedding.out_channels, W, bias=False)
            self.fc_inst = nn.Linear(inst_channels, W, bias=False)
            self.fc2 = nn.Linear(W, W)
            self.fc_out = nn.Linear(W, num_bones)

    def get_gauss(self):
        return self.log_gauss.exp()

    def bone_frames(self, bone2obj):
        """Object -> bone rotation (M,B,3,3) and translation (M,B,3) of the bone frames; a constant of the
        run while the articulation is frozen (the caller may cache it)."""
        q, t = qt.dual_quaternion_to_quaternion_translation(qt.dual_quaternion_inverse(bone2obj))
        return quaternion_to_matrix(q), t

    def bone_coords(self, xyz, bone2obj, bone_frames=None):
        """xyz (M,N,3), bone2obj ((M,B,4),(M,B,4)) -> (M,N,B,3) Gaussian-bone coordinates."""
        R, t = bone_frames if bone_frames is not None else self.bone_frames(bone2obj)
        xb = torch.einsum("mbij,mnj->mnbi", R, xyz) + t[:, None]
        return xb / self.get_gauss()[None, None]

    def forward(self, xyz, bone2obj, frame_id, inst_id, bone_frames=None):
        """-> (skin logits (M,N,B), delta (M,N,B) or None)"""
        xb = self.bone_coords(xyz, bone2obj, bone_frames)
        dist2 = xb.pow(2).sum(-1)
        if not self.delta_skin:
            return -dist2, None
        M, N = xyz.shape[:2]
        dev = xyz.device
        t_embed = (self.time_embedding.get_mean_embedding(dev).expand(M, -1) if frame_id is None
                   else self.time_embedding(frame_id))
        inst = self.inst_code(torch.zeros(M, dtype=torch.long, device=dev) if inst_id is None else inst_id)
        bias = self.fc_time(t_embed) + self.fc_inst(inst)  # (M,W): the per-frame constant part of layer 1
        h = F.relu(self.fc_xyz(xb.reshape(M, N, -1)) + bias[:, None])
        h = F.relu(self.fc2(h))
        delta = F.relu(self.fc_out(h)) * 0.1
        return -(dist2 + delta), delta


def dual_quaternion_skinning_qt(se3, skin_prob):
    """Hemisphere-aligned dual-quaternion blend -> per-point (q (M,N,4), t (M,N,3))
    (geom_utils.py:48-92 with return_qt=True)."""
    qr, qd = se3  # (M,B,4)
    anchor = skin_prob.argmax(-1)  # (M,N)
    dots = torch.einsum("mbk,mck->mbc", qr, qr)  # (M,B,B)
    B = qr.shape[1]
    sign = torch.gather(dots, 1, anchor[..., None].expand(-1, -1, B)) > 0  # (M,N,B): row `anchor` of dots
    w = skin_prob * (sign.to(skin_prob.dtype) * 2 - 1)
    qr_w = torch.bmm(w, qr)
    qd_w = torch.bmm(w, qd)
    inv = qr_w.norm(p=2, dim=-1, keepdim=True).reciprocal()
    return qt.dual_quaternion_to_quaternion_translation((qr_w * inv, qd_w * inv))


def cross_entropy_skin_loss(skin):
    """Entropy of the skinning distribution (lab4d/utils/loss_utils.py:21-42)."""
    p = skin.softmax(-1)
    return -(p * torch.log(p.clamp_min(1e-9))).sum(-1)


class SkinningWarp(nn.Module):
    """warp(xyz (M,N,1,3), frame_id, inst_id, samples_dict, return_qt=True, return_aux=True)
    (warping.py:325-444; forward direction only, which is what Stage-3 rendering uses)."""

    def __init__(self, num_frames: int, num_se3: int = 25, init_gauss_scale: float = 0.03, delta_skin: bool = True,
                 seed: int = 0):
        super().__init__()
        self.articulation = ArticulationFlatMLP(num_frames, num_se3, seed=seed)
        self.skinning_model = SkinningField(num_se3, num_frames, init_scale=init_gauss_scale, delta_skin=delta_skin)
        self.logibeta = nn.Parameter(-torch.tensor([0.01]).log())

    def forward(self, xyz, frame_id, inst_id=None, samples_dict=None, return_aux=False, return_qt=True):
        samples_dict = samples_dict or {}
        if "rest_articulation" in samples_dict and "t_articulation" in samples_dict:
            rest_art, t_art = samples_dict["rest_articulation"], samples_dict["t_articulation"]
        else:
            t_art, rest_art = self.articulation.get_vals_and_mean(frame_id)
        se3 = qt.dual_quaternion_mul(t_art, qt.dual_quaternion_inverse(rest_art))
        M, N = xyz.shape[:2]
        pts = xyz.reshape(M, N, 3)
        skin, delta = self.skinning_model(pts, rest_art, None, inst_id)  # forward warp: time-free skinning
        q, t = dual_quaternion_skinning_qt(se3, skin.softmax(-1))
        out = (q, t)
        if not return_qt:
            out = qt.quaternion_translation_apply(q, t, pts).view(xyz.shape)
        if not return_aux:
            return out
        aux = {"skin_entropy": cross_entropy_skin_loss(skin)[..., None]}
        if delta is not None:
            aux["delta_skin"] = delta.pow(2).mean(-1, keepdim=True)
        return out, aux


def apply_qt_to_gaussian(xyz, rotation, q, t, bs):
    """Moves surfel centres and orientations by per-point (q, t) (deformable_gaussian.py:1032-1046)."""
    shape = xyz.shape
    pts = qt.quaternion_translation_apply(q, t, xyz.reshape(bs, -1, 3)).view(*shape)
    rot = qt.quaternion_mul(q, rotation.reshape(bs, -1, 4)) if rotation is not None else None
    return pts, rot
