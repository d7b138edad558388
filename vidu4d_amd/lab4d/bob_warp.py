"""The "bob" (bag-of-bones) neural blend-skinning warp of Stage-3 (`--fg_motion gs-bob`), state-dict
compatible with the reference's SkinningWarp so that Stage-2 checkpoints drive real bones.

Reference: SkinningWarp (lab4d/nnutils/warping.py:325-444), SkinningField (lab4d/nnutils/skinning.py:14-142),
get_bone_coords (lab4d/utils/transforms.py:9-25), dual_quaternion_skinning (lab4d/utils/geom_utils.py:48-92),
cross_entropy_skin_loss (lab4d/utils/loss_utils.py), apply_qt_to_gaussian / forward_warp
(lab4d/nnutils/deformable_gaussian.py:1032-1046, :1395-1434).

Per frame m and canonical point x (B = 25 bones, everything a dual quaternion):
    se3_b      = t_articulation_b o rest_articulation_b^-1
    x_bone_b   = (articulation_b^-1 x) / gauss_b                                Gaussian-bone coordinates
    delta      = 0.1 relu(delta_field([x_bone (3B) | time code (128) | instance code (32)]))
    skin_b     = -(|x_bone_b|^2 + delta_b) ,  w = softmax_b(skin)
    (q, t)     = normalise(sum_b w_b s_b se3_b), s_b = +-1 aligning bone b with the arg-max bone's hemisphere
    x_t = q x q* + t ,  rot_t = q (x) rot ;  then the same with the field-to-camera (q, t).
Keys: `articulation.*` (nets.ArticulationFlatMLP), `skinning_model.log_gauss`, `skinning_model.time_embedding.*`,
`skinning_model.delta_field.{linear_1.0,linear_2.0,linear_final,inst_embedding.mapping}.*`, `logibeta`.

Results-equivalent restructurings relative to upstream (which materialises (M,N,B,4) copies of the bone dual
quaternions, geom_utils.py:66-74, and a 235-wide MLP input per point): bone coordinates come from one (B,3,3)
rotation + translation per frame, the hemisphere signs from the BxB table of bone-pair dot products, the blend
is a batched (N,B)x(B,4) product, and the time / instance columns of the first MLP layer -- identical for
every point of a frame -- are applied once per frame and added as a bias.  Values are pinned against the
imported reference in tests/test_refpy_nets.py (fixtures tests/golden/refpy_warp.npz)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import quat_transform as qt
from .nets import ArticulationFlatMLP, CondDenseStack, TimeEmbedding, fourier_dim, fourier_features



SPLIT_K_CHUNK = 2048


class _FeatureMajorLinear(torch.autograd.Function):
    """Y (O, N) = W (O, I) @ X (I, N) + b[:, None] over N surfels.  The forward and d/dX are ordinary library GEMMs; the
    weight gradient G (O, N) @ X^T (N, I) contracts over the SURFELS, a shape (64 x 75 x 200 000) for which rocBLAS runs
    16 x 256 tiles down the whole K in a handful of workgroups (722 us, rocprofv3, round 5).  It is evaluated as a batched
    GEMM over K-chunks of `chunk` surfels -- strided views of the same arrays, no copy -- and a sum over the chunks: a few
    hundred workgroups.  Same products, another summation order (float: ~1e-6 relative)."""

    @staticmethod
    def forward(ctx, b, W, X, chunk):
        ctx.save_for_backward(W, X)
        ctx.chunk = int(chunk)
        return torch.addmm(b[:, None], W, X)

    @staticmethod
    def backward(ctx, G):
        W, X = ctx.saved_tensors
        G = G.contiguous()
        gb = G.sum(1) if ctx.needs_input_grad[0] else None
        gW = None
        if ctx.needs_input_grad[1]:
            gW = contract_over_columns(G, X, ctx.chunk)
        gX = W.t().mm(G) if ctx.needs_input_grad[2] else None
        return gb, gW, gX, None


def contract_over_columns(G, X, chunk=SPLIT_K_CHUNK):
    """G (O, N) @ X (I, N)^T -> (O, I), the long contraction cut into `chunk`-column pieces that run as one batched GEMM."""
    O, N = G.shape
    n_full = N // chunk if chunk > 0 else 0
    if n_full < 4:
        return G.mm(X.t())
    body = n_full * chunk
    Gc = G[:, :body].unflatten(1, (n_full, chunk)).permute(1, 0, 2)              # (n, O, chunk), row stride N
    Xc = X[:, :body].unflatten(1, (n_full, chunk)).permute(1, 2, 0)              # (n, chunk, I), column stride N (or, for
    #                                                                              X = points^T, contiguous (chunk, 3) blocks)
    out = torch.bmm(Gc, Xc).sum(0)
    if body < N:
        out = out + G[:, body:].mm(X[:, body:].t())
    return out


def feature_major_linear(b, W, X, chunk=SPLIT_K_CHUNK):
    if chunk and (W.requires_grad or b.requires_grad) and torch.is_grad_enabled():
        return _FeatureMajorLinear.apply(b, W, X, chunk)
    return torch.addmm(b[:, None], W, X)

class SkinningField(nn.Module):
    def __init__(self, num_coords, frame_info, num_inst, D=2, W=64, num_freq_xyz=0, num_freq_t=6, inst_channels=32,
                 skips=(4,), activation=None, init_scale=0.03, delta_skin=True, symm_idx=None):
        super().__init__()
        self.num_coords = num_coords
        self.num_freq_xyz = num_freq_xyz
        self.log_gauss = nn.Parameter(torch.log(init_scale * torch.ones(num_coords, 3)))
        if delta_skin:
            self.time_embedding = TimeEmbedding(num_freq_t, frame_info)
            self.xyz_channels = fourier_dim(3 * num_coords, num_freq_xyz)
            self.delta_field = CondDenseStack(num_inst=num_inst, D=D, W=W,
                                              in_channels=self.xyz_channels + self.time_embedding.out_channels,
                                              inst_channels=inst_channels, out_channels=num_coords, skips=skips,
                                              activation=activation if activation is not None else nn.ReLU(True))
        self.symm_idx = symm_idx

    @property
    def has_delta(self) -> bool:
        return hasattr(self, "delta_field")

    def get_gauss(self):
        lg = self.log_gauss
        if self.symm_idx is not None:
            lg = (lg[self.symm_idx] + lg) / 2
        return lg.exp()

    @staticmethod
    def bone_frames(bone2obj):
        """Object -> bone rotation (M,B,3,3) and translation (M,B,3); constants of the run while the articulation
        is frozen (the caller may cache them)."""
        q, t = qt.dual_quaternion_to_quaternion_translation(qt.dual_quaternion_inverse(bone2obj))
        return qt.quaternion_to_matrix(q), t

    def bone_coords(self, xyz, bone2obj, bone_frames=None):
        """xyz (M,N,3); bone2obj ((M,B,4),(M,B,4)) -> (M,N,B,3) coordinates in the Gaussian bones' frames."""
        R, t = bone_frames if bone_frames is not None else self.bone_frames(bone2obj)
        return (torch.einsum("mbij,mnj->mnbi", R, xyz) + t[:, None]) / self.get_gauss()

    def bone_affine(self, bone2obj):
        """x_bone / gauss as ONE affine map of the canonical point for a single articulation ((1,B,4),(1,B,4)):
        returns A (3B,3) and c (3B,) with x_boneT = A @ xyz^T + c[:, None]  (feature-major, row 3b + k)."""
        R, t = self.bone_frames(bone2obj)
        ig = 1.0 / self.get_gauss()                                  # (B,3)
        A = (R[0] * ig[:, :, None]).reshape(-1, 3)
        return A, (t[0] * ig).reshape(-1)

    def delta_raw_T(self, xbT, frame_bias, split_k=SPLIT_K_CHUNK):
        """Raw delta-skin MLP output in feature-major layout: xbT (3B,N) -> (B,N).  Same weights and arithmetic as
        `forward` (first layer split into its coordinate columns and the per-frame bias), written as W @ X so that
        the (B,N) result is what csrc/lbs.hip reads with coalesced loads."""
        mlp = self.delta_field
        assert self.num_freq_xyz == 0 and not any(0 < s < mlp.D for s in mlp.skips)
        lin = mlp.linear_1[0]
        h = F.relu(feature_major_linear(frame_bias.reshape(-1), lin.weight[:, :self.xyz_channels], xbT, split_k))
        for i in range(1, mlp.D):
            li = getattr(mlp, f"linear_{i + 1}")[0]
            h = F.relu(feature_major_linear(li.bias, li.weight, h, split_k))
        return feature_major_linear(mlp.linear_final.bias, mlp.linear_final.weight, h, split_k)

    def frame_bias(self, frame_id, inst_id, M, device):
        """(M or 1, W): the first layer applied to the per-frame part of its input (time code | instance code)."""
        te = self.time_embedding
        t_embed = te.get_mean_embedding(device) if frame_id is None else te(frame_id)
        code = self.delta_field.instance_code(inst_id, None, device)
        if code.shape[0] != t_embed.shape[0]:
            t_embed, code = t_embed.expand(max(code.shape[0], t_embed.shape[0]), -1), code.expand(
                max(code.shape[0], t_embed.shape[0]), -1)
        lin = self.delta_field.linear_1[0]
        return F.linear(torch.cat((t_embed, code), -1), lin.weight[:, self.xyz_channels:], lin.bias)

    def forward(self, xyz, bone2obj, frame_id, inst_id, bone_frames=None, frame_bias=None):
        """xyz (M,N,3) -> (skin logits (M,N,B), delta (M,N,B) or None).  frame_id None = the mean time code
        (what the forward warp uses, warping.py:415-425)."""
        xb = self.bone_coords(xyz, bone2obj, bone_frames)
        dist2 = xb.pow(2).sum(-1)
        if not self.has_delta:
            return -dist2, None
        M, N = xyz.shape[:2]
        if frame_bias is None:
            frame_bias = self.frame_bias(frame_id, inst_id, M, xyz.device)
        mlp = self.delta_field
        lin = mlp.linear_1[0]
        feats = fourier_features(xb.reshape(M, N, -1), self.num_freq_xyz)
        h = F.relu(F.linear(feats, lin.weight[:, :self.xyz_channels]) + frame_bias[:, None])
        if any(0 < s < mlp.D for s in mlp.skips):  # a re-fed input needs the full row (not the bob field: D=2, skips=[4])
            t = self.time_embedding
            t_embed = t.get_mean_embedding(xyz.device) if frame_id is None else t(frame_id)
            delta = mlp(torch.cat((feats, t_embed[:, None].expand(M, N, -1)), -1), inst_id)
        else:
            for i in range(1, mlp.D):
                h = getattr(mlp, f"linear_{i + 1}")(h)
            delta = mlp.linear_final(h)
        delta = F.relu(delta) * 0.1
        return -(dist2 + delta), delta


def dual_quaternion_skinning_qt(se3, skin_prob):
    """Hemisphere-aligned dual-quaternion blend -> per-point (q (M,N,4), t (M,N,3))
    (geom_utils.py:48-92 with return_qt=True).  se3 ((M,B,4),(M,B,4)), skin_prob (M,N,B)."""
    qr, qd = se3
    B = qr.shape[1]
    anchor = skin_prob.argmax(-1)                                  # (M,N)
    dots = torch.einsum("mbk,mck->mbc", qr, qr)                    # (M,B,B) bone-pair real-part products
    sign = torch.gather(dots, 1, anchor[..., None].expand(-1, -1, B)) > 0
    w = skin_prob * (sign.to(skin_prob.dtype) * 2 - 1)
    qr_w, qd_w = torch.bmm(w, qr), torch.bmm(w, qd)
    inv = qr_w.norm(p=2, dim=-1, keepdim=True).reciprocal()
    return qt.dual_quaternion_to_quaternion_translation((qr_w * inv, qd_w * inv))


def cross_entropy_skin_loss(skin):
    """lab4d/utils/loss_utils.py cross_entropy_skin_loss: cross entropy of the logits against their own
    arg-max bone (sharpness of the assignment)."""
    shape = skin.shape
    flat = skin.reshape(-1, shape[-1])
    return F.cross_entropy(flat, flat.argmax(-1), reduction="none").view(shape[:-1])


class SkinningWarp(nn.Module):
    """warp(xyz (M,N,1,3), frame_id, inst_id, samples_dict, return_qt=True, return_aux=True): forward direction
    (canonical -> time t), which is what Stage-3 rendering uses."""

    def __init__(self, frame_info, skel_type="flat", joint_angles=None, num_freq_xyz=10, num_freq_t=6, num_se3=25,
                 init_gauss_scale=0.03, init_beta=0.01, delta_skin=True):
        super().__init__()
        if skel_type != "flat":
            raise NotImplementedError("only the bag-of-bones articulation is on the Stage-3 path")
        fo = frame_info["frame_offset"]
        self.num_frames = int(fo[-1])
        self.num_inst = len(fo) - 1
        self.articulation = (ArticulationFlatMLP(frame_info, num_se3) if num_se3 < 50
                             else ArticulationFlatMLP(frame_info, num_se3, D=2, W=32))
        self.skinning_model = SkinningField(num_se3, frame_info, self.num_inst, init_scale=init_gauss_scale,
                                            delta_skin=delta_skin)
        self.logibeta = nn.Parameter(-torch.tensor([init_beta]).log())

    def forward(self, xyz, frame_id, inst_id=None, backward=False, samples_dict=None, return_aux=False,
                return_qt=False):
        if backward:
            raise NotImplementedError("backward (time t -> canonical) warping is not on the Stage-3 path")
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
        out = (q, t) if return_qt else qt.quaternion_translation_apply(q, t, pts).view(xyz.shape)
        if not return_aux:
            return out
        aux = {"skin_entropy": cross_entropy_skin_loss(skin)[..., None, None]}
        if delta is not None:
            aux["delta_skin"] = delta.pow(2).mean(-1, keepdim=True)[..., None, :]
        return out, aux


def create_warp(fg_motion: str, data_info: dict):
    """warping.py:24-70, the bag-of-bones flavours."""
    fi = data_info["frame_info"]
    if fg_motion == "bob":
        return SkinningWarp(fi)
    if fg_motion == "bob-nosoft":
        return SkinningWarp(fi, delta_skin=False)
    raise NotImplementedError(f"fg_motion gs-{fg_motion}: only gs-bob / gs-bob-nosoft are on the Stage-3 path")


def apply_qt_to_gaussian(xyz, rotation, q, t, bs):
    """Moves surfel centres and orientations by per-point (q, t) (deformable_gaussian.py:1032-1046)."""
    shape = xyz.shape
    pts = qt.quaternion_translation_apply(q, t, xyz.reshape(bs, -1, 3)).view(*shape)
    rot = qt.quaternion_mul(q, rotation.reshape(bs, -1, 4)) if rotation is not None else None
    return pts, rot
