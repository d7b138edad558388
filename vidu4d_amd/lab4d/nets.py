"""Time-conditioned networks of the Stage-3 field, state-dict compatible with the reference's modules so
that a Stage-2 checkpoint (`--load_path .../ckpt_0020.pth`, README.md:44) populates them key for key:

    reference class                          file:line                               here
    PosEmbedding                             lab4d/nnutils/embedding.py:26-134       fourier_features()  (no parameters)
    InstEmbedding                            embedding.py:230-294                    InstanceCode
    TimeEmbedding                            embedding.py:137-227                    TimeEmbedding
    BaseMLP / CondMLP / ScaleLayer           lab4d/nnutils/base.py:8-157             DenseStack / CondDenseStack / ScaleLayer
    TimeMLP                                  lab4d/nnutils/time.py:11-133            TimeMLP
    CameraMLP                                lab4d/nnutils/pose.py:29-150            CameraMLP
    ArticulationFlatMLP                      pose.py:153-323                         ArticulationFlatMLP

Parameter and buffer NAMES (`linear_1.0.weight`, `linear_final.0.bias`, `time_embedding.mapping1.weight`,
`time_embedding.inst_embedding.mapping.weight`, `trans.3.scale`, `base_quat`, ...) and the values computed
from them are the reference's (tests/test_refpy_nets.py loads a state dict saved from the imported reference
modules with strict=True and compares every output with tests/golden/refpy_warp.npz); the code is written
from that contract, not transcribed.  Only what the forward Stage-3 path evaluates is here (no skeleton
articulation, no annealing window)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import quat_transform as qt


def make_frame_info(frame_offset, frame_offset_raw=None, frame_mapping=None) -> dict:
    """The metadata dict the reference passes around (`data_info["frame_info"]`): cumulative frame counts
    per video, the same for the unfiltered ("raw") frame ids, and the list of raw ids that are used."""
    fo = np.asarray(frame_offset, dtype=np.int64)
    fr = fo if frame_offset_raw is None else np.asarray(frame_offset_raw, dtype=np.int64)
    fm = list(range(int(fr[-1]))) if frame_mapping is None else [int(i) for i in frame_mapping]
    return {"frame_offset": fo, "frame_offset_raw": fr, "frame_mapping": fm}


def fourier_features(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    """(x, sin(2^k x), cos(2^k x))_{k < n_freq} along the last axis, per frequency the sines of all input
    channels then the cosines (embedding.py:70-109); n_freq = 0 is the identity, -1 drops the input."""
    if n_freq < 0:
        return x[..., :0]
    if n_freq == 0:
        return x
    bands = torch.pow(2.0, torch.arange(n_freq, dtype=x.dtype, device=x.device))  # 2 ** linspace(0, n-1, n)
    ang = bands[:, None] * x[..., None, :]                                            # (..., n_freq, C)
    waves = torch.stack((torch.sin(ang), torch.cos(ang)), dim=-2)                     # (..., n_freq, 2, C)
    return torch.cat((x, waves.flatten(-3)), dim=-1)


def fourier_dim(in_channels: int, n_freq: int) -> int:
    return 0 if n_freq < 0 else in_channels * (2 * n_freq + 1)


class InstanceCode(nn.Module):
    """One learnable code per video / object instance; a single-instance model answers every id with code 0."""

    def __init__(self, num_inst: int, channels: int):
        super().__init__()
        self.num_inst, self.out_channels = num_inst, channels
        if channels > 0:
            self.mapping = nn.Embedding(num_inst, channels)

    def forward(self, inst_id: torch.Tensor) -> torch.Tensor:
        if self.out_channels == 0:
            return torch.zeros(inst_id.shape + (0,), device=inst_id.device)
        return self.mapping(torch.zeros_like(inst_id) if self.num_inst == 1 else inst_id)

    def get_mean_embedding(self) -> torch.Tensor:
        return self.mapping.weight.mean(0)


class TimeEmbedding(nn.Module):
    """frame id -> code: Fourier features of the frame's position inside its video (scaled to [-1, 1] by the
    longest video) through `mapping1`, concatenated with the video's code, through `mapping2`."""

    def __init__(self, num_freq_t: int, frame_info: dict, out_channels: int = 128, time_scale: float = 1.0):
        super().__init__()
        self.num_freq_t, self.out_channels, self.time_scale = num_freq_t, out_channels, time_scale
        raw = np.asarray(frame_info["frame_offset_raw"], dtype=np.int64)
        self.frame_offset = frame_info["frame_offset"]
        self.num_frames = int(self.frame_offset[-1])
        self.num_vids = len(self.frame_offset) - 1
        self.max_ts = int((raw[1:] - raw[:-1]).max())
        vid_of = np.searchsorted(raw, np.arange(raw[-1]), side="right") - 1   # raw frame id -> video
        mapping = torch.as_tensor(list(frame_info["frame_mapping"]), dtype=torch.long)
        as_buf = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.long)  # noqa: E731
        # index tables: not part of the state dict (upstream registers them persistent=False as well)
        self.register_buffer("frame_mapping", mapping, persistent=False)
        self.register_buffer("raw_fid_to_vid", as_buf(vid_of), persistent=False)
        self.register_buffer("raw_fid_to_vstart", as_buf(raw[vid_of]), persistent=False)
        self.register_buffer("raw_fid_to_vidlen", as_buf(raw[vid_of + 1] - raw[vid_of]), persistent=False)
        self.register_buffer("frame_to_vid", as_buf(vid_of)[mapping], persistent=False)
        self.inst_embedding = InstanceCode(self.num_vids, out_channels)
        self.mapping1 = nn.Linear(fourier_dim(1, num_freq_t), out_channels)
        self.mapping2 = nn.Linear(2 * out_channels, out_channels)

    def frame_to_tid(self, frame_id: torch.Tensor) -> torch.Tensor:
        fid = frame_id.long()
        start, length = self.raw_fid_to_vstart[fid], self.raw_fid_to_vidlen[fid]
        return ((frame_id - start) - length / 2) / self.max_ts * 2 * self.time_scale

    def time_features(self, frame_id: torch.Tensor) -> torch.Tensor:
        """Fourier features of the frames' time coordinates: no parameter enters them, so integer ids look them up in a
        table built once per device by the very operations below (bit-identical; ~15 launches less per call, which is
        what a tiny-batch evaluation of the networks consists of)."""
        if frame_id.dtype in (torch.int64, torch.int32) and frame_id.dim() == 1:
            tab = self.__dict__.get("_time_feature_table")
            if tab is None or tab.device != frame_id.device:
                with torch.no_grad():
                    ids = torch.arange(self.raw_fid_to_vid.shape[0], device=frame_id.device)
                    tab = fourier_features(self.frame_to_tid(ids)[..., None], self.num_freq_t)
                self.__dict__["_time_feature_table"] = tab
            return tab[frame_id]
        return fourier_features(self.frame_to_tid(frame_id)[..., None], self.num_freq_t)

    def forward(self, frame_id=None) -> torch.Tensor:
        if frame_id is None:
            vid, frame_id = self.frame_to_vid, self.frame_mapping
        else:
            vid = self.raw_fid_to_vid[frame_id]
        coeff = self.mapping1(self.time_features(frame_id))
        return self.mapping2(torch.cat((coeff, self.inst_embedding(vid)), dim=-1))

    # The mean code over all kept frames (reference: lab4d/nnutils/embedding.py TimeEmbedding.get_mean_embedding =
    # forward(all frames).mean(0)).  Both layers are AFFINE with nothing in between, so the mean of the codes is the code of
    # the mean inputs: the mean Fourier features (no parameter enters them: a constant row) and the frame-weighted mean of the
    # videos' codes.  One row through the two layers instead of one per frame of the sequence -- with networks that train
    # the mean codes of the articulation and the skinning field are evaluated every step, forward and backward: 120-row
    # GEMMs, embedding backward, sums and fills, ~0.3 ms of a 2.5 ms step for two row vectors.  Equal to the mean of the
    # outputs up to fp32 rounding (1e-7 relative); LINEAR_MEAN = False restores the pass over all frames.
    LINEAR_MEAN = True

    def _mean_inputs(self, device):
        tab = self.__dict__.get("_mean_inputs_tab")
        if tab is None or tab[0].device != torch.device(device):
            with torch.no_grad():
                fm = self.frame_mapping.to(device)
                feat = self.time_features(fm).mean(0, keepdim=True)
                vid = self.frame_to_vid.to(device)
                w = torch.bincount(vid, minlength=self.num_vids).to(torch.float32) / float(vid.numel())
            tab = self.__dict__["_mean_inputs_tab"] = (feat, w[None])
        return tab

    def _mean_inst_code(self, w) -> torch.Tensor:
        ie = self.inst_embedding
        if ie.out_channels == 0:
            return torch.zeros(1, 0, device=w.device)
        return ie.mapping.weight[:1] if ie.num_inst == 1 else w @ ie.mapping.weight

    def get_mean_embedding(self, device=None) -> torch.Tensor:
        if not self.LINEAR_MEAN:
            return self.forward(self.frame_mapping).mean(0, keepdim=True)
        feat, w = self._mean_inputs(self.mapping1.weight.device if device is None else device)
        return self.mapping2(torch.cat((self.mapping1(feat), self._mean_inst_code(w)), dim=-1))

    def forward_and_mean(self, frame_id: torch.Tensor):
        """(codes of `frame_id` (M, C), mean code over all kept frames (1, C)) from ONE pass through the two layers."""
        M = frame_id.shape[0]
        if not self.LINEAR_MEAN:
            both = self.forward(torch.cat((frame_id.long(), self.frame_mapping)))
            return both[:M], both[M:].mean(0, keepdim=True)
        fid = frame_id.long()
        feat_m, w = self._mean_inputs(frame_id.device)
        feat = torch.cat((self.time_features(fid), feat_m))
        inst = torch.cat((self.inst_embedding(self.raw_fid_to_vid[fid]), self._mean_inst_code(w)))
        both = self.mapping2(torch.cat((self.mapping1(feat), inst), dim=-1))
        return both[:M], both[M:]


class ScaleLayer(nn.Module):
    def __init__(self, scale: float):
        super().__init__()
        self.register_buffer("scale", torch.FloatTensor([scale]))
        self.scale_value = float(scale)   # (host copy for the fused stack: reading the buffer would be a device sync)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.scale_value = float(self.scale)

    def forward(self, x):
        return x * self.scale


class DenseStack(nn.Module):
    """D hidden layers `linear_1 .. linear_D` (each Sequential(Linear, activation)) with the input re-fed at
    the layers listed in `skips`, then `linear_final` (with the activation when final_act)."""

    def __init__(self, D=8, W=256, in_channels=63, out_channels=3, skips=(4,), activation=None, final_act=False):
        super().__init__()
        act = activation if activation is not None else nn.ReLU(True)
        self.D, self.W, self.in_channels, self.out_channels, self.skips = D, W, in_channels, out_channels, tuple(skips)
        if in_channels == 0:
            return
        width = in_channels
        for i in range(D):
            fan_in = width + (in_channels if (i in self.skips and i > 0) else 0)
            setattr(self, f"linear_{i + 1}", nn.Sequential(nn.Linear(fan_in, W), act))
            width = W
        last = nn.Linear(W, out_channels)
        self.linear_final = nn.Sequential(last, act) if final_act else last

    def run_layers(self, x: torch.Tensor, first: int = 0) -> torch.Tensor:
        h = x
        for i in range(first, self.D):
            if i in self.skips:
                h = torch.cat((x, h), dim=-1)
            h = getattr(self, f"linear_{i + 1}")(h)
        return self.linear_final(h)

    def forward(self, x):
        return self.run_layers(x)


class CondDenseStack(DenseStack):
    """DenseStack on [features, instance code]."""

    def __init__(self, num_inst, D=8, W=256, in_channels=63, inst_channels=32, out_channels=3, skips=(4,),
                 activation=None, final_act=False):
        super().__init__(D=D, W=W, in_channels=in_channels + inst_channels, out_channels=out_channels, skips=skips,
                         activation=activation, final_act=final_act)
        self.inst_embedding = InstanceCode(num_inst, inst_channels)

    def instance_code(self, inst_id, lead_shape, device) -> torch.Tensor:
        """(M, C) or (1, C): the code per leading batch entry (mean code when inst_id is None)."""
        if self.inst_embedding.out_channels == 0:
            return torch.zeros(1, 0, device=device)
        if inst_id is None:
            return self.inst_embedding.get_mean_embedding()[None]
        return self.inst_embedding(inst_id)

    def forward(self, feat: torch.Tensor, inst_id) -> torch.Tensor:
        code = self.instance_code(inst_id, feat.shape[:-1], feat.device)
        code = code.view(code.shape[:1] + (1,) * (feat.dim() - 2) + code.shape[-1:]).expand(feat.shape[:-1] + (-1,))
        both = torch.cat((feat, code), dim=-1)
        return both if both.shape[-1] == 0 else self.run_layers(both)


def scaled_num_freq(frame_info: dict, num_freq_t: int) -> int:
    """The longest video sets the time bandwidth: 64 frames <-> the nominal count (time.py:34-41)."""
    if num_freq_t <= 0:
        return num_freq_t
    fo = np.asarray(frame_info["frame_offset"])
    return int(np.rint(np.log2((fo[1:] - fo[:-1]).max() / 64) + num_freq_t))


class TimeMLP(DenseStack):
    """frame id -> W features: TimeEmbedding followed by a W-wide stack (final activation on)."""

    def __init__(self, frame_info, D=5, W=256, num_freq_t=6, skips=(), activation=None, time_scale=1.0):
        super().__init__(D=D, W=W, in_channels=W, out_channels=W, skips=skips, activation=activation, final_act=True)
        self.time_embedding = TimeEmbedding(scaled_num_freq(frame_info, num_freq_t), frame_info, out_channels=W,
                                            time_scale=time_scale)

    def features(self, t_embed: torch.Tensor) -> torch.Tensor:
        return self.run_layers(t_embed)

    def fused_heads(self, t_embed: torch.Tensor, head_a, head_b):
        """(head_a(features), head_b(features)) through csrc/dense_stack.hip -- the whole stack in one launch per direction --
        or None when the stack is not of the plain kind (re-fed inputs, another activation, more than 16 rows, CPU)."""
        from . import dense_stack as ds
        if t_embed.dim() != 2 or any(0 < s < self.D for s in self.skips):
            return None
        trunk = []
        for i in range(self.D):
            lay = ds.sequential_layers(getattr(self, f"linear_{i + 1}"))
            if lay is None or len(lay) != 1 or not lay[0][1]:
                return None
            trunk += lay
        last = ds.sequential_layers(self.linear_final)
        a, b = ds.sequential_layers(head_a), ds.sequential_layers(head_b)
        if last is None or len(last) != 1 or a is None or b is None or not ds.supported(t_embed, trunk + last, a, b):
            return None
        return ds.dense_stack(t_embed, trunk + last, a, b)

    def get_frame_offset(self):
        return self.time_embedding.frame_offset

    def fit_to(self, target_fn, loss_fn=None, termination_loss=1e-4, max_iters=20000, lr=1e-3):
        """Adam until the prediction matches the prior (TimeMLP.mlp_init, time.py:77-99)."""
        opt = torch.optim.Adam(self.parameters(), lr=lr)
        for it in range(max_iters):
            opt.zero_grad()
            loss = target_fn()
            loss.backward()
            opt.step()
            if loss.item() < termination_loss:
                break
        return loss.item()


def _head(W: int, out: int, act, scale=None) -> nn.Sequential:
    layers = [nn.Linear(W, W // 2), act, nn.Linear(W // 2, out)]
    if scale is not None:
        layers.append(ScaleLayer(scale))
    return nn.Sequential(*layers)


class CameraMLP(TimeMLP):
    """frame id -> object-to-camera (unit quaternion, translation): two heads on the time features, the
    rotation composed with a learnable per-video base rotation."""

    def __init__(self, rtmat, frame_info=None, D=5, W=256, num_freq_t=6, skips=(), activation=None):
        rtmat = torch.as_tensor(np.asarray(rtmat), dtype=torch.float32)
        if frame_info is None:
            frame_info = make_frame_info([0, len(rtmat)])
        act = activation if activation is not None else nn.ReLU(True)
        super().__init__(frame_info, D=D, W=W, num_freq_t=num_freq_t, skips=skips, activation=act)
        self.trans = _head(W, 3, act)
        self.quat = _head(W, 4, act)
        self.base_quat = nn.Parameter(torch.zeros(self.time_embedding.num_vids, 4))
        self.register_buffer("init_vals", rtmat, persistent=False)

    def head_outputs(self, t_embed, fused=False):
        """(raw rotation quaternion (R, 4), translation (R, 3)): the two heads' outputs."""
        both = self.fused_heads(t_embed, self.quat, self.trans) if fused else None
        if both is None:
            feat = self.features(t_embed)
            both = self.quat(feat), self.trans(feat)
        return both

    def forward(self, t_embed, fused=False):
        quat, trans = self.head_outputs(t_embed, fused)
        return F.normalize(quat, dim=-1), trans

    def get_vals(self, frame_id=None, fused=False):
        te = self.time_embedding
        vid = te.frame_to_vid if frame_id is None else te.raw_fid_to_vid[frame_id]
        if fused and frame_id is not None and frame_id.is_cuda and frame_id.dim() == 1:
            # both normalisations and the product in one launch per direction (csrc/bone_tables.hip: ~10 + ~30 otherwise)
            from .bone_tables import camera_tail
            quat, trans = self.head_outputs(te(frame_id), fused=True)
            return camera_tail(quat, self.base_quat[vid]), trans
        quat, trans = self.forward(te(frame_id), fused=fused)
        return qt.quaternion_mul(quat, F.normalize(self.base_quat[vid], dim=-1)), trans

    def base_init(self):
        """Per-video base rotation = the prior's rotation at the video's first frame (pose.py:95-101)."""
        first = torch.as_tensor(np.asarray(self.get_frame_offset()[:-1]), dtype=torch.long)
        self.base_quat.data = qt.matrix_to_quaternion(self.init_vals[first, :3, :3])

    def prior_loss(self):
        q, t = self.get_vals()
        return F.mse_loss(qt.quaternion_translation_to_se3(q, t), self.init_vals)

    def mlp_init(self, termination_loss=1e-4, max_iters=20000):
        self.base_init()
        return self.fit_to(self.prior_loss, termination_loss=termination_loss, max_iters=max_iters)


class ArticulationFlatMLP(TimeMLP):
    """frame id -> B bone-to-object rigid transforms as dual quaternions (bag of bones): axis-angle and
    0.1-scaled translation heads on the time features."""

    def __init__(self, frame_info, num_se3, D=5, W=256, num_freq_t=6, skips=(), activation=None):
        act = activation if activation is not None else nn.ReLU(True)
        super().__init__(frame_info, D=D, W=W, num_freq_t=num_freq_t, skips=skips, activation=act)
        self.edges = None
        self.num_se3 = num_se3
        self.trans = _head(W, 3 * num_se3, act, scale=0.1)
        self.so3 = _head(W, 3 * num_se3, act)

    def head_outputs(self, t_embed, fused=False):
        """(axis-angle (..., B, 3), translation (..., B, 3)): what the two heads emit, before any quaternion algebra."""
        lead = t_embed.shape[:-1]
        both = self.fused_heads(t_embed, self.so3, self.trans) if fused else None
        if both is None:
            feat = self.features(t_embed)
            both = self.so3(feat), self.trans(feat)
        return both[0].reshape(*lead, self.num_se3, 3), both[1].reshape(*lead, self.num_se3, 3)

    def forward(self, t_embed, inst_id=None):
        so3, trans = self.head_outputs(t_embed)
        return qt.quaternion_translation_to_dual_quaternion(qt.axis_angle_to_quaternion(so3), trans)

    def get_vals(self, frame_id=None):
        return self.forward(self.time_embedding(frame_id))

    def get_mean_vals(self, inst_id=None):
        return self.forward(self.time_embedding.get_mean_embedding())

    def get_vals_and_mean(self, frame_id=None):
        if frame_id is not None and frame_id.dim() == 1 and not frame_id.is_floating_point():
            # the frames' rows and the mean code's row through the stack and the heads TOGETHER: rows of a dense layer
            # are independent, and a second pass of ~100 launches on one row is what it would cost
            M = frame_id.shape[0]
            qr, qd = self.forward(torch.cat(self.time_embedding.forward_and_mean(frame_id)))
            return (qr[:M], qd[:M]), (qr[M:].expand(M, -1, -1).contiguous(), qd[M:].expand(M, -1, -1).contiguous())
        at_t = self.get_vals(frame_id)
        rest = self.get_mean_vals()
        return at_t, (rest[0].expand_as(at_t[0]).contiguous(), rest[1].expand_as(at_t[1]).contiguous())
