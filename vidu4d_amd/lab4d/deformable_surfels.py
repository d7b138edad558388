"""DeformableSurfels: the Stage-3 4D surfel field (canonical surfels -> bob LBS warp -> camera space
-> per-frame rasterization), i.e. the hot part of `DeformableGaussian`
(reference: lab4d/nnutils/deformable_gaussian.py: __init__ :87-161, get_xyz/get_rotation overrides
:163-176, render_view :178-202, get_gs_Kcamera :927-962, apply_qt_to_gaussian :1032-1046,
query_field render loop :1175-1233, forward_warp :1395-1434).

Out of scope here (auxiliary losses dropped by --rgb_loss_only, trainer.py:477-483): cycle loss,
feature matching / reprojection, flow rendering, bone-density visualisation."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ..gs.cameras import KCamera
from ..gs.gaussian_model import GaussianModel
from ..gs.gaussian_renderer import GEOMETRY_KEYS, render
from . import quat_transform as qt
from .bob_warp import apply_qt_to_gaussian, create_warp, cross_entropy_skin_loss, feature_major_linear
from .bone_tables import bone_tables
from .lbs_fused import (lbs_apply, lbs_skin_apply, prepare_skin_field, skin_field, skin_field_shape_supported,
                        skin_field_supported, skin_field_train)
from .net_graphs import GraphedNetworks
from .nets import CameraMLP, make_frame_info


RASTER_MAX_FRAMES = 8   # (VIDU4D_SURFEL_MAX_FRAMES: frames of one stacked launch set)


class PipelineParams:
    """2DGS PipelineParams defaults (gs/arguments/__init__.py:64-70)."""
    convert_SHs_python = False
    compute_cov3D_python = False
    depth_ratio = 0.0
    debug = False


class PointCloud:
    def __init__(self, points, colors):
        self.points, self.colors = points, colors


def Kmatinv(Kmat):
    """(…,3,3) intrinsics -> inverse (lab4d/utils/geom_utils.py Kmatinv)."""
    return torch.inverse(Kmat)


class _WarpNetEval(nn.Module):
    """What the fused warp needs from the networks, as ONE callable of the step's frame ids (for graph capture,
    DeformableSurfels._graphed_warp_networks): the bones' dual quaternions relative to the rest pose, the cameras, the rest
    pose's bone map and the mean time code's first-layer bias -- the same calls forward_warp_fused makes eagerly."""

    def __init__(self, warp, camera_mlp, fused_tables=True, branches=True, fused_stacks=True):
        super().__init__()
        self.warp, self.camera_mlp, self.fused_tables, self.branches = warp, camera_mlp, fused_tables, branches
        self.fused_stacks = fused_stacks   # (the networks' dense layers as one launch per direction, csrc/dense_stack.hip)

    def _side_streams(self, device):
        st = self.__dict__.get("_streams")
        if st is None or st[0].device != device:
            st = self.__dict__["_streams"] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        return st

    def forward(self, frame_id):
        w = self.warp
        sm = w.skinning_model
        art = w.articulation
        if self.branches and frame_id.is_cuda:
            # The camera network, the skinning field's mean time code and the articulation network are three independent
            # chains of ~6 us launches that occupy a compute unit each: they run on three HIP streams (forked from and
            # joined to the calling one), so a captured graph has three branches and its critical path is the longest of
            # them instead of their sum -- forward AND backward: autograd runs a node on its forward's stream.
            main = torch.cuda.current_stream(frame_id.device)
            s_cam, s_bias = self._side_streams(frame_id.device)
            s_cam.wait_stream(main)
            with torch.cuda.stream(s_cam):
                cq, ct = self.camera_mlp.get_vals(frame_id, fused=self.fused_stacks)
                cq, ct = cq.contiguous(), ct.contiguous()
            bias = None
            if sm.has_delta:
                s_bias.wait_stream(main)
                with torch.cuda.stream(s_bias):
                    bias = sm.frame_bias(None, None, 1, frame_id.device).contiguous()
        else:
            main = cq = bias = None
        if self.fused_tables and frame_id.is_cuda and hasattr(art, "head_outputs"):
            # the heads' outputs for the frames and for the mean code in one pass, then ONE kernel per direction from there
            # to the relative bone transforms and the rest pose's scaled bone map (csrc/bone_tables.hip) instead of the
            # quaternion algebra as ~75 + ~175 elementwise launches
            M = frame_id.shape[0]
            so3, trans = art.head_outputs(torch.cat(art.time_embedding.forward_and_mean(frame_id)), fused=self.fused_stacks)
            se3_qr, se3_qd, A, c0 = bone_tables(so3[:M], trans[:M], so3[M], trans[M], 1.0 / sm.get_gauss())
            se3 = (se3_qr, se3_qd)
        else:
            t_art, rest_art = art.get_vals_and_mean(frame_id)
            se3 = qt.dual_quaternion_mul(t_art, qt.dual_quaternion_inverse(rest_art))
            A, c0 = sm.bone_affine((rest_art[0][:1], rest_art[1][:1]))
        if cq is None:
            cq, ct = self.camera_mlp.get_vals(frame_id)
            bias = sm.frame_bias(None, None, 1, frame_id.device) if sm.has_delta else None
        else:
            main.wait_stream(s_cam)
            cq.record_stream(main), ct.record_stream(main)
            if bias is not None:
                main.wait_stream(s_bias)
                bias.record_stream(main)
        out = (se3[0].contiguous(), se3[1].contiguous(), cq.contiguous(), ct.contiguous(), A.contiguous(), c0.contiguous())
        if sm.has_delta:
            out = out + (bias.contiguous(),)
        return out


class DeformableSurfels(GaussianModel):
    """State-dict keys under `fields.field_params.fg.` as upstream: `_xyz`, `_features_dc`, `_features_rest`,
    `_scaling`, `_rotation`, `_opacity`, `_regist_feat`, `logsigma`, `logibeta`, `aabb`, `learnable_bkgd`,
    `warp.*` (bob_warp.SkinningWarp), `camera_mlp.*` (nets.CameraMLP)."""

    def __init__(self, opts: dict, num_frames: int, device="cuda", data_info: dict | None = None):
        super().__init__(opts.get("sh_degree", 3), device=device)
        self.opts = opts
        data_info = dict(data_info or {})
        if "frame_info" not in data_info:
            data_info["frame_info"] = make_frame_info([0, num_frames])
        frame_info = data_info["frame_info"]
        self.frame_offset = frame_info["frame_offset"]
        self.frame_offset_raw = frame_info["frame_offset_raw"]
        self.num_frames = int(self.frame_offset[-1])
        self.num_inst = 1
        motion = opts.get("fg_motion", "gs-bob")
        assert motion.startswith("gs-"), motion
        self.fg_motion = motion[3:]
        self.warp = create_warp(self.fg_motion, data_info)
        self.logsigma = nn.Parameter(torch.tensor([1.0]).log())
        self.logibeta = nn.Parameter(-torch.tensor([0.1]).log())
        rtmat = data_info.get("rtmat")
        synthetic_cam = rtmat is None
        if synthetic_cam:  # no camera prior: every frame looks at the object from 3 units away
            rtmat = torch.eye(4).repeat(int(self.frame_offset_raw[-1]), 1, 1)
            rtmat[:, 2, 3] = 3.0
        else:  # dataset cameras: translations enter at the field's scale (deformable_gaussian.py:93, :123)
            rtmat = torch.as_tensor(np.asarray(rtmat), dtype=torch.float32).clone()
            rtmat[..., :3, 3] *= float(opts.get("init_scale", 0.1))
        self.camera_mlp = CameraMLP(rtmat, frame_info=frame_info)
        if synthetic_cam:
            self.camera_mlp.base_init()
            with torch.no_grad():  # an untrained translation head near the prior, so that synthetic runs see the object
                self.camera_mlp.trans[2].weight.mul_(0.1)
                self.camera_mlp.trans[2].bias.copy_(torch.tensor([0.0, 0.0, 3.0]))
        self.register_buffer("aabb", torch.zeros(2, 3))
        self.pipeline = PipelineParams()
        # host-side state of this model's rasterizer calls (vidu4d_amd/_C.py RasterContext); a plain attribute, not a
        # parameter / buffer: it is not part of the state dict
        from .. import _C as _native
        self.__dict__["raster_context"] = _native.RasterContext()
        self.pipeline.debug = opts.get("debug_cuda", False)
        self.background_feat = torch.zeros(3, device=self.device_)  # what render_view is handed upstream (:147, :1190)
        if opts.get("gs_learnable_bg", True):
            self.learnable_bkgd = nn.Parameter(torch.tensor([0.5, 0.5, 0.5]))
        self.cameras_extent = opts.get("cameras_extent", 1.0)
        self.to(self.device_)

    # ---- initialisation from a point sample of the Stage-2 proxy mesh (init_proxy :354-409)
    def init_from_points(self, points, colors, feat_channels: int = 16):
        import numpy as np
        pts = np.asarray(points)
        # scene radius as upstream (:402-407); it scales the densify / prune size thresholds
        self.cameras_extent = float(np.linalg.norm(pts - pts.mean(axis=0, keepdims=True), axis=-1).max() * 1.1)
        self.create_from_pcd(PointCloud(points, colors), spatial_lr_scale=self.cameras_extent)
        self._regist_feat = nn.Parameter(torch.zeros(self._xyz.shape[0], feat_channels, device=self._xyz.device))
        self.training_setup(_Args(self.opts))

    # ---- overrides used while rendering one warped frame (:163-176)
    @property
    def get_rotation(self):
        if hasattr(self, "_override_rotation"):
            if self.__dict__.get("_override_rotation_is_unit", False):
                return self._override_rotation  # (activated by the warp kernel: lbs_skin_apply(unit_rot=True))
            return self.rotation_activation(self._override_rotation)
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._override_xyz if hasattr(self, "_override_xyz") else self._xyz

    # The frames of a step share the canonical appearance: the activated copies (one exp, one sigmoid, one (N,16,3)
    # concatenation of the SH rows) are made once per step by render_frames, not once per frame; autograd sums
    # the frames' gradients into them.
    def _shared(self, key, make):
        cache = self.__dict__.get("_step_cache")
        if cache is None:
            return make()
        if key not in cache:
            cache[key] = make()
        return cache[key]

    @property
    def get_scaling(self):
        return self._shared("scaling", lambda: self.scaling_activation(self._scaling))

    @property
    def get_opacity(self):
        return self._shared("opacity", lambda: self.opacity_activation(self._opacity))

    @property
    def get_features(self):
        return self._shared("features", lambda: torch.cat((self._features_dc, self._features_rest), dim=1))

    def render_view(self, view, override_xyz=None, override_rotation=None, override_color=None, override_bkgd=None,
                    outputs=None):
        if override_xyz is not None:
            assert override_xyz.dim() == 2
            self._override_xyz = override_xyz
            self._override_rotation = override_rotation
        try:
            bkgd = self.background_feat if override_bkgd is None else override_bkgd
            rendered = render(view, self, self.pipeline, bkgd, override_color=override_color, outputs=outputs)
        finally:
            if override_xyz is not None:
                del self._override_xyz
                del self._override_rotation
        if outputs is not None and "raw" in outputs:
            return rendered  # (the caller composites the learnable background itself: lab4d/loss_fused.py)
        if hasattr(self, "learnable_bkgd"):
            rendered["render"] = rendered["render"] + (1 - rendered["acc"]) * self.learnable_bkgd[:, None, None]
        return rendered

    def get_gs_Kcamera(self, Kinvs, Hs, Ws):
        """One camera per frame from the inverse intrinsics (:927-962).  Cameras are built on the host from six
        numbers and cached on them (with --force_center_cam every frame of a video shares them).  Hand the
        intrinsics over as a HOST tensor (vidloader and synthetic_batch do): a device tensor has to be copied
        back, which makes the host wait for all queued GPU work once per step (only skipped when the very same
        tensor comes again)."""
        # the same intrinsics tensor as last time (same storage, not written since): same cameras, and no
        # device-to-host copy -- that copy would make the host wait for all queued GPU work every step
        tkey = (Kinvs.data_ptr(), Kinvs._version, tuple(Kinvs.shape), tuple(int(h) for h in Hs), tuple(int(w) for w in Ws))
        last = self.__dict__.get("_camera_last")
        if last is not None and last[0] == tkey:
            return last[1]
        cams = []
        Kh = Kinvs.detach().float().cpu()
        cache = self.__dict__.setdefault("_camera_cache", {})
        # (the model's device, also when it holds no surfel: a fit whose outlier pass pruned everything -- upstream's fixed
        # radius 0.004 / 20 neighbours, trainer.py:573-588, does that to a sparse toy cloud -- must not move to the host)
        own = getattr(self, "device_", None)
        if isinstance(getattr(self, "_xyz", None), torch.Tensor) and (self._xyz.is_cuda or self._xyz.numel()):
            dev = self._xyz.device
        else:
            dev = own if own is not None and own.type == "cuda" else Kinvs.device
        for i in range(Kh.shape[0]):
            Kinv, H, W = Kh[i], int(Hs[i]), int(Ws[i])
            left, right = Kinv[0, 2], Kinv[0, 2] + Kinv[0, 0] * W
            bottom, top = Kinv[1, 2], Kinv[1, 2] + Kinv[1, 1] * H
            key = (H, W, float(left), float(right), float(top), float(bottom), str(dev))
            if key not in cache:
                if len(cache) > 64:
                    cache.clear()
                cache[key] = KCamera(H=H, W=W, left=left, right=right, top=top, bottom=bottom,
                                     data_device=dev)
            cams.append(cache[key])
        self.__dict__["_camera_last"] = (tkey, cams, Kinvs)  # (keeps the tensor alive: its address is the key)
        return cams

    def forward_warp(self, xyz, rotation, frame_id, inst_id=None, samples_dict=None):
        """Canonical -> time t (bob LBS) -> camera space.  xyz (M,N,1,3), rotation (M,N,4)."""
        samples_dict = samples_dict or {}
        M = frame_id.shape[0]
        (q, t), aux = self.warp(xyz, frame_id, inst_id, samples_dict=samples_dict, return_qt=True, return_aux=True)
        xyz_t, rot_t = apply_qt_to_gaussian(xyz, rotation, q, t, M)
        cq, ct = samples_dict["field2cam"] if "field2cam" in samples_dict else self.camera_mlp.get_vals(frame_id)
        N = xyz.shape[1]
        cq = cq[:, None].expand(-1, N, -1)
        ct = ct[:, None].expand(-1, N, -1)
        xyz_cam, rot_cam = apply_qt_to_gaussian(xyz_t, rot_t, cq, ct, M)
        self._aux_dict = aux
        return xyz_cam, rot_cam, (q, t)

    def _warp_param_list(self):
        """The warp's and the camera's parameters as a plain list, collected once: `Module.parameters()` walks the module
        tree (0.2 ms for the bob networks -- it was asked two or three times per step, a third of the step's host time,
        tools/fit_host_probe.py)."""
        lst = self.__dict__.get("_warp_params")
        if lst is None:
            lst = self.__dict__["_warp_params"] = [p for mod in (self.warp, self.camera_mlp) for p in mod.parameters()]
        return lst

    def _graphed_warp_networks(self, frame_id):
        """frame ids (M,) -> (se3_qr, se3_qd, cam_q, cam_t, bone_A, bone_c[, frame_bias]) through hipGraphs of the networks'
        forward and backward (captured on first use and whenever M or the set of trainable parameters changes)."""
        if self.opts.get("graphed_warp_networks", True) == "inline":
            # (a caller that captures the WHOLE step -- lab4d/captured_step.py -- wants the same launches, three branches and
            # fused stacks, as nodes of ITS graph: the module itself, under autograd)
            mod = self.__dict__.get("_net_eval_inline")
            if mod is None:
                mod = self.__dict__["_net_eval_inline"] = _WarpNetEval(
                    self.warp, self.camera_mlp, bool(self.opts.get("fused_bone_tables", True)),
                    bool(self.opts.get("parallel_network_branches", True)), bool(self.opts.get("fused_dense_stacks", True)))
            return mod(frame_id)
        key = (int(frame_id.shape[0]), tuple(p.requires_grad for p in self._warp_param_list()))
        g = self.__dict__.get("_net_graph")
        if g is None or g[0] != key:
            mod = _WarpNetEval(self.warp, self.camera_mlp, bool(self.opts.get("fused_bone_tables", True)),
                               bool(self.opts.get("parallel_network_branches", True)),
                               bool(self.opts.get("fused_dense_stacks", True)))
            try:
                if self.opts.get("graphed_warp_networks", True) == "torch":   # (A/B: gradients through AccumulateGrad copies)
                    fn = torch.cuda.make_graphed_callables(mod, (frame_id.clone(),), allow_unused_input=True)
                else:
                    fn = GraphedNetworks(mod, frame_id, self._warp_param_list())
            except Exception as e:   # (capture is an optimisation: anything it cannot take runs eagerly, said once)
                print(f"graphed_warp_networks: capture failed ({type(e).__name__}: {e}); evaluating the networks eagerly")
                fn = None
            g = self.__dict__["_net_graph"] = (key, fn)
        return None if g[1] is None else g[1](frame_id)

    def warp_networks_train(self) -> bool:
        return any(p.requires_grad for p in self._warp_param_list())

    def fused_warp_ok(self, inst_id=None) -> bool:
        """The fused HIP warp applies when all frames of the batch share one instance code.  Frozen bone and camera
        networks (--gs_optim_warp=False, the README's Stage-3 command) get the cached tables and the MFMA skinning field;
        networks that TRAIN (--gs_optim_warp=True, the reference's default, lab4d/config.py:157: AdamW on them,
        trainer.py:592-598) are evaluated with autograd for the step's frames and the skinning kernel's backward reduces
        the gradients w.r.t. their bone dual quaternions and cameras (round 5; `fused_warp_trainable: False` restores the
        torch chain of rounds 1-4 for an A/B)."""
        if not self._xyz.is_cuda or not self.opts.get("fused_warp", True):
            return False
        if self.warp_networks_train() and not self.opts.get("fused_warp_trainable", True):
            return False
        return inst_id is None or len(set(inst_id.tolist())) == 1

    def _frozen_warp_table(self):
        version = sum(p._version for p in self._warp_param_list())
        tab = self.__dict__.get("_warp_table")
        if tab is None or tab["version"] != version:
            with torch.no_grad():
                # frame ids are RAW ids (vidloader.stage3_batch: frame_map[idx] + frame_offset_raw[vid]; the time
                # embeddings are raw-indexed, nets.TimeEmbedding): the table covers every raw frame, not just the kept ones
                ids = torch.arange(int(self.frame_offset_raw[-1]), device=self._xyz.device)
                t_art, rest_art = self.warp.articulation.get_vals_and_mean(ids)
                se3 = qt.dual_quaternion_mul(t_art, qt.dual_quaternion_inverse(rest_art))
                cq, ct = self.camera_mlp.get_vals(ids)
                rest1 = (rest_art[0][:1].contiguous(), rest_art[1][:1].contiguous())
                sm = self.warp.skinning_model
                A, c0 = sm.bone_affine(rest1)
                tab = {"version": version, "se3_qr": se3[0].contiguous(), "se3_qd": se3[1].contiguous(),
                       "rest1": rest1, "bone_frames": sm.bone_frames(rest1), "bone_A": A.contiguous(),
                       "bone_c": c0.contiguous(),
                       "frame_bias": sm.frame_bias(None, None, 1, ids.device) if sm.has_delta else None,
                       # ... and with every instance code: the mean time code behind it is an MLP over ALL frames
                       "frame_bias_inst": self._frame_bias_per_instance(sm, ids.device),
                       "cam_q": cq.contiguous(), "cam_t": ct.contiguous()}
            self.__dict__["_warp_table"] = tab
        return tab

    @staticmethod
    def _frame_bias_per_instance(sm, device):
        if not sm.has_delta:
            return None
        emb = getattr(sm.delta_field, "inst_embedding", None)
        if emb is None or emb.out_channels == 0:
            return None
        return sm.frame_bias(None, torch.arange(emb.mapping.weight.shape[0], device=device), 1, device).contiguous()

    @staticmethod
    def _check_frame_ids(frame_id, rows):
        """The kernels index the frozen networks' tables with the frame ids themselves and CLAMP an id outside them to the
        first / last row (csrc/lbs.hip table_row) -- the torch indexing they replace, and the reference, raise (a
        bad id would render and train against another frame's bones and camera without a word).  Checked here, once per
        tensor OBJECT: the producers of frame batches note the range of theirs on the host (`_vidu4d_host_range`:
        vidloader.stage3_batch / sequence_batch, stage3.synthetic_batch), anything else costs one device read the first time
        it is seen."""
        if getattr(frame_id, "_vidu4d_rows_checked", 0) >= rows or frame_id.numel() == 0:
            return
        host = getattr(frame_id, "_vidu4d_host_range", None)   # (lo, hi) noted by whoever built the ids on the host
        lo, hi = host if host is not None else (int(frame_id.min()), int(frame_id.max()))
        if lo < 0 or hi >= rows:
            raise IndexError(f"frame id out of range: ids span [{lo}, {hi}], the sequence has {rows} frames")
        frame_id._vidu4d_rows_checked = rows

    def forward_warp_fused(self, frame_id, inst_id=None, samples_dict=None):
        """forward_warp for frozen bones: the skinning weights of the forward warp depend on neither the
        frame nor the time code (warping.py:415-425: rest articulation, mean time embedding), so they are
        evaluated once for the step; blend + apply + camera transform run in one HIP kernel per
        direction (csrc/lbs.hip).  -> xyz_cam (M,N,3), rot_cam (M,N,4)."""
        samples_dict = samples_dict or {}
        w = self.warp
        table_rows = None   # (frame ids when se3 / cq / ct below are whole tables)
        self.__dict__["_warp_rot_is_unit"] = False
        # networks that train are evaluated for the step's frames WITH autograd (no cached table): articulation, camera,
        # the rest pose's bone map and the time-code bias are a handful of launches on (M, B, .) tensors; the delta-skin MLP
        # runs as feature-major library GEMMs (its weight gradients are GEMMs over the surfels); the skinning kernel's
        # backward hands back d/d se3 and d/d camera
        trainable = self.warp_networks_train()
        overrides = trainable or any(k in samples_dict for k in ("rest_articulation", "t_articulation", "field2cam"))
        if overrides:
            graphed = None
            if (trainable and not samples_dict and inst_id is None and frame_id.is_cuda and frame_id.dtype == torch.int64
                    and torch.is_grad_enabled() and self.opts.get("graphed_warp_networks", True)
                    and frame_id.shape[0] <= 8 and self.opts.get("fused_skin", True) and not self.opts.get("warp_aux", False)
                    and (not w.skinning_model.has_delta or w.skinning_model.num_freq_xyz == 0)):
                # networks that train: their forward for the step's frames (and, for the rest pose, for the mean over all
                # frames) is ~450 small launches and their backward ~550 autograd nodes -- 9 of the step's 12 ms of HOST
                # time (tools/fit_host_probe_optim_warp.py).  Shapes are static, so both directions are captured into
                # hipGraphs once (torch.cuda.make_graphed_callables) and replayed: two graph launches per step.
                graphed = self._graphed_warp_networks(frame_id)
            if graphed is not None:
                se3, cq, ct = (graphed[0], graphed[1]), graphed[2], graphed[3]
                rest1 = None
            elif "rest_articulation" in samples_dict and "t_articulation" in samples_dict:
                rest_art, t_art = samples_dict["rest_articulation"], samples_dict["t_articulation"]
            else:
                t_art, rest_art = w.articulation.get_vals_and_mean(frame_id)
            if graphed is None:
                se3 = qt.dual_quaternion_mul(t_art, qt.dual_quaternion_inverse(rest_art))
                rest1 = (rest_art[0][:1], rest_art[1][:1])
                cq, ct = samples_dict["field2cam"] if "field2cam" in samples_dict else self.camera_mlp.get_vals(frame_id)
        else:
            # frozen networks: bone transforms and cameras of ALL frames are constants of the run; they are
            # evaluated once (and again whenever a parameter is written) and indexed per step
            tab = self._frozen_warp_table()
            rest1 = tab["rest1"]
            # (the fused skinning kernel below indexes the tables itself -- `frame_index` -- instead of four row gathers per step)
            in_kernel = (frame_id.dtype == torch.int64 and self.opts.get("fused_skin", True)
                         and self.opts.get("table_index_in_kernel", True))
            if in_kernel:
                self._check_frame_ids(frame_id, int(tab["se3_qr"].shape[0]))
                se3, (cq, ct), table_rows = (tab["se3_qr"], tab["se3_qd"]), (tab["cam_q"], tab["cam_t"]), frame_id
            else:
                se3 = (tab["se3_qr"][frame_id], tab["se3_qd"][frame_id])
                cq, ct = tab["cam_q"][frame_id], tab["cam_t"][frame_id]
        sm = w.skinning_model
        M = frame_id.shape[0]
        iid = None if inst_id is None else inst_id[:1]
        if overrides and rest1 is None:   # (from the captured graphs)
            A, c0 = graphed[4], graphed[5]
            bias = graphed[6] if sm.has_delta else None
        elif overrides:
            A, c0 = sm.bone_affine(rest1)
            bias = sm.frame_bias(None, iid, 1, self._xyz.device) if sm.has_delta else None
        else:
            A, c0, bias = tab["bone_A"], tab["bone_c"], tab["frame_bias"]   # (bias: mean instance code)
            if sm.has_delta and iid is not None:
                per_inst = tab["frame_bias_inst"]   # (instances, W): a row look-up instead of the time-code MLP per step
                if per_inst is None or not self.opts.get("cached_frame_bias", True):
                    bias = sm.frame_bias(None, iid, 1, self._xyz.device)
                else:  # (a single-instance model answers every id with its one code: nets.InstanceCode)
                    bias = per_inst if per_inst.shape[0] == 1 else per_inst[iid]
        # (any number of frames: lbs_skin_apply runs more than 8 in groups of 8 -- VERDICT r5 missing 3)
        if self.opts.get("fused_skin", True) and (not sm.has_delta or sm.num_freq_xyz == 0):
            # bone coordinates and the delta MLP as feature-major GEMMs (4 library calls), everything else -- distances,
            # relu * 0.1, softmax, blend, apply, camera, for all frames -- in one HIP kernel per direction
            # (partial freeze, ADVICE r5: a frozen skinning field under an articulation / time code that TRAINS -- the bone map
            # or the first-layer bias then require grad, which the frozen instances take as constants: the TRAIN instances below
            # serve it, with no weight gradient asked for)
            frozen_inputs = not (overrides and any(t is not None and t.requires_grad for t in (A, c0, bias)))
            if self.opts.get("fused_skin_field", True) and skin_field_supported(sm) and frozen_inputs:
                # ... and with frozen weights those GEMMs too: one thread carries a surfel through bone map and MLP
                # (csrc/skin_field.hip), no hidden activation ever reaches HBM
                if overrides:
                    sf_tab = prepare_skin_field(sm, A, c0)
                else:
                    sf_tab = tab.get("skin_field")
                    if sf_tab is None:
                        sf_tab = tab["skin_field"] = prepare_skin_field(sm, A, c0)
                # (the bone coordinates themselves are re-evaluated by the skinning kernel from the same (3B, 3) map: the
                # (3B, N) array and its gradient never cross HBM)
                bone_map = (sf_tab["bone_A"], sf_tab["bone_c"]) if self.opts.get("fused_bone_map", True) else None
                xbT, rawT = skin_field(self._xyz, bias[0].detach(), sf_tab, want_xb=bone_map is None)
            elif (overrides and sm.has_delta and self.opts.get("fused_skin_field_trainable", True)
                  and skin_field_shape_supported(sm)):
                # weights that TRAIN: the same MFMA kernels in their TRAIN instances (hidden activations and masked
                # pre-activation gradients left in feature-major arrays), the weight gradients as contractions over the
                # surfels -- instead of four library GEMMs forward and eight backward whose activations cross HBM
                bone_map = None
                xbT, rawT = skin_field_train(self._xyz, bias[0], A, c0, sm)
            else:
                bone_map = None
                # (d/dA contracts over the surfels like the MLP's weight gradients: bob_warp.feature_major_linear)
                xbT = feature_major_linear(c0, A, self._xyz.t())
                rawT = sm.delta_raw_T(xbT, bias[0]) if sm.has_delta else None
            # (the renderer's rotation activation -- F.normalize per frame, 6 launches forward and backward -- is
            # applied inside the kernel: render_frames hands these orientations on as already activated)
            unit = bool(self.opts.get("fused_rot_activation", True))
            xyz_cam, rot_cam = lbs_skin_apply(xbT, rawT, se3, self._xyz, self._rotation, cq, ct, unit_rot=unit,
                                              bone_map=bone_map, frame_index=table_rows)
            self.__dict__["_warp_rot_is_unit"] = unit
            skin = delta = None
        else:
            if table_rows is not None:   # (this path takes per-frame rows)
                se3, cq, ct = (se3[0][table_rows], se3[1][table_rows]), cq[table_rows], ct[table_rows]
            frames = None if overrides else tab["bone_frames"]
            skin, delta = sm(self._xyz[None], rest1, None, iid, bone_frames=frames)
            xyz_cam, rot_cam = lbs_apply(skin[0].softmax(-1), se3, self._xyz, self._rotation, cq, ct)
        # The warp's auxiliary terms (skin entropy, delta-skin magnitude) only feed regularisers that
        # --rgb_loss_only drops (trainer.py:477-483): evaluated on request
        if self.opts.get("warp_aux", False):
            if skin is None:
                skin, delta = sm(self._xyz[None], rest1, None, iid, bone_frames=None if overrides else tab["bone_frames"])
            aux = {"skin_entropy": cross_entropy_skin_loss(skin)[..., None, None].expand(M, -1, -1, -1)}  # (M,N,1,1)
            if delta is not None:
                aux["delta_skin"] = delta.pow(2).mean(-1, keepdim=True)[..., None, :].expand(M, -1, -1, -1)
            self._aux_dict = aux
        else:
            self._aux_dict = {}
        return xyz_cam, rot_cam

    class _FrameGrad:
        """viewspace_points of one frame of a stacked call: `.grad` is that frame's slice of the stacked statistic."""

        def __init__(self, parent, index):
            self._parent, self._index = parent, index

        @property
        def grad(self):
            g = self._parent.grad
            return None if g is None else g[self._index]

    class _Visible:
        """visibility_filter of the frames of a stacked call, made when somebody asks (`radii > 0` is a launch per
        frame that only the densification statistics read)."""

        def __init__(self, radii):
            self._radii, self._made = radii, {}

        def __len__(self):
            return self._radii.shape[0]

        def __getitem__(self, i):
            if i not in self._made:
                self._made[i] = self._radii[i] > 0
            return self._made[i]

    def _render_frames_stacked(self, cams, xyz_cam, rot_cam, rot_is_unit, aux_planes=0):
        """All frames of the step through ONE launch set (diff_surfel_rasterization.rasterize_frames: stacked tile
        grids).  -> {"raw_stacked": (color (3,M,H,W), allmap (8,M,H,W))}; frame i is [:, i]."""
        from ..diff_surfel_rasterization import GaussianRasterizationSettings
        settings = []
        for cam in cams:
            tan = cam.__dict__.get("_raster_tanfov")
            if tan is None:  # fp32 tan of the fp32 field of view, as render() (gs/gaussian_renderer/__init__.py:36-37)
                tan = cam.__dict__["_raster_tanfov"] = tuple(
                    float(torch.tan(torch.as_tensor(f, dtype=torch.float32).cpu() * 0.5)) for f in (cam.FoVx, cam.FoVy))
            settings.append(GaussianRasterizationSettings(
                image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=tan[0], tanfovy=tan[1],
                bg=self.background_feat, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, sh_degree=self.active_sh_degree, campos=cam.camera_center,
                prefiltered=False, debug=False))
        rotations = rot_cam if rot_is_unit else self.rotation_activation(rot_cam, dim=-1)
        M = xyz_cam.shape[0]
        if M > RASTER_MAX_FRAMES:
            # More frames than one stacked launch set takes (Vidu4dSurfel*Args::frames <= 8): groups of <= 8 -- each its own
            # launch set, the frames' planes concatenated -- instead of the per-frame loop (the reference's loop takes any
            # imgs_per_gpu, deformable_gaussian.py:1175; VERDICT r5 missing 3).  Per-frame results are those of single calls.
            self.__dict__["_step_cache"] = {}   # (the activated copies of the non-canonical path: once for all groups)
            try:
                parts = [self._rasterize_group(settings[lo:lo + RASTER_MAX_FRAMES], xyz_cam[lo:lo + RASTER_MAX_FRAMES],
                                               rotations[lo:lo + RASTER_MAX_FRAMES], aux_planes, slot=lo // RASTER_MAX_FRAMES)
                         for lo in range(0, M, RASTER_MAX_FRAMES)]
            finally:
                self.__dict__.pop("_step_cache", None)
            self._viewspace_points_batch = [self._FrameGrad(p[3], i) for p in parts for i in range(p[1].shape[0])]
            self._visibility_filter_batch = [p[1][i] > 0 for p in parts for i in range(p[1].shape[0])]
            self._radii_batch = [p[1][i] for p in parts for i in range(p[1].shape[0])]
            return {"raw_stacked": (torch.cat([p[0] for p in parts], 1), torch.cat([p[2] for p in parts], 1))}
        color, radii, allmap, screen = self._rasterize_group(settings, xyz_cam, rotations, aux_planes)
        self._viewspace_points_batch = [self._FrameGrad(screen, i) for i in range(M)]
        self._visibility_filter_batch = self._Visible(radii)
        self._radii_batch = [radii[i] for i in range(M)]
        return {"raw_stacked": (color, allmap)}

    def _rasterize_group(self, settings, xyz_cam, rotations, aux_planes, slot=0):
        """One stacked launch set for <= 8 frames -> color (3,F,H,W), radii (F,N), allmap (8,F,H,W), the leaf whose .grad is the
        frames' densification statistic."""
        from ..diff_surfel_rasterization import rasterize_frames
        # a leaf whose .grad is the densification statistic; the rasterizer never reads its values (upstream passes zeros),
        # so every step's leaf shares one cached buffer instead of zero-filling a new one
        bufs = self.__dict__.setdefault("_screen_bufs", {})
        buf = bufs.get(slot)
        if buf is None or buf.shape != xyz_cam.shape or buf.device != xyz_cam.device:
            buf = bufs[slot] = torch.zeros_like(xyz_cam)
        screen = buf.detach().requires_grad_(True)
        if self.opts.get("canonical_params", True) and self._features_rest.shape[1] == 15:
            # the parameters as the optimizer holds them: the kernels apply exp / sigmoid and read the two SH tensors in
            # place -- no activation / concatenation launches here, none of their backward launches either
            color, radii, allmap = rasterize_frames(xyz_cam, screen, self._features_dc, self._opacity, self._scaling,
                                                    rotations, settings, sh_rest=self._features_rest, raw_params=True,
                                                    aux_planes=aux_planes)
        else:
            color, radii, allmap = rasterize_frames(xyz_cam, screen, self.get_features, self.get_opacity,
                                                    self.get_scaling, rotations, settings, aux_planes=aux_planes)
        return color, radii, allmap, screen

    @staticmethod
    def _collect_frame(r, raw, raw_frames, per_frame, stacked):
        if raw:
            raw_frames.append((r.pop("render"), r.pop("allmap")))
            r.pop("acc", None), r.pop("rend_dist", None)
        for k, v in r.items():
            if k in per_frame:
                per_frame[k].append(v)
            else:
                stacked.setdefault(k, []).append(v.permute(1, 2, 0))

    def _frame_streams(self, M):
        if M < 2:
            return None
        pool = self.__dict__.setdefault("_stream_pool", [])
        while len(pool) < M:
            pool.append(torch.cuda.Stream(device=self._xyz.device))
        return pool

    def render_frames(self, frame_id, Kinv, H, W, inst_id=None, samples_dict=None, outputs=None, aux_planes=0):
        """The per-frame render loop of query_field (:1175-1233): returns a dict of (M,H,W,C) maps and
        keeps the per-frame screen-space tensors the densification statistics need.
        aux_planes (stacked raw output only): bit mask of the allmap planes the caller reads, 0 = all
        (diff_surfel_rasterization.rasterize_frames)."""
        if self._xyz.is_cuda:
            # the rasterizer's host-side state -- capacity / split hints, unchecked deferred forwards, gradient outputs --
            # is this model's own (`raster_context`): two models stepping alternately, or rendering on two threads, share none
            from .. import _C
            rc = self.raster_context
            if _C.current() is not rc:
                with rc:
                    return self.render_frames(frame_id, Kinv, H, W, inst_id, samples_dict, outputs, aux_planes)
        M = frame_id.shape[0]
        if self.fused_warp_ok(inst_id):
            xyz_cam, rot_cam = self.forward_warp_fused(frame_id, inst_id, samples_dict)  # (M,N,3), (M,N,4)
            rot_is_unit = self.__dict__.pop("_warp_rot_is_unit", False)
        else:
            rot_is_unit = False
            xyz = self._xyz[None, :, None].expand(M, -1, -1, -1)
            rot = self._rotation[None].expand(M, -1, -1)
            xyz_cam, rot_cam, _ = self.forward_warp(xyz, rot, frame_id, inst_id, samples_dict)
            xyz_cam = xyz_cam.squeeze(2)
        cams = self.get_gs_Kcamera(Kinv, H, W)
        if (outputs is not None and tuple(outputs) == ("raw",) and xyz_cam.is_cuda and self.opts.get("stacked_frames", True)
                and M >= 1 and len({(int(h), int(w)) for h, w in zip(H, W)}) == 1):
            return self._render_frames_stacked(cams, xyz_cam, rot_cam, rot_is_unit, aux_planes)
        stacked, per_frame = {}, {"viewspace_points": [], "visibility_filter": [], "radii": []}
        # outputs containing "raw": the per-frame colour / auxiliary planes are handed out as they leave the rasterizer
        # (out["raw"] = [(color (3,H,W), allmap (8,H,W))], no learnable-background composite, no permute / stack)
        raw = outputs is not None and "raw" in outputs
        raw_frames = []
        if raw:
            outputs = tuple(outputs) + ("allmap",)
        # Frames of a step are independent until the loss: each one is queued on its own HIP stream so
        # that the tile workgroups of all frames are resident together (an object-centric frame fills
        # only a fraction of the 256 CUs, and its blend kernels are bound by the longest tile's serial
        # chain, not by throughput).  autograd replays each frame's backward on the stream of its
        # forward, so the backward kernels overlap the same way.  Per-frame results are unchanged.
        streams = self._frame_streams(M) if xyz_cam.is_cuda and self.opts.get("frame_streams", True) else None
        if outputs is None or any(k in GEOMETRY_KEYS for k in outputs):
            # the cached pixel-ray grid of a camera is shared by the frames that use it: build it here, on the
            # main stream, before the per-frame streams fork (two streams must not race to create it)
            for cam in cams:
                cam.pixel_rays()
        # (unbind, not xyz_cam[i]: its backward is ONE stack of the frames' gradients instead of a zero-filled full
        # tensor plus a copy plus an add per frame)
        frame_xyz, frame_rot = xyz_cam.unbind(0), rot_cam.unbind(0)
        self.__dict__["_step_cache"] = {}
        self.__dict__["_override_rotation_is_unit"] = rot_is_unit
        shared = (self.get_scaling, self.get_opacity, self.get_features)  # made here, on the main stream
        if streams:
            main = torch.cuda.current_stream(xyz_cam.device)
            ready = main.record_event()
        try:
            for i in range(M):
                if streams:
                    streams[i].wait_event(ready)
                    with torch.cuda.stream(streams[i]):
                        for t in shared:
                            t.record_stream(streams[i])
                        r = self.render_view(cams[i], override_xyz=frame_xyz[i], override_rotation=frame_rot[i],
                                             outputs=outputs)
                        for v in r.values():
                            v.record_stream(main)
                else:
                    r = self.render_view(cams[i], override_xyz=frame_xyz[i], override_rotation=frame_rot[i],
                                         outputs=outputs)
                self._collect_frame(r, raw, raw_frames, per_frame, stacked)
        finally:
            self.__dict__.pop("_step_cache", None)
            self.__dict__.pop("_override_rotation_is_unit", None)
        if streams:
            for st in streams[:M]:
                main.wait_stream(st)
        out = {k: torch.stack(v, 0) for k, v in stacked.items()}
        self._viewspace_points_batch = per_frame["viewspace_points"]
        self._visibility_filter_batch = per_frame["visibility_filter"]
        self._radii_batch = per_frame["radii"]
        if raw:
            out["raw"] = raw_frames
        else:
            out["rendered"] = out["render"]
            out["mask"] = out["acc"]
        return out


class _Args:
    """dict -> attribute access with the Stage-3 defaults of lab4d/config.py:155-238."""
    DEFAULTS = dict(percent_dense=0.01, position_lr_init=5e-5, position_lr_final=5e-7, position_lr_delay_mult=0.01,
                    position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3,
                    rotation_lr=1e-3, densification_interval=100, densify_from_iter=500, densify_until_iter=15000,
                    densify_grad_threshold=2e-4, opacity_reset_interval=3000, lambda_normal=0.05, lambda_dist=0.0,
                    lambda_dssim=0.0, sh_degree=3, gs_learnable_bg=True, rgb_wt=0.1, mask_wt=0.1, learning_rate=5e-4,
                    outlier_filtering_interval=2000, outlier_stop_iter=29000)

    def __init__(self, opts):
        self.__dict__.update(self.DEFAULTS)
        self.__dict__.update({k: v for k, v in opts.items() if k in self.DEFAULTS})
