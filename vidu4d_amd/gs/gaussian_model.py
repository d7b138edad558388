"""GaussianModel: the canonical surfel parameter set and its densify / prune / optimizer surgery
(reference: gs/scene/gaussian_model.py; create_from_pcd :127-151, activations :28-43,
replace/prune/cat optimizer surgery :270-356, densify_and_split :384-412, densify_and_clone :414-432,
densify_and_prune :434-448, add_densification_stats :450-452, PLY I/O :189-268).

Same attribute and method names and the same semantics, with two practical differences: tensors
live on the device given at construction instead of a hard-coded "cuda", and the one-off scale
initialisation (mean squared distance to the 3 nearest neighbours, simple-knn's distCUDA2) runs the
HIP kernel csrc/knn.hip on a GPU and a scipy cKDTree query for host-only construction."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .sh_utils import RGB2SH


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def build_rotation(r):
    """(N,4) w-first quaternions -> (N,3,3) (reference: gs/utils/general_utils.py build_rotation)."""
    q = r / torch.norm(r, dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def mean_knn_dist2(points: np.ndarray, k: int = 3) -> np.ndarray:
    """Mean squared distance to the k nearest neighbours (what distCUDA2 returns, simple_knn.cu:185-221)."""
    from scipy.spatial import cKDTree
    tree = cKDTree(points)
    d, _ = tree.query(points, k=k + 1)
    return (d[:, 1:] ** 2).mean(axis=1)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return helper


class GaussianModel(nn.Module):
    def setup_functions(self):
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree: int, device="cuda"):
        super().__init__()
        self.device_ = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.setup_functions()

    # ---- activated views
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- construction
    def create_from_pcd(self, pcd, spatial_lr_scale: float, dist2_fn=None):
        """pcd: object with .points (N,3) and .colors (N,3) in [0,1] (BasicPointCloud upstream)."""
        dev = self.device_
        self.spatial_lr_scale = spatial_lr_scale
        pts = np.asarray(pcd.points, dtype=np.float32)
        xyz = torch.from_numpy(pts).to(dev)
        color = RGB2SH(torch.from_numpy(np.asarray(pcd.colors, dtype=np.float32)).to(dev))
        n, m = xyz.shape[0], (self.max_sh_degree + 1) ** 2
        features = torch.zeros(n, 3, m, device=dev)
        features[:, :3, 0] = color
        if dist2_fn is not None:
            d2 = dist2_fn(pts)
        elif xyz.is_cuda:  # the HIP 3-NN kernel (csrc/knn.hip), as upstream's distCUDA2
            from ..simple_knn import distCUDA2
            d2 = distCUDA2(xyz)
        else:              # host-only construction (CPU tests): same quantity from a k-d tree
            d2 = mean_knn_dist2(pts)
        dist2 = torch.clamp_min(torch.as_tensor(d2, dtype=torch.float32, device=dev), 1e-7)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 2)
        rots = torch.rand(n, 4, device=dev)  # upstream initialises with uniform random quaternions (:141)
        opacities = self.inverse_opacity_activation(0.1 * torch.ones(n, 1, device=dev))
        self._xyz = nn.Parameter(xyz.requires_grad_(True))
        self._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self.max_radii2D = torch.zeros(n, device=dev)

    def training_setup(self, training_args):
        n = self.get_xyz.shape[0]
        dev = self.get_xyz.device
        self.percent_dense = training_args.percent_dense
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(new, "opacity")["opacity"]

    # ---- optimizer surgery (one parameter per named group, as upstream)
    _GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "regist_feat")

    def _assign(self, tensors):
        self._xyz = tensors["xyz"]
        self._features_dc = tensors["f_dc"]
        self._features_rest = tensors["f_rest"]
        self._opacity = tensors["opacity"]
        self._scaling = tensors["scaling"]
        self._rotation = tensors["rotation"]
        if "regist_feat" in tensors and hasattr(self, "_regist_feat"):
            self._regist_feat = tensors["regist_feat"]

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] != name:
                continue
            # Upstream (:270-290) zeroes the moments but files them back under the OLD parameter object before
            # swapping the parameter in, so the replacement starts WITHOUT Adam state: its next update runs
            # with a fresh step count (bias correction restarts).  Reproduced: the old state is dropped.
            self.optimizer.state.pop(group["params"][0], None)
            new = nn.Parameter(tensor.requires_grad_(True))
            group["params"][0] = new
            out[name] = new
            return out
        raise ValueError("Tensor not found in optimizer")

    def _resize_groups(self, fn_param, fn_state):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] == "bg_rgb":
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            new = nn.Parameter(fn_param(group["name"], old).requires_grad_(True))
            if state is not None:
                if "exp_avg" in state:
                    state["exp_avg"] = fn_state(group["name"], state["exp_avg"])
                    state["exp_avg_sq"] = fn_state(group["name"], state["exp_avg_sq"])
                self.optimizer.state[new] = state
            group["params"][0] = new
            out[group["name"]] = new
        return out

    def _prune_optimizer(self, mask):
        return self._resize_groups(lambda n, p: p[mask], lambda n, s: s[mask])

    def prune_points(self, mask):
        keep = ~mask
        self._assign(self._prune_optimizer(keep))
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def cat_tensors_to_optimizer(self, tensors_dict):
        return self._resize_groups(lambda n, p: torch.cat((p, tensors_dict[n]), dim=0),
                                   lambda n, s: torch.cat((s, torch.zeros_like(tensors_dict[n])), dim=0))

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation, new_regist_feat=None):
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "scaling": new_scaling, "rotation": new_rotation}
        if new_regist_feat is not None:
            d["regist_feat"] = new_regist_feat
        self._assign(self.cat_tensors_to_optimizer(d))
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    def _regist(self, mask, reps=None):
        if not hasattr(self, "_regist_feat"):
            return None
        r = self._regist_feat[mask]
        return r if reps is None else r.repeat(reps, 1)

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None, samples=None):
        """`samples` (extension, tests): the (N * selected, 3) normal draws to use instead of sampling."""
        n0, dev = self.get_xyz.shape[0], self.get_xyz.device
        padded = torch.zeros(n0, device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        stds = torch.cat([stds, torch.zeros_like(stds[:, :1])], dim=-1)  # surfels: no extent along the normal
        if samples is None:
            samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._features_dc[sel].repeat(N, 1, 1),
                                   self._features_rest[sel].repeat(N, 1, 1), self._opacity[sel].repeat(N, 1),
                                   new_scaling, self._rotation[sel].repeat(N, 1), self._regist(sel, N))
        prune = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool)))
        self.prune_points(prune)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & \
              (torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel],
                                   self._opacity[sel], self._scaling[sel], self._rotation[sel], self._regist(sel))

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None, samples=None):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator, samples=samples)
        prune_mask = (self.get_opacity < min_opacity).squeeze(-1)
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune_mask = prune_mask | big_vs | big_ws
        self.prune_points(prune_mask)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        g = viewspace_point_tensor.grad if viewspace_point_tensor.grad is not None else viewspace_point_tensor
        f = update_filter[:, None].to(self.denom.dtype)  # masked adds without nonzero() (no host sync)
        self.xyz_gradient_accum += torch.norm(g, dim=-1, keepdim=True) * f
        self.denom += f

    # ---- PLY I/O: binary little-endian, attribute order x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_*
    # with channel-major f_dc / f_rest (reference :189-268)
    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        c = lambda t: t.detach().cpu().numpy()  # noqa: E731
        xyz = c(self._xyz)
        cols = [xyz, np.zeros_like(xyz), c(self._features_dc.transpose(1, 2).flatten(start_dim=1)),
                c(self._features_rest.transpose(1, 2).flatten(start_dim=1)), c(self._opacity), c(self._scaling),
                c(self._rotation)]
        data = np.ascontiguousarray(np.concatenate(cols, axis=1).astype("<f4"))
        names = self.construct_list_of_attributes()
        assert data.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % data.shape[0]
        header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(data.tobytes())

    def load_ply(self, path):
        with open(path, "rb") as f:
            names, n = [], 0
            while True:
                line = f.readline().decode("ascii").strip()
                if line.startswith("element vertex"):
                    n = int(line.split()[-1])
                elif line.startswith("property"):
                    names.append(line.split()[-1])
                elif line == "end_header":
                    break
            data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
        col = {k: i for i, k in enumerate(names)}
        dev = self.device_
        take = lambda prefix: data[:, [col[k] for k in sorted((k for k in names if k.startswith(prefix)),  # noqa: E731
                                                                 key=lambda s: int(s.split("_")[-1]))]]
        xyz = data[:, [col["x"], col["y"], col["z"]]]
        m = (self.max_sh_degree + 1) ** 2
        f_dc = take("f_dc_").reshape(n, 3, 1)
        f_rest = take("f_rest_").reshape(n, 3, m - 1)
        t = lambda a: nn.Parameter(torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev).requires_grad_(True))  # noqa: E731
        self._xyz = t(xyz)
        self._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float32, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(f_rest, dtype=torch.float32, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = t(data[:, [col["opacity"]]])
        self._scaling = t(take("scale_"))
        self._rotation = t(take("rot_"))
        self.active_sh_degree = self.max_sh_degree
