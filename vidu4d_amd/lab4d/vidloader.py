"""Reader for the reference's preprocessed video data (`database/processed/...`), every modality.

Reference: lab4d/dataloader/vidloader.py:48-372 (RangeSampler, VidDataset), lab4d/dataloader/data_utils.py:13-31
(FrameInfo), :121-215 (config_to_dataset / section_to_dataset / load_config), :225-334 (get_data_info,
load_small_files), lab4d/utils/numpy_utils.py:97-122 (bilinear_interp), lab4d/utils/geom_utils.py:310-390
(K2mat / K2inv), lab4d/engine/model.py:401-427 (Kinv = K2inv(K) @ K2mat(crop2raw)).

On-disk layout, per video `<vid>` (= `<seqname>-0000`, ...) and crop prefix `<prefix>` ("crop-256", "full-256", ...):

    database/configs/<seqname>.config                      [data] + [data_<i>]: img_path, ks, shape, init_frame, end_frame
    processed/JPEGImages/Full-Resolution/<vid>/%05d.jpg    the kept frames (their NAMES are the raw frame ids)
    processed/JPEGImagesRaw/Full-Resolution/<vid>/*.jpg    all frames (only counted)
    processed/JPEGImages/Full-Resolution/<vid>/<prefix>.npy               (F,H,W,3) float16 rgb in [0,1]
    processed/Annotations/Full-Resolution/<vid>/<prefix>.npy              (F,H,W,2) mask, vis2d
    processed/Annotations/Full-Resolution/<vid>/<prefix>-crop2raw.npy     (F,4) fx, fy, cx, cy crop -> raw
    processed/Annotations/Full-Resolution/<vid>/<prefix>-is_detected.npy  (F,)
    processed/Depth/Full-Resolution/<vid>/<prefix>.npy                    (F,H,W) float16
    processed/FlowFW_<d>/Full-Resolution/<vid>/<prefix>.npy               (F/d,H,W,3) flow (2) + uncertainty (1)
    processed/FlowBW_<d>/Full-Resolution/<vid>/<prefix>.npy               same, backward
    processed/Features/Full-Resolution/<vid>/<prefix>-<feature_type>-01.npy  (F,112,112,16)
    processed/Cameras/Full-Resolution/<vid>/00.npy, 01-canonical.npy      (F,4,4) background / object world-to-camera

Arrays are memory-mapped like upstream (vidloader.py:146-166).  `VidDataset` keeps the reference's constructor, method
names, returned keys, dtypes and sampling behaviour (tests/test_refpy_vidloader.py compares it with the imported
reference class on the same directory); `stage3_batch` assembles what `Stage3Trainer.train_step` consumes."""
from __future__ import annotations

import configparser
import glob
import os
from pathlib import Path

import numpy as np
import torch


def K2mat(K):
    """(..., 4) intrinsics (fx, fy, cx, cy) -> (..., 3, 3) matrix (geom_utils.py:310-329)."""
    K = torch.as_tensor(K, dtype=torch.float32)
    m = torch.zeros(K.shape[:-1] + (3, 3), dtype=torch.float32, device=K.device)
    m[..., 0, 0], m[..., 1, 1] = K[..., 0], K[..., 1]
    m[..., 0, 2], m[..., 1, 2] = K[..., 2], K[..., 3]
    m[..., 2, 2] = 1.0
    return m


def K2inv(K):
    """(..., 4) intrinsics -> (..., 3, 3) inverse matrix (geom_utils.py:372-390)."""
    K = torch.as_tensor(K, dtype=torch.float32)
    m = torch.zeros(K.shape[:-1] + (3, 3), dtype=torch.float32, device=K.device)
    m[..., 0, 0], m[..., 1, 1] = 1.0 / K[..., 0], 1.0 / K[..., 1]
    m[..., 0, 2], m[..., 1, 2] = -K[..., 2] / K[..., 0], -K[..., 3] / K[..., 1]
    m[..., 2, 2] = 1.0
    return m


FEATURE_RES = 112  # side of the stored feature maps (vidloader.py:340-348)


def bilinear_interp(feat, xy_loc):
    """feat (H,W,C), xy_loc (N,2) float (x, y) -> (N,C) in feat's dtype.  The corner index is clipped to
    [0, 110] AFTER the fractional weights are taken, as upstream does (numpy_utils.py:106-121)."""
    corner = np.floor(xy_loc).astype(int)
    fx = (xy_loc[:, 0] - corner[:, 0])[:, None]
    fy = (xy_loc[:, 1] - corner[:, 1])[:, None]
    corner = np.clip(corner, 0, FEATURE_RES - 2)
    cx, cy = corner[:, 0], corner[:, 1]
    out = (feat[cy, cx] * (1 - fx) * (1 - fy) + feat[cy + 1, cx] * (1 - fx) * (fy - 0) +
           feat[cy, cx + 1] * (fx - 0) * (1 - fy) + feat[cy + 1, cx + 1] * (fx - 0) * (fy - 0))
    return out.astype(feat.dtype)


class RangeSampler:
    """Indices from [0, num_elems) without replacement; a fresh permutation whenever the current one cannot serve a
    whole request (vidloader.py:15-45)."""

    def __init__(self, num_elems):
        self.num_elems = num_elems
        self.init_queue()

    def init_queue(self):
        self.sample_queue = np.random.permutation(self.num_elems)
        self.curr_idx = 0

    def sample(self, num_samples):
        if self.curr_idx + num_samples > self.num_elems:
            self.init_queue()
        out = self.sample_queue[self.curr_idx:self.curr_idx + num_samples]
        self.curr_idx += num_samples
        return out


class FrameInfo:
    """num_frames (kept), num_frames_raw (all), frame_map: kept frame -> raw frame id, read from the file NAMES
    (data_utils.py:13-31)."""

    def __init__(self, ref_list):
        self.num_frames = len(ref_list)
        raw_dir = ref_list[0].rsplit("/", 1)[0].replace("JPEGImages", "JPEGImagesRaw")
        self.num_frames_raw = len(glob.glob(raw_dir + "/*.jpg"))
        if self.num_frames_raw <= 0:
            raise FileNotFoundError(f"no raw frames under {raw_dir}")
        self.frame_map = [int(p.split("/")[-1].split(".")[0]) for p in ref_list]


_SMALL = ("ref", "cambg", "camfg", "crop2raw", "is_detected")


class VidDataset(torch.utils.data.Dataset):
    """One video of a sequence (vidloader.py:48-372): item i is the frame pair (i, i + delta)."""

    def __init__(self, opts, rgblist, dataid, ks, raw_size):
        self.delta_list = opts["delta_list"]
        self.dict_list = self.construct_data_list(rgblist, opts["data_prefix"], opts["feature_type"])
        self.pixels_per_image = opts["pixels_per_image"]
        self.dataid = dataid
        self.load_pair = opts["load_pair"]
        self.ks = ks
        self.raw_size = raw_size
        self.img_size = np.load(self.dict_list["rgb"], mmap_mode="r").shape[1:3]  # (H, W)
        self.load_data_list(self.dict_list)
        self.idx_sampler = RangeSampler(num_elems=self.img_size[0] * self.img_size[1])
        self.frame_info = FrameInfo(self.dict_list["ref"])
        self.quant_exp = bool(opts.get("quant_exp", False))

    def construct_data_list(self, reflist, prefix, feature_type):
        first = reflist[0]
        rgb = first.replace("00000.jpg", f"{prefix}.npy")
        swap = lambda name: rgb.replace("JPEGImages", name)  # noqa: E731
        mask = swap("Annotations")
        return {
            "ref": reflist, "rgb": rgb, "mask": mask,
            "cambg": first.replace("JPEGImages", "Cameras").replace("00000.jpg", "00.npy"),
            "camfg": first.replace("JPEGImages", "Cameras").replace("00000.jpg", "01-canonical.npy"),
            "flowfw": swap("FlowFW"), "flowbw": swap("FlowBW"), "depth": swap("Depth"),
            "feature": str(Path(swap("Features")).parent) + f"/{prefix}-{feature_type}-01.npy",
            "crop2raw": mask.replace(".npy", "-crop2raw.npy"),
            "is_detected": mask.replace(".npy", "-is_detected.npy"),
        }

    def load_data_list(self, dict_list):
        self.crop2raw = np.load(dict_list["crop2raw"])
        self.is_detected = np.load(dict_list["is_detected"])
        self.mmap_list = {}
        for key, path in dict_list.items():
            if key in _SMALL:
                continue
            if key in ("flowfw", "flowbw"):
                per_delta = {}
                for delta in [1] + list(self.delta_list):
                    p = path.replace("FlowFW", f"FlowFW_{delta}").replace("FlowBW", f"FlowBW_{delta}")
                    if os.path.exists(p):
                        per_delta[delta] = np.load(p, mmap_mode="r")
                self.mmap_list[key] = per_delta
                continue
            try:
                self.mmap_list[key] = np.load(path, mmap_mode="r")
            except (OSError, ValueError):
                # upstream substitutes noise of the feature shape for ANY unreadable array (vidloader.py:161-165)
                print(f"Warning: cannot load {path}")
                self.mmap_list[key] = np.random.rand(len(self) + 1, FEATURE_RES, FEATURE_RES, 16)

    def __len__(self):
        n = len(self.dict_list["ref"])
        return n // 4 - 2 if getattr(self, "quant_exp", False) else n - 1

    def __getitem__(self, index):
        return self.load_data(index * 4 if self.quant_exp else index)

    def sample_delta(self, index):
        n = len(self.dict_list["ref"])
        choices = [1] + [d for d in self.delta_list if index % d == 0 and int(index + d) < n]
        if self.quant_exp:
            choices = [4, 8]
        return np.random.choice(choices)

    def sample_xy(self):
        if self.pixels_per_image == -1:
            return None
        idx = self.idx_sampler.sample(num_samples=self.pixels_per_image)
        # (upstream divides by the HEIGHT for both: exact for square crops)
        return np.stack([idx // self.img_size[0], idx % self.img_size[0]], axis=-1)

    def load_data(self, im0idx):
        delta = self.sample_delta(im0idx)
        xy0, xy1 = self.sample_xy(), self.sample_xy()
        out = self.read_raw(im0idx, delta, rand_xy=xy0)
        if self.load_pair:
            other = self.read_raw(im0idx + delta, -delta, rand_xy=xy1)
            for k in out:
                out[k] = np.stack([out[k], other[k]])
        return out

    def read_raw(self, im0idx, delta, rand_xy=None):
        rgb = self.read_rgb(im0idx, rand_xy=rand_xy)
        mask, vis2d, crop2raw, is_detected = self.read_mask(im0idx, rand_xy=rand_xy)
        depth = self.read_depth(im0idx, rand_xy=rand_xy)
        flow = self.read_flow(im0idx, delta, rand_xy=rand_xy)
        feature = self.read_feature(im0idx, rand_xy=rand_xy)
        if rand_xy is None:
            x0, y0 = np.meshgrid(range(self.img_size[1]), range(self.img_size[0]))
            hxy = np.stack([x0, y0, np.ones_like(x0)], axis=-1)
        else:
            hxy = np.concatenate([rand_xy, np.ones_like(rand_xy[..., :1])], -1)
        return {"rgb": rgb, "mask": mask, "depth": depth, "feature": feature, "flow": flow[..., :2],
                "flow_uct": flow[..., 2:], "vis2d": vis2d, "crop2raw": crop2raw, "is_detected": is_detected,
                "dataid": self.dataid, "frameid_sub": self.frame_info.frame_map[im0idx],
                "hxy": hxy.astype(np.float32)}

    @staticmethod
    def _pick(frame, rand_xy):
        return frame if rand_xy is None else frame[rand_xy[:, 1], rand_xy[:, 0]]

    def read_rgb(self, im0idx, rand_xy=None):
        frame = self.mmap_list["rgb"][im0idx]
        gray = frame.ndim == 2
        frame = self._pick(frame, rand_xy)
        return np.repeat(np.expand_dims(frame, -1), 3, axis=-1) if gray else frame

    def read_mask(self, im0idx, rand_xy=None):
        m = self._pick(self.mmap_list["mask"][im0idx], rand_xy)
        return m[..., :1], m[..., 1:], self.crop2raw[im0idx], self.is_detected[im0idx]

    def read_depth(self, im0idx, rand_xy=None):
        return self._pick(self.mmap_list["depth"][im0idx], rand_xy)[..., None]

    def read_feature(self, im0idx, rand_xy=None):
        feat = self.mmap_list["feature"][im0idx]
        if rand_xy is not None:
            feat = bilinear_interp(feat, rand_xy / self.img_size[0] * FEATURE_RES)
        return feat.astype(np.float32)

    def read_flow(self, im0idx, delta, rand_xy=None):
        step = abs(delta)
        if delta > 0:
            frame = self.mmap_list["flowfw"][step][im0idx // step]
        else:
            frame = self.mmap_list["flowbw"][step][im0idx // step - 1]
        return self._pick(frame, rand_xy).astype(np.float32)


def load_config(config, section, current_dict=None):
    """One section of database/configs/<seqname>.config; missing or malformed entries keep what `current_dict`
    (the [data] defaults) holds (data_utils.py:176-215)."""
    out = {} if current_dict is None else current_dict
    readers = {"rgb_path": ("img_path", str), "init_frame": ("init_frame", int), "end_frame": ("end_frame", int),
               "ks": ("ks", lambda s: [float(v) for v in s.split(" ")]),
               "raw_size": ("shape", lambda s: [int(v) for v in s.split(" ")])}
    for key, (name, conv) in readers.items():
        try:
            out[key] = conv(config.get(section, name))
        except (configparser.Error, ValueError):
            pass
    return out


def section_to_dataset(opts, config, vidid, constructor=VidDataset):
    cfg = load_config(config, "data")
    cfg = load_config(config, "data_%d" % vidid, current_dict=cfg)
    rgblist = sorted(glob.glob("%s/*.jpg" % cfg["rgb_path"]))
    if cfg["end_frame"] > -1:
        rgblist = rgblist[:cfg["end_frame"]]
    if cfg["init_frame"] > 0:
        rgblist = rgblist[cfg["init_frame"]:]
    return constructor(opts, rgblist=rgblist, dataid=vidid, ks=cfg["ks"], raw_size=cfg["raw_size"])


def config_to_datasets(opts, config_path=None):
    """-> [VidDataset] for every [data_<i>] section of database/configs/<seqname>.config (data_utils.py:121-148;
    the iteration-count padding of `duplicate_dataset` belongs to the torch DataLoader upstream wraps around it
    and is not reproduced: the Stage-3 loop below indexes frames itself)."""
    config = configparser.RawConfigParser()
    path = config_path or "database/configs/%s.config" % opts["seqname"]
    if not config.read(path):
        raise FileNotFoundError(path)
    return [section_to_dataset(opts, config, v) for v in range(len(config.sections()) - 1)]


def get_data_info(datasets):
    """Dataset metadata (data_utils.py:225-334): frame_info {frame_offset, frame_offset_raw (cumulative),
    frame_mapping}, total_frames, intrinsics (N,4), raw_size (V,2), rtmat (2,N,4,4) = background / object
    cameras, vis_info, geom_path.  (The feature PCA `apply_pca_fn`, used by upstream's visualisations only, is
    not built.)"""
    offs, offs_raw, mapping, intrinsics, raw_size = [0], [0], [], [], []
    for ds in datasets:
        fi = FrameInfo(ds.dict_list["ref"])
        offs.append(fi.num_frames)
        offs_raw.append(fi.num_frames_raw)
        mapping += [i + np.sum(offs_raw[:-1]) for i in fi.frame_map]
        intrinsics += [ds.ks] * fi.num_frames
        raw_size.append(ds.raw_size)
    frame_info = {"frame_offset": np.asarray(offs).cumsum(), "frame_offset_raw": np.asarray(offs_raw).cumsum(),
                  "frame_mapping": mapping}
    info = {"frame_info": frame_info, "total_frames": frame_info["frame_offset"][-1],
            "intrinsics": np.asarray(intrinsics), "raw_size": np.asarray(raw_size)}
    cams = {k: np.concatenate([np.load(ds.dict_list[k]).astype(np.float32) for ds in datasets], 0)
            for k in ("cambg", "camfg")}
    info["vis_info"] = {"bg": 0, "fg": 1}
    info["rtmat"] = np.stack([cams["cambg"], cams["camfg"]], 0)
    cam_dir = datasets[0].dict_list["cambg"].rsplit("/", 1)[0]
    info["geom_path"] = [f"{cam_dir}/mesh-00-centered.obj", f"{cam_dir}/mesh-01-centered.obj"]
    return info


def stage3_batch(datasets, data_info, frames, device="cpu") -> dict:
    """frames: [(video id, kept-frame index)] -> the batch `Stage3Trainer.train_step` consumes:
    {"frameid" (M,) raw frame id in the whole sequence (model.py:385-399: frameid_sub + frame_offset_raw[dataid]),
     "Kinv" (M,3,3) host tensor, "H", "W", "rgb" (M,H,W,3), "mask", "vis2d" (M,H,W,1), "is_detected" (M,),
     "depth" (M,H,W,1), "dataid" (M,)}."""
    offs_raw = data_info["frame_info"]["frame_offset_raw"]
    rows = {k: [] for k in ("rgb", "mask", "vis2d", "depth", "crop2raw", "is_detected", "frameid", "ks", "dataid")}
    for vid, idx in frames:
        ds = datasets[vid]
        rows["rgb"].append(np.asarray(ds.read_rgb(idx), dtype=np.float32))
        mask, vis2d, crop2raw, det = ds.read_mask(idx)
        rows["mask"].append(np.asarray(mask, dtype=np.float32))
        rows["vis2d"].append(np.asarray(vis2d, dtype=np.float32))
        rows["depth"].append(np.asarray(ds.read_depth(idx), dtype=np.float32) if "depth" in ds.mmap_list and
                             ds.mmap_list["depth"].ndim == 3 else np.zeros(ds.img_size + (1,), np.float32))
        rows["crop2raw"].append(crop2raw)
        rows["is_detected"].append(bool(det))
        rows["frameid"].append(int(ds.frame_info.frame_map[idx] + offs_raw[vid]))
        rows["ks"].append(ds.ks)
        rows["dataid"].append(vid)
    dev = torch.device(device)
    M = len(frames)
    H, W = datasets[frames[0][0]].img_size
    Kinv = K2inv(torch.tensor(rows["ks"], dtype=torch.float32)) @ K2mat(np.stack(rows["crop2raw"]))
    img = lambda k: torch.from_numpy(np.stack(rows[k])).to(dev)  # noqa: E731
    fid = torch.tensor(rows["frameid"], device=dev)
    fid._vidu4d_host_range = (min(rows["frameid"]), max(rows["frameid"]))   # (DeformableSurfels._check_frame_ids: no device read)
    return {"frameid": fid, "Kinv": Kinv.cpu(), "H": [int(H)] * M, "W": [int(W)] * M,
            "rgb": img("rgb"), "mask": img("mask"), "vis2d": img("vis2d"), "depth": img("depth"),
            "is_detected": torch.tensor(rows["is_detected"], device=dev), "dataid": torch.tensor(rows["dataid"], device=dev)}


class SequenceData:
    """One video read without a config file (`--intrinsics` given on the command line): rgb + annotations only."""

    def __init__(self, root: str, seq: str, prefix: str = "full-256"):
        self.seq, self.prefix = seq, prefix
        base = os.path.join(root, "%s", "Full-Resolution", seq)
        rgb = os.path.join(base % "JPEGImages", prefix + ".npy")
        ann = os.path.join(base % "Annotations", prefix + ".npy")
        for p in (rgb, ann):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        self.rgb = np.load(rgb, mmap_mode="r")
        self.annot = np.load(ann, mmap_mode="r")
        self.crop2raw = np.load(ann.replace(".npy", "-crop2raw.npy"))
        det = ann.replace(".npy", "-is_detected.npy")
        self.is_detected = np.load(det) if os.path.exists(det) else np.ones(len(self.rgb), dtype=bool)
        if self.rgb.shape[:3] != self.annot.shape[:3] or self.annot.shape[-1] < 2:
            raise ValueError(f"{seq}: rgb {self.rgb.shape} and annotation {self.annot.shape} arrays do not match")
        self.img_size = tuple(self.rgb.shape[1:3])  # (H, W)

    def __len__(self):
        return self.rgb.shape[0]

    def frame_batch(self, frame_ids, intrinsics, device="cpu", frame_offset: int = 0) -> dict:
        """frame_ids: indices into this sequence; intrinsics: (4,) or (M,4) raw-image (fx, fy, cx, cy).
        -> {"frameid" (M,), "Kinv" (M,3,3), "H", "W", "rgb" (M,H,W,3), "mask" (M,H,W,1), "vis2d" (M,H,W,1),
            "is_detected" (M,)}; frameid = frame_offset + index (the sequence's position in the dataset)."""
        idx = np.asarray(frame_ids, dtype=np.int64)
        M = len(idx)
        H, W = self.img_size
        rgb = np.stack([np.asarray(self.rgb[i]) for i in idx]).astype(np.float32)
        if rgb.ndim == 3:  # gray frames (vidloader.py:291-292)
            rgb = np.repeat(rgb[..., None], 3, axis=-1)
        ann = np.stack([np.asarray(self.annot[i]) for i in idx]).astype(np.float32)
        K = torch.as_tensor(np.asarray(intrinsics), dtype=torch.float32)
        K = K.expand(M, 4) if K.dim() == 1 else K
        Kinv = K2inv(K) @ K2mat(self.crop2raw[idx])
        dev = torch.device(device)
        fid = torch.as_tensor(idx + frame_offset, device=dev)
        if M:
            fid._vidu4d_host_range = (int(idx.min()) + int(frame_offset), int(idx.max()) + int(frame_offset))
        return {"frameid": fid, "Kinv": Kinv.cpu(), "H": [H] * M,
                "W": [W] * M, "rgb": torch.from_numpy(rgb).to(dev), "mask": torch.from_numpy(ann[..., :1].copy()).to(dev),
                "vis2d": torch.from_numpy(ann[..., 1:2].copy()).to(dev),
                "is_detected": torch.as_tensor(self.is_detected[idx].astype(bool), device=dev)}
