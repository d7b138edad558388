"""Reader for the reference's preprocessed video data, reduced to what Stage-3 (`--rgb_loss_only`) reads.

Reference: lab4d/dataloader/vidloader.py:60-372 (VidDataset), lab4d/utils/geom_utils.py:310-390
(K2mat / K2inv), lab4d/engine/model.py:401-427 (Kinv = K2inv(K) @ K2mat(crop2raw)).  On-disk layout,
per sequence `<seq>` and crop prefix `<prefix>` ("crop-256", "full-256", ...), all `.npy`:

    database/processed/JPEGImages/Full-Resolution/<seq>/<prefix>.npy               (F,H,W,3) float16 rgb in [0,1]
    database/processed/Annotations/Full-Resolution/<seq>/<prefix>.npy              (F,H,W,2) mask, vis2d
    database/processed/Annotations/Full-Resolution/<seq>/<prefix>-crop2raw.npy     (F,4) fx, fy, cx, cy crop -> raw
    database/processed/Annotations/Full-Resolution/<seq>/<prefix>-is_detected.npy  (F,)
    (+ Depth, FlowFW_<d>, FlowBW_<d>, Features, Cameras: not read here -- their losses are dropped by
     --rgb_loss_only, trainer.py:477-483)

Arrays are memory-mapped like upstream (vidloader.py:146-166); a frame batch is assembled in the layout
`Stage3Trainer.train_step` expects."""
from __future__ import annotations

import os

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


class SequenceData:
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
        return {"frameid": torch.as_tensor(idx + frame_offset, device=dev), "Kinv": Kinv.cpu(), "H": [H] * M,
                "W": [W] * M, "rgb": torch.from_numpy(rgb).to(dev), "mask": torch.from_numpy(ann[..., :1].copy()).to(dev),
                "vis2d": torch.from_numpy(ann[..., 1:2].copy()).to(dev),
                "is_detected": torch.as_tensor(self.is_detected[idx].astype(bool), device=dev)}
