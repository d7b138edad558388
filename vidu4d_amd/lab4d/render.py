"""Forward-only rendering of a Stage-3 checkpoint -- the consumer side of `lab4d/render.py` for surfel fields
(reference: lab4d/render.py:279-354: construct_test_model -> batch of all frames with H = W = render_res ->
model.evaluate under no_grad -> rendered['rgb'] clamped to [0, 1] -> `<save_dir>/rgb.pth` = {"rgb": float16
(F,H,W,3), "mask": float16 |surf_normal - rend_normal|}; save_dir = <logdir>/renderings_%04d/<viewpoint>).

    python lab4d/render.py --flagfile=logdir/<seq>-<logname>/opts.log --load_suffix latest --render_res 512

What runs here: the checkpoint (`ckpt_<suffix>.pth`, reference key layout, vidu4d_amd/lab4d/checkpoint.py) is loaded into
a DeformableSurfels model built with the checkpoint's frame count, every frame is warped and rasterized on the MI355X
path (DeformableSurfels.render_frames, frames in chunks through the stacked launch set), and the reference's outputs
are written.  Viewpoint synthesis (`--viewpoint rot-*`), the PCA feature images and the mp4 writers of
lab4d/utils/io.py are outside this build; `--viewpoint ref` is what is rendered."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from .train import STAGE3_FLAGS, parse_flags

RENDER_FLAGS = dict(load_suffix="latest", render_res=128, motion_id=0, viewpoint="ref", nowarp=False, chunk=8)


def render_sequence(model, frame_ids, H: int, W: int, chunk: int = 8) -> dict:
    """(F,H,W,C) maps of `frame_ids` (what dvr_model.evaluate returns for the surfel field: rendered, mask, rend_normal,
    surf_normal, surf_depth, ...), rendered under no_grad `chunk` frames at a time."""
    from .stage3 import make_intrinsics_inv
    outs = {}
    with torch.no_grad():
        for s in range(0, len(frame_ids), chunk):
            ids = torch.as_tensor(frame_ids[s:s + chunk], device=model._xyz.device)
            M = int(ids.shape[0])
            r = model.render_frames(ids, make_intrinsics_inv(M, H, W), [H] * M, [W] * M)
            for k, v in r.items():
                if isinstance(v, torch.Tensor) and v.dim() == 4:
                    outs.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in outs.items()}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    saved = dict(STAGE3_FLAGS)
    STAGE3_FLAGS.update(RENDER_FLAGS)
    try:
        opts, ignored = parse_flags(argv)
    finally:
        for k in RENDER_FLAGS:
            STAGE3_FLAGS.pop(k, None)
        STAGE3_FLAGS.update(saved)
    if not torch.cuda.is_available():
        raise SystemExit("rendering needs a GPU: the rasterizer has no CPU path")
    from . import checkpoint
    from .deformable_surfels import DeformableSurfels
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    logdir = os.path.join(opts["logroot"], f"{opts['seqname']}-{opts['logname']}")
    path = opts["load_path"] or os.path.join(logdir, "ckpt_%s.pth" % opts["load_suffix"])
    if not os.path.exists(path):
        raise SystemExit(f"checkpoint {path} not found")
    raw = torch.load(path, map_location="cpu", weights_only=False)
    states = checkpoint.remove_ddp_prefix(raw["model"])
    n = int(states[checkpoint.FG_PREFIX + "_xyz"].shape[0])
    model = DeformableSurfels(opts, num_frames=opts["num_frames"], device=dev)
    # (placeholders on the device: load_checkpoint re-creates the surfel tensors with the checkpoint's point count)
    for k in checkpoint.SURFEL_KEYS:
        setattr(model, k, torch.nn.Parameter(torch.empty(0, device=dev)))
    info = checkpoint.load_checkpoint(path, model, None, map_location=dev)
    model.active_sh_degree = model.max_sh_degree
    res = int(opts["render_res"])
    frames = list(range(int(opts["num_frames"])))
    rendered = render_sequence(model, frames, res, res, chunk=int(opts["chunk"]))
    rendered["rendered"].clamp_(0, 1)
    rgb = np.float16(rendered["rendered"].cpu().numpy())
    mask = np.abs(np.float16(rendered["surf_normal"].cpu().numpy()) - np.float16(rendered["rend_normal"].cpu().numpy()))
    save_dir = os.path.join(logdir, "renderings_%04d" % opts["motion_id"], opts["viewpoint"])
    os.makedirs(save_dir, exist_ok=True)
    torch.save({"rgb": rgb, "mask": mask}, os.path.join(save_dir, "rgb.pth"))
    print(f"rendered {len(frames)} frames at {res}x{res} from {path} ({n} surfels; "
          f"{len(info['unexpected_keys'])} checkpoint keys without a counterpart) -> {save_dir}/rgb.pth")
    return save_dir


if __name__ == "__main__":
    main()
