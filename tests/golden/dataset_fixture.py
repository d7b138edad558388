"""TEST INFRASTRUCTURE: writes a tiny two-video sequence in the reference's `database/` layout (seeded), for
tests/golden/make_refpy_golden.py (read back by the IMPORTED reference loader) and tests/test_refpy_vidloader.py
(read back by vidu4d_amd.lab4d.vidloader).  Layout: lab4d/dataloader/vidloader.py:81-128."""
import os

import numpy as np

SEQ = "toy-seq"
PREFIX = "crop-16"
FEATURE_TYPE = "cse"
DELTAS = [2, 4]
RES = 16
# (kept frame names = raw frame ids, number of raw frames)
VIDEOS = [([0, 1, 2, 4, 5, 6, 8], 9), ([0, 1, 2, 3, 4], 5)]
KS = [[20.0, 21.0, 8.0, 7.5], [18.0, 18.5, 8.5, 8.0]]
RAW_SIZE = [[48, 64], [40, 40]]
OPTS = {"delta_list": DELTAS, "data_prefix": PREFIX, "feature_type": FEATURE_TYPE, "pixels_per_image": 12,
        "load_pair": True, "seqname": SEQ}


def write_dataset(root, seed=0):
    """-> path of the .config file.  `root` plays the role of the directory that holds `database/`."""
    rng = np.random.default_rng(seed)
    proc = os.path.join(root, "database", "processed")
    cfg = ["[data]", "init_frame = 0", "end_frame = -1", ""]
    for v, (kept, n_raw) in enumerate(VIDEOS):
        vid = "%s-%04d" % (SEQ, v)
        F = len(kept)
        d = lambda kind: os.path.join(proc, kind, "Full-Resolution", vid)  # noqa: E731
        for kind in ("JPEGImages", "JPEGImagesRaw", "Annotations", "Depth", "Features", "Cameras"):
            os.makedirs(d(kind), exist_ok=True)
        for i in range(n_raw):
            open(os.path.join(d("JPEGImagesRaw"), "%05d.jpg" % i), "wb").close()
        for i in kept:
            open(os.path.join(d("JPEGImages"), "%05d.jpg" % i), "wb").close()
        np.save(os.path.join(d("JPEGImages"), PREFIX + ".npy"), rng.uniform(size=(F, RES, RES, 3)).astype(np.float16))
        ann = np.stack([rng.uniform(size=(F, RES, RES)) > 0.5, rng.uniform(size=(F, RES, RES)) > 0.1], -1)
        np.save(os.path.join(d("Annotations"), PREFIX + ".npy"), ann.astype(np.float16))
        c2r = np.concatenate([rng.uniform(2.0, 3.0, size=(F, 2)), rng.uniform(0.0, 10.0, size=(F, 2))], 1)
        np.save(os.path.join(d("Annotations"), PREFIX + "-crop2raw.npy"), c2r.astype(np.float32))
        np.save(os.path.join(d("Annotations"), PREFIX + "-is_detected.npy"), rng.uniform(size=F) > 0.2)
        np.save(os.path.join(d("Depth"), PREFIX + ".npy"), rng.uniform(0.5, 3.0, size=(F, RES, RES)).astype(np.float16))
        np.save(os.path.join(d("Features"), "%s-%s-01.npy" % (PREFIX, FEATURE_TYPE)),
                rng.normal(size=(F, 112, 112, 16)).astype(np.float16))
        for delta in [1] + DELTAS:
            for kind in ("FlowFW", "FlowBW"):
                p = os.path.join(proc, "%s_%d" % (kind, delta), "Full-Resolution", vid)
                os.makedirs(p, exist_ok=True)
                np.save(os.path.join(p, PREFIX + ".npy"),
                        rng.normal(size=(F // delta + 1, RES, RES, 3)).astype(np.float16))
        for name in ("00.npy", "01-canonical.npy"):
            rt = np.tile(np.eye(4, dtype=np.float32), (F, 1, 1))
            rt[:, :3, 3] = rng.normal(size=(F, 3))
            np.save(os.path.join(d("Cameras"), name), rt)
        cfg += ["[data_%d]" % v, "ks = " + " ".join(str(x) for x in KS[v]),
                "shape = " + " ".join(str(x) for x in RAW_SIZE[v]), "img_path = " + d("JPEGImages") + "/", ""]
    cdir = os.path.join(root, "database", "configs")
    os.makedirs(cdir, exist_ok=True)
    path = os.path.join(cdir, SEQ + ".config")
    with open(path, "w") as f:
        f.write("\n".join(cfg))
    return path


# what both sides read back (video id, frame index, delta); FIXED_XY = pixels for the sampled-pixel path
READS = [(0, 0, 1), (0, 2, 2), (0, 4, -2), (0, 3, -1), (0, 4, -4), (1, 1, 1), (1, 3, -1)]
FIXED_XY = np.array([[0, 0], [15, 15], [3, 7], [7, 3], [15, 0], [0, 15], [8, 8], [1, 14], [14, 1], [5, 5]])


def collect(datasets, get_data_info):
    """Runs the same reads on either implementation -> {name: array}."""
    out = {}
    for n, (v, idx, delta) in enumerate(READS):
        for tag, xy in (("full", None), ("xy", FIXED_XY)):
            d = datasets[v].read_raw(idx, delta, rand_xy=xy)
            for k, val in d.items():
                if k == "feature" and xy is None:
                    val = np.asarray(val)[::8, ::8] if n == 0 else None  # (full maps are 0.8 MB each: keep a sample of one)
                if val is not None:
                    out["read%d_%s_%s" % (n, tag, k)] = np.asarray(val)
    for v, ds in enumerate(datasets):
        np.random.seed(11 + v)
        ds.idx_sampler.init_queue()
        for rep in range(3):  # exercises sample_delta / sample_xy / the sampler's re-permutation
            d = ds[(0, 2, 3)[rep]]
            for k, val in d.items():
                out["item%d_%d_%s" % (v, rep, k)] = np.asarray(val)
        out["len%d" % v] = np.asarray(len(ds))
    info = get_data_info(datasets)
    fi = info["frame_info"]
    out["frame_offset"] = np.asarray(fi["frame_offset"])
    out["frame_offset_raw"] = np.asarray(fi["frame_offset_raw"])
    out["frame_mapping"] = np.asarray(fi["frame_mapping"])
    out["total_frames"] = np.asarray(info["total_frames"])
    out["intrinsics"] = np.asarray(info["intrinsics"])
    out["raw_size"] = np.asarray(info["raw_size"])
    out["rtmat"] = np.asarray(info["rtmat"])
    out["geom_names"] = np.asarray([os.path.basename(p) for p in info["geom_path"]])
    return out
