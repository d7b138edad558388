"""CPU estimate for a restructured blend walk: today a wave64 owns an 8x8 pixel quadrant and all 64
lanes evaluate the same list entry; in the "row walk" each 16-lane DPP row owns a 4x4 pixel block and walks ITS OWN
culled sub-list, so one wave iteration serves up to four (entry, block) pairs.  Counts, on the bench scene with the
product's cull box: (entry, quadrant) evaluations today vs wave iterations of the row walk (= per batch and wave, the
longest of its four rows), and how often two rows of a wave would be on the same entry in the same iteration.

    python tools/experiments/row_walk_sim.py [surfels] [res]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import torch_render as tr  # noqa: E402
from vidu4d_amd.synthetic import make_scene  # noqa: E402

BOX_MARGIN_PX = 0.02


def contribution_box(T, cx, cy, opacity):
    """surfel_math.h::contribution_box, vectorised (float64 is fine for a count)."""
    oa = opacity * 255.0
    ok = oa >= 1.0
    rc = 2.0 * np.log(np.maximum(oa, 1.0)) * 1.0001 + 1e-4
    r2 = np.sqrt(0.5 * rc)
    x0, x1, y0, y1 = cx - r2, cx + r2, cy - r2, cy + r2
    Tw0, Tw1, Tw2 = T[:, 6], T[:, 7], T[:, 8]
    Ux, Uy, Uz = T[:, 0] - cx * Tw0, T[:, 1] - cx * Tw1, T[:, 2] - cx * Tw2
    Vx, Vy, Vz = T[:, 3] - cy * Tw0, T[:, 4] - cy * Tw1, T[:, 5] - cy * Tw2
    d = rc * (Tw0 * Tw0 + Tw1 * Tw1) - Tw2 * Tw2
    good = d < -1e-3 * Tw2 * Tw2
    f = 1.0 / np.where(good, d, 1.0)
    ex = f * (rc * (Ux * Tw0 + Uy * Tw1) - Uz * Tw2)
    ey = f * (rc * (Vx * Tw0 + Vy * Tw1) - Vz * Tw2)
    hx = np.sqrt(np.maximum(ex * ex - f * (rc * (Ux * Ux + Uy * Uy) - Uz * Uz), 0.0))
    hy = np.sqrt(np.maximum(ey * ey - f * (rc * (Vx * Vx + Vy * Vy) - Vz * Vz), 0.0))
    BIG = 1e30
    x0 = np.where(good, np.minimum(x0, cx + ex - hx), -BIG)
    x1 = np.where(good, np.maximum(x1, cx + ex + hx), BIG)
    y0 = np.where(good, np.minimum(y0, cy + ey - hy), -BIG)
    y1 = np.where(good, np.maximum(y1, cy + ey + hy), BIG)
    mx = BOX_MARGIN_PX + 2e-3 * (x1 - x0)
    my = BOX_MARGIN_PX + 2e-3 * (y1 - y0)
    box = np.stack([x0 - mx, y0 - my, x1 + mx, y1 + my], 1)
    box[~ok] = [BIG, BIG, -BIG, -BIG]
    return box


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    sc = make_scene(n, res, None, seed=1234)
    with torch.no_grad():
        pre = tr.preprocess(sc.means3D.double(), sc.scales.double(), sc.rotations.double(), sc.opacities.double(),
                            sc.shs.double(), None, sc.viewmatrix, sc.campos, sc.width, sc.height, sc.tanfovx,
                            sc.tanfovy, 0)
        point_list, ranges, _ = tr.bin_and_sort(pre)
    T = pre["transMat"].numpy()
    cen = pre["center"].numpy()
    box = contribution_box(T, cen[:, 0], cen[:, 1], pre["opacity"].numpy())
    pl = point_list.numpy()
    rg = ranges.numpy()
    gx, gy = pre["grid"]
    print(f"{n} surfels {res}^2: {pl.shape[0]} (entry, tile) pairs, {gx * gy} tiles")

    def hits(b, x0, y0, w, h):  # pixel centres x0+0.5 .. x0+w-0.5
        return ~((b[:, 2] < x0 + 0.5) | (b[:, 0] > x0 + w - 0.5) | (b[:, 3] < y0 + 0.5) | (b[:, 1] > y0 + h - 0.5))

    tot_quadrant = 0
    for name, (bw, bh) in {"4x4 blocks": (4, 4), "8x2 strips": (8, 2)}.items():
        it_row = it_row_wholelist = pairs_row = same = iters_checked = 0
        tot_quadrant = 0
        for t in range(gx * gy):
            a, e = rg[t]
            if e <= a:
                continue
            b = box[pl[a:e]][::-1]  # back to front, as the backward stages it
            tx0, ty0 = (t % gx) * 16, (t // gx) * 16
            for q in range(4):
                qx0, qy0 = tx0 + (q & 1) * 8, ty0 + (q >> 1) * 8
                hq = hits(b, qx0, qy0, 8, 8)
                tot_quadrant += int(hq.sum())
                rows = []
                for r in range(4):
                    if (bw, bh) == (4, 4):
                        rows.append(hits(b, qx0 + (r & 1) * 4, qy0 + (r >> 1) * 4, 4, 4))
                    else:
                        rows.append(hits(b, qx0, qy0 + r * 2, 8, 2))
                rows = np.stack(rows)  # (4, L)
                pairs_row += int(rows.sum())
                it_row_wholelist += int(rows.sum(1).max())
                for s in range(0, rows.shape[1], batch):
                    rb = rows[:, s:s + batch]
                    cnt = rb.sum(1)
                    it_row += int(cnt.max())
                    # same-entry collisions: iteration i of row r handles its i-th set bit
                    m = int(cnt.max())
                    if m == 0:
                        continue
                    idx = np.full((4, m), -1 - np.arange(4)[:, None])
                    for r in range(4):
                        w = np.flatnonzero(rb[r])
                        idx[r, :w.size] = w
                    srt = np.sort(idx, 0)
                    same += int((srt[1:] == srt[:-1]).any(0).sum())
                    iters_checked += m
        print(f"{name}: (entry, quadrant) evaluations today {tot_quadrant}; (entry, row) pairs {pairs_row} "
              f"({pairs_row / tot_quadrant:.2f} per quadrant evaluation); row-walk wave iterations {it_row} "
              f"= {it_row / tot_quadrant:.3f} x today (batch {batch}), {it_row_wholelist / tot_quadrant:.3f} x with no "
              f"batch barriers; perfect balance {pairs_row / 4 / tot_quadrant:.3f}; iterations with two rows on the same "
              f"entry {same / max(1, iters_checked):.2%}")


if __name__ == "__main__":
    main()
