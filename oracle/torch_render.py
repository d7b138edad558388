"""Pure-PyTorch CPU restatement of the reference surfel rasterizer (second, independent oracle).

TEST INFRASTRUCTURE ONLY (see oracle/surfel_oracle.c): used by tests/ to cross-check the C oracle's
hand-restated analytic backward against torch.autograd, and by bench.py as the "pure-PyTorch CPU
render" baseline BASELINE.json's north_star asks to be timed beside the GPU number.  The product
package never imports it.

The forward follows /root/reference/gs/submodules/diff-surfel-rasterization/cuda_rasterizer/
forward.cu (:75-128 computeTransMat, :133-163 computeAABB, :166-260 preprocess, :265-463 render) and
rasterizer_impl.cu (:70-138 keys / ranges, :304-309 sort), vectorised per batch of tiles.  Its
autograd derivative equals the reference's *analytic* backward (backward.cu) because the places
where the reference backward is not the derivative of its forward are restated explicitly
(SURVEY.md §8a traps):
  1. alpha = min(0.99, o*G) is straight-through in backward (backward.cu:400,446);
  2. the quaternion normalisation inside quat_to_rotmat has no Jacobian in the vjp (auxiliary.h:213-257);
  3. the returned dL_dmeans2D is the densification statistic (dL_dT[2]*z*W/2, dL_dT[5]*z*H/2, 0)
     (backward.cu:645-648), not a gradient -- computed here from transMat.grad.
"""
from __future__ import annotations

import math

import torch

BLOCK = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]
FILTER_SIZE = 0.7071067811865476
NEAR, FAR = 0.2, 100.0


def _quat_to_rotmat(q):
    """auxiliary.h:188-210 with the normalisation treated as a constant (trap 2)."""
    inv = (1.0 / q.norm(dim=1, keepdim=True)).detach()
    w, x, y, z = (q * inv).unbind(1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def _sh_to_rgb(deg, means, campos, shs):
    d = means - campos
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6] + SH_C2[3] * xz * shs[:, 7]
                   + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def preprocess(means3D, scales, rotations, opacities, shs, colors_precomp, viewmatrix, campos, W, H, tanfovx,
               tanfovy, sh_degree):
    """forward.cu:166-260.  Returns a dict of per-surfel tensors; `transMat` is a graph intermediate
    with retain_grad() so that trap 3 can be evaluated after backward."""
    dt = means3D.dtype
    vm = viewmatrix.to(dt)
    Wm = vm[:3, :3].t()  # W (forward.cu:79-83): W[r][c] = vm[c][r]
    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)
    cx, cy = W / 2.0, H / 2.0
    p_view = means3D @ Wm.t() + vm[3, :3]
    R = _quat_to_rotmat(rotations)
    RS0 = R[:, :, 0] * scales[:, 0:1]
    RS1 = R[:, :, 1] * scales[:, 1:2]
    M0 = RS0 @ Wm.t()
    M1 = RS1 @ Wm.t()
    tn = R[:, :, 2] @ Wm.t()
    cos = -(tn * p_view).sum(1)
    mult = torch.where(cos > 0, torch.ones_like(cos), -torch.ones_like(cos))
    normal = tn * mult[:, None]
    Mz = torch.stack([M0[:, 2], M1[:, 2], p_view[:, 2]], 1)
    Tu = focal_x * torch.stack([M0[:, 0], M1[:, 0], p_view[:, 0]], 1) + cx * Mz
    Tv = focal_y * torch.stack([M0[:, 1], M1[:, 1], p_view[:, 1]], 1) + cy * Mz
    Tw = Mz
    transMat = torch.cat([Tu, Tv, Tw], 1)
    if transMat.requires_grad:
        transMat.retain_grad()
    Tu, Tv, Tw = transMat[:, 0:3], transMat[:, 3:6], transMat[:, 6:9]
    sgn = torch.tensor([1.0, 1.0, -1.0], dtype=dt)
    d = (sgn * Tw * Tw).sum(1)
    f = sgn[None] / d[:, None]
    center = torch.stack([(f * Tu * Tw).sum(1), (f * Tv * Tw).sum(1)], 1)
    h0 = center * center - torch.stack([(f * Tu * Tu).sum(1), (f * Tv * Tv).sum(1)], 1)
    extent = torch.sqrt(torch.clamp_min(h0.detach(), 0.0))
    radius = torch.ceil(3.0 * torch.clamp_min(extent.max(1).values.double(), FILTER_SIZE))
    radius_i = radius.to(torch.int64)
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    cen = center.detach().to(torch.float32)  # the rect arithmetic is fp32 in the reference
    rf = radius.to(torch.float32)

    def tile(v, g):
        return torch.clamp(torch.trunc(v / BLOCK).to(torch.int64), 0, g)

    rmin_x, rmin_y = tile(cen[:, 0] - rf, gx), tile(cen[:, 1] - rf, gy)
    rmax_x, rmax_y = tile(cen[:, 0] + rf + BLOCK - 1, gx), tile(cen[:, 1] + rf + BLOCK - 1, gy)
    ok = (p_view[:, 2] > 0.2) & (cos != 0) & (d != 0)
    tiles = (rmax_x - rmin_x) * (rmax_y - rmin_y)
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    radii = torch.where(tiles > 0, radius_i, torch.zeros_like(radius_i)).to(torch.int32)
    rgb = colors_precomp if colors_precomp is not None else _sh_to_rgb(sh_degree, means3D, campos.to(dt), shs)
    return dict(transMat=transMat, center=center, normal=normal, rgb=rgb, depth=p_view[:, 2].detach(),
                radii=radii, tiles_touched=tiles, rect=(rmin_x, rmin_y, rmax_x, rmax_y), grid=(gx, gy),
                opacity=opacities.reshape(-1))


def bin_and_sort(pre):
    """rasterizer_impl.cu:70-138 + :304-309: (tile | depth bits) keys, stable sort, tile ranges."""
    tiles = pre["tiles_touched"]
    rmin_x, rmin_y, rmax_x, _ = pre["rect"]
    gx, gy = pre["grid"]
    n = tiles.shape[0]
    ids = torch.repeat_interleave(torch.arange(n), tiles)
    offs = torch.cumsum(tiles, 0) - tiles
    k = torch.arange(ids.shape[0]) - offs[ids]
    wdt = (rmax_x - rmin_x)[ids]
    ty = rmin_y[ids] + k // torch.clamp_min(wdt, 1)
    tx = rmin_x[ids] + k % torch.clamp_min(wdt, 1)
    tile_id = ty * gx + tx
    dbits = pre["depth"].to(torch.float32).view(torch.int32).to(torch.int64)[ids]
    key = (tile_id << 32) | dbits
    skey, order = torch.sort(key, stable=True)
    point_list = ids[order]
    stile = skey >> 32
    t = torch.arange(gx * gy)
    start = torch.searchsorted(stile, t, right=False)
    end = torch.searchsorted(stile, t, right=True)
    ranges = torch.stack([start, end], 1)
    ranges = torch.where((end > start)[:, None], ranges, torch.zeros_like(ranges))
    return point_list, ranges, skey


def render(pre, point_list, ranges, bg, W, H, max_elems=1 << 21):
    """forward.cu:265-463, vectorised over batches of tiles with lists padded to the batch maximum."""
    dt = pre["transMat"].dtype
    gx, gy = pre["grid"]
    T9, xy_all, nrm_all, rgb_all, op_all = pre["transMat"], pre["center"], pre["normal"], pre["rgb"], pre["opacity"]
    lens = (ranges[:, 1] - ranges[:, 0])
    order = torch.argsort(lens, descending=True)
    color = torch.zeros(3, gy * BLOCK, gx * BLOCK, dtype=dt) + bg.to(dt)[:, None, None]
    others = torch.zeros(8, gy * BLOCK, gx * BLOCK, dtype=dt)
    n_contrib = torch.zeros(2, gy * BLOCK, gx * BLOCK, dtype=torch.int64)
    ly, lx = torch.meshgrid(torch.arange(BLOCK), torch.arange(BLOCK), indexing="ij")
    ly, lx = ly.reshape(-1), lx.reshape(-1)
    pos = 0
    ntiles = gx * gy
    while pos < ntiles:
        Lmax = int(lens[order[pos]])
        if Lmax == 0:
            break
        B = max(1, min(ntiles - pos, max_elems // (BLOCK * BLOCK * Lmax)))
        tsel = order[pos:pos + B]
        pos += B
        ar = torch.arange(Lmax)
        valid = ar[None, :] < lens[tsel][:, None]  # (B,L)
        gidx = torch.clamp(ranges[tsel, 0][:, None] + ar[None, :], max=point_list.shape[0] - 1)
        ids = point_list[gidx]  # (B,L)
        Tm = T9[ids]  # (B,L,9)
        Tu, Tv, Tw = Tm[..., 0:3], Tm[..., 3:6], Tm[..., 6:9]
        px = ((tsel % gx) * BLOCK)[:, None] + lx[None, :] + 0.5  # (B,256)
        py = ((tsel // gx) * BLOCK)[:, None] + ly[None, :] + 0.5
        px = px.to(dt)[:, :, None, None]
        py = py.to(dt)[:, :, None, None]
        k = -Tu[:, None] + px * Tw[:, None]  # (B,256,L,3)
        l = -Tv[:, None] + py * Tw[:, None]
        p = torch.linalg.cross(k, l, dim=-1)
        pz = p[..., 2]
        pz_ok = pz != 0
        pz_safe = torch.where(pz_ok, pz, torch.ones_like(pz))
        sx, sy = p[..., 0] / pz_safe, p[..., 1] / pz_safe
        rho3d = sx * sx + sy * sy
        dxy = xy_all[ids][:, None] - torch.cat([px, py], -1)  # (B,256,L,2)
        rho2d = 2.0 * (dxy * dxy).sum(-1)
        use3d = rho3d <= rho2d
        rho = torch.where(use3d, rho3d, rho2d)
        Twb = Tw[:, None]
        depth = torch.where(use3d, (sx * Twb[..., 0] + sy * Twb[..., 1]) + Twb[..., 2], Twb[..., 2].expand_as(sx))
        power = -0.5 * rho
        G = torch.exp(power)
        oG = op_all[ids][:, None] * G
        alpha = oG + (torch.clamp_max(oG, 0.99) - oG).detach()  # trap 1
        ok = valid[:, None] & pz_ok & (depth >= NEAR) & (power <= 0) & (alpha >= 1.0 / 255.0)
        a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
        T_incl = torch.cumprod(1 - a_eff, dim=-1)
        term = ok & (T_incl < 1e-4)
        contributes = ok & (torch.cumsum(term.to(torch.int32), -1) == 0)
        a_c = torch.where(contributes, alpha, torch.zeros_like(alpha))
        one_m = 1 - a_c
        T_incl = torch.cumprod(one_m, dim=-1)
        T_excl = T_incl / one_m
        T_final = T_incl[..., -1]
        w = a_c * T_excl
        dsafe = torch.where(contributes, depth, torch.ones_like(depth))
        m = (FAR * dsafe - FAR * NEAR) / ((FAR - NEAR) * dsafe)
        wm, wm2 = w * m, w * m * m
        d1_excl = torch.cumsum(wm, -1) - wm
        d2_excl = torch.cumsum(wm2, -1) - wm2
        A_excl = 1 - T_excl
        distortion = (w * (m * m * A_excl + d2_excl - 2 * m * d1_excl)).sum(-1)
        C = (w[..., None] * rgb_all[ids][:, None]).sum(-2)  # (B,256,3)
        Nn = (w[..., None] * nrm_all[ids][:, None]).sum(-2)
        D = (w * dsafe).sum(-1)
        idx1 = ar[None, None, :] + 1
        last = (contributes * idx1).max(-1).values
        med_mask = contributes & (T_excl > 0.5)
        med = (med_mask * idx1).max(-1).values  # 1-based, 0 = none
        med_g = torch.clamp(med - 1, min=0)[..., None]
        has_med = (med > 0).to(dt)
        med_depth = torch.gather(dsafe, -1, med_g)[..., 0] * has_med
        med_w = torch.gather(w, -1, med_g)[..., 0] * has_med
        ys = ((tsel // gx) * BLOCK)[:, None] + ly[None, :]
        xs = ((tsel % gx) * BLOCK)[:, None] + lx[None, :]
        bgc = bg.to(dt)
        col = C + T_final[..., None] * bgc
        color[:, ys, xs] = col.permute(2, 0, 1)
        others[:, ys, xs] = torch.stack([D, 1 - T_final, Nn[..., 0], Nn[..., 1], Nn[..., 2], med_depth, distortion,
                                         med_w], 0)
        n_contrib[0, ys, xs] = last
        n_contrib[1, ys, xs] = med
    return color[:, :H, :W], others[:, :H, :W], n_contrib[:, :H, :W]


def rasterize(means3D, opacities, scales, rotations, viewmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0,
              shs=None, colors_precomp=None, max_elems=1 << 21):
    """Forward of the reference op (color (3,H,W), radii (N,), others (8,H,W)) + a state dict."""
    pre = preprocess(means3D, scales, rotations, opacities, shs, colors_precomp, viewmatrix, campos, W, H,
                     float(tanfovx), float(tanfovy), sh_degree)
    with torch.no_grad():
        point_list, ranges, skey = bin_and_sort(pre)
    color, others, n_contrib = render(pre, point_list, ranges, bg, W, H, max_elems)
    state = dict(pre=pre, point_list=point_list, ranges=ranges, keys=skey, n_contrib=n_contrib)
    return color, pre["radii"], others, state


def means2D_statistic(state, W, H, tanfovx, tanfovy):
    """Trap 3: what the reference returns as dL_dmeans2D (backward.cu:645-648, :683-684)."""
    T = state["pre"]["transMat"]
    g = T.grad
    focal_x, focal_y = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    z = T.detach()[:, 8]
    vis = (state["pre"]["radii"] > 0).to(g.dtype)
    out = torch.zeros(T.shape[0], 3, dtype=g.dtype)
    out[:, 0] = g[:, 2] * z * (focal_x * tanfovx) * vis
    out[:, 1] = g[:, 5] * z * (focal_y * tanfovy) * vis
    return out
