"""numpy reference dense -> brick encoder for the tests (same rules as Volume::to_brick_grid in
volren_amd/csrc/grids.cpp: dilation 2, fp16 range rounded outwards, sequential slots, 3 min/max mips)."""
import numpy as np


def _half_down(x):
    h = np.float16(x)
    if np.float32(h) > np.float32(x):
        h = np.nextafter(h, np.float16(-np.inf))
    return h


def _half_up(x):
    h = np.float16(x)
    if np.float32(h) < np.float32(x):
        h = np.nextafter(h, np.float16(np.inf))
    return h


def encode_arrays(dense_zyx, transform=None):
    d = np.asarray(dense_zyx, np.float32)
    nz, ny, nx = d.shape
    up8 = lambda v: ((v + 7) // 8 + 7) // 8 * 8
    nbx, nby, nbz = up8(nx), up8(ny), up8(nz)
    pad = np.zeros((nbz * 8 + 4, nby * 8 + 4, nbx * 8 + 4), np.float32)
    pad[2:2 + nz, 2:2 + ny, 2:2 + nx] = d
    ind = np.zeros((nbz, nby, nbx), np.uint32)
    rng = np.zeros((nbz, nby, nbx), np.uint32)
    lo_f = np.zeros((nbz, nby, nbx), np.float32)
    hi_f = np.zeros((nbz, nby, nbx), np.float32)
    alloc = []
    for bz in range(nbz):
        for by in range(nby):
            for bx in range(nbx):
                blk = pad[bz * 8:bz * 8 + 12, by * 8:by * 8 + 12, bx * 8:bx * 8 + 12]
                hlo, hhi = _half_down(blk.min()), _half_up(blk.max())
                rng[bz, by, bx] = int(hlo.view(np.uint16)) | (int(hhi.view(np.uint16)) << 16)
                lo_f[bz, by, bx], hi_f[bz, by, bx] = np.float32(hlo), np.float32(hhi)
                if np.float32(hhi) != np.float32(hlo):
                    alloc.append((bz, by, bx))
    per_layer = nbx * nby
    layers = max(1, (len(alloc) + per_layer - 1) // per_layer)
    atlas = np.zeros((layers * 8, nby * 8, nbx * 8), np.uint8)
    for k, (bz, by, bx) in enumerate(alloc):
        px, py, pz = k % nbx, (k // nbx) % nby, k // per_layer
        ind[bz, by, bx] = (px << 22) | (py << 12) | (pz << 2)
        lo, hi = lo_f[bz, by, bx], hi_f[bz, by, bx]
        v = pad[bz * 8 + 2:bz * 8 + 10, by * 8 + 2:by * 8 + 10, bx * 8 + 2:bx * 8 + 10]
        inv = np.float32(255.0) / np.float32(hi - lo)
        q = np.floor((v - lo) * inv + np.float32(0.5))
        atlas[pz * 8:pz * 8 + 8, py * 8:py * 8 + 8, px * 8:px * 8 + 8] = np.clip(q, 0, 255).astype(np.uint8)
    mips = []
    cur_lo, cur_hi, cur_w = lo_f, hi_f, rng
    for _ in range(3):
        z, y, x = cur_lo.shape
        z2, y2, x2 = (z + 1) // 2, (y + 1) // 2, (x + 1) // 2
        plo = np.full((z2 * 2, y2 * 2, x2 * 2), np.inf, np.float32)
        phi = np.full((z2 * 2, y2 * 2, x2 * 2), -np.inf, np.float32)
        plo[:z, :y, :x] = cur_lo
        phi[:z, :y, :x] = cur_hi
        mlo = plo.reshape(z2, 2, y2, 2, x2, 2).min((1, 3, 5))
        mhi = phi.reshape(z2, 2, y2, 2, x2, 2).max((1, 3, 5))
        w = mlo.astype(np.float16).view(np.uint16).astype(np.uint32) | (mhi.astype(np.float16).view(np.uint16).astype(np.uint32) << 16)
        mips.append(((x2, y2, z2), w.reshape(-1)))
        cur_lo, cur_hi = mlo, mhi
    t = np.eye(4, dtype=np.float32).reshape(16) if transform is None else np.asarray(transform, np.float32).reshape(16)
    return dict(transform=t, n_bricks=(nbx, nby, nbz), min_maj=(float(lo_f.min()), float(hi_f.max())),
                brick_counter=len(alloc), indirection=ind.reshape(-1), rng=rng.reshape(-1),
                atlas_dim=(nbx * 8, nby * 8, layers * 8), atlas=atlas.reshape(-1), mips=mips)


def encode(dense_zyx, transform=None):
    """-> oracle.binding.Grid"""
    from oracle import binding as ob
    a = encode_arrays(dense_zyx, transform)
    g = ob.Grid()
    g.set(a["transform"], a["n_bricks"], a["min_maj"], a["brick_counter"], a["indirection"], a["rng"],
          a["atlas_dim"], a["atlas"], a["mips"])
    return g


def dense_fp16_arrays(dense_zyx, transform=None):
    """Dense fp16 grid + macro-cell ranges (8^3 cells dilated by 2 voxels, 3 min/max mips) -- same rules as
    vr::DenseGridF16 in volren_amd/csrc/grids.cpp.  Cells: ceil(dim / 8) per axis (no rounding up to 8 cells)."""
    h = np.ascontiguousarray(dense_zyx, np.float16)
    d = h.astype(np.float32)
    nz, ny, nx = d.shape
    nbx, nby, nbz = (nx + 7) // 8, (ny + 7) // 8, (nz + 7) // 8
    pad = np.zeros((nbz * 8 + 4, nby * 8 + 4, nbx * 8 + 4), np.float32)
    pad[2:2 + nz, 2:2 + ny, 2:2 + nx] = d
    lo = np.zeros((nbz, nby, nbx), np.float32)
    hi = np.zeros((nbz, nby, nbx), np.float32)
    for bz in range(nbz):
        for by in range(nby):
            blk = pad[bz * 8:bz * 8 + 12, by * 8:by * 8 + 12]
            for bx in range(nbx):
                b = blk[:, :, bx * 8:bx * 8 + 12]
                lo[bz, by, bx], hi[bz, by, bx] = b.min(), b.max()
    word = lambda l, u: l.astype(np.float16).view(np.uint16).astype(np.uint32) | (u.astype(np.float16).view(np.uint16).astype(np.uint32) << 16)
    rng = word(lo, hi)
    mips = []
    cur_lo, cur_hi = lo, hi
    for _ in range(3):
        z, y, x = cur_lo.shape
        z2, y2, x2 = (z + 1) // 2, (y + 1) // 2, (x + 1) // 2
        plo = np.full((z2 * 2, y2 * 2, x2 * 2), np.inf, np.float32)
        phi = np.full((z2 * 2, y2 * 2, x2 * 2), -np.inf, np.float32)
        plo[:z, :y, :x] = cur_lo
        phi[:z, :y, :x] = cur_hi
        cur_lo = plo.reshape(z2, 2, y2, 2, x2, 2).min((1, 3, 5))
        cur_hi = phi.reshape(z2, 2, y2, 2, x2, 2).max((1, 3, 5))
        mips.append(((x2, y2, z2), word(cur_lo, cur_hi).reshape(-1)))
    t = np.eye(4, dtype=np.float32).reshape(16) if transform is None else np.asarray(transform, np.float32).reshape(16)
    return dict(transform=t, n_bricks=(nbx, nby, nbz), min_maj=(float(d.min()), float(d.max())), rng=rng.reshape(-1), mips=mips,
                extent=(nx, ny, nz), dense=h)


def encode_dense_fp16(dense_zyx, transform=None):
    """-> oracle.binding.Grid holding the dense fp16 voxels."""
    from oracle import binding as ob
    a = dense_fp16_arrays(dense_zyx, transform)
    g = ob.Grid()
    n = int(np.prod(a["n_bricks"]))
    g.set(a["transform"], a["n_bricks"], a["min_maj"], 0, np.zeros(n, np.uint32), a["rng"], (8, 8, 8), np.zeros(512, np.uint8),
          a["mips"], extent=a["extent"], dense=a["dense"])
    return g
