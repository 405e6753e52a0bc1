"""Framebuffer tile sharding across the GPUs of one node.

The frame's 16x16 tiles (the reference's work-group size, shader/pathtracer_brick.glsl:3) are dealt to the ranks on a
diagonal interleave, owner(tx, ty) = (tx + ty) mod N, so that box-missing sky tiles and expensive cloud-core tiles mix
on every rank.  A pixel-sample depends only on (seed, pixel, sample) and read-only scene data, so ranks never talk
while rendering; the accumulated radiance is exchanged ONCE per frame with an all_gather of equal-sized compact tile
buffers (RCCL over xGMI on the GPU box, gloo in the CPU tests) followed by a local scatter back into the frame.
The sharded image is bit-identical to the single-GPU image.

This module is pure index logic + the collective call; packing/unpacking is done by the caller (HIP kernels behind
vr_pack_tiles / vr_unpack_tiles in the product, numpy in the CPU tests).
"""
import numpy as np

TILE = 16
TEXELS = TILE * TILE * 4          # floats per packed tile (RGBA32F)


def tile_grid(w, h):
    return (w + TILE - 1) // TILE, (h + TILE - 1) // TILE


def _hash32(i):
    """Avalanching 32-bit integer hash (the finaliser of MurmurHash3)."""
    i = (int(i) + 0x9E3779B9) & 0xFFFFFFFF
    i = ((i ^ (i >> 16)) * 0x85EBCA6B) & 0xFFFFFFFF
    i = ((i ^ (i >> 13)) * 0xC2B2AE35) & 0xFFFFFFFF
    return i ^ (i >> 16)


DEFAULT_SCHEME = "diagonal"


def tile_owner_lists(w, h, n_ranks, scheme=None):
    """Raster tile ids (row 0 = bottom, like the framebuffer) owned by each rank.
    scheme "diagonal": owner(tx, ty) = (tx + ty) mod N.  "hashed": the tiles in the order of a 32-bit hash of their id, dealt round
    robin (counts differ by at most one; no correlation with any direction of the image)."""
    scheme = scheme or DEFAULT_SCHEME
    tiles_x, tiles_y = tile_grid(w, h)
    lists = [[] for _ in range(n_ranks)]
    if scheme == "hashed":
        order = sorted(range(tiles_x * tiles_y), key=lambda t: (_hash32(t), t))
        for i, t in enumerate(order):
            lists[i % n_ranks].append(t)
        return [sorted(t) for t in lists]
    if scheme != "diagonal":
        raise ValueError("unknown tile deal %r" % scheme)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            lists[(tx + ty) % n_ranks].append(ty * tiles_x + tx)
    return lists


class TileShard:
    """Tile ownership of one rank plus the padded layouts the all_gather needs."""

    def __init__(self, w, h, world, rank, scheme=None):
        self.w, self.h, self.world, self.rank = int(w), int(h), int(world), int(rank)
        self.lists = tile_owner_lists(w, h, world, scheme)
        self.mine = self.lists[rank]
        self.n_max = max(1, max(len(t) for t in self.lists))
        # pack list: own tiles, padded by repeating the last one (any valid tile; the slot is ignored on unpack)
        pad_src = self.mine[-1] if self.mine else 0
        self.pack_ids = np.asarray(self.mine + [pad_src] * (self.n_max - len(self.mine)), np.int32)
        # unpack list for the gathered buffer: every rank's tiles in rank order, -1 marks padding
        self.unpack_ids = np.asarray(sum((t + [-1] * (self.n_max - len(t)) for t in self.lists), []), np.int32)

    @property
    def packed_floats(self):
        return self.n_max * TEXELS

    @property
    def gathered_floats(self):
        return self.world * self.n_max * TEXELS

    def all_gather(self, dist, gathered, packed):
        """One collective per frame.  `dist` is torch.distributed; tensors live on the backend's device."""
        if self.world == 1:
            gathered.copy_(packed)
        else:
            dist.all_gather_into_tensor(gathered, packed)


# ---- numpy reference pack/unpack (CPU tests; same layout as pack_tiles_kernel / unpack_tiles_kernel) ----
def pack_tiles_numpy(fb, tile_ids):
    h, w, _ = fb.shape
    tiles_x, _ = tile_grid(w, h)
    out = np.zeros((len(tile_ids), TILE, TILE, 4), np.float32)
    for i, t in enumerate(tile_ids):
        tx, ty = int(t) % tiles_x, int(t) // tiles_x
        blk = fb[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
        out[i, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape(-1)


def unpack_tiles_numpy(packed, tile_ids, fb):
    h, w, _ = fb.shape
    tiles_x, _ = tile_grid(w, h)
    p = np.asarray(packed, np.float32).reshape(len(tile_ids), TILE, TILE, 4)
    for i, t in enumerate(tile_ids):
        if t < 0:
            continue
        tx, ty = int(t) % tiles_x, int(t) // tiles_x
        hh, ww = min(TILE, h - ty * TILE), min(TILE, w - tx * TILE)
        fb[ty * TILE:ty * TILE + hh, tx * TILE:tx * TILE + ww] = p[i, :hh, :ww]
    return fb
