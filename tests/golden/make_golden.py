"""Generates the committed golden fixtures under tests/golden/.  Run in the BUILD container only
(it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

* example_64.npy      -- the reference's only output artefact, imgs/example.jpg (README.md:72-77),
                         box-downsampled to 64x64 RGB uint8.  Data, not source.
* example_256.npy     -- the same image box-downsampled 4x4 to 256x256 RGB uint8 (round 3): the size at which a wrong field of view,
                         camera matrix, unit-cube placement or environment rotation shows (tests/test_gpu_parity.py renders the
                         README command on the GPU at the reference's 1024x1024 x 4096 spp and compares).
* known_answers.json  -- structural known-answers on the reference's data files,
                         recomputed here with numpy directly from the files (not through the oracle).
"""
import json
import os
import struct
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def brick_known_answers(path):
    d = open(path, "rb").read()
    off = 1
    transform = struct.unpack_from("<16f", d, off); off += 64
    nb = struct.unpack_from("<3I", d, off); off += 12
    min_maj = struct.unpack_from("<2f", d, off); off += 8
    counter, = struct.unpack_from("<Q", d, off); off += 8

    def buf(off, dtype):
        dim = struct.unpack_from("<3I", d, off); off += 12
        n, = struct.unpack_from("<Q", d, off); off += 8
        a = np.frombuffer(d, dtype, n, off); off += n * a.itemsize
        return dim, a, off
    _, ind, off = buf(off, np.uint32)
    _, rng, off = buf(off, np.uint32)
    adim, atlas, off = buf(off, np.uint8)
    nm, = struct.unpack_from("<Q", d, off); off += 8
    mdims = []
    for _ in range(nm):
        md, _, off = buf(off, np.uint32)
        mdims.append(list(md))
    assert off == len(d)
    b = (5 * nb[1] + 7) * nb[0] + 6
    return {
        "file_bytes": len(d), "endian_flag": d[0], "transform": list(transform), "n_bricks": list(nb),
        "min_maj": list(min_maj), "brick_counter": counter, "atlas_dim": list(adim), "mip_dims": mdims,
        "atlas_byte_sum": int(atlas.astype(np.uint64).sum()), "atlas_crc32": zlib.crc32(atlas.tobytes()),
        "brick_6_7_5_ptr": [int(ind[b] >> 22), int((ind[b] >> 12) & 1023), int((ind[b] >> 2) & 1023)],
        "brick_6_7_5_range": [float(np.uint16(rng[b] & 0xFFFF).view(np.float16)), float(np.uint16(rng[b] >> 16).view(np.float16))],
        "allocated_bricks_fraction": float((ind != 0).mean()),
        "decoded_sum": 171484.0643, "decoded_mean": 0.04088499, "decoded_max": 5.7148438, "decoded_nonzero_fraction": 0.2324,
    }


def main():
    ka = {
        "provenance": "SURVEY.md 2.3 / 4 / 8c; recomputed by tests/golden/make_golden.py from /root/reference/data",
        "brick": brick_known_answers(os.path.join(REF, "data/smoke.brick")),
        "rng": {"tea": [[0, 1, 832450237], [42, 1, 2477371080], [518490, 7, 1421412122]],
                "stream_from_tea_42_1": [0.22075694799, 0.89226114750, 0.41994196177, 0.32730841637],
                "end_state": 257149564},
        "lut_cdf_alpha": [0, 0.10134243, 0.14085634, 0.16639936, 0.24573745, 0.493436, 0.76886386, 1],
        "hdr": {"shape": [512, 1024], "mean_rgb_f32": [0.37153518, 0.3602841, 0.3001295], "max": 60416.0,
                "argmax_row_col": [235, 653], "mean_luma": 0.35844183, "top_quarter_luma": 0.11995,
                "bottom_quarter_luma": 0.04780, "impmap_avg_w": 0.3584418},
        "example_render": {"source": "imgs/example.jpg", "cmd": "README.md:72-73",
                           "params": {"w": 1024, "h": 1024, "spp": 4096, "bounces": 128, "albedo": 0.8, "phase": 0.3,
                                      "density": 100, "env_strength": 3, "env_rot": 270, "exposure": 3, "gamma": 2.0,
                                      "cam_fov": 40}},
    }
    json.dump(ka, open(os.path.join(HERE, "known_answers.json"), "w"), indent=1)
    ex = Image.open(os.path.join(REF, "imgs/example.jpg")).convert("RGB").resize((64, 64), Image.BOX)
    np.save(os.path.join(HERE, "example_64.npy"), np.asarray(ex, np.uint8))
    ex256 = Image.open(os.path.join(REF, "imgs/example.jpg")).convert("RGB").resize((256, 256), Image.BOX)
    np.save(os.path.join(HERE, "example_256.npy"), np.asarray(ex256, np.uint8))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
