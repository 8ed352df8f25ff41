#!/usr/bin/env python3
"""Generate the committed golden fixtures from the DATA files of the reference (no reference code is run).

Run once in the build container (needs /root/reference and PIL):  python tests/golden/make_golden.py

Outputs (all data, no source text):
  tests/golden/jpeg_rows.npz   - decoded RGB rows of the three README renders at --seed=42
                                 (sample.jpg = BASELINE config 2, sample_blur.jpg = config 3,
                                 final_scene.jpg = config 5; /root/reference/README.md:10-39).
                                 Row index k is the PPM/JPEG row (0 = top); render row j = H-1-k.
  assets/earthmap.ppm          - /root/reference/earthmap.jpg (the texture asset opened by
                                 worlds.rs:180,441) decoded to binary P6 by PIL/libjpeg.  The reference
                                 decodes with jpeg-decoder 0.1.22 whose IDCT rounding may differ by
                                 +-1..2 LSB: parity on the earth sphere is unpinned beyond that.
"""
import os
import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

ROWS_RANDOM = [40, 150, 260, 370, 480, 590, 700, 760]          # render rows j (0 = bottom)
ROWS_FINAL = list(range(20, 800, 40))


def rows_of(name, rows):
    im = np.asarray(Image.open(os.path.join(REF, name)).convert("RGB"))
    h = im.shape[0]
    return np.stack([im[h - 1 - j] for j in rows]).astype(np.uint8), im.shape


def main():
    out = {}
    for key, name, rows in (("sample", "sample.jpg", ROWS_RANDOM), ("sample_blur", "sample_blur.jpg", ROWS_RANDOM),
                            ("final_scene", "final_scene.jpg", ROWS_FINAL)):
        data, shape = rows_of(name, rows)
        out[key + "_rows"] = np.array(rows, dtype=np.int32)
        out[key + "_rgb"] = data
        out[key + "_shape"] = np.array(shape, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "jpeg_rows.npz"), **out)
    earth = np.asarray(Image.open(os.path.join(REF, "earthmap.jpg")).convert("RGB"))
    os.makedirs(os.path.join(ROOT, "assets"), exist_ok=True)
    with open(os.path.join(ROOT, "assets", "earthmap.ppm"), "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (earth.shape[1], earth.shape[0]))
        f.write(np.ascontiguousarray(earth).tobytes())
    print("wrote jpeg_rows.npz and assets/earthmap.ppm", earth.shape)


if __name__ == "__main__":
    main()
