#!/usr/bin/env python3
"""Fixture from the ONE output artefact the reference publishes for this path: assets/location.png, the
matplotlib rendering (`plt.imshow(locs)`) of the README quick-start (README.md:31-53) -- 800 x 800
perspective rays from (0, 0, 3) onto trimesh's default icosphere, `intersects_closest(...,
stream_compaction=True)`, `locs[hit] = location`, produced by the reference's own OptiX path on its
author's machine.  The figure's axes area (the 800 x 800 location map resampled to 370 x 369 pixels by
matplotlib, RGB = clip(loc, 0, 1)) is stored as uint8; everything around it (ticks, margins) is dropped.

Run in the build container (reads /root/reference, which does not travel to the GPU box):
    python tests/golden/make_readme_location_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
a = np.asarray(Image.open("/root/reference/assets/location.png").convert("RGB"))
nonwhite = (a != 255).any(-1)
rows = np.nonzero(nonwhite.mean(1) > 0.5)[0]
cols = np.nonzero(nonwhite.mean(0) > 0.5)[0]
axes = a[rows.min() + 1:rows.max(), cols.min() + 1:cols.max()]          # inside the one-pixel frame
assert axes.shape == (370, 369, 3), axes.shape
np.savez_compressed(os.path.join(HERE, "reference_readme_location_axes.npz"), axes=axes,
                    source=np.array("lcp29/trimesh-ray-optix assets/location.png (README.md:53), axes area"))
print("wrote", axes.shape, "disc pixels:", int((axes.astype(int).sum(-1) > 60).sum()))
