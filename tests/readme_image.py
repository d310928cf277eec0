"""Shared by the CPU and GPU tests: compare a location map of the README quick-start with the reference's
own published rendering of it (tests/golden/reference_readme_location_axes.npz, made from
assets/location.png by tests/golden/make_readme_location_fixture.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_readme_location_axes.npz")


def compare_with_reference_image(locs_800):
    """locs_800: float32 [800, 800, 3], loc where hit else 0 (README.md:49-51).  Returns the metrics."""
    from PIL import Image
    from scipy import ndimage
    ref = np.load(GOLDEN)["axes"].astype(np.float32)
    h, w = ref.shape[:2]
    img = (np.clip(locs_800, 0.0, 1.0) * 255.0).astype(np.uint8)                # what imshow draws for float RGB
    mine = np.asarray(Image.fromarray(img).resize((w, h), Image.LANCZOS)).astype(np.float32)
    disc_ref, disc_me = ref.sum(-1) > 60, mine.sum(-1) > 60
    core = ndimage.binary_erosion(disc_ref & disc_me, iterations=4)            # away from the resampled edge
    yr, xr = np.nonzero(disc_ref)
    ym, xm = np.nonzero(disc_me)
    return {"area_ref": float(disc_ref.mean()), "area": float(disc_me.mean()), "xor": float((disc_ref ^ disc_me).mean()),
            "radius_ref": float(np.sqrt(disc_ref.sum() / np.pi)), "radius": float(np.sqrt(disc_me.sum() / np.pi)),
            "centroid_shift": float(np.hypot(yr.mean() - ym.mean(), xr.mean() - xm.mean())),
            "interior_mae": float(np.abs(mine - ref)[core].mean()), "interior_max": float(np.abs(mine - ref)[core].max()),
            "interior_pixels": int(core.sum()), "background_max": float(mine[~ndimage.binary_dilation(disc_ref, iterations=4)].max())}


def assert_matches_reference_image(m):
    # silhouette: 800 * (1/sqrt(8)) / 2 px radius on the 800-px image = 65.2 px here
    assert abs(m["area"] - m["area_ref"]) < 0.005 * m["area_ref"] + 2e-4, m
    assert abs(m["radius"] - m["radius_ref"]) < 0.3 and m["xor"] < 0.003 and m["centroid_shift"] < 1.0, m
    # location values: RGB = clip(loc, 0, 1) * 255, compared away from the resampled silhouette edge
    assert m["interior_pixels"] > 10000 and m["interior_mae"] < 1.0 and m["interior_max"] <= 8, m
    assert m["background_max"] == 0.0, m            # misses stay (0, 0, 0)
