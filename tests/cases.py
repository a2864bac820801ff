"""Shared parity cases: shapes, flag combinations and tile calls (used by the golden
generator, the CPU tests and the GPU tests)."""
import hashlib

import numpy as np

FLAG_COMBOS = {
    "none": dict(),
    "specified": dict(specified=True),
    "specified_periodic_x": dict(specified=True, periodic_x=True),
    "nested": dict(nested=True),
}

# name -> (NI, NK, NJ, tile override or None)
# tile override = dict(its=, ite=, jts=, jte=) relative to the single-patch domain
SHAPES = {
    "16x8x16": (16, 8, 16, None),
    "64x40x64": (64, 40, 64, None),               # BASELINE.json configs[0]
    "37x5x11_ragged": (37, 5, 11, None),          # nothing a multiple of anything
    "130x3x7_tile": (130, 3, 7, dict(its=60, ite=70, jts=3, jte=5)),   # interior tile call
    "70x1x9_onelevel": (70, 1, 9, None),          # k_end = 1: recurrence loop is empty
    "5x60x4_thin": (5, 60, 4, None),
}


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def make_case(pkg, shape_name, flag_name, dtype, seed=12345, aligned=False):
    ni, nk, nj, tile = SHAPES[shape_name]
    b = pkg.synth.domain_bounds(ni, nk, nj, aligned=aligned)
    if tile:
        b = b.replace(**tile)
    return pkg.synth.make_patch(b, pkg.GridConfig(**FLAG_COMBOS[flag_name]), dtype=dtype, seed=seed,
                                global_dims=(ni, nk, nj))
