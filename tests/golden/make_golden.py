"""Regenerates tests/golden/ from the REFERENCE Fortran itself.

Run in the build container only (needs /root/reference and amdflang):

    python tests/golden/make_golden.py

It compiles the reference's module_configure.f90 + module_small_step_em.f90 from where they
lie into oracle/_ref/ (oracle/Makefile `ref`; nothing is copied into the repository), runs
`advance_mu_t` on this repository's seeded synthetic inputs and stores

* golden_small.npz     -- all 7 outputs, full arrays, of 16x8x16 (every flag combination)
                          and of the ragged / tile / one-level cases (flags none, specified)
* golden_digests.json  -- sha256 of every input and output array of every case (incl. 64x40x64)

The reference ships no golden vectors of its own (its drivers diff against an absent
/data2/... directory), so these are the pinned known answers.
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import __graft_entry__ as g  # noqa: E402
import cases  # noqa: E402


FULL_SHAPES = ("37x5x11_ragged", "130x3x7_tile", "70x1x9_onelevel")   # stored as full arrays


def main():
    oracle = g.load_oracle()
    oracle.build(ref=True)
    pkg = g.load_package()
    small, digests = {}, {}
    for shape in cases.SHAPES:
        for flag in cases.FLAG_COMBOS:
            for dtype in (np.float32, np.float64):
                key = f"{shape}/{flag}/{np.dtype(dtype).name}"
                p = cases.make_case(pkg, shape, flag, dtype)
                inputs = {n: cases.digest(a) for n, a in p.arrays.items()}
                oracle.ref_advance_mu_t(*p.args())
                digests[key] = {"bounds": list(p.bounds.as_tuple()), "inputs": inputs,
                                "outputs": {n: cases.digest(p.arrays[n]) for n in pkg.synth.FIELD_NAMES}}
                if shape == "16x8x16" or (shape in FULL_SHAPES and flag in ("none", "specified")):
                    for n in pkg.synth.OUTPUTS:
                        small[f"{key}/{n}"] = p.arrays[n]
    np.savez_compressed(HERE / "golden_small.npz", **small)
    (HERE / "golden_digests.json").write_text(json.dumps(digests, indent=1, sort_keys=True))
    print(f"wrote {len(small)} arrays, {len(digests)} cases")


if __name__ == "__main__":
    main()
