#!/usr/bin/env python3
"""Generates tests/golden/msm_fixtures.json: frozen MSM results from the DEFINITION (sum of big-int double-and-add scalar
multiplications, oracle/py/ecc.py - no Pippenger, no C++, no GPU) for the seeded inputs tests/helpers.py builds.

The reference holds no MSM input -> output vector ("MSM: parity unpinned" by its tests); the group law is
canonical, so any correct implementation yields these affine points.  Sizes follow SURVEY.md section 7 step 0.
Run from the repo root:  python tests/golden/make_msm_fixtures.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.py import ecc            # noqa: E402
from tests import helpers as H       # noqa: E402


def hexpt(P, f2=False):
    if P is None:
        return None
    if f2:
        return [[hex(P[0][0]), hex(P[0][1])], [hex(P[1][0]), hex(P[1][1])]]
    return [hex(P[0]), hex(P[1])]


def main():
    out = {"_source": "tests/golden/make_msm_fixtures.py: expected = sum_i s_i * P_i by the big-int definition; inputs: "
                      "tests/helpers.py seeded_points(curve, generator, n, 100 + n) / seeded_scalars(n, 200 + n, r) (edge scalars "
                      "0, 1, r - 1, 2^64, 2^136 - 1, 2 in the first six slots when n >= 8)", "g1": {}, "g2": {}}
    for n in (1, 2, 31, 32, 33, 256, 1024):
        pts = H.seeded_points(ecc.E1_377, ecc.G1_377, n, 100 + n)
        sc = H.seeded_scalars(n, 200 + n, ecc.R377)
        out["g1"][str(n)] = hexpt(ecc.E1_377.msm(pts, sc))
    for n in (1, 2, 33, 64):
        pts = H.seeded_points(ecc.E2_377, ecc.G2_377, n, 100 + n)
        sc = H.seeded_scalars(n, 200 + n, ecc.R377)
        out["g2"][str(n)] = hexpt(ecc.E2_377.msm(pts, sc), f2=True)
    with open(os.path.join(ROOT, "tests", "golden", "msm_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", {k: list(v) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
