"""Turns the reference's committed BRDF table (include/ggx_lut.inl, the output of
`bin/cube --generate-ggx-lut`, cube.cpp:346-389) into a binary fixture.

Run in the build container only (the reference tree does not travel):
    python tests/golden/make_ggx_lut_fixture.py
Writes tests/golden/ggx_lut_committed.npy: float32 [256 roughness][256 cos][slope,bias].
Only the numbers are kept; the one comment line of the file is dropped.
"""
import os
import re
import sys

import numpy as np

SRC = "/root/reference/include/ggx_lut.inl"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ggx_lut_committed.npy")


def main():
    vals = []
    with open(SRC) as f:
        for line in f:
            if line.lstrip().startswith("//"):
                continue
            vals += [float(t) for t in re.findall(r"[-+]?[0-9.]+(?:e[-+]?[0-9]+)?", line)]
    a = np.asarray(vals, dtype=np.float64)
    assert a.size == 256 * 256 * 2, a.size
    np.save(DST, a.reshape(256, 256, 2).astype(np.float32))
    print("wrote", DST, a.reshape(256, 256, 2)[0, :2])


if __name__ == "__main__":
    sys.exit(main())
