"""The algebra behind csrc/lambertian_o48.hip (tools/o48_identities.py): the kernel's tables are the group's, and
the transforms reproduce the pair-by-pair sum over two orbits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import o48_identities  # noqa: E402


def test_tables_are_the_groups():
    assert o48_identities.tables_are_consistent()


def test_the_kernels_tables_are_the_checked_ones():
    """The tables in the HIP source are the ones the numpy statement uses."""
    src = open(os.path.join(ROOT, "vulkan-renderer_amd", "csrc", "lambertian_o48.hip")).read()
    flat = lambda t: str(t).replace("[", "{").replace("]", "}")
    for name, table in (("PERM[6][3]", [list(p) for p in o48_identities.PERMS]), ("PAR[6]", o48_identities.PAR),
                        ("INV[6]", o48_identities.INV), ("EVEN[3][3]", o48_identities.EVEN), ("ODD[3][3]", o48_identities.ODD)):
        line = [l for l in src.splitlines() if "const int " + name in l]
        assert len(line) == 1, name
        assert flat(table).replace(" ", "") in line[0].replace(" ", ""), (name, line[0])
    body = src[src.index("const int R2I[6][2][2]"):]
    body = body[:body.index(";")].replace(" ", "").replace("\\", "").replace("\n", "")
    assert flat(o48_identities.R2I).replace(" ", "") in body


def test_transforms_equal_the_pair_by_pair_sum():
    e64, e32 = o48_identities.residuals(1024)
    assert e64 < 1e-12, e64           # an identity
    assert e32 < 2e-7, e32            # the kernel's float32 arithmetic, 256 records a chunk
