"""How far the Lambertian split forms may cancel: a 64 x 64/face environment with a light of rising strength on
one face, 32 x 32/face out, for the 48-fold form and the hexadecet form and several values of the knob
lamb_split_limit: octets redone with exact weights, worst texel against the CPU statement, RGBE texels identical.

    python tools/lamb_limit_sweep.py > profiles/<round>_lamb_limit_sweep.txt      (on an MI355X)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import vulkan_renderer_amd as vra  # noqa: E402
from helpers import byte_identical_fraction, decode_strip, rel_err  # noqa: E402


def main():
    oracle.lib()
    ibl = vra.Ibl(0)
    torch = ibl.torch
    base = decode_strip(oracle, vra.synth.hdr_environment(64, seed=1234))
    ibl.set_counting(True)
    try:
        for o48 in (1, 0):
            ibl.set_tuning("lamb_o48", o48)
            for strength in (1.5e3, 3e3, 6e3, 1e4, 3e4):
                env = base.copy()
                env[0, 20:24, 30:34, :3] = np.float32(strength)
                want = np.asarray(oracle.lambertian(env, 32)).reshape(-1, 4)
                env_d = torch.from_numpy(env).to(ibl.device)
                row = []
                for limit in (16, 32, 64, 1e9):
                    ibl.set_tuning("lamb_split_limit", limit)
                    got = ibl.lambertian(env_d, 32).cpu().numpy().reshape(-1, 4)
                    n = ibl.last_exact_octets()
                    frac = byte_identical_fraction(oracle.rgbe_encode(got), oracle.rgbe_encode(want))
                    row.append("limit %g: %3d redone, max %.2e, bytes %.5f" % (limit, n, rel_err(got, want).max(), frac))
                print("%s light %6g | %s" % ("48-fold  " if o48 else "hexadecet", strength, " | ".join(row)), flush=True)
    finally:
        ibl.set_counting(False)
        ibl.set_tuning("lamb_split_limit", 16)
        ibl.set_tuning("lamb_o48", 1)
    print("(the 48-fold form compares its measure with twice the knob: its column 'limit 16' is a limit of 32 on that measure)")


if __name__ == "__main__":
    main()
