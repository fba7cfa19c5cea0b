#!/usr/bin/env python3
"""Register use of every kernel of one HIP source, from the compiler's own report
(-Rpass-analysis=kernel-resource-usage): name, SGPRs, VGPRs, spills, scratch, occupancy.

    python tools/kernel_resources.py vulkan-renderer_amd/csrc/ggx_octets.hip [pattern]

Exit status 1 if a kernel whose name contains `pattern` spills scalar or vector registers or
uses scratch memory."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vulkan-renderer_amd", "csrc"),
         "-Rpass-analysis=kernel-resource-usage"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        return [l.split("(")[0] for l in out] if len(out) == len(names) else names
    except OSError:
        return names


def resources(source, extra=()):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-c", source, "-o", os.devnull],
                       capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        text = m.group(1)
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.split(":", 1)
            try:
                cur[k.strip()] = int(v)
            except ValueError:
                cur[k.strip()] = v.strip()
    for k, n in zip(kernels, demangle([k["name"] for k in kernels])):
        k["name"] = n
    return kernels


def main():
    source = sys.argv[1]
    pattern = sys.argv[2] if len(sys.argv) > 2 else None
    extra = ["-fno-slp-vectorize"] if source.endswith(("lambertian_split.hip", "lambertian_hex.hip", "lambertian_o48.hip")) else []
    bad = 0
    for k in resources(source, extra):
        spill = k.get("SGPRs Spill", 0), k.get("VGPRs Spill", 0), k.get("ScratchSize [bytes/lane]", 0)
        mark = ""
        if pattern and pattern in k["name"] and any(spill):
            bad, mark = 1, "   <-- spills"
        print("%-70s sgpr %3d vgpr %3d agpr %3d  spill s/v %d/%d scratch %d occupancy %s%s" % (
            k["name"][-70:], k.get("TotalSGPRs", -1), k.get("VGPRs", -1), k.get("AGPRs", 0), spill[0], spill[1], spill[2],
            k.get("Occupancy [waves/SIMD]", "?"), mark))
    return bad


if __name__ == "__main__":
    sys.exit(main())
