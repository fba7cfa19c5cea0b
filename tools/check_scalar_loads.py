#!/usr/bin/env python3
"""Build-time guard for the hand-written scalar loads (s_load_dword* in inline asm whose
s_waitcnt sits in a later asm statement): between such a load and the next `s_waitcnt lgkmcnt(0)`
on every path, no instruction may read or write the registers the load fills -- the compiler does not
see inside inline asm, so a copy or a spill it places there would move registers that are not loaded yet.

    python tools/check_scalar_loads.py vulkan-renderer_amd/csrc/ggx_octets.hip [more sources]

Compiles each source to gfx950 assembly, walks every function's control-flow graph with the set of
registers in flight (union over predecessors, to a fixed point) and reports every instruction that
touches one.  Exit status 1 if there is any."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vulkan-renderer_amd", "csrc"),
         "-S", "--cuda-device-only", "-Wno-unused-command-line-argument"]
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def assembly(source):
    extra = ["-fno-slp-vectorize"] if source.endswith(("lambertian_split.hip", "lambertian_hex.hip", "lambertian_o48.hip")) else []
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", f.name, source], capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr)
        return open(f.name).read().splitlines()


def functions(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                yield name, body
                name = None
            else:
                body.append(ln)


def check_function(name, body):
    """-> list of (line text, registers) violations, number of hand-written loads."""
    # instructions: (text, in_asm); basic blocks by labels and branches
    blocks, cur, labels, in_asm = [], {"label": None, "ins": []}, {}, False
    for ln in body:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        cur["ins"].append((t, in_asm))
        if t.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"] is not None:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            labels[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b["ins"][-1][0] if b["ins"] else ""
        s = []
        if last.startswith("s_endpgm"):
            pass
        elif last.startswith("s_branch"):
            s.append(labels[last.split()[-1]])
        else:
            if last.startswith("s_cbranch"):
                s.append(labels[last.split()[-1]])
            if i + 1 < len(blocks):
                s.append(i + 1)
        succ.append(s)
    state_in = [set() for _ in blocks]
    work, bad, n_loads = list(range(len(blocks))), {}, 0
    seen_loads = set()
    while work:
        i = work.pop()
        pending = set(state_in[i])
        for k, (t, in_asm) in enumerate(blocks[i]["ins"]):
            op = t.split()[0]
            if op == "s_waitcnt" and "lgkmcnt(0)" in t:
                pending.clear()
                continue
            regs = sregs(t)
            hit = regs & pending
            if hit:
                bad[(i, k)] = (t, sorted(hit))
            if in_asm and op.startswith("s_load_dword"):
                if (i, k) not in seen_loads:
                    seen_loads.add((i, k))
                    n_loads += 1
                pending |= sregs(t.split(",")[0])
        for j in succ[i]:
            if not pending <= state_in[j]:
                state_in[j] |= pending
                work.append(j)
    return list(bad.values()), n_loads


def main():
    status = 0
    for source in sys.argv[1:]:
        total = 0
        for name, body in functions(assembly(source)):
            bad, n_loads = check_function(name, body)
            total += n_loads
            for text, regs in bad:
                status = 1
                print("%s: %s\n    touches s%s while a hand-written scalar load into it is in flight: %s"
                      % (os.path.basename(source), name, regs, text))
        print("%s: %d hand-written scalar loads checked" % (os.path.basename(source), total))
    return status


if __name__ == "__main__":
    sys.exit(main())
