"""Build-time guards of the hand-written kernels, run on the compiler's output (no GPU needed):
registers of hand-written scalar loads are not touched while the load is in flight, and the hot
kernels neither spill vector registers nor use scratch memory."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "vulkan-renderer_amd", "csrc")

import check_scalar_loads  # noqa: E402
import kernel_resources  # noqa: E402


def test_checker_sees_a_register_touched_in_flight():
    body = """
	s_mov_b32 s4, 0
	;;#ASMSTART
	s_load_dwordx4 s[8:11], s[2:3], 0x0
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
.LBB0_1:
	v_mov_b32_e32 v0, s12
	s_branch .LBB0_3
.LBB0_2:
	v_writelane_b32 v40, s9, 3
.LBB0_3:
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v1, s8
	s_endpgm
""".splitlines()
    bad, n = check_scalar_loads.check_function("k", body)
    assert n == 1
    assert len(bad) == 1 and bad[0][1] == [9] and "v_writelane" in bad[0][0]
    # a loop: the load at the bottom is still in flight at the top of the next iteration
    loop = """
.LBB0_1:
	s_add_u32 s20, s8, 1
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	;;#ASMSTART
	s_load_dwordx2 s[8:9], s[2:3], 0x0
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	s_endpgm
""".splitlines()
    bad, n = check_scalar_loads.check_function("k", loop)
    assert n == 1 and len(bad) == 1 and "s_add_u32" in bad[0][0]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("source", ["ggx_octets.hip", "ggx_hex.hip", "lambertian_split.hip", "lambertian_hex.hip",
                                    "lambertian_o48.hip"])
def test_hand_written_scalar_loads_are_left_alone(source):
    lines = check_scalar_loads.assembly(os.path.join(CSRC, source))
    loads = 0
    for name, body in check_scalar_loads.functions(lines):
        bad, n = check_scalar_loads.check_function(name, body)
        loads += n
        assert not bad, (name, bad[:3])
    assert loads >= 10          # the kernels do use them


@pytest.mark.timeout(600)
@pytest.mark.parametrize("source,kernel", [("ggx_octets.hip", "ggx_oct_pairs"), ("lambertian_split.hip", "ls_pairs"),
                                           ("lambertian_hex.hip", "lh_pairs"), ("lambertian_o48.hip", "lo_pairs")])
def test_hot_kernels_have_no_scratch(source, kernel):
    extra = ["-fno-slp-vectorize"] if source.startswith("lambertian_") else []
    found = 0
    for k in kernel_resources.resources(os.path.join(CSRC, source), extra):
        if kernel in k["name"]:
            found += 1
            assert k.get("VGPRs Spill", 0) == 0 and k.get("ScratchSize [bytes/lane]", 0) == 0, k
    assert found >= 1


@pytest.mark.timeout(600)
@pytest.mark.parametrize("source,kernel", [("lambertian_split.hip", "ls_pairs"), ("lambertian_hex.hip", "lh_pairs"),
                                           ("lambertian_o48.hip", "lo_pairs")])
def test_loads_nobody_waits_for_have_no_destination_register(source, kernel):
    """These kernels touch records ahead with a load that nobody waits for.  Such a load must not name a
    register: its data arrives whenever it arrives, and a compiler that only knows where the instruction
    stands is free to hand the register to another value in between (a build that did faulted the GPU).
    So the touch is a load into LDS; every load into a register has a reader; and the wave waits for
    everything on its way into LDS before it ends."""
    import re
    lines = check_scalar_loads.assembly(os.path.join(CSRC, source))
    checked = 0
    for name, body in check_scalar_loads.functions(lines):
        if kernel not in name:
            continue
        code = [l.split(";")[0].strip() for l in body]
        code = [l for l in code if l and not l.startswith(".")]
        to_lds = [i for i, l in enumerate(code) if l.startswith("global_load_lds_dword")]
        assert to_lds, name
        assert any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in code[to_lds[-1]:]), name
        for i, l in enumerate(code):
            m = re.match(r"global_load_dword(?:x\d)? (v\d+|v\[\d+:\d+\]),", l)
            if not m:
                continue
            regs = re.findall(r"\d+", m.group(1))
            lo, hi = int(regs[0]), int(regs[-1])

            def reads(text):
                ops = text.split(None, 1)[1] if " " in text else ""
                srcs = ops.split(",", 1)[1] if "," in ops else ""          # everything behind the first operand
                if text.startswith(("global_store", "scratch_store", "ds_write", "buffer_store")):
                    srcs = ops
                singles = [int(v) for v in re.findall(r"\bv(\d+)\b", srcs)]
                ranges = [(int(a), int(b)) for a, b in re.findall(r"v\[(\d+):(\d+)\]", srcs)]
                return any(lo <= v <= hi for v in singles) or any(a <= hi and lo <= b for a, b in ranges)
            assert any(reads(o) for o in code[i + 1:] + code[:i]), (name, l)
        checked += 1
    assert checked >= 1
