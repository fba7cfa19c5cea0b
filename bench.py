#!/usr/bin/env python3
"""Benchmark of the IBL precompute hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c4|c3|c5|c2|c1] [--no-extras]

A "step" is one pass of the hot path over one synthetic environment that is
already resident in HBM.  The default workload is configuration 4 of the task's list, the one
the metric ("env-prefilter GSamples/s at 1/2/4/8 MI355X") is quoted on across
1-8 GPUs: the brute-force Lambertian convolution of a 1024x1024/face environment
into 128x128/face (6.18e11 (output, input) texel pairs).  With N > 1 (one process
per GPU, launched by torch.distributed.run) the output texels are split into N
shards that keep mirror-image texels together (the kernel evaluates the 8 sign
flips of a direction at once), every rank bakes its shard against the full input,
packs its texels, and one RCCL all-gather per cube followed by a scatter assembles
the cube on every rank; total work is fixed, so the scaling is "strong".

Rank 0 prints ONE JSON line.  `value` is samples of the whole job per second.
`roofline` prices the FP32 arithmetic the dominant kernel EXECUTED -- the records its
wavefronts walked (and, for the kernels that work pattern by pattern, the sign patterns
they evaluated), counted by the kernel itself in an untimed pass, times the flop of the
instructions its inner loop issues -- over that kernel's mean duration (HIP events on its
own stream, read from the library's ring of event pairs after the timed loop) against the
157.3 TFLOP/s FP32 vector peak; `algorithmic_equiv_tflops` prices the same launch at the
flop count a pair-by-pair evaluation would need (the default Lambertian kernel needs 18
vector instructions for 64 texel pairs, so this figure exceeds the peak many times over).
`cpu_baseline` is the CPU oracle (a port of the reference's loop, recomputing the input
direction per pair as the reference does) timed on a bounded sample of the same workload.
After the headline the run also times the same workload on the exact-weights kernel
(`c4_exact_weights_*`) and the GGX chains of configurations 3 and 5 (`c3_*`, `c5_*` keys),
since the full bake spends 99 % of its time there.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3     # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
FLOP_PER_LAMBERTIAN_PAIR = 12       # pair by pair: dot 5, max 1, 3 FMA 6
# What lambertian_oct_pairs issues per lane (its inner loop, hipcc -S): per record 3 multiplies,
# 3 adds for the pattern sums, 3 adds + 3 compares for the ballots (12, of which the always-live
# pattern's own add is counted with the pattern below: 11); per evaluated pattern one (clamped)
# add and 12 packed FMAs of 2 x 2 flop (49).
LAMB_FLOP_PER_LANE_RECORD = 11
LAMB_FLOP_PER_LANE_PATTERN = 49
# ls_pairs (the default: clamped cosine split into even and odd halves) per lane and record, as
# issued (hipcc -S): 2 multiplies and 2 FMAs for a+b, a-b and c (6), max, min, max, median (4),
# 4 adds, 6 packed FMAs of 2 x 2 flop (24): 38.  No per-pattern part.
LAMB_SPLIT_FLOP_PER_LANE_RECORD = 38
# lh_pairs (the default where the sizes allow: the same split over hexadecets, 256 texel pairs per lane
# and record) as issued (hipcc -S): 2 packed multiplies and a multiply for the products (5), 2 packed
# adds (4), max, min, max, median for each of the two weight sets (8), 6 packed adds (12), 18 packed
# FMAs of 2 x 2 flop (72): 101.
LAMB_HEX_FLOP_PER_LANE_RECORD = 101
# lo_pairs (the default where the sizes allow: the split over the orbits of all 48 symmetries of the cube,
# 2304 texel pairs per lane and record) as issued (hipcc -S): 4 packed multiplies and a multiply for the nine
# products (9), 38 packed adds (76), max, min, max, median for each of the six permutations (24), 4 adds,
# 98 packed FMAs of 2 x 2 flop (392): 505.
LAMB_O48_FLOP_PER_LANE_RECORD = 505
# ggx_oct_pairs per lane, record and pattern: cosine 5 (every record walked); then, for the records
# evaluated (all of them at narrow lobes; at wide lobes those some lane of the wave accepts):
# u/t/d 5, rcp + sqrt 2, two clamped ops 2, three multiplies 3, the weight sum's FMA 2 and 12 packed
# FMAs (48): 62.
GGX_FLOP_PER_LANE_WALKED = 5
GGX_FLOP_PER_LANE_EVALUATED = 62
# Matrix form of ggx_oct_pairs (the default for inputs of 64 x 64 and more): the vector pipe keeps the
# cosine (5) and the weight (u, t, d as FMAs 6, sqrt + rcp 2, the window's clamped subtraction 1, four
# multiplies 4: 13), and cuts the weight into two bf16 pieces (2 conversions, 1 v_dot2c_f32_bf16 counted as a
# fused multiply-add: 4): 17 per record evaluated.  Where the window ends above the horizon and the input is 1024 x 1024 or
# more the weight is u, t, d 6, the clamped subtraction 1, d^2, t d^2 2, an FMA 2, rsq 1, four
# multiplies 4: 16, with the cut 20.  The 25 sums are v_mfma_f32_32x32x16_bf16: 3 partial
# products x 32 columns x 2 flop = 192 per lane and record on the matrix pipe (peak 2.5 PFLOP/s dense).
GGX_MX_FLOP_PER_LANE_EVALUATED = 17
GGX_MX_RSQ_FLOP_PER_LANE_EVALUATED = 20
GGX_MX_MATRIX_FLOP_PER_LANE_RECORD = 192
# The 16-fold kernel (ggx_hex_pairs: sign flips and the exchange of x and y; the default where the output
# faces are a multiple of 16 wide) evaluates the same weight per lane and record -- the same vector
# instructions -- and multiplies it into 64 record columns instead of 32: 384 per lane and record on the
# matrix pipe; a weight then serves 16 texel pairs instead of 8.
GGX_HEX_MATRIX_FLOP_PER_LANE_RECORD = 384
BF16_MATRIX_PEAK_TFLOPS = 2500.0
FLOP_PER_GGX_ACCEPTED = 30          # pair by pair
FLOP_PER_GGX_REJECTED = 8
FLOP_PER_LUT_SAMPLE = 40


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--input-downscale", type=int, default=1,
                    help="c3/c5: box-filter the environment by this power of two first, as the tool's "
                         "--input-downscale does (mip sizes still derive from the full size)")
    ap.add_argument("--ggx-samples", type=int, default=0,
                    help="c3/c5: use the sampled GGX mode with this many samples per output texel (an addition; "
                         "0 = the reference's quadrature, the measured configuration)")
    ap.add_argument("--ggx-lod-bias", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-extras", action="store_true",
                    help="c4 only: skip the GGX chains of configurations 3 and 5 that are timed after the headline")
    return ap.parse_args()


class Workload:
    """One benchmark configuration: how to run a step on this rank and how much
    work the whole job does."""

    def __init__(self, name, args, ibl, rank, world, dist):
        import numpy as np
        import vulkan_renderer_amd as vra
        from vulkan_renderer_amd import capi, sharding
        self.name, self.ibl, self.rank, self.world, self.dist = name, ibl, rank, world, dist
        # several ranks on one GPU over gloo (rehearsal only): collectives go through host tensors,
        # see sharding.gather_cube
        self.via_host = world > 1 and dist.get_backend() == "gloo"
        # IBL_BENCH_EXCHANGE_ALWAYS=1: a single rank also packs, all-gathers (over RCCL) and scatters every cube,
        # which puts the fixed cost of the exchange step inside the timed step of a one-GPU run
        self.always = os.environ.get("IBL_BENCH_EXCHANGE_ALWAYS") == "1"
        self.sharding, self.capi = sharding, capi
        t = ibl.torch
        self.kernel_ms = []
        self.pending_kernel_ms = 0
        self.executed_flop_per_launch = None
        if name in ("c1", "c4"):
            self.in_w, self.out_w = (64, 32) if name == "c1" else (1024, 128)
            strip = vra.synth.hdr_environment(self.in_w, seed=args.seed)
            rgbe = t.from_numpy(strip).to(ibl.device)
            self.env = ibl.rgbe_decode(rgbe).reshape(6, self.in_w, self.in_w, 4).contiguous()
            self.n_out = 6 * self.out_w * self.out_w
            self.n_in = 6 * self.in_w * self.in_w
            # output texels this rank bakes: the library's partition for these sizes (orbits, hexadecets or octets kept whole)
            idx = capi.shard_indices(capi.SHARD_LAMBERTIAN, self.in_w, self.out_w, 1, 1, rank, world)
            self.count = int((idx != 0xFFFFFFFF).sum())
            self.out = t.zeros((6, self.out_w, self.out_w, 4), dtype=t.float32, device=ibl.device)
            self.samples = self.n_out * self.n_in
            self.flop_per_launch = FLOP_PER_LAMBERTIAN_PAIR * self.count * self.n_in
            # what the kernel executes, counted by the kernel itself (untimed pass)
            ibl.set_counting(True)
            ibl.lambertian_shard(self.env, self.out_w, rank, world, out=self.out)
            c = ibl.last_counts()
            ibl.set_counting(False)
            self.split = ibl.get_tuning("lamb_split") != 0
            # (the hexadecet form: even, mirror-exact faces -- every power of two -- with an output of a multiple of 16 >= 32)
            self.hex = (self.split and ibl.get_tuning("lamb_hex") != 0 and ibl.get_tuning("general_only") == 0
                        and self.out_w % 16 == 0 and self.out_w >= 32 and self.in_w % 16 == 0 and self.in_w >= 32
                        and self.in_w & (self.in_w - 1) == 0 and self.out_w & (self.out_w - 1) == 0)
            # (the 48-fold form serves the same sizes)
            self.o48 = self.hex and ibl.get_tuning("lamb_o48") != 0
            self.exact_octets = ibl.last_exact_octets() if self.split else None
            if c["wave_records"] and self.split:
                self.executed_flop_per_launch = 64.0 * c["wave_records"] * (
                    LAMB_O48_FLOP_PER_LANE_RECORD if self.o48 else
                    LAMB_HEX_FLOP_PER_LANE_RECORD if self.hex else LAMB_SPLIT_FLOP_PER_LANE_RECORD)
                self.patterns_per_record = None
            elif c["wave_records"]:
                self.executed_flop_per_launch = 64.0 * (LAMB_FLOP_PER_LANE_RECORD * c["wave_records"]
                                                        + LAMB_FLOP_PER_LANE_PATTERN * c["wave_patterns"])
                self.patterns_per_record = c["wave_patterns"] / c["wave_records"]
            self.kernel = (("lo_pairs" if self.o48 else "lh_pairs" if self.hex else "ls_pairs") if self.split
                           else "lambertian_oct_pairs")
            self.desc = ("C%s: Lambertian irradiance %dx%d/face -> %dx%d/face, all (output, input) texel pairs"
                         % (name[1], self.in_w, self.in_w, self.out_w, self.out_w))
            self.bytes_per_launch = 24 * self.n_in + 16 * self.count
        elif name in ("c3", "c5"):
            self.in_w, self.min_size = (512, 16) if name == "c3" else (2048, 16)
            strip = vra.synth.hdr_environment(self.in_w, seed=args.seed)
            rgbe = t.from_numpy(strip).to(ibl.device)
            full = ibl.rgbe_decode(rgbe).reshape(6, self.in_w, self.in_w, 4).contiguous()
            self.down = args.input_downscale
            self.env = ibl.downscale(full, self.down) if self.down > 1 else full   # cube.cpp:472-504
            self.env_w = self.in_w // self.down
            self.mips = ibl.ggx_mip_levels(self.in_w, self.in_w, self.min_size)    # cube.cpp:473-474,513
            self.levels = []
            for m in range(1, self.mips + 1):
                ow = self.in_w >> m
                n_out = 6 * ow * ow
                self.levels.append(dict(m=m, ow=ow, n_out=n_out,
                                        out=t.zeros((6, ow, ow, 4), dtype=t.float32, device=ibl.device)))
            self.ggx_samples, self.lod_bias = args.ggx_samples, args.ggx_lod_bias
            # work counts, measured once by the counting kernels outside the timed region; every rank
            # counts its own shard and the sums over the ranks are the job's
            ibl.set_counting(True)
            mine = []
            for lv in self.levels:
                ibl.ggx_level_shard(self.env, lv["ow"], lv["m"], self.mips, rank, world, out=lv["out"])
                c = ibl.last_counts()
                lv["my_tested"], lv["my_accepted"], lv["my_wave_records"] = c["tested"], c["accepted"], c["wave_records"]
                lv["my_wave_evaluated"] = c["wave_patterns"]
                mine += [c["tested"], c["accepted"]]
            ibl.set_counting(False)
            tot = t.tensor(mine, dtype=t.float64, device="cpu" if self.via_host else ibl.device)
            if world > 1:
                dist.all_reduce(tot)
            tot = [int(v) for v in tot.cpu().tolist()]
            for i, lv in enumerate(self.levels):
                lv["tested"], lv["accepted"] = tot[2 * i], tot[2 * i + 1]
            tested = sum(lv["tested"] for lv in self.levels)
            accepted = sum(lv["accepted"] for lv in self.levels)
            self.samples = tested
            self.accepted = accepted
            lv1 = self.levels[0]
            self.flop_per_launch = (FLOP_PER_GGX_ACCEPTED * lv1["my_accepted"]
                                    + FLOP_PER_GGX_REJECTED * (lv1["my_tested"] - lv1["my_accepted"]))
            # the matrix form takes whole groups of 16 records: patches of 4 x 4 texels and more
            self.matrix = (ibl.get_tuning("ggx_matrix") != 0 and self.env_w >= 64 and not args.ggx_samples
                           and bool(lv1["my_wave_records"]))
            self.matrix_flop_per_launch = None
            # the 16-fold kernel serves level 1 (the library's rule: ggx_hex_eligible)
            self.hex = (self.matrix and ibl.get_tuning("ggx_hex") != 0 and lv1["ow"] % 16 == 0 and lv1["ow"] >= 32
                        and (self.env_w & (self.env_w - 1)) == 0)
            if lv1["my_wave_records"] and self.matrix:
                # which weight form level 1 takes: the library's rule (ggx_oct_launch)
                rsq_knob = ibl.get_tuning("ggx_rsq")
                narrow = ibl.ggx_level_params(1, self.mips, self.env_w, self.env_w)["cutoff"] >= 0.51
                self.rsq = narrow and (rsq_knob == 2 or (rsq_knob == 1 and self.env_w >= 1024))
                per_record = GGX_MX_RSQ_FLOP_PER_LANE_EVALUATED if self.rsq else GGX_MX_FLOP_PER_LANE_EVALUATED
                self.executed_flop_per_launch = 64.0 * (GGX_FLOP_PER_LANE_WALKED + per_record) * lv1["my_wave_records"]
                self.matrix_flop_per_launch = 64.0 * lv1["my_wave_records"] * (
                    GGX_HEX_MATRIX_FLOP_PER_LANE_RECORD if self.hex else GGX_MX_MATRIX_FLOP_PER_LANE_RECORD)
            elif lv1["my_wave_records"]:
                self.executed_flop_per_launch = 64.0 * (GGX_FLOP_PER_LANE_WALKED * lv1["my_wave_records"]
                                                        + GGX_FLOP_PER_LANE_EVALUATED * lv1["my_wave_evaluated"])
            self.kernel = ("ggx_hex_pairs (level 1)" if self.hex else
                           "ggx_oct_pairs%s (level 1)" % (", matrix form" if self.matrix else ""))
            self.desc = ("C%s: GGX prefilter chain of a %dx%d/face environment, %d levels, --input-downscale %d; "
                         "samples = in-patch texel tests (%.3g tested, %.3g accepted)"
                         % (name[1], self.in_w, self.in_w, self.mips, self.down, tested, accepted))
            self.bytes_per_launch = 32 * 6 * self.env_w * self.env_w + 16 * lv1["n_out"] // world
            if self.ggx_samples:
                # sampled mode: a sample is one (output texel, half vector) evaluation; ranks take
                # contiguous ranges of output texels and gather them
                self.samples = sum(lv["n_out"] for lv in self.levels) * self.ggx_samples
                for lv in self.levels:
                    lv["begin"], lv["count"] = sharding.shard_range(lv["n_out"], rank, world)
                self.flop_per_launch = 150.0 * self.levels[0]["count"] * self.ggx_samples
                self.executed_flop_per_launch = None
                self.kernel = "ggx_sampled_texels (level 1)"
                self.desc = ("C%s (sampled GGX mode, not the reference's algorithm): chain of a %dx%d/face environment, "
                             "%d levels, --input-downscale %d, %d samples per output texel, lod bias %g"
                             % (name[1], self.in_w, self.in_w, self.mips, self.down, self.ggx_samples, self.lod_bias))
                self.bytes_per_launch = 16 * 6 * self.env_w * self.env_w * 4 // 3 + 16 * self.levels[0]["count"]
            if name == "c5":
                # plus the Lambertian irradiance of the 1024x1024/face (2x box-filtered) environment,
                # or of the downscaled one when --input-downscale is coarser than that
                lam_down = max(2, self.down)
                self.lam_env = ibl.downscale(full, lam_down)
                self.lam_in = self.in_w // lam_down
                self.lam_w = 128
                self.lam_out = t.zeros((6, self.lam_w, self.lam_w, 4), dtype=t.float32, device=ibl.device)
                pairs = 6 * self.lam_w * self.lam_w * 6 * self.lam_in * self.lam_in
                self.samples += pairs
                self.desc += "; + Lambertian %dx%d/face -> 128x128/face (%.3g pairs)" % (self.lam_in, self.lam_in, pairs)
        elif name == "c2":
            self.s = 32
            self.n_cells = 256 * 256
            self.begin, self.count = sharding.shard_range(self.n_cells, rank, world)
            self.out = t.zeros((256, 256, 2), dtype=t.float32, device=ibl.device)
            self.samples = self.n_cells * self.s * self.s
            self.flop_per_launch = FLOP_PER_LUT_SAMPLE * self.count * self.s * self.s
            self.kernel = "brdf_lut_cells"
            self.desc = "C2: split-sum BRDF table 256x256 cells, 32x32 = 1024 strata per cell"
            self.bytes_per_launch = 8 * self.count
        self.np = np

    def collect_kernel_ms(self):
        """Kernel durations of the steps whose event pairs have not been read yet."""
        for back in range(self.pending_kernel_ms - 1, -1, -1):
            self.kernel_ms.append(self.ibl.kernel_ms_back(back))
        self.pending_kernel_ms = 0

    def _gather(self, kind, in_w, cube, mip=1, mips=1):
        """The exchange step: pack this rank's shard, one all-gather, scatter every shard."""
        self.sharding.gather_cube(self.dist, self.ibl, kind, in_w, cube, mip, mips, self.rank, self.world,
                                  via_host=self.via_host, always=self.always)

    def step(self):
        ibl, d = self.ibl, self.dist
        if self.name in ("c1", "c4"):
            ibl.lambertian_shard(self.env, self.out_w, self.rank, self.world, out=self.out)
            # (kernel times are read after the loop, from the library's ring of event pairs: a wait
            # here would expose the host's launch latency once per 4 ms step)
            self.pending_kernel_ms += 1 if self.count else 0
            if self.pending_kernel_ms == 60:
                self.collect_kernel_ms()
            self._gather(self.capi.SHARD_LAMBERTIAN, self.in_w, self.out)
        elif self.name in ("c3", "c5"):
            if self.name == "c5":
                ibl.lambertian_shard(self.lam_env, self.lam_w, self.rank, self.world, out=self.lam_out)
                self._gather(self.capi.SHARD_LAMBERTIAN, self.lam_in, self.lam_out)
            if self.ggx_samples:
                pyr = ibl.env_pyramid(self.env)
                for lv in self.levels:
                    ibl.ggx_level_sampled(pyr, self.env_w, lv["ow"], lv["m"], self.mips, self.ggx_samples, self.lod_bias,
                                          out_begin=lv["begin"], out_count=lv["count"], out=lv["out"])
                    if lv["m"] == 1:
                        self.kernel_ms.append(ibl.last_kernel_ms() if lv["count"] else 0.0)
                    if self.world > 1:
                        self.sharding.all_gather_ranges(d, lv["out"].view(-1, 4), lv["n_out"], self.rank, self.world,
                                                        via_host=self.via_host)
                return
            for lv in self.levels:
                ibl.ggx_level_shard(self.env, lv["ow"], lv["m"], self.mips, self.rank, self.world, out=lv["out"])
                if lv["m"] == 1:
                    self.kernel_ms.append(ibl.last_kernel_ms() if lv["my_wave_records"] or self.world == 1 else 0.0)
                self._gather(self.capi.SHARD_GGX, self.env_w, lv["out"], lv["m"], self.mips)
        else:
            ibl.brdf_lut(256, 256, self.s, seed=1, cell_begin=self.begin, cell_count=self.count, out=self.out)
            self.kernel_ms.append(ibl.last_kernel_ms() if self.count else 0.0)
            self.sharding.all_gather_ranges(d, self.out.view(-1, 2), self.n_cells, self.rank, self.world,
                                            via_host=self.via_host)


def cpu_baseline(wl, target_seconds):
    """Times the CPU oracle (reference-faithful loop) on a bounded sample of the
    same workload and checks the GPU result on that sample while it is at it."""
    import numpy as np
    import oracle
    ncores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
    env = wl.env.cpu().numpy() if hasattr(wl, "env") else None
    if wl.name in ("c1", "c4"):
        n_out = wl.n_out
        pilot = min(n_out, 32 * ncores)
        t0 = time.time()
        oracle.lambertian(env, wl.out_w, out_begin=n_out // 2, out_count=pilot, faithful=True)
        rate = pilot / max(time.time() - t0, 1e-3)
        count = int(max(pilot, min(n_out, rate * target_seconds)))
        begin = (n_out - count) // 2
        t0 = time.time()
        want = oracle.lambertian(env, wl.out_w, out_begin=begin, out_count=count, faithful=True)
        dt = time.time() - t0
        got = wl.out.cpu().numpy().reshape(-1, 4)[begin:begin + count]
        want = want.reshape(-1, 4)[begin:begin + count]
        units = count * wl.n_in
        sample = "%d of %d output texels (range %d..%d) x all %d input texels" % (count, n_out, begin, begin + count, wl.n_in)
    elif wl.name in ("c3", "c5") and wl.ggx_samples:
        # sampled mode: the CPU statement of the same estimator, same samples
        lv = wl.levels[1]
        n_out = lv["n_out"]
        pyr = oracle.env_pyramid(env)
        pilot = min(n_out, 32 * ncores)
        t0 = time.time()
        oracle.ggx_level_sampled(pyr, wl.env_w, lv["ow"], lv["m"], wl.mips, wl.ggx_samples, wl.lod_bias, n_out // 2, pilot)
        rate = pilot / max(time.time() - t0, 1e-3)
        count = int(max(pilot, min(n_out, rate * target_seconds)))
        begin = (n_out - count) // 2
        t0 = time.time()
        want = oracle.ggx_level_sampled(pyr, wl.env_w, lv["ow"], lv["m"], wl.mips, wl.ggx_samples, wl.lod_bias, begin, count)
        dt = time.time() - t0
        got = lv["out"].cpu().numpy().reshape(-1, 4)[begin:begin + count]
        want = want.reshape(-1, 4)[begin:begin + count]
        units = count * wl.ggx_samples
        sample = ("sampled mode, level 2 of %d, %d of %d output texels (range %d..%d) x %d samples"
                  % (wl.mips, count, n_out, begin, begin + count, wl.ggx_samples))
    elif wl.name in ("c3", "c5"):
        lv = wl.levels[1]          # level 2: representative cost, full hemisphere
        n_out = lv["n_out"]
        pilot = min(n_out, 2 * ncores)
        t0 = time.time()
        _, c = oracle.ggx_level(env, lv["ow"], lv["ow"], lv["m"], wl.mips, out_begin=n_out // 2, out_count=pilot,
                                want_counts=True)
        rate = pilot / max(time.time() - t0, 1e-3)
        count = int(max(pilot, min(n_out, rate * target_seconds)))
        begin = (n_out - count) // 2
        t0 = time.time()
        want, c = oracle.ggx_level(env, lv["ow"], lv["ow"], lv["m"], wl.mips, out_begin=begin, out_count=count,
                                   want_counts=True)
        dt = time.time() - t0
        got = lv["out"].cpu().numpy().reshape(-1, 4)[begin:begin + count]
        want = want.reshape(-1, 4)[begin:begin + count]
        units = c["tested"]
        sample = "level 2 of %d, %d of %d output texels (range %d..%d)" % (wl.mips, count, n_out, begin, begin + count)
    else:
        t0 = time.time()
        want = oracle.brdf_lut(256, 256, wl.s, seed=1, mode=oracle.MODE_COUNTER)
        pilot = max(time.time() - t0, 1e-3)
        reps = max(1, int(target_seconds / pilot))
        t0 = time.time()
        for _ in range(reps):
            want = oracle.brdf_lut(256, 256, wl.s, seed=1, mode=oracle.MODE_COUNTER)
        dt = time.time() - t0
        got = wl.out.cpu().numpy().reshape(-1, 2).astype(np.float64)
        want = want.reshape(-1, 2)
        units = reps * 256 * 256 * wl.s * wl.s
        sample = "%d passes over the whole 256x256 table" % reps
        return dict(value=units / dt / 1e9, unit="GSamples/s", cores=ncores, kind="port", sample=sample,
                    seconds=round(dt, 2), max_abs_err=float(np.abs(got - want).max()))
    err = np.abs(got[:, :3].astype(np.float64) - want[:, :3]) / np.maximum(np.abs(want[:, :3]), 1e-30)
    same = float((oracle.rgbe_encode(got) == oracle.rgbe_encode(want)).all(axis=1).mean())
    return dict(value=units / dt / 1e9, unit="GSamples/s", cores=ncores, kind="port", sample=sample,
                seconds=round(dt, 2), max_rel_err=float(err.max()), rgbe_byte_identical=same)


def run_timed(wl, steps, warmup, torch, dist, world, device):
    """Warm-up, then `steps` steps between two fences; returns (seconds of the slowest rank, mean
    duration in ms of the dominant kernel on the slowest rank)."""
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        wl.step()
    fence()
    wl.collect_kernel_ms()
    wl.kernel_ms.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    fence()
    elapsed = time.perf_counter() - t0
    wl.collect_kernel_ms()
    kernel_ms = sum(wl.kernel_ms) / max(len(wl.kernel_ms), 1)
    if world > 1:
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if wl.via_host else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tt[0].item()), float(tt[1].item())
    return elapsed, kernel_ms


def chain_extra(name, args, ibl, rank, world, dist, torch, steps, cpu_seconds):
    """The GGX chain of configuration 3 or 5 (C5 with its Lambertian), timed like the headline;
    returns flat keys for the JSON line."""
    wl = Workload(name, args, ibl, rank, world, dist)
    elapsed, kernel_ms = run_timed(wl, steps, 1, torch, dist, world, ibl.device)
    per_chain = elapsed / steps
    out = {
        name + "_workload": wl.desc,
        name + "_steps": steps,
        name + "_ms_per_chain": round(per_chain * 1e3, 3),
        name + "_gsamples": round(wl.samples / per_chain / 1e9, 1),
        name + "_level1_kernel_ms": round(kernel_ms, 4),
    }
    if wl.executed_flop_per_launch and kernel_ms > 0:
        tf = wl.executed_flop_per_launch / (kernel_ms * 1e-3) / 1e12
        out[name + "_level1_executed_tflops"] = round(tf, 2)
        out[name + "_level1_frac_of_fp32_peak"] = round(tf / FP32_VECTOR_PEAK_TFLOPS, 4)
        out[name + "_level1_algorithmic_equiv_tflops"] = round(wl.flop_per_launch / (kernel_ms * 1e-3) / 1e12, 2)
        if getattr(wl, "matrix_flop_per_launch", None):
            # the two pipes' shares of the kernel's time, and their sum
            mtf = wl.matrix_flop_per_launch / (kernel_ms * 1e-3) / 1e12
            out[name + "_level1_matrix_bf16_tflops"] = round(mtf, 1)
            out[name + "_level1_frac_of_bf16_matrix_peak"] = round(mtf / BF16_MATRIX_PEAK_TFLOPS, 4)
            out[name + "_level1_both_pipes_frac"] = round(tf / FP32_VECTOR_PEAK_TFLOPS + mtf / BF16_MATRIX_PEAK_TFLOPS, 4)
    if world == 1 and cpu_seconds > 0:
        cb = cpu_baseline(wl, cpu_seconds)
        out[name + "_cpu_port_gsamples"] = round(cb["value"], 4)
        out[name + "_cpu_cores"] = cb["cores"]
        out[name + "_cpu_sample"] = cb["sample"]
        out[name + "_cpu_sample_max_rel_err"] = cb.get("max_rel_err")
        out[name + "_cpu_sample_rgbe_byte_identical"] = cb.get("rgbe_byte_identical")
    if getattr(wl, "matrix", False):
        # Like for like with the reference, which multiplies in FP32: the same chain with the sums on the vector
        # pipe (ggx_matrix 0), timed the same way in the same run.
        out[name + "_kernel"] = wl.kernel
        out[name + "_products"] = ("2 bf16 pieces per factor (16 significant bits) on the matrix pipe; patches whose "
                                   "radiance sits in a few texels: 3 pieces per factor, every product to 2^-25")
        ibl.set_tuning("ggx_matrix", 0)
        try:
            # what the vector form of level 1 executes, counted by its own counting variant
            lv1 = wl.levels[0]
            ibl.set_counting(True)
            ibl.ggx_level_shard(wl.env, lv1["ow"], lv1["m"], wl.mips, rank, world, out=lv1["out"])
            c = ibl.last_counts()
            ibl.set_counting(False)
            wl.vector_executed_flop_per_launch = (64.0 * (GGX_FLOP_PER_LANE_WALKED * c["wave_records"]
                                                          + GGX_FLOP_PER_LANE_EVALUATED * c["wave_patterns"])
                                                  if c["wave_records"] else None)
            wl.kernel_ms.clear()
            f_steps = max(1, steps // 2)
            f_elapsed, f_kernel_ms = run_timed(wl, f_steps, 1, torch, dist, world, ibl.device)
        finally:
            ibl.set_tuning("ggx_matrix", 1)
        f_chain = f_elapsed / f_steps
        out[name + "_fp32_kernel"] = "ggx_oct_pairs, vector form"
        out[name + "_fp32_products"] = "FP32 on the vector pipe throughout (ggx_matrix 0)"
        out[name + "_fp32_steps"] = f_steps
        out[name + "_fp32_ms_per_chain"] = round(f_chain * 1e3, 3)
        out[name + "_fp32_gsamples"] = round(wl.samples / f_chain / 1e9, 1)
        out[name + "_fp32_level1_kernel_ms"] = round(f_kernel_ms, 4)
        if wl.vector_executed_flop_per_launch and f_kernel_ms > 0:
            tf = wl.vector_executed_flop_per_launch / (f_kernel_ms * 1e-3) / 1e12
            out[name + "_fp32_level1_executed_tflops"] = round(tf, 2)
            out[name + "_fp32_level1_frac_of_fp32_peak"] = round(tf / FP32_VECTOR_PEAK_TFLOPS, 4)
    del wl
    torch.cuda.empty_cache()
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(gpus, argv, environ, port=None):
    """`python bench.py --gpus N` without a launcher around it: the command and environment of the
    child that starts one rank per GPU (torch.distributed.run on this node), or None when this
    process already is a rank (WORLD_SIZE set) or a single GPU is asked for.  Pure: nothing is
    started and nothing touches the GPU here."""
    if gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "16")                 # the CPU baseline is not run with N > 1
    env["IBL_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port()),
           os.path.abspath(__file__)] + list(argv)
    return cmd, env


def main():
    args = parse_args()
    launch = self_launch_command(args.gpus, sys.argv[1:], os.environ)
    if launch is not None:
        # Nothing in this process has touched the GPU (torch is not even imported): the ranks are fresh
        # child processes, rank 0 prints the one JSON line on the stdout they inherit, and this
        # process passes their exit status on.
        import subprocess
        raise SystemExit(subprocess.run(launch[0], env=launch[1]).returncode)
    import torch
    import torch.distributed as dist
    import vulkan_renderer_amd as vra

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # Rehearsal knobs for a one-GPU box (not used by the driver): all ranks on cuda:0
    # over gloo, since RCCL refuses two ranks on one device.
    backend = os.environ.get("IBL_BENCH_BACKEND", "nccl")
    if os.environ.get("IBL_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    grouped = world > 1 or os.environ.get("IBL_BENCH_EXCHANGE_ALWAYS") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    ibl = vra.Ibl(local_rank)
    ibl.set_timing(True)
    wl = Workload(args.workload, args, ibl, rank, world, dist)
    elapsed, kernel_ms = run_timed(wl, args.steps, args.warmup, torch, dist, world, ibl.device)

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = wl.samples / (elapsed / args.steps) / 1e9
        algorithmic = wl.flop_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        if wl.executed_flop_per_launch and kernel_ms > 0:
            achieved = wl.executed_flop_per_launch / (kernel_ms * 1e-3) / 1e12
            basis = "executed: records and sign patterns counted by the kernel x the instruction counts of its inner loop"
        else:
            achieved = algorithmic
            basis = "algorithmic flop count (this kernel evaluates pair by pair)"
        line = {
            "metric": "env-prefilter GSamples/s",
            "value": round(value, 3),
            "unit": "GSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.desc,
                       "parallelism": "output shards over %d GPU(s), one all-gather per cube%s" % (
                           world, " (run for the single rank too)" if wl.always else ""),
                       "seed": args.seed},
            "roofline": {
                "bound": "fp32_valu",
                "kernel": wl.kernel,
                "achieved": round(achieved, 3),
                "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4),
                "traffic": None,
                "basis": basis,
                "kernel_ms": round(kernel_ms, 4),
                "executed_flop_per_launch": wl.executed_flop_per_launch,
                "algorithmic_equiv_tflops": round(algorithmic, 3),
                "algorithmic_flop_per_launch": wl.flop_per_launch,
                "algorithmic_bytes_per_launch": wl.bytes_per_launch,
            },
        }
        if getattr(wl, "patterns_per_record", None):
            line["roofline"]["sign_patterns_per_record"] = round(wl.patterns_per_record, 3)
        if getattr(wl, "matrix_flop_per_launch", None) and kernel_ms > 0:
            # the sums of this kernel run on the matrix pipe; its time is the sum of the two pipes' shares
            mtf = wl.matrix_flop_per_launch / (kernel_ms * 1e-3) / 1e12
            line["roofline"]["matrix_bf16_tflops"] = round(mtf, 1)
            line["roofline"]["frac_of_bf16_matrix_peak"] = round(mtf / BF16_MATRIX_PEAK_TFLOPS, 4)
            line["roofline"]["both_pipes_frac"] = round(achieved / FP32_VECTOR_PEAK_TFLOPS + mtf / BF16_MATRIX_PEAK_TFLOPS, 4)
        if getattr(wl, "rsq", False):
            line["roofline"]["weight_form"] = "one v_rsq_f32 per record (window ends above the horizon)"
        if getattr(wl, "exact_octets", None) is not None:
            # output octets whose even and odd halves cancel too far and were redone with exact weights
            line["roofline"]["exact_weight_octets"] = wl.exact_octets
        try:   # measured HBM traffic of the same launch, from the committed rocprofv3 PMC passes
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                table = json.load(f)
            tr = table.get(wl.name)
            if tr and not tr["kernel"].startswith(wl.kernel):       # the same workload on another kernel
                tr = next((v for k, v in table.items() if k.startswith(wl.name + "_") and v["kernel"].startswith(wl.kernel)), None)
            if tr and world == 1 and not getattr(wl, "ggx_samples", 0):
                line["roofline"]["traffic"] = tr["bytes"]
                line["roofline"]["traffic_source"] = tr["source"]
                line["roofline"]["traffic_measured_in_run"] = False     # a constant from the committed PMC passes
        except OSError:
            pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)

    # The same workload on the exact-weights octet kernel (every pair weighted as the reference
    # weights it; the default falls back to it octet by octet where its split cancels too far).
    if args.workload == "c4" and not args.no_extras and getattr(wl, "split", False):
        ibl.set_tuning("lamb_split", 0)
        try:
            wx = Workload("c4", args, ibl, rank, world, dist)
            ex_elapsed, ex_kernel_ms = run_timed(wx, 3, 1, torch, dist, world, ibl.device)
            if line is not None:
                line["c4_exact_weights_ms_per_step"] = round(ex_elapsed / 3 * 1e3, 4)
                line["c4_exact_weights_kernel"] = wx.kernel
                line["c4_exact_weights_kernel_ms"] = round(ex_kernel_ms, 4)
                if wx.executed_flop_per_launch and ex_kernel_ms > 0:
                    tf = wx.executed_flop_per_launch / (ex_kernel_ms * 1e-3) / 1e12
                    line["c4_exact_weights_executed_tflops"] = round(tf, 2)
                    line["c4_exact_weights_frac_of_fp32_peak"] = round(tf / FP32_VECTOR_PEAK_TFLOPS, 4)
            del wx
        finally:
            ibl.set_tuning("lamb_split", 1)

    # ... on the split form over hexadecets (16 symmetries, every cosine the reference's to the bit) ...
    if args.workload == "c4" and not args.no_extras and getattr(wl, "o48", False):
        ibl.set_tuning("lamb_o48", 0)
        try:
            wh = Workload("c4", args, ibl, rank, world, dist)
            h_elapsed, h_kernel_ms = run_timed(wh, 10, 5, torch, dist, world, ibl.device)
            if line is not None:
                line["c4_hexadecets_ms_per_step"] = round(h_elapsed / 10 * 1e3, 4)
                line["c4_hexadecets_kernel"] = wh.kernel
                line["c4_hexadecets_kernel_ms"] = round(h_kernel_ms, 4)
                if wh.executed_flop_per_launch and h_kernel_ms > 0:
                    line["c4_hexadecets_frac_of_fp32_peak"] = round(
                        wh.executed_flop_per_launch / (h_kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)
            del wh
        finally:
            ibl.set_tuning("lamb_o48", 1)

    # ... and on the split form over octets (what the two replaced where the sizes allow it).
    if args.workload == "c4" and not args.no_extras and getattr(wl, "hex", False):
        ibl.set_tuning("lamb_hex", 0)
        try:
            wo = Workload("c4", args, ibl, rank, world, dist)
            o_elapsed, o_kernel_ms = run_timed(wo, 10, 5, torch, dist, world, ibl.device)
            if line is not None:
                line["c4_octets_split_ms_per_step"] = round(o_elapsed / 10 * 1e3, 4)
                line["c4_octets_split_kernel"] = wo.kernel
                line["c4_octets_split_kernel_ms"] = round(o_kernel_ms, 4)
            del wo
        finally:
            ibl.set_tuning("lamb_hex", 1)

    # The rest of the bake: the GGX chains of configurations 3 and 5, every rank taking part.
    if args.workload == "c4" and not args.no_extras:
        del wl
        torch.cuda.empty_cache()
        cpu_s = 0.0 if args.no_cpu_baseline else 5.0
        for name, steps in (("c3", 5), ("c5", 2)):
            try:
                extra = chain_extra(name, args, ibl, rank, world, dist, torch, steps, cpu_s)
            except Exception as e:     # the headline above stands on its own
                extra = {name + "_error": "%s: %s" % (type(e).__name__, e)}
            if line is not None:
                line.update(extra)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
