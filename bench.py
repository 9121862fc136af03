#!/usr/bin/env python3
"""bench.py -- frames/s of the fixed-budget global optimisation on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c5|c2] [--scaling strong|weak]
                                                            (N > 1 without a launcher: starts the N ranks itself, as below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one complete pass of the hot path: FittingOP.fitting(mode='global') with the
reference's fixed budget (500 Adam iterations, phase split 400/100, global_optimization.py:672,
:564) over one synthetic clip whose inputs are already resident in HBM; the final 6D->angle-axis
conversion and the device->host copy of the results are inside the timed region, model / scene
upload is not.  Workload at every N: BASELINE config 3 -- 1024-frame clip,
500k-point scene, 500 contact vertices; with N > 1 the SAME clip is sharded over the ranks
(strong scaling), exchanging 2-frame halos + the scale gradient per iteration over RCCL.
--config c5 / c2: BASELINE config 5 (512 frames, 2 M-point scene, all 10 475 vertices as contacts) / config 2 (256 frames,
100 k points) instead; --scaling weak: every rank fits its OWN whole clip (no data-path collective; `value` = all ranks'
frames over the slowest rank's time) -- the arrangement that pays on 8 GPUs for a clip one GPU already fits.

Rank 0 prints ONE JSON line.
  `value`                   loss terms evaluated where they reach a gradient;
  `with_reference_logging`  the like-for-like figure: every term of every iteration evaluated and kept, as the reference
                            prints them (:573-575, :587-589) -- read the two as a pair;
  `exact_fp32`              the same step with the dense products as exact fp32 MFMA chains (FDCAP_GEMM_SPLIT3=0);
  `roofline`                the dominant kernel (in-loop Chamfer NN launch): its duration is measured live with HIP events
                            on the launch stream; the counters that say WHICH resource bounds it come from the committed
                            rocprofv3 PMC summary of the configuration (profiles/r5_<config>_pmc_summary.json,
                            tools/run_prof_r5.sh); `contract` keeps SURVEY §8d's algorithmic-bytes figure;
                            `roofline.per_kernel`: EVERY launch of a phase-1 iteration (eight since the blend forward and the skinning forward are one) -- microseconds from the committed
                            kernel trace, executed flops or algorithmic bytes, fraction of the peak that binds each;
  `other_configs`           BASELINE configs 2 and 5 measured in the same run (ms per fit, us per iteration, their Chamfer
                            launch and what bounds it, from their own PMC summaries);
  `setup`                   what is NOT in `value`: seconds in fdcap_ctx_create / fdcap_set_scene / fdcap_set_contact_ids, the first
                            fit against a steady one, and `cli_end_to_end` = frames / (registration + one fit) for a 300-frame clip
                            (the reference's real clip length; one process = one clip = one scene there);
  `cpu_baseline`            the oracle timed on this host's cores on a bounded sample.

--dry-run: no GPU, no kernels, `value` null -- only the multi-process plumbing of this file (gloo instead of RCCL: rendezvous,
LOCAL_RANK, frame shards, the per-iteration all-gather, barriers, max over ranks, rank-0-only JSON); tests/test_bench_plumbing.py
runs it under torch.distributed.run with two processes.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E 8 TB/s
BF16_DENSE_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA (same rate on gfx950)
FP32_MFMA_TFLOPS = 157.3       # fp32 MFMA (64 FLOP/clk/SIMD at 2.4 GHz)
NUM_SIMD = 1024                # 256 CUs x 4
NUM_XCD = 8
WAVES_PER_SIMD = 8
METRIC = "frames/sec global-opt (fixed iters), 1024f clip/500k-pt scene; Chamfer GB/s"
# BASELINE.json configs that fit one GPU, by the names DESIGN.md uses: (frames, scene points, all mesh vertices as contacts?)
CONFIGS = {"c3": (1024, 500_000, False), "c5": (512, 2_000_000, True), "c2": (256, 100_000, False)}
CONFIG_NAMES = {"c3": "BASELINE config 3", "c5": "BASELINE config 5 (Chamfer stress: dense scene, every vertex a contact)", "c2": "BASELINE config 2"}


def pmc_summary_path(cfg):
    """the newest committed PMC summary of the configuration (tools/run_prof_r6.sh; r5's where r6 has none)"""
    for rnd in ("r6", "r5"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_pmc_summary.json")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", f"r6_{cfg}_pmc_summary.json")
# VALU issue cost per wave64 instruction on one gfx950 SIMD, MEASURED (tools/valu_issue_probe.hip, 8 waves per SIMD, per physical
# SIMD; profiles/r4_valu_issue_probe.txt): v_fma / v_add / v_mul / v_sub_f32, v_and_b32, v_add_u32 issue every ~2.2 cycles (the
# guide's "2 cycles"); v_min3 / v_min / v_med3 / v_cmp_f32, v_alignbit_b32, 64-bit shifts and the packed fp32 forms every ~4.16.
# There is no single "VALU peak": a kernel is priced by its own mix (SQ_INSTS_VALU_{FMA,ADD,MUL}_F32 + _INT32 counted in the fast
# class -- an upper bound on it, some integer forms are slow --, everything else that is not an MFMA in the slow class).
VALU_CYC_FAST, VALU_CYC_SLOW = 2.22, 4.16


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="BASELINE configuration: sets --frames / --scene / --all-contacts "
                    "(c3 = the quoted one; explicit size flags override)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong", help="N > 1: strong = ONE clip sharded over the ranks with the "
                    "per-iteration halo exchange (BASELINE config 3's arrangement); weak = one whole clip per rank, no data-path collective")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs block (configs 2 and 5 measured in the same run)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--scene", type=int, default=None)
    ap.add_argument("--contacts-per-leg", type=int, default=250)
    ap.add_argument("--all-contacts", action="store_true", help="every mesh vertex is a contact vertex (BASELINE config 5)")
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--verts", type=int, default=10475)
    ap.add_argument("--lbs-nnz", type=int, default=4, help="non-zero skinning weights per vertex of the synthetic body model "
                    "(4 = SURVEY 8d's spec; a real SMPLX_NEUTRAL.npz is not promised to be 4-sparse: 8 / 12 run the wider packed skinning forms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-logging-run", action="store_true", help="skip the secondary run that evaluates every loss term every iteration")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the child run with FDCAP_GEMM_SPLIT3=0")
    ap.add_argument("--value-only", action="store_true", help="timed steps only: no roofline / secondary figures (child runs)")
    ap.add_argument("--profile-logging", action="store_true", help="profiling aid, with --value-only: the timed steps are the "
                    "every-iteration-logging step (the line says so in config.workload; never the headline)")
    ap.add_argument("--cpu-sample-frames", type=int, default=0, help="0 = pick from a ~20 s budget")
    ap.add_argument("--dry-run", action="store_true", help="plumbing only, on CPU with gloo (see the module docstring)")
    args = ap.parse_args(argv)
    f, ns, allc = CONFIGS[args.config]
    if args.frames is None:
        args.frames = f
    if args.scene is None:
        args.scene = ns
    args.all_contacts = args.all_contacts or allc
    return args


def which_config(N, ns, nc, verts, lbs_nnz):
    """name of the BASELINE configuration these sizes are (None: none of them)"""
    for name, (f, s_, allc) in CONFIGS.items():
        if (N, ns) == (f, s_) and nc == (verts if allc else 500) and verts == 10475 and lbs_nnz == 4:
            return name
    return None


# ---- multi-process plumbing (shared by the real run and --dry-run) -----------------------------------------------------
def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N rank processes here --
    the same `python -m torch.distributed.run` line the driver uses for the scaling run -- as a CHILD of this process, before
    anything in this process has touched a GPU, pass its output through and return its exit code.  (Never exec: a process
    that has initialised the GPU must not be replaced; this one has not, and a child is right either way.)"""
    import socket
    if not args.dry_run:
        have = torch.cuda.device_count()                       # counting devices does not initialise the runtime on this image
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s); one rank per GPU is the only arrangement "
                             f"the metric is defined for (no ranks were started)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # (the host driver only supports dmabuf IPC: RCCL needs it)
    p = subprocess.run(cmd, env=env)
    return p.returncode


class Ranks:
    def __init__(self, dry, gpus=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if gpus is not None and gpus != self.world:
            raise SystemExit(f"bench.py --gpus {gpus} inside a launcher that started WORLD_SIZE={self.world} ranks: the two must agree "
                             f"(n_gpus in the JSON line is the number of ranks that ran)")
        self.dry = dry
        self.group = None
        # FDCAP_FORCE_EXCHANGE=1: a one-rank RCCL group, to measure what the exchange path itself costs
        if self.world > 1 or os.environ.get("FDCAP_FORCE_EXCHANGE") == "1":
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local))
            self.group = dist.group.WORLD
            assert dist.get_world_size() == self.world, (dist.get_world_size(), self.world)

    def barrier(self):
        if not self.dry:
            torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        if not self.dry:
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        import torch.distributed as dist
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if self.dry else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, steps):
        """EXACTLY `steps` calls of fn between two barriers; the slowest rank's wall time."""
        self.barrier()
        t0 = time.perf_counter()
        res = None
        for _ in range(steps):
            res = fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), res

    def finish(self, out):
        if self.group is not None:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        if self.rank == 0:
            # librccl writes its version banner through C stdio (block-buffered when stdout is a pipe): push it out first
            # so the JSON line is the last thing this process prints
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            print(json.dumps(out), flush=True)


def base_line(args, rk, nc, value, dt):
    N, ns = args.frames, args.scene
    cfg = which_config(N, ns, nc, args.verts, args.lbs_nnz) if args.iters == 500 else None
    weak = args.scaling == "weak"
    return {"metric": METRIC, "value": value, "unit": "frames/s", "n_gpus": rk.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 values and fp32 accumulation throughout; dense products on the 16-bit matrix cores as two-plane fp16 splits of "
                          "the power-of-two-scaled fp32 operands, three products per term (measured error below the fp32 MFMA chain's: "
                          "tools/panel_error_probe.py; FDCAP_GEMM_SPLIT3=0: v_mfma_f32 chains, see exact_fp32)",
            "config": {"workload": f"{CONFIG_NAMES[cfg] if cfg else 'non-default sizes (not a BASELINE configuration)'}: {N}-frame clip, "
                                   f"{ns}-pt scene, {nc} contact verts, {args.iters} Adam iterations (phase split 0.8), full loss; "
                                   + (f"one whole clip per GPU, {rk.world} GPU(s), no data-path collective" if weak else
                                      f"frames sharded over {rk.world} GPU(s)"),
                       "name": cfg,
                       "frames": N, "scene_points": ns, "contact_verts": nc, "iters": args.iters, "body_verts": args.verts,
                       "lbs_weights_per_vertex": args.lbs_nnz,
                       "frame_iterations_per_s": None if value is None else value * args.iters}}


def dry_run(args):
    """The plumbing of main() without a GPU: same rendezvous variables, shards, collectives per iteration and JSON rules."""
    import fdcap_amd  # noqa: F401
    from fdcap_amd.dist import FrameShard, allgather_packed
    rk = Ranks(dry=True, gpus=args.gpus)
    shard = FrameShard(args.frames, rk.group, rank=rk.rank, world=rk.world)
    xl = 4 * (78 + 16) + 8                                   # fdcap_exchange_len(): the iteration's one message
    send, gathered = torch.zeros(xl), torch.zeros(rk.world, xl)

    def one_step():
        for ii in range(min(args.iters, 5)):
            send.fill_(float(rk.rank * 1000 + ii))
            if rk.world > 1:
                allgather_packed(shard, send, gathered)
                assert [float(gathered[r, 0]) for r in range(rk.world)] == [r * 1000.0 + ii for r in range(rk.world)]
        return shard.n_local

    for _ in range(args.warmup):
        one_step()
    dt, _ = rk.timed(one_step, args.steps)
    mine = {"rank": rk.rank, "local_rank": rk.local, "device": f"cuda:{rk.local}", "frames": [shard.frame0, shard.frame0 + shard.n_local]}
    ranks = [mine]
    if rk.world > 1:
        import torch.distributed as dist
        ranks = [None] * rk.world
        dist.all_gather_object(ranks, mine)
    out = base_line(args, rk, 2 * args.contacts_per_leg, None, dt)
    out.update({"dry_run": True, "ranks": ranks, "backend": "gloo"})
    rk.finish(out)


# ---- CPU baseline --------------------------------------------------------------------------------------------------------
def cpu_baseline(bm, vp, clip, scene, vid, args):
    """The oracle (PyTorch-CPU restatement of cal_loss + Adam, golden-checked against the reference's own loop) on a bounded
    sample of the same workload: the first F frames of the clip against the FULL scene; 1 warm-up + 3 timed iterations of
    each phase, weighted by the fixed budget's 400 / 100 split."""
    from oracle import rotrepr
    from oracle.chamfer import nn_direct
    from oracle.fitting import FittingOracle, reference_host_loop_overhead
    from oracle.smplx import SMPLXOracle
    from oracle.vposer import VPoserDecoder
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    smpl, vpo = SMPLXOracle(bm), VPoserDecoder.from_data(vp)
    # torch's intra-op pool degrades badly when it is wider than what these tensor sizes can use (256 threads: 274 s per
    # iteration against 2-4 s at 16-32): pick the fastest width on the dominant op (one frame's contact set vs 64k scene points)
    qs = torch.randn(len(vid), 3)
    ss = torch.tensor(scene[:65536])
    best_t, cores = None, 1
    for cand in [c for c in (4, 8, 16, 32, 64) if c <= avail] or [avail]:
        torch.set_num_threads(cand)
        nn_direct(qs, ss)
        t0 = time.perf_counter()
        nn_direct(qs, ss)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, cores = dt, cand
    torch.set_num_threads(cores)
    P = int(np.ceil(args.iters * 0.8 - 1e-12))

    def run(F, timed):
        f = FittingOracle(smpl, vpo, scene, vid, clip.camerapose_lines[:F], F, num_iter=args.iters)
        x78 = rotrepr.convert_to_6D_rot(torch.tensor(clip.body_params[:F]))
        idx1 = f.init(x78)
        x78 = x78.detach()
        t1, t2 = [], []
        for k in range(timed + 1):                             # phase 1 (ii < 0.8 num_iter): contact + smoothing + rec
            t0 = time.perf_counter()
            f.step(k, x78, idx1)
            t1.append(time.perf_counter() - t0)
        for k in range(timed + 1):                             # phase 2: rec + world smoothing + smoothing (Chamfer still evaluated: it is printed)
            t0 = time.perf_counter()
            f.step(P + k, x78, idx1)
            t2.append(time.perf_counter() - t0)
        with torch.no_grad():                                  # share of the Chamfer forward (as cal_loss issues it) in an iteration
            from oracle.chamfer import chamferDist
            _, verts, _ = f.forward_world()
            contact = verts[:, vid, :].contiguous()
            t0 = time.perf_counter()
            chamferDist(True)(contact, f.s_verts_batch)
            t_nn = time.perf_counter() - t0
        return float(np.mean(t1[1:])), float(np.mean(t2[1:])), t_nn

    F = args.cpu_sample_frames
    if F <= 0:
        p1, p2, _ = run(4, 1)
        per_frame = (p1 + p2) / 2.0 / 4.0                      # measured seconds per frame-iteration
        F = int(max(2, min(args.frames, 12, 15.0 / (8 * max(per_frame, 1e-6)))))   # (2 phases x 4 iterations; bounded: the host is shared)
    t1, t2, t_nn = run(F, 3)
    n1, n2 = P, args.iters - P
    fps = F / (n1 * t1 + n2 * t2)
    ov = reference_host_loop_overhead(min(args.frames, 300))
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32, {cores} threads = fastest of 4..64 on {avail} usable cores): first {F} of the clip's "
                      f"{args.frames} frames vs the full {len(scene)}-pt scene; 3 timed iterations after 1 warm-up in EACH phase "
                      f"(phase 1 {t1 * 1e3:.0f} ms/iter, phase 2 {t2 * 1e3:.0f} ms/iter), extrapolated to the fixed budget "
                      f"{n1} x phase 1 + {n2} x phase 2; the Chamfer forward alone is {t_nn * 1e3:.0f} ms of an iteration "
                      f"({100 * t_nn / max(t1, 1e-9):.0f} % of phase 1): the baseline is a Chamfer benchmark",
            "sample_frames": F, "timed_iterations_per_phase": 3, "ms_per_iter_phase1": t1 * 1e3, "ms_per_iter_phase2": t2 * 1e3,
            "chamfer_forward_ms": t_nn * 1e3,
            "reference_python_loop_overhead": {
                "frames": min(args.frames, 300), "ms_per_iter_reference_loops": ov["reference_loops"] * 1e3,
                "ms_per_iter_vectorised": ov["vectorised"] * 1e3,
                "note": "forward + backward of the two host loops every reference iteration runs (per-frame body2world, "
                        "global_optimization.py:191-206; 23 x 3 x W cal_dctloss, :232-246, result unused in mode 'global') around a trivial "
                        "stand-in body, against the vectorised forms the oracle (and `value` above) uses; not included in `value`"}}


# ---- roofline from live durations + the committed PMC summary ---------------------------------------------------------
def counter_fracs(k, live_seconds=None):
    """What a kernel's PMC totals (summed over the device, per launch) say about the resources it uses.
    GRBM_GUI_ACTIVE is summed over the 8 XCDs -> /8 = busy cycles of the launch; SQ_WAVE_CYCLES counts quad-cycles."""
    if not k or not k.get("GRBM_GUI_ACTIVE"):
        return None
    cyc = k["GRBM_GUI_ACTIVE"] / NUM_XCD
    dur = k["duration_us_under_pmc"] * 1e-6
    out = {"clock_ghz_under_pmc": cyc / dur / 1e9, "duration_us_under_pmc": k["duration_us_under_pmc"], "dispatches": k.get("dispatches")}
    if out["clock_ghz_under_pmc"] > 2.45:
        # GRBM_GUI_ACTIVE also counts the front end's activity around a launch: for kernels of a few microseconds the implied
        # clock exceeds the chip's 2.4 GHz maximum.  Then the launch's cycles are taken as duration x 2.4 GHz -- the most the
        # chip can have run -- so every cycle-normalised fraction below is a LOWER bound, and says so.
        cyc = dur * 2.4e9
        out["clock_ghz_under_pmc"] = 2.4
        out["note"] = ("launch too short for GRBM_GUI_ACTIVE to give its clock: cycles = duration x the 2.4 GHz maximum, "
                       "cycle-normalised fractions are lower bounds")
    if k.get("SQ_INSTS_VALU") is not None:
        out["valu_insts_per_launch"] = k["SQ_INSTS_VALU"]
        mix = [k.get(n) for n in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_INT32")]
        if all(v is not None for v in mix) and k.get("SQ_INSTS_MFMA") is not None:
            fast = float(sum(mix))
            slow = k["SQ_INSTS_VALU"] - fast - k["SQ_INSTS_MFMA"]          # (SQ_INSTS_VALU counts the MFMAs too; they issue on the matrix pipe)
            busy = fast * VALU_CYC_FAST + slow * VALU_CYC_SLOW
            out.update({"valu_busy_simd_cycles_per_launch": busy, "valu_issue_frac": busy / (NUM_SIMD * cyc),
                        "valu_mix": {"fast_class_2.2cyc": fast, "slow_class_4.16cyc": slow, "mfma": k["SQ_INSTS_MFMA"]}})
        else:
            # no instruction-mix pass: bounds only (every instruction in the fast / in the slow class)
            out["valu_issue_frac_bounds"] = [k["SQ_INSTS_VALU"] * VALU_CYC_FAST / (NUM_SIMD * cyc), k["SQ_INSTS_VALU"] * VALU_CYC_SLOW / (NUM_SIMD * cyc)]
        if k.get("SQ_ACTIVE_INST_ANY") is not None and k.get("SQ_WAVE_CYCLES"):
            wc = k["SQ_WAVE_CYCLES"]
            out["wave_time_split"] = {"issuing": k["SQ_ACTIVE_INST_ANY"] / wc, "waiting_to_issue": (k.get("SQ_WAIT_INST_ANY") or 0) / wc,
                                      "waiting_on_waitcnt": (k.get("SQ_WAIT_ANY") or 0) / wc}
    if k.get("SQ_WAVES"):
        out["waves_per_launch"] = k["SQ_WAVES"]
        if k.get("SQ_INSTS_VALU") is not None:
            out["valu_insts_per_wave"] = k["SQ_INSTS_VALU"] / k["SQ_WAVES"]
    if k.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        out["mfma_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (NUM_SIMD * cyc)
    if k.get("SQ_WAVE_CYCLES") is not None:
        out["mean_waves_per_simd"] = k["SQ_WAVE_CYCLES"] * 4.0 / (NUM_SIMD * cyc)
        out["max_waves_per_simd"] = WAVES_PER_SIMD
    if k.get("hbm_bytes") is not None:
        out["hbm_bytes_per_launch"] = k["hbm_bytes"]
        out["hbm_frac_on_counter_bytes"] = k["hbm_bytes"] / (live_seconds or dur) / 1e9 / HBM_PEAK_GBS
    return out


def load_pmc(cfg):
    """the committed PMC summary of a BASELINE configuration (tools/run_prof_r5.sh): {} when there is none"""
    path = pmc_summary_path(cfg) if cfg else None
    if path and os.path.exists(path):
        return json.load(open(path)), os.path.relpath(path, ROOT)
    return {}, None


def build_problem(frames, scene_pts, all_contacts, verts, lbs_nnz, contacts_per_leg, iters, group):
    import fdcap_amd  # noqa: F401
    from fdcap_amd import synth
    from fdcap_amd.fitting import FittingOP
    from fdcap_amd.io import read_camerapose
    bm = synth.make_body_model(verts, seed=0, lbs_nnz=lbs_nnz)
    vp = synth.make_vposer(seed=1)
    clip = synth.make_clip(frames, seed=3)
    scene = synth.make_scene(scene_pts, seed=2)
    left, right = synth.make_contact_ids(bm.v_template, per_part=contacts_per_leg, seed=4)
    vid = np.arange(verts) if all_contacts else np.concatenate([left, right])
    fop = FittingOP({"num_iter": iters}, {}, frames, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines), group=group)
    return fop, torch.tensor(clip.body_params).cuda(), bm, vp, clip, scene, vid


def setup_block(fop, first_fit_s, steady_fit_s, frames):
    """What a one-clip process pays around its fit: host seconds of the three registration calls (model constants,
    scene, contact set -- each returns with its device work done; r6: the tables and panels are built by device kernels), and
    what the FIRST fit costs beyond a steady one (fdcap_opt_create's allocations, the seeding launch nn_seed_kernel, first
    launches of every kernel).  None of it is inside `value` (SURVEY §8d excludes model / scene upload); it is published so the
    line also says what the reference's usage -- one process, one clip, one scene (:655-714) -- would see."""
    s = dict(fop.setup_s)
    tot = sum(s.values())
    out = {"ctx_create_s": s.get("ctx_create"), "set_scene_s": s.get("set_scene"), "set_contact_ids_s": s.get("set_contact_ids"),
           "registration_total_s": tot, "first_fit_s": first_fit_s, "steady_fit_s": steady_fit_s,
           "first_fit_extra_s": None if first_fit_s is None else first_fit_s - steady_fit_s}
    if first_fit_s is not None:
        out["end_to_end_frames_per_s"] = frames / (tot + first_fit_s)
        out["end_to_end_note"] = "frames / (registration + the process's FIRST fit): a one-clip process, model and scene arrays already in host memory"
    return out


def cli_end_to_end(args):
    """The reference's real clip length (300 frames, :41-42, utils/split_frames.py:21-34) against the quoted scene, as ONE process
    would see it: registration + one fit, nothing warmed up but the HIP runtime and the code objects (this process has run fits)."""
    t0 = time.perf_counter()
    fop, body_gpu, *_ = build_problem(300, args.scene, args.all_contacts, args.verts, args.lbs_nnz, args.contacts_per_leg, args.iters, None)
    t_build = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b, sc, cam = fop.fitting(body_gpu, "global")
    b, cam = b.cpu(), cam.cpu()
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    reg = sum(fop.setup_s.values())
    fop.close()
    return {"frames": 300, "scene_points": args.scene, "registration_s": reg, "one_fit_s": t_fit, "value": 300 / (reg + t_fit), "unit": "frames/s",
            "synthetic_inputs_generated_in_s": t_build - reg,
            "note": "300-frame clip (the reference's clip length), 500 iterations: frames / (fdcap_ctx_create + fdcap_set_scene + "
                    "fdcap_set_contact_ids + one fit incl. fdcap_opt_create and the seeding launch); file I/O and the synthetic generator excluded"}


def time_nn_launches(fop, one_step, iters):
    """HIP events around every in-loop Chamfer launch of one more (untimed) fit + the two stand-alone timings"""
    import ctypes
    from fdcap_amd import capi
    lib, h = fop.ctx.lib, fop.ctx.handle
    ms_inloop, n_inloop = ctypes.c_float(0), ctypes.c_int32(0)
    capi.check(lib.fdcap_opt_nn_timing(h, iters), "fdcap_opt_nn_timing")
    one_step()
    capi.check(lib.fdcap_opt_nn_timing_read(h, ctypes.byref(ms_inloop), ctypes.byref(n_inloop)), "fdcap_opt_nn_timing_read")
    capi.check(lib.fdcap_opt_nn_timing(h, 0), "fdcap_opt_nn_timing")
    ms_loop = ctypes.c_float(0)
    capi.check(lib.fdcap_opt_time_chamfer(h, 10, 0, ctypes.byref(ms_loop), capi.current_stream()), "fdcap_opt_time_chamfer")
    return ms_inloop.value, n_inloop.value, ms_loop.value


LT_STAGES = ("vposer_fwd", "pose_fwd", "contact_fwd", "nn_in_loop_all", "skin_bwd", "blend_bwd", "pose_bwd", "vposer_bwd")   # include/fdcap.h FDCAP_LT_*


def time_all_launches(fop, one_step, iters, steady_fit_s=None):
    """One more (untimed) fit with a HIP event at every boundary between two launches of an iteration (fdcap_opt_launch_timing):
    mean microseconds per launch and phase, measured on THIS box in THIS run (the table's microseconds used to be
    copies from the committed trace)."""
    import ctypes
    from fdcap_amd import capi
    lib, h = fop.ctx.lib, fop.ctx.handle
    n = len(LT_STAGES)
    capi.check(lib.fdcap_opt_launch_timing(h, 10 * iters + 16), "fdcap_opt_launch_timing")
    one_step()
    us = (ctypes.c_float * (2 * n))()
    cnt = (ctypes.c_int32 * (2 * n))()
    capi.check(lib.fdcap_opt_launch_timing_read(h, us, cnt), "fdcap_opt_launch_timing_read")
    capi.check(lib.fdcap_opt_launch_timing(h, 0), "fdcap_opt_launch_timing")
    out = {}
    for ph in (0, 1):
        d = {LT_STAGES[i]: {"us": float(us[ph * n + i]), "launches": int(cnt[ph * n + i])} for i in range(n) if cnt[ph * n + i]}
        out["phase1" if ph == 0 else "phase2"] = d
    if steady_fit_s:
        # the events are not free: kernels run back to back (the host is ahead), so the sum of all event-to-event intervals of the
        # fit minus a plain fit's wall time, per interval, is what one event adds to the interval it closes
        tot = sum(v["us"] * v["launches"] for d in out.values() for v in d.values())
        nint = sum(v["launches"] for d in out.values() for v in d.values())
        ov = max(0.0, (tot - steady_fit_s * 1e6) / max(nint, 1))
        out["event_overhead_us_per_launch"] = ov
        out["event_overhead_note"] = ("(sum of all event-to-event intervals of the instrumented fit - wall time of a plain fit) / intervals: what "
                                      "recording an event adds to the interval it closes; us_live_corrected = us - this")
        for d in (out["phase1"], out["phase2"]):
            for v in d.values():
                v["us_corrected"] = v["us"] - ov
    return out


def nn_roofline(pk, src, sec_loop, ms_steady, n_timed, alg_bytes):
    """The in-loop launch is an exact PRUNED search (seeds, k-d cells, kept work lists; bit-identical to the full scan): it touches
    ~1 % of the pairs the algorithmic byte count pays for, so bytes-over-time says nothing about a hardware limit.  What bounds it
    is VALU issue while the machine is full, then a drain at part occupancy: the top-level fraction is that
    resource's, from the PMC counters; `contract` keeps the §8d figure."""
    nn = counter_fracs(pk.get("nn_in_loop_all") or pk.get("nn_in_loop"), sec_loop) if pk else None
    roofline = {
        "kernel": "fdc::nn_stream4_kernel (Chamfer body->scene NN forward as issued in the loop: seeded + k-d-cell-culled exact scan, "
                  "kept work lists, bf16-split MFMA filter + fp32 re-evaluation)",
        "ms_per_launch": sec_loop * 1e3, "launches_timed": n_timed, "steady_state_ms_per_launch": ms_steady,
        "timing": "HIP events around every NN launch of one whole fit, on its launch stream (mean); steady_state = back-to-back "
                  "launches at the converged state",
        "counters_from": src if nn else None}
    if nn and nn.get("valu_busy_simd_cycles_per_launch"):
        clock = nn["clock_ghz_under_pmc"] * 1e9
        ach = nn["valu_busy_simd_cycles_per_launch"] / sec_loop / 1e9
        peak = NUM_SIMD * clock / 1e9
        roofline.update({"bound": "valu_issue", "achieved": ach, "peak": peak, "unit": "G SIMD-cycles/s of VALU issue", "frac": ach / peak,
                         "frac_under_counters": nn.get("valu_issue_frac"),        # the same busy cycles over the launch's own cycles in the PMC pass
                         "us_per_launch_under_counters": nn.get("duration_us_under_pmc"),
                         "traffic": nn.get("hbm_bytes_per_launch"), "hbm_frac_on_counter_bytes": nn.get("hbm_frac_on_counter_bytes"),
                         "mfma_busy_frac": nn.get("mfma_busy_frac"), "mean_waves_per_simd": nn.get("mean_waves_per_simd"),
                         "max_waves_per_simd": WAVES_PER_SIMD, "valu_insts_per_wave": nn.get("valu_insts_per_wave"),
                         "valu_mix": nn.get("valu_mix"), "wave_time_split": nn.get("wave_time_split"),
                         "valu_cycles_per_instruction": {"fast_class": VALU_CYC_FAST, "slow_class": VALU_CYC_SLOW,
                                                         "measured_by": "tools/valu_issue_probe.hip -> profiles/r4_valu_issue_probe.txt"},
                         "clock_ghz_under_pmc": nn["clock_ghz_under_pmc"],
                         "note": "achieved = cycles one SIMD's VALU port is occupied per launch (instruction counts of the PMC mix pass x the measured "
                                 "cycles of their class, summed over the SIMDs) / live launch time; peak = 1024 SIMDs x the clock the launch "
                                 "sustained in the PMC pass.  frac is the mean over the whole launch: the port is saturated while all 8 wave "
                                 "slots per SIMD are filled and idles through the drain (mean_waves_per_simd); HBM and the matrix pipe are far "
                                 "from their limits"})
        wts = nn.get("wave_time_split") or {}
        roofline["reading"] = (
            f"VALU issue {roofline['frac']:.2f} of the SIMDs' cycles over the live launch time; {nn.get('mean_waves_per_simd', 0):.1f} of {WAVES_PER_SIMD} wave slots filled "
            f"on average; a wave's life: {wts.get('issuing', 0):.0%} issuing, {wts.get('waiting_to_issue', 0):.0%} waiting to issue, "
            f"{wts.get('waiting_on_waitcnt', 0):.0%} at s_waitcnt; matrix pipe busy {nn.get('mfma_busy_frac') or 0:.2f}; HBM {nn.get('hbm_frac_on_counter_bytes') or 0:.2f} on counter bytes; "
            f"clock under the counters {nn['clock_ghz_under_pmc']:.2f} GHz.  " +
            ("VALU-saturated (and, for long launches, at the chip's power limit): only fewer executed instructions help."
             if roofline["frac"] > 0.85 else
             "Latency- and issue-mix-bound: the port is saturated while all wave slots are filled and idles through the launch's drain."))
        if roofline["frac"] > 0.97:
            roofline["frac_note"] = ("saturated: the instruction prices are measured means (+-3 %) and the counters come from another run of the same "
                                     "configuration, so a VALU-bound launch can read a few per cent above 1; long launches of this configuration also run "
                                     "below the 2.4 GHz maximum (clock_ghz_under_pmc: the chip's power limit), which `peak` reflects")
    else:
        roofline.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                         "note": "no committed PMC summary for these sizes (profiles/r5_<config>_pmc_summary.json exist for BASELINE configs 3, 5 "
                                 "and 2 at 500 iterations on one GPU): only the live launch time and the contract figure below"})
    roofline["contract"] = {
        "bound": "hbm", "algorithmic_bytes": alg_bytes, "bytes_per_unit": "F * (12 Ns + 20 Nc): the reference op re-reads a scene copy per frame",
        "achieved": alg_bytes / sec_loop / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac_on_algorithmic_bytes": alg_bytes / sec_loop / 1e9 / HBM_PEAK_GBS,
        "note": "SURVEY §8d's contract figure.  > 1 because exact pruning does not read what the byte count pays for -- an algorithmic win, "
                "not a bandwidth measurement"}
    return roofline, nn


def per_kernel_table(pk, F, nc, K, src, live=None):
    """Every launch of a phase-1 iteration next to the peak that binds it.  Microseconds: kernel-trace averages of the committed
    profile (the same command's rocprofv3 --kernel-trace --stats).  Dense products: EXECUTED 16-bit MFMA flops (SQ_INSTS_MFMA of
    the PMC mix pass x 16 384 flop per v_mfma_f32_16x16x32_f16 wave instruction; the fp32-equivalent useful flops beside them)
    against the dense fp16 / bf16 peak.  Per-frame kernels: ALGORITHMIC bytes (what the launch must read and write once; formulas
    in DESIGN §5) against HBM, with the counter traffic (FETCH x 2 + WRITE, guide's gfx950 correction) beside them."""
    if not pk:
        return None
    fl = lambda *dims: 2.0 * F * sum(a * b for a, b in dims)
    rows = [
        # key, what, bound, useful fp32 flops | algorithmic bytes
        ("vposer_fwd", "VPoser decode 32-512-512-126 (A7)", "mfma", fl((32, 512), (512, 512), (512, 126))),
        ("pose_fwd", "6D/PCA -> rotations, joint regression, kinematic chain, world joints (A8 K5-K9, A9)", "hbm", 4.0 * F * 2791),
        ("blend_skin_fwd", "pose+shape blend offsets [F,496]x[496,3Nc] + linear-blend skinning + scale + world transform of the contact set, "
                           "one launch (A8 K6/K8/K10, A10; r5: blend_skin_fwd_kernel)", "mfma", fl((496, 3 * nc))),
        ("blend_fwd", "pose+shape blend offsets of the contact set [F,496]x[496,3Nc] (A8 K6/K8)", "mfma", fl((496, 3 * nc))),
        ("skin_fwd", "linear-blend skinning + scale + world transform of the contact set (A8 K10, A10)", "hbm", F * (24.0 * nc + 2720)),
        ("nn_in_loop_all", "Chamfer NN forward (A12): see the top-level roofline", "valu_issue", None),
        ("skin_bwd", "contact robustifier + skinning backward (A13, A8 K10 bwd)", "hbm", F * ((52.0 + 4.0 * (K > 4)) * nc + 5344)),
        ("blend_bwd", "data gradient of the blend product [F,3Nc]x[3Nc,496]", "mfma", fl((3 * nc, 496))),
        ("pose_bwd", "chain / rotation / joint-regression backward + parameter-space losses (A14)", "hbm", 4.0 * F * 3224),
        ("vposer_bwd", "VPoser data gradient 126-512-512-32", "mfma", fl((126, 512), (512, 512), (512, 32))),
    ]
    out, total, total_live = [], 0.0, 0.0
    lv1 = (live or {}).get("phase1", {})
    # the contact forward is one launch at clip size (blend_skin_fwd) and two below: the live event pair spans both either way
    live_key = {"blend_skin_fwd": "contact_fwd", "blend_fwd": "contact_fwd", "skin_fwd": None}
    for key, what, bound, work in rows:
        k = pk.get(key)
        if not k or not k.get("duration_us_trace"):
            continue
        us_trace = k["duration_us_trace"]
        total += us_trace
        lk = live_key.get(key, key)
        us_live = lv1.get(lk, {}).get("us_corrected", lv1.get(lk, {}).get("us")) if lk else None
        if us_live:
            total_live += us_live
        us = us_live or us_trace                                  # fractions below: on the LIVE time when this run measured one
        e = {"kernel": k.get("name"), "computes": what, "us": us, "us_live": us_live, "us_trace": us_trace, "bound": bound,
             "traffic": k.get("hbm_bytes")}
        if key == "blend_fwd" and us_live:
            e["us_live_note"] = "live figure spans this launch AND skin_fwd (one event pair around the contact forward)"
        c = counter_fracs(k) or {}
        if c.get("mfma_busy_frac") is not None:
            e["mfma_busy_frac"] = c["mfma_busy_frac"]
        if c.get("mean_waves_per_simd") is not None:
            e["mean_waves_per_simd"] = c["mean_waves_per_simd"]
        if bound == "mfma":
            ex = k.get("SQ_INSTS_MFMA")
            e.update({"useful_fp32_flop": work, "unit": "TFLOP/s (fp16 MFMA executed, dense)", "peak": BF16_DENSE_TFLOPS})
            if ex is not None:
                e.update({"executed_mfma_flop": ex * 16384.0, "achieved": ex * 16384.0 / us / 1e6, "frac": ex * 16384.0 / us / 1e6 / BF16_DENSE_TFLOPS})
            e["floor_us"] = None if ex is None else ex * 16384.0 / (BF16_DENSE_TFLOPS * 1e6)
        elif bound == "hbm":
            e.update({"algorithmic_bytes": work, "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": work / us / 1e3, "frac": work / us / 1e3 / HBM_PEAK_GBS,
                      "floor_us": work / (HBM_PEAK_GBS * 1e3),
                      "traffic_over_algorithmic": None if not k.get("hbm_bytes") else k["hbm_bytes"] / work})
        else:
            e.update({"frac": None, "note": "priced on VALU issue in the top-level roofline block (live launch time)"})
        out.append(e)
    return {"source": src, "phase1_iteration_us": total_live or total, "phase1_iteration_us_trace": total,
            "phase1_iteration_us_live": total_live or None, "phase2_live": (live or {}).get("phase2"),
            "event_overhead_us_per_launch": (live or {}).get("event_overhead_us_per_launch"), "kernels": out,
            "timing": "us_live: HIP events on the launch stream between consecutive launches of every iteration of one whole (extra, untimed) fit of "
                      "THIS run, mean over the fit's phase-1 iterations, ~1 us of dependent-launch gap included, the events' own cost (event_overhead_us_per_launch, "
                      "estimated from the instrumented fit's total against a plain fit) subtracted; us_trace: rocprofv3 kernel-trace "
                      "average of the committed profile (another box, another day); counters (flops, traffic) always from the committed PMC passes",
            "note": "floor_us = the kernel's work at the binding peak; every launch also pays ~1.2 us fixed + ~0.2 us per MB it leaves dirty in "
                    "L2 (profiles/r4_launch_overhead_probe.txt).  Phase 2 of a fit issues four of them (VPoser and pose, both ways)"}


def other_config(name, args):
    """One more BASELINE configuration in the same run: a fresh optimiser of its sizes, 1 warm-up + 1 timed fit, the Chamfer
    launch timed in the loop, its own committed PMC summary for the bound."""
    f, ns, allc = CONFIGS[name]
    fop, body_gpu, bm, vp, clip, scene, vid = build_problem(f, ns, allc, 10475, 4, 250, args.iters, None)

    def one_step():
        b, sc, cam = fop.fitting(body_gpu, "global")
        return b.cpu(), sc, cam.cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one_step()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert np.isfinite(res[0].numpy()).all()
    ms_in, n_in, ms_steady = time_nn_launches(fop, one_step, args.iters)
    nc = len(vid)
    pmc, src = load_pmc(name if args.iters == 500 else None)
    pk = pmc.get("kernels", {})
    alg = f * (12.0 * ns + 20.0 * nc)
    rl, nn = nn_roofline(pk, src, (ms_in if n_in else ms_steady) * 1e-3, ms_steady, n_in, alg)
    P = int(np.ceil(args.iters * 0.8 - 1e-12))
    out = {"workload": f"{CONFIG_NAMES[name]}: {f}-frame clip, {ns}-pt scene, {nc} contact verts, {args.iters} iterations, 1 GPU",
           "value": f / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "us_per_iteration": dt * 1e6 / args.iters,
           "chamfer_launch_us": rl["ms_per_launch"] * 1e3, "chamfer_launch_steady_us": ms_steady * 1e3,
           "chamfer_share_of_step": rl["ms_per_launch"] * P / (dt * 1e3),
           "bound": rl.get("bound"), "frac": rl.get("frac"), "hbm_frac_on_counter_bytes": rl.get("hbm_frac_on_counter_bytes"),
           "mfma_busy_frac": rl.get("mfma_busy_frac"), "mean_waves_per_simd": rl.get("mean_waves_per_simd"),
           "traffic": rl.get("traffic"), "contract_frac_on_algorithmic_bytes": rl["contract"]["frac_on_algorithmic_bytes"],
           "counters_from": rl.get("counters_from"), "reading": rl.get("reading"), "setup": setup_block(fop, first, dt, f)}
    live = time_all_launches(fop, one_step, args.iters, dt)
    out["launches_live_us"] = live
    pkt = per_kernel_table(pk, f, nc, 4, src, live) if name == "c2" else None
    if pkt:
        out["phase1_iteration_us_by_trace"] = pkt["phase1_iteration_us_trace"]
        out["phase1_iteration_us_live"] = pkt["phase1_iteration_us_live"]
    fop.close()
    del fop
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # no launcher around us: be the launcher (before any GPU call)
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback; --dry-run checks the multi-process plumbing only)")
    rk = Ranks(dry=False, gpus=args.gpus)
    torch.cuda.set_device(rk.local)
    import ctypes
    import fdcap_amd  # noqa: F401
    from fdcap_amd import capi

    N = args.frames
    weak = args.scaling == "weak"
    torch.zeros(1, device="cuda")                          # the HIP runtime's own start-up is nobody's set-up cost: keep it out of setup.ctx_create_s
    torch.cuda.synchronize()
    # weak: every rank owns a whole clip (the same synthetic one) -- the process group only serves the barriers and the max over ranks
    fop, body_gpu, bm, vp, clip, scene, vid = build_problem(N, args.scene, args.all_contacts, args.verts, args.lbs_nnz,
                                                            args.contacts_per_leg, args.iters, None if weak else rk.group)
    if rk.group is not None:                               # create the RCCL communicator outside the timed region
        import torch.distributed as dist
        warm = torch.zeros(8, device="cuda")
        allw = torch.zeros(rk.world, 8, device="cuda")
        dist.all_gather_into_tensor(allw, warm)
        dist.all_reduce(warm)
    torch.cuda.synchronize()

    def one_step(log_every=0):
        body_rec, scale, cam = fop.fitting(body_gpu, "global", log_every=log_every)
        return body_rec.cpu(), scale, cam.cpu()          # D->H of the results is part of the step

    if args.profile_logging and not args.value_only:
        raise SystemExit("--profile-logging goes with --value-only")
    timed_step = (lambda: one_step(log_every=1)) if args.profile_logging else one_step
    first_fit_s = None
    for w in range(args.warmup):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timed_step()
        torch.cuda.synchronize()
        if w == 0:
            first_fit_s = time.perf_counter() - t0           # the process's first fit (setup block)
    dt, res = rk.timed(timed_step, args.steps)
    assert np.isfinite(res[0].numpy()).all()
    nc, ns, nl = len(vid), len(scene), fop.shard.n_local
    clips = rk.world if weak else 1
    out = base_line(args, rk, nc, clips * N * args.steps / dt, dt)
    if rk.world > 1:
        out["config"]["ranks_seen_by_rccl"] = int(torch.distributed.get_world_size())
        if not weak:      # which iteration schedule rank 0's last fit kept after timing both (same results either way)
            out["config"]["exchange_schedule"] = "next forward's head under the all-gather" if getattr(fop, "exchange_overlap", False) else "plain"
            out["config"]["exchange_inside_library"] = bool(getattr(fop, "_c_comm", False))
    if (rk.world > 1 or os.environ.get("FDCAP_FORCE_EXCHANGE") == "1") and not weak and getattr(fop, "_c_comm", False):
        # what one iteration's exchange costs on THIS group (every rank makes the call; rank 0's figures go into the line).  Measured
        # after the timed steps: the call steps the optimiser state, the next fit re-initialises it.
        xus, gus = ctypes.c_float(0), ctypes.c_float(0)
        capi.check(fop.ctx.lib.fdcap_opt_time_exchange(fop.ctx.handle, 300, ctypes.byref(xus), ctypes.byref(gus), capi.current_stream()),
                   "fdcap_opt_time_exchange")
        out["config"]["exchange_us_measured"] = {"whole_tail": float(xus.value), "allgather_alone": float(gus.value), "ranks": rk.world,
                                                 "note": "mean of 300 back-to-back calls between two HIP events on the compute stream"}
    out["setup"] = setup_block(fop, first_fit_s, dt / max(args.steps, 1), N if weak or rk.world == 1 else nl)
    if args.value_only:
        if args.profile_logging:
            out["config"]["workload"] += " [--profile-logging: every loss term of every iteration evaluated and kept]"
        return rk.finish(out)
    # Like-for-like figure (never `value`): the same step with EVERY loss term evaluated in EVERY iteration, as the reference's
    # loop prints them (:573-575, :587-589; phase 2 then also runs the contact forward it otherwise has no use for).  The
    # partial sums go to a device-side history and are read back once, after the last iteration.
    n_log_steps = 0 if args.no_logging_run else min(args.steps, 2)
    dt_log, _ = rk.timed(lambda: one_step(log_every=1), n_log_steps)
    if n_log_steps:
        out["with_reference_logging"] = {
            "value": clips * N * n_log_steps / dt_log, "unit": "frames/s", "ms_per_step": dt_log / n_log_steps * 1e3,
            "note": "same step, every loss term of every iteration evaluated and kept (the reference prints them every iteration); "
                    "`value` evaluates them only where they reach a gradient -- optimised parameters are identical either way"}

    if ns == 0:
        raise SystemExit('bench.py needs a scene (the roofline kernel is the Chamfer NN); BASELINE config 1 is a parity-test case')
    lib, h = fop.ctx.lib, fop.ctx.handle
    # the dominant kernel as the loop issues it: HIP events around every contact forward of one more (untimed) step
    ms_inloop, n_inloop, ms_steady = time_nn_launches(fop, one_step, args.iters)
    ms_bf = ctypes.c_float(0)
    capi.check(lib.fdcap_opt_time_chamfer(h, 3 if nc <= 1000 else 1, 1, ctypes.byref(ms_bf), capi.current_stream()), "fdcap_opt_time_chamfer")
    alg_bytes = nl * (12.0 * ns + 20.0 * nc)              # scene once PER FRAME + queries + dist/idx
    pairs = float(nl) * nc * ns
    sec_bf, sec_loop = ms_bf.value * 1e-3, (ms_inloop if n_inloop else ms_steady) * 1e-3
    # the committed counters describe ONE GPU holding the whole clip at the fixed budget: sharded runs get live times only
    cfg = which_config(N, ns, nc, args.verts, args.lbs_nnz) if (args.iters == 500 and nl == N) else None
    pmc, pmc_src = load_pmc(cfg)
    pk = pmc.get("kernels", {})
    quoted = cfg == "c3"
    roofline, nn = nn_roofline(pk, pmc_src, sec_loop, ms_steady, n_inloop, alg_bytes)
    # (the brute-force launch and the wide GEMM are not in the r5 per-configuration command: their counters come from the passes of
    #  the full bench command, profiles/r5_c3_ops_pmc_summary.json -- r4's summary before that file existed)
    ops_pk = {}
    for name in ("r6_c3_ops_pmc_summary.json", "r5_c3_ops_pmc_summary.json", "r4_pmc_summary.json"):
        fn = os.path.join(ROOT, "profiles", name)
        if quoted and os.path.exists(fn):
            ops_pk = json.load(open(fn)).get("kernels", {})
            break
    bf = counter_fracs(pk.get("nn_bruteforce") or ops_pk.get("nn_bruteforce"), sec_bf) if quoted else None
    roofline["brute_force"] = {
        "kernel": "fdc::nn_mfma_kernel<4> (every (query, scene point) pair visited: the launch the algorithmic byte count describes; "
                  "operator API for foreign targets, not part of the loop)",
        "ms_per_launch": ms_bf.value, "bound": "mfma", "achieved": 32 * pairs / sec_bf / 1e12, "peak": BF16_DENSE_TFLOPS,
        "unit": "TFLOP/s (bf16 MFMA, dense)", "frac": 32 * pairs / sec_bf / 1e12 / BF16_DENSE_TFLOPS, "mfma_flop_per_pair": 32,
        "pairs_per_s": pairs / sec_bf, "mfma_busy_frac": None if not bf else bf.get("mfma_busy_frac"),
        "traffic": None if not bf else bf.get("hbm_bytes_per_launch"),
        "hbm": {"achieved": alg_bytes / sec_bf / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / sec_bf / 1e9 / HBM_PEAK_GBS,
                "on": "algorithmic bytes"}}
    live = time_all_launches(fop, one_step, args.iters, dt / max(args.steps, 1))
    roofline["per_kernel"] = per_kernel_table(pk, nl, nc, args.lbs_nnz, pmc_src, live)
    if roofline["per_kernel"] is None:                      # no committed counters for these sizes: the live times alone
        roofline["per_kernel"] = {"source": None, "launches_live_us": live}
    # the two other readings of the dominant kernel, up where the fraction is: what the counters say it moves
    # against HBM, and SURVEY 8d's contract figure (algorithmic bytes of the every-frame scene copy / launch time: > 1 = pruning)
    roofline["contract_frac"] = roofline["contract"]["frac_on_algorithmic_bytes"]
    roofline.setdefault("hbm_frac_on_counter_bytes", None)
    out["roofline"] = roofline
    out["dtype_note"] = ("values and accumulation fp32; dense products: 2 x fp16 planes of power-of-two-scaled fp32 operands, three MFMA products per "
                         "term, error <= 1e-6 * sum|a||b| vs fp64 (tests/test_gpu_panel.py); strict fp32 products = exact_fp32 below")

    # north-star item: the full-mesh pose + shape blendshape GEMM (body-model operator / output meshes; the loop itself only
    # needs the contact-vertex columns, whose two products are listed under roofline.per_kernel)
    ms_g = ctypes.c_float(0)
    capi.check(lib.fdcap_time_blend_gemm(h, nl, 5, ctypes.byref(ms_g), capi.current_stream()), "fdcap_time_blend_gemm")
    gflop = 2.0 * nl * 496 * 3 * args.verts / 1e9         # operand rows [pose feature 486 | betas 10]
    split3 = os.environ.get("FDCAP_GEMM_SPLIT3", "1") != "0"
    tf = gflop / ms_g.value                                 # useful fp32 multiply-adds, TFLOP/s
    wide = counter_fracs(ops_pk.get("blend_wide")) if quoted else None
    if split3:
        ex = 3.0 * tf * 512.0 / 496.0                       # three fp16 MFMAs per product term (r5; six bf16 ones until then), K padded 496 -> 512
        blend = {"kernel": "fdc::panel_gemm3_wide_kernel<2> (pose + shape blendshapes [F,496] x [496,3V]; fp32 operands as two scaled fp16 parts, "
                           "three v_mfma_f32_16x16x32_f16 per 32 columns, fp32 accumulation, static operand in fragment order)",
                 "ms_per_launch": ms_g.value, "bound": "mfma", "achieved": ex, "peak": BF16_DENSE_TFLOPS, "unit": "TFLOP/s (fp16 MFMA executed, dense)",
                 "frac": ex / BF16_DENSE_TFLOPS, "mfma_busy_frac": None if not wide else wide.get("mfma_busy_frac"),
                 "fp32_equivalent": {"achieved": tf, "unit": "TFLOP/s of fp32 multiply-adds", "fp32_mfma_peak": FP32_MFMA_TFLOPS,
                                     "ratio_to_fp32_mfma_peak": tf / FP32_MFMA_TFLOPS,
                                     "note": "the product's useful work against the pipe an exact-fp32 chain would run on (not a fraction of "
                                             "the pipe that executes it)"}}
    else:
        blend = {"kernel": "fdc::panel_gemm_wide_kernel<2> (pose + shape blendshapes [F,496] x [496,3V], v_mfma_f32_16x16x4_f32, static operand "
                           "in MFMA fragment order)",
                 "ms_per_launch": ms_g.value, "bound": "mfma", "achieved": tf, "peak": FP32_MFMA_TFLOPS, "unit": "TFLOP/s (fp32 MFMA)",
                 "frac": tf / FP32_MFMA_TFLOPS, "mfma_busy_frac": None if not wide else wide.get("mfma_busy_frac")}
    out["blendshape_gemm"] = blend

    single = rk.world == 1
    if single and split3 and not args.no_exact_fp32:
        # the same step on exact fp32 MFMA chains: the switch is read once per process, so a child process runs it
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 3)), "--warmup", "1", "--value-only",
               "--frames", str(N), "--scene", str(ns), "--iters", str(args.iters), "--verts", str(args.verts),
               "--contacts-per-leg", str(args.contacts_per_leg), "--lbs-nnz", str(args.lbs_nnz)] + (["--all-contacts"] if args.all_contacts else [])
        try:
            p = subprocess.run(cmd, env=dict(os.environ, FDCAP_GEMM_SPLIT3="0"), capture_output=True, text=True, timeout=600)
            child = json.loads(p.stdout.strip().splitlines()[-1])
            out["exact_fp32"] = {"value": child["value"], "unit": "frames/s", "ms_per_step": child["ms_per_step"], "steps": child["steps"],
                                 "note": "FDCAP_GEMM_SPLIT3=0: every dense product as a v_mfma_f32 chain (child process, same workload)"}
        except Exception as e:  # noqa: BLE001 -- a secondary figure must not take the line down
            out["exact_fp32"] = {"value": None, "error": repr(e)[:200]}
    if single and quoted and not args.no_other_configs:
        # the other one-GPU configurations of BASELINE.json, measured here so that the driver's one line carries them
        fop.close()
        torch.cuda.empty_cache()
        oc = {}
        for name in ("c2", "c5"):
            try:
                oc[name] = other_config(name, args)
            except Exception as e:  # noqa: BLE001 -- secondary figures must not take the line down
                oc[name] = {"error": repr(e)[:300]}
        out["other_configs"] = oc
        try:
            out["setup"]["cli_end_to_end"] = cli_end_to_end(args)
        except Exception as e:  # noqa: BLE001
            out["setup"]["cli_end_to_end"] = {"error": repr(e)[:300]}
    if single and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(bm, vp, clip, scene, vid, args)
    rk.finish(out if rk.rank == 0 else None)


if __name__ == "__main__":
    main()
