"""bench.py's multi-process plumbing on CPU: the driver launches the 8-GPU scaling run as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`;
`--dry-run` takes exactly that path with gloo in place of RCCL and no kernels (value null), so rendezvous variables, the frame
partition, the per-iteration all-gather, barrier placement, the max over ranks and the rank-0-only JSON line are covered here."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except json.JSONDecodeError:
                pass
    return out


def test_bench_under_torchrun_two_ranks_prints_one_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout                       # rank 0 only
    d = lines[0]
    assert p.stdout.strip().splitlines()[-1].strip().startswith("{")       # ... and it is the last thing printed
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True and d["value"] is None
    assert d["scaling"] == "strong" and d["higher_is_better"] is True and d["unit"] == "frames/s"
    assert d["metric"].startswith("frames/sec global-opt")
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert [r["device"] for r in ranks] == ["cuda:0", "cuda:1"]            # one device per process, from LOCAL_RANK
    assert ranks[0]["frames"] == [0, 512] and ranks[1]["frames"] == [512, 1024]
    assert d["ms_per_step"] > 0


def test_plain_bench_with_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2 ...` with no launcher around it (the shape of the driver's 1-GPU command line with another N):
    the file starts its own ranks; n_gpus is the number of ranks that ran, not the flag echoed back."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["dry_run"] is True
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["device"] for r in ranks] == ["cuda:0", "cuda:1"]
    assert ranks[0]["frames"] == [0, 512] and ranks[1]["frames"] == [512, 1024]


def test_gpus_flag_must_agree_with_the_launcher():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert p.returncode != 0 and "must agree" in (p.stderr + p.stdout)
    assert not _json_lines(p.stdout)


def test_more_gpus_than_the_node_has_is_refused_before_any_rank_starts():
    import torch
    if torch.cuda.device_count() >= 64:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "no ranks were started" in (p.stderr + p.stdout)


def test_bench_dry_run_single_process():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=120,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["ranks"][0]["frames"] == [0, 1024]


def test_bench_refuses_to_run_the_product_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "needs a GPU" in (p.stderr + p.stdout)
    assert not _json_lines(p.stdout)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config_presets_and_the_blocks_built_from_the_committed_profiles():
    """r5: `--config` presets, and the two blocks of the line that are arithmetic on COMMITTED files (no GPU needed to check them):
    `roofline.per_kernel` from profiles/r6_c3_pmc_summary.json (r5's until r6) -- every launch of a phase-1 iteration (eight since the blend forward and
    the skinning forward are one launch), each with its time, its
    work and a fraction of its binding peak in (0, 1) -- and the counter fractions of configs 5 and 2 (`other_configs`)."""
    b = _bench_module()
    a = b.parse(["--config", "c5"])
    assert (a.frames, a.scene, a.all_contacts, a.scaling) == (512, 2_000_000, True, "strong")
    a = b.parse(["--config", "c2", "--scaling", "weak", "--frames", "64"])
    assert (a.frames, a.scene, a.all_contacts, a.scaling) == (64, 100_000, False, "weak")
    assert b.which_config(1024, 500_000, 500, 10475, 4) == "c3" and b.which_config(512, 2_000_000, 10475, 10475, 4) == "c5"
    assert b.which_config(1024, 500_000, 500, 10475, 8) is None and b.which_config(1000, 500_000, 500, 10475, 4) is None
    pmc, src = b.load_pmc("c3")
    assert src == os.path.join("profiles", "r6_c3_pmc_summary.json")          # (the newest committed round)
    t = b.per_kernel_table(pmc["kernels"], 1024, 500, 4, src)
    ks = t["kernels"]
    assert len(ks) == 8 and abs(sum(k["us"] for k in ks) - t["phase1_iteration_us"]) < 1e-6
    assert all(k["us_live"] is None and k["us"] == k["us_trace"] for k in ks)       # no live times without a GPU: the trace's
    # r6: with live times (fdcap_opt_launch_timing) the table's microseconds and fractions are the LIVE ones, the trace's stay beside them
    live = {"phase1": {n: {"us": 10.0 + i, "us_corrected": 8.0 + i, "launches": 400} for i, n in enumerate(b.LT_STAGES)}, "phase2": {},
            "event_overhead_us_per_launch": 2.0}
    tl = b.per_kernel_table(pmc["kernels"], 1024, 500, 4, src, live)
    assert [k["us_live"] for k in tl["kernels"]] == [8.0 + i for i in range(8)] and all(k["us"] == k["us_live"] for k in tl["kernels"])
    assert abs(tl["phase1_iteration_us_live"] - sum(8.0 + i for i in range(8))) < 1e-9 and tl["phase1_iteration_us_trace"] == t["phase1_iteration_us"]
    assert 120.0 < t["phase1_iteration_us"] < 170.0
    assert [k["bound"] for k in ks] == ["mfma", "hbm", "mfma", "valu_issue", "hbm", "mfma", "hbm", "mfma"]
    for k in ks:
        if k["bound"] != "valu_issue":
            assert 0.0 < k["frac"] < 1.0 and 0.0 < k["floor_us"] < k["us"], k
    rl, _ = b.nn_roofline(pmc["kernels"], src, 60e-6, 0.052, 400, 1024 * (12.0 * 500_000 + 20.0 * 500))
    assert rl["bound"] == "valu_issue" and 0.3 < rl["frac"] < 1.0 and rl["contract"]["frac_on_algorithmic_bytes"] > 1.0
    for cfg, t_launch in (("c5", 0.9e-3), ("c2", 18e-6)):
        pk, s2 = b.load_pmc(cfg)
        rl, nn = b.nn_roofline(pk["kernels"], s2, t_launch, t_launch * 1e3, 400, 1.0)
        assert rl["bound"] == "valu_issue" and 0.1 < rl["frac"] < 1.0 and 0.0 < rl["hbm_frac_on_counter_bytes"] < 1.0, (cfg, rl)
    # (config 2's launch is too short for the clock counter: its fractions are lower bounds and say so)
    assert "lower bounds" in (nn.get("note") or "")
