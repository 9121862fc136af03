"""Frame sharding and the per-iteration exchange (fdcap_amd/dist.py) on CPU with the gloo backend, world sizes 2, 3 and 8:
the sharded optimisation -- host harness standing in for the kernels, SAME schedule as fitting.FittingOP.fitting runs on the
GPUs: backward -> Adam on the owned rows -> pack [boundary rows | d loss / d scale partial] -> ONE all-gather -> unpack the
halo rows, sum the partials in rank order, Adam on `scale` -- must reproduce the single-rank run.  The message layout used
here (tests/host_pipeline.py xch_pack / xch_unpack) is checked against the kernels' own message in
tests/test_gpu_sharded.py::test_exchange_message_layout_matches_the_host_mirror."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fdcap_amd  # noqa: F401
from fdcap_amd import synth
from fdcap_amd.dist import FrameShard, allgather_packed, allreduce_scalars, exchange_halos
from fdcap_amd.fitting import find_outliers, first_phase2_iter
from fdcap_amd.io import read_camerapose
from tests.host_pipeline import XCH_LEN, HostPipeline, f32, xch_pack, xch_unpack
from tests.test_host_math import CFG, hp_ptr

N, ITERS = 14, 8


def _inputs(N=N):
    bm = synth.make_body_model(260, seed=11)
    vp = synth.make_vposer(seed=12)
    clip = synth.make_clip(N, seed=13)
    scene = synth.make_scene(600, seed=14)
    l, r = synth.make_contact_ids(bm.v_template, per_part=10, seed=15)
    return bm, vp, clip, scene, np.concatenate([l, r])


def _sharded_fit(rank, world, group, N=N):
    bm, vp, clip, scene, vid = _inputs(N)
    hp = HostPipeline(bm, vp, scene, vid)
    x78 = np.zeros((N, 78), np.float32)
    hp.lib.h_75_to_78(hp_ptr(f32(clip.body_params)), N, hp_ptr(x78))
    idx1, pos = find_outliers(x78)
    init = x78.copy()
    if idx1.size:
        init[idx1] = x78[pos]
    mask = np.ones(N, np.float32)
    mask[idx1] = 0
    cam0 = read_camerapose(clip.camerapose_lines).reshape(N, 16)
    sh = FrameShard(N, group, rank=rank, world=world)
    lo, hi, nl = sh.frame0, sh.frame0 + sh.n_local, sh.n_local
    R = nl + 4
    rows_x, rows_cam = torch.zeros(R, 78), torch.zeros(R, 16)
    X0 = np.zeros((R, 78), np.float32)
    M = np.zeros(R, np.float32)
    rows_x[2:2 + nl] = torch.from_numpy(init[lo:hi])
    rows_cam[2:2 + nl] = torch.from_numpy(cam0[lo:hi])
    X0[2:2 + nl] = x78[lo:hi]
    M[2:2 + nl] = mask[lo:hi]
    exchange_halos(sh, rows_x, rows_cam)
    scale = np.array([1.8], np.float32)
    st = {k: np.zeros(s, np.float32) for k, s in (("mX", (nl, 78)), ("vX", (nl, 78)), ("mC", (nl, 16)), ("vC", (nl, 16)),
                                                   ("mS", (1,)), ("vS", (1,)))}
    P = first_phase2_iter(ITERS)
    send, gathered = torch.zeros(XCH_LEN), torch.zeros(world, XCH_LEN)
    for ii in range(ITERS):
        out = hp.backward(rows_x.numpy(), X0, M, rows_cam.numpy(), float(scale[0]), N, lo, 2, nl, ii >= P, CFG)
        losses = torch.from_numpy(out["losses"].copy())
        allreduce_scalars(sh, torch.zeros(1), losses)                       # logging iterations only, in production
        # fdcap_opt_step_rows_and_pack: Adam on the owned rows, then the message
        Xo = f32(rows_x[2:2 + nl].numpy())
        hp.adam(Xo, st["mX"], st["vX"], out["dX"], 0.005, ii + 1)
        rows_x[2:2 + nl] = torch.from_numpy(Xo)
        if ii >= P + 1:
            Co = f32(rows_cam[2:2 + nl].numpy())
            hp.adam(Co, st["mC"], st["vC"], f32(out["dCAM"].reshape(nl, 16)), 0.005, ii - P)
            rows_cam[2:2 + nl] = torch.from_numpy(Co)
        send.copy_(torch.from_numpy(xch_pack(rows_x.numpy(), rows_cam.numpy(), nl, out["dscale"])))
        # the iteration's one collective
        if world > 1:
            allgather_packed(sh, send, gathered)
        else:
            gathered[0] = send
        # fdcap_opt_unpack_and_step_scale: halo rows, scale gradient summed in rank order, Adam on scale
        rx, rc = rows_x.numpy(), rows_cam.numpy()                           # (views: written in place)
        dscale = np.array([xch_unpack(gathered.numpy(), rank, world, nl, rx, rc)], np.float32)
        if ii < P:
            hp.adam(scale, st["mS"], st["vS"], dscale, 0.005, ii + 1)
    return lo, hi, rows_x[2:2 + nl].numpy().copy(), rows_cam[2:2 + nl].numpy().copy(), float(scale[0]), losses.numpy()


def _worker(rank, world, port, q, n=N):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _sharded_fit(rank, world, dist.group.WORLD, n))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_frame_shard_partition():
    for n, w in ((1024, 8), (300, 7), (14, 3), (9, 4)):
        spans = [FrameShard(n, None, rank=r, world=w) for r in range(w)]
        assert spans[0].frame0 == 0 and sum(s.n_local for s in spans) == n
        for a, b in zip(spans, spans[1:]):
            assert a.frame0 + a.n_local == b.frame0
        assert max(s.n_local for s in spans) - min(s.n_local for s in spans) <= 1
    with pytest.raises(ValueError):
        FrameShard(5, None, rank=0, world=4)


def test_message_round_trip_on_the_bench_partition():
    """xch_pack / xch_unpack over BASELINE config 3's partition (1024 frames, 8 ranks), no process group: every rank's halo
    rows come out as its neighbours' boundary rows and every rank forms the same scale-gradient bits."""
    world, n = 8, 1024
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, 78)).astype(np.float32)
    C = rng.standard_normal((n, 16)).astype(np.float32)
    parts = rng.standard_normal(world).astype(np.float32)
    shards = [FrameShard(n, None, rank=r, world=world) for r in range(world)]
    rows, msgs = [], []
    for sh in shards:
        rx, rc = np.zeros((sh.n_local + 4, 78), np.float32), np.zeros((sh.n_local + 4, 16), np.float32)
        rx[2:2 + sh.n_local], rc[2:2 + sh.n_local] = X[sh.frame0:sh.frame0 + sh.n_local], C[sh.frame0:sh.frame0 + sh.n_local]
        rows.append((rx, rc))
        msgs.append(xch_pack(rx, rc, sh.n_local, parts[sh.rank]))
    gathered = np.stack(msgs)
    sums = []
    for sh, (rx, rc) in zip(shards, rows):
        sums.append(xch_unpack(gathered, sh.rank, world, sh.n_local, rx, rc))
        lo, hi = sh.frame0, sh.frame0 + sh.n_local
        for k, f in ((0, lo - 2), (1, lo - 1), (sh.n_local + 2, hi), (sh.n_local + 3, hi + 1)):
            if 0 <= f < n:
                np.testing.assert_array_equal(rx[k], X[f])
                np.testing.assert_array_equal(rc[k], C[f])
            else:
                assert not rx[k].any() and not rc[k].any()               # clip ends keep their zeros
    assert len(set(sums)) == 1


@pytest.mark.parametrize("world,n", [(2, N), (3, N), (8, 20)])
def test_sharded_run_matches_single_rank(world, n):
    ref = _sharded_fit(0, 1, None, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = np.concatenate([r[3] for r in res])
    C = np.concatenate([r[4] for r in res])
    assert [r[1] for r in res] == [FrameShard(n, None, rank=i, world=world).frame0 for i in range(world)]
    # per-frame arithmetic is identical; only the order of the scale-gradient sum differs
    np.testing.assert_allclose(X, ref[2], rtol=0, atol=2e-6)
    np.testing.assert_allclose(C, ref[3], rtol=0, atol=2e-6)
    assert len({r[5] for r in res}) == 1                                  # `scale` is bit-identical on every rank
    for r in res:
        assert abs(r[5] - ref[4]) < 1e-6
        np.testing.assert_allclose(r[6][:5], ref[5][:5], rtol=1e-6)


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fdcap_amd.dist import agree_on, same_on_all_ranks
        sh = FrameShard(40, dist.group.WORLD)
        # per-rank configuration that differs (verbose on rank 0 only): every rank takes rank 0's value
        got = agree_on(sh, 1 if rank == 0 else 0)
        lo, hi = same_on_all_ranks(sh, 100 + rank)          # (checkpoint files from different iterations: min != max)
        lo2, hi2 = same_on_all_ranks(sh, 250)
        q.put((rank, got, lo, hi, lo2, hi2))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_rank_zero_decides_collective_schedules_and_checkpoint_sets_are_checked():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [1, 1]                    # rank 1 said 0, follows rank 0
    assert all(r[2:4] == (100, 101) for r in res) and all(r[4:6] == (250, 250) for r in res)
