"""The sharded optimiser on the REAL kernels: 2 and 3 ranks share the one GPU of the test box
(process group backend gloo, halo rows staged through the host -- RCCL refuses two ranks on one
device), each owning a block of frames with halo rows, vs the single-rank run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fdcap_amd  # noqa: F401
from fdcap_amd import synth
from fdcap_amd.dist import FrameShard
from fdcap_amd.io import read_camerapose

pytestmark = pytest.mark.gpu
N, ITERS = 22, 10


def _inputs(n=N):
    bm = synth.make_body_model(300, seed=31)
    vp = synth.make_vposer(seed=32)
    clip = synth.make_clip(n, seed=33)
    scene = synth.make_scene(9000, seed=34)
    l, r = synth.make_contact_ids(bm.v_template, per_part=24, seed=35)
    return bm, vp, clip, scene, np.concatenate([l, r])


def _fit(group, mode="global", n=N):
    from fdcap_amd.fitting import FittingOP
    bm, vp, clip, scene, vid = _inputs(n)
    fop = FittingOP({"num_iter": ITERS}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                    camera_ext=read_camerapose(clip.camerapose_lines), group=group)
    body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), mode, log_every=1)
    tot = np.array(fop.log.total) if mode == "global" else np.array(fop.log2)[:, 5]
    out = (fop.shard.frame0, body.cpu().numpy(), float(scale), cam.cpu().numpy(), tot)
    fop.close()
    return out


def _worker(rank, world, port, q, mode="global", n=N):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _fit(dist.group.WORLD, mode, n))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# (800 frames over two ranks: shards large enough for the clip-sized kernel forms -- two row blocks per fragment stream in
# the blend products -- next to halo rows)
@pytest.mark.parametrize("world,mode,n", [(2, "global", N), (3, "global", N), (2, "local", N), (2, "global", 800)])
def test_sharded_gpu_run_matches_single_rank(world, mode, n):
    ref = _fit(None, mode, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, n)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [FrameShard(n, None, rank=i, world=world).frame0 for i in range(world)]
    body = np.concatenate([r[2] for r in res])
    cam = np.concatenate([r[4] for r in res])
    # identical per-frame arithmetic; only the order of the scale-gradient sum differs (1 ulp of dscale),
    # which Adam turns into <= ~1e-6 on scale after 8 steps
    np.testing.assert_allclose(body, ref[1], rtol=0, atol=5e-6)
    np.testing.assert_allclose(cam, ref[3], rtol=0, atol=5e-6)
    for r in res:
        assert abs(r[3] - ref[2]) < 2e-6
        np.testing.assert_allclose(r[5], ref[4], rtol=2e-6)      # all-reduced loss totals, every iteration


def _rccl_worker(port, q):
    """One rank, backend nccl (= RCCL): the sharded iteration tail with its real collective on device tensors."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FDCAP_FORCE_EXCHANGE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        q.put(_fit(dist.group.WORLD, "global"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_exchange_path_over_rccl_on_one_rank_equals_the_plain_loop():
    """The multi-GPU iteration tail (Adam on rows + pack, RCCL all_gather_into_tensor, unpack + Adam on scale, logging
    all-reduce) on a one-rank RCCL group: same arithmetic as the single-GPU loop, so results must be identical."""
    ref = _fit(None, "global")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    np.testing.assert_array_equal(res[1], ref[1])
    assert res[2] == ref[2]
    np.testing.assert_array_equal(res[3], ref[3])
    np.testing.assert_allclose(res[4], ref[4], rtol=1e-12)
