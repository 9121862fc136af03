#!/usr/bin/env python3
"""Development tool: two-rank runs (one GPU, gloo) with and without the forward ahead of the exchange: where do they first differ?"""
import sys, os, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q, n, iters):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_gpu_sharded import _inputs
        from fdcap_amd.fitting import FittingOP
        from fdcap_amd.io import read_camerapose
        bm, vp, clip, scene, vid = _inputs(n)
        fop = FittingOP({"num_iter": iters}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                        camera_ext=read_camerapose(clip.camerapose_lines), group=dist.group.WORLD)
        body, scale, cam = fop.fitting(torch.tensor(clip.body_params).cuda(), "global", log_every=1)
        L = fop.log
        terms = np.array([L.l_rec, L.l_vposer, L.loss_smoothing, L.loss_contact, L.loss_world_smoothing, L.total]).T
        q.put((rank, body.cpu().numpy(), float(scale), cam.cpu().numpy(), terms))
    finally:
        dist.barrier(); dist.destroy_process_group()


def run(n, iters, ov):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["FDCAP_XCH_OVERLAP"] = ov
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, n, iters)) for r in range(2)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ref = run(n, iters, "0")
    names = ["rec", "vposer", "smooth", "contact", "world", "total"]
    bad = {"0": 0, "1": 0}
    for k in range(reps):
        for ov in ("0", "1"):
            r = run(n, iters, ov)
            ok = True
            for a, b in zip(ref, r):
                if not (np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])):
                    ok = False
                    d = np.argwhere(a[4] != b[4])
                    rows = sorted(set(np.argwhere(a[1] != b[1])[:, 0].tolist()))
                    print(f"  overlap {ov} run {k} rank {a[0]}: body rows differing {rows[:12]} (of {len(rows)}), scale equal {a[2] == b[2]}, cam equal {np.array_equal(a[3], b[3])}; "
                          f"first log differences (iteration, term, ref, got): {[(int(i), names[j], float(a[4][i, j]), float(b[4][i, j])) for i, j in d[:4]]}", flush=True)
            bad[ov] += not ok
    print("n", n, "iters", iters, ": mismatching runs plain", bad["0"], "ahead", bad["1"], "of", reps, flush=True)
