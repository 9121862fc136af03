#!/usr/bin/env python3
"""Development tool: two ranks on one GPU (gloo), manual iteration loop with per-frame hashes of X, dist, idx, dX after every
backward; repeated runs are compared to find the first (iteration, buffer, frame) that differs."""
import sys, os, socket, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hsum(t):          # raw rows (kept whole: the comparison reports frames, columns and magnitudes)
    return t.contiguous().view(t.shape[0], -1).cpu().numpy().copy()


def _worker(rank, world, port, q, n, iters):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.test_gpu_sharded import _inputs
        from fdcap_amd import capi
        from fdcap_amd.fitting import FittingOP, first_phase2_iter
        from fdcap_amd.dist import allgather_packed
        from fdcap_amd.io import read_camerapose
        bm, vp, clip, scene, vid = _inputs(n)
        if os.environ.get("SHIFT_VA") == "1" and rank > 0:      # different virtual addresses in the two processes
            import ctypes as _ct
            hip = _ct.CDLL("libamdhip64.so"); _p = _ct.c_void_p()
            hip.hipMalloc(_ct.byref(_p), _ct.c_size_t(rank * (192 << 20) + (3 << 20)))
        fop = FittingOP({"num_iter": iters}, {}, n, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=vid,
                        camera_ext=read_camerapose(clip.camerapose_lines), group=dist.group.WORLD)
        lib, h = fop.ctx.lib, fop.ctx.handle
        body = torch.tensor(clip.body_params).cuda()
        x78 = torch.empty(n, capi.XDIM, device="cuda")
        capi.check(lib.fdcap_params_75_to_78(capi.dptr(body), n, capi.dptr(x78), capi.current_stream()), "75->78")
        fop._mode = "global"; fop.init(x78)
        P = first_phase2_iter(iters)
        nl, nc = fop.shard.n_local, len(vid)
        d = torch.empty(nl, nc, device="cuda"); i = torch.empty(nl, nc, device="cuda", dtype=torch.int32)
        gx = torch.empty(nl, capi.XDIM, device="cuda"); gc = torch.empty(nl, 16, device="cuda")
        raw = None
        if os.environ.get("FDCAP_LIB", "").endswith("_dbg.so"):
            raw = ctypes.CDLL(capi.LIB_PATH)
            raw.fdcap_debug_rows.restype = ctypes.c_int
            raw.fdcap_debug_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            big = torch.empty(nl * 2048, device="cuda")
        trace = []
        for ii in range(iters):
            st = capi.current_stream()
            hx = hsum(fop._rows_x[2:2 + nl]); hs = hsum(fop._scale.view(1, 1)) if hasattr(fop, "_scale") else np.zeros((1, 1))
            capi.check(lib.fdcap_opt_backward(h, ii, P, 0, st), "b")
            rec = {"X": hx, "scale": hs}
            if ii < P:
                capi.check(lib.fdcap_opt_get_contact(h, capi.dptr(d), capi.dptr(i), st), "gc")
                rec["dist"] = hsum(d); rec["idx"] = hsum(i)
            if raw is not None:
                for which, name in enumerate(("O", "PF", "A", "M", "Voff", "Vw", "G", "Opart0", "H2", "Jw")):
                    w = raw.fdcap_debug_rows(h, which, ctypes.c_void_p(big.data_ptr()), st)
                    assert w > 0, (which, w)
                    rec[name] = hsum(big[: nl * w].view(nl, w))
            capi.check(lib.fdcap_opt_get_grads(h, capi.dptr(gx), capi.dptr(gc), st), "gg")
            rec["dX"] = hsum(gx); rec["dCAM"] = hsum(gc)
            trace.append(rec)
            capi.check(lib.fdcap_opt_step_rows_and_pack(h, ii, P, capi.dptr(fop._xch_send), st), "p")
            allgather_packed(fop.shard, fop._xch_send, fop._xch_all)
            capi.check(lib.fdcap_opt_unpack_and_step_scale(h, ii, P, capi.dptr(fop._xch_all), fop.shard.rank, fop.shard.world, st), "u")
        torch.cuda.synchronize()
        if raw is not None:
            bad = (ctypes.c_uint * 128)()
            raw.fdcap_debug_stage_bad.argtypes = [ctypes.c_void_p]
            raw.fdcap_debug_stage_bad(bad)
            if bad[0]:
                names = {1: "Jd", 2: "hand_comp", 3: "Jt", 4: "hand_mean", 5: "x", 6: "cam", 7: "parents", 8: "order", 9: "child_list", 10: "depth", 11: "child_start", 12: "level_start", 20: "Op0", 21: "Op1", 22: "Op2", 23: "Op3"}
                recs = [tuple(bad[8 * (k + 1): 8 * (k + 1) + 7]) for k in range(min(bad[0], 15))]
                print(f"STAGE rank {rank}: {bad[0]} staged words differ from their source: " + "; ".join(f"{names.get(t, t)}[{i}] lds {a:#x} src {b:#x} row {row} blk {blk}" for t, i, a, b, row, blk, ln in recs), flush=True)
        q.put((rank, trace))
    finally:
        dist.barrier(); dist.destroy_process_group()


WORLD = int(os.environ.get("WORLD", "2"))


def run(n, iters):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q, n, iters)) for r in range(WORLD)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=300) for _ in range(WORLD)), key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    bg = None
    if os.environ.get("BG"):                                  # another process keeps the GPU busy: "fit" (the same kernels) or "matmul"
        import subprocess, time
        from tools.contention_probe import LOAD
        bg = subprocess.Popen([sys.executable, "-c", LOAD, os.environ["BG"], "600"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        time.sleep(10)
    ref = run(n, iters)
    bad = 0
    for k in range(reps):
        r = run(n, iters)
        first = None
        for (rk, ta), (_, tb) in zip(ref, r):
            for ii, (a, b) in enumerate(zip(ta, tb)):
                for key in ("X", "scale", "Opart0", "H2", "O", "G", "A", "M", "Jw", "PF", "Voff", "Vw", "dist", "idx", "dX", "dCAM"):
                    if key in a and not np.array_equal(a[key], b[key]):
                        rows = sorted(set(np.nonzero(a[key].reshape(a[key].shape[0], -1) != b[key].reshape(b[key].shape[0], -1))[0].tolist()))
                        if first is None or ii < first[1]:
                            first = (rk, ii, key, rows[:8], len(rows))
                        break
                if first is not None and first[0] == rk and first[1] == ii:
                    break
        if first is not None:
            bad += 1
            rk, ii = first[0], first[1]
            a, b = ref[rk][1][ii], r[rk][1][ii]
            msg = []
            for key in a:
                if not np.array_equal(a[key], b[key]):
                    av, bv = a[key].reshape(a[key].shape[0], -1), b[key].reshape(b[key].shape[0], -1)
                    fr, col = np.nonzero(av != bv)
                    if av.dtype.kind == "f":
                        mag = float(np.abs(av[fr, col].astype(np.float64) - bv[fr, col]).max())
                        rel = float((np.abs(av[fr, col].astype(np.float64) - bv[fr, col]) / np.maximum(np.abs(av[fr, col]), 1e-30)).max())
                    else:
                        mag = rel = -1.0
                    msg.append(f"{key}: frames {sorted(set(fr.tolist()))[:4]} cols {sorted(set(col.tolist()))[:12]} ({len(col)} elements) max|d| {mag:.3g} max rel {rel:.3g}")
            print(f"  run {k}: first difference rank {rk} iteration {ii}: " + " | ".join(msg), flush=True)
    print(f"world {WORLD} n {n} iters {iters} background {os.environ.get('BG')}: {bad} of {reps} runs differ from the first", flush=True)
    if bg is not None:
        bg.terminate(); bg.wait()
