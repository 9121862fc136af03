#!/usr/bin/env python3
"""How long the host needs to ISSUE one optimiser iteration (ctypes calls + the torch.distributed collective of the sharded
tail), against what the GPU needs to run it.  Development tool.  FDCAP_FORCE_EXCHANGE=1 includes the exchange path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fdcap_amd
from fdcap_amd import capi, synth
from fdcap_amd.fitting import FittingOP
from fdcap_amd.io import read_camerapose
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
group = None
if os.environ.get("FDCAP_FORCE_EXCHANGE") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    group = dist.group.WORLD
bm = synth.make_body_model(10475, seed=0); vp = synth.make_vposer(seed=1); clip = synth.make_clip(N, seed=3)
scene = synth.make_scene(500000, seed=2); l, r = synth.make_contact_ids(bm.v_template, per_part=250, seed=4)
fop = FittingOP({"num_iter": 500}, {}, N, body_model=bm, vposer=vp, scene_verts=scene, contact_ids=np.concatenate([l, r]),
                camera_ext=read_camerapose(clip.camerapose_lines), group=group)
body = torch.tensor(clip.body_params).cuda()
fop.fitting(body, "global"); torch.cuda.synchronize()
from fdcap_amd.dist import allgather_packed
lib, h = fop.ctx.lib, fop.ctx.handle
def run(label, body_fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ii in range(400):
        body_fn(ii, capi.current_stream())
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    capi.check(lib.fdcap_opt_sync(h, capi.current_stream()), "sync")
    print(f"frames {N} {label}: host issued 400 phase-1 iterations in {1e3*(t1-t0):.1f} ms ({(t1-t0)*1e6/400:.1f} us/iter), GPU done after {1e3*(t2-t0):.1f} ms ({(t2-t0)*1e6/400:.1f} us/iter)", flush=True)


BIG = 10 ** 6


def run_c_loop(label, flags=0):
    import ctypes
    torch.cuda.synchronize()
    nl = ctypes.c_int32(0)
    t0 = time.perf_counter()
    capi.check(lib.fdcap_opt_run(h, 0, 400, BIG, BIG, 0, None, 0, flags, ctypes.byref(nl), capi.current_stream()), "run")
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    capi.check(lib.fdcap_opt_sync(h, capi.current_stream()), "sync")
    print(f"frames {N} {label}: host issued 400 phase-1 iterations in {1e3*(t1-t0):.1f} ms ({(t1-t0)*1e6/400:.1f} us/iter), GPU done after {1e3*(t2-t0):.1f} ms ({(t2-t0)*1e6/400:.1f} us/iter)", flush=True)


if group is None:
    def two_calls(ii, st):
        capi.check(lib.fdcap_opt_backward(h, ii, BIG, 0, st), "b")
        capi.check(lib.fdcap_opt_step(h, ii, BIG, st), "s")
    run("one GPU, fdcap_opt_backward + fdcap_opt_step (r3: 10 launches, 2 calls)", two_calls)
    run("one GPU, fdcap_opt_backward_and_step (r4: 9 launches, 1 call)", lambda ii, st: capi.check(lib.fdcap_opt_backward_and_step(h, ii, BIG, 0, st), "bs"))
    run_c_loop("one GPU, fdcap_opt_run (r4: the loop in the library, ONE call for the 400 iterations)")
else:
    def torch_tail(overlap):
        def f(ii, st):
            capi.check(lib.fdcap_opt_backward(h, ii, BIG, 0, st), "b")
            capi.check(lib.fdcap_opt_step_rows_and_pack(h, ii, BIG, capi.dptr(fop._xch_send), st), "p")
            ahead = (lambda ii=ii, st=st: capi.check(lib.fdcap_opt_forward_ahead(h, ii + 1, BIG, 0, st), "a")) if overlap else None
            allgather_packed(fop.shard, fop._xch_send, fop._xch_all, ahead)
            capi.check(lib.fdcap_opt_unpack_and_step_scale(h, ii, BIG, capi.dptr(fop._xch_all), 0, 1, st), "u")
        return f

    def c_tail(ii, st):
        capi.check(lib.fdcap_opt_backward(h, ii, BIG, 0, st), "b")
        capi.check(lib.fdcap_opt_exchange(h, ii, BIG, st), "x")
    run("one-rank RCCL group, torch.distributed all-gather between pack and unpack (r3)", torch_tail(False))
    run("one-rank RCCL group, the same + forward ahead of the exchange (asynchronous collective, r3)", torch_tail(True))
    if fop._c_comm:
        run("one-rank RCCL group, fdcap_opt_exchange: ncclAllGather on the compute stream from C (r4)", c_tail)
        run("one-rank RCCL group, fdcap_opt_exchange again", c_tail)
        run_c_loop("one-rank RCCL group, fdcap_opt_run (backward + fdcap_opt_exchange per iteration, one call for the 400)", 2)
    dist.destroy_process_group()
