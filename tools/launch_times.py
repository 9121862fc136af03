#!/usr/bin/env python3
"""Live per-launch times of an iteration at given clip sizes (fdcap_opt_launch_timing through bench.time_all_launches):
usage: python tools/launch_times.py [--log] [--config c3|c5|c2] [--scene NS] [--lbs-nnz K] [--per-leg N] frames [frames ...]     (one rank's share of a sharded clip = a smaller clip)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    argv = sys.argv[1:]
    cfg, log_every = "c3", 0
    if argv and argv[0] == "--log":                      # the every-iteration-logging fit (bench.py's with_reference_logging)
        log_every, argv = 1, argv[1:]
    if argv and argv[0] == "--config":
        cfg, argv = argv[1], argv[2:]
    ns_over = None
    if argv and argv[0] == "--scene":                    # scene points (overrides the configuration's)
        ns_over, argv = int(argv[1]), argv[2:]
    lbs = 4
    if argv and argv[0] == "--lbs-nnz":                  # skinning weights per vertex of the synthetic body model
        lbs, argv = int(argv[1]), argv[2:]
    per_leg = 250
    if argv and argv[0] == "--per-leg":                  # contact vertices per leg (BASELINE: 250 -> 500 contact vertices)
        per_leg, argv = int(argv[1]), argv[2:]
    _, ns, allc = bench.CONFIGS[cfg]
    ns = ns_over or ns
    for frames in [int(a) for a in argv]:
        fop, body_gpu, *_ = bench.build_problem(frames, ns, allc, 10475, lbs, per_leg, 500, None)

        def one_step():
            b, sc, cam = fop.fitting(body_gpu, "global", log_every=log_every)
            return b.cpu(), sc, cam.cpu()
        one_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_step(); one_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        live = bench.time_all_launches(fop, one_step, 500, dt)
        p1 = {k: round(v["us_corrected"], 2) for k, v in live["phase1"].items()}
        p2 = {k: round(v["us_corrected"], 2) for k, v in live["phase2"].items()}
        print(f"{cfg} ns {ns:8d} nc {len(fop.vid):5d} frames {frames:5d}: {dt * 1e3:7.2f} ms/fit  {dt * 1e6 / 500:7.1f} us/iteration | phase 1 sum {sum(p1.values()):6.1f} us {json.dumps(p1)} | "
              f"phase 2 sum {sum(p2.values()):5.1f} us {json.dumps(p2)} | event overhead {live['event_overhead_us_per_launch']:.2f} us", flush=True)
        fop.close()
        del fop
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
