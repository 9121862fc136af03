"""FittingOP mirror: the reference's optimiser driver (global_optimization.py:141-188, :450-489,
:491-635, :637-653) over the fused HIP iteration (include/fdcap.h fdcap_opt_*).

Same constructor dictionaries, same `fitting(body_gpu, mode) -> (body_rec, scale, camera_ext)`
and `save_result(...)`.  Differences that are documented deviations, not behaviour changes:
clip length is free (the reference hard-codes 300), an empty outlier set is legal, the scene is
stored once, and log-only terms are evaluated only when logging is requested."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import capi, io
from .dist import FrameShard, agree_on, allgather_packed, allreduce_scalars, exchange_halos, same_on_all_ranks

# loss-total constants buried in fitting() (global_optimization.py:564, :570, :582)
PHASE_SPLIT = 0.8
PHASE1_CONTACT = 0.1
LOCAL_PHASE1_CONTACT = 0.2      # mode 'local' (:511)
LOCAL_SECOND_LOOP = 0.4         # int(0.4 * num_iter) iterations of cal_loss2 (:536)
PHASE1_SMOOTH = 1.0
PHASE2_WORLD = 1.0
PHASE2_SMOOTH = 0.5
SCALE_INIT = 1.8
OUTLIER_FACTOR = 1.8
VERBOSE_FLUSH = 50              # a verbose fit reads its device-side loss history back every this many logged iterations
# mode 'dct' (:41-45, :595-630)
BATCH_FRAME_NUM = 60            # frames per DCT window (:41)
DCT_NUM = 5                     # coefficients per trajectory (:43)
DCT_NUM_ITER = 10000            # forced inside fitting() (:596)
DCT_PHASE_SPLIT = 0.95          # (:601)
DCT_PHASE1_WEIGHT = 10.0        # loss = loss_dct*10 (:607)
DCT_PHASE2 = (0.0001, 0.5, 0.1)  # loss_dct, loss_rec, loss_contact (:620)

DEFAULT_FITTINGCONFIG = {
    "scene_verts_path": None, "camera_path": None, "human_model_path": "./models",
    "vposer_ckpt_path": "./vposer/", "init_lr_h": 0.005, "num_iter": 500, "batch_size": 1,
    "device": "cuda", "contact_id_folder": "./body_segments", "contact_part": ["L_Leg", "R_Leg"],
    "verbose": False,
}
DEFAULT_LOSSCONFIG = {"weight_loss_rec": 1, "weight_loss_vposer": 0.001, "weight_contact": 0.1,
                      "weight_collision": 0.5}


def first_phase2_iter(num_iter: int) -> int:
    """Smallest ii with not (ii < num_iter*0.8) (:564)."""
    return int(math.ceil(num_iter * PHASE_SPLIT - 1e-12))


def is_logging_iteration(ii: int, num_iter: int, log_every: int) -> bool:
    """Every log_every-th iteration and the fit's last one (the reference prints every iteration: log_every = 1)."""
    return bool(log_every) and (ii % log_every == 0 or ii == num_iter - 1)


def stretch_end(ii: int, num_iter: int, log_every: int = 0, snapshot_at=frozenset(), check_finite_every: int = 0,
                checkpoint_every: int = 0, flush_every: int = 0, unflushed_rows: int = 0) -> int:
    """The loop runs inside the library (fdcap_opt_run) in stretches [ii, end): `end` is the number of iterations done when this
    side next has something to do -- a snapshot after `end` steps, a finite check or a checkpoint every k-th iteration (no
    checkpoint after the last one), or a read-back of the loss history once `flush_every` logged rows wait (0: never inside the
    loop; `unflushed_rows` wait already) -- else num_iter.  Pure host logic (tests/test_host_math.py)."""
    end, rows = ii, unflushed_rows
    while end < num_iter:
        end += 1
        logged = is_logging_iteration(end - 1, num_iter, log_every)
        rows += 1 if logged else 0
        if (end in snapshot_at or (check_finite_every and end % check_finite_every == 0) or
                (checkpoint_every and end % checkpoint_every == 0 and end < num_iter) or
                (flush_every and logged and rows >= flush_every)):
            break
    return end


def find_outliers(x78: np.ndarray):
    """init() :459-487.  x78 [N,78] fp32 (6D form).  Returns (idx1 outlier rows, pos nearest
    inlier for each, ties -> lower index).  Generalised from the hard-coded 300 to N."""
    x78 = np.asarray(x78, dtype=np.float32)
    n = x78.shape[0]
    z = x78[:, 19:51]
    stats = np.sum(z * z, axis=1, dtype=np.float32)
    avg = np.float32(np.sum(stats, dtype=np.float32) / np.float32(n))
    idx1 = np.where(stats > avg * np.float32(OUTLIER_FACTOR))[0]
    temp = np.ones(n)
    temp[idx1] = 0.0
    index_one = np.where(temp == 1)[0]
    index_zero = np.where(temp == 0)[0]
    if index_zero.size == 0 or index_one.size == 0:
        return idx1, np.zeros(0, dtype=np.int64)
    diff = np.abs(index_zero[:, None] - index_one[None, :])
    pos = index_one[np.argmin(diff, axis=1)]
    return idx1, pos


@dataclass
class FitLog:
    """Per-logged-iteration losses in the reference's print order (:573-575, :587-589)."""
    iters: list
    l_rec: list
    l_vposer: list
    loss_smoothing: list
    loss_contact: list
    loss_world_smoothing: list
    total: list


class FittingOP:
    def __init__(self, fittingconfig, lossconfig, num_body, body_model=None, vposer=None, scene_verts=None,
                 contact_ids=None, camera_ext=None, group=None, legacy_zero_grad=False, n_left=None,
                 dct_mtx=None, c_dct_init=None, dct_num_iter=DCT_NUM_ITER):
        """`body_model` / `vposer`: objects with the SMPL-X npz / VPoser state-dict arrays
        (synth.BodyModelData / synth.VPoserData or assets.load_*).  `scene_verts`, `contact_ids`,
        `camera_ext` override the paths in `fittingconfig` when given (synthetic runs).
        `dct_mtx` [60,5]: load_dct_base() (:131-136; default: `fittingconfig['dct_mat_path']` if that .mat
        exists, else the orthonormal DCT-II basis); `c_dct_init` [N//60,23,3,5]: start of c_dct (the
        reference draws torch.randn, :186 -- so does the default); `dct_num_iter`: the 10000 of :596.
        `group`: torch.distributed process group for frame sharding (None = single GPU);
        `num_body` is the clip length N (the reference's batch_size = num_body, :152)."""
        import torch
        cfg = dict(DEFAULT_FITTINGCONFIG)
        cfg.update(fittingconfig or {})
        lcfg = dict(DEFAULT_LOSSCONFIG)
        lcfg.update(lossconfig or {})
        for k, v in cfg.items():
            setattr(self, k, v)
        for k, v in lcfg.items():
            setattr(self, k, v)
        self.snapshot_hook = None      # tools: called with k right after the snapshot of step k was taken (`fitting(snapshot_at=...)`)
        self.batch_size = self.num_body = int(num_body)
        self.legacy_zero_grad = bool(legacy_zero_grad)
        if not torch.cuda.is_available():
            raise capi.FdcapError("no HIP device: the fdcap_amd optimiser only runs on the GPU")
        self.device = torch.device("cuda", torch.cuda.current_device())
        if body_model is None or vposer is None:
            from . import assets
            body_model = body_model or assets.load_smplx_npz(self.human_model_path)
            vposer = vposer or assets.load_vposer_snapshot(self.vposer_ckpt_path)
        # host seconds of the three registration calls (each returns with its device work done): what a one-clip process pays
        # before its first iteration -- bench.py publishes them as `setup` (the reference's counterpart: __init__, :142-188)
        import time
        t0 = time.perf_counter()
        self.ctx = capi.Context(body_model, vposer)
        self.setup_s = {"ctx_create": time.perf_counter() - t0}
        if scene_verts is None and self.scene_verts_path:
            scene_verts = io.read_scene_points(self.scene_verts_path)
        if scene_verts is None:
            scene_verts = np.zeros((0, 3), np.float32)
        t0 = time.perf_counter()
        self.ctx.set_scene(scene_verts)
        self.setup_s["set_scene"] = time.perf_counter() - t0
        if contact_ids is None:
            parts = [io.read_contact_ids(self.contact_id_folder, [p]) for p in self.contact_part]
            contact_ids = np.concatenate(parts)
            n_left = len(parts[0])
        self.vid = np.asarray(contact_ids, dtype=np.int64)
        # mode 'local' treats the two contact parts separately (L_Leg ids first, :341-347)
        self.n_left = int(n_left) if n_left is not None else len(self.vid) // 2
        t0 = time.perf_counter()
        self.ctx.set_contact_ids(self.vid)
        self.setup_s["set_contact_ids"] = time.perf_counter() - t0
        if camera_ext is None and self.camera_path:
            camera_ext = io.read_camerapose(self.camera_path)
        self._camera_ext_init = None if camera_ext is None else np.asarray(camera_ext, np.float32).reshape(-1, 4, 4)
        self.shard = FrameShard(self.num_body, group)
        self.group = group
        # The exchange inside the library (fdcap_comm_* / fdcap_opt_exchange: RCCL on the compute stream, two C calls per
        # iteration) whenever the group runs over RCCL; FDCAP_C_COMM=0 keeps torch.distributed's collectives (the path the
        # gloo tests -- ranks sharing one GPU, which RCCL refuses -- always take).
        self._c_comm = False
        if group is not None:
            import os
            import torch.distributed as dist
            if dist.get_backend(group) == "nccl":
                from .dist import agree_on
                if agree_on(self.shard, int(os.environ.get("FDCAP_C_COMM", "1") != "0")):      # (rank 0's setting, for everyone)
                    self._init_c_comm()
        self.dct_mtx = None if dct_mtx is None else np.ascontiguousarray(dct_mtx, dtype=np.float32)
        self._c_dct_init = c_dct_init
        self.dct_num_iter = int(dct_num_iter)
        self.c_dct = None
        self.scale = None
        self.camera_ext = None
        self.body_rotation_rec = None
        self.log = None
        self.idx1 = None

    def _init_c_comm(self):
        """Rank 0 draws the RCCL unique id, torch.distributed carries it to the other ranks (the out-of-band channel any
        launcher has), every rank joins the library's own communicator.  Whether the library's exchange is used is ONE decision
        for the whole group (a rank on the other path would issue other collectives and hang its peers): every step that can
        fail on one rank alone -- binding librccl, creating the communicator, a first sum over it -- is followed by an
        agreement over torch.distributed, and any failure anywhere sends every rank to torch.distributed's collectives
        (RCCL as well; same kernels either side of the all-gather, same results), with a warning on rank 0."""
        import ctypes
        import warnings
        import torch
        import torch.distributed as dist
        from .dist import same_on_all_ranks
        lib, h = self.ctx.lib, self.ctx.handle
        sh = self.shard

        def all_ok(rc, what):
            lo, hi = same_on_all_ranks(sh, int(rc))
            if lo == 0 and hi == 0:
                return True
            if sh.rank == 0:
                why = lib.fdcap_comm_last_error(h) or b""
                warnings.warn(f"fdcap: {what} failed on some rank (codes {lo}..{hi}; this rank: {rc} {why.decode()!r}); "
                              "using torch.distributed's collectives instead")
            return False

        idb = (ctypes.c_uint8 * 128)()
        rc = lib.fdcap_comm_unique_id(idb)                  # (every rank: this is where librccl is bound; only rank 0's id is used)
        if not all_ok(rc, "binding librccl (fdcap_comm_unique_id)"):
            return
        t = torch.tensor(list(idb), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, src=sh.global_rank(0), group=self.group)
        idb = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
        rc = lib.fdcap_comm_create(h, idb, sh.rank, sh.world)
        if not all_ok(rc, "fdcap_comm_create"):
            lib.fdcap_comm_destroy(h)
            return
        # first use: sum of (rank + 1) over the communicator, on every rank
        probe = torch.tensor([float(sh.rank + 1)], dtype=torch.float64, device=self.device)
        rc = lib.fdcap_comm_allreduce_f64(h, capi.dptr(probe), 1, capi.current_stream())
        got = float(probe.item()) if rc == 0 else -1.0
        if not all_ok(rc if rc else int(got != sh.world * (sh.world + 1) / 2), "the first all-reduce over the library's communicator"):
            lib.fdcap_comm_destroy(h)
            return
        self._c_comm = True

    def _halos(self):
        """Halo rows <- the neighbouring ranks' boundary rows as they are."""
        if self._c_comm:
            capi.check(self.ctx.lib.fdcap_opt_halo_exchange(self.ctx.handle, capi.current_stream()), "fdcap_opt_halo_exchange")
        else:
            exchange_halos(self.shard, self._rows_x, self._rows_cam)

    def _sum_over_ranks(self, t64):
        """In-place sum of a float64 device tensor over the ranks (logged loss partial sums)."""
        import torch
        if self._c_comm:
            capi.check(self.ctx.lib.fdcap_comm_allreduce_f64(self.ctx.handle, capi.dptr(t64), t64.numel(), capi.current_stream()),
                       "fdcap_comm_allreduce_f64")
        else:
            allreduce_scalars(self.shard, torch.zeros(1, device=self.device), t64)

    # ---- :450-489 -------------------------------------------------------------------------
    def init(self, body_data_rotation):
        """body_data_rotation: [N,78] device tensor (whole clip).  Host-side index logic on a
        copy; returns idx1 like the reference and stages the optimiser's inputs."""
        import torch
        x78 = body_data_rotation.detach().cpu().numpy()
        idx1, pos = find_outliers(x78)
        init78 = x78.copy()
        if idx1.size and pos.size:
            init78[idx1, :] = x78[pos, :]
        mask = np.ones(self.num_body, np.float32)
        mask[idx1] = 0.0
        if self._camera_ext_init is None or self._camera_ext_init.shape[0] != self.num_body:
            raise capi.FdcapError("camera_ext / camerapose.txt must have one pose per frame (:455)")
        sh = self.shard
        lo, hi = sh.frame0, sh.frame0 + sh.n_local
        local = getattr(self, "_mode", "global") == "local"
        oc = capi.OptConfig(self.num_body, sh.n_local, sh.frame0, float(self.init_lr_h), float(self.weight_loss_rec),
                            float(self.weight_loss_vposer), float(self.weight_contact),
                            LOCAL_PHASE1_CONTACT if local else PHASE1_CONTACT, PHASE1_SMOOTH,
                            0.0 if local else PHASE2_WORLD, PHASE2_SMOOTH, SCALE_INIT, int(self.legacy_zero_grad))
        lib = self.ctx.lib
        import ctypes
        dev = self.device
        R = sh.n_local + 4
        # optimiser state is owned here (torch tensors) and registered with the library, so the
        # same storage is what RCCL exchanges / reduces
        self._rows_x = torch.zeros(R, capi.XDIM, device=dev)
        self._rows_cam = torch.zeros(R, 16, device=dev)
        self._scale = torch.zeros(1, device=dev)
        self._dscale = torch.zeros(1, device=dev)
        self._losses = torch.zeros(capi.NUM_LOSSES, device=dev, dtype=torch.float64)
        torch.cuda.current_stream().synchronize()
        capi.check(lib.fdcap_opt_create(self.ctx.handle, ctypes.byref(oc), capi.dptr(self._rows_x),
                                        capi.dptr(self._rows_cam), capi.dptr(self._scale), capi.dptr(self._dscale),
                                        capi.dptr(self._losses)), "fdcap_opt_create")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_data, d_init = body_data_rotation[lo:hi].contiguous(), t(init78[lo:hi])
        d_mask, d_cam = t(mask[lo:hi]), t(self._camera_ext_init[lo:hi].reshape(-1, 16))
        st = capi.current_stream()
        capi.check(lib.fdcap_opt_set_inputs(self.ctx.handle, capi.dptr(d_data), capi.dptr(d_init), capi.dptr(d_mask),
                                            capi.dptr(d_cam), st), "fdcap_opt_set_inputs")
        torch.cuda.current_stream().synchronize()
        self._halos()
        xl = int(lib.fdcap_exchange_len())
        self._xch_send = torch.zeros(xl, device=dev)
        self._xch_all = torch.zeros(self.shard.world, xl, device=dev)
        self.idx1 = idx1
        return idx1

    # ---- :491-635 -------------------------------------------------------------------------
    def fitting(self, body_data, mode="global", log_every=0, checkpoint_every=0, checkpoint_path=None, resume=None,
                check_finite_every=0, snapshot_at=()):
        """body_data: [N,75] (device tensor or numpy), SMPLify-X layout (:64-76).
        Returns (body_rec [N_local,75] device tensor, scale numpy scalar, camera_ext [N_local,4,4])
        -- the whole clip when not sharded, exactly the reference's triple (:635).
        Not in the reference:
          checkpoint_every=k, checkpoint_path: after every k-th iteration write parameters + Adam moments to
            `checkpoint_path` (sharded runs: one file per rank, suffix .rank<r>).  All three modes (r5): mode 'local' counts its
            second loop's iterations on from num_iter and keeps detect_contact's weights in the file; mode 'dct' counts the
            iterations of its own budget (`dct_num_iter`, the 10000 of :596), cuts the one-launch first phase at the checkpoint
            iterations and keeps c_dct with its Adam moments (with log_every: k must be a multiple of it);
          resume=path: continue such a run from where the file left off -- bit-identical to the uninterrupted run;
          check_finite_every=k: every k iterations count the non-finite parameters on the device and raise if any
            (the opt-in counterpart of the reference's set_detect_anomaly(True), :561); all three modes -- mode 'dct' checks the
            DCT coefficients and their Adam moments through its first phase (every lcm(k, log_every) iterations there), the body
            parameters from then on;
          snapshot_at=(k, ...) (mode 'global'): after the k-th optimiser step keep device copies of (body_rotation_rec [N_local,78],
            scale, camera_ext [N_local,16]) in self.snapshots[k] -- no host sync; for trajectory comparisons (tests/test_gpu_parity500.py).
        Sharded runs: every argument of this call and `num_iter` must be the same on all ranks (they decide which collectives
        are issued).  `verbose` may differ -- prints come from rank 0 only, and rank 0's flag alone decides whether the loss
        history is read back (all-reduced) during the loop, so a rank-0-only verbose run is legal."""
        import torch
        if mode not in ("global", "local", "dct"):
            raise ValueError("mode must be 'local', 'global' or 'dct' (global_optimization.py:660)")
        snapshot_at = frozenset(int(k) for k in snapshot_at)
        self.snapshots = {}
        if snapshot_at and mode != "global":
            raise ValueError("snapshot_at is implemented for mode 'global'")
        if checkpoint_every and not checkpoint_path:
            raise ValueError("checkpoint_every needs checkpoint_path")
        if mode == "dct" and checkpoint_every and log_every and checkpoint_every % log_every:
            raise ValueError("mode 'dct': checkpoint_every must be a multiple of log_every (the first phase's objective history is "
                             "written by the launch itself, every log_every-th iteration counted from the launch's first)")
        self._mode = mode
        lib, h = self.ctx.lib, self.ctx.handle
        dev = self.device
        if not torch.is_tensor(body_data):
            body_data = torch.from_numpy(np.ascontiguousarray(body_data, dtype=np.float32))
        body_data = body_data.to(dev, torch.float32).contiguous()
        n = body_data.shape[0]
        if n != self.num_body:
            raise capi.FdcapError(f"body_data has {n} frames, FittingOP was built for {self.num_body}")
        st = capi.current_stream()
        x78 = torch.empty(n, capi.XDIM, device=dev)
        capi.check(lib.fdcap_params_75_to_78(capi.dptr(body_data), n, capi.dptr(x78), st), "fdcap_params_75_to_78")
        self.init(x78)                                                                      # :495
        P = first_phase2_iter(self.num_iter)
        self._ck_extra = {}
        ii0 = self._load_checkpoint(resume) if resume else 0
        self._ck = (int(checkpoint_every), checkpoint_path, int(check_finite_every))
        log = FitLog([], [], [], [], [], [], [])
        # FDCAP_FORCE_EXCHANGE=1: run the sharded iteration tail (pack -> all-gather -> unpack) even on a one-rank
        # group -- lets a single-GPU box exercise the RCCL calls of the multi-GPU path
        import os
        multi = self.shard.world > 1 or (self.group is not None and os.environ.get("FDCAP_FORCE_EXCHANGE") == "1")
        # Forward ahead of the exchange (fdcap_opt_forward_ahead): FDCAP_XCH_OVERLAP=1 always, =0 never; otherwise the rank
        # times eight iterations of either schedule early in phase 1 and keeps the faster one.  The two give the same bits,
        # and the choice is local to a rank (the collective is the same call either way).  Why not simply "always": handing
        # work between the collective's stream and this one costs ~17 us per iteration on a one-rank RCCL group
        # (tools/host_issue_probe.py), so the overlap only pays when the all-gather takes longer than that to come back.
        env_ov = os.environ.get("FDCAP_XCH_OVERLAP", "auto")
        overlap = multi and env_ov == "1" and not self._c_comm        # (the library's own exchange runs on the compute stream: nothing to overlap)
        tune = None
        if multi and not self._c_comm and env_ov not in ("0", "1") and mode != "dct" and ii0 + 18 <= min(P, self.num_iter):
            tune = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.exchange_overlap = overlap
        if mode == "dct":
            self._dct_loops(lib, h, multi, log_every, ii0)
        # The reference prints every loss term in every iteration (:573-575, :587-589) with five .item() syncs.  Here the
        # partial sums of a logging iteration are written into a device-side history row (no host sync inside the loop) and
        # read back -- and, when sharded, all-reduced -- once after the last iteration.
        # FDCAP_DEFER_STEP=0: every optimiser step as its own launch (A/B; the results are the same bits either way)
        defer = os.environ.get("FDCAP_DEFER_STEP", "1") != "0"
        n_log = 0
        logged = []
        # in-loop read-backs of the loss history are collectives when sharded: one flag for the whole group (rank 0's)
        flush_in_loop = bool(agree_on(self.shard, 1 if self.verbose else 0)) if multi else bool(self.verbose)
        if log_every and mode != "dct":
            n_log = sum(1 for ii in range(ii0, self.num_iter) if ii % log_every == 0 or ii == self.num_iter - 1)
            hist = torch.zeros(max(n_log, 1), capi.NUM_LOSSES, device=dev, dtype=torch.float64)
        flushed = 0

        def flush(upto):
            # history rows [flushed, upto) -> log entries (and the reference's per-iteration print when verbose)
            nonlocal flushed
            if upto <= flushed:
                return
            part = hist[flushed:upto]
            if multi:
                part = part.clone()
                self._sum_over_ranks(part)
            rows = part.cpu().numpy()
            for k in range(upto - flushed):
                self._append_log(log, logged[flushed + k], logged[flushed + k] >= P, rows[k])
            flushed = upto

        # The loop inside the library (fdcap_opt_run: one C call per stretch of iterations between two things THIS side has to do --
        # a verbose flush, a snapshot, a finite check, a checkpoint; none of them in a plain fit: one call for all 500) wherever the
        # iteration is C calls only: one rank, or ranks exchanging through the library's communicator.  FDCAP_C_LOOP=0 keeps the
        # Python `for` below (same calls in the same order: same bits).
        c_loop = os.environ.get("FDCAP_C_LOOP", "1") != "0" and mode != "dct" and (not multi or self._c_comm)
        if c_loop:
            import ctypes
            n_done = ctypes.c_int32(0)
            st = capi.current_stream()

            def is_log(i):
                return is_logging_iteration(i, self.num_iter, log_every)

            ii = ii0
            try:
                while ii < self.num_iter:
                    # the stretch [ii, end): ends after the first iteration that leaves this side something to do
                    end = stretch_end(ii, self.num_iter, log_every, snapshot_at, check_finite_every, checkpoint_every,
                                      VERBOSE_FLUSH if flush_in_loop else 0, len(logged) - flushed)
                    rows = [i for i in range(ii, end) if is_log(i)]
                    k0 = len(logged)
                    capi.check(lib.fdcap_opt_run(h, ii, end, self.num_iter, P, int(log_every or 0),
                                                 capi.dptr(hist[k0]) if rows else None, len(rows), (0 if defer else 1) | (2 if multi else 0),
                                                 ctypes.byref(n_done), st), "fdcap_opt_run")
                    assert n_done.value == len(rows)
                    logged.extend(rows)
                    last = end - 1
                    if flush_in_loop and is_log(last) and len(logged) - flushed >= VERBOSE_FLUSH:
                        flush(len(logged))
                    if end in snapshot_at:
                        capi.check(lib.fdcap_opt_sync(h, st), "fdcap_opt_sync")
                        nl_ = self.shard.n_local
                        self.snapshots[end] = (self._rows_x[2:2 + nl_].clone(), self._scale.clone(), self._rows_cam[2:2 + nl_].clone())
                        if self.snapshot_hook:
                            self.snapshot_hook(end)
                    if check_finite_every and end % check_finite_every == 0:
                        self._check_finite(last)
                    if checkpoint_every and end % checkpoint_every == 0 and end < self.num_iter:
                        self._save_checkpoint(checkpoint_path, end)
                    ii = end
            finally:
                if logged:                                   # (fdcap_opt_run re-registers the old output itself; after an error, make sure)
                    capi.check(lib.fdcap_opt_set_loss_output(h, capi.dptr(self._losses)), "fdcap_opt_set_loss_output")
        try:
            for ii in range(ii0, self.num_iter if (mode != "dct" and not c_loop) else 0):       # :560
                do_log = bool(log_every) and (ii % log_every == 0 or ii == self.num_iter - 1)
                st = capi.current_stream()
                if tune is not None:                     # iterations ii0+2..9 plain, ii0+10..17 with the forward ahead
                    k = ii - ii0
                    if k in (2, 10, 18):
                        tune[(k - 2) // 8].record()
                    if k == 10:
                        overlap = True
                    elif k == 18:
                        tune[2].synchronize()
                        overlap = tune[1].elapsed_time(tune[2]) < 0.97 * tune[0].elapsed_time(tune[1])
                        self.exchange_overlap = overlap
                        tune = None
                if do_log:                               # this iteration's partial sums go straight into their history row
                    capi.check(lib.fdcap_opt_set_loss_output(h, capi.dptr(hist[len(logged)])), "fdcap_opt_set_loss_output")
                    logged.append(ii)
                if defer and not multi and ii + 1 < self.num_iter:
                    # loss.backward() + optimizer.step() in one call, the step without a launch of its own (fdcap.h): `scale` (and a
                    # logging iteration's printed sums) ride in the backward's last launch, the rows are stepped by the next
                    # iteration's first two launches; whoever reads the registered tensors before that calls fdcap_opt_sync first
                    # (snapshots below; export_state does it itself)
                    capi.check(lib.fdcap_opt_backward_and_step(h, ii, P, 2 if do_log else 0, st), "fdcap_opt_backward_and_step")
                    stepped = True
                else:
                    # (2: the logged sums are delivered by the step launch that follows -- one launch less per logged iteration)
                    capi.check(lib.fdcap_opt_backward(h, ii, P, 2 if do_log else 0, st), "fdcap_opt_backward")
                    stepped = False
                if stepped:
                    pass
                elif multi and self._c_comm:
                    # the same tail inside the library: Adam on the rows + message, ncclAllGather on this stream, unpack + scale
                    capi.check(lib.fdcap_opt_exchange(h, ii, P, st), "fdcap_opt_exchange")
                elif multi:
                    # one collective per iteration: boundary rows (after Adam) + the scale-gradient partial travel
                    # together; every rank then sums the partials in rank order and steps `scale` identically
                    capi.check(lib.fdcap_opt_step_rows_and_pack(h, ii, P, capi.dptr(self._xch_send), st), "step_rows_and_pack")
                    ahead = None
                    if overlap and ii + 1 < self.num_iter:
                        # the next iteration's decoder / pose state / blend product of the owned rows need neither `scale`
                        # nor the halo rows: they run while the messages travel
                        nxt_log = bool(log_every) and ((ii + 1) % log_every == 0 or ii + 1 == self.num_iter - 1)

                        def ahead(ii=ii, nxt_log=nxt_log, st=st):
                            capi.check(lib.fdcap_opt_forward_ahead(h, ii + 1, P, 2 if nxt_log else 0, st), "fdcap_opt_forward_ahead")
                    allgather_packed(self.shard, self._xch_send, self._xch_all, ahead)
                    capi.check(lib.fdcap_opt_unpack_and_step_scale(h, ii, P, capi.dptr(self._xch_all), self.shard.rank,
                                                                   self.shard.world, st), "unpack_and_step_scale")
                else:
                    capi.check(lib.fdcap_opt_step(h, ii, P, st), "fdcap_opt_step")
                # the reference prints every iteration live (:573-575); a verbose fit shows its progress in batches of
                # VERBOSE_FLUSH logged iterations (one read-back each) instead of only after the last one
                if flush_in_loop and do_log and len(logged) - flushed >= VERBOSE_FLUSH:
                    flush(len(logged))
                if ii + 1 in snapshot_at:
                    capi.check(lib.fdcap_opt_sync(h, st), "fdcap_opt_sync")
                    nl_ = self.shard.n_local
                    self.snapshots[ii + 1] = (self._rows_x[2:2 + nl_].clone(), self._scale.clone(), self._rows_cam[2:2 + nl_].clone())
                    if self.snapshot_hook:
                        self.snapshot_hook(ii + 1)
                if check_finite_every and (ii + 1) % check_finite_every == 0:
                    self._check_finite(ii)
                if checkpoint_every and (ii + 1) % checkpoint_every == 0 and ii + 1 < self.num_iter:
                    self._save_checkpoint(checkpoint_path, ii + 1)
        finally:
            # never leave the library pointing into `hist` (freed with this frame if the loop raised)
            if logged:
                capi.check(lib.fdcap_opt_set_loss_output(h, capi.dptr(self._losses)), "fdcap_opt_set_loss_output")
        flush(len(logged))
        if mode == "local":
            self._local_second_loop(lib, h, multi, log_every, max(ii0 - self.num_iter, 0))
        nl = self.shard.n_local
        body_rec = torch.empty(nl, capi.PDIM, device=dev)
        scale = torch.empty(1, device=dev)
        cam = torch.empty(nl, 16, device=dev)
        capi.check(lib.fdcap_opt_get_results(h, capi.dptr(body_rec), capi.dptr(scale), capi.dptr(cam),
                                             capi.current_stream()), "fdcap_opt_get_results")
        self.scale = scale
        self.camera_ext = cam.view(nl, 4, 4)
        self.body_rotation_rec = self._rows_x[2:2 + nl]
        self.log = log
        return body_rec, scale.detach().cpu().numpy().squeeze(), self.camera_ext

    # ---- checkpoint / resume / finite check (not in the reference) ---------------------
    def _ckpt_file(self, path):
        return path if self.shard.world == 1 else f"{path}.rank{self.shard.rank}"

    def _budget(self):
        """the iteration budget a checkpoint belongs to (mode 'dct' runs its own, :596)"""
        return int(self.dct_num_iter if getattr(self, "_mode", "global") == "dct" else self.num_iter)

    def _save_checkpoint(self, path, next_iter, extra_fn=None, **extra):
        """`extra_fn`: callable returning more arrays for the file (mode 'dct': c_dct and its moments, read from the device).
        EVERYTHING that can fail on one rank alone -- the allocation, the library's state export, the extras' read-back, the
        write -- runs inside the try whose outcome the ranks then agree on (a HIP error or an out-of-memory in the
        export used to raise before the agreement and leave the peers waiting in it)."""
        import os
        import torch
        lib, h, nl = self.ctx.lib, self.ctx.handle, self.shard.n_local
        fn = self._ckpt_file(path)
        tmp = fn + ".tmp.npz"
        err = None
        try:
            state = torch.empty(int(lib.fdcap_opt_state_len(h)), device=self.device)
            capi.check(lib.fdcap_opt_export_state(h, capi.dptr(state), capi.current_stream()), "fdcap_opt_export_state")   # (applies a deferred step first)
            if extra_fn is not None:
                extra = dict(extra, **extra_fn())
            np.savez(tmp, next_iter=np.int64(next_iter), num_iter=np.int64(self._budget()), n_total=np.int64(self.num_body),
                     frame0=np.int64(self.shard.frame0), n_local=np.int64(nl), rows_x=self._rows_x[2:2 + nl].cpu().numpy(),
                     rows_cam=self._rows_cam[2:2 + nl].cpu().numpy(), scale=self._scale.cpu().numpy(), state=state.cpu().numpy(),
                     mode=np.array(getattr(self, "_mode", "global")),
                     **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in extra.items()})
        except Exception as e:  # noqa: BLE001 -- agreed on below: one rank's failed write must not leave its peers in a collective
            err = e
        # sharded: nobody replaces its file before every rank has written its new one (a crash during the writes leaves the
        # previous, mutually consistent set); a crash between the renames is caught on resume (next_iter must agree).  The
        # agreement doubles as that barrier AND carries every rank's outcome (a bare barrier turned one rank's
        # disk-full exception into a group hang): any failure anywhere -> nobody renames, everybody raises.
        lo, hi = same_on_all_ranks(self.shard, 0 if err is None else 1)
        if hi != 0:
            try:
                os.remove(tmp)
            except OSError:
                pass
            if err is not None:
                raise err
            raise capi.FdcapError(f"checkpoint for iteration {next_iter}: another rank could not write its file; none was replaced")
        os.replace(tmp, fn)                                      # (a crash mid-write leaves the previous checkpoint intact)

    def _load_checkpoint(self, path):
        """After init(): parameters and Adam moments <- the file; returns the iteration to continue with."""
        import torch
        lib, h, nl = self.ctx.lib, self.ctx.handle, self.shard.n_local
        with np.load(self._ckpt_file(path)) as ck:
            if (int(ck["n_total"]), int(ck["frame0"]), int(ck["n_local"]), int(ck["num_iter"])) != \
                    (self.num_body, self.shard.frame0, nl, self._budget()):
                raise capi.FdcapError("checkpoint was written for another clip length / sharding / iteration budget")
            ck_mode = str(ck["mode"]) if "mode" in ck.files else "global"
            if ck_mode != self._mode:
                raise capi.FdcapError(f"checkpoint was written by a mode {ck_mode!r} fit, this is mode {self._mode!r}")
            self._ck_extra = {k: np.array(ck[k]) for k in ("contact_weight", "c_dct", "dct_m", "dct_v") if k in ck.files}
            t = lambda k: torch.from_numpy(np.ascontiguousarray(ck[k], dtype=np.float32)).to(self.device)
            self._rows_x[2:2 + nl] = t("rows_x")
            self._rows_cam[2:2 + nl] = t("rows_cam")
            self._scale.copy_(t("scale"))
            state = t("state").contiguous()
            nxt = int(ck["next_iter"])
        if state.numel() != int(lib.fdcap_opt_state_len(h)):
            raise capi.FdcapError("checkpoint state has the wrong size")
        lo, hi = same_on_all_ranks(self.shard, nxt)
        if lo != hi:
            raise capi.FdcapError(f"the ranks' checkpoint files are from different iterations ({lo} .. {hi}): a run died between "
                                  f"two ranks' writes; resume from an older, complete set")
        capi.check(lib.fdcap_opt_import_state(h, capi.dptr(state), capi.current_stream()), "fdcap_opt_import_state")
        torch.cuda.current_stream().synchronize()
        self._halos()
        return nxt

    def _check_finite(self, ii):
        import torch
        cnt = torch.zeros(1, device=self.device, dtype=torch.int32)
        capi.check(self.ctx.lib.fdcap_opt_check_finite(self.ctx.handle, capi.dptr(cnt), capi.current_stream()), "fdcap_opt_check_finite")
        bad = int(cnt.cpu())
        if self.shard.world > 1:
            t = torch.tensor([float(bad)], device=self.device)
            allreduce_scalars(self.shard, t)
            bad = int(t.cpu())
        if bad:
            raise capi.FdcapError(f"{bad} non-finite optimiser parameters after iteration {ii}")

    # ---- :595-630 -------------------------------------------------------------------------
    def _dct_loops(self, lib, h, multi, log_every, ii0=0):
        """mode 'dct'.  First 95 % of the iterations: only c_dct moves (loss_dct*10, :601-607) against frozen
        joint trajectories -- ONE launch; iteration ceil(0.95*num_iter) is a no-op (every leaf's flag was
        flipped after its forward, :615-618, so nothing receives a gradient); the rest optimise
        body_rotation_rec + scale with loss_dct*1e-4 + loss_rec*0.5 + loss_contact*0.1 (:620)."""
        import torch
        # legacy_zero_grad (torch < 2: zero_grad() zeroes gradients instead of dropping them): body / scale / camera never
        # hold a gradient before the switch, so the first phase is the same; from iteration ceil(0.95 num_iter) on the
        # frozen c_dct keeps a zero gradient and Adam keeps stepping it on its decaying moments.
        legacy = self.legacy_zero_grad
        N, T, dev = self.num_body, BATCH_FRAME_NUM, self.device
        W = N // T
        if W < 1:
            raise capi.FdcapError(f"mode 'dct' needs at least one {T}-frame window (:41-42); clip has {N} frames")
        sh = self.shard
        if multi and any(sh.bounds(r)[0] % T for r in range(sh.world)):
            raise capi.FdcapError(f"mode 'dct': frame shards must start on {T}-frame window boundaries "
                                  f"(N={N}, world={sh.world})")
        if self.dct_mtx is None:
            self.dct_mtx = io.load_dct_base(getattr(self, "dct_mat_path", None), T, DCT_NUM)
        D = np.ascontiguousarray(self.dct_mtx, np.float32)
        C = D.shape[1]
        ck_every, ck_path, finite_every = getattr(self, "_ck", (0, None, 0))
        ex = getattr(self, "_ck_extra", {})
        resumed = "c_dct" in ex
        if ii0 and not resumed:
            raise capi.FdcapError("the checkpoint holds no c_dct: not written by a mode 'dct' fit")
        if ii0 and log_every and ii0 % log_every:
            raise ValueError(f"mode 'dct': resuming at iteration {ii0} needs a log_every that divides it")
        c0 = ex["c_dct"] if resumed else self._c_dct_init
        c0 = torch.randn(W, 23, 3, C) if c0 is None else torch.as_tensor(np.asarray(c0, dtype=np.float32).reshape(W, 23, 3, -1))
        if tuple(c0.shape) != (W, 23, 3, C):
            raise capi.FdcapError(f"c_dct_init must be [{W},23,3,{C}], got {tuple(c0.shape)}")
        c0 = c0.to(dev, torch.float32).contiguous()
        st = capi.current_stream()
        import ctypes
        capi.check(lib.fdcap_opt_set_dct(h, D.ctypes.data_as(ctypes.c_void_p), D.shape[0], C, capi.dptr(c0), st),
                   "fdcap_opt_set_dct")
        if resumed:                                            # Adam's moments of c_dct as the interrupted fit left them
            dm = torch.as_tensor(ex["dct_m"]).to(dev, torch.float32).contiguous()
            dv = torch.as_tensor(ex["dct_v"]).to(dev, torch.float32).contiguous()
            capi.check(lib.fdcap_opt_set_dct_state(h, capi.dptr(dm), capi.dptr(dv), st), "fdcap_opt_set_dct_state")
        num_iter = self.dct_num_iter
        P = int(math.ceil(num_iter * DCT_PHASE_SPLIT - 1e-9))           # first ii with not (ii < num_iter*0.95)
        w0, w1 = ctypes.c_int32(), ctypes.c_int32()
        lib.fdcap_opt_dct_windows(h, ctypes.byref(w0), ctypes.byref(w1))
        ntraj = 69 * (w1.value - w0.value)

        def save(next_iter):                                   # parameters + Adam moments + c_dct with ITS moments
            def dct_state():                                   # (runs inside _save_checkpoint's agreed-on try)
                cd_ = torch.empty(W, 69 * C, device=dev)
                dm_, dv_ = torch.empty_like(cd_), torch.empty_like(cd_)
                capi.check(lib.fdcap_opt_get_dct(h, capi.dptr(cd_), capi.current_stream()), "fdcap_opt_get_dct")
                capi.check(lib.fdcap_opt_get_dct_state(h, capi.dptr(dm_), capi.dptr(dv_), capi.current_stream()), "fdcap_opt_get_dct_state")
                return {"c_dct": cd_, "dct_m": dm_, "dct_v": dv_}
            self._save_checkpoint(ck_path, next_iter, extra_fn=dct_state)

        # first phase [ii0, P): ONE launch -- or one per stretch between two checkpoint iterations (the launch keeps coefficients
        # and moments in registers and writes them back at its end; Adam's step counter runs on through step0: same bits)
        hist = None
        if log_every and ntraj and P > ii0:
            hist = torch.zeros((P - ii0 + log_every - 1) // log_every, ntraj, device=dev)
        def check_dct_finite(done):                            # phase 1 moves only c_dct: count ITS non-finite entries (and its moments')
            cd_ = torch.empty(W, 69 * C, device=dev)
            dm_, dv_ = torch.empty_like(cd_), torch.empty_like(cd_)
            capi.check(lib.fdcap_opt_get_dct(h, capi.dptr(cd_), capi.current_stream()), "fdcap_opt_get_dct")
            capi.check(lib.fdcap_opt_get_dct_state(h, capi.dptr(dm_), capi.dptr(dv_), capi.current_stream()), "fdcap_opt_get_dct_state")
            own = slice(w0.value, w1.value)                    # (sharded: the windows this rank fits)
            bad = sum((~torch.isfinite(t[own])).sum() for t in (cd_, dm_, dv_)).to(torch.float32).reshape(1)
            if multi:
                allreduce_scalars(sh, bad)
            if int(bad.cpu()):
                raise capi.FdcapError(f"{int(bad.cpu())} non-finite DCT coefficients / Adam moments after iteration {done - 1}")

        # (a stretch's logged rows are counted from its first iteration: a cut must fall on a logging iteration -- checkpoints
        #  already require it; the finite check of this phase therefore runs every lcm(check_finite_every, log_every) iterations)
        finite_cut = (finite_every * log_every // math.gcd(finite_every, log_every)) if (finite_every and log_every) else finite_every
        ii = ii0
        while ii < P:
            end = P                                            # cut at the next checkpoint / finite-check iteration, if any
            for every in (ck_every, finite_cut):
                if every:
                    end = min(end, (ii // every + 1) * every)
            row = capi.dptr(hist[(ii - ii0) // log_every]) if hist is not None else None
            capi.check(lib.fdcap_opt_dct_fit(h, end - ii, ii, DCT_PHASE1_WEIGHT, row, max(int(log_every), 1), st), "fdcap_opt_dct_fit")
            ii = end
            if finite_cut and (ii % finite_cut == 0 or ii == P):
                check_dct_finite(ii)
            if ck_every and ii % ck_every == 0 and ii < num_iter:
                save(ii)                                       # (sharded: every rank's own windows; merged after the phase, below)
        self.log_dct = []
        if hist is not None:
            part = hist.sum(dim=1, dtype=torch.float64)
            if multi:
                allreduce_scalars(sh, torch.zeros(1, device=dev), part)
            self.log_dct = [[ii0 + k * log_every, float(v) / (69 * W)] for k, v in enumerate(part.cpu().numpy())]

        def merge_windows():                                   # every rank ends up with all windows' coefficients
            import torch.distributed as dist
            full = torch.zeros(W, 69 * C, device=dev)
            capi.check(lib.fdcap_opt_get_dct(h, capi.dptr(full), capi.current_stream()), "fdcap_opt_get_dct")
            own = torch.zeros_like(full)
            own[w0.value:w1.value] = full[w0.value:w1.value]
            if dist.get_backend(self.group) == "gloo":
                t = own.cpu(); dist.all_reduce(t, group=self.group); own = t.to(dev)
            else:
                dist.all_reduce(own, group=self.group)
            capi.check(lib.fdcap_opt_set_dct_coef(h, capi.dptr(own.contiguous()), capi.current_stream()), "fdcap_opt_set_dct_coef")

        if multi and ii0 <= P:                                 # (a checkpoint from after the phase holds merged coefficients)
            merge_windows()
        if legacy and num_iter > P and ii0 <= P:               # iteration P: nothing receives a gradient, c_dct coasts
            capi.check(lib.fdcap_opt_dct_fit(h, 1, P, 0.0, None, 1, st), "fdcap_opt_dct_fit")
        BIG = 2 ** 30
        wd, wr, wc = DCT_PHASE2
        self.log2 = []
        for k in range(max(ii0 - P - 1, 0), max(num_iter - P - 1, 0)):     # ii = P + 1 + k; Adam step counters k + 1
            ii = P + 1 + k
            do_log = bool(log_every) and (ii % log_every == 0 or ii == num_iter - 1)
            st = capi.current_stream()
            capi.check(lib.fdcap_opt_backward_dct(h, wd, wr, wc, 1 if do_log else 0, st), "fdcap_opt_backward_dct")
            if do_log:
                s = self._losses.clone()
                if multi:
                    allreduce_scalars(sh, torch.zeros(1, device=dev), s)
                s = s.cpu().numpy()
                nc = max(self.ctx.num_contact, 1)
                l_rec = self.weight_loss_rec * s[0] / (N * capi.XDIM)
                l_vp = self.weight_loss_vposer * s[1] / (N * 32)
                l_sm = s[2] / ((N - 2) * capi.XDIM) if N >= 3 else float("nan")
                l_con = self.weight_contact * s[3] / (N * nc)
                l_dct = s[7] / (69 * W)
                self.log2.append([ii, l_rec, l_vp, l_sm, l_con, l_dct, wd * l_dct + wr * l_rec + wc * l_con])
                if self.verbose and sh.rank == 0:
                    print('[INFO][fitting] iter={:d}, l_rec={:f}, l_vposer={:f}, loss_smoothing={:f}, loss_contact={:f}, '
                          'loss_dct={:f}, total_loss={:f}'.format(*self.log2[-1]))
            if multi and self._c_comm:
                capi.check(lib.fdcap_opt_exchange(h, k, BIG, st), "fdcap_opt_exchange")
            elif multi:
                capi.check(lib.fdcap_opt_step_rows_and_pack(h, k, BIG, capi.dptr(self._xch_send), st), "step_rows_and_pack")
                allgather_packed(sh, self._xch_send, self._xch_all)
                capi.check(lib.fdcap_opt_unpack_and_step_scale(h, k, BIG, capi.dptr(self._xch_all), sh.rank, sh.world, st),
                           "unpack_and_step_scale")
            else:
                capi.check(lib.fdcap_opt_step(h, k, BIG, st), "fdcap_opt_step")
            if legacy:                                         # optimizer.step() of this iteration also moves the frozen c_dct
                capi.check(lib.fdcap_opt_dct_fit(h, 1, ii, 0.0, None, 1, st), "fdcap_opt_dct_fit")
            if finite_every and (ii + 1) % finite_every == 0:
                self._check_finite(ii)
            if ck_every and (ii + 1) % ck_every == 0 and ii + 1 < num_iter:
                save(ii + 1)
        if multi and legacy:                                   # (each rank coasted its own windows)
            merge_windows()
        cd = torch.empty(W, 23, 3, C, device=dev)
        capi.check(lib.fdcap_opt_get_dct(h, capi.dptr(cd), capi.current_stream()), "fdcap_opt_get_dct")
        self.c_dct = cd

    def _local_second_loop(self, lib, h, multi, log_every, jj0=0):
        """detect_contact + the cal_loss2 loop of mode 'local' (:534-556).  jj0 > 0: resumed inside this loop -- the contact
        weights (a function of the parameters as the FIRST loop left them) come from the checkpoint."""
        import torch
        nl = self.shard.n_local
        ck_every, ck_path, finite_every = getattr(self, "_ck", (0, None, 0))
        ex = getattr(self, "_ck_extra", {})
        if jj0 and "contact_weight" not in ex:
            raise capi.FdcapError("the checkpoint holds no contact weights: not written inside mode 'local''s second loop")
        w_local = torch.empty(nl, device=self.device)
        if jj0:
            weight = torch.as_tensor(ex["contact_weight"]).to(self.device, torch.float32).contiguous()
        else:
            capi.check(lib.fdcap_opt_detect_contact(h, self.n_left, capi.dptr(w_local), capi.current_stream()),
                       "fdcap_opt_detect_contact")
        if jj0:
            pass
        elif multi:                                            # every rank needs the weight of its right neighbour's first frame
            import torch.distributed as dist
            parts = [torch.empty(self.shard.bounds(r)[1] - self.shard.bounds(r)[0], device=self.device)
                     for r in range(self.shard.world)]
            if dist.get_backend(self.group) == "gloo":
                cpu_parts = [p.cpu() for p in parts]
                dist.all_gather(cpu_parts, w_local.cpu(), group=self.group)
                parts = [p.to(self.device) for p in cpu_parts]
            else:
                dist.all_gather(parts, w_local, group=self.group)
            weight = torch.cat(parts).contiguous()
        else:
            weight = w_local
        self.contact_weight = weight
        self.log2 = []
        n2 = int(LOCAL_SECOND_LOOP * self.num_iter)
        for jj in range(jj0, n2):
            st = capi.current_stream()
            capi.check(lib.fdcap_opt_backward_local2(h, capi.dptr(weight), self.n_left, st), "fdcap_opt_backward_local2")
            if log_every and (jj % log_every == 0):
                s = self._losses.clone()
                if multi:
                    allreduce_scalars(self.shard, torch.zeros(1, device=self.device), s)
                s = s.cpu().numpy()
                N = self.num_body
                l_rec = self.weight_loss_rec * s[0] / (N * capi.XDIM)
                l_loc = s[2] / ((N - 2) * capi.XDIM)
                l_sm = s[5] / ((N - 2) * 3 * self.ctx.num_verts)
                l_cs = s[6]
                self.log2.append([jj, l_rec, l_loc, l_sm, l_cs, l_sm + l_loc + l_rec + l_cs])
                if self.verbose and self.shard.rank == 0:
                    print('[INFO][fitting] iter={:d}, l_rec={:f}, loss_local_smoothing={:f}, loss_smoothing={:f}, '
                          'loss_contact_smoothing={:f}, total_loss={:f}'.format(*self.log2[-1]))
            capi.check(lib.fdcap_opt_step_x(h, self.num_iter + jj + 1, st), "fdcap_opt_step_x")
            if multi:
                self._halos()
            done = self.num_iter + jj + 1                    # iterations of the whole fit made so far
            if finite_every and done % finite_every == 0:
                self._check_finite(done - 1)
            if ck_every and done % ck_every == 0 and jj + 1 < n2:
                self._save_checkpoint(ck_path, done, contact_weight=weight)

    def _append_log(self, log, ii, phase2, s=None):
        s = self._losses.cpu().numpy() if s is None else s
        N, nc = self.num_body, max(self.ctx.num_contact, 1)
        l_rec = self.weight_loss_rec * s[0] / (N * capi.XDIM)
        l_vp = self.weight_loss_vposer * s[1] / (N * 32)
        l_sm = s[2] / ((N - 2) * capi.XDIM) if N >= 3 else float("nan")
        l_con = self.weight_contact * s[3] / (N * nc)
        l_ws = s[4] / ((N - 1) * 69) if N >= 2 else float("nan")
        local = getattr(self, "_mode", "global") == "local"
        total = (l_rec + (0.0 if local else PHASE2_WORLD * l_ws) + PHASE2_SMOOTH * l_sm) if phase2 else \
            ((LOCAL_PHASE1_CONTACT if local else PHASE1_CONTACT) * l_con + PHASE1_SMOOTH * l_sm + l_rec)
        log.iters.append(ii); log.l_rec.append(l_rec); log.l_vposer.append(l_vp)
        log.loss_smoothing.append(l_sm); log.loss_contact.append(l_con)
        log.loss_world_smoothing.append(l_ws); log.total.append(total)
        if self.verbose and self.shard.rank == 0:
            if phase2:
                print('[INFO][fitting] iter={:d}, l_rec={:f}, l_vposer={:f}, loss_smoothing={:f}, loss_contact={:f}, '
                      'loss_world_smoothing={:f}, total_loss={:f}'.format(ii, l_rec, l_vp, l_sm, l_con, l_ws, total))
            else:
                print('[INFO][fitting] iter={:d}, l_rec={:f}, l_vposer={:f}, loss_smoothing={:f}, loss_contact={:f}, '
                      'total_loss={:f}'.format(ii, l_rec, l_vp, l_sm, l_con, total))

    # ---- :637-653 -------------------------------------------------------------------------
    def save_result(self, body_rec, scale, camera_ext, fit_path):
        import torch
        br = body_rec.detach().cpu().numpy() if torch.is_tensor(body_rec) else np.asarray(body_rec)
        ce = camera_ext.detach().cpu().numpy() if torch.is_tensor(camera_ext) else np.asarray(camera_ext)
        return io.save_result(br, scale, ce, fit_path)

    def close(self):
        if getattr(self, "ctx", None) is not None:
            self.ctx.close()
            self.ctx = None
