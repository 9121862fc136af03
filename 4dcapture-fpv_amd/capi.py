"""ctypes binding of include/fdcap.h (libfdcap_hip.so, built in-tree by __graft_entry__.build()).

The library is the product: if it is missing or cannot be loaded this module raises; there is
no CPU fallback anywhere in the package.  torch is used only to own device memory and streams:
tensors cross the boundary as raw device pointers."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDCAP_LIB") or os.path.join(_HERE, "libfdcap_hip.so")

NUM_LOSSES = 8
XDIM = 78
PDIM = 75


class FdcapError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [("num_verts", c_int32), ("v_template", POINTER(c_float)), ("shapedirs", POINTER(c_float)),
                ("num_shape", c_int32), ("posedirs", POINTER(c_float)), ("J_regressor", POINTER(c_float)),
                ("parents", POINTER(c_int32)), ("lbs_weights", POINTER(c_float)),
                ("hands_componentsl", POINTER(c_float)), ("hands_componentsr", POINTER(c_float)),
                ("hands_meanl", POINTER(c_float)), ("hands_meanr", POINTER(c_float)),
                ("vp_fc1_w", POINTER(c_float)), ("vp_fc1_b", POINTER(c_float)), ("vp_fc2_w", POINTER(c_float)),
                ("vp_fc2_b", POINTER(c_float)), ("vp_out_w", POINTER(c_float)), ("vp_out_b", POINTER(c_float))]


class Fit2dStage(ctypes.Structure):
    _fields_ = [("fx", c_float), ("fy", c_float), ("cx", c_float), ("cy", c_float), ("rho", c_float),
                ("w_data", c_float), ("w_pose", c_float), ("w_shape", c_float), ("w_hand", c_float)]


class LbfgsConfig(ctypes.Structure):
    """include/fdcap.h fdcap_lbfgs_config: torch.optim.LBFGS's arguments + SMPLify-X's loop around optimizer.step()."""
    _fields_ = [("dim", c_int32), ("history", c_int32), ("max_iter", c_int32), ("max_eval", c_int32), ("max_steps", c_int32),
                ("max_ls", c_int32), ("lr", c_float), ("tolerance_grad", c_float), ("tolerance_change", c_float),
                ("ftol", c_float), ("gtol", c_float)]


class OptConfig(ctypes.Structure):
    _fields_ = [("n_total", c_int32), ("n_local", c_int32), ("frame0", c_int32), ("lr", c_float),
                ("weight_loss_rec", c_float), ("weight_loss_vposer", c_float), ("weight_contact", c_float),
                ("phase1_contact", c_float), ("phase1_smooth", c_float), ("phase2_world", c_float),
                ("phase2_smooth", c_float), ("scale_init", c_float), ("legacy_zero_grad", c_int32)]


# every symbol include/fdcap.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "fdcap_version": (c_char_p, []),
    "fdcap_build_info": (c_char_p, []),
    "fdcap_ctx_create": (c_int32, [POINTER(ModelDesc), POINTER(c_void_p)]),
    "fdcap_ctx_destroy": (None, [c_void_p]),
    "fdcap_set_scene": (c_int32, [c_void_p, c_void_p, c_int64]),
    "fdcap_debug_scene_hash": (c_int32, [c_void_p, c_void_p]),
    "fdcap_debug_kernel_forms": (c_int32, [c_void_p, c_int32, c_int32]),
    "fdcap_set_contact_ids": (c_int32, [c_void_p, c_void_p, c_int32]),
    "fdcap_chamfer_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_chamfer_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "fdcap_chamfer_fwd_scene": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "fdcap_chamfer_bwd_scene": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_set_nn_kernel": (c_int32, [c_int32]),
    "fdcap_vposer_decode": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "fdcap_vposer_decode_bwd": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_smplx_backward": (c_int32, [c_void_p] * 7 + [c_int32] + [c_void_p] * 9),
    "fdcap_body_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "fdcap_world_mesh": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_params_75_to_78": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p]),
    "fdcap_params_78_to_75": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p]),
    "fdcap_smplx_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                       c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_create": (c_int32, [c_void_p, POINTER(OptConfig), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_set_inputs": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_backward": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "fdcap_opt_step": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "fdcap_opt_backward_and_step": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "fdcap_opt_sync": (c_int32, [c_void_p, c_void_p]),
    "fdcap_opt_run": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "fdcap_comm_unique_id": (c_int32, [c_void_p]),
    "fdcap_comm_create": (c_int32, [c_void_p, c_void_p, c_int32, c_int32]),
    "fdcap_comm_destroy": (c_int32, [c_void_p]),
    "fdcap_comm_last_error": (c_char_p, [c_void_p]),
    "fdcap_opt_halo_exchange": (c_int32, [c_void_p, c_void_p]),
    "fdcap_opt_exchange": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "fdcap_opt_time_exchange": (c_int32, [c_void_p, c_int32, POINTER(c_float), POINTER(c_float), c_void_p]),
    "fdcap_comm_allreduce_f64": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "fdcap_opt_state_len": (c_int32, [c_void_p]),
    "fdcap_opt_export_state": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_import_state": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_check_finite": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_set_loss_output": (c_int32, [c_void_p, c_void_p]),
    "fdcap_opt_detect_contact": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p]),
    "fdcap_opt_backward_local2": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "fdcap_opt_step_x": (c_int32, [c_void_p, c_int32, c_void_p]),
    "fdcap_opt_set_dct": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "fdcap_opt_dct_fit": (c_int32, [c_void_p, c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p]),
    "fdcap_opt_backward_dct": (c_int32, [c_void_p, c_float, c_float, c_float, c_int32, c_void_p]),
    "fdcap_opt_set_dct_coef": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_get_dct": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_get_dct_state": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_set_dct_state": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_dct_windows": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "fdcap_frame_smoother": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_float, c_float,
                                        c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "fdcap_opt_set_keypoints": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_backward_fit2d": (c_int32, [c_void_p, POINTER(Fit2dStage), c_int32, c_void_p]),
    "fdcap_opt_reset_adam": (c_int32, [c_void_p, c_void_p]),
    "fdcap_lbfgs_create": (c_int32, [c_int32, POINTER(LbfgsConfig), POINTER(c_void_p)]),
    "fdcap_lbfgs_destroy": (None, [c_void_p]),
    "fdcap_lbfgs_reset": (c_int32, [c_void_p, c_void_p]),
    "fdcap_lbfgs_advance": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "fdcap_lbfgs_get_stats": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_lbfgs_finalize": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "fdcap_opt_fit2d_lbfgs": (c_int32, [c_void_p, POINTER(Fit2dStage), POINTER(LbfgsConfig), c_int32, POINTER(c_int32), c_void_p]),
    "fdcap_opt_fit2d_lbfgs_stats": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_step_rows_and_pack": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "fdcap_opt_unpack_and_step_scale": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "fdcap_opt_forward_ahead": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "fdcap_exchange_len": (c_int32, []),
    "fdcap_opt_get_results": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_destroy": (None, [c_void_p]),
    "fdcap_opt_forward_world": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_get_contact": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_get_grads": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdcap_panel_gemm": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int32,
                                   c_void_p]),
    "fdcap_time_blend_gemm": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_float), c_void_p]),
    "fdcap_opt_nn_timing": (c_int32, [c_void_p, c_int32]),
    "fdcap_opt_nn_timing_read": (c_int32, [c_void_p, POINTER(c_float), POINTER(c_int32)]),
    "fdcap_opt_launch_timing": (c_int32, [c_void_p, c_int32]),
    "fdcap_opt_launch_timing_read": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "fdcap_opt_time_chamfer": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_float), c_void_p]),
}

_lib = None


def load_library(path: str | None = None) -> ctypes.CDLL:
    """Load libfdcap_hip.so and bind every declared symbol.  Raises FdcapError when absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FdcapError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    info = lib.fdcap_build_info().decode()
    if "packed_fp32=off" not in info and os.environ.get("FDCAP_ALLOW_PK_F32") != "1":
        raise FdcapError(f"{p} was built with packed fp32 instructions ({info}): results are not reliable next to another kernel's "
                         "MFMAs on the same CU (NOTES.md section 6).  Rebuild with __graft_entry__.build(), or set "
                         "FDCAP_ALLOW_PK_F32=1 for an instrumentation variant.")
    if path is None:
        _lib = lib
    return lib


def count_packed_fp32(path: str | None = None) -> int:
    """Number of v_pk_{fma,mul,add}_f32 instructions in the library's gfx950 code object (llvm-objdump of the extracted
    bundle, ~1 s).  The build requirement of csrc/fdcap.hip says 0; __graft_entry__.build() and tests/test_io_and_abi.py check."""
    import re
    import shutil
    import subprocess
    import tempfile
    objdump = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not os.path.exists(objdump):
        raise FdcapError(f"{objdump} not found (set LLVM_OBJDUMP)")
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(path or LIB_PATH, so)
        subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=tmp)     # writes lib.so.<k>.<triple>
        objs = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not objs:
            raise FdcapError("no gfx950 code object found in the library")
        n = 0
        for f in objs:
            txt = subprocess.run([objdump, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            n += len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", txt))
        return n


def check(code: int, what: str) -> None:
    if code == 0:
        return
    if code < 0:
        names = {-1: "FDCAP_E_ARG", -2: "FDCAP_E_STATE", -3: "FDCAP_E_NODEVICE", -4: "FDCAP_E_COMM (RCCL)"}
        raise FdcapError(f"{what}: {names.get(code, code)}")
    raise FdcapError(f"{what}: hipError_t {code}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def dptr(t):
    """Raw device pointer of a contiguous CUDA/HIP torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise FdcapError("expected a device tensor (the HIP path has no host fallback)")
    if not t.is_contiguous():
        raise FdcapError("expected a contiguous tensor")
    return c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class Context:
    """Owns one fdcap_ctx: body-model + VPoser constants on the current device."""

    def __init__(self, body_model, vposer):
        self.lib = load_library()
        bm, vp = body_model, vposer
        shapedirs = _f32(bm.shapedirs)
        if shapedirs.ndim != 3 or shapedirs.shape[2] < 10:
            raise FdcapError("shapedirs must be [V,3,>=10]")
        keep = dict(
            v_template=_f32(bm.v_template), shapedirs=shapedirs, posedirs=_f32(bm.posedirs),
            J_regressor=_f32(bm.J_regressor), parents=np.ascontiguousarray(bm.parents, dtype=np.int32),
            lbs_weights=_f32(bm.lbs_weights), hcl=_f32(bm.hands_componentsl[:12]),
            hcr=_f32(bm.hands_componentsr[:12]), hml=_f32(bm.hands_meanl), hmr=_f32(bm.hands_meanr),
            w1=_f32(vp.fc1_w), b1=_f32(vp.fc1_b), w2=_f32(vp.fc2_w), b2=_f32(vp.fc2_b), w3=_f32(vp.out_w),
            b3=_f32(vp.out_b))
        V = keep["v_template"].shape[0]
        assert keep["posedirs"].shape == (486, 3 * V), keep["posedirs"].shape
        assert keep["J_regressor"].shape == (55, V) and keep["lbs_weights"].shape == (V, 55)
        assert keep["w1"].shape == (512, 32) and keep["w2"].shape == (512, 512) and keep["w3"].shape == (126, 512)
        md = ModelDesc(V, _fp(keep["v_template"]), _fp(keep["shapedirs"]), shapedirs.shape[2], _fp(keep["posedirs"]),
                       _fp(keep["J_regressor"]), keep["parents"].ctypes.data_as(POINTER(c_int32)),
                       _fp(keep["lbs_weights"]), _fp(keep["hcl"]), _fp(keep["hcr"]), _fp(keep["hml"]), _fp(keep["hmr"]),
                       _fp(keep["w1"]), _fp(keep["b1"]), _fp(keep["w2"]), _fp(keep["b2"]), _fp(keep["w3"]), _fp(keep["b3"]))
        h = c_void_p()
        check(self.lib.fdcap_ctx_create(ctypes.byref(md), ctypes.byref(h)), "fdcap_ctx_create")
        self.handle = h
        self.num_verts = V
        self.num_scene = 0
        self.num_contact = 0

    def set_scene(self, scene_xyz):
        s = _f32(np.asarray(scene_xyz).reshape(-1, 3))
        check(self.lib.fdcap_set_scene(self.handle, s.ctypes.data_as(c_void_p), s.shape[0]), "fdcap_set_scene")
        self.num_scene = s.shape[0]
        self._scene_host = s                   # (ops.chamferDist recognises a target tensor that holds exactly these points)
        self._scene_dev = None
        self._scene_gen = getattr(self, "_scene_gen", 0) + 1     # ... and forgets what it concluded about an earlier scene

    def set_contact_ids(self, vid):
        v = np.ascontiguousarray(vid, dtype=np.int64)
        check(self.lib.fdcap_set_contact_ids(self.handle, v.ctypes.data_as(c_void_p), v.shape[0]),
              "fdcap_set_contact_ids")
        self.num_contact = v.shape[0]

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.fdcap_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
