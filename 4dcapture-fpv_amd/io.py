"""File boundary of the hot path: body_gen -> smoothed_body pickles, camerapose.txt, scene.

Mirrors /root/reference/global_optimization.py:64-76 (body_params_parse), :688-707 (loader),
:208-230 (camerapose parsing), :637-653 + cvae.py:189-208 (save_result schema as consumed by
global_vis.py:116-129 / local_vis.py:309-313).  Host-side only; numpy, no torch.
"""
from __future__ import annotations

import glob
import json
import os
import pickle
import struct

import numpy as np

PARAM_KEYS = ("transl", "global_orient", "betas", "body_pose", "left_hand_pose",
              "right_hand_pose", "camera_translation")
PARAM_DIMS = (3, 3, 10, 32, 12, 12, 3)
# column ranges of the [N,75] layout and of the optimised [N,78] layout (A2)
SLICES_75 = {"transl": (0, 3), "global_orient": (3, 6), "betas": (6, 16), "body_pose": (16, 48),
             "left_hand_pose": (48, 60), "right_hand_pose": (60, 72), "camera_translation": (72, 75)}


def body_params_parse(body_params_batch: dict) -> np.ndarray:
    """global_optimization.py:64-76: concatenate the 7 keys in this order; extra keys ignored."""
    return np.concatenate([np.asarray(body_params_batch[k]) for k in PARAM_KEYS], axis=-1)


def load_body_gen(body_path: str) -> np.ndarray:
    """global_optimization.py:688-707: sorted glob of <body_path>/results/*/*.pkl -> [N,75] fp32."""
    files = sorted(glob.glob(os.path.join(body_path, "results/*/*.pkl")))
    if not files:
        raise FileNotFoundError(f"no SMPLify-X results under {body_path}/results/*/*.pkl")
    rows = []
    for fn in files:
        with open(fn, "rb") as f:
            try:
                d = pickle.load(f)
            except UnicodeDecodeError:
                f.seek(0)
                d = pickle.load(f, encoding="latin1")
        rows.append(body_params_parse(d))
    return np.vstack(rows).astype(np.float32)


def load_dct_base(mat_path=None, num_frames: int = 60, num_coef: int = 5) -> np.ndarray:
    """global_optimization.py:131-136: `loadmat(DCT_MAT_PATH)['D'][:DCT_NUM].T` -> [60,5].  The .mat
    (../Data/DCT_Basis/60.mat, :45) is not part of the repository; without it the orthonormal DCT-II
    basis of the same shape is generated (row k = sqrt((1 or 2)/T) cos(pi (n + 1/2) k / T))."""
    if mat_path and os.path.exists(mat_path):
        import scipy.io as sio
        mtx = sio.loadmat(mat_path, squeeze_me=True, struct_as_record=False)["D"]
        return np.ascontiguousarray(np.array(mtx[:num_coef]).T, dtype=np.float32)
    n = np.arange(num_frames)
    D = np.zeros((num_coef, num_frames))
    for k in range(num_coef):
        a = np.sqrt(1.0 / num_frames) if k == 0 else np.sqrt(2.0 / num_frames)
        D[k] = a * np.cos(np.pi * (n + 0.5) * k / num_frames)
    return np.ascontiguousarray(D.T, dtype=np.float32)


def body_params_encapsulate(body_rec: np.ndarray, scale, camera_ext: np.ndarray) -> list:
    """The 3-argument form save_result calls (:644) but cvae.py never defines; the 1-argument
    version (cvae.py:189-208) gives the slicing, the consumers give the two extra keys."""
    body_rec = np.asarray(body_rec, dtype=np.float32)
    camera_ext = np.asarray(camera_ext, dtype=np.float32)
    out = []
    for b in range(body_rec.shape[0]):
        d = {k: body_rec[b:b + 1, lo:hi] for k, (lo, hi) in SLICES_75.items()}
        d["scale"] = np.float32(scale)
        d["camera_ext"] = camera_ext[b]
        out.append(d)
    return out


def save_result(body_rec, scale, camera_ext, fit_path: str) -> list:
    """global_optimization.py:637-653: <fit_path>/body_gen_%06d.pkl, index from 0."""
    os.makedirs(fit_path, exist_ok=True)
    files = []
    for i, d in enumerate(body_params_encapsulate(body_rec, scale, camera_ext)):
        fn = fit_path + "/body_gen_" + str(i).zfill(6) + ".pkl"
        with open(fn, "wb") as f:
            pickle.dump(d, f)
        files.append(fn)
    return files


def write_body_gen(body_params: np.ndarray, body_path: str) -> None:
    """Inverse of load_body_gen: lay a [N,75] array out as SMPLify-X result folders."""
    for i, row in enumerate(np.asarray(body_params, dtype=np.float32)):
        d = os.path.join(body_path, "results", "frame_%06d" % i)
        os.makedirs(d, exist_ok=True)
        rec = {k: row[None, lo:hi].copy() for k, (lo, hi) in SLICES_75.items()}
        with open(os.path.join(d, "000.pkl"), "wb") as f:
            pickle.dump(rec, f)


def qvec2rotmat(q) -> np.ndarray:
    """COLMAP quaternion (w,x,y,z) -> rotation (global_optimization.py:51-61), float64."""
    w, x, y, z = (float(v) for v in q)
    return np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
        [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def read_camerapose(path_or_lines) -> np.ndarray:
    """global_optimization.py:208-230: each ' qw qx qy qz tx ty tz' line (leading space, so the
    fields are items[1:8]) is a world->camera pose; returns its inverse, [N,4,4] fp32 computed
    in float64."""
    if isinstance(path_or_lines, (str, os.PathLike)):
        with open(path_or_lines) as f:
            lines = [ln.rstrip("\n") for ln in f]
    else:
        lines = [str(ln).rstrip("\n") for ln in path_or_lines]
    out = np.zeros((len(lines), 4, 4), dtype=np.float64)
    for i, line in enumerate(lines):
        items = line.split(" ")
        E = np.eye(4)
        E[:3, :3] = qvec2rotmat([float(v) for v in items[1:5]])
        E[:3, 3] = [float(v) for v in items[5:8]]
        out[i] = np.linalg.inv(E)
    return out.astype(np.float32)


def read_contact_ids(folder: str, parts=("L_Leg", "R_Leg")) -> np.ndarray:
    """global_optimization.py:79-94: per part `list(set(verts_ind))`, parts concatenated.  The
    reference re-reads the JSON files every iteration; once is enough."""
    ids = []
    for part in parts:
        with open(os.path.join(folder, part + ".json")) as f:
            data = json.load(f)
        ids.append(np.array(list(set(data["verts_ind"])), dtype=np.int64))
    return np.concatenate(ids)


_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h",
              "ushort": "H", "uint16": "H", "int": "i", "int32": "i", "uint": "I", "uint32": "I",
              "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_scene_points(path: str) -> np.ndarray:
    """Vertex positions of the scene: PLY (ascii / binary_little_endian / binary_big_endian;
    stands in for o3d.io.read_triangle_mesh(...).vertices, :173-174), or .xyz / .txt rows
    (utils/pointcloud_helper.py output), or .npy."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path).astype(np.float32).reshape(-1, 3)
    if ext in (".xyz", ".txt"):
        return np.loadtxt(path, dtype=np.float64, usecols=(0, 1, 2)).astype(np.float32).reshape(-1, 3)
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        nvert = 0
        props = []
        in_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    nvert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list property on vertex element is unsupported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        ix, iy, iz = names.index("x"), names.index("y"), names.index("z")
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(nvert)]
            arr = np.array([[float(r[ix]), float(r[iy]), float(r[iz])] for r in rows],
                           dtype=np.float64).reshape(-1, 3)
            return arr.astype(np.float32)
        endian = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, endian + t) for n, t in props])
        raw = np.frombuffer(f.read(dt.itemsize * nvert), dtype=dt, count=nvert)
        return np.stack([raw["x"], raw["y"], raw["z"]], 1).astype(np.float32)


def write_ply_points(path: str, pts: np.ndarray, binary: bool = True) -> None:
    pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
    with open(path, "wb") as f:
        fmt = "binary_little_endian" if binary else "ascii"
        f.write((f"ply\nformat {fmt} 1.0\nelement vertex {len(pts)}\nproperty float x\n"
                 "property float y\nproperty float z\nelement face 0\n"
                 "property list uchar int vertex_indices\nend_header\n").encode())
        if binary:
            f.write(struct.pack("<%df" % pts.size, *pts.ravel().tolist()) if pts.size < 4096
                    else pts.astype("<f4").tobytes())
        else:
            for p in pts:
                f.write(("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2]))).encode())


def load_smoothed_body(fit_path: str):
    """Read <fit_path>/body_gen_%06d.pkl back (what global_vis.py:105-124 iterates over):
    -> (body_rec [N,75], scale, camera_ext [N,4,4])."""
    files = sorted(glob.glob(os.path.join(fit_path, "body_gen_*.pkl")))
    if not files:
        raise FileNotFoundError(f"no body_gen_*.pkl under {fit_path}")
    rows, cams, scale = [], [], None
    for fn in files:
        with open(fn, "rb") as f:
            d = pickle.load(f)
        rows.append(body_params_parse(d))
        cams.append(np.asarray(d["camera_ext"], dtype=np.float32).reshape(4, 4))
        scale = np.float32(d["scale"])
    return np.vstack(rows).astype(np.float32), scale, np.stack(cams)


def colmap_images_to_camerapose(images_txt: str, out_path: str | None = None) -> list:
    """utils/camerapose_helper.py:15-29: COLMAP images.txt -> camerapose.txt lines
    ' qw qx qy qz tx ty tz' (first 3 header lines skipped as the reference does, only rows whose
    last field names a jpg)."""
    with open(images_txt) as f:
        lines = [ln.rstrip("\n") for ln in f][3:]
    out = []
    for line in lines:
        items = line.split(" ")
        if "jpg" in items[-1]:
            out.append(" " + " ".join(items[1:8]))
    if out_path:
        with open(out_path, "w") as f:
            f.write("\n".join(out) + ("\n" if out else ""))
    return out


def colmap_points_to_xyz(points3d_txt: str, out_path: str | None = None) -> np.ndarray:
    """utils/pointcloud_helper.py:15-27: COLMAP points3D.txt -> ' x y z r g b' rows; returns xyz [n,3]."""
    with open(points3d_txt) as f:
        lines = [ln.rstrip("\n") for ln in f][3:]
    rows = [ln.split(" ") for ln in lines if ln.strip()]
    if out_path:
        with open(out_path, "w") as f:
            for it in rows:
                f.write(" " + " ".join(it[1:7]) + "\n")
    return np.array([[float(it[1]), float(it[2]), float(it[3])] for it in rows], dtype=np.float32).reshape(-1, 3)
