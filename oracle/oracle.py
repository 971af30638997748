"""TEST INFRASTRUCTURE: ctypes wrapper over oracle/libgs_oracle.so (the CPU
restatement of the reference hot path, oracle/gs_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is the checker, never the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("GS_ORACLE_LIB") or os.path.join(_HERE, "libgs_oracle.so")   # (GS_ORACLE_LIB: the sanitizer build, tests only)


def build_sanitized():
    """oracle/libgs_oracle_san.so: gs_oracle.c under -fsanitize=address,undefined (loaded by a child process with libasan preloaded)"""
    out = os.path.join(_HERE, "libgs_oracle_san.so")
    src = os.path.join(_HERE, "gs_oracle.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgs_oracle_san.so"])
    return out


def build(force=False):
    src = os.path.join(_HERE, "gs_oracle.c")
    if os.environ.get("GS_ORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def build_gl_ref():
    """oracle/_ref/gl_ref: the harness that draws the reference's own shaders on Mesa (oracle/gl_ref.c).  Only where the
    reference is present (the build container): its output -- tests/golden/gl_*.bin -- is what travels."""
    src, out = os.path.join(_HERE, "gl_ref.c"), os.path.join(_HERE, "_ref", "gl_ref")
    if not os.path.exists("/root/reference/index.js") or not os.path.exists("/usr/include/GL/internal/dri_interface.h"):
        return None
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-Wall", "-o", out, src, "-ldl"])
    return out


class ProjT(C.Structure):
    _fields_ = [("visible", C.c_int32)] + [(n, C.c_float) for n in (
        "cx", "cy", "ax", "ay", "bx", "by", "v1x", "v1y", "v2x", "v2y", "zndc", "r", "g", "b", "alpha")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.gso_sort.restype = C.c_size_t
        L.gso_sort.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gso_pack.restype = None
        L.gso_pack.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gso_model_view.argtypes = [C.c_void_p] * 3
        L.gso_projection.argtypes = [C.c_void_p] * 2
        L.gso_tick.argtypes = [C.c_void_p] * 5
        L.gso_focal.restype = C.c_double
        L.gso_focal.argtypes = [C.c_void_p, C.c_double]
        L.gso_project.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                  C.c_float, C.POINTER(ProjT)]
        L.gso_render.restype = C.c_int
        L.gso_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_float,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gso_render_scene.restype = C.c_int
        L.gso_render_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_float,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
        L.gso_js_exp.restype = C.c_double
        L.gso_js_exp.argtypes = [C.c_double]
        L.gso_ply_to_splat.restype = C.c_int
        L.gso_ply_to_splat.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def sort(rows, view, cutout=None):
    """rows: [N,4] (x,y,z,size) or [N,16] worker matrices (f32).  -> uint32[V]"""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    if rows.ndim == 1:
        rows = rows.reshape(-1, 4)
    n, stride = rows.shape
    view = np.ascontiguousarray(view, dtype=np.float32)
    cut = None if cutout is None else np.ascontiguousarray(cutout, dtype=np.float32)
    out = np.zeros(max(n, 1), dtype=np.uint32)
    ptr = C.c_void_p(rows.ctypes.data + (12 * 4 if stride == 16 else 0))
    v = lib().gso_sort(ptr, n, stride, _p(view), _p(cut), _p(out))
    return out[:v].copy()


def order_sum(idx):
    """Position-sensitive checksum of a sorted-index array: sum of value * (position + 1) mod 2^32 (oracle/worker_sort.js
    prints the same figure for the JS restatement)."""
    idx = np.asarray(idx, dtype=np.uint64)
    return int((idx * np.arange(1, idx.size + 1, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).sum() & np.uint64(0xFFFFFFFF))


def pack(rows_bytes):
    rows = np.ascontiguousarray(np.frombuffer(bytes(rows_bytes), dtype=np.uint8))
    n = rows.size // 32
    cs = np.zeros((n, 4), np.float32)
    cc = np.zeros((n, 4), np.uint32)
    mats = np.zeros((n, 16), np.float32)
    lib().gso_pack(_p(rows), n, _p(cs), _p(cc), _p(mats))
    return cs, cc, mats


def model_view(cam_world, obj_world):
    a = np.ascontiguousarray(cam_world, np.float64)
    b = np.ascontiguousarray(obj_world, np.float64)
    o = np.zeros(16, np.float64)
    lib().gso_model_view(_p(a), _p(b), _p(o))
    return o


def projection(proj):
    a = np.ascontiguousarray(proj, np.float64)
    o = np.zeros(16, np.float64)
    lib().gso_projection(_p(a), _p(o))
    return o


def tick(cam_world, obj_world, cutout_world=None):
    a = np.ascontiguousarray(cam_world, np.float64)
    b = np.ascontiguousarray(obj_world, np.float64)
    c = None if cutout_world is None else np.ascontiguousarray(cutout_world, np.float64)
    view = np.zeros(4, np.float32)
    cut = np.zeros(16, np.float32)
    lib().gso_tick(_p(a), _p(b), _p(c), _p(view), _p(cut))
    return view, (cut if c is not None else None)


def focal(gs_proj, vh):
    a = np.ascontiguousarray(gs_proj, np.float64)
    return lib().gso_focal(_p(a), float(vh))


def project(cs, cc, idx, mv, proj, focal_, vw, vh):
    cs = np.ascontiguousarray(cs, np.float32)
    cc = np.ascontiguousarray(cc, np.uint32)
    mv = np.ascontiguousarray(mv, np.float32)
    proj = np.ascontiguousarray(proj, np.float32)
    o = ProjT()
    lib().gso_project(_p(cs), _p(cc), int(idx), _p(mv), _p(proj), float(np.float32(focal_)), float(vw), float(vh),
                      C.byref(o))
    return o


def render(cs, cc, sorted_idx, mv, proj, focal_, W, H, x0=0, x1=None, bg=(0, 0, 0, 1), want_f32=True, scene_depth=None,
           scene_rgba=None):
    """-> (rgba8 [H,SW,4] top-down, f32 image or None, fragment count).  scene_depth [H,W] f32 window depth and
    scene_rgba [H,W,4] u8 are the opaque scene the splats are depth-tested against / composited over."""
    x1 = W if x1 is None else x1
    cs = np.ascontiguousarray(cs, np.float32)
    cc = np.ascontiguousarray(cc, np.uint32)
    si = np.ascontiguousarray(sorted_idx, np.uint32)
    mv = np.ascontiguousarray(mv, np.float32)
    proj = np.ascontiguousarray(proj, np.float32)
    bg = np.ascontiguousarray(bg, np.float32)
    sw = x1 - x0
    u8 = np.zeros((H, sw, 4), np.uint8)
    f32 = np.zeros((H, sw, 4), np.float32) if want_f32 else None
    fr = C.c_uint64(0)
    sd = None if scene_depth is None else np.ascontiguousarray(scene_depth, np.float32)
    sc = None if scene_rgba is None else np.ascontiguousarray(scene_rgba, np.uint8)
    rc = lib().gso_render_scene(_p(cs), _p(cc), _p(si), si.size, _p(mv), _p(proj), float(np.float32(focal_)), W, H, x0, x1,
                                _p(bg), _p(sd), _p(sc), _p(f32), _p(u8), C.byref(fr))
    if rc != 0:
        raise MemoryError("gso_render")
    return u8, f32, fr.value


def js_exp(x):
    """Math.exp as V8 evaluates it (fdlibm e_exp.c), elementwise over a float64 array."""
    x = np.ascontiguousarray(x, np.float64)
    f = lib().gso_js_exp
    return np.array([f(float(v)) for v in x.ravel()], np.float64).reshape(x.shape)


class PlyError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def ply_to_splat(ply_bytes):
    buf = np.frombuffer(bytes(ply_bytes), dtype=np.uint8)
    n = C.c_size_t(0)
    err = C.create_string_buffer(256)
    rc = lib().gso_ply_to_splat(_p(buf), buf.size, None, C.byref(n), err, 256)
    if rc != 0:
        raise PlyError(rc, err.value.decode())
    out = np.zeros(n.value * 32, np.uint8)
    rc = lib().gso_ply_to_splat(_p(buf), buf.size, _p(out), C.byref(n), err, 256)
    if rc != 0:
        raise PlyError(rc, err.value.decode())
    return out
