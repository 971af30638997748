"""ctypes binding of the C ABI (include/gs_splat.h -> csrc/libgs_splat_hip.so).

Plumbing only: every call goes straight into the HIP library.  There is no Python or CPU fallback -- if the
library is missing, or no GPU is usable, this raises.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import build as _build

GS_OK = 0
E_BADARG, E_PLY_HEADER, E_PLY_PROP, E_HIP, E_OOM, E_NODEVICE, E_STATE, E_PLY_DATA, E_RETRY = -1, -2, -3, -4, -5, -6, -7, -8, -9
RENDER_FLIP_Y, RENDER_COUNT_FRAGS, RENDER_NO_EARLY_OUT, RENDER_ASYNC, RENDER_COUNT_EVALUATED = 1, 2, 4, 8, 16
OPT_PROFILE, OPT_TERMINATION, OPT_NEAR_PERMILLE, OPT_RECORD_STAGED, OPT_PIPELINE_DEPTH, OPT_WIDE_PAIRS, OPT_ENQUEUE_THREADS = 1, 2, 3, 4, 5, 6, 7
OPT_COMM_SELF_COPY = 8
OPT_BLEND_SPLIT = 9
OPT_FRAME_BATCH = 10
OPT_SORT_NEAR = 11
OPT_COMM_TRANSPORT = 12
OPT_AUTO_RETRY = 13
OPT_SORT_SHARE = 15
OPT_BINNING = 16
OPT_SUBTILE = 17
TRANSPORT_RCCL, TRANSPORT_INPROC = 0, 1
COMM_ID_BYTES = 128
BUF_CENTER_SCALE, BUF_COV_COLOR, BUF_SORT_ROWS, BUF_SORTED, BUF_PROJECTED, BUF_TILE_COUNT, BUF_TILE_STATS, BUF_UNSAT_MASK = 0, 1, 2, 3, 4, 5, 6, 7

EXPORTS = [
    "gs_create", "gs_destroy", "gs_last_error", "gs_version", "gs_device_count", "gs_clear", "gs_push_splat", "gs_push_matrices", "gs_load_ply",
    "gs_ply_to_splat", "gs_ply_to_splat_gpu", "gs_count", "gs_sort", "gs_render", "gs_render_device", "gs_render_stereo", "gs_set_scene", "gs_sync",
    "gs_set_stream", "gs_frame_stream", "gs_frame_status_device", "gs_frame_lane", "gs_lane_stream", "gs_wait_stream", "gs_stream_wait_frame",
    "gs_model_view_matrix", "gs_projection_matrix", "gs_tick_uniforms", "gs_focal", "gs_scaled_size", "gs_set_option",
    "gs_get_stats", "gs_download",
    "gs_comm_unique_id", "gs_comm_init", "gs_comm_destroy", "gs_partition", "gs_render_gathered", "gs_read_gathered",
    "gs_host_alloc", "gs_host_free", "gs_sort_for", "gs_sort_gathered", "gs_gathered_size", "gs_sort_begin", "gs_sort_poll",
    "gs_create_multi", "gs_multi_destroy", "gs_multi_last_error", "gs_multi_devices", "gs_multi_ctx", "gs_multi_clear",
    "gs_multi_push_splat", "gs_multi_load_ply", "gs_multi_count", "gs_multi_set_option", "gs_multi_sort", "gs_multi_render",
    "gs_multi_render_device", "gs_multi_read", "gs_multi_sync",
]


class Piece(C.Structure):
    _fields_ = [("view", C.c_int32), ("x0", C.c_int32), ("x1", C.c_int32), ("owner", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("model_view", C.c_float * 16), ("projection", C.c_float * 16), ("fb_width", C.c_int32), ("fb_height", C.c_int32),
                ("x0", C.c_int32), ("x1", C.c_int32), ("focal", C.c_float), ("background", C.c_float * 4), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_splats", C.c_uint64), ("n_sorted", C.c_uint64), ("n_visible", C.c_uint64), ("n_pairs", C.c_uint64),
                ("n_frags", C.c_uint64), ("n_tiles", C.c_uint64), ("ms_sort", C.c_float), ("ms_project", C.c_float),
                ("ms_bin", C.c_float), ("ms_blend", C.c_float), ("ms_render", C.c_float), ("blend_launches", C.c_uint32),
                ("prof_frames", C.c_uint32), ("sum_ms_sort", C.c_float), ("sum_ms_project", C.c_float), ("sum_ms_bin", C.c_float),
                ("sum_ms_blend", C.c_float), ("acc_frames", C.c_uint64), ("acc_sorted", C.c_uint64), ("acc_visible", C.c_uint64),
                ("acc_pairs", C.c_uint64), ("unsat_tiles", C.c_uint32), ("near_permille", C.c_uint32),
                ("sort_records", C.c_uint32), ("retried_frames", C.c_uint32), ("spec_sorts", C.c_uint32), ("spec_misses", C.c_uint32), ("need_splats", C.c_uint32),
                ("sort_mode", C.c_uint32), ("subtile", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class GsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gs_splat error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


_lib = None


def library_path():
    return _build.LIB


def _hip_runtime_mapped():
    try:
        with open("/proc/self/maps") as f:
            return any("libamdhip64" in line for line in f)
    except OSError:
        return False


def hip_runtime():
    """ctypes handle of the HIP runtime THIS process has mapped (the one libgs_splat_hip.so talks to): for helpers that call
    hipMalloc / hipMemcpy next to the library (tests, bench.py).  dlopen("libamdhip64.so") by NAME may pick another copy than
    the one already loaded by path (torch's wheel and /opt/rocm both ship one), and memory from one is no use to the other."""
    load()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line and "/" in line:
                    return C.CDLL(line[line.index("/"):].strip())
    except OSError:
        pass
    return C.CDLL("libamdhip64.so")


def load(build_if_missing=True):
    """dlopen the HIP library (building it first if absent and a compiler is present)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_build.LIB):
        if not build_if_missing:
            raise FileNotFoundError(_build.LIB + " not built; run __graft_entry__.build()")
        _build.build_lib()
    # A process that also uses torch must have torch's HIP runtime loaded first: the wheel ships its own libamdhip64 /
    # libhsa-runtime64 under the same sonames as /opt/rocm's, the first one loaded serves both, and torch finds no GPU through the
    # other one ("No HIP GPUs are available").  Loading that ONE shared object is enough -- importing torch (1-2 s, hundreds of MB)
    # is not a ctypes loader's business; nothing happens when no torch is installed, when a HIP runtime is already mapped, or with
    # GS_SPLAT_NO_TORCH_PRELOAD set.
    if "torch" not in sys.modules and os.environ.get("GS_SPLAT_NO_TORCH_PRELOAD") is None and not _hip_runtime_mapped():
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
            if cand and os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except (ImportError, OSError, ValueError):
            pass
    # GS_SPLAT_LIB: load another build of the same library (A/B measurements of kernel variants); never a different backend
    L = C.CDLL(os.environ.get("GS_SPLAT_LIB") or _build.LIB)
    vp, sz, i32, u32p, f32 = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.c_float
    L.gs_create.argtypes = [i32, C.POINTER(vp)]
    L.gs_destroy.argtypes = [vp]
    L.gs_last_error.argtypes = [vp]; L.gs_last_error.restype = C.c_char_p
    L.gs_version.restype = C.c_uint32
    L.gs_clear.argtypes = [vp]
    L.gs_push_splat.argtypes = [vp, vp, sz]
    L.gs_push_matrices.argtypes = [vp, vp, sz]
    L.gs_load_ply.argtypes = [vp, vp, sz]
    L.gs_ply_to_splat.argtypes = [vp, sz, vp, C.POINTER(sz), C.c_char_p, sz]
    L.gs_ply_to_splat_gpu.argtypes = [vp, vp, sz, vp, C.POINTER(sz)]
    L.gs_count.argtypes = [vp]; L.gs_count.restype = sz
    L.gs_sort.argtypes = [vp, vp, vp, vp, u32p]
    L.gs_sort_begin.argtypes = [vp, vp, vp]
    L.gs_sort_poll.argtypes = [vp, i32, vp, u32p, C.POINTER(C.c_int)]
    L.gs_render.argtypes = [vp, C.POINTER(RenderParams), vp, sz]
    L.gs_render_device.argtypes = [vp, C.POINTER(RenderParams), vp]
    L.gs_render_stereo.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(vp), sz]
    L.gs_set_scene.argtypes = [vp, vp, vp, i32, i32]
    L.gs_sync.argtypes = [vp]
    L.gs_set_stream.argtypes = [vp, vp]
    L.gs_wait_stream.argtypes = [vp, vp]
    L.gs_frame_stream.argtypes = [vp]; L.gs_frame_stream.restype = C.c_void_p
    if hasattr(L, "gs_frame_status_device"):                   # (an older build loaded through GS_SPLAT_LIB for an A/B run has none)
        L.gs_frame_status_device.argtypes = [vp, C.POINTER(C.c_void_p)]; L.gs_frame_status_device.restype = C.c_int
    L.gs_frame_lane.argtypes = [vp]; L.gs_frame_lane.restype = C.c_int
    L.gs_lane_stream.argtypes = [vp, C.c_int]; L.gs_lane_stream.restype = C.c_void_p
    L.gs_stream_wait_frame.argtypes = [vp, vp]
    L.gs_model_view_matrix.argtypes = [vp, vp, vp]; L.gs_model_view_matrix.restype = None
    L.gs_projection_matrix.argtypes = [vp, vp]; L.gs_projection_matrix.restype = None
    L.gs_tick_uniforms.argtypes = [vp, vp, vp, vp, vp]; L.gs_tick_uniforms.restype = None
    L.gs_focal.argtypes = [vp, C.c_double]; L.gs_focal.restype = C.c_double
    L.gs_scaled_size.argtypes = [i32, i32, C.c_double, C.POINTER(i32), C.POINTER(i32)]; L.gs_scaled_size.restype = None
    L.gs_set_option.argtypes = [vp, i32, C.c_int64]
    L.gs_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.gs_download.argtypes = [vp, i32, vp, sz]
    L.gs_comm_unique_id.argtypes = [vp, vp]
    L.gs_comm_init.argtypes = [vp, vp, i32, i32]
    L.gs_comm_destroy.argtypes = [vp]
    L.gs_partition.argtypes = [i32, C.POINTER(i32), i32, C.POINTER(Piece), i32]
    L.gs_render_gathered.argtypes = [vp, C.POINTER(RenderParams), i32, i32, C.POINTER(vp), C.c_uint32]
    L.gs_read_gathered.argtypes = [vp, i32, vp, sz]
    L.gs_sort_for.argtypes = [vp, vp, vp, C.POINTER(RenderParams), vp, u32p]
    L.gs_sort_gathered.argtypes = [vp, vp, vp, C.POINTER(RenderParams), i32]
    L.gs_gathered_size.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.gs_host_alloc.argtypes = [sz]; L.gs_host_alloc.restype = C.c_void_p
    L.gs_host_free.argtypes = [vp]; L.gs_host_free.restype = None
    L.gs_create_multi.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    L.gs_multi_destroy.argtypes = [vp]
    L.gs_multi_last_error.argtypes = [vp]; L.gs_multi_last_error.restype = C.c_char_p
    L.gs_multi_devices.argtypes = [vp]
    L.gs_multi_ctx.argtypes = [vp, i32]; L.gs_multi_ctx.restype = vp
    L.gs_multi_clear.argtypes = [vp]
    L.gs_multi_push_splat.argtypes = [vp, vp, sz]
    L.gs_multi_load_ply.argtypes = [vp, vp, sz]
    L.gs_multi_count.argtypes = [vp]; L.gs_multi_count.restype = sz
    L.gs_multi_set_option.argtypes = [vp, i32, C.c_int64]
    L.gs_multi_sort.argtypes = [vp, vp, vp, C.POINTER(RenderParams), i32]
    L.gs_multi_render.argtypes = [vp, C.POINTER(RenderParams), i32, C.POINTER(vp), sz, C.c_uint32]
    L.gs_multi_render_device.argtypes = [vp, C.POINTER(RenderParams), i32, C.POINTER(vp), C.c_uint32]
    L.gs_multi_read.argtypes = [vp, i32, vp, sz]
    L.gs_multi_sync.argtypes = [vp]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- host helpers (no context needed) ------------------------------------------------------------------

def host_frame(height, width):
    """An H x W x 4 uint8 array over page-locked memory (gs_host_alloc); keep the returned owner alive, free with owner.free()."""
    L = load()
    n = int(height) * int(width) * 4
    p = L.gs_host_alloc(n)
    if not p:
        raise MemoryError("gs_host_alloc(%d)" % n)

    class _Owner:
        def free(self):
            if self.p:
                L.gs_host_free(self.p); self.p = None
    o = _Owner(); o.p = p
    arr = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p)).reshape(int(height), int(width), 4)
    return arr, o


def device_count():
    return int(load().gs_device_count())


def partition(widths, world):
    """gs_partition: [(view, x0, x1, owner rank)] for one view (column strips) or two (XR eyes), in gather order."""
    widths = [int(w) for w in widths]
    arr = (C.c_int * len(widths))(*widths)
    out = (Piece * 128)()
    n = load().gs_partition(len(widths), arr, int(world), out, 128)
    if n < 0:
        raise GsError(n, "gs_partition(%r, world=%d)" % (widths, world))
    return [(out[i].view, out[i].x0, out[i].x1, out[i].owner) for i in range(n)]



def model_view_matrix(cam_world, obj_world):
    a = np.ascontiguousarray(cam_world, np.float64); b = np.ascontiguousarray(obj_world, np.float64); o = np.zeros(16, np.float64)
    load().gs_model_view_matrix(_p(a), _p(b), _p(o))
    return o


def projection_matrix(proj):
    a = np.ascontiguousarray(proj, np.float64); o = np.zeros(16, np.float64)
    load().gs_projection_matrix(_p(a), _p(o))
    return o


def tick_uniforms(cam_world, obj_world, cutout_world=None):
    a = np.ascontiguousarray(cam_world, np.float64); b = np.ascontiguousarray(obj_world, np.float64)
    c = None if cutout_world is None else np.ascontiguousarray(cutout_world, np.float64)
    view = np.zeros(4, np.float32); cut = np.zeros(16, np.float32)
    load().gs_tick_uniforms(_p(a), _p(b), _p(c), _p(view), _p(cut))
    return view, (cut if c is not None else None)


def focal(gs_proj, viewport_h):
    a = np.ascontiguousarray(gs_proj, np.float64)
    return load().gs_focal(_p(a), float(viewport_h))


def scaled_size(css_w, css_h, ratio):
    w, h = C.c_int(0), C.c_int(0)
    load().gs_scaled_size(int(css_w), int(css_h), float(ratio), C.byref(w), C.byref(h))
    return w.value, h.value


def ply_to_splat(ply_bytes):
    """processPlyBuffer (index.js:600-745): PLY bytes -> uint8 array of 32-byte .splat rows."""
    buf = np.frombuffer(bytes(ply_bytes), np.uint8)
    n = C.c_size_t(0); err = C.create_string_buffer(256)
    rc = load().gs_ply_to_splat(_p(buf), buf.size, None, C.byref(n), err, 256)
    if rc != GS_OK:
        raise GsError(rc, err.value.decode())
    out = np.zeros(n.value * 32, np.uint8)
    rc = load().gs_ply_to_splat(_p(buf), buf.size, _p(out), C.byref(n), err, 256)
    if rc != GS_OK:
        raise GsError(rc, err.value.decode())
    return out


def make_params(mv, proj, width, height, x0=0, x1=None, focal_=0.0, background=(0.0, 0.0, 0.0, 1.0), flags=0):
    p = RenderParams()
    p.model_view[:] = [float(v) for v in np.asarray(mv, np.float32)]
    p.projection[:] = [float(v) for v in np.asarray(proj, np.float32)]
    p.fb_width, p.fb_height = int(width), int(height)
    p.x0, p.x1 = int(x0), int(width if x1 is None else x1)
    p.focal = float(np.float32(focal_))
    p.background[:] = [float(v) for v in background]
    p.flags = int(flags)
    return p


class Context:
    """One gs_ctx: one component instance on one GPU."""

    def __init__(self, device=0):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.gs_create(int(device), C.byref(h))
        if rc != GS_OK:
            raise GsError(rc, self._L.gs_last_error(None).decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.gs_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != GS_OK:
            raise GsError(rc, self._L.gs_last_error(self._h).decode())

    # ingest
    def clear(self):
        self._ck(self._L.gs_clear(self._h))

    def push_splat(self, rows):
        rows = np.ascontiguousarray(np.asarray(rows).view(np.uint8).reshape(-1))
        if rows.size % 32:
            raise ValueError("rows must be a multiple of 32 bytes")
        self._ck(self._L.gs_push_splat(self._h, _p(rows), rows.size // 32))

    def push_matrices(self, matrices):
        m = np.ascontiguousarray(matrices, np.float32).reshape(-1)
        if m.size % 16:
            raise ValueError("matrices must be a multiple of 16 floats")
        self._ck(self._L.gs_push_matrices(self._h, _p(m), m.size // 16))

    def ply_to_splat(self, ply_bytes):
        """processPlyBuffer on this context's GPU -> uint8[32 * n] (same bytes as the host converter)."""
        buf = np.frombuffer(bytes(ply_bytes), np.uint8)
        n = C.c_size_t(0)
        self._ck(self._L.gs_ply_to_splat_gpu(self._h, _p(buf), buf.size, None, C.byref(n)))
        out = np.zeros(n.value * 32, np.uint8)
        if n.value:
            self._ck(self._L.gs_ply_to_splat_gpu(self._h, _p(buf), buf.size, _p(out), C.byref(n)))
        return out

    def load_ply(self, ply_bytes):
        buf = np.frombuffer(bytes(ply_bytes), np.uint8)
        self._ck(self._L.gs_load_ply(self._h, _p(buf), buf.size))

    def count(self):
        return self._L.gs_count(self._h)

    # sort
    def sort(self, view, cutout=None, want_indices=True):
        view = np.ascontiguousarray(view, np.float32)
        cut = None if cutout is None else np.ascontiguousarray(cutout, np.float32)
        if not want_indices:
            self._ck(self._L.gs_sort(self._h, _p(view), _p(cut), None, None))
            return None
        out = np.zeros(max(self.count(), 1), np.uint32)
        n = C.c_uint32(0)
        self._ck(self._L.gs_sort(self._h, _p(view), _p(cut), _p(out), C.byref(n)))
        return out[:n.value].copy()

    def sort_begin(self, view, cutout=None):
        """gs_sort_begin: post the sort and return (the reference's tick, index.js:438-455); draws keep using the last completed order"""
        view = np.ascontiguousarray(view, np.float32)
        cut = None if cutout is None else np.ascontiguousarray(cutout, np.float32)
        self._ck(self._L.gs_sort_begin(self._h, _p(view), _p(cut)))

    def sort_poll(self, wait=False, want_indices=True):
        """gs_sort_poll: None while the sort begun with sort_begin() runs; else the new order (installed for the draws that follow), or
        True with want_indices=False"""
        out = np.zeros(max(self.count(), 1), np.uint32) if want_indices else None
        n, done = C.c_uint32(0), C.c_int(0)
        self._ck(self._L.gs_sort_poll(self._h, 1 if wait else 0, _p(out), C.byref(n), C.byref(done)))
        if not done.value:
            return None
        return out[:n.value].copy() if want_indices else True

    def sort_for(self, view, cutout, strip_params, want_indices=True):
        """gs_sort_for: the order of the splats that can reach strip_params' column strip (a sub-sequence of sort()'s result)."""
        view = np.ascontiguousarray(view, np.float32)
        cut = None if cutout is None else np.ascontiguousarray(cutout, np.float32)
        if not want_indices:
            self._ck(self._L.gs_sort_for(self._h, _p(view), _p(cut), C.byref(strip_params), None, None))
            return None
        out = np.zeros(max(self.count(), 1), np.uint32)
        n = C.c_uint32(0)
        self._ck(self._L.gs_sort_for(self._h, _p(view), _p(cut), C.byref(strip_params), _p(out), C.byref(n)))
        return out[:n.value].copy()

    def sort_gathered(self, view, cutout, views):
        """the sort of a frame drawn with render_gathered(views): only this rank's strip when it owns exactly one"""
        views = list(views) if isinstance(views, (list, tuple)) else [views]
        arr = (RenderParams * len(views))(*views)
        view = np.ascontiguousarray(view, np.float32)
        cut = None if cutout is None else np.ascontiguousarray(cutout, np.float32)
        self._ck(self._L.gs_sort_gathered(self._h, _p(view), _p(cut), arr, len(views)))

    # render
    def render(self, params, flip=False):
        sw = max(params.x1 - params.x0, 0)
        out = np.zeros((max(params.fb_height, 0), sw, 4), np.uint8)
        scratch = out if out.size else np.zeros(4, np.uint8)          # let the library reject bad sizes itself
        self._ck(self._L.gs_render(self._h, C.byref(params), _p(scratch), 0))
        return out

    def render_into(self, params, out):
        """gs_render into a caller-owned H x w x 4 uint8 array (e.g. page-locked memory from host_frame())."""
        self._ck(self._L.gs_render(self._h, C.byref(params), _p(out), out.strides[0]))
        return out

    def render_device(self, params, device_ptr=None):
        self._ck(self._L.gs_render_device(self._h, C.byref(params), C.c_void_p(device_ptr) if device_ptr else None))

    def render_stereo(self, left, right):
        arr = (RenderParams * 2)(left, right)
        o0 = np.zeros((left.fb_height, left.x1 - left.x0, 4), np.uint8)
        o1 = np.zeros((right.fb_height, right.x1 - right.x0, 4), np.uint8)
        outs = (C.c_void_p * 2)(o0.ctypes.data, o1.ctypes.data)
        self._ck(self._L.gs_render_stereo(self._h, arr, outs, 0))
        return o0, o1

    def set_scene(self, depth=None, rgba=None):
        """Opaque-scene inputs (depthTest LEQUAL against `depth` [H,W] f32 window depth; blend over `rgba` [H,W,4] u8)."""
        if depth is None and rgba is None:
            self._ck(self._L.gs_set_scene(self._h, None, None, 0, 0))
            return
        d = None if depth is None else np.ascontiguousarray(depth, np.float32)
        c = None if rgba is None else np.ascontiguousarray(rgba, np.uint8)
        h, w = (d.shape if d is not None else c.shape[:2])
        self._ck(self._L.gs_set_scene(self._h, _p(d), _p(c), int(w), int(h)))

    def sync(self):
        self._ck(self._L.gs_sync(self._h))

    def set_stream(self, stream_ptr):
        self._ck(self._L.gs_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def frame_stream(self):
        """hipStream_t (as an int) of the lane the current frame was enqueued on."""
        return int(self._L.gs_frame_stream(self._h) or 0)

    def frame_status_device(self):
        """device address (int) of the current frame's completion word: one uint32, 0 = complete, else drawn again at sync()"""
        p = C.c_void_p()
        self._ck(self._L.gs_frame_status_device(self._h, C.byref(p)))
        return int(p.value or 0)

    def frame_status(self):
        """the completion word read back once the frame's stream is idle (a test's view of it: a GPU-side consumer reads the word itself)"""
        addr, st = self.frame_status_device(), self.frame_stream()
        hip = hip_runtime()
        hip.hipStreamSynchronize(C.c_void_p(st))
        v = C.c_uint32(0)
        hip.hipMemcpy(C.byref(v), C.c_void_p(addr), C.c_size_t(4), 2)
        return int(v.value)

    def frame_lane(self):
        """Index of the pipeline lane the current frame went to (no waiting)."""
        return int(self._L.gs_frame_lane(self._h))

    def lane_stream(self, lane):
        """hipStream_t of a lane, once its worker thread has enqueued everything handed to it so far."""
        return int(self._L.gs_lane_stream(self._h, int(lane)) or 0)

    def wait_stream(self, stream_ptr):
        """The next frame starts (on the GPU) only after everything queued on the given hipStream_t so far."""
        self._ck(self._L.gs_wait_stream(self._h, C.c_void_p(stream_ptr)))

    def stream_wait_frame(self, stream_ptr):
        """The given hipStream_t waits (on the GPU) for the frame enqueued last."""
        self._ck(self._L.gs_stream_wait_frame(self._h, C.c_void_p(stream_ptr)))

    # ---- several GPUs (one Context per GPU; see gs_splat.h) ----
    def comm_unique_id(self, transport=None):
        """rank 0: the id every rank passes to comm_init.  transport: TRANSPORT_RCCL / TRANSPORT_INPROC (None = as set)."""
        if transport is not None:
            self.set_option(OPT_COMM_TRANSPORT, transport)
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        self._ck(self._L.gs_comm_unique_id(self._h, buf))
        return bytes(buf)

    def comm_init(self, uid, rank, world):
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        self._ck(self._L.gs_comm_init(self._h, buf, int(rank), int(world)))

    def render_gathered(self, views, root=0, device_frames=None, flags=0):
        """views: one RenderParams (column strips over the ranks) or two (XR eyes).  Asynchronous with RENDER_ASYNC."""
        views = list(views) if isinstance(views, (list, tuple)) else [views]
        arr = (RenderParams * len(views))(*views)
        ptrs = None
        if device_frames is not None:
            ptrs = (C.c_void_p * len(views))(*[C.c_void_p(int(p)) if p else None for p in device_frames])
        self._ck(self._L.gs_render_gathered(self._h, arr, len(views), int(root), ptrs, int(flags)))

    def gathered_size(self, view):
        w, h = C.c_int(0), C.c_int(0)
        self._ck(self._L.gs_gathered_size(self._h, int(view), C.byref(w), C.byref(h)))
        return w.value, h.value

    def read_gathered(self, view, width=None, height=None):
        """root: view `view` of the last gathered frame as H x W x 4 uint8 (the size is the library's; width / height, if
        given, are checked against it)."""
        w, h = self.gathered_size(view)
        if (width is not None and int(width) != w) or (height is not None and int(height) != h):
            raise ValueError("gathered frame %d is %dx%d, not %sx%s" % (view, w, h, width, height))
        out = np.empty((h, w, 4), np.uint8)
        self._ck(self._L.gs_read_gathered(self._h, int(view), _p(out), 0))
        return out

    def set_option(self, opt, value):
        self._ck(self._L.gs_set_option(self._h, int(opt), int(value)))

    def stats(self):
        s = Stats()
        self._ck(self._L.gs_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def download(self, which, count, dtype, width):
        out = np.zeros((count, width), dtype)
        self._ck(self._L.gs_download(self._h, int(which), _p(out), out.nbytes))
        return out


class Multi:
    """gs_multi: one host process, several GPUs (or several "devices" on one GPU): the single-context calls over all of them."""

    def __init__(self, devices):
        self._L = load()
        devs = [int(d) for d in devices]
        arr = (C.c_int * len(devs))(*devs)
        h = C.c_void_p()
        rc = self._L.gs_create_multi(arr, len(devs), C.byref(h))
        if rc != GS_OK:
            raise GsError(rc, self._L.gs_multi_last_error(None).decode())
        self._h = h
        self.devices = devs

    def close(self):
        if getattr(self, "_h", None):
            self._L.gs_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != GS_OK:
            raise GsError(rc, self._L.gs_multi_last_error(self._h).decode())

    def ctx_stats(self, i):
        """gs_get_stats of the context on devices[i] (call sync() first)"""
        s = Stats()
        h = self._L.gs_multi_ctx(self._h, int(i))
        rc = self._L.gs_get_stats(h, C.byref(s))
        if rc != GS_OK:
            raise GsError(rc, self._L.gs_last_error(h).decode())
        return s.as_dict()

    def clear(self):
        self._ck(self._L.gs_multi_clear(self._h))

    def push_splat(self, rows):
        rows = np.ascontiguousarray(np.asarray(rows).view(np.uint8).reshape(-1))
        if rows.size % 32:
            raise ValueError("rows must be a multiple of 32 bytes")
        self._ck(self._L.gs_multi_push_splat(self._h, _p(rows), rows.size // 32))

    def load_ply(self, ply_bytes):
        buf = np.frombuffer(bytes(ply_bytes), np.uint8)
        self._ck(self._L.gs_multi_load_ply(self._h, _p(buf), buf.size))

    def count(self):
        return self._L.gs_multi_count(self._h)

    def set_option(self, opt, value):
        self._ck(self._L.gs_multi_set_option(self._h, int(opt), int(value)))

    @staticmethod
    def _views(views):
        views = list(views) if isinstance(views, (list, tuple)) else [views]
        return (RenderParams * len(views))(*views), len(views)

    def sort(self, view, cutout, views):
        arr, n = self._views(views)
        view = np.ascontiguousarray(view, np.float32)
        cut = None if cutout is None else np.ascontiguousarray(cutout, np.float32)
        self._ck(self._L.gs_multi_sort(self._h, _p(view), _p(cut), arr, n))

    def render(self, views, frames, flags=0):
        """host-direct: frames = one H x W x 4 uint8 array per view (page-locked: host_frame()); strips are copied into them"""
        arr, n = self._views(views)
        frames = list(frames) if isinstance(frames, (list, tuple)) else [frames]
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        self._ck(self._L.gs_multi_render(self._h, arr, n, ptrs, frames[0].strides[0], int(flags)))

    def render_device(self, views, device_frames=None, flags=0):
        arr, n = self._views(views)
        ptrs = None
        if device_frames is not None:
            ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) if p else None for p in device_frames])
        self._ck(self._L.gs_multi_render_device(self._h, arr, n, ptrs, int(flags)))

    def read(self, view, width, height):
        out = np.empty((int(height), int(width), 4), np.uint8)
        self._ck(self._L.gs_multi_read(self._h, int(view), _p(out), 0))
        return out

    def sync(self):
        self._ck(self._L.gs_multi_sync(self._h))
