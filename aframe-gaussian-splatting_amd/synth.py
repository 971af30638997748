"""Synthetic inputs for tests and bench: the real train.splat / bicycle.ply are remote URLs
(index.html:13) and unavailable offline, so the workloads are `.splat`-layout rows with train.splat-like
statistics, and cameras taken from the reference's demo pages.

Everything here is host-side input generation; none of it is on the hot path.
"""
import math

import numpy as np

SEED_BASE = 0x5EED0000
N_TRAIN = 1 << 20            # C1/C2/C4  "train.splat (~1M gaussians)"
N_BICYCLE = 6 * (1 << 20)    # C3        "bicycle.ply (~6M gaussians)"
N_20M = 20 * (1 << 20)       # C5


def make_splat_rows(n, seed=SEED_BASE + 2, order_by_importance=True):
    """n x 32-byte .splat rows (f32 pos[3], f32 scale[3], u8 rgba[4], u8 quat_wxyz[4]; index.js:344-359, 671-676)
    as a uint8 array of n*32 bytes.  80 % of the positions ~ N(0, diag(2.5,1,2.5)^2), 20 % uniform "floaters" in
    [-8,8]^3; ln scale ~ N(-4.2, 0.9^2) clipped to [-7,-1]; quaternion uniform on S^3; opacity =
    sigmoid(N(0.5, 2.5^2)); rows ordered by importance (scale product x opacity) descending like
    processPlyBuffer (index.js:653-668)."""
    g = np.random.Generator(np.random.PCG64(seed))
    pos = g.normal(0.0, 1.0, (n, 3)) * np.array([2.5, 1.0, 2.5])
    fl = g.random(n) < 0.2
    pos[fl] = g.uniform(-8.0, 8.0, (int(fl.sum()), 3))
    lns = np.clip(g.normal(-4.2, 0.9, (n, 3)), -7.0, -1.0)
    q = g.normal(0.0, 1.0, (n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = 1.0 / (1.0 + np.exp(-g.normal(0.5, 2.5, n)))
    rgb = g.integers(0, 256, (n, 3))
    if order_by_importance:
        imp = np.exp(lns.sum(axis=1)) * op
        o = np.argsort(-imp, kind="stable")
        pos, lns, q, op, rgb = pos[o], lns[o], q[o], op[o], rgb[o]
    rows = np.zeros((n, 32), np.uint8)
    rows[:, 0:12] = pos.astype("<f4").view(np.uint8).reshape(n, 12)
    rows[:, 12:24] = np.exp(lns).astype("<f4").view(np.uint8).reshape(n, 12)
    rows[:, 24:27] = rgb.astype(np.uint8)
    rows[:, 27] = np.clip(np.rint(op * 255.0), 0, 255).astype(np.uint8)
    rows[:, 28:32] = np.clip(np.rint(q * 128.0 + 128.0), 0, 255).astype(np.uint8)
    return rows.reshape(-1)


def make_splat_rows_fast(n, seed=SEED_BASE + 5, block=1 << 21):
    """The same distributions as make_splat_rows for the 20 M-row configuration (C5), generated in float32 blocks and
    left in generation order (processPlyBuffer's importance order matters to progressive loading, not to the frame):
    about 3x faster on the host, which is what the 20 M tests and bench runs spend most of their time on."""
    g = np.random.Generator(np.random.PCG64(seed))
    rows = np.empty((n, 32), np.uint8)
    sig = np.array([2.5, 1.0, 2.5], np.float32)
    for a in range(0, n, block):
        m = min(block, n - a)
        pos = g.standard_normal((m, 3), dtype=np.float32) * sig
        fl = g.random(m, dtype=np.float32) < 0.2
        pos[fl] = g.random((int(fl.sum()), 3), dtype=np.float32) * np.float32(16.0) - np.float32(8.0)
        lns = np.clip(g.standard_normal((m, 3), dtype=np.float32) * np.float32(0.9) - np.float32(4.2), -7.0, -1.0)
        q = g.standard_normal((m, 4), dtype=np.float32)
        q /= np.sqrt((q * q).sum(axis=1, keepdims=True))
        op = 1.0 / (1.0 + np.exp(-(g.standard_normal(m, dtype=np.float32) * np.float32(2.5) + np.float32(0.5))))
        r = rows[a:a + m]
        r[:, 0:12] = pos.astype("<f4").view(np.uint8).reshape(m, 12)
        r[:, 12:24] = np.exp(lns).astype("<f4").view(np.uint8).reshape(m, 12)
        r[:, 24:27] = g.integers(0, 256, (m, 3), dtype=np.uint8)
        r[:, 27] = np.clip(np.rint(op * 255.0), 0, 255).astype(np.uint8)
        r[:, 28:32] = np.clip(np.rint(q * 128.0 + 128.0), 0, 255).astype(np.uint8)
    return rows.reshape(-1)


def rows_to_inria_ply(rows_u8):
    """Inverse of processPlyBuffer for synthetic data: .splat rows -> INRIA-layout binary PLY bytes
    (62 float props, 248 B/row) so the `.ply` loader path (index.js:600-745) can be exercised."""
    rows = np.asarray(rows_u8, np.uint8).reshape(-1, 32)
    n = rows.shape[0]
    pos = rows[:, 0:12].copy().view("<f4").reshape(n, 3)
    sc = rows[:, 12:24].copy().view("<f4").reshape(n, 3).astype(np.float64)
    rgba = rows[:, 24:28].astype(np.float64)
    q = (rows[:, 28:32].astype(np.float64) - 128.0) / 128.0
    props = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + ["f_rest_%d" % i for i in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    body = np.zeros((n, len(props)), "<f4")
    body[:, 0:3] = pos
    body[:, 6:9] = ((rgba[:, 0:3] + 0.25) / 255.0 - 0.5) / 0.28209479177387814
    a = np.clip((rgba[:, 3] + 0.25) / 255.0, 1e-4, 1 - 1e-4)
    body[:, 54] = np.log(a / (1 - a))
    body[:, 55:58] = np.log(np.maximum(sc, 1e-30))
    body[:, 58:62] = q
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join("property float %s\n" % p for p in props) + "end_header\n"
    return hdr.encode("ascii") + body.tobytes()


# ---------------------------------------------------------------- cameras (three.js conventions, column-major)

def compose(pos, yaw_deg=0.0, scale=(1.0, 1.0, 1.0)):
    """Object world matrix: T * R_y(yaw) * S, as 16 column-major f64 (three.js Matrix4.compose)."""
    h = yaw_deg * math.pi / 360.0
    x, y, z, w = 0.0, math.sin(h), 0.0, math.cos(h)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz, wx, wy, wz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2, w * x2, w * y2, w * z2
    sx, sy, sz = scale
    return np.array([(1 - (yy + zz)) * sx, (xy + wz) * sx, (xz - wy) * sx, 0,
                     (xy - wz) * sy, (1 - (xx + zz)) * sy, (yz + wx) * sy, 0,
                     (xz + wy) * sz, (yz - wx) * sz, (1 - (xx + yy)) * sz, 0,
                     pos[0], pos[1], pos[2], 1], np.float64)


def perspective(fov_deg, aspect, near=0.005, far=10000.0):
    """three.js PerspectiveCamera.updateProjectionMatrix (A-Frame defaults: fov 80, near 0.005, far 10000)."""
    top = near * math.tan(math.radians(fov_deg) * 0.5)
    h, w = 2 * top, aspect * 2 * top
    return frustum(-0.5 * w, 0.5 * w, top, top - h, near, far)


def frustum(left, right, top, bottom, near, far):
    x, y = 2 * near / (right - left), 2 * near / (top - bottom)
    a, b = (right + left) / (right - left), (top + bottom) / (top - bottom)
    c, d = -(far + near) / (far - near), -2 * far * near / (far - near)
    return np.array([x, 0, 0, 0, 0, y, 0, 0, a, b, c, -1, 0, 0, d, 0], np.float64)


def uniforms(cam_world, obj_world, proj, vw, vh, cutout_world=None, capi=None):
    """The per-frame uniforms the reference computes in tick / onBeforeRender (index.js:184-195, 438-487).
    Uses the product's host helpers (capi) when given, else numpy (S = diag(1,-1,1,1) conjugation)."""
    if capi is not None:
        mv = capi.model_view_matrix(cam_world, obj_world)
        gp = capi.projection_matrix(proj)
        view, cut = capi.tick_uniforms(cam_world, obj_world, cutout_world)
        focal = capi.focal(gp, vh)
    else:
        S = np.diag([1.0, -1.0, 1.0, 1.0])
        C = np.asarray(cam_world, np.float64).reshape(4, 4).T
        M = np.asarray(obj_world, np.float64).reshape(4, 4).T
        mvm = S @ (np.linalg.inv(C) @ M) @ S
        mv = mvm.T.reshape(-1).copy()
        gp = (np.asarray(proj, np.float64).reshape(4, 4).T @ S).T.reshape(-1).copy()
        view = np.array([mv[2], mv[6], mv[10], mv[14]], np.float32)
        cut = None
        if cutout_world is not None:
            B = np.asarray(cutout_world, np.float64).reshape(4, 4).T
            cut = (np.linalg.inv(B) @ M).T.reshape(-1).astype(np.float32)
        focal = (vh / 2.0) * abs(gp[5])
    return {"gs_mv": np.asarray(mv, np.float64), "gs_proj": np.asarray(gp, np.float64), "view": np.asarray(view, np.float32),
            "cutout": cut, "focal": float(focal), "vw": int(vw), "vh": int(vh)}


def index_html_camera(vw=1920, vh=1080, yaw_deg=0.0, capi=None):
    """index.html:13 pose: A-Frame default camera at (0,1.6,0); entity at (0,1.5,-2) with yaw (the benchmark orbit)."""
    return uniforms(compose((0.0, 1.6, 0.0)), compose((0.0, 1.5, -2.0), yaw_deg), perspective(80.0, vw / vh), vw, vh, capi=capi)


def outside_cloud_camera(vw=1920, vh=1080, yaw_deg=0.0, capi=None, distance=7.5):
    """The scene seen from OUTSIDE: the entity 3 sigma (7.5 units) in front of the A-Frame default camera instead of index.html's 2
    (where the camera sits inside the sigma = 2.5 cloud and the nearest splats fill the screen): sky around the cloud, thin
    coverage at its rim, many small splats per tile -- the regime between the saturated headline pose and the unsaturated scene."""
    return uniforms(compose((0.0, 1.6, 0.0)), compose((0.0, 1.6, -float(distance)), yaw_deg), perspective(80.0, vw / vh), vw, vh, capi=capi)


def cutout_demo_camera(vw=1920, vh=1080, yaw_deg=0.0, capi=None):
    """cutout-demo.html:22-24 pose: camera (5.132,1.6,7.237); entity scale 2 at (0,0.8,-2); cutout box
    scale (4.17,2.95,3.89) at (0.8145,1.73322,-2.35981)."""
    return uniforms(compose((5.132, 1.6, 7.237)), compose((0.0, 0.8, -2.0), yaw_deg, (2.0, 2.0, 2.0)),
                    perspective(80.0, vw / vh), vw, vh,
                    cutout_world=compose((0.8145, 1.73322, -2.35981), 0.0, (4.17, 2.95, 3.89)), capi=capi)


def xr_eye_cameras(yaw_deg=0.0, xr_pixel_ratio=0.5, capi=None, cant_deg=0.0):
    """C4: two eyes +-0.032 m on x, asymmetric Quest-3-like frusta, 2064x2208 x xrPixelRatio each; the sort uses
    the head camera (index.js:441) -- returned as the third element.  cant_deg: each eye turned outward about y by
    that angle (canted displays: Index, Pimax), so an eye's view direction differs from the head camera's."""
    w, h = int(math.floor(2064 * xr_pixel_ratio)), int(math.floor(2208 * xr_pixel_ratio))
    near, far = 0.005, 10000.0
    obj = compose((0.0, 1.5, -2.0), yaw_deg)
    eyes = []
    for sx in (-1.0, 1.0):
        # tan half-angles: outer 54 deg, inner 40 deg, up 44 deg, down 55 deg
        lo, ro = (math.tan(math.radians(54)), math.tan(math.radians(40))) if sx < 0 else (math.tan(math.radians(40)), math.tan(math.radians(54)))
        proj = frustum(-lo * near, ro * near, math.tan(math.radians(44)) * near, -math.tan(math.radians(55)) * near, near, far)
        eyes.append(uniforms(compose((0.032 * sx, 1.6, 0.0), -sx * cant_deg), obj, proj, w, h, capi=capi))
    head = uniforms(compose((0.0, 1.6, 0.0)), obj, perspective(80.0, w / h), w, h, capi=capi)
    return eyes[0], eyes[1], head
