"""Seeded synthetic workloads of BASELINE.json's configs.

The reference ships no data files (its tests generate inputs in-test, T/test_motion_model.cpp:24,
SURVEY.md section 4), so every workload here is generated from a SplitMix64 stream: the same seed
gives bit-identical float32 arrays on every machine, which is what lets the CPU oracle and the HIP
path be compared index-for-index.

C1  scan_pair_2d      1000-beam 270deg scan of a 10 m x 8 m room with two boxes, SE(2)
C2  cloud_pair_3d     100k-pt cloud on an analytic surface + 4 walls, with normals, SE(3)
C4  batch_3d          K independent C2-style problems (loop-closure candidates)
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n uint64 values of the SplitMix64 stream started at ``seed``."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(seed, n, lo=0.0, hi=1.0):
    u = (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def normal(seed, n, sigma=1.0):
    """Box-Muller on two SplitMix64 streams."""
    u1 = np.maximum(uniform(seed, n), 1e-300)
    u2 = uniform(seed ^ 0x5DEECE66D, n)
    return sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ---- transforms (float64 helpers for building ground truth) ---------------------------
def rpy_to_R(roll, pitch, yaw):
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def se3(t, rpy):
    T = np.zeros((3, 4))
    T[:, :3] = rpy_to_R(*rpy)
    T[:, 3] = t
    return T


def se3_inv(T):
    Ti = np.zeros((3, 4))
    Ti[:, :3] = T[:, :3].T
    Ti[:, 3] = -T[:, :3].T @ T[:, 3]
    return Ti


def se3_mul(A, B):
    C = np.zeros((3, 4))
    C[:, :3] = A[:, :3] @ B[:, :3]
    C[:, 3] = A[:, :3] @ B[:, 3] + A[:, 3]
    return C


def se2(tx, ty, theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, tx], [s, c, ty], [0, 0, 1.0]])


def identity(dim):
    return np.eye(3, dtype=np.float32) if dim == 2 else np.eye(3, 4, dtype=np.float32)


# ---- C2: analytic surface + walls -----------------------------------------------------
def _surface(x, y):
    return 0.3 * np.sin(1.3 * x) * np.cos(0.9 * y)


def _surface_normal(x, y):
    dzdx = 0.3 * 1.3 * np.cos(1.3 * x) * np.cos(0.9 * y)
    dzdy = -0.3 * 0.9 * np.sin(1.3 * x) * np.sin(0.9 * y)
    n = np.stack([-dzdx, -dzdy, np.ones_like(x)], axis=1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def scene_3d(n, seed, noise_sigma=0.0):
    """n points (+unit normals) on z = 0.3 sin(1.3x) cos(0.9y), (x,y) in [-5,5]^2, plus 4 walls.

    80 % of the points lie on the surface (stratified jitter), 5 % on each wall
    (x = +-5, y = +-5, z in [-0.5, 2])."""
    n_wall = n // 20
    n_surf = n - 4 * n_wall
    g = int(np.ceil(np.sqrt(n_surf)))
    cell = np.arange(n_surf)
    jx = uniform(seed * 7919 + 1, n_surf)
    jy = uniform(seed * 7919 + 2, n_surf)
    x = -5.0 + 10.0 * ((cell % g) + jx) / g
    y = -5.0 + 10.0 * ((cell // g) + jy) / g
    y = np.minimum(y, 5.0)
    pts = [np.stack([x, y, _surface(x, y)], axis=1)]
    nrm = [_surface_normal(x, y)]
    walls = [((-5.0, None), (1.0, 0.0, 0.0)), ((5.0, None), (-1.0, 0.0, 0.0)),
             ((None, -5.0), (0.0, 1.0, 0.0)), ((None, 5.0), (0.0, -1.0, 0.0))]
    for w, ((wx, wy), nvec) in enumerate(walls):
        a = uniform(seed * 7919 + 10 + 2 * w, n_wall, -5.0, 5.0)
        z = uniform(seed * 7919 + 11 + 2 * w, n_wall, -0.5, 2.0)
        if wx is not None:
            p = np.stack([np.full(n_wall, wx), a, z], axis=1)
        else:
            p = np.stack([a, np.full(n_wall, wy), z], axis=1)
        pts.append(p)
        nrm.append(np.tile(np.array(nvec), (n_wall, 1)))
    P = np.concatenate(pts, axis=0)
    N = np.concatenate(nrm, axis=0)
    if noise_sigma > 0:
        P = P + N * normal(seed * 7919 + 99, P.shape[0], noise_sigma)[:, None]
    return P, N


def cloud_pair_3d(n=100_000, seed=2000, t=(0.05, -0.03, 0.02), rpy_deg=(1.0, -1.5, 2.0), noise_sigma=0.0):
    """C2: returns dict(fixed, fixed_normals, moving, moving_normals, X_gt) as float32 (X_gt 3x4).

    fixed and moving sample the same scene with different seeds; moving is expressed in the
    moving frame, i.e. fixed ~= X_gt * moving."""
    X_gt = se3(np.array(t), np.deg2rad(np.array(rpy_deg)))
    Pf, Nf = scene_3d(n, seed, noise_sigma)
    Pm_w, Nm_w = scene_3d(n, seed + 1, noise_sigma)
    Xi = se3_inv(X_gt)
    Pm = Pm_w @ Xi[:, :3].T + Xi[:, 3]
    Nm = Nm_w @ Xi[:, :3].T
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"fixed": f32(Pf), "fixed_normals": f32(Nf), "moving": f32(Pm), "moving_normals": f32(Nm),
            "X_gt": f32(X_gt)}


def batch_3d(K=256, n=50_000, seed=4000, shared_fixed_group=8, t_max=0.2, rpy_max_deg=5.0, only=None):
    """C4: K alignment problems.  The fixed cloud is shared per group of ``shared_fixed_group``
    problems (setFixed once per query map, multi_loop_detector_brute_force_impl.cpp:63);
    0 or 1 => all distinct.  Ground truths: t ~ U[-t_max,t_max]^3, rpy ~ U[-rpy_max,rpy_max]^3.
    ``only``: generate just these problems of the K (a rank of a sharded job makes its own shard; every problem is seeded
    by its index, so the result is the same as picking them out of the full list) -- returns them in the given order."""
    out = []
    fixed_cache = {}
    for k in (range(K) if only is None else only):
        grp = k // shared_fixed_group if shared_fixed_group > 1 else k
        if grp not in fixed_cache:
            Pf, Nf = scene_3d(n, seed + 10 * grp)
            fixed_cache = {grp: (np.ascontiguousarray(Pf, np.float32), np.ascontiguousarray(Nf, np.float32))}
        t = uniform(seed * 31 + 3 * k, 3, -t_max, t_max)
        rpy = np.deg2rad(uniform(seed * 31 + 3 * k + 1, 3, -rpy_max_deg, rpy_max_deg))
        X_gt = se3(t, rpy)
        Pm_w, Nm_w = scene_3d(n, seed + 10 * grp + 1 + (k % max(shared_fixed_group, 1)))
        Xi = se3_inv(X_gt)
        Pm = Pm_w @ Xi[:, :3].T + Xi[:, 3]
        Nm = Nm_w @ Xi[:, :3].T
        out.append({"group": grp, "fixed": fixed_cache[grp][0], "fixed_normals": fixed_cache[grp][1],
                    "moving": np.ascontiguousarray(Pm, np.float32),
                    "moving_normals": np.ascontiguousarray(Nm, np.float32),
                    "X_gt": np.ascontiguousarray(X_gt, np.float32)})
    return out


# ---- C1: 2D laser scan -------------------------------------------------------------------
def _room_segments():
    segs = []

    def box(x0, y0, x1, y1):
        segs.extend([((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1)), ((x0, y1), (x0, y0))])

    box(-5.0, -4.0, 5.0, 4.0)  # 10 m x 8 m room
    box(1.5, 1.0, 2.5, 2.2)  # interior boxes
    box(-3.2, -2.6, -2.0, -1.2)
    return np.array(segs, dtype=np.float64)  # (S, 2, 2)


def scan_2d(pose, beams=1000, fov_deg=270.0, sigma=0.0, seed=0):
    """Ray-cast scan taken from SE(2) ``pose`` (3x3, sensor in world); returns points and unit
    normals in the sensor frame.  Beams that hit nothing are dropped (none in a closed room)."""
    segs = _room_segments()
    ang = np.deg2rad(np.linspace(-fov_deg / 2, fov_deg / 2, beams))
    c, s = pose[0, 0], pose[1, 0]
    o = pose[:2, 2]
    d = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    dw = d @ np.array([[c, s], [-s, c]])  # rotate directions into the world
    a = segs[:, 0, :]
    e = segs[:, 1, :] - a
    # solve o + r*dw = a + u*e for every (beam, segment)
    den = dw[:, None, 0] * e[None, :, 1] - dw[:, None, 1] * e[None, :, 0]
    ao = a[None, :, :] - o[None, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (ao[..., 0] * e[None, :, 1] - ao[..., 1] * e[None, :, 0]) / den
        u = (ao[..., 0] * dw[:, None, 1] - ao[..., 1] * dw[:, None, 0]) / den
    ok = (np.abs(den) > 1e-12) & (r > 1e-6) & (u >= 0.0) & (u <= 1.0)
    r = np.where(ok, r, np.inf)
    hit = np.argmin(r, axis=1)
    rng = r[np.arange(beams), hit]
    valid = np.isfinite(rng)
    if sigma > 0:
        rng = rng + normal(seed * 104729 + 5, beams, sigma)
    pts = d * rng[:, None]
    ew = e[hit]
    nw = np.stack([-ew[:, 1], ew[:, 0]], axis=1)
    nw = nw / np.linalg.norm(nw, axis=1, keepdims=True)
    flip = np.sum(nw * dw, axis=1) > 0  # normals face the sensor
    nw[flip] *= -1
    ns = nw @ np.array([[c, -s], [s, c]])  # world -> sensor frame
    return pts[valid], ns[valid]


def scan_pair_2d(beams=1000, t=(0.10, 0.05), theta_deg=3.0, sigma=0.0, seed=1000):
    """C1: fixed scan from the origin, moving scan from pose X_gt; fixed ~= X_gt * moving."""
    X_gt = se2(t[0], t[1], np.deg2rad(theta_deg))
    Pf, Nf = scan_2d(se2(0, 0, 0), beams, sigma=sigma, seed=seed)
    Pm, Nm = scan_2d(X_gt, beams, sigma=sigma, seed=seed + 1)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"fixed": f32(Pf), "fixed_normals": f32(Nf), "moving": f32(Pm), "moving_normals": f32(Nm),
            "X_gt": f32(X_gt)}


# ---- C5: pose graphs ---------------------------------------------------------------------------------
def _quat_v2t(v):
    x, y, z = v[3:6]
    w = np.sqrt(max(0.0, 1.0 - x * x - y * y - z * z))
    T = np.zeros((3, 4))
    T[:, :3] = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T[:, 3] = v[:3]
    return T


def pose_graph_3d(V=50_000, E=200_000, seed=5000, sigma_t=0.01, sigma_r=0.005, spacing=0.5, loop_radius=1.5):
    """C5: V SE(3) poses on a 3-D lawn-mower trajectory, V-1 odometry edges + (E-V+1) loop
    edges between poses closer than loop_radius; Z = true relative pose (+) N(0, sigma); Omega = I; initial guess =
    odometry integration; pose 0 fixed.  Returns dict(poses_gt, poses_init, ij, Z) (float32 / int32)."""
    row = max(2, int(round(np.sqrt(V / 4.0))))  # poses per sweep line
    gt = np.zeros((V, 3, 4))
    for v in range(V):
        line, k = divmod(v, row)
        layer, line_in_layer = divmod(line, row)
        x = (k if line % 2 == 0 else row - 1 - k) * spacing
        y = line_in_layer * spacing
        z = layer * spacing * 2.0
        yaw = 0.0 if line % 2 == 0 else np.pi
        gt[v] = se3(np.array([x, y, z]), np.array([0.02 * np.sin(0.1 * v), 0.03 * np.cos(0.07 * v), yaw]))
    # candidate loop pairs: neighbours on adjacent sweep lines / layers
    pos = gt[:, :, 3]
    n_loop = E - (V - 1)
    cand_i = (uniform(seed + 1, 4 * n_loop) * V).astype(np.int64)
    offs = np.array([2 * row - 1, 2 * row, 2 * row + 1, row * row, row * row + 1, 3])
    cand_j = cand_i + offs[(uniform(seed + 2, 4 * n_loop) * len(offs)).astype(np.int64)]
    ok = (cand_j < V) & (cand_j >= 0)
    d = np.linalg.norm(pos[np.minimum(cand_j, V - 1)] - pos[cand_i], axis=1)
    ok &= d < loop_radius
    ci, cj = cand_i[ok], cand_j[ok]
    if ci.size < n_loop:  # fall back to short-range skips to reach the requested edge count
        extra = n_loop - ci.size
        ei = (uniform(seed + 3, extra) * max(V - 3, 1)).astype(np.int64)
        ci = np.concatenate([ci, ei])
        cj = np.concatenate([cj, ei + 2])
    ci, cj = ci[:n_loop], cj[:n_loop]
    ij = np.concatenate([np.stack([np.arange(V - 1), np.arange(1, V)], 1), np.stack([ci, cj], 1)]).astype(np.int32)
    Et = ij.shape[0]
    nt = normal(seed + 10, 3 * Et, sigma_t).reshape(Et, 3)
    nr = normal(seed + 11, 3 * Et, sigma_r).reshape(Et, 3)
    Z = np.zeros((Et, 3, 4))
    for e in range(Et):
        rel = se3_mul(se3_inv(gt[ij[e, 0]]), gt[ij[e, 1]])
        Z[e] = se3_mul(rel, _quat_v2t(np.concatenate([nt[e], 0.5 * nr[e]])))
    init = np.zeros((V, 3, 4))
    init[0] = gt[0]
    for v in range(1, V):
        init[v] = se3_mul(init[v - 1], Z[v - 1])
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"poses_gt": f32(gt), "poses_init": f32(init), "ij": ij, "Z": f32(Z)}


def pose_graph_2d(V=400, E=900, seed=5100, sigma_t=0.01, sigma_r=0.005):
    """small SE(2) graph: square-ish loop trajectory + random loop closures."""
    gt = np.zeros((V, 3, 3))
    for v in range(V):
        a = 2 * np.pi * v / V
        gt[v] = se2(6 * np.cos(a) + 0.5 * np.cos(5 * a), 6 * np.sin(a), a + np.pi / 2)
    n_loop = E - (V - 1)
    ci = (uniform(seed + 1, n_loop) * V).astype(np.int64)
    cj = (ci + 1 + (uniform(seed + 2, n_loop) * 12).astype(np.int64)) % V
    keep = ci != cj
    ij = np.concatenate([np.stack([np.arange(V - 1), np.arange(1, V)], 1), np.stack([ci[keep], cj[keep]], 1)]).astype(np.int32)
    Et = ij.shape[0]
    n = normal(seed + 10, 3 * Et).reshape(Et, 3) * np.array([sigma_t, sigma_t, sigma_r])
    Z = np.zeros((Et, 3, 3))
    for e in range(Et):
        Z[e] = np.linalg.inv(gt[ij[e, 0]]) @ gt[ij[e, 1]] @ se2(*n[e])
    init = np.zeros((V, 3, 3))
    init[0] = gt[0]
    for v in range(1, V):
        init[v] = init[v - 1] @ Z[v - 1]
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"poses_gt": f32(gt), "poses_init": f32(init), "ij": ij, "Z": f32(Z)}


# ---- C3: RGB-D shaped organised clouds ----------------------------------------------------------------
def default_camera(rows=480, cols=640):
    f = 525.0 * cols / 640.0
    return np.array([[f, 0, (cols - 1) / 2.0], [0, f, (rows - 1) / 2.0], [0, 0, 1.0]])


def render_depth(T_w_cam, K, rows, cols, iters=30):
    """Ray-cast the analytic surface z = 0.3 sin(1.3x) cos(0.9y) from camera pose T_w_cam (3x4, camera in world).
    Returns (points, normals) in the CAMERA frame, each (rows*cols, 3), organised row-major."""
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    d = np.stack([(c - K[0, 2]) / K[0, 0], (r - K[1, 2]) / K[1, 1], np.ones_like(c, dtype=np.float64)], axis=-1).reshape(-1, 3)
    R, p = T_w_cam[:, :3], T_w_cam[:, 3]
    dw = d @ R.T
    t = np.full(d.shape[0], (p[2] - 0.0) / np.maximum(-dw[:, 2], 1e-9))
    for _ in range(iters):  # fixed-point iteration on the ray parameter (the surface is gentle)
        x = p[0] + t * dw[:, 0]
        y = p[1] + t * dw[:, 1]
        t = (_surface(x, y) - p[2]) / dw[:, 2]
    x = p[0] + t * dw[:, 0]
    y = p[1] + t * dw[:, 1]
    pts = d * t[:, None]
    n_w = _surface_normal(x, y)
    n_c = n_w @ R  # world -> camera: R^T n
    return pts, n_c


def rgbd_pair(rows=480, cols=640, seed=3000, t=(0.03, 0.01, -0.02), rpy_deg=(0.5, 1.0, -0.5), hole_fraction=0.02,
              depth_min=0.4, depth_max=8.0):
    """C3: a fixed organised cloud (+normals, invalid pixels = NaN) rendered from camera 1 and the unprojected render
    from camera 2 as the moving cloud; X_gt maps moving (camera 2 frame) into fixed (camera 1 frame)."""
    K = default_camera(rows, cols)
    T_w_c1 = np.zeros((3, 4))
    T_w_c1[:, :3] = np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])  # looking down from 4 m
    T_w_c1[:, 3] = [0.0, 0.0, 4.0]
    X_gt = se3(np.array(t), np.deg2rad(np.array(rpy_deg)))
    T_w_c2 = se3_mul(T_w_c1, X_gt)
    Pf, Nf = render_depth(T_w_c1, K, rows, cols)
    Pm, Nm = render_depth(T_w_c2, K, rows, cols)
    okf = (Pf[:, 2] >= depth_min) & (Pf[:, 2] <= depth_max) & (uniform(seed + 1, rows * cols) >= hole_fraction)
    okm = (Pm[:, 2] >= depth_min) & (Pm[:, 2] <= depth_max) & (uniform(seed + 2, rows * cols) >= hole_fraction)
    Pf = Pf.copy()
    Pf[~okf] = np.nan
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {"fixed": f32(Pf), "fixed_normals": f32(Nf), "moving": f32(Pm[okm]), "moving_normals": f32(Nm[okm]),
            "X_gt": f32(X_gt), "K": f32(K), "rows": rows, "cols": cols, "depth_min": depth_min, "depth_max": depth_max}
