"""-m gpu: scene clipping / correspondence-based merging on the HIP path (srrg2_scene_*, through the C ABI) against
the CPU oracle (oracle/o_scene.c), bit for bit; and the tracker cycle clip -> align -> merge with the clouds staying
in device memory (S/trackers/tracker_slice_processor_impl.cpp:111-205)."""
import numpy as np
import pytest

from helpers import cue_config
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import mapping
from srrg2_slam_interfaces_amd import synthetic as syn
from test_oracle_scene import _clouds

pytestmark = pytest.mark.gpu
f32 = np.float32


def _bindings(oracle, product):
    return oracle.scene_binding(), product.scene_binding(0)


def _same_scene(a, b):
    pa, na = a.get()
    pb, nb = b.get()
    assert pa.shape == pb.shape
    assert pa.tobytes() == pb.tobytes()
    assert na.tobytes() == nb.tobytes()


@pytest.mark.parametrize("dim", [3, 2])
def test_clip_ball_parity(oracle, product, dim):
    rng = np.random.default_rng(11 + dim)
    n = 200_000
    sp = rng.uniform(-30, 30, (n, dim)).astype(f32)
    sn = rng.normal(size=(n, dim)).astype(f32)
    sp[::997] = np.nan
    pose = (syn.se3(np.array([3.0, -2.0, 0.5]), np.deg2rad(np.array([10.0, 5.0, -20.0]))) if dim == 3
            else syn.se2(3.0, -2.0, 0.7)).astype(f32)
    out = []
    for b in _bindings(oracle, product):
        full, clipped = mapping.Scene(b, dim), mapping.Scene(b, dim)
        full.set(sp, sn)
        cl = mapping.SceneClipperBall(b, range_max=12.0)
        cl.set_full_scene(full); cl.set_clipped_scene_in_robot(clipped); cl.set_robot_in_local_map(pose)
        cl.compute()
        out.append((cl, clipped))
    (c_ref, s_ref), (c_gpu, s_gpu) = out
    assert c_ref.status() == c_gpu.status() == mapping.CLIPPER_SUCCESSFUL
    assert np.array_equal(c_ref.global_indices(), c_gpu.global_indices())
    _same_scene(s_ref, s_gpu)
    assert 1000 < s_gpu.size() < n


def test_clip_into_a_scene_that_has_room_or_not(oracle, product):
    """Round 6, last: a clipped scene with room from the clip before gets its scatter launched behind the scan WITHOUT the host having
    seen the total (one wait per clip); a total beyond the room repeats the scatter.  One pair of scenes through a small clip, a larger
    one (beyond the room: repeated), a smaller one again (within the room: speculated), an empty one, and a full one."""
    rng = np.random.default_rng(77)
    n = 120_000
    sp = rng.uniform(-30, 30, (n, 3)).astype(f32)
    sn = rng.normal(size=(n, 3)).astype(f32)
    sp[::1013] = np.nan
    pose = syn.se3(np.array([3.0, -2.0, 0.5]), np.deg2rad(np.array([10.0, 5.0, -20.0]))).astype(f32)
    pairs = []
    for b in _bindings(oracle, product):
        full, clipped = mapping.Scene(b, 3), mapping.Scene(b, 3)
        full.set(sp, sn)
        pairs.append((b, full, clipped))
    sizes = []
    for r in (4.0, 15.0, 8.0, 0.01, 500.0, 6.0):
        got = []
        for b, full, clipped in pairs:
            cl = mapping.SceneClipperBall(b, range_max=r)
            cl.set_full_scene(full); cl.set_clipped_scene_in_robot(clipped); cl.set_robot_in_local_map(pose)
            cl.compute()
            got.append((cl.global_indices(), clipped))
        assert np.array_equal(got[0][0], got[1][0]), r
        _same_scene(got[0][1], got[1][1])
        sizes.append(got[1][1].size())
    assert sizes[0] < sizes[2] < sizes[1] < sizes[4] and sizes[3] == 0


@pytest.mark.parametrize("target", [200, 20])
@pytest.mark.parametrize("with_corr", [True, False])
def test_merge_parity_with_duplicates(oracle, product, target, with_corr):
    sp, sn, mp, mn, T, corr = _clouds(21, ns=5000, nm=3000)
    params = mapping.MergerParams(50.0, 0.25, target)
    arr = np.zeros(len(corr), dtype=[("fixed_idx", np.int32), ("moving_idx", np.int32), ("response", np.float32)])
    for k, c in enumerate(corr):
        arr[k] = c
    res, scenes = [], []
    for b in _bindings(oracle, product):
        scene, meas = mapping.Scene(b, 3), mapping.Scene(b, 3)
        scene.set(sp, sn); meas.set(mp, mn)
        mg = mapping.MergerCorrespondenceHomo(b, params)
        mg.set_scene(scene); mg.set_measurement(meas); mg.set_measurement_in_scene(T)
        if with_corr:
            mg.set_correspondences(arr)
        res.append(mg.compute())
        scenes.append(scene)
        assert mg.status() == mapping.MERGER_SUCCESS
    assert res[0] == res[1], res
    _same_scene(*scenes)
    if with_corr:
        assert res[1]["num_merged"] > 100
        # a second merge into the grown scene (capacity growth keeps the old points)
    with pytest.raises(RuntimeError):
        b = product.scene_binding(0)
        scene, meas = mapping.Scene(b, 3), mapping.Scene(b, 3)
        scene.set(sp, sn); meas.set(mp, mn)
        mg = mapping.MergerCorrespondenceHomo(b, params)
        mg.set_scene(scene); mg.set_measurement(meas); mg.set_measurement_in_scene(T)
        bad = arr[:1].copy(); bad["fixed_idx"] = len(sp)
        mg.set_correspondences(bad)
        mg.compute()


@pytest.mark.parametrize("kept", [False, True])
def test_tracker_cycle_stays_on_device(oracle, product, kept):
    """clip -> align (clipped scene = moving, measurement = fixed) -> merge, three frames; the product side feeds the
    aligner with the scene's device arrays and merges from the aligner's device-side correspondences."""
    kind = abi.SE3_QUAT_RIGHT
    frames = []
    poses = [syn.se3(np.array([0.05 * k, -0.03 * k, 0.01 * k]), np.deg2rad(np.array([0.6 * k, -0.4 * k, 0.8 * k]))) for k in range(4)]
    for k in range(4):
        P, N = syn.scene_3d(30_000, 300 + k)  # the same surfaces, sampled afresh every frame (world frame)
        Xi = syn.se3_inv(poses[k])           # measurement in the robot frame of frame k
        frames.append((np.ascontiguousarray(P @ Xi[:, :3].T + Xi[:, 3], f32), np.ascontiguousarray(N @ Xi[:, :3].T, f32)))
    params = mapping.MergerParams(50.0, 0.01, 10 ** 9)  # merge close points, always append the rest
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05)
    results = []
    for side, b in zip(("oracle", "gpu"), _bindings(oracle, product)):
        al = oracle.OracleAligner(kind) if side == "oracle" else product.MultiAligner(kind)
        si = al.add_slice(cfg)
        scene, clipped, meas = mapping.Scene(b, 3), mapping.Scene(b, 3), mapping.Scene(b, 3)
        mg = mapping.MergerCorrespondenceHomo(b, params)
        cl = mapping.SceneClipperBall(b, range_max=6.0)
        # frame 0 starts the local map: merge without correspondences (merger_correspondence_homo_impl.cpp:30-41)
        meas.set(*frames[0])
        mg.set_scene(scene); mg.set_measurement(meas); mg.set_measurement_in_scene(syn.identity(3))
        mg.compute()
        robot_in_map = syn.identity(3).astype(f32)
        log = []
        for k in range(1, 4):
            meas.set(*frames[k])
            cl.set_full_scene(scene); cl.set_clipped_scene_in_robot(clipped); cl.set_robot_in_local_map(robot_in_map)
            cl.compute()
            if side == "gpu":
                cp, cn, n = clipped.device_arrays()
                al.set_cloud_device("set_moving", si, cp, 16, cn, 16, n, kept=kept)  # (kept: SRRG2_MEM_DEVICE_KEPT, no wait for the ingest)
                mp_, mn_, m = meas.device_arrays()
                al.set_cloud_device("set_fixed", si, mp_, 16, mn_, 16, m, kept=kept)
            else:
                al.set_moving(si, *clipped.get())
                al.set_fixed(si, *meas.get())
            # moving (clipped scene, robot frame of the previous estimate) in fixed (measurement = current robot frame)
            al.set_moving_in_fixed(syn.identity(3))
            al.compute()
            assert al.status() == abi.SUCCESS
            X = al.moving_in_fixed()  # previous robot frame in the current one
            Xm = np.vstack([X, [0, 0, 0, 1]]).astype(np.float64)
            Rm = np.vstack([robot_in_map, [0, 0, 0, 1]]).astype(np.float64)
            robot_in_map = (Rm @ np.linalg.inv(Xm))[:3].astype(f32)
            mg.set_measurement_in_scene(robot_in_map)
            if side == "gpu":
                res = mg.compute_from_aligner(al, si, clipped)
            else:
                c = al.correspondences(si)
                l2g = cl.global_indices()
                flipped = np.zeros(len(c), dtype=c.dtype)
                flipped["fixed_idx"] = l2g[c["moving_idx"]]   # tracker_slice_processor_impl.cpp:177-180
                flipped["moving_idx"] = c["fixed_idx"]
                flipped["response"] = c["response"]
                mg.set_correspondences(flipped)
                res = mg.compute()
            log.append((res, X.copy(), clipped.size()))
        results.append((scene, log, robot_in_map))
    (s_ref, log_ref, pose_ref), (s_gpu, log_gpu, pose_gpu) = results
    for (r, X, nc), (g, Y, mc) in zip(log_ref, log_gpu):
        assert r == g and nc == mc
        assert X.tobytes() == Y.tobytes()
        assert r["num_merged"] > 1000 and r["num_added"] > 0
    _same_scene(s_ref, s_gpu)
    assert pose_ref.tobytes() == pose_gpu.tobytes()
    assert np.max(np.abs(pose_gpu - poses[3].astype(f32))) < 2e-2
