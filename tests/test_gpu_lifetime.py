"""-m gpu: handles release what they allocate (create / use / destroy in a loop leaves the device memory level where it
was), and several handles live side by side."""
import numpy as np
import pytest

from helpers import cue_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import mapping
from srrg2_slam_interfaces_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _free_bytes():
    """hipMemGetInfo of the HIP runtime the product library itself is linked against"""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    assert hip.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_no_device_memory_leak(product):
    d = syn.cloud_pair_3d(n=50_000, seed=31)
    g = syn.pose_graph_3d(V=2000, E=6000, seed=32)

    def cycle():
        al = product.MultiAligner(abi.SE3_QUAT_RIGHT)
        setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05))
        al.compute()
        al.correspondences(0)
        b = product.scene_binding(0)
        s1, s2 = mapping.Scene(b, 3), mapping.Scene(b, 3)
        s1.set(d["fixed"], d["fixed_normals"])
        cl = mapping.SceneClipperBall(b, 5.0)
        cl.set_full_scene(s1); cl.set_clipped_scene_in_robot(s2); cl.set_robot_in_local_map(syn.identity(3))
        cl.compute()
        pg = product.PoseGraph(abi.SE3_QUAT_RIGHT, 0)
        pg.set_graph(g["poses_init"], g["ij"], g["Z"])
        pg.solve()
        for h in (al, s1, s2, pg):
            h.close()

    for _ in range(3):
        cycle()  # allocator warm-up
    before = _free_bytes()
    for _ in range(25):
        cycle()
    after = _free_bytes()
    assert before - after < 8 << 20, (before, after)  # nothing accumulates (the clouds alone are ~10 MB per cycle)


def test_handles_side_by_side(oracle, product):
    """two aligners with different clouds interleaved: neither disturbs the other's state"""
    d1 = syn.cloud_pair_3d(n=8000, seed=41)
    d2 = syn.cloud_pair_3d(n=12000, seed=42, t=(0.02, 0.04, -0.03))
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05)
    a1, a2 = product.MultiAligner(abi.SE3_QUAT_RIGHT), product.MultiAligner(abi.SE3_QUAT_RIGHT)
    setup_pair(a1, d1, cfg)
    setup_pair(a2, d2, cfg)
    a1.compute(); a2.compute()
    X1, X2 = a1.moving_in_fixed().copy(), a2.moving_in_fixed().copy()
    c1 = a1.correspondences(0)
    a2.set_moving_in_fixed(syn.identity(3)); a2.compute()
    assert np.array_equal(a1.correspondences(0), c1) and a1.moving_in_fixed().tobytes() == X1.tobytes()
    a1.set_moving_in_fixed(syn.identity(3)); a1.compute()
    assert a1.moving_in_fixed().tobytes() == X1.tobytes() and a2.moving_in_fixed().tobytes() == X2.tobytes()
    r1 = oracle.OracleAligner(abi.SE3_QUAT_RIGHT)
    setup_pair(r1, d1, cfg)
    r1.compute()
    assert r1.moving_in_fixed().tobytes() == X1.tobytes()


def test_distinct_handles_from_distinct_threads(product):
    """one handle = one non-thread-safe object; distinct handles are usable from distinct threads:
    four threads align concurrently (ctypes releases the GIL), results equal the sequential ones bit for bit."""
    import threading

    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05)
    data = [syn.cloud_pair_3d(n=20000 + 3000 * k, seed=60 + k) for k in range(4)]

    def run(k, out, reps):
        al = product.MultiAligner(abi.SE3_QUAT_RIGHT)
        setup_pair(al, data[k], cfg)
        for _ in range(reps):
            al.set_moving_in_fixed(syn.identity(3))
            al.compute()
        out[k] = (al.moving_in_fixed().copy(), al.correspondences(0).copy(), al.status())
        al.close()

    seq, par = {}, {}
    for k in range(4):
        run(k, seq, 1)
    threads = [threading.Thread(target=run, args=(k, par, 20)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(4):
        assert par[k][2] == seq[k][2] == abi.SUCCESS
        assert par[k][0].tobytes() == seq[k][0].tobytes()
        assert np.array_equal(par[k][1], seq[k][1])
