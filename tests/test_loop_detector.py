"""Loop-closure candidate driver: batched alignments + the accept gates of
multi_loop_detector_brute_force_impl.cpp:80-112.  CPU leg on the oracle, gpu leg on the HIP library."""
import numpy as np
import pytest

from helpers import cue_config
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import loop_detector as ld
from srrg2_slam_interfaces_amd import slices as sl
from srrg2_slam_interfaces_amd import synthetic as syn

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def _aligner(backend, oracle):
    if backend == "oracle":
        al = oracle.OracleAligner(abi.SE3_QUAT_RIGHT)
    else:
        import srrg2_slam_interfaces_amd as pkg

        al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
    al.add_slice(cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.35, abi.ROBUST_CAUCHY, 0.05))
    return al


@pytest.mark.parametrize("backend", BACKENDS)
def test_gates_and_closure_records(backend, oracle):
    probs = syn.batch_3d(K=4, n=3000, seed=4400)
    hints = [ld.ClosureHint(100 + k, p["moving"], p["moving_normals"]) for k, p in enumerate(probs)]
    # candidate 2 comes from a different place: its cloud does not overlap -> the aligner itself fails
    hints[2] = ld.ClosureHint(102, probs[2]["moving"] + np.float32(30.0), probs[2]["moving_normals"])
    hints.append(ld.ClosureHint(999, None))  # a hint without a local map is skipped (:66-70)
    pose = syn.se3(np.array([0.5, 0.1, 0.0]), np.array([0.0, 0.0, 0.3])).astype(np.float32)
    det = ld.MultiLoopDetectorBruteForce(_aligner(backend, oracle), relocalize_min_inliers=500,
                                         relocalize_max_chi_inliers=0.005, relocalize_min_inliers_ratio=0.7)
    closures = det.compute(7, probs[0]["fixed"], probs[0]["fixed_normals"], hints, pose)
    assert det.attempted_closures == [100, 101, 102, 103]
    assert [c["target"] for c in closures] == [100, 101, 103]
    assert det.drops == [(102, "ALIGNER DROP [code: 3]")]
    for c, k in zip(closures, (0, 1, 3)):
        assert c["source"] == 7 and c["num_inliers"] >= 500 and c["chi_inliers"] <= 0.005
        assert c["num_inliers"] / c["num_correspondences"] >= 0.7
        assert np.max(np.abs(c["measurement"] - probs[k]["X_gt"])) < 2e-2
        assert np.allclose(sl.compose(c["measurement"], c["pose_in_target"]), pose, atol=1e-5)
        assert np.array_equal(c["information"], np.eye(6, dtype=np.float32))
    # each gate in turn
    det.relocalize_min_inliers = 10 ** 6
    assert det.compute(7, probs[0]["fixed"], probs[0]["fixed_normals"], hints[:2], pose) == []
    assert [d[1] for d in det.drops] == ["NUM_INLIERS DROP"] * 2
    det.relocalize_min_inliers, det.relocalize_max_chi_inliers = 500, 1e-12
    det.compute(7, probs[0]["fixed"], probs[0]["fixed_normals"], hints[:2], pose)
    assert [d[1] for d in det.drops] == ["MAX_CHI_INLIERS DROP"] * 2
    det.relocalize_max_chi_inliers, det.relocalize_min_inliers_ratio = 0.005, 1.01
    det.compute(7, probs[0]["fixed"], probs[0]["fixed_normals"], hints[:2], pose)
    assert [d[1] for d in det.drops] == ["MIN_INLIERS_RATIO DROP"] * 2
    with pytest.raises(RuntimeError):
        ld.MultiLoopDetectorBruteForce(None)


def test_breadth_first_selector():
    """LocalMapSelectorBreadthFirst_::compute, local_map_selector_breadth_first_impl.cpp:12-101: hop-count visit over the
    enabled factors, range growing with the graph distance, aggressive relocalisation beyond a graph distance."""
    n = 30
    est = {k: syn.se3(np.array([0.5 * k, 0.0, 0.0]), np.zeros(3)).astype(np.float32) for k in range(n)}
    # a corridor walked forth and back: maps 29..15 lie next to 0..14
    for k in range(15, n):
        est[k] = syn.se3(np.array([0.5 * (29 - k), 0.3, 0.0]), np.array([0.0, 0.0, np.pi])).astype(np.float32)
    factors = [(k, k + 1, True) for k in range(n - 1)] + [(3, 20, False)]  # a disabled closure must not shorten paths
    sel = ld.LocalMapSelectorBreadthFirst(relocalize_range_scale=2, aggressive_relocalize_graph_distance=10,
                                          aggressive_relocalize_graph_max_range=20,
                                          aggressive_relocalize_range_increase_per_edge=0.1, max_local_map_distance=1.0)
    source = 29
    hints = sel.compute(est, factors, source, est[source])
    assert sel.costs[0] == 29 and sel.costs[28] == 1 and sel.costs[20] == 9
    got = {h["target"]: h for h in hints}
    for k in range(n - 1):
        c = 29 - k
        rng = min(2 * c * 0.1 + 1, 20.0)
        dist = np.linalg.norm((sl.compose(sl.inverse(est[source]), est[k]))[:, 3])
        assert (k in got) == (dist <= rng), (k, dist, rng)
    assert 0 in got and 1 in got and 28 in got and 27 in got  # spatial neighbours far on the graph and graph neighbours
    # aggressive: the initial guess of far-on-the-graph candidates has no translation, near ones keep it
    assert np.all(got[0]["initial_guess"][:, 3] == 0) and got[0]["cost"] == 29
    assert np.allclose(got[28]["initial_guess"], sl.compose(sl.inverse(est[29]), est[28]))
    assert np.array_equal(got[0]["information"], np.eye(6, dtype=np.float32))
    # unreachable local maps are skipped
    est[99] = est[29].copy()
    assert 99 not in {h["target"] for h in sel.compute(est, factors, source, est[source])}
    with pytest.raises(RuntimeError):
        sel.compute(est, factors, 1234, est[source])


def test_relocalizer_without_aligner_uses_detector_statistics():
    """multi_relocalizer_impl.cpp:27-66"""
    I = sl.identity(3)
    far = I.copy(); far[0, 3] = 5.0
    cands = [dict(target=1, pose_in_target=I, chi_inliers=0.004, num_correspondences=900),
             dict(target=2, pose_in_target=far, chi_inliers=0.001, num_correspondences=2000),   # beyond max_translation
             dict(target=3, pose_in_target=I, chi_inliers=0.003, num_correspondences=800),     # fewer correspondences
             dict(target=4, pose_in_target=I, chi_inliers=0.0035, num_correspondences=1000),
             dict(target=5, pose_in_target=I, chi_inliers=0.0050, num_correspondences=3000)]   # higher chi
    rel = ld.MultiRelocalizer(None, max_translation=3.0)
    assert rel.compute(cands) == 4
    assert rel.relocalized_closure["target"] == 4
    assert [d[1] for d in rel.drops] == ["MAX_TRANSITION DROP", "LOW_MIN_CORRESPONDENCE DROP", "HIGH_CHI_INLIERS DROP"]
    assert rel.compute([]) is None and rel.relocalization_map is None


@pytest.mark.parametrize("backend", BACKENDS)
def test_relocalizer_with_aligner(backend, oracle):
    """multi_relocalizer_impl.cpp:68-140: every near candidate is re-aligned against the measurement, gated, and the
    lowest chi per inlier wins; robot_in_local_map = moving_in_fixed^-1."""
    probs = syn.batch_3d(K=3, n=3000, seed=4500)
    cands = []
    for k, p in enumerate(probs):
        X = p["X_gt"].astype(np.float32)
        cands.append(dict(target=200 + k, moving=p["moving"], moving_normals=p["moving_normals"],
                          pose_in_target=sl.inverse(X)))
    # candidate 1: noisy copy of its cloud -> higher chi per inlier than the others
    rng = np.random.default_rng(5)
    cands[1]["moving"] = (cands[1]["moving"] + rng.normal(scale=0.004, size=cands[1]["moving"].shape)).astype(np.float32)
    far = sl.identity(3); far[0, 3] = 10.0
    cands.append(dict(target=299, moving=probs[0]["moving"], moving_normals=probs[0]["moving_normals"], pose_in_target=far))
    rel = ld.MultiRelocalizer(_aligner(backend, oracle), max_translation=3.0, relocalize_min_inliers=500,
                              relocalize_max_chi_inliers=0.005, relocalize_min_inliers_ratio=0.7)
    best = rel.compute(cands, probs[0]["fixed"], probs[0]["fixed_normals"])
    assert best in (200, 202) and rel.relocalized_closure["target"] == best
    assert ("299", "MAX_TRANSITION DROP") not in rel.drops and (299, "MAX_TRANSITION DROP") in rel.drops
    k = best - 200
    assert np.max(np.abs(sl.inverse(rel.robot_in_local_map) - probs[k]["X_gt"])) < 2e-2
    rel.relocalize_min_inliers = 10 ** 6
    assert rel.compute(cands, probs[0]["fixed"], probs[0]["fixed_normals"]) is None
    assert [d[1] for d in rel.drops].count("NUM_INLIERS DROP") == 3
