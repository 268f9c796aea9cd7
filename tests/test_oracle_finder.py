"""CPU: the oracle's voxel-grid finder must equal its brute-force finder bit for bit (SURVEY.md section 8c:
'grid == brute force' is the self-consistency pin of the correspondence search)."""
import numpy as np
import pytest

from helpers import cue_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn


def _corr(oracle, kind, data, cfg, brute, guess=None):
    al = oracle.OracleAligner(kind)
    al.set_bruteforce(brute)
    setup_pair(al, data, cfg, guess)
    acc, k = al.linearize_once(0)
    return al.correspondences(0), acc, k


@pytest.mark.parametrize("gate,cell", [(0.25, 0.0), (0.25, 0.3), (0.05, 0.0), (1.0, 0.11), (0.6, 2.0)])
def test_grid_equals_bruteforce_3d(oracle, gate, cell):
    d = syn.cloud_pair_3d(n=3000, seed=11)
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, gate, normal_cos=0.5)
    cfg.finder_cell_size = cell
    cb, accb, kb = _corr(oracle, abi.SE3_QUAT_RIGHT, d, cfg, True)
    cg, accg, kg = _corr(oracle, abi.SE3_QUAT_RIGHT, d, cfg, False)
    assert len(cb) > 100
    assert cb.tobytes() == cg.tobytes()
    assert kb == kg and np.array_equal(accb, accg)


@pytest.mark.parametrize("gate", [0.1, 0.5, 3.0])
def test_grid_equals_bruteforce_2d(oracle, gate):
    d = syn.scan_pair_2d(beams=1000, sigma=0.01)
    cfg = cue_config(abi.SE2_RIGHT, abi.SLICE_P2P, gate)
    cb, accb, _ = _corr(oracle, abi.SE2_RIGHT, d, cfg, True)
    cg, accg, _ = _corr(oracle, abi.SE2_RIGHT, d, cfg, False)
    assert len(cb) > 100 and cb.tobytes() == cg.tobytes() and np.array_equal(accb, accg)


def test_ties_pick_smallest_fixed_index_and_gate_is_inclusive(oracle):
    fixed = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, 0, 5], [0, 0, 5]], np.float32)
    moving = np.array([[0, 0, 0], [0, 0, 4], [10, 10, 10], [np.nan, 0, 0]], np.float32)
    d = {"fixed": fixed, "moving": moving}
    for brute in (True, False):
        cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2P, 1.0)
        c, _, _ = _corr(oracle, abi.SE3_QUAT_RIGHT, d, cfg, brute)
        # point 0: three fixed points at distance exactly 1 (== gate, inclusive) -> smallest index 0
        # point 1: two coincident fixed points 3 and 4 -> 3; point 2 too far; point 3 is NaN -> skipped
        assert c["moving_idx"].tolist() == [0, 1]
        assert c["fixed_idx"].tolist() == [0, 3]
        assert c["response"].tolist() == [1.0, 1.0]


def test_nonfinite_fixed_points_never_match(oracle):
    d = syn.cloud_pair_3d(n=2000, seed=5)
    d["fixed"][::3, 2] = np.inf
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2P, 0.5)
    cb, _, _ = _corr(oracle, abi.SE3_QUAT_RIGHT, d, cfg, True)
    cg, _, _ = _corr(oracle, abi.SE3_QUAT_RIGHT, d, cfg, False)
    assert cb.tobytes() == cg.tobytes() and not np.any(cb["fixed_idx"] % 3 == 0)
