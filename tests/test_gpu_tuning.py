"""-m gpu: the strategy knobs of a handle (srrg2_aligner_tuning, include/srrg2_slam_amd.h) set through the ABI instead of the
environment: defaults, round trip, and -- what the knobs promise -- the same bits under every setting.  The search passes of
batches are the round-3 kernel k_icp_step_tile (tiles of 416 / 504 candidates in LDS) or k_icp_step (per-lane gathers)."""
import numpy as np
import pytest

from helpers import assert_same_run, cue_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def test_tuning_defaults_and_round_trip(product, monkeypatch):
    for name in ("SRRG2_AMD_LDS_TILE", "SRRG2_AMD_FAST_FROM", "SRRG2_AMD_CELL_TARGET", "SRRG2_AMD_TUNE", "SRRG2_AMD_SEARCH_LISTS"):
        monkeypatch.delenv(name, raising=False)
    al = product.MultiAligner(abi.SE3_QUAT_RIGHT)
    t = al.tuning()
    assert (t.strategy_mask, t.queue_probe_iteration, t.small_max_points, t.fast_from_iteration) == (0, 1, 1024, 3)
    assert (t.lds_tile, t.fast_gather, t.msort_key_bits, t.queue_min_points) == (-1, -1, 0, 90000)
    assert (t.search_lists, t.search_team, t.batch_pipeline, t.fused_control) == (-1, 0, -1, -1)
    assert t.cell_target == pytest.approx(8.0)
    al.set_tuning(lds_tile=2, fast_from_iteration=1, cell_target=6.0)
    t = al.tuning()
    assert (t.lds_tile, t.fast_from_iteration) == (2, 1) and t.cell_target == pytest.approx(6.0)
    assert t.small_max_points == 1024  # (untouched fields keep their values)
    with pytest.raises(KeyError):
        al.set_tuning(no_such_knob=1)
    # iteration 0 has no previous neighbours for the converged-pass kernel to certify against: rejected, and an
    # environment value below 1 is clamped when the handle is created
    with pytest.raises(Exception):
        al.set_tuning(fast_from_iteration=0)
    assert al.tuning().fast_from_iteration == 1
    monkeypatch.setenv("SRRG2_AMD_FAST_FROM", "0")
    monkeypatch.setenv("SRRG2_AMD_MSORT_BITS", "40")
    t0 = product.MultiAligner(abi.SE3_QUAT_RIGHT).tuning()
    assert (t0.fast_from_iteration, t0.msort_key_bits) == (1, 18)
    monkeypatch.delenv("SRRG2_AMD_FAST_FROM")
    monkeypatch.delenv("SRRG2_AMD_MSORT_BITS")
    # the environment is read once, when a handle is created
    monkeypatch.setenv("SRRG2_AMD_LDS_TILE", "0")
    assert al.tuning().lds_tile == 2
    assert product.MultiAligner(abi.SE3_QUAT_RIGHT).tuning().lds_tile == 0


@pytest.mark.parametrize("plane", [True, False])
def test_batches_give_the_same_bits_under_every_search_pass_kernel(oracle, product, plane):
    """10 ragged alignments per launch against a partially overlapping fixed cloud: per-lane gathers, tiles of 416 and of
    504 candidates, two cell sizes -- and the oracle."""
    K = 10
    probs = syn.batch_3d(K=K, n=6000, seed=9100, shared_fixed_group=64, t_max=0.12, rpy_max_deg=3.0)
    fixed, fixed_n = probs[0]["fixed"], probs[0]["fixed_normals"]
    keep = fixed[:, 1] <= np.quantile(fixed[:, 1], 0.8)
    fixed, fixed_n = fixed[keep], fixed_n[keep]
    movs = [p["moving"][: 6000 - 311 * k] for k, p in enumerate(probs)]
    nrms = [p["moving_normals"][: len(m)] for p, m in zip(probs, movs)]
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE if plane else abi.SLICE_P2P, 0.3, abi.ROBUST_CAUCHY, 0.05, 0.7)

    def run(al, **knobs):
        if knobs:
            al.set_tuning(**knobs)
        al.set_params(max_iterations=6)
        si = al.add_slice(cfg)
        al.set_fixed(si, fixed, fixed_n)
        return al.compute_batch(movs, [syn.identity(3)] * K, nrms)

    ref = run(oracle.OracleAligner(abi.SE3_QUAT_RIGHT))
    for knobs in ({"lds_tile": 0, "search_lists": 0}, {"lds_tile": 1, "search_lists": 0}, {"lds_tile": 2, "search_lists": 0},
                  {"lds_tile": 1, "cell_target": 3.0, "search_lists": 0}, {"lds_tile": 1, "msort_key_bits": -1, "search_lists": 0},
                  {"lds_tile": 1, "fast_from_iteration": 1, "search_lists": 0},
                  {"search_lists": 1}, {"search_lists": 2, "cell_target": 3.0}, {"search_lists": 1, "cell_target": 20.0},
                  {"search_lists": 1, "fast_from_iteration": 1}, {"search_lists": 1, "fast_from_iteration": 100},
                  {"search_lists": 1, "search_team": 4}, {"search_lists": 1, "search_team": 4, "fast_from_iteration": 100},
                  {"batch_pipeline": 0}, {"batch_pipeline": 1}, {"batch_pipeline": 1, "search_lists": 0},
                  {"batch_pipeline": 3}, {"batch_pipeline": 4}, {"batch_pipeline": 8}, {"batch_pipeline": 5, "search_lists": 0},
                  {"fused_control": 0}, {"fused_control": 1}, {"fused_control": 1, "batch_pipeline": 0},
                  {"fused_control": 1, "fast_from_iteration": 1}, {"fused_control": 1, "fast_from_iteration": 100},
                  {"fused_control": 0, "batch_pipeline": 0},
                  {"batch_pipeline": 0, "search_lists": 0, "lds_tile": 0}):
        got = run(product.MultiAligner(abi.SE3_QUAT_RIGHT), **knobs)
        for r, g in zip(ref, got):
            assert r["status"] == g["status"], knobs
            assert r["num_iterations"] == g["num_iterations"], knobs
            assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes(), knobs
            assert r["last"] == g["last"], knobs


def test_single_alignment_on_wave_tiles(oracle, product):
    """lds_tile = 1 with the deferred-search kernel off puts a single alignment on k_icp_step_tile too"""
    d = syn.cloud_pair_3d(n=30000, seed=9200)
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05, 0.8)
    runs = []
    for al, knobs in ((oracle.OracleAligner(abi.SE3_QUAT_RIGHT), None),
                      (product.MultiAligner(abi.SE3_QUAT_RIGHT), {"lds_tile": 1, "queue_min_points": 1 << 30}),
                      (product.MultiAligner(abi.SE3_QUAT_RIGHT), {"lds_tile": 2, "queue_min_points": 1 << 30})):
        if knobs:
            al.set_tuning(**knobs)
        al.set_params(max_iterations=8)
        setup_pair(al, d, cfg)
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1])
    assert_same_run(runs[0], runs[2])


def test_single_alignments_get_their_lists_from_the_second_compute_on(oracle, product, monkeypatch):
    """automatic policy (search_lists = -1): the first compute() on a fixed cloud walks the grid (a tracker sets a new fixed
    cloud every frame and should not pay for lists it uses once), the following ones the cell neighbour lists; a new fixed
    cloud starts over.  Same bits every time -- and after a pipelined batch on the same handle."""
    for name in ("SRRG2_AMD_SEARCH_LISTS", "SRRG2_AMD_SEARCH_TEAM", "SRRG2_AMD_BATCH_PIPELINE"):
        monkeypatch.delenv(name, raising=False)
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=20000, seed=9300)
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05, 0.8)
    ref = oracle.OracleAligner(kind)
    setup_pair(ref, d, cfg)
    ref.compute()
    al = product.MultiAligner(kind)
    setup_pair(al, d, cfg)
    for _ in range(3):  # grid walk, then lists twice
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        assert_same_run(ref, al)
    # a batch of 9 (pipelined: 5 + 4) against the same fixed cloud, then the single alignment again
    movs = [d["moving"][: 20000 - 777 * k] for k in range(9)]
    nrms = [d["moving_normals"][: len(m)] for m in movs]
    got = al.compute_batch(movs, [syn.identity(3)] * 9, nrms)
    assert got[0]["moving_in_fixed"].tobytes() == ref.moving_in_fixed().tobytes()
    al.set_moving(0, d["moving"], d["moving_normals"])
    al.set_moving_in_fixed(syn.identity(3))
    al.compute()
    assert_same_run(ref, al)
    # a new fixed cloud: back to the grid walk, same answer as a fresh handle
    d2 = syn.cloud_pair_3d(n=15000, seed=9301)
    ref2 = oracle.OracleAligner(kind)
    setup_pair(ref2, d2, cfg)
    ref2.compute()
    al.set_fixed(0, d2["fixed"], d2["fixed_normals"])
    al.set_moving(0, d2["moving"], d2["moving_normals"])
    for _ in range(2):
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        assert_same_run(ref2, al)
