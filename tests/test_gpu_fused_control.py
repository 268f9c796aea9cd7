"""Fused control steps (round 5): the control step of an ICP iteration -- slot-set sums, one Gauss-Newton
step (multi_aligner_impl.cpp:112-121), statistics, termination criterion (aligner_termination_criteria_impl.cpp:24-65) --
runs on one wave inside the next iteration's first pass kernel, with matrices spread over lanes.  Every outcome of the state
machine through that path, against the oracle AND against the same library with one control launch per iteration."""
import numpy as np
import pytest

from helpers import assert_same_run, cue_config, prior_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _runs(oracle, product, kind, build, knobs_list):
    out = []
    for make, knobs in [(lambda: oracle.OracleAligner(kind), None)] + [(lambda: product.MultiAligner(kind), k) for k in knobs_list]:
        al = make()
        if knobs:
            al.set_tuning(**knobs)
        build(al)
        out.append(al)
    return out


FUSED = {"search_lists": 2, "fused_control": 1}    # lists (hence the fused launches) from the first compute() on
UNFUSED = {"search_lists": 2, "fused_control": 0}


@pytest.mark.parametrize("kind,slice_kind", [(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE), (abi.SE3_EULER_RIGHT, abi.SLICE_P2P),
                                             (abi.SE2_RIGHT, abi.SLICE_P2P), (abi.SE2_RIGHT, abi.SLICE_P2PLANE)])
def test_termination_criterion_inlier_only_run_and_pruning(oracle, product, kind, slice_kind):
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=3000, sigma=0.01, seed=1234)
        gate, thr = 0.5, 0.002
        if slice_kind == abi.SLICE_P2PLANE:  # (a 2-D "plane" factor needs normals: the scans' own, from neighbouring beams)
            for which in ("fixed", "moving"):
                p = d[which]
                t = np.roll(p, -1, axis=0) - np.roll(p, 1, axis=0)
                n = np.stack([-t[:, 1], t[:, 0]], axis=1)
                d[which + "_normals"] = (n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-9)).astype(np.float32)
    else:
        d = syn.cloud_pair_3d(n=15000, seed=2200, noise_sigma=0.01)
        gate, thr = 0.25, 0.0005
    cfg = cue_config(kind, slice_kind, gate, abi.ROBUST_CAUCHY, thr)

    def build(al):
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True)
        al.set_termination_criteria(abi.default_termination_params())
        setup_pair(al, d, cfg)
        al.compute()
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()  # (a second compute() on the same clouds: the same path again, from a used handle)

    ref, fused, unfused = _runs(oracle, product, kind, build, [FUSED, UNFUSED])
    assert ref.status() == abi.SUCCESS
    if kind != abi.SE2_RIGHT or slice_kind == abi.SLICE_P2PLANE:
        assert len(ref.iteration_stats()) < 24  # (the criterion fired)
    assert_same_run(ref, fused)
    assert_same_run(ref, unfused)
    assert fused.information().tobytes() == unfused.information().tobytes()


def test_every_way_an_alignment_of_a_batch_can_end(oracle, product):
    """One launch, five fates: converges; empty cloud (Fail: no statistics, multi_aligner_impl.cpp:75-78); a guess so far off
    that iteration 0 finds nothing (Fail); few points and a large min_num_inliers (NotEnoughInliers, :81-85); a cloud that loses
    its correspondences after a few iterations is not constructible on purpose -- instead min_num_correspondences above what a
    partial overlap offers, which ends NotEnoughCorrespondences at iteration 0 -> Fail as well (the quirk of :75-78)."""
    kind = abi.SE3_QUAT_RIGHT
    probs = syn.batch_3d(K=6, n=9000, seed=8800, shared_fixed_group=64, t_max=0.1, rpy_max_deg=2.0)
    fixed, fixed_n = probs[0]["fixed"], probs[0]["fixed_normals"]
    far = syn.se3(np.array([30.0, 0.0, 0.0]), np.zeros(3)).astype(np.float32)
    movs = [probs[0]["moving"], probs[1]["moving"][:0], probs[2]["moving"], probs[3]["moving"][::110][:80], probs[4]["moving"][:2500],
            probs[5]["moving"]]
    nrms = [probs[0]["moving_normals"], probs[1]["moving_normals"][:0], probs[2]["moving_normals"], probs[3]["moving_normals"][::110][:80],
            probs[4]["moving_normals"][:2500], probs[5]["moving_normals"]]
    guesses = [syn.identity(3), syn.identity(3), far, syn.identity(3), syn.identity(3), syn.identity(3)]
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.3, abi.ROBUST_CAUCHY, 0.05, 0.7)

    def run(al):
        al.set_params(max_iterations=7, min_num_inliers=100)
        al.add_slice(cfg)
        al.set_fixed(0, fixed, fixed_n)
        return al.compute_batch(movs, guesses, nrms)

    ref = run(oracle.OracleAligner(kind))
    assert [r["status"] for r in ref] == [abi.SUCCESS, abi.FAIL, abi.FAIL, abi.NOT_ENOUGH_INLIERS, abi.SUCCESS, abi.SUCCESS]
    for knobs in (FUSED, UNFUSED, dict(FUSED, batch_pipeline=0), dict(FUSED, batch_pipeline=3)):
        al = product.MultiAligner(kind)
        al.set_tuning(**knobs)
        got = run(al)
        for r, g in zip(ref, got):
            assert r["status"] == g["status"] and r["num_iterations"] == g["num_iterations"], knobs
            assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes(), knobs
            assert r["last"] == g["last"] and r["num_correspondences"] == g["num_correspondences"], knobs
            assert np.asarray(r["information"]).tobytes() == np.asarray(g["information"]).tobytes(), knobs


def test_a_solver_failure_keeps_the_estimate(oracle, product):
    """All moving points on one line seen through a point-to-plane factor with parallel normals: H is singular, the L D L^T
    meets a pivot <= 0, solver status != Success and the estimate stays (multi_aligner_impl.cpp:118-121) -- through the
    lane-distributed factorisation too."""
    kind = abi.SE3_QUAT_RIGHT
    n = 4000
    x = np.linspace(-3, 3, n, dtype=np.float32)
    fixed = np.stack([x, np.zeros(n, np.float32), np.zeros(n, np.float32)], axis=1)
    fn = np.tile(np.array([[0, 0, 1]], np.float32), (n, 1))
    moving = fixed + np.array([0.01, 0.0, 0.02], np.float32)
    d = {"fixed": fixed, "fixed_normals": fn, "moving": moving, "moving_normals": fn.copy()}
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.25)

    def build(al):
        al.set_params(max_iterations=4, min_num_inliers=10)
        setup_pair(al, d, cfg)
        al.compute()

    ref, fused, unfused = _runs(oracle, product, kind, build, [FUSED, UNFUSED])
    assert all(s_["solver_status"] != 0 for s_ in ref.iteration_stats())
    assert_same_run(ref, fused)
    assert_same_run(ref, unfused)


def test_projective_slices_sharing_one_association(oracle, product):
    """C3's shape at a quarter of the resolution: the control step of BOTH slices in the z-buffer kernel's prologue"""
    kind = abi.SE3_QUAT_RIGHT
    r = syn.rgbd_pair(rows=120, cols=160, seed=3100)

    def build(al):
        al.set_params(max_iterations=8, min_num_inliers=10, enable_inlier_only_runs=True)
        for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
            c = abi.default_slice_config(kind)
            c.kind, c.finder, c.finder_max_distance = sk, abi.FINDER_PROJECTIVE, 0.05
            c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05 if sk == abi.SLICE_P2PLANE else 4.0
            for i, v in enumerate(r["K"].reshape(-1)):
                c.camera_matrix[i] = v
            c.image_rows, c.image_cols, c.depth_min, c.depth_max = r["rows"], r["cols"], r["depth_min"], r["depth_max"]
            si = al.add_slice(c)
            if si == 0 or not hasattr(al, "share_clouds") or isinstance(al, oracle.OracleAligner):
                al.set_fixed(si, r["fixed"], r["fixed_normals"])
                al.set_moving(si, r["moving"], r["moving_normals"])
            else:
                al.share_clouds(si, 0)
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()

    ref, fused, unfused = _runs(oracle, product, kind, build, [{"fused_control": 1}, {"fused_control": 0}])
    assert ref.status() == abi.SUCCESS
    assert_same_run(ref, fused, slices=(0, 1))
    assert_same_run(ref, unfused, slices=(0, 1))


@pytest.mark.parametrize("slice_kind", [abi.SLICE_P2PLANE, abi.SLICE_P2P])
def test_partial_overlap_through_the_one_linearisation_flow(oracle, product, slice_kind):
    """A single alignment on the lists with fused control steps runs its converged passes as certificates -> searches -> ONE
    linearisation of all lanes (icp_step_fast_body, `defer`).  A cropped fixed cloud and a large initial error keep MANY
    certificates failing per wave for several passes (the pooled list search inside that flow, not only the four-at-a-time
    team scans), neighbours change and points drop out of the gate: every iteration must still be the from-scratch search's."""
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=30000, seed=2601, t=(0.08, -0.05, 0.03), rpy_deg=(1.5, -2.0, 2.5))
    d = {k: v.copy() for k, v in d.items()}
    keep = d["fixed"][:, 0] <= np.quantile(d["fixed"][:, 0], 0.6)
    d["fixed"], d["fixed_normals"] = d["fixed"][keep], d["fixed_normals"][keep]
    cfg = cue_config(kind, slice_kind, 0.25, abi.ROBUST_CAUCHY, 0.05)

    def build(al):
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True)
        setup_pair(al, d, cfg)
        al.compute()

    early = dict(FUSED, fast_from_iteration=1)  # (the converged-pass kernel while the estimate still moves)
    ref, fused, fused_early, unfused = _runs(oracle, product, kind, build, [FUSED, early, UNFUSED])
    assert ref.status() == abi.SUCCESS
    for run in (fused, fused_early, unfused):
        assert_same_run(ref, run)
    assert fused.information().tobytes() == unfused.information().tobytes() == fused_early.information().tobytes()


def test_polling_waves_apply_the_control_step_themselves():
    """a fused pass whose record is stale polls workgroup (problem, 0) of the SAME launch -- dispatch
    order is not a HIP guarantee.  Past SRRG2_FUSED_POLL_LIMIT polls the polling wave applies the control step itself, in
    registers, from inputs that are read-only for the whole launch (the previous epoch's records, the previous round's slot
    sets: three buffers, zeroed two rounds later), and proceeds: no trap, no dependence on scheduling.  Proven on a library
    built with -DSRRG2_FUSED_STALL (the designated wave sleeps ~0.3 ms first): a single alignment with the termination criterion
    and an inlier-only run, SE(2), a batch with an empty cloud, two projective slices on one association -- the oracle's bits in
    every scenario, and the fallback counter moves in each."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stall = os.path.join(root, "srrg2_slam_interfaces_amd", "lib", "libsrrg2_slam_amd_stall.so")
    assert os.path.exists(stall), "build it with `make -C srrg2_slam_interfaces_amd/csrc stall` (__graft_entry__.build() does)"
    env = dict(os.environ, SRRG2_AMD_LIB=stall)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers_scripts", "fused_stall_check.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "STALL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("kind,slice_kind", [(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE), (abi.SE2_RIGHT, abi.SLICE_P2P)])
def test_grid_search_passes_with_fused_control_steps(oracle, product, kind, slice_kind, monkeypatch):
    """Round 6: the search pass on the GRID (no cell neighbour lists: a tracker's first compute() on every new fixed cloud) has a
    fused instantiation too (k_icp_step_fused: the control step of the previous iteration in its prologue, open points finished
    inside the kernel instead of the deferred-search queue).  run_compute takes it on the evidence of the handle's previous
    compute(); SRRG2_AMD_FUSED_GRID_MAX=-1 (read at create) forces it.  Termination criterion, inlier-only run and pruning, a
    cropped fixed cloud (many far searches: what the queue exists for), a second compute() on a new fixed cloud: the oracle's bits."""
    monkeypatch.setenv("SRRG2_AMD_FUSED_GRID_MAX", "-1")
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=3000, sigma=0.01, seed=1234)
        gate, thr = 0.5, 0.002
    else:
        d = syn.cloud_pair_3d(n=110000, seed=2200, noise_sigma=0.01, t=(0.08, -0.05, 0.03), rpy_deg=(1.5, -2.0, 2.5))
        d = {k: v.copy() for k, v in d.items()}
        keep = d["fixed"][:, 0] <= np.quantile(d["fixed"][:, 0], 0.7)
        d["fixed"], d["fixed_normals"] = d["fixed"][keep], d["fixed_normals"][keep]
        gate, thr = 0.25, 0.0005
    cfg = cue_config(kind, slice_kind, gate, abi.ROBUST_CAUCHY, thr)

    def build(al):
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True)
        al.set_termination_criteria(abi.default_termination_params())
        setup_pair(al, d, cfg)
        al.compute()
        al.set_fixed(0, d["fixed"], d.get("fixed_normals"))  # (a new fixed cloud: no lists again)
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()

    grid = {"search_lists": 0, "fused_control": 1}  # (never lists: every search pass on the grid kernel)
    ref, fused, unfused = _runs(oracle, product, kind, build, [grid, dict(grid, fused_control=0)])
    assert ref.status() == abi.SUCCESS
    assert_same_run(ref, fused)
    assert_same_run(ref, unfused)
    assert fused.information().tobytes() == unfused.information().tobytes()


LAUNCHED_PRIORS = {"search_lists": 2, "fused_control": 2}  # (cue-only aligners fused, prior + cue aligners on control launches)


def _prior_scenario(kind, slice_kind):
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=3000, sigma=0.01, seed=1234)
        gate, thr = 0.5, 0.002
        if slice_kind == abi.SLICE_P2PLANE:
            for which in ("fixed", "moving"):
                p = d[which]
                t = np.roll(p, -1, axis=0) - np.roll(p, 1, axis=0)
                n = np.stack([-t[:, 1], t[:, 0]], axis=1)
                d[which + "_normals"] = (n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        Za = syn.se2(0.03, -0.02, 0.01).astype(np.float32)
        Zb = syn.se2(-0.05, 0.04, -0.02).astype(np.float32)
        info_a, info_b = [10.0, 10.0, 100.0], [3.0, 5.0, 40.0]
    else:
        d = syn.cloud_pair_3d(n=15000, seed=2200, noise_sigma=0.01)
        gate, thr = 0.25, 0.0005
        Za = syn.se3(np.array([0.04, -0.02, 0.01]), np.deg2rad([0.5, -1.0, 1.5])).astype(np.float32)
        Zb = syn.se3(np.array([-0.03, 0.05, 0.02]), np.deg2rad([-1.0, 0.5, -0.7])).astype(np.float32)
        info_a, info_b = [10, 10, 10, 100, 100, 100], [3, 4, 5, 30, 40, 50]
    return d, cue_config(kind, slice_kind, gate, abi.ROBUST_CAUCHY, thr), (Za, info_a), (Zb, info_b)


@pytest.mark.parametrize("kind,slice_kind", [(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE), (abi.SE3_QUAT_RIGHT, abi.SLICE_P2P),
                                             (abi.SE2_RIGHT, abi.SLICE_P2P), (abi.SE2_RIGHT, abi.SLICE_P2PLANE)])
@pytest.mark.parametrize("layout", ["cue_prior", "prior_cue", "prior_cue_prior", "cue_prior_prior_robust"])
def test_prior_slices_next_to_the_cue_slice(oracle, product, kind, slice_kind, layout):
    """Round 6, late (VERDICT r5 missing #5): an aligner with prior slices next to its cue slice -- an odometry prior, a motion
    model: what a tracker configures (S/instances.cpp:35-38) -- has fused control steps too: the control wave linearises the prior
    factors itself (wave_prior: prior_linearize with H spread over the lanes), in slice order.  Priors before and behind the cue
    slice, two of them (three terms: the order of the sums shows), one that overrides the initial guess
    (aligner_slice_odometry_prior.cpp:19,34), one with a robustifier that kernelises it (an outlier in the statistics; Clamp in the
    inlier-only run suppresses its weight), the termination criterion, the inlier-only run, a second compute() on the used handle:
    the oracle's bits, through the fused steps, through control launches (fused_control = 2) and without fused steps at all."""
    d, cfg, (Za, info_a), (Zb, info_b) = _prior_scenario(kind, slice_kind)

    def build(al):
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True)
        al.set_termination_criteria(abi.default_termination_params())
        priors = []
        if layout.startswith("prior"):
            priors.append((al.add_slice(prior_config(kind, info=info_a, sets_guess=1)), Za))
        cue = setup_pair(al, d, cfg)
        if layout in ("cue_prior", "prior_cue_prior"):
            priors.append((al.add_slice(prior_config(kind, info=info_b, sets_guess=0)), Zb))
        if layout == "cue_prior_prior_robust":
            priors.append((al.add_slice(prior_config(kind, info=info_a, sets_guess=0)), Za))
            c = prior_config(kind, info=info_b, sets_guess=0)
            c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 1e-4  # (chi of Zb against the estimate is above it)
            priors.append((al.add_slice(c), Zb))
        for pi, Z in priors:
            al.set_prior_measurement(pi, Z)
        al.compute()
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        al._cue = cue

    ref, fused, launched, unfused = _runs(oracle, product, kind, build, [FUSED, LAUNCHED_PRIORS, UNFUSED])
    assert ref.status() == abi.SUCCESS
    if layout == "cue_prior_prior_robust":
        assert any(s_["num_outliers"] > 0 for s_ in ref.iteration_stats())
    for run in (fused, launched, unfused):
        assert_same_run(ref, run, slices=(ref._cue,))
    assert fused.information().tobytes() == unfused.information().tobytes() == launched.information().tobytes()


@pytest.mark.parametrize("kind,slice_kind", [(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE), (abi.SE2_RIGHT, abi.SLICE_P2P)])
def test_prior_slices_on_a_new_fixed_cloud_and_in_a_batch(oracle, product, kind, slice_kind, monkeypatch):
    """... on the grid search passes (a tracker's frame: a new fixed cloud every compute(), k_icp_step_fused) and -- SE(3) -- for the
    alignments of a batch (compute_batch: one cue slice plus prior slices), where every alignment's control wave linearises the
    same prior at ITS estimate."""
    monkeypatch.setenv("SRRG2_AMD_FUSED_GRID_MAX", "-1")
    d, cfg, (Za, info_a), _ = _prior_scenario(kind, slice_kind)

    def build(al):
        al.set_params(max_iterations=10, min_num_inliers=10, enable_inlier_only_runs=True)
        cue = setup_pair(al, d, cfg)
        pi = al.add_slice(prior_config(kind, info=info_a, sets_guess=0))
        al.set_prior_measurement(pi, Za)
        al.compute()
        al.set_fixed(cue, d["fixed"], d.get("fixed_normals"))  # (a new fixed cloud: no lists)
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()

    grid = {"search_lists": 0, "fused_control": 1}
    ref, fused, launched = _runs(oracle, product, kind, build, [grid, dict(grid, fused_control=2)])
    assert ref.status() == abi.SUCCESS
    assert_same_run(ref, fused)
    assert_same_run(ref, launched)
    if kind != abi.SE3_QUAT_RIGHT:
        return
    probs = syn.batch_3d(K=20, n=6000, seed=8810, shared_fixed_group=64, t_max=0.1, rpy_max_deg=2.0)
    movs = [p["moving"] for p in probs]
    nrms = [p["moving_normals"] for p in probs]
    guesses = [syn.identity(3)] * len(probs)

    def run(al):
        al.set_params(max_iterations=7, min_num_inliers=100)
        al.add_slice(cue_config(kind, abi.SLICE_P2PLANE, 0.3, abi.ROBUST_CAUCHY, 0.05, 0.7))
        al.set_fixed(0, probs[0]["fixed"], probs[0]["fixed_normals"])
        pi = al.add_slice(prior_config(kind, info=info_a, sets_guess=0))
        al.set_prior_measurement(pi, Za)
        return al.compute_batch(movs, guesses, nrms)

    want = run(oracle.OracleAligner(kind))
    for knobs in (FUSED, LAUNCHED_PRIORS, dict(FUSED, batch_pipeline=3)):
        al = product.MultiAligner(kind)
        al.set_tuning(**knobs)
        got = run(al)
        for r, g in zip(want, got):
            assert r["status"] == g["status"] and r["num_iterations"] == g["num_iterations"], knobs
            assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes(), knobs
            assert r["last"] == g["last"] and r["num_correspondences"] == g["num_correspondences"], knobs
            assert np.asarray(r["information"]).tobytes() == np.asarray(g["information"]).tobytes(), knobs


@pytest.mark.parametrize("prior_first", [False, True])
def test_prior_slice_next_to_projective_slices_sharing_one_association(oracle, product, prior_first):
    """... and beside the projective slices of an RGB-D tracker (a motion model next to point-to-plane + reprojection on one
    association): the control step of the z-buffer kernel's prologue linearises the prior between / behind the pack's sums."""
    kind = abi.SE3_QUAT_RIGHT
    r = syn.rgbd_pair(rows=120, cols=160, seed=3100)
    Z = syn.se3(np.array([0.03, 0.01, -0.02]), np.deg2rad([0.4, 0.9, -0.6])).astype(np.float32)

    def build(al):
        al.set_params(max_iterations=8, min_num_inliers=10, enable_inlier_only_runs=True)
        pi = al.add_slice(prior_config(kind, info=[50, 50, 50, 500, 500, 500], sets_guess=0)) if prior_first else None
        first = None
        for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
            c = abi.default_slice_config(kind)
            c.kind, c.finder, c.finder_max_distance = sk, abi.FINDER_PROJECTIVE, 0.05
            c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05 if sk == abi.SLICE_P2PLANE else 4.0
            for i, v in enumerate(r["K"].reshape(-1)):
                c.camera_matrix[i] = v
            c.image_rows, c.image_cols, c.depth_min, c.depth_max = r["rows"], r["cols"], r["depth_min"], r["depth_max"]
            si = al.add_slice(c)
            if first is None or isinstance(al, oracle.OracleAligner):
                al.set_fixed(si, r["fixed"], r["fixed_normals"])
                al.set_moving(si, r["moving"], r["moving_normals"])
                first = si if first is None else first
            else:
                al.share_clouds(si, first)
        if pi is None:
            pi = al.add_slice(prior_config(kind, info=[50, 50, 50, 500, 500, 500], sets_guess=0))
        al.set_prior_measurement(pi, Z)
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        al._cues = (first, first + 1)

    ref, fused, launched, unfused = _runs(oracle, product, kind, build, [{"fused_control": 1}, {"fused_control": 2}, {"fused_control": 0}])
    assert ref.status() == abi.SUCCESS
    for run in (fused, launched, unfused):
        assert_same_run(ref, run, slices=ref._cues)


@pytest.mark.parametrize("kind,slice_kind", [(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE), (abi.SE2_RIGHT, abi.SLICE_P2P)])
def test_prologue_inside_the_first_pass(oracle, product, kind, slice_kind, monkeypatch):
    """Round 6, late: from a handle's second compute() on, a single alignment with fused control steps has NO k_icp_init launch --
    the first pass derives the finder transform and the exponent from its arguments, wave 0 of its first workgroup writes state,
    records and tables on its way, and the slot sets were left zeroed by the previous compute()'s final step (k_icp_final_wave).
    One handle through everything that could leave something behind: a run the termination criterion stops early (sums left in a
    buffer its control steps never got to), an inlier-only run, a new initial guess, a moving cloud of another size, an EMPTY moving
    cloud (Fail: no statistics) and the compute() after it, a guess so far off that iteration 0 finds nothing, a new fixed cloud (the
    grid kernel's first pass), a prior slice that overrides the guess (cue slice at index 1), the launch path forced in between
    (SRRG2_AMD_TUNE bit 23) -- the oracle's bits after every compute(), and the path asserted where it must be taken."""
    monkeypatch.setenv("SRRG2_AMD_FUSED_GRID_MAX", "-1")
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=3000, sigma=0.01, seed=1234)
        gate, thr = 0.5, 0.002
        far = syn.se2(40.0, 0.0, 0.0).astype(np.float32)
        nudge = syn.se2(0.02, -0.01, 0.004).astype(np.float32)
        Z, info = syn.se2(0.03, -0.02, 0.01).astype(np.float32), [10.0, 10.0, 100.0]
    else:
        d = syn.cloud_pair_3d(n=15000, seed=2200, noise_sigma=0.01)
        gate, thr = 0.25, 0.0005
        far = syn.se3(np.array([30.0, 0.0, 0.0]), np.zeros(3)).astype(np.float32)
        nudge = syn.se3(np.array([0.02, -0.01, 0.015]), np.deg2rad([0.3, -0.2, 0.4])).astype(np.float32)
        Z, info = syn.se3(np.array([0.04, -0.02, 0.01]), np.deg2rad([0.5, -1.0, 1.5])).astype(np.float32), [10, 10, 10, 100, 100, 100]
    cfg = cue_config(kind, slice_kind, gate, abi.ROBUST_CAUCHY, thr)
    half = {k: (v[: len(v) // 2] if k.startswith("moving") else v) for k, v in d.items()}

    def script(al, with_prior):
        """yields after every compute(); the second element: must the prologue have ridden in the first pass?"""
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True)
        al.set_termination_criteria(abi.default_termination_params())
        if with_prior:
            pi = al.add_slice(prior_config(kind, info=info, sets_guess=1))
            al.set_prior_measurement(pi, Z)
        cue = setup_pair(al, d, cfg)
        al.compute()
        yield "first", False, cue
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        yield "second (the criterion stops it early)", True, cue
        al.set_moving_in_fixed(nudge)
        al.compute()
        yield "a new guess", True, cue
        al.set_moving(cue, half["moving"], half.get("moving_normals"))
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        yield "a smaller moving cloud", True, cue
        al.set_moving(cue, d["moving"][:0], None if d.get("moving_normals") is None else d["moving_normals"][:0])
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        yield "an empty moving cloud", False, cue
        al.set_moving(cue, d["moving"], d.get("moving_normals"))
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        yield "after the empty one", None, cue
        al.set_moving_in_fixed(far)
        al.compute()
        yield "nothing within the gate", None, cue
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        yield "after the failure", None, cue
        al.set_fixed(cue, d["fixed"], d.get("fixed_normals"))
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        yield "a new fixed cloud (grid kernel)", True, cue
        al.set_moving_in_fixed(nudge)
        al.compute()
        yield "and again on it", True, cue

    for with_prior in (False, True):
        for knobs in ({"search_lists": 2, "fused_control": 1}, {"search_lists": 0, "fused_control": 1},
                      {"search_lists": 2, "fused_control": 1, "strategy_mask": 1 << 23}):
            ref, got = oracle.OracleAligner(kind), product.MultiAligner(kind)
            got.set_tuning(**knobs)
            forced_launch = "strategy_mask" in knobs
            for (what, must, cue), _ in zip(script(ref, with_prior), script(got, with_prior)):
                if not with_prior:  # (a prior slice overrides every guess and keeps an empty cue slice's run alive: other statuses)
                    assert ref.status() == (abi.FAIL if what in ("an empty moving cloud", "nothing within the gate") else abi.SUCCESS), what
                try:
                    assert_same_run(ref, got, slices=(cue,))
                except AssertionError as e:
                    raise AssertionError("%s | %s | prior %s | path %d | statuses %d %d | stats %d %d" % (
                        what, knobs, with_prior, got.last_compute_path(), ref.status(), got.status(), len(ref.iteration_stats()),
                        len(got.iteration_stats()))) from e
                path = got.last_compute_path()
                if forced_launch:
                    assert not path & abi.PATH_PROLOGUE_IN_PASS, (what, knobs)
                elif must is not None:
                    assert bool(path & abi.PATH_PROLOGUE_IN_PASS) == must, (what, knobs, with_prior, path)


def test_prologue_inside_the_first_pass_of_a_small_batch(oracle, product):
    """... and for batches of up to eight alignments (the first pass reads the problems' rows of the pinned tables): two calls on one
    handle, the second one without a k_icp_init launch, ragged clouds, an empty one, as two halves on two streams and as one launch."""
    kind = abi.SE3_QUAT_RIGHT
    probs = syn.batch_3d(K=8, n=9000, seed=8820, shared_fixed_group=64, t_max=0.1, rpy_max_deg=2.0)
    movs = [p["moving"] for p in probs]
    nrms = [p["moving_normals"] for p in probs]
    movs[2], nrms[2] = movs[2][:0], nrms[2][:0]
    movs[5], nrms[5] = movs[5][:3000], nrms[5][:3000]
    guesses = [syn.identity(3)] * 8
    nudged = [syn.se3(np.array([0.01 * k, -0.01, 0.005]), np.deg2rad([0.2, -0.1 * k, 0.3])).astype(np.float32) for k in range(8)]
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.3, abi.ROBUST_CAUCHY, 0.05, 0.7)

    def run(al):
        al.set_params(max_iterations=7, min_num_inliers=100)
        al.add_slice(cfg)
        al.set_fixed(0, probs[0]["fixed"], probs[0]["fixed_normals"])
        first = al.compute_batch(movs, guesses, nrms)
        first = [dict(r) for r in first]
        path1 = al.last_compute_path()
        second = [dict(r) for r in al.compute_batch(movs, nudged, nrms)]
        path2 = al.last_compute_path()
        # (fewer alignments than the call before: the slot sets of the first three problems in the NEW layout were among those left
        # zeroed; then a single alignment on the same handle)
        third = [dict(r) for r in al.compute_batch(movs[4:7], nudged[4:7], nrms[4:7])]
        al.set_moving(0, movs[7], nrms[7])
        al.set_moving_in_fixed(nudged[7])
        al.compute()
        single = {"status": al.status(), "X": al.moving_in_fixed().tobytes(), "H": al.information().tobytes(),
                  "n": len(al.iteration_stats())}
        return first, second, path1, path2, third, single, al.last_compute_path()

    want = run(oracle.OracleAligner(kind))
    for knobs in (FUSED, dict(FUSED, batch_pipeline=0), dict(FUSED, strategy_mask=1 << 23)):
        al = product.MultiAligner(kind)
        al.set_tuning(**knobs)
        got = run(al)
        assert not got[2] & abi.PATH_PROLOGUE_IN_PASS
        assert bool(got[3] & abi.PATH_PROLOGUE_IN_PASS) == ("strategy_mask" not in knobs), knobs
        assert bool(got[6] & abi.PATH_PROLOGUE_IN_PASS) == ("strategy_mask" not in knobs), knobs
        assert want[5] == got[5], knobs
        for a, b in ((want[0], got[0]), (want[1], got[1]), (want[4], got[4])):
            for r, g in zip(a, b):
                assert r["status"] == g["status"] and r["num_iterations"] == g["num_iterations"], knobs
                assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes(), knobs
                assert r["last"] == g["last"] and r["num_correspondences"] == g["num_correspondences"], knobs
                assert np.asarray(r["information"]).tobytes() == np.asarray(g["information"]).tobytes(), knobs


@pytest.mark.parametrize("with_prior", [False, True])
def test_prologue_inside_the_z_buffer_pass_of_a_projective_pack(oracle, product, with_prior):
    """... and for projective slices that share one association (C3's shape): from the handle's second compute() on the z-buffer pass
    of the first iteration carries the prologue (k_proj_zbuf_fz_init), and the last step of every compute() is one wave
    (k_icp_final_wave_pack).  Three computes with different guesses, an inlier-only run, optionally a motion-model prior that overrides
    the guess: the oracle's bits, the path asserted."""
    kind = abi.SE3_QUAT_RIGHT
    r = syn.rgbd_pair(rows=120, cols=160, seed=3100)
    Z = syn.se3(np.array([0.03, 0.01, -0.02]), np.deg2rad([0.4, 0.9, -0.6])).astype(np.float32)
    guesses = [syn.identity(3), syn.se3(np.array([0.01, -0.01, 0.0]), np.deg2rad([0.2, 0.0, -0.1])).astype(np.float32), syn.identity(3)]

    def script(al):
        al.set_params(max_iterations=8, min_num_inliers=10, enable_inlier_only_runs=True)
        first = None
        for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
            c = abi.default_slice_config(kind)
            c.kind, c.finder, c.finder_max_distance = sk, abi.FINDER_PROJECTIVE, 0.05
            c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05 if sk == abi.SLICE_P2PLANE else 4.0
            for i, v in enumerate(r["K"].reshape(-1)):
                c.camera_matrix[i] = v
            c.image_rows, c.image_cols, c.depth_min, c.depth_max = r["rows"], r["cols"], r["depth_min"], r["depth_max"]
            si = al.add_slice(c)
            if first is None or isinstance(al, oracle.OracleAligner):
                al.set_fixed(si, r["fixed"], r["fixed_normals"])
                al.set_moving(si, r["moving"], r["moving_normals"])
                first = si if first is None else first
            else:
                al.share_clouds(si, first)
        if with_prior:
            pi = al.add_slice(prior_config(kind, info=[50, 50, 50, 500, 500, 500], sets_guess=1))
            al.set_prior_measurement(pi, Z)
        for g in guesses:
            al.set_moving_in_fixed(g)
            al.compute()
            yield (first, first + 1)

    for knobs in ({"fused_control": 1}, {"fused_control": 1, "strategy_mask": 1 << 23}, {"fused_control": 1, "strategy_mask": 1 << 25}):
        ref, got = oracle.OracleAligner(kind), product.MultiAligner(kind)
        got.set_tuning(**knobs)
        for k, (cues, _) in enumerate(zip(script(ref), script(got))):
            assert ref.status() == abi.SUCCESS
            assert_same_run(ref, got, slices=cues)
            assert got.information().tobytes() == ref.information().tobytes()
            path = got.last_compute_path()
            if "strategy_mask" not in knobs:
                assert path & abi.PATH_FINAL_WAVE, (k, path)
                assert bool(path & abi.PATH_PROLOGUE_IN_PASS) == (k > 0), (k, path)
            else:
                assert not path & abi.PATH_PROLOGUE_IN_PASS, (k, knobs, path)
