"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import assert_same_run, cue_config, prior_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn

pytestmark = pytest.mark.gpu


# (deferred-search kernel for single alignments?, first iteration run by the converged-pass kernel, points per thread of
# that kernel, batches hand their failed certificates to the deferred-search kernel?, kept neighbours gathered from the
# fixed cloud?, search passes over the cell neighbour lists: 0 never / 1 batches / 2 every alignment, lanes per moving point
# of that kernel: None = automatic, 4 for single alignments and 1 for batches)
_PATHS = {
    "deferred-search kernel": ("0", None, None, None, None, "1", None),
    "searches finished in the step kernel": ("1000000000", None, None, None, None, "0", None),
    "converged-pass kernel from iteration 1, deferred searches, 4 points per thread": ("0", "1", "4", "1", "0", "0", None),
    "converged-pass kernel from iteration 1, wave-cooperative searches, gathered neighbours": ("1000000000", "1", "1", "0", "1", "1", "4"),
    "converged-pass kernel from iteration 2, 2 points per thread, streamed neighbours": ("1000000000", "2", "2", "0", "0", "0", None),
    "converged-pass kernel never": ("0", "1000000", None, None, None, "0", None),
    "cell neighbour lists for every search pass": ("0", None, None, None, None, "2", None),
    "cell neighbour lists with one lane per point, converged-pass kernel never, gathered neighbours": ("0", "1000000", None, None, "1", "2", "1"),
}


@pytest.fixture(autouse=True, params=list(_PATHS))
def search_path(request, monkeypatch):
    """Single alignments defer their open searches to k_icp_step_queue from 90 000 moving points on and finish them
    inside k_icp_step below (SRRG2_AMD_QUEUE_MIN); the converged-pass kernel k_icp_step_fast takes over from iteration 3
    (SRRG2_AMD_FAST_FROM) with 1 / 2 / 4 points per thread (SRRG2_AMD_FAST_PPT), its failed certificates go to the
    deferred-search kernel or are searched by their wave (SRRG2_AMD_FAST_QUEUE for batches), kept neighbours are streamed
    from per-point arrays or gathered from the fixed cloud (SRRG2_AMD_FAST_GATHER); the search passes walk the grid
    (k_icp_step / k_icp_step_tile) or the cell neighbour lists (k_icp_step_cnl, SRRG2_AMD_SEARCH_LISTS).  Every scenario of
    this module runs on all of these paths: they must give the same bits."""
    qmin, fast_from, ppt, fq, gather, lists, team = _PATHS[request.param]
    monkeypatch.setenv("SRRG2_AMD_QUEUE_MIN", qmin)
    monkeypatch.setenv("SRRG2_AMD_FAST_MIN", "0")  # (the converged-pass kernel also on this module's small clouds)
    monkeypatch.setenv("SRRG2_AMD_SMALL_MAX", "1024" if lists != "2" else "0")  # (lists: also the small clouds of this module)
    for name, val in (("SRRG2_AMD_FAST_FROM", fast_from), ("SRRG2_AMD_FAST_PPT", ppt), ("SRRG2_AMD_FAST_QUEUE", fq),
                      ("SRRG2_AMD_FAST_GATHER", gather), ("SRRG2_AMD_SEARCH_LISTS", lists), ("SRRG2_AMD_SEARCH_TEAM", team)):
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, val)


def _pair(oracle, product, kind):
    return oracle.OracleAligner(kind), product.MultiAligner(kind)


@pytest.mark.parametrize("kind", [abi.SE3_QUAT_RIGHT, abi.SE3_EULER_RIGHT])
@pytest.mark.parametrize("slice_kind", [abi.SLICE_P2PLANE, abi.SLICE_P2P])
def test_se3_icp_parity(oracle, product, kind, slice_kind):
    d = syn.cloud_pair_3d(n=20000, seed=2000)
    a_ref, a_gpu = _run_both(oracle, product, kind, d,
                             cue_config(kind, slice_kind, 0.25, abi.ROBUST_CAUCHY, 0.05, 0.8))
    assert a_ref.status() == abi.SUCCESS
    assert_same_run(a_ref, a_gpu)
    # converged close to the ground truth (different samplings of the same surface)
    tol = 5e-3 if slice_kind == abi.SLICE_P2PLANE else 3e-2  # point-to-point ICP converges slowly
    assert np.max(np.abs(a_gpu.moving_in_fixed() - d["X_gt"])) < tol


def _run_both(oracle, product, kind, data, cfg, params=None, term=None, guess=None, moving_normals=True):
    out = []
    for al in _pair(oracle, product, kind):
        if params:
            al.set_params(**params)
        if term is not None:
            al.set_termination_criteria(term)
        setup_pair(al, data, cfg, guess, moving_normals)
        al.compute()
        out.append(al)
    return out


def test_single_iteration_bitwise(oracle, product):
    """One finder pass + one Gauss-Newton step from the initial guess."""
    d = syn.cloud_pair_3d(n=30000, seed=2100)
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_SATURATED, 0.01)
    a_ref, a_gpu = _run_both(oracle, product, abi.SE3_QUAT_RIGHT, d, cfg, params=dict(max_iterations=1))
    assert_same_run(a_ref, a_gpu)


@pytest.mark.parametrize("slice_kind", [abi.SLICE_P2P, abi.SLICE_P2PLANE])
@pytest.mark.parametrize("sigma", [0.0, 0.01])
def test_se2_icp_parity_c1(oracle, product, slice_kind, sigma):
    """BASELINE config C1: SE(2) ICP on a 1k-beam scan pair."""
    d = syn.scan_pair_2d(beams=1000, sigma=sigma)
    cfg = cue_config(abi.SE2_RIGHT, slice_kind, 0.5, abi.ROBUST_CAUCHY if sigma else abi.ROBUST_NONE, 0.05)
    a_ref, a_gpu = _run_both(oracle, product, abi.SE2_RIGHT, d, cfg)
    assert a_ref.status() == abi.SUCCESS
    assert_same_run(a_ref, a_gpu)
    if sigma == 0.0:
        assert np.max(np.abs(a_gpu.moving_in_fixed() - d["X_gt"])) < 2e-2


def test_termination_inlier_runs_and_pruning(oracle, product):
    d = syn.cloud_pair_3d(n=15000, seed=2200, noise_sigma=0.01)
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.0005)
    params = dict(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True,
                  keep_only_inlier_correspondences=True)
    a_ref, a_gpu = _run_both(oracle, product, abi.SE3_QUAT_RIGHT, d, cfg, params=params,
                             term=abi.default_termination_params())
    assert a_ref.status() == abi.SUCCESS
    assert len(a_ref.iteration_stats()) < 24  # the criterion fired before 2 x 12 iterations
    assert_same_run(a_ref, a_gpu)
    assert np.all(a_gpu.factor_status(0) == abi.FACTOR_INLIER)


def test_prior_plus_cue_slices(oracle, product):
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=10000, seed=2300)
    Z = syn.se3(np.array([0.04, -0.02, 0.01]), np.deg2rad([0.5, -1.0, 1.5])).astype(np.float32)
    runs = []
    for al in _pair(oracle, product, kind):
        setup_pair(al, d, cue_config(kind, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05))
        pi = al.add_slice(prior_config(kind, info=[10, 10, 10, 100, 100, 100]))
        al.set_prior_measurement(pi, Z)
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1])


def test_prior_only_slices(oracle, product):
    for kind, Z in ((abi.SE3_QUAT_RIGHT, syn.se3(np.array([0.3, -0.2, 0.1]), np.deg2rad([20., -30., 45.]))),
                    (abi.SE2_RIGHT, syn.se2(0.4, -0.1, 0.7))):
        runs = []
        for al in _pair(oracle, product, kind):
            al.set_params(min_num_inliers=0)
            pi = al.add_slice(prior_config(kind))
            al.set_prior_measurement(pi, Z.astype(np.float32))
            al.compute()
            runs.append(al)
        assert runs[0].status() == abi.SUCCESS
        assert_same_run(runs[0], runs[1], slices=())


def test_edge_cases(oracle, product):
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=4000, seed=2400)
    # (a) nothing within the gate at the first iteration -> Fail
    far = dict(d)
    far["moving"] = d["moving"] + np.float32(50.0)
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.25)
    a_ref, a_gpu = _run_both(oracle, product, kind, far, cfg)
    assert a_ref.status() == abi.FAIL
    assert_same_run(a_ref, a_gpu)
    # (b) NaN / inf points on both sides are skipped
    bad = {k: v.copy() for k, v in d.items()}
    bad["moving"][::7, 1] = np.nan
    bad["fixed"][::11, 0] = np.inf
    a_ref, a_gpu = _run_both(oracle, product, kind, bad, cfg)
    assert_same_run(a_ref, a_gpu)
    # (c) empty moving cloud
    emp = dict(d)
    emp["moving"] = np.zeros((0, 3), np.float32)
    emp["moving_normals"] = np.zeros((0, 3), np.float32)
    a_ref, a_gpu = _run_both(oracle, product, kind, emp, cfg)
    assert a_ref.status() == abi.FAIL
    assert_same_run(a_ref, a_gpu)
    # (d) too few inliers
    a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg, params=dict(min_num_inliers=10 ** 6))
    assert a_ref.status() == abi.NOT_ENOUGH_INLIERS
    assert_same_run(a_ref, a_gpu)
    # (e) wide gate on a coarse cloud exercises the second search phase of the grid finder
    cfg2 = cue_config(kind, abi.SLICE_P2P, 1.5)
    sparse = {k: (v[::40] if v.ndim == 2 and v.shape[0] > 12 else v) for k, v in d.items()}
    a_ref, a_gpu = _run_both(oracle, product, kind, sparse, cfg2)
    assert_same_run(a_ref, a_gpu)


def test_batch_matches_sequential_and_oracle(oracle, product):
    kind = abi.SE3_QUAT_RIGHT
    probs = syn.batch_3d(K=6, n=6000, seed=4100, shared_fixed_group=8)
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.35, abi.ROBUST_CAUCHY, 0.05)
    res = []
    for al in _pair(oracle, product, kind):
        si = al.add_slice(cfg)
        al.set_fixed(si, probs[0]["fixed"], probs[0]["fixed_normals"])
        guesses = [syn.identity(3)] * len(probs)
        res.append(al.compute_batch([p["moving"] for p in probs], guesses, [p["moving_normals"] for p in probs]))
    for r, g in zip(*res):
        assert r["status"] == g["status"] == abi.SUCCESS
        assert r["num_iterations"] == g["num_iterations"]
        assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes()
        assert r["last"] == g["last"]
    for p, g in zip(probs, res[1]):
        assert np.max(np.abs(g["moving_in_fixed"] - p["X_gt"])) < 2e-2


def test_errors_are_loud(product):
    al = product.MultiAligner(abi.SE3_QUAT_RIGHT)
    with pytest.raises(RuntimeError):
        al.compute() if al.add_slice(cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2P, 0.1)) == 0 else None
    with pytest.raises(RuntimeError):
        al.set_fixed(5, np.zeros((1, 3), np.float32))


def test_c3_projective_and_reprojection_slices(oracle, product):
    """BASELINE config C3 (reduced resolution here, full 640x480 in test_c3_full_resolution): projective finder
    with point-to-plane and with pinhole-reprojection factors, alone and as a 2-slice MultiAligner."""
    from helpers import projective_config

    kind = abi.SE3_QUAT_RIGHT
    d = syn.rgbd_pair(rows=120, cols=160)
    for sk, rob in ((abi.SLICE_P2PLANE, abi.ROBUST_CAUCHY), (abi.SLICE_REPROJECTION, abi.ROBUST_NONE)):
        guess = syn.se3(np.array([0.01, 0.0, -0.01]), np.deg2rad([0.2, 0.4, -0.1])).astype(np.float32)
        cfg = projective_config(kind, sk, d, gate=0.05, robust=rob, thr=1e-4, normal_cos=0.5)
        a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg, guess=guess)
        assert a_ref.status() == abi.SUCCESS
        assert_same_run(a_ref, a_gpu)
    runs = []
    for al in _pair(oracle, product, kind):
        s0 = al.add_slice(projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05))
        s1 = al.add_slice(projective_config(kind, abi.SLICE_REPROJECTION, d, gate=0.05))
        for si in (s0, s1):
            al.set_fixed(si, d["fixed"], d["fixed_normals"])
            al.set_moving(si, d["moving"], d["moving_normals"])
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1], slices=(0, 1))


def test_c3_full_resolution(oracle, product):
    """640 x 480 depth image, projective + point-to-plane: parity and convergence to the ground truth."""
    from helpers import projective_config

    kind = abi.SE3_QUAT_RIGHT
    d = syn.rgbd_pair()
    cfg = projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05)
    a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg)
    assert a_ref.status() == abi.SUCCESS
    assert_same_run(a_ref, a_gpu)
    assert np.max(np.abs(a_gpu.moving_in_fixed() - d["X_gt"])) < 2e-4
    assert a_gpu.iteration_stats()[-1]["num_correspondences"] > 250000


def test_c3_full_resolution_two_slices(oracle, product):
    """BASELINE config C3 exactly as benchmarked (bench.py --workload c3): MultiAligner with 2 slices -- projective +
    point-to-plane, projective + reprojection -- on the 640 x 480 pair, both slices' passes in one launch pair
    (k_proj_zbuf_pack / k_icp_step_proj_pack).  Against the oracle, bit for bit, at the benchmark's own size."""
    from helpers import projective_config

    kind = abi.SE3_QUAT_RIGHT
    d = syn.rgbd_pair(seed=3000)
    runs = []
    for al in _pair(oracle, product, kind):
        for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
            si = al.add_slice(projective_config(kind, sk, d, gate=0.05))
            al.set_fixed(si, d["fixed"], d["fixed_normals"])
            al.set_moving(si, d["moving"], d["moving_normals"])
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1], slices=(0, 1))
    assert len(runs[1].iteration_stats()) == 10
    assert runs[1].iteration_stats()[-1]["num_correspondences"] > 400000  # both slices count


@pytest.mark.parametrize("variant", ["c3", "three slices, normal gate, robust kernels, inlier-only run", "different gates"])
def test_projective_slices_sharing_their_clouds(oracle, product, variant):
    """srrg2_aligner_share_clouds: two slices that read the same clouds (in the reference: the same fixed_slice_name /
    moving_slice_name, aligner_slice_processor_base_impl.cpp:27-50) and agree on their finder parameters share ONE
    association pass per iteration (k_icp_step_proj_fused).  The oracle runs every slice's own finder, as the reference
    does: same correspondences, statistics and estimate, bit for bit -- also with a third slice, a normal gate on one of
    them, different robustifiers and the inlier-only second run; slices with different gates share the clouds but not
    the association (the unfused launch pair)."""
    from helpers import projective_config

    kind = abi.SE3_QUAT_RIGHT
    rows, cols = (480, 640) if variant == "c3" else (120, 160)
    d = syn.rgbd_pair(rows=rows, cols=cols, seed=3000 if variant == "c3" else 3300)
    if variant == "c3":
        cfgs = [projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05), projective_config(kind, abi.SLICE_REPROJECTION, d, gate=0.05)]
        params = dict(max_iterations=10)
    elif variant == "different gates":
        cfgs = [projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05), projective_config(kind, abi.SLICE_REPROJECTION, d, gate=0.08)]
        params = dict(max_iterations=5)
    else:
        cfgs = [projective_config(kind, abi.SLICE_REPROJECTION, d, gate=0.05, robust=abi.ROBUST_CAUCHY, thr=2.0),
                projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05, robust=abi.ROBUST_SATURATED, thr=1e-4, normal_cos=0.9),
                projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05)]
        params = dict(max_iterations=4, min_num_inliers=10, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True)
    runs = []
    for al in _pair(oracle, product, kind):
        al.set_params(**params)
        for k, c in enumerate(cfgs):
            si = al.add_slice(c)
            if k == 0:
                al.set_fixed(si, d["fixed"], d["fixed_normals"])
                al.set_moving(si, d["moving"], d["moving_normals"])
            else:
                al.share_clouds(si, 0)
                with pytest.raises(Exception):  # (the clouds are the source's to set)
                    al.set_moving(si, d["moving"], d["moving_normals"])
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1], slices=tuple(range(len(cfgs))))
    # ... and the same bits as a product aligner whose slices each hold their own copy of the clouds
    own = product.MultiAligner(kind)
    own.set_params(**params)
    for c in cfgs:
        si = own.add_slice(c)
        own.set_fixed(si, d["fixed"], d["fixed_normals"])
        own.set_moving(si, d["moving"], d["moving_normals"])
    own.set_moving_in_fixed(syn.identity(3))
    own.compute()
    assert own.moving_in_fixed().tobytes() == runs[1].moving_in_fixed().tobytes()
    assert own.information().tobytes() == runs[1].information().tobytes()
    # a new moving cloud on the source serves the sharing slices too
    runs[1].set_moving(0, d["moving"][::2], d["moving_normals"][::2])
    runs[0].set_moving(0, d["moving"][::2], d["moving_normals"][::2])
    for al in runs:
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
    assert_same_run(runs[0], runs[1], slices=tuple(range(len(cfgs))))


def test_projective_correspondences_are_derived_on_demand(oracle, product):
    """The projective passes no longer store their correspondence records (9 bytes per point, slice and iteration): they are
    derived when asked for, from the z-buffer of the last executed pass (k_proj_zbuf_last / k_proj_records).  Asking twice
    gives the same records, asking for the second slice of a sharing pair first gives the same records, a run that stops
    on its termination criterion before max_iterations reports the records of the pass it stopped after (the oracle's), and
    once a cloud has been replaced the records of the previous run are gone (as for the nearest-neighbour slices)."""
    from helpers import projective_config

    kind = abi.SE3_QUAT_RIGHT
    d = syn.rgbd_pair(rows=120, cols=160, seed=3400)
    cfgs = [projective_config(kind, abi.SLICE_P2PLANE, d, gate=0.05, robust=abi.ROBUST_CAUCHY, thr=1e-3),
            projective_config(kind, abi.SLICE_REPROJECTION, d, gate=0.05)]
    runs = []
    for al in _pair(oracle, product, kind):
        al.set_params(max_iterations=12)
        tc = abi.TerminationParams()
        tc.window_size, tc.num_correspondences_range, tc.num_inliers_range, tc.num_outliers_range = 3, 100000, 100000, 1000000
        tc.chi_epsilon = 0.5
        al.set_termination_criteria(tc)
        for k, c in enumerate(cfgs):
            si = al.add_slice(c)
            if k == 0:
                al.set_fixed(si, d["fixed"], d["fixed_normals"])
                al.set_moving(si, d["moving"], d["moving_normals"])
            else:
                al.share_clouds(si, 0)
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        runs.append(al)
    assert len(runs[0].iteration_stats()) < 12  # (stopped by the criterion: the last pass is not the last possible one)
    second_first = [runs[1].correspondences(1), runs[1].correspondences(0)]
    assert_same_run(runs[0], runs[1], slices=(0, 1))
    again = [runs[1].correspondences(1), runs[1].correspondences(0)]
    for a, b in zip(second_first, again):
        assert a.tobytes() == b.tobytes()
    assert len(second_first[0]) > 1000
    runs[1].set_moving(0, d["moving"][::2], d["moving_normals"][::2])
    assert len(runs[1].correspondences(0)) == 0


@pytest.mark.parametrize("offset", [0.0, 900.0, -7000.0])
@pytest.mark.parametrize("cell", [0.0, 0.05, 0.4])
def test_ball_trimmed_search_is_exact(oracle, product, offset, cell):
    """The finder trims its scans to the ball of the best candidate known so far (previous iteration's neighbour,
    first-phase candidate).  Large coordinate offsets (coarse float32 spacing), cells much smaller / larger than the
    gate and a mixed-density cloud stress the conservative margins: the result must stay the exact (d2, index) minimum."""
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=12000, seed=2500, t=(0.12, -0.08, 0.05), rpy_deg=(2.0, -2.5, 3.0))
    d = {k: v.copy() for k, v in d.items()}
    d["fixed"] = np.ascontiguousarray(d["fixed"][::1])
    # thin out half of the fixed cloud: neighbours at several cell radii
    keep = np.ones(len(d["fixed"]), bool)
    keep[: len(keep) // 2][1::2] = False
    keep[: len(keep) // 4][::3] = False
    d["fixed"], d["fixed_normals"] = d["fixed"][keep], d["fixed_normals"][keep]
    # both clouds far from the origin: same relative geometry, coarse float32 spacing (4.9e-4 m at 7000 m)
    d["fixed"] = d["fixed"] + np.float32(offset)
    d["moving"] = d["moving"] + np.float32(offset)
    guess = syn.identity(3)
    cfg = cue_config(kind, abi.SLICE_P2P, 0.45, abi.ROBUST_CAUCHY, 0.05)
    cfg.finder_cell_size = cell
    a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg, params=dict(max_iterations=6), guess=guess)
    assert_same_run(a_ref, a_gpu)
    assert a_ref.iteration_stats()[0]["num_correspondences"] > 1000


@pytest.mark.parametrize("slice_kind", [abi.SLICE_P2PLANE, abi.SLICE_P2P])
def test_partial_overlap_parity(oracle, product, slice_kind):
    """40 % of the moving cloud has no neighbour within the gate (the fixed cloud is cropped): those points are
    certified 'no match' by an empty scan reaching beyond the gate and then skip their searches; points just outside
    the gate become (unmatched) nearest neighbours.  Results must stay those of the from-scratch search."""
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=30000, seed=2600, t=(0.08, -0.05, 0.03), rpy_deg=(1.5, -2.0, 2.5))
    d = {k: v.copy() for k, v in d.items()}
    keep = d["fixed"][:, 0] <= np.quantile(d["fixed"][:, 0], 0.6)
    d["fixed"], d["fixed_normals"] = d["fixed"][keep], d["fixed_normals"][keep]
    cfg = cue_config(kind, slice_kind, 0.25, abi.ROBUST_CAUCHY, 0.05)
    a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg,
                             params=dict(max_iterations=12, enable_inlier_only_runs=True))
    assert a_ref.status() == abi.SUCCESS
    assert_same_run(a_ref, a_gpu)
    n_corr = a_ref.iteration_stats()[-1]["num_correspondences"]
    assert 0.4 * 30000 < n_corr < 0.8 * 30000


@pytest.mark.parametrize("seed", list(range(64)))
def test_randomised_configurations(oracle, product, seed):
    """Random clouds, gates, cell sizes, overlaps, noise levels, robustifiers, iteration counts and flavours: the
    finder's shortcuts (ball-trimmed scans, exclusion radii, certificates for unmatched points, deferred searches)
    must never change a bit of the from-scratch result."""
    rng = np.random.default_rng(1000 + seed)
    dim3 = bool(rng.integers(0, 4))  # mostly 3-D
    gate = float(rng.choice([0.1, 0.25, 0.6, 1.5]))
    cell = float(rng.choice([0.0, 0.0, gate / 6, gate / 2, gate * 1.5]))
    iters = int(rng.integers(2, 15))
    rob = int(rng.choice([abi.ROBUST_NONE, abi.ROBUST_CAUCHY, abi.ROBUST_SATURATED, abi.ROBUST_CLAMP]))
    inlier_runs = bool(rng.integers(0, 2))
    if dim3:
        kind = int(rng.choice([abi.SE3_QUAT_RIGHT, abi.SE3_EULER_RIGHT]))
        n = int(rng.integers(500, 20000))
        t = rng.uniform(-0.5, 0.5, 3) * gate
        rpy = rng.uniform(-3.0, 3.0, 3)
        d = syn.cloud_pair_3d(n=n, seed=3000 + seed, t=tuple(t), rpy_deg=tuple(rpy),
                              noise_sigma=float(rng.choice([0.0, 0.01])))
        d = {k: v.copy() for k, v in d.items()}
        keep = d["fixed"][:, 1] <= np.quantile(d["fixed"][:, 1], float(rng.choice([1.0, 0.8, 0.5])))
        d["fixed"], d["fixed_normals"] = d["fixed"][keep], d["fixed_normals"][keep]
        if rng.integers(0, 2):  # thin out a part of the fixed cloud: neighbours at several cell radii
            m = np.ones(len(d["fixed"]), bool)
            m[: len(m) // 3][::2] = False
            d["fixed"], d["fixed_normals"] = d["fixed"][m], d["fixed_normals"][m]
    else:
        kind = abi.SE2_RIGHT
        d = syn.scan_pair_2d(beams=int(rng.integers(200, 3000)), sigma=float(rng.choice([0.0, 0.01])))
    slice_kind = int(rng.choice([abi.SLICE_P2PLANE, abi.SLICE_P2P]))
    cfg = cue_config(kind, slice_kind, gate, rob, float(rng.choice([0.01, 0.05, 0.5])),
                     float(rng.choice([-1.0, 0.5, 0.8])))
    cfg.finder_cell_size = cell
    a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg,
                             params=dict(max_iterations=iters, enable_inlier_only_runs=inlier_runs,
                                         keep_only_inlier_correspondences=bool(rng.integers(0, 2))),
                             moving_normals=bool(rng.integers(0, 4)))
    assert_same_run(a_ref, a_gpu)


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomised_batches(oracle, product, seed):
    """compute_batch (more than 4 alignments per launch: searches finish inside the step kernel, no deferred-search
    kernel) on random batches against the oracle."""
    rng = np.random.default_rng(5000 + seed)
    kind = int(rng.choice([abi.SE3_QUAT_RIGHT, abi.SE3_EULER_RIGHT]))
    K = int(rng.integers(5, 13))
    gate = float(rng.choice([0.2, 0.35, 0.8]))
    probs = syn.batch_3d(K=K, n=int(rng.integers(1000, 8000)), seed=6000 + 17 * seed, shared_fixed_group=64,
                         t_max=float(rng.uniform(0.05, 0.6)) * gate, rpy_max_deg=float(rng.uniform(0.5, 4.0)))
    fixed, fixed_n = probs[0]["fixed"].copy(), probs[0]["fixed_normals"].copy()
    keep = fixed[:, 0] <= np.quantile(fixed[:, 0], float(rng.choice([1.0, 0.7])))
    fixed, fixed_n = fixed[keep], fixed_n[keep]
    cfg = cue_config(kind, int(rng.choice([abi.SLICE_P2PLANE, abi.SLICE_P2P])), gate,
                     int(rng.choice([abi.ROBUST_NONE, abi.ROBUST_CAUCHY])), 0.05, float(rng.choice([-1.0, 0.7])))
    cfg.finder_cell_size = float(rng.choice([0.0, gate / 5, gate]))
    movs = [p["moving"][: int(rng.integers(len(p["moving"]) // 2, len(p["moving"]) + 1))] for p in probs]  # ragged
    nrms = [p["moving_normals"][: len(m)] for p, m in zip(probs, movs)]
    res = []
    for al in _pair(oracle, product, kind):
        al.set_params(max_iterations=int(4 + seed % 9), enable_inlier_only_runs=bool(seed % 2))
        si = al.add_slice(cfg)
        al.set_fixed(si, fixed, fixed_n)
        res.append(al.compute_batch(movs, [syn.identity(3)] * K, nrms))
    for r, g in zip(*res):
        assert r["status"] == g["status"]
        assert r["num_iterations"] == g["num_iterations"]
        assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes()
        assert r["last"] == g["last"]


@pytest.mark.parametrize("dim,K", [(3, 6), (3, 40), (2, 9)])
def test_batch_sort_edge_cases(oracle, product, dim, K):
    """Batches go through the one-workgroup-per-cloud Morton sort (5-bit keys up to 32 clouds, 4-bit beyond): clouds with
    non-finite points, an empty cloud, a single point, a degenerate (collinear) cloud, SE(2) batches."""
    kind = abi.SE3_QUAT_RIGHT if dim == 3 else abi.SE2_RIGHT
    rng = np.random.default_rng(8800 + K)
    if dim == 3:
        probs = syn.batch_3d(K=K, n=1500, seed=8800, shared_fixed_group=64, t_max=0.05, rpy_max_deg=1.0)
        fixed, fixed_n = probs[0]["fixed"], probs[0]["fixed_normals"]
        movs = [p["moving"].copy() for p in probs]
        nrms = [p["moving_normals"].copy() for p in probs]
        ident, gate, sk = syn.identity(3), 0.3, abi.SLICE_P2PLANE
    else:
        d = syn.scan_pair_2d(sigma=0.0)
        fixed, fixed_n = d["fixed"], d["fixed_normals"]
        movs = [np.ascontiguousarray(d["moving"][rng.permutation(len(d["moving"]))[: 600 + 20 * k]]) for k in range(K)]
        nrms = [np.zeros_like(m) for m in movs]
        ident, gate, sk = syn.identity(2), 0.5, abi.SLICE_P2P
    movs[1][::7] = np.nan                       # non-finite points (last Morton cell, never matched)
    movs[2] = movs[2][:0]; nrms[2] = nrms[2][:0]  # empty cloud
    movs[3] = movs[3][:1]; nrms[3] = nrms[3][:1]  # one point
    movs[4][:, 1:] = movs[4][0, 1:]             # collinear: zero extent on the other axes
    res = []
    for al in _pair(oracle, product, kind):
        al.set_params(max_iterations=6)
        si = al.add_slice(cue_config(kind, sk, gate, abi.ROBUST_CAUCHY, 0.05))
        al.set_fixed(si, fixed, fixed_n)
        res.append(al.compute_batch(movs, [ident] * K, nrms))
    for r, g in zip(*res):
        assert r["status"] == g["status"]
        assert r["num_iterations"] == g["num_iterations"]
        assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes()
        assert r["last"] == g["last"]


@pytest.mark.parametrize("dim", [3, 2])
def test_ties_and_duplicates(oracle, product, dim):
    """Fixed points on a lattice (every moving point at a cell centre is equidistant to 2^dim of them), exact duplicates
    in both clouds: ties resolve to the smallest fixed index in every search mode, and exclusion radii that equal the
    neighbour's distance never certify a skip."""
    kind = abi.SE3_QUAT_RIGHT if dim == 3 else abi.SE2_RIGHT
    g = np.arange(-6, 7, dtype=np.float32) * np.float32(0.25)
    mesh = np.stack(np.meshgrid(*([g] * dim), indexing="ij"), -1).reshape(-1, dim).astype(np.float32)
    fixed = np.concatenate([mesh, mesh[::3]])                       # duplicates with larger indices
    centres = (mesh + np.float32(0.125))[: len(mesh) // 2]           # equidistant to the lattice corners
    moving = np.concatenate([centres, centres[::5], mesh[1::7]]).astype(np.float32)  # duplicates, exact hits
    fn = np.zeros_like(fixed); fn[:, -1] = 1
    mn = np.zeros_like(moving); mn[:, -1] = 1
    d = dict(fixed=fixed, fixed_normals=fn, moving=moving, moving_normals=mn)
    for gate, cell in ((0.3, 0.0), (0.3, 0.1), (0.6, 0.25)):
        cfg = cue_config(kind, abi.SLICE_P2P, gate, abi.ROBUST_CAUCHY, 0.05)
        cfg.finder_cell_size = cell
        guess = (syn.se3(np.array([0.01, -0.02, 0.015]), np.deg2rad(np.array([0.5, -0.3, 0.8]))) if dim == 3
                 else syn.se2(0.01, -0.02, 0.01)).astype(np.float32)
        a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg, params=dict(max_iterations=8), guess=guess)
        assert_same_run(a_ref, a_gpu)
        assert a_ref.iteration_stats()[0]["num_correspondences"] > len(moving) // 2
    # exactly at the lattice: identity guess, every centre ties 2^dim ways
    cfg = cue_config(kind, abi.SLICE_P2P, 0.3)
    a_ref, a_gpu = _run_both(oracle, product, kind, d, cfg, params=dict(max_iterations=3))
    assert_same_run(a_ref, a_gpu)


def test_set_fixed_box_through_rows_and_polled_words(oracle, product):
    """Round 6, last: set_fixed's bounding box, count and norm come from the rows the blocks of k_ingest_bbox leave and its last block
    reduces (no atomics on shared words), through pinned words the host polls; SRRG2_AMD_TUNE bit 21 keeps the atomics, the copy and
    the wait.  One point, a handful, fewer points than a block, a cloud with NaNs, a million points (512 blocks): the same grid either
    way -- the oracle's bits where the oracle is quick enough to ask -- and twice on the same handle (the kernel re-arms its own
    ticket and clears the normals' norm for the next call)."""
    kind = abi.SE3_QUAT_RIGHT
    for n in (1, 7, 300, 70_000, 1_000_000):
        d = syn.cloud_pair_3d(n=max(n, 8), seed=5)
        f, fn = d["fixed"][:n].copy(), d["fixed_normals"][:n]
        m, mn = d["moving"][:max(n // 3, 1)], d["moving_normals"][:max(n // 3, 1)]
        if n >= 300:
            f[::97] = np.nan
        runs = []
        makers = [lambda: product.MultiAligner(kind), lambda: product.MultiAligner(kind)]
        if n <= 70_000:
            makers.append(lambda: oracle.OracleAligner(kind))
        for k, make in enumerate(makers):
            al = make()
            if k == 1:
                al.set_tuning(strategy_mask=1 << 21)
            al.set_params(max_iterations=4, min_num_inliers=0)
            si = al.add_slice(cue_config(kind, abi.SLICE_P2PLANE, 0.25))
            for rep in range(2):
                al.set_fixed(si, f, fn)
                al.set_moving(si, m, mn)
                al.set_moving_in_fixed(syn.identity(3))
                al.compute()
            runs.append((al.status(), al.moving_in_fixed().tobytes(), al.num_correspondences(), len(al.iteration_stats())))
        assert all(r == runs[0] for r in runs[1:]), (n, [r[0::2] for r in runs])
