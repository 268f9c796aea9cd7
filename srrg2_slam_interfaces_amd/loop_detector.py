"""Loop-closure candidate drivers on top of the batched aligner: the breadth-first
local-map selector that proposes candidates, the brute-force detector and the relocalizer that align them.

Mirror of MultiLoopDetectorBruteForce_::compute()
(S/registration/loop_detector/multi_loop_detector_brute_force_impl.cpp:12-132): the fixed scene is set ONCE
(:63), every hint is one independent alignment (:64-79) -- here ONE compute_batch() call instead of the
sequential loop -- followed by the accept gates (:80-112) and the closure record (:120-131).
PARAM names and defaults: multi_loop_detector_brute_force.h:20-41.
"""
import numpy as np

from . import _abi as abi
from . import slices as sl


class ClosureHint:
    """LocalMapSelector_::ClosureHint: a candidate local map and the initial guess of moving-in-fixed."""

    def __init__(self, local_map_id, moving, moving_normals=None, initial_guess=None):
        self.local_map_id = local_map_id
        self.moving = moving
        self.moving_normals = moving_normals
        self.initial_guess = initial_guess


class MultiLoopDetectorBruteForce:
    def __init__(self, relocalize_aligner, relocalize_min_inliers=500, relocalize_max_chi_inliers=0.005,
                 relocalize_min_inliers_ratio=0.7):
        if relocalize_aligner is None:
            raise RuntimeError("MultiLoopDetectorBruteForce_::compute| no aligner")  # :52-54
        self.relocalize_aligner = relocalize_aligner
        self.relocalize_min_inliers = relocalize_min_inliers
        self.relocalize_max_chi_inliers = relocalize_max_chi_inliers
        self.relocalize_min_inliers_ratio = relocalize_min_inliers_ratio
        self.attempted_closures = []
        self.detected_closures = []
        self.drops = []

    def compute(self, source_local_map_id, fixed, fixed_normals, hints, pose_in_current=None):
        al = self.relocalize_aligner
        dim = al.dim
        pose_in_current = sl.identity(dim) if pose_in_current is None else np.asarray(pose_in_current, np.float32)
        self.attempted_closures = [h.local_map_id for h in hints if h.moving is not None]  # :71-75
        self.detected_closures, self.drops = [], []
        hints = [h for h in hints if h.moving is not None]
        if not hints:
            return self.detected_closures
        al.set_fixed(0, fixed, fixed_normals)  # aligner->setFixed(...) once, :63
        guesses = [sl.identity(dim) if h.initial_guess is None else h.initial_guess for h in hints]
        normals = [h.moving_normals for h in hints]
        results = al.compute_batch([h.moving for h in hints], guesses,
                                   normals if all(n is not None for n in normals) else None)
        for h, r in zip(hints, results):
            if r["status"] != abi.SUCCESS:  # :80-84
                self.drops.append((h.local_map_id, "ALIGNER DROP [code: %d]" % r["status"]))
                continue
            last = r["last"]
            # aligner->numCorrespondences() AFTER compute(), i.e. after _pruneCorrespondences (:89)
            num_correspondences = r.get("num_correspondences", last["num_correspondences"])
            num_inliers = last["num_inliers"]
            chi_inliers = np.float32(last["chi_inliers"]) / np.float32(num_inliers)  # :91
            if num_inliers < self.relocalize_min_inliers:  # :94-97
                self.drops.append((h.local_map_id, "NUM_INLIERS DROP"))
                continue
            if chi_inliers > np.float32(self.relocalize_max_chi_inliers):  # :99-103
                self.drops.append((h.local_map_id, "MAX_CHI_INLIERS DROP"))
                continue
            inlier_ratio = np.float32(num_inliers) / np.float32(num_correspondences)  # :105
            if inlier_ratio < np.float32(self.relocalize_min_inliers_ratio):  # :107-111
                self.drops.append((h.local_map_id, "MIN_INLIERS_RATIO DROP"))
                continue
            X = r["moving_in_fixed"]
            self.detected_closures.append({  # LoopClosure_ ctor arguments, :120-131
                "source": source_local_map_id,
                "target": h.local_map_id,
                "measurement": X,
                "information": np.eye(3 if dim == 2 else 6, dtype=np.float32),
                "pose_in_target": sl.compose(sl.inverse(X), pose_in_current),  # :120
                "chi_inliers": float(chi_inliers),
                "num_inliers": int(num_inliers),
                "num_correspondences": int(num_correspondences),
            })
        return self.detected_closures


class LocalMapSelectorBreadthFirst:
    """LocalMapSelectorBreadthFirst_::compute() (S/registration/local_map_selectors/
    local_map_selector_breadth_first_impl.cpp:12-101): a uniform-cost (hop count) visit of the pose graph from the
    current local map over its enabled factors, then one ClosureHint per local map whose origin lies within a range
    that grows with the graph distance.  PARAM names/defaults: local_map_selector_breadth_first.h:24-48.

    The visit is srrg2_solver's FactorGraphVisit with FactorGraphVisitCostUniform (:52-59; un-vendored): restated as a
    breadth-first search; local maps the visit does not reach are skipped."""

    def __init__(self, relocalize_range_scale=2, aggressive_relocalize_graph_distance=10,
                 aggressive_relocalize_graph_max_range=20, aggressive_relocalize_range_increase_per_edge=0.1,
                 max_local_map_distance=1.0):
        self.relocalize_range_scale = relocalize_range_scale
        self.aggressive_relocalize_graph_distance = aggressive_relocalize_graph_distance
        self.aggressive_relocalize_graph_max_range = aggressive_relocalize_graph_max_range
        self.aggressive_relocalize_range_increase_per_edge = aggressive_relocalize_range_increase_per_edge
        self.max_local_map_distance = max_local_map_distance
        self.hints = []
        self.costs = {}

    def compute(self, estimates, factors, source_id, robot_in_world):
        """estimates: {local map id: pose}; factors: iterable of (i, j, enabled); returns the hints as dicts
        {target, initial_guess (target in source), information, cost}."""
        if source_id not in estimates:
            raise RuntimeError("LocalMapSelectorBreadthFirst_::compute| _current_local_map is NULL")
        adj = {}
        for (i, j, enabled) in factors:
            if not enabled:
                continue
            adj.setdefault(i, []).append(j)
            adj.setdefault(j, []).append(i)
        cost, frontier = {source_id: 0}, [source_id]
        while frontier:
            nxt = []
            for v in frontier:
                for w in adj.get(v, ()):
                    if w not in cost:
                        cost[w] = cost[v] + 1
                        nxt.append(w)
            frontier = nxt
        self.costs = cost
        dim = 2 if np.asarray(robot_in_world).shape == (3, 3) else 3
        world_in_robot = sl.inverse(np.asarray(robot_in_world, np.float32))
        source_inv = sl.inverse(np.asarray(estimates[source_id], np.float32))
        self.hints = []
        for vid in sorted(estimates):  # graph->variables() is ordered by id
            if vid == source_id or vid not in cost:
                continue
            target = np.asarray(estimates[vid], np.float32)
            target_in_robot = sl.compose(world_in_robot, target)  # :69
            guess = sl.compose(source_inv, target)  # :70-71
            c = float(cost[vid])
            range_scale = self.relocalize_range_scale * c * self.aggressive_relocalize_range_increase_per_edge + 1  # :78-80
            range_scale = min(range_scale, float(self.aggressive_relocalize_graph_max_range))
            t = target_in_robot[:2, 2] if dim == 2 else target_in_robot[:, 3]
            if float(np.linalg.norm(t)) > self.max_local_map_distance * range_scale:  # :82-85
                continue
            if c > self.aggressive_relocalize_graph_distance:  # :88-90 aggressive relocalization
                guess = guess.copy()
                if dim == 2:
                    guess[:2, 2] = 0
                else:
                    guess[:, 3] = 0
            self.hints.append({"target": vid, "initial_guess": guess,
                               "information": np.eye(3 if dim == 2 else 6, dtype=np.float32), "cost": c})
        return self.hints


class MultiRelocalizer:
    """MultiRelocalizer_::compute() (S/registration/relocalization/multi_relocalizer_impl.cpp:12-145).  The closure
    candidates come from the loop detector (dicts as MultiLoopDetectorBruteForce.detected_closures plus, for the
    aligner branch, the target local map's cloud).  Without an aligner the best closure is chosen on the detector's
    statistics (:27-66); with one every candidate within max_translation is re-aligned against the current measurement
    -- here in ONE compute_batch() instead of the sequential loop -- gated like the detector (:104-121) and the lowest
    chi per inlier wins (:128-137).  PARAMs: multi_relocalizer.h:29-43, relocalizer.h:22."""

    def __init__(self, aligner=None, max_translation=3.0, relocalize_min_inliers=500, relocalize_max_chi_inliers=0.005,
                 relocalize_min_inliers_ratio=0.7):
        self.aligner = aligner
        self.max_translation = max_translation
        self.relocalize_min_inliers = relocalize_min_inliers
        self.relocalize_max_chi_inliers = relocalize_max_chi_inliers
        self.relocalize_min_inliers_ratio = relocalize_min_inliers_ratio
        self.relocalized_closure = None
        self.relocalization_map = None
        self.robot_in_local_map = None
        self.drops = []

    @staticmethod
    def _tnorm(T):
        T = np.asarray(T, np.float32)
        return float(np.linalg.norm(T[:2, 2] if T.shape == (3, 3) else T[:, 3]))

    def compute(self, closure_candidates, fixed=None, fixed_normals=None):
        self.relocalized_closure = self.relocalization_map = None
        self.drops = []
        dim = None
        for c in closure_candidates:
            dim = 2 if np.asarray(c["pose_in_target"]).shape == (3, 3) else 3
            break
        self.robot_in_local_map = sl.identity(dim or 3)
        near = []
        for c in closure_candidates:
            if self._tnorm(c["pose_in_target"]) > self.max_translation:  # :38-42, :84-88
                self.drops.append((c["target"], "MAX_TRANSITION DROP"))
                continue
            near.append(c)
        if self.aligner is None:
            best = None
            for c in near:
                if best is not None:
                    if c["chi_inliers"] > best["chi_inliers"]:  # :45-49
                        self.drops.append((c["target"], "HIGH_CHI_INLIERS DROP"))
                        continue
                    if c["num_correspondences"] < best["num_correspondences"]:  # :50-54
                        self.drops.append((c["target"], "LOW_MIN_CORRESPONDENCE DROP"))
                        continue
                best = c
            if best is not None:  # :62-66
                self.relocalized_closure = best
                self.relocalization_map = best["target"]
                self.robot_in_local_map = np.asarray(best["pose_in_target"], np.float32)
            return self.relocalization_map
        if not near:
            return None
        al = self.aligner
        al.set_fixed(0, fixed, fixed_normals)  # aligner->setFixed(&tracker->measurementContainer()), :78
        guesses = [sl.inverse(np.asarray(c["pose_in_target"], np.float32)) for c in near]  # :91
        normals = [c.get("moving_normals") for c in near]
        results = al.compute_batch([c["moving"] for c in near], guesses,
                                   normals if all(n is not None for n in normals) else None)
        best_chi_average = np.float32(np.finfo(np.float32).max)
        for c, r in zip(near, results):
            if r["status"] != abi.SUCCESS:  # :93-97
                self.drops.append((c["target"], "ALIGNER DROP [code: %d]" % r["status"]))
                continue
            last = r["last"]
            # numCorrespondences() after compute(), i.e. after pruning (multi_relocalizer_impl.cpp:101)
            num_inliers, num_correspondences = last["num_inliers"], r.get("num_correspondences", last["num_correspondences"])
            chi_inliers = np.float32(last["chi_inliers"]) / np.float32(num_inliers)  # :102
            if num_inliers < self.relocalize_min_inliers:  # :108-111
                self.drops.append((c["target"], "NUM_INLIERS DROP"))
                continue
            if chi_inliers > np.float32(self.relocalize_max_chi_inliers):  # :113-117
                self.drops.append((c["target"], "MAX_CHI_INLIERS DROP"))
                continue
            if np.float32(num_inliers) / np.float32(num_correspondences) < np.float32(self.relocalize_min_inliers_ratio):
                self.drops.append((c["target"], "MIN_INLIERS_RATIO DROP"))  # :119-125
                continue
            if chi_inliers < best_chi_average:  # :131-137
                self.relocalization_map = c["target"]
                self.robot_in_local_map = sl.inverse(r["moving_in_fixed"])
                best_chi_average = chi_inliers
                self.relocalized_closure = c
        return self.relocalization_map


class MultiLoopDetectorHBST:
    """The alignment half of MultiLoopDetectorHBST_ (S/registration/loop_detector/multi_loop_detector_hbst_impl.cpp):
    ``_computeAlignments`` (:257-377) -- per reference local map with enough descriptor matches, a one-variable
    Gauss-Newton solve with the matches kept locked, starting from the identity -- and the accept gates and closure
    record of ``_addLoopClosure`` (:379-447).  Here all candidates go through ONE compute_batch_correspondences().
    The descriptor tree that produces the matches (srrg_hbst) is out of scope: matches are an input.
    PARAMs: loop_detector.h (relocalize_min_inliers / max_chi_inliers / min_inliers_ratio)."""

    def __init__(self, relocalize_aligner, relocalize_min_inliers=500, relocalize_max_chi_inliers=0.005,
                 relocalize_min_inliers_ratio=0.7):
        if relocalize_aligner is None:
            raise RuntimeError("MultiLoopDetectorHBST::computeAlignments|ERROR: aligner not set")  # :264-268
        self.relocalize_aligner = relocalize_aligner
        self.relocalize_min_inliers = relocalize_min_inliers
        self.relocalize_max_chi_inliers = relocalize_max_chi_inliers
        self.relocalize_min_inliers_ratio = relocalize_min_inliers_ratio
        self.detected_closures = []
        self.drops = []

    def compute_alignments(self, query_id, fixed, fixed_normals, candidates, pose_in_query=None):
        """candidates: list of dicts {reference, moving, moving_normals or None, correspondences (fixed_idx = query
        point, moving_idx = reference point)}"""
        al = self.relocalize_aligner
        dim = al.dim
        pose_in_query = sl.identity(dim) if pose_in_query is None else np.asarray(pose_in_query, np.float32)
        self.detected_closures, self.drops = [], []
        todo = []
        for c in candidates:
            if len(c["correspondences"]) < self.relocalize_min_inliers:  # :309-314
                self.drops.append((c["reference"], "ALIGNER DROP [code: %d]" % abi.NOT_ENOUGH_CORRESPONDENCES))
                continue
            todo.append(c)
        if not todo:
            return self.detected_closures
        al.set_fixed(0, fixed, fixed_normals)
        normals = [c.get("moving_normals") for c in todo]
        results = al.compute_batch_correspondences(
            [c["moving"] for c in todo], [c["correspondences"] for c in todo], [sl.identity(dim)] * len(todo),  # :335
            normals if all(n is not None for n in normals) else None)
        for c, r in zip(todo, results):
            if r["status"] != abi.SUCCESS:  # :346-356
                self.drops.append((c["reference"], "ALIGNER DROP [code: %d]" % r["status"]))
                continue
            last = r["last"]
            num_inliers = last["num_inliers"]
            num_correspondences = num_inliers + last["num_outliers"] + last["num_suppressed"]  # :388-389
            chi_inliers = np.float32(last["chi_inliers"]) / np.float32(num_inliers)
            if num_inliers < self.relocalize_min_inliers:  # :394-398
                self.drops.append((c["reference"], "NUM_INLIERS DROP"))
                continue
            if chi_inliers > np.float32(self.relocalize_max_chi_inliers):  # :400-404
                self.drops.append((c["reference"], "MAX_CHI_INLIERS DROP"))
                continue
            if np.float32(num_inliers) / np.float32(num_correspondences) < np.float32(self.relocalize_min_inliers_ratio):
                self.drops.append((c["reference"], "MIN_INLIERS_RATIO DROP"))  # :406-413
                continue
            reference_in_query = r["moving_in_fixed"]
            D = 3 if dim == 2 else 6
            info = np.eye(D, dtype=np.float32)
            info[dim - 1, dim - 1] = 1e-3  # :429-430 (reduced weight along the last translation axis)
            self.detected_closures.append({
                "source": query_id, "target": c["reference"], "measurement": reference_in_query, "information": info,
                "pose_in_target": sl.compose(sl.inverse(reference_in_query), pose_in_query),  # :420
                "chi_inliers": float(chi_inliers), "num_inliers": int(num_inliers),
                "num_correspondences": int(num_correspondences), "correspondences": c["correspondences"]})
        return self.detected_closures
