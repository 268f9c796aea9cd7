"""Brute-force loop-closure candidate driver on top of the batched aligner.

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
            num_correspondences = last["num_correspondences"]
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
