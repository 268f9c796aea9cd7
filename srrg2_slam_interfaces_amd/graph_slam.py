"""Host-side mirror of the pose-graph lifecycle of ``MultiGraphSLAM_`` (S/system/multi_graph_slam_impl.cpp) on top of
the incremental ``srrg2_posegraph_*`` interface: the graph lives in device memory, local maps and closures are
appended, closures are validated, ``optimize()`` runs the GN/PCG solve only when a closure was accepted.

Only the graph bookkeeping is mirrored (makeNewMap :52-90, loopValidate :227-297, optimize :300-317); the tracker,
the local-map content and the loop detector are other components (out of scope / other rows).
"""
import numpy as np

REJECTED, ACCEPTED, PENDING = 0, 1, 2


class GraphSLAMLifecycle:
    def __init__(self, posegraph, default_information=None):
        self.graph = posegraph
        self.default_info = (np.eye(posegraph.D, dtype=np.float32) if default_information is None
                             else np.asarray(default_information, np.float32))  # _default_info
        self.current_local_map = None  # graph id
        self.num_valid_closures = 0
        self.closures = {}  # factor id -> (source, target)

    def make_new_map(self, robot_in_world, robot_in_local_map, info_scale=1.0):
        """makeNewMap(info_scale_), :52-90: a variable with the current robot pose; an odometry factor from the
        previous local map with measurement robot_in_local_map and information default_info * info_scale; the very
        first local map is Fixed (:86)."""
        previous = self.current_local_map
        vid = self.graph.add_variable(robot_in_world, fixed=previous is None)
        if previous is not None:
            self.graph.add_factor(previous, vid, robot_in_local_map, self.default_info * np.float32(info_scale), enabled=True)
        self.current_local_map = vid
        return vid

    def loop_validate(self, detected_closures, validator=None):
        """loopValidate(), :227-297.  detected_closures: iterable of (source_id, target_id, Z, information or None).
        Closures enter the graph disabled (:238-241).  Without a validator all of them are accepted (:245-251);
        with one (a callable: list of factor ids -> list of REJECTED / ACCEPTED / PENDING) rejected closures are
        removed from the graph (:279-281) and accepted ones enabled (:283-286).  Returns the accepted factor ids."""
        self.num_valid_closures = 0
        ids = []
        for (i, j, Z, info) in detected_closures:
            fid = self.graph.add_factor(i, j, Z, info, enabled=False)
            self.closures[fid] = (i, j)
            ids.append(fid)
        if not ids:
            return []
        if validator is None:
            for fid in ids:
                self.graph.set_factor_enabled(fid, True)
            self.num_valid_closures = len(ids)
            return ids
        accepted = []
        for fid, verdict in zip(ids, validator(ids)):
            if verdict == REJECTED:
                self.graph.remove_factor(fid)
                del self.closures[fid]
            elif verdict == ACCEPTED:
                self.num_valid_closures += 1
                self.graph.set_factor_enabled(fid, True)
                accepted.append(fid)
        return accepted

    def optimize(self, params=None):
        """optimize(), :300-317: nothing to do without a valid closure; else bindFactors + global_solver->compute()."""
        if not self.num_valid_closures:
            return []
        return self.graph.solve(params)
