"""mkz_mpc_path_follower_amd -- MI355X-native batched kinematic-bicycle MPC solver.

Drop-in for ONE path of govvijaycal/mkz_mpc_path_follower: the per-step nonlinear MPC solve of
scripts/mpc_utils/MKZMPCPathFollower.jl as driven by scripts/mpc_cmd_pub.jl.  Importing the
package does not touch the GPU; constructing a solver does, and fails loudly without one.
"""
__all__ = ["BatchMPC", "KinematicMPC", "synthetic"]


def __getattr__(name):
    if name == "BatchMPC":
        from .solver import BatchMPC
        return BatchMPC
    if name == "KinematicMPC":
        from .kinematic_mpc import KinematicMPC
        return KinematicMPC
    if name == "synthetic":
        from . import synthetic
        return synthetic
    raise AttributeError(name)
