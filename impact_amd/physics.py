"""Host-side mirror of the reference's rigid-body / constraint interface for the hot path
(engine/crates/impact_physics), on top of the C ABI (`ivx_world_*`):

  RigidBodyManager        src/rigid_body.rs:71-78, 373-395   (dynamic + kinematic bodies, momenta / configurations)
  ConstraintManager       src/constraint.rs:33-39, 193-309   (prepare_constraints, compute_and_apply_constrained_state)
  ConstraintSolverConfig  src/constraint/solver.rs:41-57, 374-384
  perform_physics_step    src/lib.rs:31-110

All arithmetic on body state runs in HIP kernels (impact_amd/csrc/physics.hip); nothing falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import sys
import weakref

import numpy as np

from . import capi
from .capi import CONTACT_DTYPE, KINEMATIC_BODY_DTYPE, RIGID_BODY_DTYPE, SOLVER_CONFIG_DTYPE, check, ptr
from .voxel import Context

_live_worlds: "weakref.WeakSet" = weakref.WeakSet()


class ConstraintSolverConfig:
    def __init__(self, n_iterations=8, old_impulse_weight=0.4, n_positional_correction_iterations=3, positional_correction_factor=0.2):
        self.n_iterations = n_iterations
        self.old_impulse_weight = old_impulse_weight
        self.n_positional_correction_iterations = n_positional_correction_iterations
        self.positional_correction_factor = positional_correction_factor

    def as_record(self):
        r = np.zeros(1, dtype=SOLVER_CONFIG_DTYPE)
        r[0] = (self.n_iterations, self.old_impulse_weight, self.n_positional_correction_iterations, self.positional_correction_factor)
        return r


SOLVER_STATIONARY = 255  # `ivx_world_set_solver_groups`: the chain-stationary solve wherever the schedule allows it


class PhysicsWorld:
    """RigidBodyManager + ConstraintManager living in HBM (`ivx_world`)."""

    def __init__(self, ctx: Context, config: ConstraintSolverConfig | None = None):
        self.ctx = ctx
        cfg = (config or ConstraintSolverConfig()).as_record()
        h = C.c_void_p()
        check(capi.lib().ivx_world_create(ctx.h, ptr(cfg), C.byref(h)))
        self.h = h
        self.n_dynamic = self.n_kinematic = 0
        self.n_prepared = 0
        _live_worlds.add(self)
        # worlds must go before the context at interpreter exit
        from . import voxel as _v

        _v._live_grids.add(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                capi.lib().ivx_world_destroy(self.h)
            self.h = None

    def __del__(self):
        if not sys.is_finalizing():
            try:
                self.close()
            except Exception:
                pass

    # ---- RigidBodyManager ------------------------------------------------------------------
    def set_bodies(self, dynamic, kinematic=None):
        dyn = np.ascontiguousarray(dynamic, dtype=RIGID_BODY_DTYPE)
        kin = np.zeros(0, dtype=KINEMATIC_BODY_DTYPE) if kinematic is None else np.ascontiguousarray(kinematic, dtype=KINEMATIC_BODY_DTYPE)
        check(capi.lib().ivx_world_set_bodies(self.h, ptr(dyn) if len(dyn) else None, len(dyn), ptr(kin) if len(kin) else None, len(kin)))
        self.n_dynamic, self.n_kinematic = len(dyn), len(kin)

    def bodies(self):
        dyn = np.zeros(self.n_dynamic, dtype=RIGID_BODY_DTYPE)
        kin = np.zeros(self.n_kinematic, dtype=KINEMATIC_BODY_DTYPE)
        check(capi.lib().ivx_world_get_bodies(self.h, ptr(dyn) if len(dyn) else None, ptr(kin) if len(kin) else None))
        return dyn, kin

    def advance_dynamic_rigid_body_momenta(self, step_duration: float):
        check(capi.lib().ivx_world_advance_momenta(self.h, step_duration))

    def advance_rigid_body_configurations(self, step_duration: float):
        check(capi.lib().ivx_world_advance_configurations(self.h, step_duration))

    # ---- ConstraintManager -------------------------------------------------------------------
    def prepare_constraints(self, contacts) -> int:
        """`prepare_constraints` over the explicit contact list of this step (constraint.rs:193-287)"""
        c = np.ascontiguousarray(contacts, dtype=CONTACT_DTYPE)
        n = C.c_size_t(0)
        check(capi.lib().ivx_world_set_contacts(self.h, ptr(c) if len(c) else None, len(c), C.byref(n)))
        self.n_prepared = int(n.value)
        return self.n_prepared

    def prepare_constraints_again(self):
        check(capi.lib().ivx_world_prepare(self.h))

    def compute_and_apply_constrained_state(self):
        check(capi.lib().ivx_world_solve(self.h))

    def contact_state(self):
        ids = np.zeros(max(self.n_prepared, 1), dtype=np.uint64)
        imp = np.zeros((max(self.n_prepared, 1), 3), dtype=np.float32)
        n = C.c_size_t(0)
        check(capi.lib().ivx_world_contact_state(self.h, ptr(ids), ptr(imp), len(ids), C.byref(n)))
        return ids[: n.value], imp[: n.value]

    def set_solver_groups(self, groups: int = 0):
        """workgroups the solve is spread over (`ivx_world_set_solver_groups`): 0 = automatic, 1 = one workgroup with the bodies in LDS,
        2..16 = tiles of the level schedule handed to that many workgroups, 255 (`SOLVER_STATIONARY`) = the chain-stationary solve"""
        check(capi.lib().ivx_world_set_solver_groups(self.h, int(groups)))

    def solver_info(self) -> dict:
        out = np.zeros(8, dtype=np.uint32)
        check(capi.lib().ivx_world_solver_info(self.h, ptr(out)))
        return {"workgroups": int(out[0]), "levels": [int(out[1]), int(out[2])], "widest_level": [int(out[3]), int(out[4])], "chains": int(out[5]),
                "contacts": int(out[6]), "kernel": ("one_workgroup", "tile_dataflow", "chain_stationary")[min(int(out[7]), 2)]}

    # ---- perform_physics_step ------------------------------------------------------------------
    def step(self, step_duration: float) -> np.ndarray:
        """one physics step over the resident bodies and contacts (`ivx_world_step`)"""
        out = np.zeros(1, dtype=capi.PHYSICS_RESULT_DTYPE)
        check(capi.lib().ivx_world_step(self.h, step_duration, ptr(out)))
        return out[0]

    def set_spherical_joints(self, body_pairs) -> None:
        """`ConstraintManager::add_spherical_joint` for all joints at once: (n, 2) body references (KINEMATIC_BODY flag for kinematic bodies).
        The reference's joint applies no impulse (constraint/spherical_joint.rs:62-88); its bodies become constrained bodies of every step."""
        bp = np.ascontiguousarray(np.asarray(body_pairs, dtype=np.uint32).reshape(-1, 2))
        check(capi.lib().ivx_world_set_spherical_joints(self.h, ptr(bp) if len(bp) else None, len(bp)))

    def step_enqueue(self, step_duration: float) -> None:
        """the same step, only enqueued on the context's stream (`ivx_world_step_enqueue`); read the bodies back with
        `get_bodies` after the next wait on the stream"""
        check(capi.lib().ivx_world_step_enqueue(self.h, step_duration))

    def perform_physics_step(self, contacts, step_duration: float) -> np.ndarray:
        self.prepare_constraints(contacts)
        return self.step(step_duration)


# ---- helpers mirroring the reference's constructors ------------------------------------------------
def uniform_sphere_body(radius: float, mass_density: float, position, velocity=(0.0, 0.0, 0.0)) -> np.ndarray:
    """`InertialProperties::of_uniform_sphere` + `DynamicRigidBody::new` with zero angular velocity
    (inertia.rs:155-168, rigid_body.rs:411-441); f32 arithmetic like the reference."""
    f32 = np.float32
    r = f32(radius)
    mass = f32(f32(f32(4.0 / 3.0) * f32(np.pi)) * f32(r * r * r)) * f32(mass_density)
    moi = f32(f32(f32(2.0 / 5.0) * mass) * f32(r * r))
    b = np.zeros((), dtype=RIGID_BODY_DTYPE)
    b["mass"] = mass
    b["inertia"] = (np.eye(3, dtype=np.float32) * moi).reshape(-1)
    b["inv_inertia"] = (np.eye(3, dtype=np.float32) * (f32(1.0) / moi)).reshape(-1)
    b["position"] = position
    b["orientation"] = (0.0, 0.0, 0.0, 1.0)
    b["momentum"] = np.asarray(velocity, dtype=np.float32) * mass
    return b
