#!/usr/bin/env python3
"""Throughput benchmark of the voxel hot path (BASELINE.json metric:
"voxels stepped/sec + remesh tris/sec, 512^3 grid, 1/2/4/8 MI355X").

A step = one pass of the per-frame voxel path over one SDF-defined grid that is already resident in
HBM: SDF sample -> derived state (flags, chunk state, occupied ranges) -> connected regions ->
Surface Nets remesh -> mass/inertia reduction -> rigid-body step of the object's own body (momenta,
constraint solve over its — empty — contact list, configuration). N = 1: the config-2 asteroid scaled
x2.05 (502^3 grid -> 32^3 chunks = 512^3 stored voxels). N > 1 (one process per GPU, launched by
torch.distributed.run): weak scaling — N such asteroids in a row joined by a thin bar, one per x-slab of
32 chunk planes, one-voxel face halos and boundary chunk state exchanged over RCCL (torch.distributed
"nccl"), cross-rank region equivalences and the 10 mass moments in one small all-gather.

The rigid-body solver's own workload (BASELINE config 4: 4096 bodies, 46 080 contacts) does not scale
with voxels; it is timed separately on rank 0 and reported under "pile" in the same JSON line.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the byte counts behind `roofline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


STAGE_KERNELS = {  # which kernels make up a timed stage (names as rocprofv3 reports them)
    "sdf_sample": ["k_sdf_super", "k_sdf_prepass", "k_sdf_eval"],
    "derive": ["k_chunk_pre", "k_derive"],  # k_derive also labels the chunk-local regions and leaves the chunk moments (fused sweep)
    "occupied": ["k_occupied_reduce"],
    "ccl_local": ["k_ccl_local_exact"],  # what is left of the stage after the fusion
    "ccl_merge": ["k_ccl_merge_columns", "k_ccl_merge_multi"],
    "ccl_resolve": ["k_ccl_flatten", "k_scan_groups", "k_ccl_assign"],
    "sn_count": ["k_sn_count"],
    "sn_scan": ["k_sn_scan"],
    "sn_emit": ["k_sn_emit"],
    "inertia": ["k_inertia_sum", "k_inertia_final"],
}


def stage_bytes(n_voxels, n_chunks, exposed_chunks, n_vertices, n_indices):
    """Algorithmic HBM bytes per launch of each timed stage: SURVEY.md §8(d)'s per-voxel figures x the stored voxels of the
    launch for the plane sweeps (the figures are per voxel of the GRID, whether or not a chunk's planes are materialised: see
    DESIGN.md §4 on compact planes — measured traffic is therefore far below these numbers for a solid body); the mesher is
    charged for the padded tiles of the chunks it meshes plus its output. The derive stage runs the chunk-local region
    labelling and the chunk moments in the same sweep (R sdf + type, W flags + label = 4 B/voxel, SURVEY's fused-sweep
    accounting), so the `ccl_local` and `inertia` stages are left with their list / reduction kernels only."""
    return {
        "sdf_sample": 2.0 * n_voxels,                        # W sdf + type
        "derive": 4.0 * n_voxels,                            # R sdf + type, W flags + label (fused sweep)
        "occupied": 4.0 * n_chunks,                          # R one packed box per chunk
        "ccl_local": 0.0,                                    # fused into derive; k_ccl_local_exact walks a (usually empty) list
        "ccl_merge": 9.0 * 3 * n_chunks,                     # chunk records + touch bytes of a chunk column and its two neighbours
        "ccl_resolve": 16.0 * n_chunks,                      # (chunk, region) table entries in use, a few per chunk
        "sn_count": 1.0 * 5832 / 8 * exposed_chunks,         # 18^3 sign bits per exposed chunk
        "sn_scan": 36.0 * n_chunks,                          # counts in, offsets/ranks/records out
        "sn_emit": 2.0 * 5832 * exposed_chunks + 40.0 * n_vertices + 12.0 * n_indices,  # tile + (pos,nrm,vmat) + (idx u32, imat 8B)
        "inertia": 88.0 * n_chunks,                          # chunk records + per-chunk moment slots
    }


def measured_traffic(stage):
    """HBM bytes per launch of the stage's kernels from the committed PMC passes (profiles/round1/pmc_traffic.json, produced by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE rocprofv3 runs of this same workload), or None."""
    path = os.path.join(ROOT, "profiles", "round1", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    total, found = 0.0, False
    for k in STAGE_KERNELS.get(stage, []):
        for name in d:  # (template kernels are reported as "void k_name<args>")
            if name == k or name.startswith("void " + k + "<"):
                total += d[name]["hbm_bytes"]
                found = True
    return total if found else None


def cpu_baseline(scale):
    """The oracle (single thread, -O2) over the same workload, timed on this box's host cores."""
    import oracle_lib as ol
    from impact_amd import scenes

    graph = scenes.asteroid_scene(scale)
    t0 = time.perf_counter()
    o = ol.OracleObject.from_sdf(graph, 1.0, 0)
    o.update_occupied_voxel_ranges()
    o.compute_all_derived_state()
    t1 = time.perf_counter()
    m = o.mesh()
    t2 = time.perf_counter()
    o.inertia()
    t3 = time.perf_counter()
    cc = o.chunk_counts
    nvox = cc[0] * cc[1] * cc[2] * 4096
    # contact generation between two voxel bodies: probes of this body, then mutual contacts with a half-size copy
    global _CPU_COLLIDE
    t6 = time.perf_counter()
    pa = o.collision_probes(m)
    t7 = time.perf_counter()
    ob = ol.OracleObject.from_sdf(scenes.asteroid_scene(0.5 * scale), 1.0, 0)
    ob.update_occupied_voxel_ranges()
    ob.compute_all_derived_state()
    pb = ob.collision_probes(ob.mesh())
    ca, cb = o.center_of_mass(), ob.center_of_mass()
    qa, ta, qb, tb = collide_poses(ca, cb, scale)
    t8 = time.perf_counter()
    wi = o.mutual_contacts(pa, ca, qa, ta, ob, pb, cb, qb, tb, cap=1 << 20)[0]
    t9 = time.perf_counter()
    _CPU_COLLIDE = {"probes_ms": 1e3 * (t7 - t6), "mutual_ms": 1e3 * (t9 - t8), "probes": int(len(pa[0])), "contacts": int(len(wi)), "cores": 1,
                    "kind": "port"}
    del ob
    # the edit op on the same object: one absorbing sphere at the surface (EDIT_* below), derived state refreshed
    c = np.array([0.5 * (a + b_) for a, b_ in o.info()["occupied_voxel_ranges"]], dtype=np.float32) + EDIT_OFFSET * np.float32(scale)
    t4 = time.perf_counter()
    er = o.absorb_sphere(c, EDIT_RADIUS * scale + 2.0, EDIT_RADIUS * scale)
    t5 = time.perf_counter()
    global _CPU_EDIT
    _CPU_EDIT = {"ms": 1e3 * (t5 - t4), "emptied_voxels": int(er["emptied_by_type"].sum()), "cores": 1, "kind": "port"}
    return {
        "value": nvox / (t3 - t0),
        "unit": "voxels/s",
        "cores": 1,
        "kind": "port",
        "sample": f"full N=1 workload once ({cc[0] * 16}^3 stored voxels): generate+derive {t1 - t0:.2f}s, remesh {t2 - t1:.2f}s "
                  f"({m.indices.size // 3 / (t2 - t1):.3g} tris/s), inertia {t3 - t2:.2f}s; single thread",
    }


def collide_poses(com_a, com_b, scale):
    """world -> object transforms for the collide benchmark: body A with its centre of mass at the world origin, the half-size body
    B rotated and pushed ~4 % of its radius into A's side"""
    qa = np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)
    ta = com_a.astype(np.float32)
    axis = np.array([0.3, 0.1, 1.0]) / np.linalg.norm([0.3, 0.1, 1.0])
    qb = np.array([*(axis * np.sin(0.35)), np.cos(0.35)], dtype=np.float32)
    d = np.array([0.6, 0.64, 0.48])
    centre_b = d / np.linalg.norm(d) * (96.0 + 48.0 - 4.0) * scale
    x, y, z, w = [float(v) for v in qb]
    b = np.array([x, y, z])
    rot = centre_b * (w * w - b @ b) + b * (2 * (centre_b @ b)) + np.cross(b, centre_b) * (2 * w)
    tb = (com_b.astype(np.float64) - rot).astype(np.float32)
    return qa, ta, qb, tb


def collide_benchmark(ctx, scale, reps=5):
    """SURVEY §8f item 1 on the N=1 workload: collision probes of the meshed body (`ivx_collision_probes_recompute`), then the
    contacts between it and a half-size copy pushed into its side (`ivx_mutual_voxel_object_contacts`, results on the host)."""
    from impact_amd import capi, scenes
    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject, VoxelObjectMesh

    objs = []
    for sc in (scale, 0.5 * scale):
        gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(sc), 0)
        obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
        obj.set_sdf_program(gen)
        obj.set_densities(np.ones(256, dtype=np.float32))
        r = obj.step(capi.STAGE_ALL)
        m32 = np.asarray(r["moments"]["m32"], dtype=np.float32).reshape(-1)
        com = (m32[1:4] * (np.float32(1.0) / m32[0])).astype(np.float32)  # derive_center_of_mass (object/inertia.rs:167-169)
        VoxelObjectMesh.create(obj)
        objs.append((obj, com))
    (a, ca), (b, cb) = objs
    qa, ta, qb, tb = collide_poses(ca, cb, scale)
    t_p, t_m, n_probes, n_contacts = [], [], 0, 0
    for _ in range(reps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        n_probes = a.collision_probes_recompute()
        t1 = time.perf_counter()
        b.collision_probes_recompute()
        ctx.synchronize()
        t2 = time.perf_counter()
        n_contacts = len(a.mutual_contacts(qa, ta, ca, b, qb, tb, cb, 1, 2, 0, 1, capacity=1 << 20))
        t3 = time.perf_counter()
        t_p.append(t1 - t0)
        t_m.append(t3 - t2)
    a.close()
    b.close()
    out = {"workload": "probes of the N=1 body; contacts between it and a half-size copy pushed into its side",
           "probes_ms": round(1e3 * float(np.mean(t_p[1:])), 4), "mutual_ms": round(1e3 * float(np.mean(t_m[1:])), 4), "probes": n_probes,
           "contacts": n_contacts}
    if _CPU_COLLIDE is not None:
        out["cpu_baseline"] = dict(_CPU_COLLIDE)
        out["cpu_baseline"]["parity"] = ("same probe and contact counts" if (_CPU_COLLIDE["probes"], _CPU_COLLIDE["contacts"]) == (n_probes, n_contacts)
                                         else "MISMATCH")
    return out


_CPU_COLLIDE = None
EDIT_OFFSET = np.array([110.0, 6.0, -4.0], dtype=np.float32)  # from the centre of the body, at scale 1: inside the tip of the +x bump
EDIT_RADIUS = 15.0
_CPU_EDIT = None


def edit_benchmark(ctx, scale, reps=5):
    """SURVEY §8f item 2 on the N=1 workload: an absorbing sphere bites into the asteroid (`ivx_absorb_sphere`: the edit kernel,
    the derived-state + region refresh of the whole object, results back on the host), then the full remesh the bite invalidates.
    Each repetition starts from the freshly generated body."""
    from impact_amd import capi, scenes
    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject, VoxelObjectMesh

    gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(scale), 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    mesh = VoxelObjectMesh(obj)
    t_edit, t_remesh, t_sync, emptied, touched, invalidated = [], [], [], 0, 0, 0
    for _ in range(reps + 1):
        obj.step(capi.STAGE_ALL)
        c = np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + EDIT_OFFSET * np.float32(scale)
        mesh.sync_with_voxel_object(np.zeros(obj.n_chunks, dtype=np.uint8))  # (the submesh bookkeeping of the fresh mesh, once per full remesh)
        ctx.synchronize()
        t0 = time.perf_counter()
        r = obj.absorb_sphere(c, EDIT_RADIUS * scale + 2.0, EDIT_RADIUS * scale, want_invalidated=True)
        t1 = time.perf_counter()
        mesh.sync_with_voxel_object(r["invalidated"])  # the incremental remesh of the invalidated chunks (mesh.rs:355-456) ...
        t2 = time.perf_counter()
        obj.step(capi.STAGE_REMESH)  # ... and the full one, for comparison
        t3 = time.perf_counter()
        t_edit.append(t1 - t0)
        t_sync.append(t2 - t1)
        t_remesh.append(t3 - t2)
        emptied, touched, invalidated = r["emptied_voxels"], r["touched_chunks"], int(r["invalidated"].sum())
    obj.close()
    out = {"workload": f"absorbing sphere r={EDIT_RADIUS * scale:.1f} voxels at the surface of the N=1 body",
           "edit_ms": round(1e3 * float(np.mean(t_edit[1:])), 4), "remesh_after_ms": round(1e3 * float(np.mean(t_remesh[1:])), 4),
           "sync_after_ms": round(1e3 * float(np.mean(t_sync[1:])), 4), "emptied_voxels": emptied, "touched_chunks": touched,
           "invalidated_chunks": invalidated}
    if _CPU_EDIT is not None:
        out["cpu_baseline"] = dict(_CPU_EDIT)
        out["cpu_baseline"]["parity"] = "same emptied voxel count" if _CPU_EDIT["emptied_voxels"] == emptied else "MISMATCH"
    return out


def pile_benchmark(ctx, with_cpu, steps=10):
    """BASELINE config 4 on one GPU: 16^3 spheres, 46 080 contacts resident in HBM, 8 velocity + 3 positional
    sweeps per step in the reference's exact order (dependency-level schedule)."""
    from impact_amd import capi, scenes
    from impact_amd.physics import PhysicsWorld

    bodies, contacts = scenes.sphere_pile_scene(16)
    w = PhysicsWorld(ctx)
    w.set_bodies(bodies)
    t0 = time.perf_counter()
    w.prepare_constraints(contacts)
    host_ms = 1e3 * (time.perf_counter() - t0)
    w.step(0.005)
    t0 = time.perf_counter()
    w.prepare_constraints(contacts)  # the per-frame case: the same contact ids as last frame (cache hits, buffers sized)
    host_warm_ms = 1e3 * (time.perf_counter() - t0)
    for _ in range(2):
        w.step(0.005)
    acc = np.zeros(5)
    t0 = time.perf_counter()
    for _ in range(steps):
        acc += w.step(0.005)["stage_ms"]
    wall = (time.perf_counter() - t0) / steps
    r = w.step(0.005)
    sweeps = 1 + 8 + 3
    out = {
        "workload": "16^3 lattice of unit-density spheres r=0.5 at spacing 0.95: 4096 bodies, 46080 contacts, dt 0.005, 8+3 sweeps + warm start",
        "ms_per_step": 1e3 * wall,
        "contact_sweeps_per_s": len(contacts) * sweeps / wall,
        "levels": [int(r["n_levels"][0]), int(r["n_levels"][1])],
        "stage_ms": {k: round(float(v) / steps, 4) for k, v in zip(capi.PHYSICS_STAGE_NAMES, acc)},
        "set_contacts_next_frame_host_ms": round(host_warm_ms, 3), "set_contacts_host_ms": round(host_ms, 3),
    }
    if with_cpu:
        import oracle_lib as ol

        o = ol.OraclePhysics(bodies, config=(8, 0.4, 3, 0.2))
        o.step(contacts, 0.005)
        t0 = time.perf_counter()
        for _ in range(5):
            o.step(contacts, 0.005)
        cpu = (time.perf_counter() - t0) / 5
        out["cpu_baseline"] = {"value": len(contacts) * sweeps / cpu, "unit": "contact sweeps/s", "cores": 1, "kind": "port",
                               "sample": f"same pile, 5 steps, {1e3 * cpu:.1f} ms/step, single thread"}
    w.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=2.05, help="asteroid scale (2.05 -> 512^3 stored grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pile", action="store_true", help="skip the separate rigid-body pile timing (config 4)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch

    from impact_amd import capi, scenes
    from impact_amd.voxel import Context, SDFVoxelGenerator, VoxelObject

    # one rank per GPU; on a box with fewer GPUs than ranks (single-GPU protocol check with the gloo
    # backend) ranks share devices
    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    ctx = Context(device)
    dist = None
    # IVX_BENCH_FORCE_SLABS=1: run the slab protocol (and its RCCL calls) even at world size 1 — a single-GPU check of the
    # N>1 code path under torch.distributed.run; never the default
    slabs = world > 1 or os.environ.get("IVX_BENCH_FORCE_SLABS") == "1"
    if slabs:
        import torch.distributed as dist_mod

        dist = dist_mod
        backend = os.environ.get("IVX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            except Exception as e:  # RCCL unusable on this box: keep the run alive through host staging and say so
                print(f"[bench] rank {rank}: RCCL init failed ({e}); falling back to gloo", file=sys.stderr)
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)

    dens = np.ones(256, dtype=np.float32)
    # the voxel object's own rigid body (setup_dynamic_rigid_body_for_voxel_object, impact_voxel/src/setup.rs:581-616):
    # integrated every step; its contact list is empty in this workload
    from impact_amd.physics import PhysicsWorld, uniform_sphere_body

    body_world = PhysicsWorld(ctx)
    body_world.set_bodies(np.array([uniform_sphere_body(100.0, 1.0, (0.0, 0.0, 0.0), (0.1, 0.0, 0.0))]))
    body_world.prepare_constraints(np.zeros(0, dtype=capi.CONTACT_DTYPE))
    if not slabs:
        gen = SDFVoxelGenerator(1.0, scenes.asteroid_scene(args.scale), 0)
        cc = gen.chunk_counts()
        obj = VoxelObject(ctx, cc, 1.0)
        obj.set_sdf_program(gen)
        obj.set_densities(dens)

        def step():
            # the voxel stages and the object's rigid-body step are enqueued back to back; one wait covers both
            obj.step_enqueue(capi.STAGE_ALL)
            body_world.step_enqueue(0.005)
            return obj.step_collect()

        workload = f"config-2 SDF asteroid x{args.scale} -> {gen.grid_shape()[0]}^3 grid = {cc[0] * 16}^3 stored voxels ({obj.n_chunks} chunks)"
        parallelism = "single GPU"
    else:
        from impact_amd.distributed import SlabStepper, TorchComm

        # weak scaling the way BASELINE.json's config 5 states it: the config-2 asteroid with all lengths scaled so that every
        # rank keeps the 512^3 workload's voxel count (N = 8: scale x2 again -> the 1024^3 grid in 8 slabs of 128 planes)
        mg_scale = args.scale * world ** (1.0 / 3.0)
        stepper = SlabStepper(ctx, scenes.asteroid_scene(mg_scale), dens, rank, world, torch)
        comm = TorchComm(dist, torch, rank, world)
        obj = stepper.obj

        class _Res(dict):
            pass

        def step():
            body_world.step_enqueue(0.005)  # on the same stream, ahead of the slab's kernels; the protocol's one wait covers it
            r = comm.run(stepper)
            return {"stage_ms": r.stage_ms, "mesh": {"n_vertices": r.mesh_counts[0], "n_indices": r.mesh_counts[1]},
                    "region_count": r.region_count}

        workload = (f"config-2 SDF asteroid x{mg_scale:.3f} -> {stepper.global_shape} stored grid (config 5 at N=8), "
                    f"x-slabs of {obj.chunk_counts[0]} chunk planes per rank ({obj.n_chunks} chunks on rank 0)")
        parallelism = f"x-slab domain decomposition over {world} GPUs, 1-voxel halos + region equivalences over RCCL"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    stage_sum = np.zeros(capi.N_TIMED_STAGES, dtype=np.float64)
    for _ in range(args.steps):
        res = step()
        stage_sum += res["stage_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_vox_rank = obj.n_voxels
    if dist is not None:
        t = torch.tensor([n_vox_rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        n_vox_total = int(t.item())
    else:
        n_vox_total = n_vox_rank
    tris_rank = int(res["mesh"]["n_indices"]) // 3
    if dist is not None:
        t = torch.tensor([tris_rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        tris_total = int(t.item())
    else:
        tris_total = tris_rank
    stage_ms = stage_sum / args.steps
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        # roofline of the dominant kernel (largest average launch duration measured with HIP events on the
        # library's stream), algorithmic bytes from DESIGN.md
        _, _, _, _, info = obj.download(sdf=False, types=False, flags=False, labels=False)
        exposed = int(np.count_nonzero((info["kind"] == 2) & ((info["flags"] & 0x3F) != 0x3F)))
        counters = obj.stage_counters()
        sb = stage_bytes(n_vox_rank, obj.n_chunks, exposed, int(res["mesh"]["n_vertices"]), int(res["mesh"]["n_indices"]))
        dom = int(np.argmax(stage_ms))
        name = capi.STAGE_NAMES[dom]
        achieved = sb[name] / (stage_ms[dom] * 1e-3) / 1e9
        remesh_ms = float(stage_ms[6] + stage_ms[7] + stage_ms[8])
        out = {
            "metric": "voxels stepped/sec + remesh tris/sec, 512^3 grid, 1/2/4/8 MI355X",
            "value": n_vox_total / (elapsed / args.steps),
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i8 voxels, f32 SDF/mesh arithmetic, f64 moments",
            "data": "synthetic",
            "config": {"workload": workload, "parallelism": parallelism, "voxels_per_gpu": n_vox_rank, "regions": int(res["region_count"]),
                       "triangles": tris_total, "vertices_rank0": int(res["mesh"]["n_vertices"]), "exposed_chunks_rank0": exposed,
                       "evaluated_chunks_rank0": counters["evaluated_chunks"], "meshed_chunks_rank0": counters["meshed_chunks"]},
            "remesh_tris_per_s": tris_rank / (remesh_ms * 1e-3) if remesh_ms > 0 else None,
            "remesh_ms": remesh_ms,
            "stage_ms": {capi.STAGE_NAMES[i]: round(float(stage_ms[i]), 4) for i in range(capi.N_TIMED_STAGES)},
            "stage_gbs": {capi.STAGE_NAMES[i]: round(sb[capi.STAGE_NAMES[i]] / (stage_ms[i] * 1e-3) / 1e9, 1) if stage_ms[i] > 0 and sb[capi.STAGE_NAMES[i]] > 0 else None
                          for i in range(capi.N_TIMED_STAGES)},
            "roofline": {"bound": "hbm", "kernel": "+".join(STAGE_KERNELS[name]), "stage": name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes": sb[name],
                         "traffic": measured_traffic(name) if world == 1 and abs(args.scale - 2.05) < 1e-9 else None},
            "step_roofline": {"algorithmic_bytes": float(sum(sb.values())), "achieved": float(sum(sb.values())) / (float(stage_ms.sum()) * 1e-3) / 1e9,
                              "unit": "GB/s", "frac": float(sum(sb.values())) / (float(stage_ms.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scale)
        elif world == 1:
            out["cpu_baseline"] = None
        if not args.no_pile:
            out["pile"] = pile_benchmark(ctx, with_cpu=(world == 1 and not args.no_cpu_baseline))
            if world == 1:
                out["edit"] = edit_benchmark(ctx, args.scale)
                out["collide"] = collide_benchmark(ctx, args.scale)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    body_world.close()
    obj.close()
    ctx.close()


if __name__ == "__main__":
    main()
