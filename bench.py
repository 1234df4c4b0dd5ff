#!/usr/bin/env python3
"""Throughput benchmark of the voxel hot path (BASELINE.json metric:
"voxels stepped/sec + remesh tris/sec, 512^3 grid, 1/2/4/8 MI355X").

A step = one pass of the per-frame voxel path over one SDF-defined grid that is already resident in HBM: SDF sample ->
derived state (flags, chunk state, occupied ranges) -> connected regions -> Surface Nets remesh -> mass/inertia
reduction -> rigid-body step of the object's own body. The headline workload is the config-2 asteroid scaled x2.05
(502^3 grid -> 32^3 chunks = 512^3 stored voxels).

N > 1 (one process per GPU, launched by torch.distributed.run): the grid is cut into N chunk-aligned x-slabs, one-voxel
face halos and boundary chunk state exchanged over RCCL, cross-rank region equivalences and the 10 mass moments in one
small all-gather.
  --scaling strong (default)  the SAME 512^3 grid on every N — the configuration the metric and its 1 -> 8 target are quoted on;
  --scaling weak              BASELINE config 5: all lengths x N^(1/3), every rank keeps the 512^3 voxel count (N = 8: 1024^3).

Beside the headline, rank 0 at N = 1 reports (each with the oracle timed beside it and a parity verdict):
  dense    all-surface workload (32 perforated plates, every chunk NonUniform and meshed): here the per-voxel byte accounting of
           SURVEY §8d applies to the whole grid, so `roofline.frac` of this leg IS an HBM fraction
  config2  256^3 asteroid: Surface Nets remesh + mass/inertia
  config3  256^3 fracture: seven split-offs (8 objects), then the eight octant copies by polyhedron clip
  pile     config 4: 4096 bodies / 46 080 contacts, exact-order sequential impulses
  frame    voxel step + pile step back to back (the two serial stages of a frame the headline leaves out)
  edit, collide   SURVEY §8f items 1-2 on the headline body

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0    # the guide's measured float4 copy: the practical ceiling
# G wave-instructions/s: 256 CUs x 4 SIMD-32s, one wave64 VALU instruction per TWO cycles per SIMD once two or more waves share it
# (MI355X_MICROARCH.md, "Wave scheduling" and the constants table: `v_fma_f32` 2 cyc, one wave alone 4) at 2.4 GHz = 1228.8. Rounds 1-3 divided
# by 614.4 (one per four cycles — what ONE wave sustains, and what SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycle per instruction looked
# like): with four to eight waves per SIMD that is not the ceiling. Measured beside it (tools/valu_peak.hip, tools/valu_ops.hip on this part,
# >= 2 waves per SIMD, the clock the chip holds under that load): f32 add / mul / fma and 32-bit logic ~880, shifts / bit-field / 3-operand
# integer / v_cndmask ~560, packed f32 ~520 G wave-instructions/s.
VALU_PEAK_GWI = 256 * 4 * 2.4 / 2.0
VALU_MEASURED_GWI = {"f32 add/mul/fma, logic": 880.0, "shifts, bit-field, 3-operand integer, v_cndmask": 560.0, "packed f32": 520.0}
def _default_profile_dir():
    """the newest profiles/round* directory that holds PMC traffic summaries (or --profiles)"""
    import glob

    dirs = sorted(d for d in glob.glob(os.path.join(ROOT, "profiles", "round*")) if glob.glob(os.path.join(d, "pmc_traffic_*.json")))
    return dirs[-1] if dirs else os.path.join(ROOT, "profiles", "round2")


PROFILE_DIR = _default_profile_dir()
AHEAD = os.environ.get("IVX_BENCH_SAMPLE_AHEAD", "1") != "0"  # legs that sample the resident program step after step run its pre-pass a step ahead

STAGE_KERNELS = {  # which kernels make up a timed slot (names as rocprofv3 reports them); include/impact_voxel_hip.h, IVX_N_TIMED_STAGES
    "sdf_sample": ["k_sdf_super", "k_sdf_prepass", "k_sdf_eval"],
    "derive": ["k_chunk_pre", "k_derive"],  # k_derive also labels the chunk-local regions and leaves the chunk moments (fused sweep)
    "post1": ["k_step_post1"],    # roles: mesher count | region merge by chunk columns | exact local numbering | occupied slots | moment partials
    "post2": ["k_step_post2"],    # roles: multi-region merge | mesher scan | moments and occupied ranges final
    "emit": ["k_step_emit"],      # roles: region forest flatten | mesher emit
    "assign": ["k_step_assign"],  # component ids
}


def stage_bytes(n_voxels, n_chunks, exposed_chunks, n_vertices, n_indices):
    """Algorithmic HBM bytes per step of each timed slot: SURVEY.md §8(d)'s per-voxel figures x the voxels the slot is
    charged for, the mesher for the padded tiles of the chunks it meshes plus its output. `n_voxels` is what the caller
    decides to charge: the STORED grid for the `effective` figures (compact planes: most of a solid body's chunks are 8-byte
    records that never touch the planes, so an effective rate can exceed what the HBM moves), or the ACTIVE voxels
    (chunks with planes) for a figure that can be compared with the HBM peak."""
    out = {
        "sdf_sample": 2.0 * n_voxels,                        # W sdf + type
        "derive": 4.0 * n_voxels,                            # R sdf + type, W flags + label (fused sweep)
        # 18^3 sign bits per exposed chunk (mesher count) + chunk records and touch bytes of a chunk column and its two neighbours
        # (region merge) + one packed box per chunk (occupied) + chunk records and per-chunk moment slots (moment sums)
        "post1": 5832.0 / 8 * exposed_chunks + (27.0 + 4.0 + 88.0) * n_chunks,
        "post2": 36.0 * n_chunks,                            # mesher scan: counts in, offsets / ranks / emit records out
        # mesher emit: tile + (pos, nrm) + (idx u32, imat 8 B); flatten: (chunk, region) table entries in use
        # (24 B per vertex. Rounds 1-2 charged 40: the mesher wrote a 16-byte material record per vertex as scratch; it no longer does)
        "emit": 2.0 * 5832 * exposed_chunks + 24.0 * n_vertices + 12.0 * n_indices + 8.0 * n_chunks,
        "assign": 8.0 * n_chunks,
    }
    for k in ("unused6", "unused7", "unused8", "unused9"):
        out[k] = 0.0
    return out


def load_profile(name):
    try:
        return json.load(open(os.path.join(PROFILE_DIR, name)))
    except (OSError, ValueError):
        return None


def _profile_rel():
    return os.path.relpath(PROFILE_DIR, ROOT)


def _profile_current(d):
    """a committed PMC summary is quoted only for the kernels it was measured on: the summary carries the hash of the kernel sources
    (tools/source_sha.py); another tree's numbers are not this tree's traffic"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_sha import source_sha16

    return d.get("_source_sha16") == source_sha16()


def counted_traffic(workload):
    """HBM bytes PER STEP of every kernel of the voxel step, from the committed PMC passes of the builder's own run of this
    workload (<profile dir>/pmc_traffic_<workload>.json, written by tools/pmc_traffic.py from separate FETCH_SIZE /
    WRITE_SIZE rocprofv3 runs: per-launch average x launches per step). A committed constant, not something this run
    measured: `traffic_source` says so in the output. A summary that lacks a kernel of a timed slot is an error (the kernels were
    renamed or re-split and the summary is of something else); a summary measured on other kernel sources is not quoted."""
    d = load_profile(f"pmc_traffic_{workload}.json")
    if not d:
        return None, None
    if not _profile_current(d):
        return None, (f"{_profile_rel()}/pmc_traffic_{workload}.json was measured on other kernel sources (source hash {d.get('_source_sha16')}): not quoted; "
                      "tools/profile_round.sh regenerates it")
    per_stage = {}
    for stage, kernels in STAGE_KERNELS.items():
        tot, found = 0.0, False
        for k in kernels:
            for name, rec in d.items():
                if isinstance(rec, dict) and (name == k or name.startswith("void " + k + "<") or name.startswith(k + "<")):
                    tot += rec["hbm_bytes_per_step"]
                    found = True
        if not found and stage != "sdf_sample":  # (k_sdf_super runs only for programs of more than 2048 nodes)
            raise RuntimeError(f"{_profile_rel()}/pmc_traffic_{workload}.json has none of the kernels {kernels} of the timed slot '{stage}': "
                               "STAGE_KERNELS and the summary are out of step (re-run tools/profile_round.sh)")
        per_stage[stage] = tot if found else None
    return per_stage, (f"{_profile_rel()}/pmc_traffic_{workload}.json (builder's rocprofv3 --pmc run of this workload on these kernel sources; per-launch average x "
                       "launches per step)")


def roofline_block(stage_ms, sb_effective, sb_active, workload_key):
    """`roofline` for the stage with the largest measured time + `step_roofline` for the whole step."""
    from impact_amd import capi

    names = capi.STAGE_NAMES
    dom = int(np.argmax(stage_ms))
    name = names[dom]
    t = stage_ms[dom] * 1e-3
    traffic, source = counted_traffic(workload_key)
    eff = sb_effective[name] / t / 1e9
    act = sb_active[name] / t / 1e9
    ran = [k for k in STAGE_KERNELS[name] if k != "k_sdf_super"]  # (k_sdf_super runs only for programs of more than 2048 nodes: none of the bench's)
    rl = {"bound": "hbm", "kernel": "+".join(ran), "stage": name, "achieved": act, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": act / HBM_PEAK_GBS, "algorithmic_bytes": sb_active[name],
          "accounting": "SURVEY §8d bytes per voxel x ACTIVE voxels (chunks that have planes); the stored-grid figure is `effective_*`",
          "effective_achieved": eff, "effective_frac": eff / HBM_PEAK_GBS, "effective_algorithmic_bytes": sb_effective[name],
          "traffic": None, "traffic_source": source, "counter_frac": None}
    if traffic and traffic.get(name) is not None:
        rl["traffic"] = traffic[name]
        rl["counter_frac"] = traffic[name] / t / 1e9 / HBM_PEAK_GBS
    if name == "sdf_sample":
        # the slot is two launches: the interval pre-pass (latency of one workgroup, no voxel bytes) and the evaluator, which is bound by
        # VALU issue — an HBM fraction says nothing about it; `valu_roofline` carries the figure that does (1228.8 G wave-instructions/s by
        # the guide, 880 measured for simple ops). The dominant KERNEL's own HBM fraction by its share of the slot's time in the committed trace:
        ks = load_profile(f"kernel_share_{workload_key}.json") if workload_key else None
        if ks and _profile_current(ks) and ks.get("k_sdf_eval_share_of_sdf_sample"):
            share = float(ks["k_sdf_eval_share_of_sdf_sample"])
            rl["dominant_kernel"] = "k_sdf_eval"
            rl["dominant_kernel_frac"] = act / share / HBM_PEAK_GBS
            rl["dominant_kernel_source"] = f"{_profile_rel()}/kernel_share_{workload_key}.json (k_sdf_eval's share of the slot's time in the rocprofv3 kernel trace: {share:.3f})"
        rl["bound_note"] = "the evaluator is VALU-bound: see `valu_roofline` (bound: valu); the HBM fraction here is the contract's figure for the dominant slot"
    tt = float(stage_ms.sum()) * 1e-3
    srl = {"algorithmic_bytes": float(sum(sb_active.values())), "achieved": float(sum(sb_active.values())) / tt / 1e9, "unit": "GB/s",
           "frac": float(sum(sb_active.values())) / tt / 1e9 / HBM_PEAK_GBS,
           "effective_algorithmic_bytes": float(sum(sb_effective.values())),
           "effective_frac": float(sum(sb_effective.values())) / tt / 1e9 / HBM_PEAK_GBS, "counter_bytes": None, "counter_frac": None}
    if traffic and all(v is not None for k, v in traffic.items() if sb_effective.get(k, 0) > 0):
        cb = float(sum(v for v in traffic.values() if v is not None))
        srl["counter_bytes"] = cb
        srl["counter_frac"] = cb / tt / 1e9 / HBM_PEAK_GBS
    # the sampler is bound by VALU issue, not by HBM: its instruction count from the committed SQ pass against the issue peak
    valu = load_profile(f"pmc_valu_{workload_key}.json")
    vrl = None
    if valu and stage_ms[0] > 0 and _profile_current(valu):
        wi = sum(rec["SQ_INSTS_VALU_per_step"] for k, rec in valu.items() if isinstance(rec, dict) and any(
            k == n or k.startswith("void " + n + "<") or k.startswith(n + "<") for n in STAGE_KERNELS["sdf_sample"]))
        ach = wi / (stage_ms[0] * 1e-3) / 1e9
        act = sum(rec.get("SQ_ACTIVE_INST_VALU_per_step", 0.0) for k, rec in valu.items() if isinstance(rec, dict) and any(
            k == n or k.startswith("void " + n + "<") or k.startswith(n + "<") for n in STAGE_KERNELS["sdf_sample"]))
        vrl = {"stage": "sdf_sample", "bound": "valu", "achieved": ach, "peak": VALU_PEAK_GWI, "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_GWI,
               "peak_accounting": "one wave64 instruction per SIMD per 2 cycles at 2.4 GHz (the guide); `measured_issue_rates` = what this part sustained "
                                  "per opcode class in tools/valu_peak.hip / valu_ops.hip at the clock it holds under load",
               "measured_issue_rates": VALU_MEASURED_GWI, "frac_of_measured_simple_op_rate": ach / VALU_MEASURED_GWI["f32 add/mul/fma, logic"],
               "wave_instructions_per_step": wi, "active_quad_cycles_per_step": act,
               "source": f"{_profile_rel()}/pmc_valu_{workload_key}.json (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU; builder's run on these kernel sources)"}
    return rl, srl, vrl


# ---------------------------------------------------------------------------------------------------------------------------
# CPU legs (the oracle: tests/oracle_lib.py -> oracle/liboracle.so). Only these functions touch it.
def cpu_voxel_step(graph, threads=1):
    """the oracle over one workload, timed: (object with derived state, mesh, timings dict)"""
    import oracle_lib as ol

    t0 = time.perf_counter()
    if threads > 1:
        o = ol.OracleObject.from_sdf_parallel(graph, 1.0, 0, threads)
    else:
        o = ol.OracleObject.from_sdf(graph, 1.0, 0)
        o.update_occupied_voxel_ranges()
        o.compute_all_derived_state()
    t1 = time.perf_counter()
    m = o.mesh_parallel(threads) if threads > 1 else o.mesh()
    t2 = t1 + o.last_mesh_seconds  # (without the export of the mesh buffers into numpy arrays)
    t2b = time.perf_counter()
    if threads > 1:
        o.inertia_parallel(threads)
    else:
        o.inertia()
    t3 = time.perf_counter()
    return o, m, {"generate_derive_s": t1 - t0, "remesh_s": t2 - t1, "inertia_s": t3 - t2b, "total_s": (t2 - t0) + (t3 - t2b)}


def cpu_baseline(graph, obj, res, what, all_cores=True):
    """`cpu_baseline` (1 thread) + `cpu_baseline_all_cores` + `parity` of the GPU step just timed against the oracle object"""
    import oracle_lib as ol
    import parity_util as pu

    o, m, t = cpu_voxel_step(graph, 1)
    cc = o.chunk_counts
    nvox = cc[0] * cc[1] * cc[2] * 4096
    base = {"value": nvox / t["total_s"], "unit": "voxels/s", "cores": 1, "kind": "port",
            "sample": f"{what} once ({cc[0] * 16}x{cc[1] * 16}x{cc[2] * 16} stored voxels): generate+derive {t['generate_derive_s']:.2f}s, remesh "
                      f"{t['remesh_s']:.2f}s ({m.indices.size // 3 / max(t['remesh_s'], 1e-9):.3g} tris/s), inertia {t['inertia_s']:.2f}s; single thread"}
    parity = pu.step_parity(o, obj, res) if obj is not None else None
    allc = None
    if all_cores and hasattr(ol.OracleObject, "from_sdf_parallel"):
        n = os.cpu_count() or 1
        try:
            n = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            pass
        n = min(n, 16)  # the share of the host a one-GPU box grants (more threads than that only adds contention on these sizes)
        if n > 1:
            _, m2, t2 = cpu_voxel_step(graph, n)
            allc = {"value": nvox / t2["total_s"], "unit": "voxels/s", "cores": n, "kind": "port",
                    "sample": f"same workload, OpenMP over chunks ({n} threads): generate+derive {t2['generate_derive_s']:.2f}s, remesh {t2['remesh_s']:.2f}s, "
                              f"inertia {t2['inertia_s']:.2f}s; the cross-chunk adjacency pass and the region resolve stay serial, as in the reference",
                    "same_triangles": bool(m2.indices.size == m.indices.size)}
    return base, allc, parity, o, m


def cpu_baseline_all_cores(graph, obj, res, what):
    """the oracle's OpenMP entry points alone over one workload (no single-thread pass) + the parity of the GPU object just timed"""
    import parity_util as pu

    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    n = max(1, min(n, 16))
    o, m, t = cpu_voxel_step(graph, n)
    cc = o.chunk_counts
    nvox = cc[0] * cc[1] * cc[2] * 4096
    base = {"value": nvox / t["total_s"], "unit": "voxels/s", "cores": n, "kind": "port",
            "sample": f"{what} once ({cc[0] * 16}x{cc[1] * 16}x{cc[2] * 16} stored voxels), OpenMP over chunks ({n} threads): generate+derive "
                      f"{t['generate_derive_s']:.2f}s, remesh {t['remesh_s']:.2f}s ({m.indices.size // 3 / max(t['remesh_s'], 1e-9):.3g} tris/s), inertia "
                      f"{t['inertia_s']:.2f}s; the cross-chunk adjacency pass and the region resolve stay serial, as in the reference"}
    parity = pu.step_parity(o, obj, res, mesh=m) if obj is not None else None
    return base, parity


def compact_parity(p):
    """a `parity` block as the line prints it: the verdict, ONE digest over everything the GPU side was compared on, and the names of what
    differed (none when equal) — the per-buffer digest pairs are what tests/test_gpu_parity.py asserts on; the line only has to carry the verdict"""
    if not isinstance(p, dict) or "voxel_sha" not in p:
        return p
    import hashlib

    pairs = {k: v for k, v in p.items() if isinstance(v, list) and len(v) == 2}
    digest = hashlib.sha256("|".join(f"{k}={v[0]}" for k, v in sorted(pairs.items())).encode()).hexdigest()[:16]
    return {"equal": bool(p.get("equal")), "digest": digest, "compared": sorted(pairs), "differing": sorted(k for k, v in pairs.items() if v[0] != v[1]),
            "triangles": p["triangles"][0], "moments_rel": p.get("moments_rel")}


def compact_parities(x):
    if isinstance(x, dict):
        return {k: (compact_parity(v) if isinstance(v, dict) and "voxel_sha" in v else compact_parities(v)) for k, v in x.items()}
    return x


def summary_block(out):
    """the figures a reader of the line's tail needs, as its LAST key (the driver's record keeps the tail of the line)"""
    def g(*path):
        x = out
        for k in path:
            if not isinstance(x, dict) or x.get(k) is None:
                return None
            x = x[k]
        return round(x, 4) if isinstance(x, float) else x

    parities, failed = [], []

    def walk(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                if k in ("parity", "parity_sample") and isinstance(v, dict) and "equal" in v:
                    parities.append(bool(v["equal"]))
                    if not v["equal"]:
                        failed.append(f"{path}/{k}: " + ",".join(str(n) for n in (v.get("differing") or [n_ for n_, b in v.items() if b is False])))
                elif k == "parity" and isinstance(v, dict):
                    for n, b in v.items():
                        if isinstance(b, bool):
                            parities.append(b)
                            if not b:
                                failed.append(f"{path}/{k}/{n}")
                elif k == "parity" and isinstance(v, str):  # (the legs whose comparison is a sentence: "MISMATCH" when it failed)
                    parities.append(v != "MISMATCH")
                    if v == "MISMATCH":
                        failed.append(f"{path}/{k}")
                else:
                    walk(v, f"{path}/{k}")

    walk(out)
    return {"ms_per_step": g("ms_per_step"), "ms_per_step_isolated": g("sample_ahead", "isolated", "ms_per_step"), "sdf_sample_ms": g("stage_ms", "sdf_sample"),
            "sdf_sample_isolated_ms": g("sample_ahead", "isolated", "sdf_sample_ms"), "roofline_frac": g("roofline", "frac"),
            "valu_frac": g("valu_roofline", "frac"), "step_counter_frac": g("step_roofline", "counter_frac"),
            "dense": {"ms_per_step": g("dense", "ms_per_step"), "ms_per_step_isolated": g("dense", "ms_per_step_isolated"), "emit_ms": g("dense", "emit_ms"), "roofline_kernel": g("dense", "roofline", "kernel"),
                      "roofline_frac": g("dense", "roofline", "frac"), "counter_frac": g("dense", "roofline", "counter_frac"),
                      "step_counter_frac": g("dense", "step_roofline", "counter_frac")},
            "pile": {"ms_per_step": g("pile", "ms_per_step"), "solve_ms": g("pile", "stage_ms", "solve"), "kernel": g("pile", "solver", "kernel"),
                     "churn_ms_per_frame": g("pile", "churn", "ms_per_frame_spread"), "churn_set_contacts_host_ms": g("pile", "churn", "set_contacts_host_ms_spread")},
            "frame_pipeline_ms": g("frame", "pipeline", "ms_per_frame"), "frame_two_streams_ms": g("frame", "ms_per_frame_two_streams"),
            "edit_plus_sync_ms": g("edit", "edit_plus_sync_ms"), "edit_and_sync_overlapped_ms": g("edit", "edit_and_sync_overlapped_ms"),
            "fragments": {"cut_ms": g("fragments", "cut_ms"), "first_step_many_ms": g("fragments", "first_step_many_ms"), "frame_many_ms": g("fragments", "frame_many_ms"),
                          "frame_looped_ms": g("fragments", "frame_looped_ms")},
            "fragments_frame": {"ms_batched": g("fragments_frame", "ms_batched"), "ms_looped": g("fragments_frame", "ms_looped"),
                                "ms_batched_world_on_own_context": g("fragments_frame", "ms_batched_world_on_own_context"),
                                "probes_sync_ms": [g("fragments_frame", "probes_and_pairs", "probes_sync", "ms_batched"),
                                                   g("fragments_frame", "probes_and_pairs", "probes_sync", "ms_looped")],
                                "host_spreads_ms": {"probes_sync": g("fragments_frame", "probes_and_pairs", "probes_sync", "ms_batched_spread"),
                                                    "mutual_pairs": g("fragments_frame", "probes_and_pairs", "mutual_pairs", "ms_batched_spread"),
                                                    "contacts_many": g("fragments_frame", "batched_host_ms_spread", "contacts_many"),
                                                    "set_contacts": g("fragments_frame", "batched_host_ms_spread", "set_contacts"),
                                                    "frame": g("fragments_frame", "batched_host_ms_spread", "frame"), "cut": g("fragments", "cut_ms_spread")},
                                "mutual_pairs_ms": [g("fragments_frame", "probes_and_pairs", "mutual_pairs", "ms_batched"),
                                                    g("fragments_frame", "probes_and_pairs", "mutual_pairs", "ms_looped")]},
            "config3_split_loop_ms": g("config3", "split_loop_ms"), "config5_single_grid_ms": g("config5_one_gpu", "single_grid_ms"), "config5_single_grid_isolated_ms": g("config5_one_gpu", "single_grid_isolated_ms"),
            "strong_scaling_bound": {"512": g("strong_512", "strong_scaling_bound"), "1024": g("config5_one_gpu", "strong_scaling_bound"),
                                     "strong_512_eight_slabs_one_gpu_ms": g("strong_512", "eight_slabs_one_gpu_ms")},
            "config5_eight_slabs_one_gpu_ms": g("config5_one_gpu", "eight_slabs_one_gpu_ms"),
            "cpu_baseline_voxels_per_s": g("cpu_baseline", "value"), "parity_all_equal": (all(parities) if parities else None), "parity_blocks": len(parities), "parity_failed": failed}


# ---------------------------------------------------------------------------------------------------------------------------
def make_object(ctx, graph, sample_ahead=False):
    """`sample_ahead`: for a leg that samples the resident program step after step (ivx_grid_set_sample_ahead, see main)"""
    from impact_amd.voxel import SDFVoxelGenerator, VoxelObject

    gen = SDFVoxelGenerator(1.0, graph, 0)
    obj = VoxelObject(ctx, gen.chunk_counts(), 1.0)
    obj.set_sdf_program(gen)
    obj.set_densities(np.ones(256, dtype=np.float32))
    if sample_ahead and os.environ.get("IVX_BENCH_SAMPLE_AHEAD", "1") != "0":
        obj.set_sample_ahead(True)
    return gen, obj


def time_steps(ctx, obj, stages, steps, warmup):
    from impact_amd import capi

    gc.collect()  # (the collector itself is off for the whole run, see main(): collections happen here, between timed regions)

    for _ in range(warmup):
        res = obj.step(stages)
    ctx.synchronize()
    acc = np.zeros(capi.N_TIMED_STAGES)
    t0 = time.perf_counter()
    walls, samples = [], []
    for _ in range(steps):
        tw = time.perf_counter()
        res = obj.step(stages)
        walls.append(time.perf_counter() - tw)
        acc += res["stage_ms"]
        samples.append(np.array(res["stage_ms"], dtype=np.float64))
    ctx.synchronize()
    time_steps.last_samples = np.array(samples)  # [step][slot] ms: min / median / max of a slot over the timed steps (dense_benchmark)
    if os.environ.get("IVX_BENCH_TRACE"):
        print("[bench] time_steps walls (ms):", np.round(1e3 * np.array(walls), 3).tolist(), file=sys.stderr)
    return res, 1e3 * (time.perf_counter() - t0) / steps, acc / steps


def remesh_only_ms(ctx, obj, steps=10):
    """Surface Nets alone (count + scan + emit) over the resident, derived object: the sum of the three launches' HIP-event times"""
    from impact_amd import capi

    _, _, sm = time_steps(ctx, obj, capi.STAGE_REMESH, steps, 2)
    return float(sm[2] + sm[3] + sm[4])


def chunk_census(obj):
    info = obj.download(sdf=False, types=False, flags=False, labels=False)[4]
    exposed = int(np.count_nonzero((info["kind"] == 2) & ((info["flags"] & 0x3F) != 0x3F)))
    non_uniform = int(np.count_nonzero(info["kind"] == 2))
    return exposed, non_uniform


def dense_benchmark(ctx, args, with_cpu):
    """the all-surface workload: every chunk NonUniform, every chunk meshed"""
    from impact_amd import capi, scenes

    n = args.dense_chunks
    graph = scenes.plates_scene(n)
    gen, obj = make_object(ctx, graph)
    # the same step first with its pre-pass as its own first kernel (`ms_per_step_isolated`), then — the leg's figures — with the pre-pass a step ahead
    _, ms_isolated, stage_isolated = time_steps(ctx, obj, capi.STAGE_ALL, max(3, args.steps // 4), max(3, 3 * args.warmup))
    obj.set_sample_ahead(os.environ.get("IVX_BENCH_SAMPLE_AHEAD", "1") != "0")
    # (warm-up: this leg starts after seconds of host-side oracle work with the GPU idle — over the first ~50 steps the per-step wall still falls
    # by 6 %, clocks ramping up, so two warm-up steps as in round 3 measured the ramp)
    res, ms, stage_ms = time_steps(ctx, obj, capi.STAGE_ALL, max(3, args.steps // 2), max(3, 3 * args.warmup))
    exposed, non_uniform = chunk_census(obj)
    counters = obj.stage_counters()
    active = max(non_uniform, counters["evaluated_chunks"]) * 4096
    sb_eff = stage_bytes(obj.n_voxels, obj.n_chunks, exposed, int(res["mesh"]["n_vertices"]), int(res["mesh"]["n_indices"]))
    sb_act = stage_bytes(active, obj.n_chunks, exposed, int(res["mesh"]["n_vertices"]), int(res["mesh"]["n_indices"]))
    rl, srl, _ = roofline_block(stage_ms, sb_eff, sb_act, "dense")
    emit = time_steps.last_samples[:, 4]  # the mesher's launch, step by step: which of its two run-to-run modes this process has
    emit_ms = {"min": round(float(emit.min()), 4), "median": round(float(np.median(emit)), 4), "max": round(float(emit.max()), 4), "steps": int(emit.size)}
    remesh_ms = remesh_only_ms(ctx, obj, max(3, args.steps // 4))
    out = {"workload": f"{n} perforated plates, one per chunk layer: {gen.grid_shape()} grid = {(n * 16)}^3 stored voxels, {non_uniform} of {obj.n_chunks} chunks "
                       f"NonUniform, {exposed} meshed",
           "ms_per_step": ms, "ms_per_step_isolated": ms_isolated, "sdf_sample_isolated_ms": round(float(stage_isolated[0]), 4),
           "voxels_per_s": obj.n_voxels / (ms * 1e-3), "active_voxels": active, "triangles": int(res["mesh"]["n_indices"]) // 3,
           "remesh_tris_per_s": int(res["mesh"]["n_indices"]) // 3 / (remesh_ms * 1e-3), "regions": int(res["region_count"]),
           "stage_ms": {capi.STAGE_NAMES[i]: round(float(stage_ms[i]), 4) for i in range(capi.N_TIMED_STAGES)},
           "stage_gbs": {capi.STAGE_NAMES[i]: round(sb_act[capi.STAGE_NAMES[i]] / (stage_ms[i] * 1e-3) / 1e9, 1) if stage_ms[i] > 0 and sb_act[capi.STAGE_NAMES[i]] > 0 else None
                         for i in range(capi.N_TIMED_STAGES)},
           "emit_ms": emit_ms, "roofline": rl, "step_roofline": srl}
    if with_cpu:
        # bounded CPU sample: the same scene at 1/64 of the volume (128^3), GPU and oracle both, with the parity verdict; the
        # full-size parity check is tests/test_gpu_parity.py::test_dense_workload_512
        small = scenes.plates_scene(max(2, n // 4))
        _, sobj = make_object(ctx, small)
        sres = sobj.step(capi.STAGE_ALL)
        base, allc, parity, _, _ = cpu_baseline(small, sobj, sres, f"the same scene with {max(2, n // 4)} plates", all_cores=True)
        out["cpu_baseline_one_thread_sample"], out["cpu_baseline_all_cores_sample"], out["parity_sample"] = base, allc, parity
        sobj.close()
        # ... and the TIMED workload itself on all the host cores the box grants (seconds; one thread would take the half minute), with the
        # parity verdict of the object just timed
        out["cpu_baseline"], out["parity"] = cpu_baseline_all_cores(graph, obj, res, "the timed all-surface workload")
        out["cpu_baseline_all_cores"] = out["cpu_baseline"]
    obj.close()
    return out


def config2_benchmark(ctx, args, with_cpu):
    """BASELINE config 2: 256^3 SDF asteroid, Surface Nets remesh + mass/inertia recompute (the voxels already resident)"""
    from impact_amd import capi, scenes

    graph = scenes.asteroid_scene(1.0)
    gen, obj = make_object(ctx, graph)
    obj.step(capi.STAGE_ALL)
    res, ms, stage_ms = time_steps(ctx, obj, capi.STAGE_REMESH | capi.STAGE_INERTIA, args.steps, 2)
    full, ms_full, _ = time_steps(ctx, obj, capi.STAGE_ALL, args.steps, 1)
    tris = int(res["mesh"]["n_indices"]) // 3
    out = {"workload": f"config-2 asteroid: {gen.grid_shape()[0]}^3 grid = {obj.chunk_counts[0] * 16}^3 stored voxels; remesh + inertia of the resident object",
           "remesh_inertia_ms": ms, "remesh_tris_per_s": tris / (remesh_only_ms(ctx, obj) * 1e-3), "triangles": tris,
           "full_step_ms": ms_full, "voxels_per_s_full_step": obj.n_voxels / (ms_full * 1e-3)}
    if with_cpu:
        base, allc, parity, _, _ = cpu_baseline(graph, obj, full, "config 2")
        out["cpu_baseline"], out["cpu_baseline_all_cores"], out["parity"] = base, allc, parity
    obj.close()
    return out


def config5_benchmark(ctx, args):
    """BASELINE config 5 on the ONE GPU there is: the 1024^3 grid (config-2 asteroid x4.2, 64^3 chunks) as one resident grid, and
    domain-decomposed in 8 x-slabs of 8 chunk planes with all eight slabs on this GPU (the in-process communicator: neighbour exchange
    and all-gather are device copies; the driver code is what 8 RCCL ranks run). tests/test_gpu_slabs.py::test_config5_1024_in_8_slabs
    holds the two equal, voxel byte for voxel byte."""
    from impact_amd import capi, scenes
    from impact_amd.distributed import NativeComm, NativeSlabStepper, NativeStepGroup

    graph = scenes.asteroid_scene(4.2)
    gen, obj = make_object(ctx, graph)
    steps = max(5, args.steps // 10)
    _, ms_isolated, _ = time_steps(ctx, obj, capi.STAGE_ALL, steps, 2)
    obj.set_sample_ahead(os.environ.get("IVX_BENCH_SAMPLE_AHEAD", "1") != "0")  # (the slabs below as well: sample_ahead=True)
    res, ms, stage_ms = time_steps(ctx, obj, capi.STAGE_ALL, steps, 2)
    tris = int(res["mesh"]["n_indices"]) // 3
    n_vox = obj.n_voxels
    out = {"workload": f"config-2 asteroid x4.2: {obj.chunk_counts[0] * 16}^3 stored voxels, one GPU", "single_grid_ms": ms, "single_grid_isolated_ms": ms_isolated,
           "single_grid_voxels_per_s": n_vox / (ms * 1e-3), "triangles": tris, "stage_ms": {capi.STAGE_NAMES[i]: round(float(stage_ms[i]), 4) for i in range(capi.N_TIMED_STAGES) if stage_ms[i] > 0}}
    obj.close()
    comm = NativeComm(ctx, 8, local=True)
    steppers = [NativeSlabStepper(ctx, comm, graph, np.ones(256, dtype=np.float32), r, sample_ahead=AHEAD) for r in range(8)]
    group = NativeStepGroup(steppers)
    for s_ in steppers:  # (no event records around the stage slots: 12 per slab and step, ~2-3 us each on the queue; the N > 1 timed region keeps one slot's)
        s_.obj.set_stage_timing(0)
    for _ in range(2):
        group.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        rec = group.step()
    slab_ms = 1e3 * (time.perf_counter() - t0) / steps
    out["eight_slabs_one_gpu_ms"] = slab_ms
    out["eight_slabs_note"] = "stage-slot event records off (round 4 timed this leg with all twelve per slab on: 0.3 ms of the 1.84)"
    out["eight_slabs_triangles"] = int(sum(int(r["mesh"]["n_indices"]) for r in rec)) // 3
    out["eight_slabs_regions"] = int(rec[0]["region_count"])
    out["same_triangles"] = out["eight_slabs_triangles"] == tris
    out["strong_scaling_bound"] = round(ms / (slab_ms / 8.0), 2)
    out["strong_scaling_bound_what"] = ("single_grid_ms / (eight_slabs_one_gpu_ms / 8): the speed-up eight GPUs could reach on this grid if the links were free — what one "
                                        "rank's launches, waits and exchanges cost by themselves, measured with all eight ranks taking turns on one GPU")
    for s_ in steppers:
        s_.close()
    comm.close()
    return out


def strong_512_benchmark(ctx, args, single_ms, tris):
    """The metric's own 512^3 grid cut into 8 x-slabs of 4 chunk planes, all eight on this GPU (the in-process communicator): what one rank of the
    strong-scaling leg costs before any link is involved."""
    from impact_amd import scenes
    from impact_amd.distributed import NativeComm, NativeSlabStepper, NativeStepGroup

    graph = scenes.asteroid_scene(2.05)
    comm = NativeComm(ctx, 8, local=True)
    steppers = [NativeSlabStepper(ctx, comm, graph, np.ones(256, dtype=np.float32), r, sample_ahead=AHEAD) for r in range(8)]
    group = NativeStepGroup(steppers)
    for s_ in steppers:
        s_.obj.set_stage_timing(0)
    steps = max(10, args.steps // 10)
    for _ in range(3):
        group.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        rec = group.step()
    slab_ms = 1e3 * (time.perf_counter() - t0) / steps
    out = {"workload": "the headline 512^3 grid as 8 x-slabs of 4 chunk planes, all on one GPU (in-process communicator: exchanges are device copies)",
           "single_grid_ms": single_ms, "eight_slabs_one_gpu_ms": slab_ms, "per_rank_ms": slab_ms / 8.0, "strong_scaling_bound": round(single_ms / (slab_ms / 8.0), 2),
           "eight_slabs_triangles": int(sum(int(r["mesh"]["n_indices"]) for r in rec)) // 3, "eight_slabs_regions": int(rec[0]["region_count"])}
    out["same_triangles"] = out["eight_slabs_triangles"] == tris
    for s_ in steppers:
        s_.close()
    comm.close()
    return out


def octant_boxes(centre):
    import itertools

    for sx, sy, sz in itertools.product((-1, 1), repeat=3):
        lo = [centre if s > 0 else -10.0 for s in (sx, sy, sz)]
        hi = [300.0 if s > 0 else centre for s in (sx, sy, sz)]
        planes = []
        for d in range(3):
            nrm = [0.0, 0.0, 0.0]
            nrm[d] = 1.0
            planes.append((*nrm, float(hi[d])))
            nrm[d] = -1.0
            planes.append((*nrm, -float(lo[d])))
        yield np.array(planes, dtype=np.float32), np.array([*lo, *hi], dtype=np.float32)


def config3_benchmark(ctx, args, with_cpu):
    """BASELINE config 3: the 256^3 fracture body. (a) the split-off loop of `handle_voxel_object_after_removing_voxels`
    (interaction.rs:256: while find_two_disconnected_regions -> extract_disconnected_region): seven extractions leave 8 objects;
    (b) polyhedron COPY of each octant (3 cutting + 3 far planes), one call per fragment and — when the library has it — all
    eight in one batched call (fracturing.rs:1047-1189 runs the fragments of an impact in parallel)."""
    from impact_amd import capi, scenes
    from impact_amd.voxel import SDFVoxelGenerator

    graph = scenes.fracture_scene()
    gen0 = SDFVoxelGenerator(1.0, graph, 0)
    centre = float(gen0.shifted_grid_center[0]) + 0.5
    boxes = list(octant_boxes(centre))
    reps = 3
    t_split, t_clip, t_batch, t_all = [], [], [], []
    n_children = n_copies = 0
    child_voxels, copy_voxels, all_voxels = [], [], []
    for rep in range(reps):
        _, obj = make_object(ctx, graph)
        obj.step(capi.STAGE_ALL)
        ctx.synchronize()
        # copies first (they leave the parent untouched)
        t0 = time.perf_counter()
        kids = []
        for planes, aabb in boxes:
            rc, child, _ = obj.copy_polyhedron(aabb, planes)
            if rc == 1:
                kids.append(child)
        ctx.synchronize()
        t_clip.append(time.perf_counter() - t0)
        n_copies = len(kids)
        if rep == reps - 1:
            copy_voxels = [int(k.describe_regions()["voxel_count"].sum()) for k in kids]
        for k in kids:
            k.close()
        if hasattr(obj, "copy_polyhedra"):
            t0 = time.perf_counter()
            kids = obj.copy_polyhedra([b[1] for b in boxes], [b[0] for b in boxes])
            ctx.synchronize()
            t_batch.append(time.perf_counter() - t0)
            for _, k, _ in kids:
                if k is not None:
                    k.close()
        t0 = time.perf_counter()
        kids = []
        while True:
            rc, child, _, moved = obj.extract_any_disconnected_region()
            if rc == 0:
                break
            if rc == 1:
                kids.append((child, int(moved["voxel_count"])))
        ctx.synchronize()
        t_split.append(time.perf_counter() - t0)
        n_children = len(kids)
        if rep == reps - 1:
            child_voxels = [v for _, v in kids]
        for k, _ in kids:
            k.close()
        obj.close()
        if hasattr(obj, "extract_all_disconnected_regions"):  # the same loop as ONE call (ivx_split_off_all) on a fresh body
            _, obj2 = make_object(ctx, graph)
            obj2.step(capi.STAGE_ALL)
            ctx.synchronize()
            t0 = time.perf_counter()
            got = obj2.extract_all_disconnected_regions()
            ctx.synchronize()
            t_all.append(time.perf_counter() - t0)
            if rep == reps - 1:
                all_voxels = [int(m["voxel_count"]) for rc, _, _, m in got if rc == 1]
            for _, k, _, _ in got:
                if k is not None:
                    k.close()
            obj2.close()
    out = {"workload": "config-3 fracture body (256^3 stored voxels, 8 octants): split-off loop until one region is left; polyhedron copy of each octant",
           "split_loop_ms": 1e3 * float(np.mean(t_all[1:])) if t_all else 1e3 * float(np.mean(t_split[1:])),
           "split_loop_call": "ivx_split_off_all (the loop as one call)" if t_all else "ivx_split_off_smallest_region, looped",
           "split_loop_one_by_one_ms": 1e3 * float(np.mean(t_split[1:])), "split_offs": n_children, "objects_after": n_children + 1,
           "octant_copies_ms": 1e3 * float(np.mean(t_clip[1:])), "octant_copies": n_copies,
           "octant_copies_batched_ms": 1e3 * float(np.mean(t_batch[1:])) if t_batch else None}
    if with_cpu:
        import oracle_lib as ol

        o = ol.OracleObject.from_sdf(graph, 1.0, 0)
        o.update_occupied_voxel_ranges()
        o.compute_all_derived_state()
        t0 = time.perf_counter()
        cvox = []
        for planes, aabb in boxes:
            rc, co, _ = o.clip_polyhedron(planes, aabb, copy=True)
            if rc == 1:
                cvox.append(int(np.count_nonzero((co.export_dense()[2] & 1) == 0)))
        t1 = time.perf_counter()
        svox = []
        while True:
            rc, co, _ = o.split_off_smallest_region()
            if rc == 0:
                break
            if rc == 1:
                svox.append(int(np.count_nonzero((co.export_dense()[2] & 1) == 0)))
        t2 = time.perf_counter()
        out["cpu_baseline"] = {"split_loop_ms": 1e3 * (t2 - t1), "octant_copies_ms": 1e3 * (t1 - t0), "cores": 1, "kind": "port",
                               "sample": "the same operations once (the copies' time includes exporting each child to count its voxels)"}
        out["parity"] = {"split_child_voxels": [child_voxels, svox], "split_all_child_voxels": [all_voxels, svox] if t_all else None,
                         "copy_child_voxels": [copy_voxels, cvox],
                         "equal": bool(child_voxels == svox and copy_voxels == cvox and (not t_all or all_voxels == svox))}
    return out


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


def collide_benchmark(ctx, scale, o_big, m_big, reps=5):
    """SURVEY §8f item 1 on the headline body: collision probes of the meshed body (`ivx_collision_probes_recompute`), then the
    contacts between it and a half-size copy pushed into its side (`ivx_mutual_voxel_object_contacts`, results on the host)."""
    from impact_amd import capi, scenes
    from impact_amd.voxel import VoxelObjectMesh

    objs = []
    for sc in (scale, 0.5 * scale):
        _, obj = make_object(ctx, scenes.asteroid_scene(sc))
        r = obj.step(capi.STAGE_ALL)
        m32 = np.asarray(r["moments"]["m32"], dtype=np.float32).reshape(-1)
        com = (m32[1:4] * (np.float32(1.0) / m32[0])).astype(np.float32)  # derive_center_of_mass (object/inertia.rs:167-169)
        VoxelObjectMesh.create(obj)
        objs.append((obj, com))
    (a, ca), (b, cb) = objs
    qa, ta, qb, tb = collide_poses(ca, cb, scale)
    t_p, t_m, n_probes, contacts = [], [], 0, None
    for _ in range(reps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        n_probes = a.collision_probes_recompute()
        t1 = time.perf_counter()
        b.collision_probes_recompute()
        ctx.synchronize()
        t2 = time.perf_counter()
        contacts = a.mutual_contacts(qa, ta, ca, b, qb, tb, cb, 1, 2, 0, 1, capacity=1 << 20)
        t3 = time.perf_counter()
        t_p.append(t1 - t0)
        t_m.append(t3 - t2)
    a.close()
    b.close()
    out = {"workload": "probes of the headline body; contacts between it and a half-size copy pushed into its side",
           "probes_ms": round(1e3 * float(np.mean(t_p[1:])), 4), "mutual_ms": round(1e3 * float(np.mean(t_m[1:])), 4), "probes": n_probes,
           "contacts": int(len(contacts))}
    if o_big is not None:
        import oracle_lib as ol

        t6 = time.perf_counter()
        pa = o_big.collision_probes(m_big)
        t7 = time.perf_counter()
        ob = ol.OracleObject.from_sdf(scenes.asteroid_scene(0.5 * scale), 1.0, 0)
        ob.update_occupied_voxel_ranges()
        ob.compute_all_derived_state()
        pb = ob.collision_probes(ob.mesh())
        # (the same poses and centres of mass as the GPU call: they are inputs of the contact generation, not what is being compared)
        oca, ocb, oqa, ota, oqb, otb = ca, cb, qa, ta, qb, tb
        t8 = time.perf_counter()
        wi, opos, onrm, odep = o_big.mutual_contacts(pa, oca, oqa, ota, ob, pb, ocb, oqb, otb, cap=1 << 20)
        t9 = time.perf_counter()
        same = ((int(len(pa[0])), int(len(wi))) == (n_probes, int(len(contacts))) and bool(np.array_equal(opos.view(np.uint32), contacts["position"].view(np.uint32)))
                and bool(np.array_equal(onrm.view(np.uint32), contacts["normal"].view(np.uint32))) and bool(np.array_equal(odep.view(np.uint32), contacts["depth"].view(np.uint32))))
        out["cpu_baseline"] = {"probes_ms": 1e3 * (t7 - t6), "mutual_ms": 1e3 * (t9 - t8), "probes": int(len(pa[0])), "contacts": int(len(wi)), "cores": 1,
                               "kind": "port", "parity": "same probe count; contact positions, normals, depths bit-equal" if same else "MISMATCH"}
    return out


EDIT_OFFSET = np.array([110.0, 6.0, -4.0], dtype=np.float32)  # from the centre of the body, at scale 1: inside the tip of the +x bump
EDIT_RADIUS = 15.0


def edit_benchmark(ctx, scale, o_big, reps=5):
    """SURVEY §8f item 2 on the headline body: an absorbing sphere bites into the asteroid (`ivx_absorb_sphere`: the edit kernel,
    the derived-state + region refresh of the whole object, results back on the host), then the remesh the bite invalidates.
    Each repetition starts from the freshly generated body."""
    from impact_amd import capi, scenes
    from impact_amd.voxel import VoxelObjectMesh

    _, obj = make_object(ctx, scenes.asteroid_scene(scale))
    mesh = VoxelObjectMesh(obj)
    t_edit, t_remesh, t_sync, emptied, touched, invalidated = [], [], [], 0, 0, 0
    sdf_after = None
    for rep in range(reps + 1):
        obj.step(capi.STAGE_ALL)
        c = np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + EDIT_OFFSET * np.float32(scale)
        mesh.sync_with_voxel_object(np.zeros(obj.n_chunks, dtype=np.uint8))  # (the submesh bookkeeping of the fresh mesh, once per full remesh)
        ctx.synchronize()
        t0 = time.perf_counter()
        r = obj.absorb_sphere(c, EDIT_RADIUS * scale + 2.0, EDIT_RADIUS * scale, want_invalidated=True)
        t1 = time.perf_counter()
        mesh.sync_with_voxel_object(r["invalidated"])  # the incremental remesh of the invalidated chunks (mesh.rs:355-456) ...
        t2 = time.perf_counter()
        n_idx_sync, n_vtx_sync = int(mesh.counts["n_indices"]), int(mesh.counts["n_vertices"])
        obj.step(capi.STAGE_REMESH)  # ... and the full one, for comparison
        t3 = time.perf_counter()
        if rep == reps and o_big is not None:  # (a remesh changes neither voxels nor labels)
            sdf_after = obj.download(types=False, flags=False, labels=True, info=False)
        t_edit.append(t1 - t0)
        t_sync.append(t2 - t1)
        t_remesh.append(t3 - t2)
        emptied, touched, invalidated = r["emptied_voxels"], r["touched_chunks"], int(r["invalidated"].sum())
    # the same edit and sync with the sync enqueued while the edit is in flight (ivx_mesh_sync_enqueue(NULL): placed from the mesh needs the
    # edit's count role delivers early): edit enqueue -> sync enqueue -> edit collect -> sync collect
    t_over = []
    obj.set_early_mesh_needs(True)
    for rep in range(reps + 1):
        obj.step(capi.STAGE_ALL)
        c = np.array([0.5 * (a + b) for a, b in obj.update_occupied_voxel_ranges()], dtype=np.float32) + EDIT_OFFSET * np.float32(scale)
        mesh.sync_with_voxel_object(np.zeros(obj.n_chunks, dtype=np.uint8))
        ctx.synchronize()
        t0 = time.perf_counter()
        obj.absorb_sphere_enqueue(c, EDIT_RADIUS * scale + 2.0, EDIT_RADIUS * scale)
        mesh.sync_enqueue(None)
        r2 = obj.absorb_collect(want_invalidated=True)
        mesh.sync_collect()
        t_over.append(time.perf_counter() - t0)
    same_overlapped = (r2["emptied_voxels"], r2["touched_chunks"], int(r2["invalidated"].sum())) == (emptied, touched, invalidated) and \
        int(mesh.counts["n_indices"]) == n_idx_sync and int(mesh.counts["n_vertices"]) == n_vtx_sync
    obj.close()
    out = {"workload": f"absorbing sphere r={EDIT_RADIUS * scale:.1f} voxels at the surface of the headline body",
           "edit_and_sync_overlapped_ms": round(1e3 * float(np.mean(t_over[1:])), 4), "parity": {"overlapped_equals_sequential": bool(same_overlapped)},
           "edit_ms": round(1e3 * float(np.mean(t_edit[1:])), 4), "remesh_after_ms": round(1e3 * float(np.mean(t_remesh[1:])), 4),
           "sync_after_ms": round(1e3 * float(np.mean(t_sync[1:])), 4),
           "edit_plus_sync_ms": round(1e3 * float(np.mean(t_edit[1:]) + np.mean(t_sync[1:])), 4), "emptied_voxels": emptied, "touched_chunks": touched,
           "invalidated_chunks": invalidated}
    if o_big is not None:
        c = np.array([0.5 * (a + b_) for a, b_ in o_big.info()["occupied_voxel_ranges"]], dtype=np.float32) + EDIT_OFFSET * np.float32(scale)
        t4 = time.perf_counter()
        er = o_big.absorb_sphere(c, EDIT_RADIUS * scale + 2.0, EDIT_RADIUS * scale)
        t5 = time.perf_counter()
        o_sdf, _, _, o_lab, _ = o_big.export_dense()
        same = int(er["emptied_by_type"].sum()) == emptied and bool(np.array_equal(o_sdf, sdf_after[0])) and bool(np.array_equal(o_lab, sdf_after[3]))
        out["cpu_baseline"] = {"ms": 1e3 * (t5 - t4), "emptied_voxels": int(er["emptied_by_type"].sum()), "cores": 1, "kind": "port",
                               "parity": "same voxel bytes and chunk-local labels after the edit" if same else "MISMATCH"}
    return out


def fragments_benchmark(ctx, with_cpu, n_axis=5, frames=6, warm=2):
    """Many objects per frame (the reference's manager loops over every voxel object each frame: impact_voxel/src/lib.rs:729-733; fragments
    come into being together: interaction/fracturing.rs:1047-1189). The config-2 body (256^3) is cut into the Voronoi cells of a jittered
    n^3 lattice of fracture points (`ivx_copy_polyhedra`); then, per frame and per fragment: one absorbing-sphere edit, the incremental remesh
    of what it invalidated, the moments — once object by object (the single-object calls, a wait each) and once through the `_many` calls
    (one launch per chain position for ALL objects, csrc/many.hpp), on two identical sets of fragments; the oracle does the same per object
    on the host cores."""
    from impact_amd import capi, many, scenes
    from impact_amd import fracturing as fr
    from impact_amd.voxel import VoxelObjectMesh

    graph = scenes.asteroid_scene(1.0)
    dens = np.ones(256, dtype=np.float32)
    _, body = make_object(ctx, graph)
    body.step(capi.STAGE_ALL)
    cc = np.asarray(body.chunk_counts, dtype=np.float32) * 16.0
    rng = np.random.default_rng(11)
    ax = [(np.arange(n_axis) + 0.5) * (c / n_axis) for c in cc]
    pts = (np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3) + rng.uniform(-4.0, 4.0, (n_axis ** 3, 3))).astype(np.float32)
    sets, tets = fr.fragment_plane_sets(pts, np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32))

    def cut():
        res = body.copy_polyhedra([s_[2] for s_ in sets], [s_[1] for s_ in sets])
        objs = [child for rc, child, _ in res if rc == 1]
        for o_ in objs:
            o_.set_densities(dens)
        return objs, [k for k, (rc, _, _) in enumerate(res) if rc == 1]

    # the cut: the first one of the process pays what a first use pays (the body's compact planes written out for the clips to read, the
    # merged kernels' code objects, the allocator's first big block) — reported apart; `cut_ms` is the median of the three after it
    t0 = time.perf_counter()
    a_objs, kept = cut()
    ctx.synchronize()
    t_cut_first = time.perf_counter() - t0
    t_cuts = []
    for _ in range(3):
        for o_ in a_objs:
            o_.close()
        ctx.synchronize()
        t0 = time.perf_counter()
        a_objs, kept = cut()
        ctx.synchronize()
        t_cuts.append(time.perf_counter() - t0)
    t_cut = float(np.median(t_cuts))
    cut_spread = spread3([1e3 * t for t in t_cuts])
    b_objs, _ = cut()
    n = len(a_objs)
    stages0 = capi.STAGE_ALL & ~capi.STAGE_SAMPLE
    ctx.synchronize()
    t0 = time.perf_counter()
    ra = [o_.step(stages0) for o_ in a_objs]
    ctx.synchronize()
    t1 = time.perf_counter()
    rb = many.voxel_step_many(b_objs, stages0)
    ctx.synchronize()
    t2 = time.perf_counter()
    first_loop_ms, first_many_ms = 1e3 * (t1 - t0), 1e3 * (t2 - t1)
    a_mesh, b_mesh = [], []
    for o_, r_ in zip(a_objs, ra):
        m_ = VoxelObjectMesh(o_)
        m_.counts = r_["mesh"].copy()
        a_mesh.append(m_)
    for o_, r_ in zip(b_objs, rb):
        m_ = VoxelObjectMesh(o_)
        m_.counts = r_["mesh"].copy()
        b_mesh.append(m_)
    occ = [np.asarray(r_["occupied"], dtype=np.float32).reshape(-1)[6:].reshape(3, 2) for r_ in rb]
    zero = [np.zeros(o_.n_chunks, dtype=np.uint8) for o_ in a_objs]
    for m_, z in zip(a_mesh, zero):  # (the submesh bookkeeping of the fresh meshes, once per full remesh)
        m_.sync_with_voxel_object(z)
    many.mesh_sync_many(b_mesh, zero)

    def frame_edits(f):
        cs, rs = [], []
        for oc in occ:
            c = 0.5 * (oc[:, 0] + oc[:, 1])
            c[f % 3] = oc[f % 3, 1] - 1.0 - 2.0 * (f // 3)
            cs.append(c.astype(np.float32))
            rs.append(4.0 + (f % 3))
        return cs, rs

    t_loop, t_many = [], []
    for f in range(frames + warm):  # (the first frames grow each object's edit scratch and mesh buffers once: warm-up, for both loops)
        cs, rs = frame_edits(f)
        ctx.synchronize()
        t0 = time.perf_counter()
        for o_, m_, c, r in zip(a_objs, a_mesh, cs, rs):
            e_ = o_.absorb_sphere(c, r + 2.0, r, dens)
            m_.sync_with_voxel_object(e_["invalidated"])
            o_.step(capi.STAGE_INERTIA)
        t1 = time.perf_counter()
        eb = many.absorb_sphere_many(b_objs, cs, [r + 2.0 for r in rs], rs, dens)
        many.mesh_sync_many(b_mesh, [e_["invalidated"] for e_ in eb])
        mom = many.voxel_step_many(b_objs, capi.STAGE_INERTIA)
        t2 = time.perf_counter()
        if f >= warm:
            t_loop.append(t1 - t0)
            t_many.append(t2 - t1)
    # looped and batched objects hold the same bytes
    same = True
    for x, y, mx, my in zip(a_objs, b_objs, a_mesh, b_mesh):
        for u, v in zip(x.download(), y.download()):
            same = same and bool(np.array_equal(u, v))
        for u, v in zip(mx.download(), my.download()):
            same = same and bool(np.array_equal(np.ascontiguousarray(u).view(np.uint8), np.ascontiguousarray(v).view(np.uint8)))
    loop_ms, many_ms = 1e3 * float(np.mean(t_loop)), 1e3 * float(np.mean(t_many))
    out = {"workload": f"config-2 body (256^3) cut into the Voronoi cells of a jittered {n_axis}^3 lattice: {n} fragments of "
                       f"{int(np.median([o_.n_chunks for o_ in a_objs]))} chunks (median); per frame and fragment: one absorbing sphere (r 4-6 voxels), the "
                       "incremental remesh of what it invalidated, the ten moments",
           "objects": n, "frames": frames, "warmup_frames": warm, "cut_ms": round(1e3 * t_cut, 3), "cut_ms_spread": cut_spread, "cut_first_ms": round(1e3 * t_cut_first, 3),
           "cut_what": "ivx_copy_polyhedra (children's grids from one block; clip, derived state, regions, occupied ranges and mass of all recorded and issued as one "
                       "launch per chain position) + the children's wrappers and density tables; median of 3 cuts of the same body, the process's first apart",
           "first_step_looped_ms": round(first_loop_ms, 3), "first_step_many_ms": round(first_many_ms, 3),
           "frame_looped_ms": round(loop_ms, 4), "frame_many_ms": round(many_ms, 4), "speedup": round(loop_ms / many_ms, 2),
           "objects_per_s_looped": n / (loop_ms * 1e-3), "objects_per_s_many": n / (many_ms * 1e-3),
           "parity": {"looped_equals_many": same}}
    if with_cpu:
        import oracle_lib as ol
        import parity_util as pu

        o_body = ol.OracleObject.from_sdf(graph, 1.0, 0)
        o_body.update_occupied_voxel_ranges()
        o_body.compute_all_derived_state()
        o_objs = []
        for k in kept:
            rc, co, _ = o_body.clip_polyhedron(sets[k][1], sets[k][2], copy=True)
            assert rc == 1
            co.update_occupied_voxel_ranges()
            o_objs.append(co)
        o_mesh = [ol.OracleMeshHandle(o_) for o_ in o_objs]
        t_cpu = []
        for f in range(frames + warm):
            cs, rs = frame_edits(f)
            t0 = time.perf_counter()
            for o_, m_, c, r in zip(o_objs, o_mesh, cs, rs):
                e_ = o_.absorb_sphere(c, r + 2.0, r, dens)
                m_.sync(e_["invalidated"])
                o_.inertia(dens)
            t_cpu.append(time.perf_counter() - t0)
        equal = True
        for o_, g_, om, gm in zip(o_objs, b_objs, o_mesh, b_mesh):
            o_sdf, o_typ, o_flg, o_lab, _ = o_.export_dense()
            g_sdf, g_typ, g_flg, g_lab, _ = g_.download()
            ne = (o_flg & 1) == 0
            equal = equal and bool(np.array_equal(o_sdf, g_sdf)) and bool(np.array_equal(o_lab, g_lab)) and bool(np.array_equal(o_flg[ne], g_flg[ne]))
            want = om.get()
            pos, nrm, idx, im, sub = gm.download()
            equal = equal and len(sub) == len(want.submeshes) and bool(np.array_equal(sub["index_offset"], want.submeshes[:, 3]))
            for sm in want.submeshes:
                io, ic, vo, vc = int(sm[3]), int(sm[4]), int(sm[13]), int(sm[14])
                equal = equal and bool(np.array_equal(idx[io:io + ic], want.indices[io:io + ic])) and bool(
                    np.array_equal(pos[vo:vo + vc].view(np.uint32), want.positions[vo:vo + vc].view(np.uint32)))
        out["parity"]["every_object_equals_the_oracle"] = equal
        cpu_ms = 1e3 * float(np.mean(t_cpu[warm:]))
        out["cpu_baseline"] = {"frame_ms": cpu_ms, "objects_per_s": n / (cpu_ms * 1e-3), "cores": 1, "kind": "port",
                               "sample": f"the same {frames} frames over the same {n} fragments, object by object, single thread"}
        # the cut itself on the host cores, fragments in parallel as FracturingProcess::execute_in_parallel runs them (fracturing.rs:1047-1189)
        from concurrent.futures import ThreadPoolExecutor

        workers = 1
        try:
            workers = max(1, min(16, len(os.sched_getaffinity(0))))
        except (AttributeError, OSError):
            pass

        def one_cut(k):
            return o_body.clip_polyhedron(sets[k][1], sets[k][2], copy=True)[0]

        t0 = time.perf_counter()
        with ThreadPoolExecutor(workers) as ex:
            rcs = list(ex.map(one_cut, range(len(sets))))
        cut_cpu = time.perf_counter() - t0
        out["cpu_baseline_cut"] = {"cut_ms": round(1e3 * cut_cpu, 2), "cores": workers, "kind": "port", "fragments": int(sum(1 for r_ in rcs if r_ == 1)),
                                   "sample": f"the same {len(sets)} polyhedron copies of the same body, in parallel on {workers} host threads"}
    for o_ in a_objs + b_objs:
        o_.close()
    tets.close()
    body.close()
    return out


def fragments_probes_and_pairs(ctx, a_objs, b_objs, a_mesh, b_mesh, occ, dens, rounds=6):
    """The two other per-object loops of a frame (collidable.rs:394-433 probe sync behind every mesh sync; collidable.rs:859-1049 mutual contacts
    of every pair the broad phase found) on the fragments of `fragments_frame`: object by object (`a`) and through `ivx_collision_probes_sync_many`
    / `ivx_mutual_voxel_object_contacts_many` (`b`, the twins in the same state). Pairs: fragment k against fragment k + 1, B pushed against A's
    high-x side two voxels deep (each in its own frame). Timed apart from `ms_looped` / `ms_batched`."""
    from impact_amd import many

    n = len(a_objs)
    for o_ in a_objs + b_objs:
        o_.collision_probes_recompute()
    t_sync = np.zeros(2)
    sync_batched, pairs_batched = [], []  # (per round: the host-bound legs report min / median / max beside the mean)
    for f in range(rounds):
        cs, rs = [], []
        for oc in occ:
            c = 0.5 * (oc[:, 0] + oc[:, 1])
            c[0] = oc[0, 1] - 1.0 - 1.5 * f  # bites into the high-x side, where the neighbour touches
            cs.append(c.astype(np.float32))
            rs.append(3.0 + (f % 2))
        inv_a = []
        for o_, m_, c, r in zip(a_objs, a_mesh, cs, rs):
            e_ = o_.absorb_sphere(c, r + 2.0, r, dens)
            m_.sync_with_voxel_object(e_["invalidated"])
            inv_a.append(e_["invalidated"])
        eb = many.absorb_sphere_many(b_objs, cs, [r + 2.0 for r in rs], rs, dens)
        inv_b = [e_["invalidated"] for e_ in eb]
        many.mesh_sync_many(b_mesh, inv_b)
        ctx.synchronize()
        t0 = time.perf_counter()
        for o_, iv in zip(a_objs, inv_a):
            o_.collision_probes_sync(iv)
        t1 = time.perf_counter()
        many.collision_probes_sync_many(b_objs, inv_b)
        t2 = time.perf_counter()
        if f:
            t_sync += (t1 - t0, t2 - t1)
            sync_batched.append(1e3 * (t2 - t1))
    same_probes = True
    for x, y in zip(a_objs, b_objs):
        (pa, ea), (pb, eb_) = x.collision_probes(), y.collision_probes()
        same_probes = same_probes and np.array_equal(ea, eb_) and all(np.array_equal(pa[e[3]:e[4]].view(np.uint32), pb[e[3]:e[4]].view(np.uint32)) for e in ea)
    ident = np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)
    zero3 = np.zeros(3, dtype=np.float32)
    resp = (0.2, 0.7, 0.5)
    pairs_a, pairs_b, args = [], [], []
    for k in range(n - 1):
        ca, cb = 0.5 * (occ[k][:, 0] + occ[k][:, 1]), 0.5 * (occ[k + 1][:, 0] + occ[k + 1][:, 1])
        tb = np.array([occ[k + 1][0, 0] - occ[k][0, 1] + 6.0, cb[1] - ca[1], cb[2] - ca[2]], dtype=np.float32)  # world (= A's frame) -> B's frame
        args.append((ident, zero3, ca.astype(np.float32), ident, tb, cb.astype(np.float32), 500 + k, 501 + k, k, k + 1, resp))
        for objs, pairs in ((a_objs, pairs_a), (b_objs, pairs_b)):
            pairs.append(dict(a=objs[k], b=objs[k + 1], rotation_a=ident, translation_a=zero3, center_of_mass_a=ca, rotation_b=ident, translation_b=tb,
                              center_of_mass_b=cb, collidable_id_a=500 + k, collidable_id_b=501 + k, body_a=k, body_b=k + 1, response=resp))
    qb = many.mutual_queries(pairs_b)
    t_pairs = np.zeros(2)
    same_pairs, n_contacts = True, 0
    for f in range(rounds):
        ctx.synchronize()
        t0 = time.perf_counter()
        lists = [a_objs[k].mutual_contacts(g[0], g[1], g[2], a_objs[k + 1], *g[3:], capacity=8192) for k, g in enumerate(args)]
        t1 = time.perf_counter()
        got, off = many.mutual_voxel_object_contacts_many(qb)
        t2 = time.perf_counter()
        if f:
            t_pairs += (t1 - t0, t2 - t1)
            pairs_batched.append(1e3 * (t2 - t1))
        want = np.concatenate(lists) if lists else got[:0]
        same_pairs = same_pairs and want.tobytes() == got.tobytes()
        n_contacts = len(got)
    d = max(rounds - 1, 1)
    return {"probes_sync": {"ms_looped": round(1e3 * t_sync[0] / d, 4), "ms_batched": round(1e3 * t_sync[1] / d, 4),
                            "ms_batched_spread": spread3(sync_batched) if sync_batched else None},
            "mutual_pairs": {"pairs": n - 1, "contacts": int(n_contacts), "ms_looped": round(1e3 * t_pairs[0] / d, 4), "ms_batched": round(1e3 * t_pairs[1] / d, 4),
                             "ms_batched_spread": spread3(pairs_batched) if pairs_batched else None},
            "parity": {"probes_looped_equal_batched": bool(same_probes), "pairs_looped_equal_batched": bool(same_pairs)}}


def fragments_frame_benchmark(ctx, with_cpu, n_axis=5, frames=6, warm=2):
    """A FRAME of many objects with the rigid-body side in it (engine/src/tasks.rs:376-434 in the reference's order): the fragments of the
    `fragments` leg, each a rigid body lying on a ground plane — per frame: contact generation of every fragment against the plane (f1,
    collidable.rs:1176-1208) -> this frame's contacts into the solver (ivx_world_set_contacts) -> solve + integrate -> one absorbing-sphere edit
    per fragment -> incremental remesh -> moments. Once object by object (the single-object calls, a wait or two each) and once through the
    many-object calls (`ivx_voxel_object_contacts_many`, `ivx_absorb_sphere_many`, `ivx_mesh_sync_many`, `ivx_voxel_step_many`); the oracle
    does the same frame on one host thread. The bodies' poses are not fed back into the voxel objects' transforms (every frame finds the same
    geometry: the figure is the cost of the frame's calls, not a simulation)."""
    from impact_amd import capi, many, scenes
    from impact_amd import fracturing as fr
    from impact_amd.physics import PhysicsWorld, uniform_sphere_body
    from impact_amd.voxel import VoxelObjectMesh

    graph = scenes.asteroid_scene(1.0)
    dens = np.ones(256, dtype=np.float32)
    _, body = make_object(ctx, graph)
    body.step(capi.STAGE_ALL)
    cc = np.asarray(body.chunk_counts, dtype=np.float32) * 16.0
    rng = np.random.default_rng(11)
    ax = [(np.arange(n_axis) + 0.5) * (c / n_axis) for c in cc]
    pts = (np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3) + rng.uniform(-4.0, 4.0, (n_axis ** 3, 3))).astype(np.float32)
    sets, tets = fr.fragment_plane_sets(pts, np.array([0, 0, 0, cc[0], cc[1], cc[2]], dtype=np.float32))

    def cut():
        res = body.copy_polyhedra([s_[2] for s_ in sets], [s_[1] for s_ in sets])
        objs = [child for rc, child, _ in res if rc == 1]
        for o_ in objs:
            o_.set_densities(dens)
        return objs, [k for k, (rc, _, _) in enumerate(res) if rc == 1]

    a_objs, kept = cut()
    b_objs, _ = cut()
    n = len(a_objs)
    stages0 = capi.STAGE_ALL & ~capi.STAGE_SAMPLE
    rb = many.voxel_step_many(b_objs, stages0)
    ra = many.voxel_step_many(a_objs, stages0)
    a_mesh, b_mesh = [], []
    for objs, rs_, meshes in ((a_objs, ra, a_mesh), (b_objs, rb, b_mesh)):
        for o_, r_ in zip(objs, rs_):
            m_ = VoxelObjectMesh(o_)
            m_.counts = r_["mesh"].copy()
            meshes.append(m_)
    zero = [np.zeros(o_.n_chunks, dtype=np.uint8) for o_ in a_objs]
    many.mesh_sync_many(a_mesh, zero)
    many.mesh_sync_many(b_mesh, zero)
    occ = [np.asarray(r_["occupied"], dtype=np.float32).reshape(-1)[6:].reshape(3, 2) for r_ in rb]
    # every fragment a dynamic body (a sphere's inertia of its size: the solver does not care), the ground a kinematic one
    bodies = np.array([uniform_sphere_body(8.0, 1.0, (20.0 * (k % 9), 0.0, 20.0 * (k // 9))) for k in range(n)])
    bodies["total_force"][:, 1] = np.float32(-9.81) * bodies["mass"]
    ground = np.zeros(1, dtype=capi.KINEMATIC_BODY_DTYPE)
    ground["orientation"] = (0, 0, 0, 1)
    ground["angular_axis"] = (0, 1, 0)
    ident = np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)
    zero3 = np.zeros(3, dtype=np.float32)
    up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
    resp = (0.2, 0.7, 0.5)
    disp = [float(oc[1, 0] + 3.0) for oc in occ]  # a ground plane three voxels into each fragment's underside (its own frame)
    q = many.collidable_queries(n)
    for k in range(n):
        q[k]["mode"], q[k]["shape3"], q[k]["shape1"], q[k]["response"] = 1, up, disp[k], resp
        q[k]["collidable_id_a"], q[k]["collidable_id_b"], q[k]["body_a"], q[k]["body_b"] = 1000 + k, 7, k, 0x80000000
    wa, wb = PhysicsWorld(ctx), PhysicsWorld(ctx)
    for w in (wa, wb):
        w.set_bodies(bodies, ground)

    def frame_edits(f):
        cs, rs = [], []
        for oc in occ:
            c = 0.5 * (oc[:, 0] + oc[:, 1])
            c[1] = oc[1, 1] - 1.0 - 2.0 * f
            cs.append(c.astype(np.float32))
            rs.append(4.0 + (f % 3))
        return cs, rs

    t_loop, t_many = [], []
    parts = np.zeros(6)
    part_samples = []
    n_contacts = 0
    same_contacts = True
    for f in range(frames + warm):
        cs, rs = frame_edits(f)
        ctx.synchronize()
        t0 = time.perf_counter()
        lists = [o_.plane_contacts(ident, zero3, up, disp[k], 1000 + k, 7, k, 0x80000000, resp, capacity=4096) for k, o_ in enumerate(a_objs)]
        ca = np.concatenate(lists) if lists else np.zeros(0, dtype=capi.CONTACT_DTYPE)
        wa.prepare_constraints(ca)
        wa.step_enqueue(0.005)
        for o_, m_, c, r in zip(a_objs, a_mesh, cs, rs):
            e_ = o_.absorb_sphere(c, r + 2.0, r, dens)
            m_.sync_with_voxel_object(e_["invalidated"])
            o_.step(capi.STAGE_INERTIA)
        ctx.synchronize()
        t1 = time.perf_counter()
        cb, off = many.voxel_object_contacts_many(b_objs, q)
        ta = time.perf_counter()
        wb.prepare_constraints(cb)
        tb = time.perf_counter()
        wb.step_enqueue(0.005)
        tc = time.perf_counter()
        eb = many.absorb_sphere_many(b_objs, cs, [r + 2.0 for r in rs], rs, dens)
        td = time.perf_counter()
        many.mesh_sync_many(b_mesh, [e_["invalidated"] for e_ in eb])
        te = time.perf_counter()
        many.voxel_step_many(b_objs, capi.STAGE_INERTIA)
        ctx.synchronize()
        t2 = time.perf_counter()
        if f >= warm:
            parts += (ta - t1, tb - ta, tc - tb, td - tc, te - td, t2 - te)
            part_samples.append((1e3 * (ta - t1), 1e3 * (tb - ta)))
        same_contacts = same_contacts and ca.tobytes() == cb.tobytes()
        n_contacts = len(cb)
        if f >= warm:
            t_loop.append(t1 - t0)
            t_many.append(t2 - t1)
    same = same_contacts
    for x, y in zip(a_objs, b_objs):
        for u, v in zip(x.download(), y.download()):
            same = same and bool(np.array_equal(u, v))
    da, db = wa.bodies()[0], wb.bodies()[0]
    same_bodies = all(np.array_equal(da[f_].view(np.uint32), db[f_].view(np.uint32)) for f_ in ("position", "orientation", "momentum", "angular_momentum"))
    loop_ms, many_ms = 1e3 * float(np.mean(t_loop)), 1e3 * float(np.mean(t_many))
    out = {"workload": f"{n} fragments (the `fragments` leg's) as rigid bodies on a ground plane; per frame: plane contacts of every fragment (f1) -> "
                       "ivx_world_set_contacts -> solve + integrate -> an absorbing sphere per fragment -> incremental remesh -> moments",
           "objects": n, "contacts_per_frame": n_contacts, "frames": frames, "ms_looped": round(loop_ms, 4), "ms_batched": round(many_ms, 4),
           "speedup": round(loop_ms / many_ms, 2),
           "batched_host_ms": {k_: round(1e3 * float(v_) / frames, 4) for k_, v_ in zip(("contacts_many", "set_contacts", "step_enqueue", "absorb_many (waits for the solve too)",
                                                                                        "mesh_sync_many", "moments_many"), parts)},
           "batched_host_ms_spread": {"contacts_many": spread3([p_[0] for p_ in part_samples]), "set_contacts": spread3([p_[1] for p_ in part_samples]),
                                      "frame": spread3([1e3 * t_ for t_ in t_many])},
           "parity": {"looped_equals_batched": bool(same), "bodies_equal": bool(same_bodies)}}
    # the solve of this contact set by itself, on the kernel the schedule picks and on the chain-stationary one
    wb.prepare_constraints(cb)
    r_ = wb.step(0.005)
    out["solver"] = dict(wb.solver_info(), solve_ms=round(float(r_["stage_ms"][2]), 4))
    wb.set_solver_groups(255)
    wb.prepare_constraints(cb)
    r_ = wb.step(0.005)
    out["solver_chain_stationary"] = {"kernel": wb.solver_info()["kernel"], "solve_ms": round(float(r_["stage_ms"][2]), 4)}
    wb.set_solver_groups(0)
    if with_cpu:
        import oracle_lib as ol
        from test_gpu_contacts import oracle_plane_contact_list

        o_body = ol.OracleObject.from_sdf(graph, 1.0, 0)
        o_body.update_occupied_voxel_ranges()
        o_body.compute_all_derived_state()
        o_objs = []
        for k in kept:
            rc, co, _ = o_body.clip_polyhedron(sets[k][1], sets[k][2], copy=True)
            co.update_occupied_voxel_ranges()
            o_objs.append(co)
        o_mesh = [ol.OracleMeshHandle(o_) for o_ in o_objs]
        op = ol.OraclePhysics(bodies, ground, config=(8, 0.4, 3, 0.2))
        t_cpu, equal = [], True
        for f in range(frames + warm):
            cs, rs = frame_edits(f)
            t0 = time.perf_counter()
            lists = [oracle_plane_contact_list(o_, ident, zero3, up, disp[k], 1000 + k, 7, k, 0x80000000, resp) for k, o_ in enumerate(o_objs)]
            co_ = np.concatenate(lists)
            op.step(co_, 0.005)
            for o_, m_, c, r in zip(o_objs, o_mesh, cs, rs):
                e_ = o_.absorb_sphere(c, r + 2.0, r, dens)
                m_.sync(e_["invalidated"])
                o_.inertia(dens)
            t_cpu.append(time.perf_counter() - t0)
            if f == frames + warm - 1:
                equal = co_.tobytes() == cb.tobytes()
        od = op.bodies()[0]
        rel = 0.0
        for f_ in ("position", "orientation", "momentum", "angular_momentum"):
            g64, o64 = db[f_].astype(np.float64), od[f_].astype(np.float64)
            scale = np.maximum(np.linalg.norm(o64, axis=1, keepdims=True), max(float(np.abs(o64).max()), 1e-30) * 1e-4)
            rel = max(rel, float((np.abs(g64 - o64) / scale).max()))
        for o_, g_ in zip(o_objs, b_objs):
            o_sdf, o_typ, o_flg, o_lab, _ = o_.export_dense()
            g_sdf, g_typ, g_flg, g_lab, _ = g_.download()
            equal = equal and bool(np.array_equal(o_sdf, g_sdf)) and bool(np.array_equal(o_lab, g_lab))
        out["parity"]["last_frame_contacts_and_voxels_equal_the_oracle"] = bool(equal)
        out["parity"]["body_state_max_rel_err"] = rel
        out["parity"]["equal"] = bool(same and same_bodies and equal and rel <= 1e-5)
        cpu_ms = 1e3 * float(np.mean(t_cpu[warm:]))
        out["cpu_baseline"] = {"frame_ms": round(cpu_ms, 2), "cores": 1, "kind": "port",
                               "sample": f"the same {frames} frames (contacts built into records by a Python loop over the oracle's hits: part of the time), single thread"}
    # The batched frame once more with the rigid bodies' world on a context of its own (its own stream): the solve — six workgroups for these
    # 81 bodies, 0.65 ms of dependent levels — runs beside the fragments' edits, re-mesh and moments instead of ahead of them (the edits read
    # nothing of it). Behind the oracle's comparison: further frames on the batched set.
    from impact_amd.voxel import Context

    ctx2 = Context(ctx.device)
    wc = PhysicsWorld(ctx2)
    wc.set_bodies(bodies, ground)
    t_two = []
    for f in range(frames + warm, 2 * (frames + warm)):
        cs, rs = frame_edits(f % (frames + warm))
        for o_, m_, c, r in zip(a_objs, a_mesh, cs, rs):  # (the twins keep step, untimed: the leg behind this one compares the two sets)
            e_ = o_.absorb_sphere(c, r + 2.0, r, dens)
            m_.sync_with_voxel_object(e_["invalidated"])
            o_.step(capi.STAGE_INERTIA)
        ctx.synchronize()
        ctx2.synchronize()
        t1 = time.perf_counter()
        cb2, _ = many.voxel_object_contacts_many(b_objs, q)
        wc.prepare_constraints(cb2)
        wc.step_enqueue(0.005)
        eb = many.absorb_sphere_many(b_objs, cs, [r + 2.0 for r in rs], rs, dens)
        many.mesh_sync_many(b_mesh, [e_["invalidated"] for e_ in eb])
        many.voxel_step_many(b_objs, capi.STAGE_INERTIA)
        ctx.synchronize()
        ctx2.synchronize()
        if f >= frames + 2 * warm:
            t_two.append(time.perf_counter() - t1)
    out["ms_batched_world_on_own_context"] = round(1e3 * float(np.mean(t_two)), 4)
    wc.close()
    ctx2.close()
    # (behind the oracle's comparison: this leg edits the fragments further)
    out["probes_and_pairs"] = fragments_probes_and_pairs(ctx, a_objs, b_objs, a_mesh, b_mesh, occ, dens)
    for w in (wa, wb):
        w.close()
    for o_ in a_objs + b_objs:
        o_.close()
    tets.close()
    body.close()
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
    host_warm = []
    for _ in range(8):  # the per-frame case: the same contact ids as last frame (cache hits, buffers sized); the first such call sizes the staging buffer
        t0 = time.perf_counter()
        w.prepare_constraints(contacts)
        host_warm.append(1e3 * (time.perf_counter() - t0))
        w.step(0.005)
    host_warm_ms = float(np.median(host_warm[1:]))
    acc = np.zeros(5)
    t0 = time.perf_counter()
    for _ in range(steps):
        acc += w.step(0.005)["stage_ms"]
    wall = (time.perf_counter() - t0) / steps
    r = w.step(0.005)
    n_steps_done = 1 + len(host_warm) + steps + 1
    sweeps = 1 + 8 + 3
    out = {
        "workload": "16^3 lattice of unit-density spheres r=0.5 at spacing 0.95: 4096 bodies, 46080 contacts, dt 0.005, 8+3 sweeps + warm start",
        "ms_per_step": 1e3 * wall,
        "contact_sweeps_per_s": len(contacts) * sweeps / wall,
        "levels": [int(r["n_levels"][0]), int(r["n_levels"][1])],
        "stage_ms": {k: round(float(v) / steps, 4) for k, v in zip(capi.PHYSICS_STAGE_NAMES, acc)},
        "set_contacts_next_frame_host_ms": round(host_warm_ms, 3), "set_contacts_next_frame_host_ms_spread": spread3(host_warm[1:]),
        "set_contacts_host_ms": round(host_ms, 3), "set_contacts_host_ms_what": "the world's first call: buffers allocated, every id new",
    }
    info = w.solver_info() if hasattr(w, "solver_info") else None
    if info:
        out["solver"] = info
    if with_cpu:
        import oracle_lib as ol

        o = ol.OraclePhysics(bodies, config=(8, 0.4, 3, 0.2))
        t0 = time.perf_counter()
        for _ in range(n_steps_done):
            o.step(contacts, 0.005)
        cpu = (time.perf_counter() - t0) / n_steps_done
        gd, od = w.bodies()[0], o.bodies()[0]
        rel = 0.0
        for f in ("position", "orientation", "momentum", "angular_momentum"):
            g64, o64 = gd[f].astype(np.float64), od[f].astype(np.float64)
            scale = np.maximum(np.linalg.norm(o64, axis=1, keepdims=True), max(float(np.abs(o64).max()), 1e-30) * 1e-4)
            rel = max(rel, float((np.abs(g64 - o64) / scale).max()))
        out["cpu_baseline"] = {"value": len(contacts) * sweeps / cpu, "unit": "contact sweeps/s", "cores": 1, "kind": "port",
                               "sample": f"same pile, {n_steps_done} steps, {1e3 * cpu:.1f} ms/step, single thread"}
        out["parity"] = {"steps": n_steps_done, "body_state_max_rel_err": rel, "equal": bool(rel <= 1e-5)}
    return out, w


def pile_churn_benchmark(ctx, with_cpu, frames=60):
    """The config-4 pile with a contact set that CHANGES every frame (scenes.pile_churn_frames: a tenth of the manifolds leave, the tenth that
    left the frame before returns): `ivx_world_set_contacts` takes its general path — ids looked up, slots swap-removed and appended in the
    reference's order, the dependency schedule rebuilt — and the solve runs on the order that history leaves (more levels than the pristine
    lattice's). Timed per frame: set_contacts + step together, host work included; the oracle steps through the same lists beside it."""
    from impact_amd import scenes
    from impact_amd.physics import PhysicsWorld

    bodies, contacts = scenes.sphere_pile_scene(16)
    lists = scenes.pile_churn_frames(contacts, frames + 4)
    w = PhysicsWorld(ctx)
    w.set_bodies(bodies)
    w.prepare_constraints(contacts)
    w.step(0.005)
    t_set, t_frame, levels = [], [], []
    for f, cs in enumerate(lists):
        ctx.synchronize()
        t0 = time.perf_counter()
        w.prepare_constraints(cs)
        t1 = time.perf_counter()
        r = w.step(0.005)
        t2 = time.perf_counter()
        if f >= 4:  # (the first frames size the buffers)
            t_set.append(1e3 * (t1 - t0))
            t_frame.append(1e3 * (t2 - t0))
            levels.append(int(r["n_levels"][0]) + int(r["n_levels"][1]))
    out = {"workload": f"the config-4 pile, {frames} frames: each frame a tenth of the 11 520 manifolds is missing, the tenth missing the frame before is back "
                       "(41 472 of 46 080 contacts per frame); ivx_world_set_contacts (general path) + ivx_world_step per frame, one wait",
           "ms_per_frame": round(float(np.median(t_frame)), 4), "ms_per_frame_spread": spread3(t_frame),
           "set_contacts_host_ms": round(float(np.median(t_set)), 4), "set_contacts_host_ms_spread": spread3(t_set),
           "levels_velocity_plus_positional": {"min": int(min(levels)), "median": int(np.median(levels)), "max": int(max(levels)), "pristine_order": 183 + 147},
           "solver": w.solver_info()}
    if with_cpu:
        import oracle_lib as ol

        o = ol.OraclePhysics(bodies, config=(8, 0.4, 3, 0.2))
        o.step(contacts, 0.005)
        t0 = time.perf_counter()
        for cs in lists:
            o.step(cs, 0.005)
        cpu = (time.perf_counter() - t0) / len(lists)
        gd, od = w.bodies()[0], o.bodies()[0]
        rel = 0.0
        for f_ in ("position", "orientation", "momentum", "angular_momentum"):
            g64, o64 = gd[f_].astype(np.float64), od[f_].astype(np.float64)
            scale = np.maximum(np.linalg.norm(o64, axis=1, keepdims=True), max(float(np.abs(o64).max()), 1e-30) * 1e-4)
            rel = max(rel, float((np.abs(g64 - o64) / scale).max()))
        out["cpu_baseline"] = {"ms_per_frame": round(1e3 * cpu, 2), "cores": 1, "kind": "port", "sample": f"the same {len(lists)} frames, single thread"}
        out["parity"] = {"frames": len(lists) + 1, "body_state_max_rel_err": rel, "equal": bool(rel <= 1e-5)}
    w.close()
    return out


def spread3(samples):
    """min / median / max of a host-bound leg's per-call times (ms): the driver's run and the builder's differ by more than a median shows"""
    a = np.asarray(samples, dtype=np.float64)
    return {"min": round(float(a.min()), 4), "median": round(float(np.median(a)), 4), "max": round(float(a.max()), 4), "n": int(a.size)}


def main():
    # Python's cyclic collector stays off while the bench runs (as `timeit` keeps it): with torch imported a full collection walks ~170 000
    # objects in 36 ms, and one of them landed in the all-surface leg's 50 steps on every run (+0.8 ms per step on paper). Collections are
    # made by hand between the timed regions (time_steps, and in front of every leg below).
    gc.disable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (a step is a third of a millisecond: 200 of them are 0.06 s, and short runs measure the clocks ramping up — 0.312 ms per step
    # over 20 steps, 0.302 over 200 on the same box)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scale", type=float, default=2.05, help="asteroid scale (2.05 -> 512^3 stored grid)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = the same 512^3 grid in N x-slabs (the metric's configuration); weak = BASELINE config 5 (lengths x N^(1/3))")
    ap.add_argument("--workload", choices=("asteroid", "dense"), default="asteroid", help="the timed step's scene (dense = all-surface plates)")
    ap.add_argument("--dense-chunks", type=int, default=32)
    ap.add_argument("--profiles", default=None, help="directory of the committed PMC summaries `roofline.traffic` is quoted from (default: the newest profiles/round* that has them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pile", action="store_true", help="skip the separate legs (dense, config2, config3, pile, frame, edit, collide)")
    ap.add_argument("--plain", action="store_true", help="profiling runs: nothing but full steps (no remesh-only timing pass), so that every kernel launch rocprofv3 sees belongs to a step")
    args = ap.parse_args()
    if args.profiles:
        global PROFILE_DIR
        PROFILE_DIR = os.path.abspath(args.profiles)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch

    from impact_amd import capi, scenes
    from impact_amd.voxel import Context

    # one rank per GPU; on a box with fewer GPUs than ranks (single-GPU protocol check with the gloo backend) ranks share devices
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
        # the backend is a launch-time choice, the same on every rank; an RCCL failure ends the job (no per-rank fallback:
        # ranks on different backends would deadlock, and a host-staged number must not pass for an RCCL one)
        backend = os.environ.get("IVX_BENCH_BACKEND", "nccl")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            else:
                dist.init_process_group(backend)
        except Exception as e:
            print(f"[bench] rank {rank}: init_process_group({backend!r}) failed: {e}", file=sys.stderr)
            sys.exit(3)

    dens = np.ones(256, dtype=np.float32)
    # the voxel object's own rigid body (setup_dynamic_rigid_body_for_voxel_object, impact_voxel/src/setup.rs:581-616):
    # integrated every step; its contact list is empty in this workload
    from impact_amd.physics import PhysicsWorld, uniform_sphere_body

    body_world = PhysicsWorld(ctx)
    body_world.set_bodies(np.array([uniform_sphere_body(100.0, 1.0, (0.0, 0.0, 0.0), (0.1, 0.0, 0.0))]))
    body_world.prepare_constraints(np.zeros(0, dtype=capi.CONTACT_DTYPE))
    graph = scenes.asteroid_scene(args.scale) if args.workload == "asteroid" else scenes.plates_scene(args.dense_chunks)
    scene_name = f"config-2 SDF asteroid x{args.scale}" if args.workload == "asteroid" else f"{args.dense_chunks} perforated plates (all-surface)"
    scaling = "strong"
    if not slabs:
        gen, obj = make_object(ctx, graph)
        cc = gen.chunk_counts()
        # The step samples the same resident program every time: its interval pre-pass — which reads nothing but the program — is enqueued
        # a step ahead, on the context's second stream behind the evaluator of the step before (ivx_grid_set_sample_ahead). All of a step's
        # work is still done once per step; back-to-back steps just do not wait for it. `isolated` below is the same step without that.
        sample_ahead = os.environ.get("IVX_BENCH_SAMPLE_AHEAD", "1") != "0"
        obj.set_sample_ahead(sample_ahead)

        def step():
            # the voxel stages and the object's rigid-body step are enqueued back to back; one wait covers both
            obj.step_enqueue(capi.STAGE_ALL)
            body_world.step_enqueue(0.005)
            return obj.step_collect()

        workload = f"{scene_name} -> {gen.grid_shape()[0]}^3 grid = {cc[0] * 16}^3 stored voxels ({obj.n_chunks} chunks)"
        parallelism = "single GPU"
    else:
        from impact_amd.distributed import NativeComm, NativeSlabStepper, NativeStepGroup, SlabStepper, TorchComm

        native_comm = None
        if dist.get_backend() == "nccl":
            # the per-step protocol runs inside the library (ivx_slabs_step_*: grouped ncclSend / ncclRecv + one ncclAllGather on the
            # library's stream); this script only hands the communicator's unique id from rank 0 to the others
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(NativeComm.unique_id()), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(uid, 0)
            native_comm = NativeComm(ctx, world, rank, bytes(uid.cpu().numpy().tobytes()))

        def slab_leg(mode):
            """(graph, scene name, stepper, voxel object, step function, transport) of the x-slab decomposition: `strong` = the SAME 512^3 grid of the
            metric on every N; `weak` = BASELINE.json's config 5, the config-2 asteroid with all lengths scaled so that every rank keeps the 512^3
            workload's voxel count (N = 8: scale x2 again -> the 1024^3 grid in 8 slabs of 128 planes)"""
            g, nm = graph, scene_name
            if mode == "weak" and args.workload == "asteroid":
                g = scenes.asteroid_scene(args.scale * world ** (1.0 / 3.0))
                nm = f"config-2 SDF asteroid x{args.scale * world ** (1.0 / 3.0):.3f} (config 5 at N=8)"
            if native_comm is not None:
                st = NativeSlabStepper(ctx, native_comm, g, dens, rank, sample_ahead=AHEAD)
                group = NativeStepGroup([st])

                def fn():
                    body_world.step_enqueue(0.005)  # on the same stream, ahead of the slab's kernels; the protocol's one wait covers it
                    o = group.step()[0]
                    return {"stage_ms": o["stage_ms"], "mesh": o["mesh"], "region_count": o["region_count"]}

                return g, nm, st, st.obj, fn, "RCCL (ncclSend / ncclRecv / ncclAllGather inside the library)"
            # host-staged protocol check (IVX_BENCH_BACKEND=gloo): the same phases driven from Python over torch.distributed
            st = SlabStepper(ctx, g, dens, rank, world, torch)
            tc = TorchComm(dist, torch, rank, world)

            def fn():
                body_world.step_enqueue(0.005)
                r = tc.run(st)
                return {"stage_ms": r.stage_ms, "mesh": {"n_vertices": r.mesh_counts[0], "n_indices": r.mesh_counts[1]}, "region_count": r.region_count}

            return g, nm, st, st.obj, fn, f"torch.distributed {dist.get_backend()} (host-staged protocol check)"

        scaling = args.scaling if args.workload == "asteroid" else "strong"
        graph, scene_name, stepper, obj, step, transport = slab_leg(scaling)
        workload = (f"{scene_name} -> {stepper.global_shape} stored grid, x-slabs of {obj.chunk_counts[0]} chunk planes per rank "
                    f"({obj.n_chunks} chunks on rank 0)")
        parallelism = f"x-slab domain decomposition over {world} GPUs, 1-voxel halos + region equivalences over {transport}"
        if native_comm is not None and world > 1:
            parallelism += ("; neighbour exchanges on the communicator's own stream beside the slab's interior work (IVX_SLAB_OVERLAP=1)"
                            if os.environ.get("IVX_SLAB_OVERLAP", "0")[:1] == "1"
                            else "; neighbour exchanges on the compute stream (the default; IVX_SLAB_OVERLAP=1 for the overlapped form)")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # Warm-up with every slot timed: it names the dominant slot. The timed region then records events around that slot only (two
    # records per step instead of twelve: an event record costs ~2-3 us on the GPU's queue, 36 us per step with all of them) —
    # `roofline` is computed from those live times; the other slots' times come from a short pass of their own afterwards.
    obj.set_stage_timing(0xFFFFFFFF)
    res = None
    for _ in range(max(args.warmup, 1)):
        res = step()
    dom = int(np.argmax(np.asarray(res["stage_ms"], dtype=np.float64)))
    if dist is not None:  # every rank times the same slot
        t = torch.tensor([dom], dtype=torch.int64, device="cuda")
        dist.broadcast(t, 0)
        dom = int(t.item())
    obj.set_stage_timing(1 << dom)
    barrier()
    t0 = time.perf_counter()
    dom_sum = 0.0
    for _ in range(args.steps):
        res = step()
        dom_sum += float(res["stage_ms"][dom])
    barrier()
    elapsed = time.perf_counter() - t0
    obj.set_stage_timing(0xFFFFFFFF)
    stage_sum = np.zeros(capi.N_TIMED_STAGES, dtype=np.float64)
    n_stage_pass = max(3, min(args.steps, 10))
    for _ in range(n_stage_pass):
        res = step()
        stage_sum += res["stage_ms"]
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    isolated = None
    if not slabs and sample_ahead and not args.plain:
        # the same step with its pre-pass as its own first kernel (what a step that is not followed by another costs)
        obj.set_sample_ahead(False)
        n_iso = max(5, min(args.steps, 100))
        for _ in range(3):
            step()
        iso_stage = np.zeros(capi.N_TIMED_STAGES, dtype=np.float64)
        for _ in range(3):
            iso_stage += step()["stage_ms"]
        obj.set_stage_timing(1 << dom)  # (as in the timed region)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_iso):
            step()
        barrier()
        isolated = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / n_iso, "steps": n_iso, "sdf_sample_ms": round(float(iso_stage[0]) / 3, 4),
                    "what": "ivx_grid_set_sample_ahead(0): the interval pre-pass is the sample stage's first kernel, every step waits for it"}
        obj.set_stage_timing(0xFFFFFFFF)
        obj.set_sample_ahead(True)
        step()  # (the legs below find the object as the timed region left it)

    # N > 1: the other scaling mode in the same run (the metric's 1 -> 8 target is about the strong leg, config 5 is the weak one)
    other_leg = None
    if slabs and args.workload == "asteroid":
        mode2 = "weak" if scaling == "strong" else "strong"
        _, nm2, st2, obj2, step2, _ = slab_leg(mode2)
        for _ in range(max(2, min(args.warmup, 5))):
            r2 = step2()
        n2 = max(5, min(args.steps, 50))
        barrier()
        t2 = time.perf_counter()
        for _ in range(n2):
            r2 = step2()
        barrier()
        e2 = time.perf_counter() - t2
        t = torch.tensor([e2], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2 = float(t.item())
        t = torch.tensor([obj2.n_voxels, int(r2["mesh"]["n_indices"]) // 3], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        other_leg = {"scaling": mode2, "workload": f"{nm2} -> {st2.global_shape} stored grid, x-slabs of {obj2.chunk_counts[0]} chunk planes per rank",
                     "steps": n2, "ms_per_step": 1e3 * e2 / n2, "value": int(t[0].item()) / (e2 / n2), "unit": "voxels/s", "triangles": int(t[1].item()),
                     "regions": int(r2["region_count"])}
        st2.close()

    n_vox_rank = obj.n_voxels
    tris_rank = int(res["mesh"]["n_indices"]) // 3
    if dist is not None:
        t = torch.tensor([n_vox_rank, tris_rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        n_vox_total, tris_total = int(t[0].item()), int(t[1].item())
    else:
        n_vox_total, tris_total = n_vox_rank, tris_rank
    stage_ms = stage_sum / n_stage_pass
    stage_ms[dom] = dom_sum / args.steps  # the dominant slot: measured inside the timed region
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        exposed, non_uniform = chunk_census(obj)
        counters = obj.stage_counters()
        # chunks whose voxels are touched per step: the ones that have planes (NonUniform) and the ones the sampler evaluates per voxel
        active = max(non_uniform, counters["evaluated_chunks"]) * 4096
        nv, ni = int(res["mesh"]["n_vertices"]), int(res["mesh"]["n_indices"])
        sb_eff = stage_bytes(n_vox_rank, obj.n_chunks, exposed, nv, ni)
        sb_act = stage_bytes(active, obj.n_chunks, exposed, nv, ni)
        key = ("headline" if abs(args.scale - 2.05) < 1e-9 else None) if args.workload == "asteroid" else ("dense" if args.dense_chunks == 32 else None)
        rl, srl, vrl = roofline_block(stage_ms, sb_eff, sb_act, key if world == 1 else None)
        # N = 1: Surface Nets timed alone over the resident object; N > 1: the three launches that host it inside the step (they
        # also carry the region / moment table roles)
        remesh_ms = remesh_only_ms(ctx, obj) if not (slabs or args.plain) else float(stage_ms[2] + stage_ms[3] + stage_ms[4])
        out = {
            "metric": "voxels stepped/sec + remesh tris/sec, 512^3 grid, 1/2/4/8 MI355X",
            "value": n_vox_total / (elapsed / args.steps),
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "stage_timing": f"timed region: events around slot {dom} ({capi.STAGE_NAMES[dom]}) only; `stage_ms` of the other slots from {n_stage_pass} further steps with every slot timed",
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "i8 voxels, f32 SDF/mesh arithmetic, f64 moments",
            "data": "synthetic",
            "config": {"workload": workload, "parallelism": parallelism, "voxels_per_gpu": n_vox_rank, "active_voxels_rank0": active,
                       "regions": int(res["region_count"]), "triangles": tris_total, "vertices_rank0": nv, "exposed_chunks_rank0": exposed,
                       "non_uniform_chunks_rank0": non_uniform, "evaluated_chunks_rank0": counters["evaluated_chunks"],
                       "meshed_chunks_rank0": counters["meshed_chunks"]},
            "remesh_tris_per_s": tris_rank / (remesh_ms * 1e-3) if remesh_ms > 0 else None,
            "remesh_ms": remesh_ms,
            "active_voxels_per_s": active * world / (elapsed / args.steps),
            "headline_active": {"value": active * world / (elapsed / args.steps), "unit": "active voxels/s",
                                "note": "voxels of the chunks that have planes (NonUniform or evaluated): what the kernels touch; `value` counts every stored "
                                        "voxel — Void / Uniform chunks are settled from their 8-byte records, as in the reference's sparse store"},
            "stage_ms": {capi.STAGE_NAMES[i]: round(float(stage_ms[i]), 4) for i in range(capi.N_TIMED_STAGES)},
            "stage_gbs": {capi.STAGE_NAMES[i]: round(sb_act[capi.STAGE_NAMES[i]] / (stage_ms[i] * 1e-3) / 1e9, 1)
                          if stage_ms[i] > 0 and sb_act[capi.STAGE_NAMES[i]] > 0 else None for i in range(capi.N_TIMED_STAGES)},
            "stage_gbs_accounting": "active-voxel algorithmic bytes / live HIP-event stage time (never above the 6.3 TB/s copy rate by construction of the bytes)",
            "roofline": rl,
            "step_roofline": srl,
        }
        if not slabs:
            out["sample_ahead"] = {"on": bool(sample_ahead), "isolated": isolated,
                                   "what": "the sample stage's interval pre-pass (k_sdf_prepass: reads only the resident program) is enqueued one step ahead on the "
                                           "context's second stream, behind the evaluator of the step before; `stage_ms.sdf_sample` is then the evaluator's launch(es) "
                                           "(+ the wait for the pre-pass, normally over); every step still runs one pre-pass"}
        if vrl:
            out["valu_roofline"] = vrl
        if slabs:
            # both scaling modes of this N in one line (`value` / `scaling` above are the mode asked for: strong unless --scaling weak)
            main_leg = {"scaling": scaling, "workload": workload, "steps": args.steps, "ms_per_step": ms_per_step, "value": out["value"], "unit": "voxels/s",
                        "triangles": tris_total, "regions": int(res["region_count"])}
            out["scaling_legs"] = {main_leg["scaling"]: main_leg}
            if other_leg:
                out["scaling_legs"][other_leg["scaling"]] = other_leg
            out["ranks"] = {"world_size": world, "backend": dist.get_backend(),
                            "communicator_ranks": (native_comm.info()["nranks"] if native_comm is not None else dist.get_world_size()),
                            "communicator": ("RCCL communicator made by the library (ncclCommCount)" if native_comm is not None else "torch.distributed process group")}
        o_big = m_big = None
        if world == 1 and not args.no_cpu_baseline:
            what = "full N=1 workload" if args.workload == "asteroid" else "the timed scene"
            if args.workload == "dense":  # (the 512^3 all-surface oracle takes half a minute: bounded sample instead, see dense_benchmark)
                out["cpu_baseline"] = None
            else:
                base, allc, parity, o_big, m_big = cpu_baseline(graph, obj, res, what)
                out["cpu_baseline"], out["cpu_baseline_all_cores"], out["parity"] = base, allc, parity
        elif world == 1:
            out["cpu_baseline"] = None
        if not args.no_pile and world == 1:
            with_cpu = not args.no_cpu_baseline
            gc.collect()
            out["dense"] = dense_benchmark(ctx, args, with_cpu) if args.workload == "asteroid" else None
            gc.collect()
            out["config2"] = config2_benchmark(ctx, args, with_cpu)
            gc.collect()
            out["config3"] = config3_benchmark(ctx, args, with_cpu)
            gc.collect()
            out["config5_one_gpu"] = config5_benchmark(ctx, args)
            gc.collect()
            out["strong_512"] = strong_512_benchmark(ctx, args, out["ms_per_step"], int(out["config"]["triangles"]))
            gc.collect()
            out["fragments"] = fragments_benchmark(ctx, with_cpu)
            gc.collect()
            out["fragments_frame"] = fragments_frame_benchmark(ctx, with_cpu)
            gc.collect()
            pile, w = pile_benchmark(ctx, with_cpu)
            out["pile"] = pile
            gc.collect()
            out["pile"]["churn"] = pile_churn_benchmark(ctx, with_cpu)
            gc.collect()
            # the full frame: the voxel step of the headline body + the pile's solve, enqueued back to back, one wait
            for _ in range(2):
                obj.step_enqueue(capi.STAGE_ALL)
                w.step_enqueue(0.005)
                obj.step_collect()
            t0 = time.perf_counter()
            for _ in range(10):
                obj.step_enqueue(capi.STAGE_ALL)
                w.step_enqueue(0.005)
                obj.step_collect()
            frame_ms = 1e3 * (time.perf_counter() - t0) / 10
            w.close()
            out["frame"] = {"workload": "the timed voxel step + one config-4 pile step (resident contacts), enqueued back to back, one wait",
                            "ms_per_frame": frame_ms, "voxels_per_s": n_vox_rank / (frame_ms * 1e-3)}
            # the same frame with the rigid-body world on a context (HIP stream) of its own: the solve keeps a handful of CUs busy
            # (one workgroup per 256 chains of the widest level), the voxel step runs beside it on the rest of the chip
            from impact_amd import scenes as _scenes
            from impact_amd.physics import PhysicsWorld as _World
            ctx2 = Context(device)
            pb, pc = _scenes.sphere_pile_scene(16)
            w2 = _World(ctx2)
            w2.set_bodies(pb)
            w2.prepare_constraints(pc)
            w2.step(0.005)
            w2.prepare_constraints(pc)
            for _ in range(2):
                w2.step_enqueue(0.005)
                obj.step_enqueue(capi.STAGE_ALL)
                obj.step_collect()
                ctx2.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                w2.step_enqueue(0.005)
                obj.step_enqueue(capi.STAGE_ALL)
                obj.step_collect()
                ctx2.synchronize()
            frame2_ms = 1e3 * (time.perf_counter() - t0) / 10
            out["frame"]["ms_per_frame_two_streams"] = frame2_ms
            out["frame"]["two_streams"] = "the pile's world on a second ivx_ctx (its own stream): solve and voxel step overlap, two waits"
            # The whole per-frame pipeline, host work on the clock (engine/src/tasks.rs:405-434, 515-550 in the reference's order): contact generation
            # of the voxel body against a ground plane (f1, ivx_plane_voxel_object_contacts) -> this frame's contacts into the solver
            # (ivx_world_set_contacts: the pile's 46 080, same ids as the frame before -> its one-pass path) -> solve + integrate -> voxel step ->
            # one voxel edit (an absorbing sphere, ivx_absorb_sphere: edit + re-derive). The world runs on its own stream; nothing is resident
            # from a frame before except what the engine keeps resident (bodies, the SDF program, the solver's schedule).
            m32 = np.asarray(res["moments"]["m32"], dtype=np.float32).reshape(-1)
            com = (m32[1:4] * (np.float32(1.0) / m32[0])).astype(np.float32)
            lo = np.array([a for a, _ in obj.update_occupied_voxel_ranges()], dtype=np.float32)
            hi = np.array([b for _, b in obj.update_occupied_voxel_ranges()], dtype=np.float32)
            ident = np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)
            plane_n = np.array([0.0, 1.0, 0.0], dtype=np.float32)
            plane_d = float(lo[1] + 3.0)  # a ground plane three voxels into the body's underside
            edit_c = np.array([com[0], hi[1], com[2]], dtype=np.float32)  # a bite at the top (the first frame removes it, the rest re-run the path)
            t_parts = np.zeros(5)
            n_frames, n_plane = 10, 0
            for it in range(2 + n_frames):
                if it == 2:
                    ctx.synchronize()
                    ctx2.synchronize()
                    t_parts[:] = 0.0
                    t0 = time.perf_counter()
                ta = time.perf_counter()
                pcs = obj.plane_contacts(ident, np.zeros(3, dtype=np.float32), plane_n, plane_d, 7, 8, 0, 0x80000000)
                tb = time.perf_counter()
                w2.prepare_constraints(pc)
                tc = time.perf_counter()
                w2.step_enqueue(0.005)
                obj.step_enqueue(capi.STAGE_ALL)
                td = time.perf_counter()
                obj.step_collect()
                te = time.perf_counter()
                obj.absorb_sphere(edit_c, 10.0, 8.0, want_invalidated=False)
                ctx2.synchronize()
                tf = time.perf_counter()
                t_parts += (tb - ta, tc - tb, td - tc, te - td, tf - te)
                n_plane = len(pcs)
            pipe_ms = 1e3 * (time.perf_counter() - t0) / n_frames
            obj.step(capi.STAGE_ALL)  # (the body as generated again for the legs below)
            w2.close()
            ctx2.close()
            out["frame"]["pipeline"] = {
                "what": "per frame, host work included: voxel-body contacts against a ground plane (f1) -> ivx_world_set_contacts(46 080 pile contacts, same ids) "
                        "-> solve + integrate (own stream) -> voxel step -> one absorbing-sphere edit with its re-derive; two contexts",
                "ms_per_frame": pipe_ms, "voxels_per_s": n_vox_rank / (pipe_ms * 1e-3), "plane_contacts": n_plane,
                "host_ms": {k: round(1e3 * float(v) / n_frames, 4) for k, v in zip(("contact_generation", "set_contacts", "enqueue", "wait_voxel_step", "edit_and_wait_world"), t_parts)}}
            if args.workload == "asteroid":
                out["collide"] = collide_benchmark(ctx, args.scale, o_big if with_cpu else None, m_big)
                out["edit"] = edit_benchmark(ctx, args.scale, o_big if with_cpu else None)  # (last: the oracle's edit changes o_big)
        elif not args.no_pile:
            out["pile"], w = pile_benchmark(ctx, with_cpu=False)
            w.close()
        out = compact_parities(out)
        out["summary"] = summary_block(out)  # (last key: the tail of the line)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    body_world.close()
    if slabs:
        stepper.close()  # (the slab's buffers, then its grid)
        if native_comm is not None:
            native_comm.close()
    else:
        obj.close()
    ctx.close()


if __name__ == "__main__":
    main()
