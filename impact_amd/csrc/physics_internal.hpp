// Device-side layouts and the host-side world object of the rigid-body / contact-solver part
// (a15-a19). See physics.hip for the kernels and physics_api.hip for the bookkeeping.
#pragma once
#include <cstdint>
#include <vector>

#include "ivx_internal.hpp"

// ConstrainedBody (impact_physics/src/constraint.rs:137-150), 96 bytes. Dynamic bodies first (same index as
// in the rigid-body array), kinematic bodies after them.
struct PhysBody {
    float inv_mass;
    float inv_inertia[9];  // world space, column-major
    float pos[3];
    float q[4];
    float v[3];
    float w[3];
    float pad;
};
static_assert(sizeof(PhysBody) == 96, "PhysBody layout");

// PreparedContact (constraint/contact.rs:63-86) + the constrained-body indices of its pair, 96 bytes
struct PhysContact {
    float local_a[3], local_b[3], normal[3], tangent[3], bitangent[3];
    float m_n, m_t, m_b, friction, target;
    float world_b[3];  // the contact point on body B in world space as the velocity phase sees it: qrot(q_b, local_b) + pos_b, fixed for the step
    uint32_t pad;
};
static_assert(sizeof(PhysContact) == 96, "PhysContact layout");

#define PHYS_ITEM_WARM 0u
#define PHYS_ITEM_VELOCITY 1u
#define PHYS_ITEM_POSITIONAL 2u
#define PHYS_LEVEL_TILE 4096
// A chain is at most this many contacts: what the chain-stationary solve keeps of a chain in one lane's registers (longer manifolds are
// consecutive chains on the same pair; the dependency schedule orders them like any two items that share a body)
#define PHYS_CHAIN_MAX 4u
// the chain-stationary solve (physics.hip, k_solve_cs): waves per workgroup, and the most workgroups one phase may ask for (its working
// workgroups share one XCD: 32 CUs, one workgroup of 12 x 64 threads — three waves per SIMD at up to 168 registers — on each)
#define PHYS_CS_WAVES 12u
#define PHYS_CS_MAX_GROUPS 32u
#define PHYS_SOLVER_STATIONARY 255u  // ivx_world_set_solver_groups: force the chain-stationary solve where the schedule allows it

struct ivx_world {
    ivx_ctx* ctx;
    ivx_solver_config cfg;
    uint32_t n_dyn, n_kin;
    size_t body_cap;
    ivx_rigid_body* dyn;
    ivx_kinematic_body* kin;
    PhysBody* cb;
    uint8_t* touched;
    // contacts of the current step, in ConstraintCache order (= solve order)
    uint32_t n_contacts, n_prev;
    size_t contact_cap, item_cap, item_bodies_cap, level_cap;
    ivx_contact* contacts;
    int32_t* prev_slot;
    PhysContact* pc[2];
    float* acc[2];  // float4 per contact
    int cur;
    uint32_t* item_tags;  // [4 * items] per item: the versions it finds its bodies a and b at (a body's record carries the number of items that
                          // have touched it this phase), the sweep tag it finds on its contacts' accumulated impulses, the tag it leaves there
    size_t item_tags_cap;
    std::vector<uint32_t> item_tags_host, scratch_count;
    uint32_t* items;
    uint32_t* item_bodies;  // uint2 per item: constrained-body indices of the chain's pair
    uint32_t* level_start;
    uint32_t n_levels[2], item_offset[2], level_offset[2];
    uint32_t max_level_items[2];  // widest level of each phase's schedule
    // packed per-item records of the multi-workgroup solve (physics.hip, k_pack_items): tiles of 64 consecutive items of a level.
    // tile_base: per level (indexed like level_start) the phase-relative index of its first tile; tile_first: per tile the phase-relative
    // index of its first item | (items in the tile - 1) << 26
    uint32_t* tile_base;
    uint32_t* tile_first;
    size_t tile_base_cap, tile_first_cap;
    uint32_t n_tiles[2], tile_offset[2];
    // (what an item waits for is in the shared records themselves: version tags, item_tags below)
    float* packed[2];
    size_t packed_cap[2];
    // kinematic orientations in the positional phase (physics.hip, ReplayView): the positional chains of every kinematic body in solve order
    // (CSR: kin_offsets[n_kin + 1], kin_list: phase-relative item | side << 31), per-item counts and tables of the two passes
    // the chain-stationary solve (physics.hip, k_solve_cs): every chain lives in a pair of lanes of one wave for the whole phase (even lane: body
    // A's side, odd: B's). Per phase: tiles of 32 chains in the order of their first level; per slot (tile * 64 + lane) the chain word (first
    // contact | length << 24, ~0 = empty), the lane's own body and the other one (constrained-body indices), and degree | rank << 16 of the
    // chain on the own body — the version that body's record has when sweep s of the chain starts is s * degree + rank; per tile its rounds
    // in level order (64-bit lane masks, the level beside each). cs_slot_of: positional phase with kinematic bodies only, [(tile * 32 + pair)
    // * passes + sweep] -> the item's index in the level schedule (what ReplayView is indexed by).
    struct CsSchedule {
        uint32_t n_tiles = 0;  // 0: this phase cannot run chain-stationary (too many chains, a body with more than 65535 chains)
        uint32_t slot_offset = 0, round_start_offset = 0, round_offset = 0;  // into cs_item / cs_bodies / cs_vers, cs_round_start, cs_round_mask
    } cs[2];
    uint32_t* cs_item;
    uint32_t* cs_bodies;
    uint32_t* cs_vers;
    uint32_t* cs_round_start;
    uint64_t* cs_round_mask;
    uint32_t* cs_round_level;  // the level a round's items lie on (from 1): what a wave naps by until its round comes near
    uint32_t* cs_slot_of;
    size_t cs_item_cap, cs_bodies_cap, cs_vers_cap, cs_round_start_cap, cs_round_mask_cap, cs_round_level_cap, cs_slot_of_cap;
    std::vector<uint32_t> cs_item_host, cs_bodies_host, cs_vers_host, cs_round_start_host, cs_round_level_host, cs_slot_of_host;
    std::vector<uint64_t> cs_round_mask_host;
    uint32_t solver_kind_used;  // 0: one workgroup, 1: tile dataflow on several (k_solve_mg), 2: chain-stationary (k_solve_cs)
    uint32_t* kin_offsets;
    uint32_t* kin_list;
    uint32_t* kin_applied;
    float* kin_qstart;  // float4 per (item, side): the orientation a kinematic body has when the chain starts (pass 2)
    float* kin_snap;
    size_t kin_offsets_cap, kin_list_cap, kin_applied_cap, kin_qstart_cap, kin_snap_cap;
    uint32_t n_kin_items;  // positional items that involve a kinematic body (0: the phase runs once, as without kinematic bodies)
    // the solve on several workgroups (physics.hip, k_solve_mg): the phase's mutable body state as shared 32-byte records, the
    // monotonic arrival counter of the grid barrier + an error word (a bounded poll gave up), how many arrivals have been used up
    // SphericalJoint constraints (constraint/spherical_joint.rs): the reference's joint computes no impulse and no correction (:62-88);
    // all it does is register its two bodies as constrained bodies, which get their velocities written back after the solve
    uint32_t* joint_refs;  // device: body references (IVX_KINEMATIC_BODY flag) of all joints' anchors
    uint32_t n_joint_refs;
    std::vector<uint32_t> joint_refs_host;  // the same references on the host (the step's body count, ivx_world_step)
    uint32_t n_bodies_stat;   // constrained bodies of the resident contacts and joints (ivx_physics_result::n_bodies), while `n_bodies_stat_valid`
    int n_bodies_stat_valid;  // (a function of the contact list, the joints and the body counts: recounted on the host when one of them changes)
    float* dynst;
    uint32_t* barrier_words;
    uint32_t* mg_err_host;  // host-mapped word the multi-workgroup solve sets when its grid barrier gave up (checked at every wait on the stream)
    uint32_t* mg_err_dev;   // its device-side address
    int mg_disabled;        // a barrier timed out in this world before: the solve stays on the single-workgroup kernel
    uint32_t barrier_count;   // arrivals counted so far on the velocity phase's grid-barrier counter (barrier_words[0])
    uint32_t barrier_count1;  // ... on the positional phase's (barrier_words[1]): the two phases run side by side, each on its own counter
    hipStream_t side_stream;  // the positional phase's stream (ivx_launch_phys_solve); created on first use
    hipEvent_t ev_fork, ev_join;
    uint32_t solver_groups_forced, solver_groups_used;
    int schedule_valid, prepared_fresh;
    hipEvent_t ev[5];
    int ev_ready;
    // host-side ConstraintCache<ContactID, _> bookkeeping (constraint/solver.rs:60-66, 386-452)
    struct Entry {
        uint64_t id;
        int32_t prev_slot;
        uint32_t src;
        bool prepared;
    };
    std::vector<Entry> cache;
    // id -> slot: open addressing (physics_api.hip, id_find_pos); a value of ~0 marks an empty entry
    std::vector<uint64_t> id_keys;
    std::vector<uint32_t> id_vals;
    size_t id_used;
    std::vector<ivx_contact> effective;  // contacts of this step after interlock replacement (only when a manifold is interlocked)
    std::vector<uint32_t> slot_bodies;   // (body_a, body_b) of every resident contact in slot order (the contacts themselves: stage_contacts)
    // the resident contacts in slot order on the host: a pinned block, the source of their upload (an asynchronous copy; `stage_ev` follows it)
    ivx_contact* stage_contacts;
    size_t stage_contacts_cap;
    hipEvent_t stage_ev;
    int stage_ev_ready, stage_busy;
    // the general path's uploads (contacts in cache order, warm-start sources, a new schedule): one pinned staging block, asynchronous copies
    // behind an event — the path used to make seventeen blocking copies and two waits per frame whose contact set had changed
    char* stage_sched;
    size_t stage_sched_cap;
    hipEvent_t stage_sched_ev;
    int stage_sched_ev_ready, stage_sched_busy;
    std::vector<int32_t> prev_slot_host;
    std::vector<uint32_t> item_bodies_host, items_host, level_start_host, tile_base_host, tile_first_host, scratch_level, scratch_last, chain_start;
    std::vector<uint32_t> kin_offsets_host, kin_list_host;
    // what ivx_world_set_contacts builds of the schedule (build_levels): per phase the level of every item (pass-major, chains in solve order)
    // and whether the chain-stationary form exists; the forms the kernels read are built on demand (ivx_world_ensure_form: bit 0 the level-
    // ordered items and tiles, bit 1 the chain-stationary tiles)
    std::vector<uint32_t> lvl_host[2], cs_slot_of_level, scratch_rank, scratch_order, scratch_first;
    std::vector<uint64_t> scratch_mask, scratch_bits;
    uint32_t phase_items[2];
    int cs_feasible[2];
    uint32_t forms_built;
    std::vector<uint32_t> chain_bodies, prev_chain_start, prev_chain_bodies;  // body pair per chain; last frame's chains (an unchanged contact structure keeps its schedule)
};

int ivx_launch_phys_prepare_bodies(ivx_world* w);
int ivx_launch_phys_prepare_contacts(ivx_world* w, const int32_t* d_prev_slot);
int ivx_launch_phys_mark_joint_bodies(ivx_world* w);
int ivx_launch_phys_pre_solve(ivx_world* w, float dt);
int ivx_launch_phys_solve(ivx_world* w);
int ivx_launch_phys_free_step(ivx_world* w, float dt);
int ivx_world_ensure_form(ivx_world* w, uint32_t form);  // physics_api.hip
int ivx_launch_phys_post_solve(ivx_world* w, float dt, int write_back, int advance);
