// C++ host-side mirror of the reference's interface for the deformable-voxel path, header-only, above the C ABI of
// include/impact_voxel_hip.h. The reference is Rust; its types on this path are concrete structs, so the mirror
// keeps their names, method names and argument meaning (paths relative to /root/reference/engine/crates/):
//   SDFNode / SDFGraph / SDFGenerator        impact_voxel/src/generation/sdf/atomic.rs:55-181, 228-596, 1019-1148
//   SDFVoxelGenerator                        impact_voxel/src/generation.rs:204-258
//   VoxelObject                              impact_voxel/src/object.rs:239-404, 1136-1198; split_detection.rs:193-301; extraction.rs:78-119
//   VoxelObjectMesh                          impact_voxel/src/mesh.rs:286-456
//   VoxelObjectInertialPropertyManager       impact_voxel/src/object/inertia.rs:125-169
//   apply_sphere_absorption & co             impact_voxel/src/interaction/absorption.rs:801-1079
//   for_each_*_voxel_object_contact          impact_voxel/src/collidable.rs:859-1286
//   perform_physics_step / ConstraintSolver  impact_physics/src/lib.rs:31-110
// Errors: every C status other than IVX_OK becomes an impact_voxel::Error carrying ivx_last_error() — where the reference would
// return Err or panic on a violated precondition. Nothing here computes: it owns handles, sizes buffers and forwards.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "impact_voxel_hip.h"

namespace impact_voxel {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) {
    if (rc != IVX_OK) throw Error(rc, ivx_last_error());
}

class Context {
public:
    explicit Context(int device = 0, void* stream = nullptr) { check(ivx_init(device, stream, &ctx_)); }
    ~Context() {
        if (ctx_) ivx_shutdown(ctx_);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    void synchronize() { check(ivx_synchronize(ctx_)); }
    ivx_ctx* handle() const { return ctx_; }

private:
    ivx_ctx* ctx_ = nullptr;
};

// ---- SDF graph ---------------------------------------------------------------------------------------------------------------
using SDFNodeID = uint32_t;
struct SDFNode {
    ivx_sdf_node rec{};
    static SDFNode make(uint32_t kind, uint32_t c1, uint32_t c2, std::array<float, 4> p) {
        SDFNode n;
        n.rec.kind = kind, n.rec.child1 = c1, n.rec.child2 = c2, n.rec.pad = 0;
        for (int i = 0; i < 4; ++i) n.rec.p[i] = p[i];
        return n;
    }
    static SDFNode new_sphere(float radius) { return make(0, 0, 0, {radius, 0, 0, 0}); }
    static SDFNode new_capsule(float segment_length, float radius) { return make(1, 0, 0, {segment_length, radius, 0, 0}); }
    static SDFNode new_box(std::array<float, 3> extents) { return make(2, 0, 0, {extents[0], extents[1], extents[2], 0}); }
    static SDFNode new_translation(SDFNodeID child, std::array<float, 3> t) { return make(3, child, 0, {t[0], t[1], t[2], 0}); }
    static SDFNode new_rotation(SDFNodeID child, std::array<float, 4> quaternion_xyzw) { return make(4, child, 0, quaternion_xyzw); }
    static SDFNode new_scaling(SDFNodeID child, float scaling) { return make(5, child, 0, {scaling, 0, 0, 0}); }
    static SDFNode new_union(SDFNodeID a, SDFNodeID b, float smoothness) { return make(7, a, b, {smoothness, 0, 0, 0}); }
    static SDFNode new_subtraction(SDFNodeID a, SDFNodeID b, float smoothness) { return make(8, a, b, {smoothness, 0, 0, 0}); }
    static SDFNode new_intersection(SDFNodeID a, SDFNodeID b, float smoothness) { return make(9, a, b, {smoothness, 0, 0, 0}); }
};
class SDFGraph {
public:
    SDFNodeID add_node(const SDFNode& n) {  // the last node added is the root (atomic.rs:1037-1046)
        nodes_.push_back(n.rec);
        root_ = (SDFNodeID)nodes_.size() - 1;
        return root_;
    }
    const std::vector<ivx_sdf_node>& nodes() const { return nodes_; }
    SDFNodeID root_node_id() const { return root_; }

private:
    std::vector<ivx_sdf_node> nodes_;
    SDFNodeID root_ = 0;
};
class SDFGenerator {  // SDFGraph::build (atomic.rs:228-596)
public:
    explicit SDFGenerator(const SDFGraph& g) {
        nodes_.resize(4 * g.nodes().size() + 64);
        size_t n = 0;
        check(ivx_sdf_compile(g.nodes().data(), g.nodes().size(), g.root_node_id(), nodes_.data(), nodes_.size(), &n, domain_, &stack_size_));
        nodes_.resize(n);
    }
    const std::vector<ivx_sdf_processed_node>& nodes() const { return nodes_; }
    uint32_t stack_size() const { return stack_size_; }
    const float* domain() const { return domain_; }

private:
    std::vector<ivx_sdf_processed_node> nodes_;
    float domain_[6] = {0, 0, 0, 0, 0, 0};
    uint32_t stack_size_ = 0;
};
class SDFVoxelGenerator {  // generation.rs:204-258 with a SameVoxelTypeGenerator
public:
    SDFVoxelGenerator(float voxel_extent, SDFGenerator generator, uint8_t voxel_type = 0)
        : extent_(voxel_extent), gen_(std::move(generator)), type_(voxel_type) {
        check(ivx_sdf_grid_shape(gen_.domain(), shape_, center_));
    }
    float voxel_extent() const { return extent_; }
    std::array<uint32_t, 3> grid_shape() const { return {shape_[0], shape_[1], shape_[2]}; }
    std::array<uint32_t, 3> chunk_counts() const { return {(shape_[0] + 15) / 16, (shape_[1] + 15) / 16, (shape_[2] + 15) / 16}; }
    const SDFGenerator& sdf_generator() const { return gen_; }
    const uint32_t* shape() const { return shape_; }
    const float* shifted_grid_center() const { return center_; }
    uint8_t voxel_type() const { return type_; }

private:
    float extent_;
    SDFGenerator gen_;
    uint8_t type_;
    uint32_t shape_[3] = {0, 0, 0};
    float center_[3] = {0, 0, 0};
};

// ---- voxel object -------------------------------------------------------------------------------------------------------------
struct Isometry3 {  // rotation (x, y, z, w) then translation
    std::array<float, 4> rotation{0, 0, 0, 1};
    std::array<float, 3> translation{0, 0, 0};
};
struct AbsorptionOutcome {
    ivx_absorb_result result{};
    std::vector<uint32_t> emptied_by_type;
    std::vector<uint8_t> invalidated_mesh_chunks;
};
class VoxelObject;
struct DisconnectedVoxelObject {  // extraction.rs:47-58
    std::unique_ptr<VoxelObject> voxel_object;
    std::array<uint32_t, 3> origin_offset_in_parent{0, 0, 0};
    ivx_region_desc moved{};
};

class VoxelObject {
public:
    VoxelObject(Context& ctx, std::array<uint32_t, 3> chunk_counts, float voxel_extent) : ctx_(&ctx), extent_(voxel_extent), cc_(chunk_counts) {
        check(ivx_grid_create(ctx.handle(), cc_.data(), voxel_extent, 0, cc_[0], &g_));
    }
    ~VoxelObject() {
        if (g_) ivx_grid_destroy(g_);
    }
    VoxelObject(const VoxelObject&) = delete;
    VoxelObject& operator=(const VoxelObject&) = delete;

    static std::unique_ptr<VoxelObject> generate_without_derived_state(Context& ctx, const SDFVoxelGenerator& gen) {
        auto o = std::make_unique<VoxelObject>(ctx, gen.chunk_counts(), gen.voxel_extent());
        const auto& n = gen.sdf_generator().nodes();
        check(ivx_sdf_sample(o->g_, n.data(), n.size(), gen.sdf_generator().stack_size(), gen.shape(), gen.shifted_grid_center(), gen.voxel_type()));
        return o;
    }
    static std::unique_ptr<VoxelObject> generate(Context& ctx, const SDFVoxelGenerator& gen) {  // object.rs:239-244
        auto o = generate_without_derived_state(ctx, gen);
        o->compute_all_derived_state();       // (the reference updates the ranges first; here they are reduced from per-chunk boxes the derive
        o->update_occupied_voxel_ranges();    //  sweep leaves, the result is the same: emptiness does not change in between)
        return o;
    }
    void compute_all_derived_state() {  // object.rs:1136-1145
        check(ivx_derive_state(g_));
        check(ivx_label_regions(g_, &regions_));
    }
    std::array<uint32_t, 12> update_occupied_voxel_ranges() {
        std::array<uint32_t, 12> r{};
        check(ivx_occupied_ranges(g_, r.data()));
        return r;
    }
    uint32_t count_regions() {
        check(ivx_label_regions(g_, &regions_));
        return regions_;
    }
    std::optional<DisconnectedVoxelObject> extract_any_disconnected_region() {  // extraction.rs:78-119
        ivx_grid* child = nullptr;
        DisconnectedVoxelObject d;
        int outcome = 0;
        check(ivx_split_off_smallest_region(g_, &child, d.origin_offset_in_parent.data(), &outcome, &d.moved));
        if (outcome != 1) return std::nullopt;
        d.voxel_object = std::unique_ptr<VoxelObject>(new VoxelObject(*ctx_, child, extent_));
        return d;
    }
    // interaction/absorption.rs:801-889 with the shape in the object's normalized space
    AbsorptionOutcome absorb_sphere(std::array<float, 3> center, float influence_radius, float sphere_radius, const std::array<float, 256>& densities) {
        AbsorptionOutcome a = fresh_outcome();
        check(ivx_absorb_sphere(g_, center.data(), influence_radius, sphere_radius, densities.data(), &a.result, a.emptied_by_type.data(),
                                a.invalidated_mesh_chunks.data()));
        return a;
    }
    AbsorptionOutcome absorb_capsule(std::array<float, 3> segment_start, std::array<float, 3> segment_vector, float influence_radius, float capsule_radius,
                                     const std::array<float, 256>& densities) {
        AbsorptionOutcome a = fresh_outcome();
        check(ivx_absorb_capsule(g_, segment_start.data(), segment_vector.data(), influence_radius, capsule_radius, densities.data(), &a.result,
                                 a.emptied_by_type.data(), a.invalidated_mesh_chunks.data()));
        return a;
    }
    // collidable.rs:1098-1127 (sphere collidable in world space, transform_to_object_space of this object)
    std::vector<ivx_contact> sphere_contacts(const Isometry3& to_object, std::array<float, 3> center, float radius, uint64_t id_a, uint64_t id_b, uint32_t body_a,
                                             uint32_t body_b, std::array<float, 3> response, size_t capacity = 65536) {
        std::vector<ivx_contact> out(capacity);
        size_t n = 0;
        check(ivx_sphere_voxel_object_contacts(g_, to_object.rotation.data(), to_object.translation.data(), center.data(), radius, id_a, id_b, body_a, body_b,
                                               response.data(), out.data(), capacity, &n));
        out.resize(n);
        return out;
    }
    // collidable.rs:859-1049: contacts between this object (A) and `other` (B); both need current collision probes
    std::vector<ivx_contact> mutual_contacts(const Isometry3& to_object, std::array<float, 3> center_of_mass, VoxelObject& other, const Isometry3& other_to_object,
                                             std::array<float, 3> other_center_of_mass, uint64_t id_a, uint64_t id_b, uint32_t body_a, uint32_t body_b,
                                             std::array<float, 3> response, size_t capacity = 65536) {
        std::vector<ivx_contact> out(capacity);
        size_t n = 0;
        check(ivx_mutual_voxel_object_contacts(g_, to_object.rotation.data(), to_object.translation.data(), center_of_mass.data(), other.g_,
                                               other_to_object.rotation.data(), other_to_object.translation.data(), other_center_of_mass.data(), id_a, id_b,
                                               body_a, body_b, response.data(), out.data(), capacity, &n));
        out.resize(n);
        return out;
    }
    // VoxelObjectCollisionProbes (collidable.rs:361-433): recompute over the current mesh, or follow a mesh sync
    size_t recompute_collision_probes() {
        size_t n = 0;
        check(ivx_collision_probes_recompute(g_, &n));
        return n;
    }
    size_t sync_collision_probes(const std::vector<uint8_t>& invalidated_mesh_chunks) {
        size_t n = 0;
        check(ivx_collision_probes_sync(g_, invalidated_mesh_chunks.data(), &n));
        return n;
    }
    struct Dense {
        std::vector<int8_t> sdf;
        std::vector<uint8_t> type, flags, local_labels;
        std::vector<ivx_chunk_info> info;
    };
    Dense download() const {
        Dense d;
        const size_t nc = (size_t)cc_[0] * cc_[1] * cc_[2], nv = nc * 4096;
        d.sdf.resize(nv), d.type.resize(nv), d.flags.resize(nv), d.local_labels.resize(nv), d.info.resize(nc);
        check(ivx_grid_download_dense(g_, d.sdf.data(), d.type.data(), d.flags.data(), d.local_labels.data(), d.info.data(), nv));
        return d;
    }
    float voxel_extent() const { return extent_; }
    std::array<uint32_t, 3> chunk_counts() const { return cc_; }
    size_t n_chunks() const { return (size_t)cc_[0] * cc_[1] * cc_[2]; }
    ivx_grid* handle() const { return g_; }

private:
    VoxelObject(Context& ctx, ivx_grid* adopted, float extent) : ctx_(&ctx), g_(adopted), extent_(extent) {
        uint32_t c[3];
        check(ivx_grid_chunk_counts(g_, c));
        cc_ = {c[0], c[1], c[2]};
    }
    AbsorptionOutcome fresh_outcome() const {
        AbsorptionOutcome a;
        a.emptied_by_type.assign(256, 0);
        a.invalidated_mesh_chunks.assign(n_chunks(), 0);
        return a;
    }
    Context* ctx_;
    ivx_grid* g_ = nullptr;
    float extent_;
    std::array<uint32_t, 3> cc_{0, 0, 0};
    uint32_t regions_ = 0;
};

class VoxelObjectMesh {  // mesh.rs:44-58, 286-456; the buffers stay in HBM
public:
    static VoxelObjectMesh create(VoxelObject& o) {
        VoxelObjectMesh m(o);
        m.recreate();
        return m;
    }
    void recreate() { check(ivx_remesh(o_->handle(), &counts_)); }
    void sync_with_voxel_object(const std::vector<uint8_t>& invalidated_mesh_chunks) { check(ivx_mesh_sync(o_->handle(), invalidated_mesh_chunks.data(), &counts_)); }
    size_t n_vertices() const { return counts_.n_vertices; }
    size_t n_indices() const { return counts_.n_indices; }
    size_t n_chunks() const { return counts_.n_submeshes; }
    struct Buffers {
        std::vector<float> positions, normal_vectors;
        std::vector<uint32_t> indices;
        std::vector<uint8_t> index_materials;
        std::vector<ivx_submesh> chunk_submeshes;
    };
    Buffers download() const {
        Buffers b;
        b.positions.resize(3 * n_vertices()), b.normal_vectors.resize(3 * n_vertices()), b.indices.resize(n_indices());
        b.index_materials.resize(8 * n_indices()), b.chunk_submeshes.resize(n_chunks());
        check(ivx_mesh_download(o_->handle(), b.positions.data(), b.normal_vectors.data(), b.indices.data(), b.index_materials.data(), b.chunk_submeshes.data()));
        return b;
    }

private:
    explicit VoxelObjectMesh(VoxelObject& o) : o_(&o) {}
    VoxelObject* o_;
    ivx_mesh_counts counts_{};
};

class VoxelObjectInertialPropertyManager {  // object/inertia.rs:20-25, 125-169
public:
    static VoxelObjectInertialPropertyManager initialized_from(VoxelObject& o, const std::array<float, 256>& voxel_type_densities) {
        VoxelObjectInertialPropertyManager m;
        check(ivx_inertia(o.handle(), voxel_type_densities.data(), &m.moments_));
        return m;
    }
    double mass() const { return moments_.m64[0]; }
    std::array<double, 3> derive_center_of_mass() const { return {moments_.m64[1] / moments_.m64[0], moments_.m64[2] / moments_.m64[0], moments_.m64[3] / moments_.m64[0]}; }
    const ivx_moments& moments() const { return moments_; }

private:
    ivx_moments moments_{};
};

// ---- rigid bodies + constraint solver (impact_physics/src/lib.rs:31-110) --------------------------------------------------------------
class PhysicsWorld {
public:
    PhysicsWorld(Context& ctx, const ivx_solver_config& config) { check(ivx_world_create(ctx.handle(), &config, &w_)); }
    ~PhysicsWorld() {
        if (w_) ivx_world_destroy(w_);
    }
    PhysicsWorld(const PhysicsWorld&) = delete;
    PhysicsWorld& operator=(const PhysicsWorld&) = delete;
    void set_bodies(const std::vector<ivx_rigid_body>& dynamic, const std::vector<ivx_kinematic_body>& kinematic = {}) {
        n_dyn_ = dynamic.size(), n_kin_ = kinematic.size();
        check(ivx_world_set_bodies(w_, dynamic.data(), dynamic.size(), kinematic.data(), kinematic.size()));
    }
    ivx_physics_result perform_physics_step(const std::vector<ivx_contact>& contacts, float step_duration) {
        size_t prepared = 0;
        check(ivx_world_set_contacts(w_, contacts.data(), contacts.size(), &prepared));
        ivx_physics_result r{};
        check(ivx_world_step(w_, step_duration, &r));
        return r;
    }
    std::vector<ivx_rigid_body> dynamic_bodies() {
        std::vector<ivx_rigid_body> d(n_dyn_);
        check(ivx_world_get_bodies(w_, d.data(), nullptr));
        return d;
    }

private:
    ivx_world* w_ = nullptr;
    size_t n_dyn_ = 0, n_kin_ = 0;
};

}  // namespace impact_voxel
