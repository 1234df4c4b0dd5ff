// Parity check written against the C++ host mirror (include/impact_voxel.hpp), in the shape of the reference's own object / mesh tests:
// BASELINE config 1 (a 30^3 box in a 32^3 grid) and a sphere are generated, meshed, edited and re-meshed through the mirror on the GPU and
// through the oracle's C API on the host; voxel planes, chunk records, meshes and the synced mesh must agree bit for bit, the moments to 1e-5,
// and a ball dropped on the ground must step like the oracle's solver. TEST INFRASTRUCTURE: links the oracle, never part of the product.
#include <cstdio>
#include <cstring>
#include <vector>

#include "impact_voxel.hpp"
#include "oracle.h"

using namespace impact_voxel;

static int failures = 0;
#define EXPECT(cond, ...)                     \
    do {                                      \
        if (!(cond)) {                        \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);         \
            std::printf("\n");                \
            ++failures;                       \
        }                                     \
    } while (0)

static void compare_objects(const char* what, VoxelObject& g, const orc_object* o) {
    const VoxelObject::Dense d = g.download();
    const size_t nv = d.sdf.size(), nc = d.info.size();
    std::vector<int8_t> sdf(nv);
    std::vector<uint8_t> type(nv), flags(nv), labels(nv);
    std::vector<orc_chunk_info> info(nc);
    orc_export_dense(o, sdf.data(), type.data(), flags.data(), labels.data(), info.data());
    static_assert(sizeof(orc_chunk_info) == sizeof(ivx_chunk_info), "chunk record layout");
    EXPECT(std::memcmp(info.data(), d.info.data(), nc * sizeof(ivx_chunk_info)) == 0, "%s: chunk records differ", what);
    size_t bad = 0;
    for (size_t c = 0; c < nc; ++c) {
        if (d.info[c].kind != 2) continue;  // (Void / Uniform chunks are their record)
        for (size_t v = c * 4096; v < (c + 1) * 4096; ++v) {
            const bool empty = sdf[v] >= 0;
            bad += sdf[v] != d.sdf[v] || type[v] != d.type[v] || labels[v] != d.local_labels[v] || (!empty && flags[v] != d.flags[v]) ||
                   (empty && ((flags[v] ^ d.flags[v]) & 1u));
        }
    }
    EXPECT(bad == 0, "%s: %zu voxels differ", what, bad);
}

static void compare_meshes(const char* what, const VoxelObjectMesh& gm, const orc_mesh* om) {
    uint32_t cnt[3];
    orc_mesh_counts(om, cnt);
    EXPECT(cnt[0] == gm.n_vertices() && cnt[1] == gm.n_indices() && cnt[2] == gm.n_chunks(), "%s: mesh counts %u/%u/%u vs %zu/%zu/%zu", what, cnt[0], cnt[1],
           cnt[2], gm.n_vertices(), gm.n_indices(), gm.n_chunks());
    if (cnt[0] != gm.n_vertices() || cnt[1] != gm.n_indices() || cnt[2] != gm.n_chunks()) return;
    std::vector<float> pos(3 * cnt[0]), nrm(3 * cnt[0]);
    std::vector<uint32_t> idx(cnt[1]), sub(16 * (size_t)cnt[2]);
    std::vector<uint8_t> im(8 * (size_t)cnt[1]);
    orc_mesh_get(om, pos.data(), nrm.data(), idx.data(), im.data(), sub.data());
    const VoxelObjectMesh::Buffers b = gm.download();
    static_assert(sizeof(ivx_submesh) == 64, "submesh layout");
    EXPECT(std::memcmp(sub.data(), b.chunk_submeshes.data(), sub.size() * 4) == 0, "%s: submesh tables differ", what);
    for (uint32_t s = 0; s < cnt[2]; ++s) {  // live ranges only (a synced mesh has freed ranges with stale bytes)
        const uint32_t* q = sub.data() + 16 * (size_t)s;
        const size_t ioff = q[3], icnt = q[4], voff = q[13], vcnt = q[14];
        EXPECT(std::memcmp(&pos[3 * voff], &b.positions[3 * voff], 12 * vcnt) == 0, "%s: positions of submesh %u differ", what, s);
        EXPECT(std::memcmp(&nrm[3 * voff], &b.normal_vectors[3 * voff], 12 * vcnt) == 0, "%s: normals of submesh %u differ", what, s);
        EXPECT(std::memcmp(&idx[ioff], &b.indices[ioff], 4 * icnt) == 0, "%s: indices of submesh %u differ", what, s);
        EXPECT(std::memcmp(&im[8 * ioff], &b.index_materials[8 * ioff], 8 * icnt) == 0, "%s: index materials of submesh %u differ", what, s);
    }
}

int main() {
    try {
        Context ctx(0);
        std::array<float, 256> dens;
        dens.fill(1.0f);
        struct Scene {
            const char* name;
            SDFGraph graph;
            float extent;
        };
        std::vector<Scene> scenes(2);
        scenes[0].name = "config 1: box 30^3";
        scenes[0].graph.add_node(SDFNode::new_box({30.0f, 30.0f, 30.0f}));
        scenes[0].extent = 1.0f;
        scenes[1].name = "sphere r=26 minus a box, extent 0.5";
        {
            SDFGraph& g = scenes[1].graph;
            const SDFNodeID s = g.add_node(SDFNode::new_sphere(26.0f));
            const SDFNodeID b = g.add_node(SDFNode::new_box({60.0f, 6.0f, 9.0f}));
            const SDFNodeID t = g.add_node(SDFNode::new_translation(b, {0.0f, 11.0f, -3.0f}));
            g.add_node(SDFNode::new_subtraction(s, t, 2.0f));
            scenes[1].extent = 0.5f;
        }
        for (Scene& sc : scenes) {
            SDFVoxelGenerator gen(sc.extent, SDFGenerator(sc.graph), 0);
            auto obj = VoxelObject::generate(ctx, gen);
            static_assert(sizeof(orc_sdf_node) == sizeof(ivx_sdf_node), "node layout");
            orc_object* o = orc_object_from_sdf(reinterpret_cast<const orc_sdf_node*>(sc.graph.nodes().data()), (int)sc.graph.nodes().size(),
                                                sc.graph.root_node_id(), sc.extent, 0);
            orc_update_occupied_voxel_ranges(o);
            orc_compute_all_derived_state(o);
            compare_objects(sc.name, *obj, o);
            VoxelObjectMesh mesh = VoxelObjectMesh::create(*obj);
            orc_mesh* om = orc_mesh_recreate(o);
            compare_meshes(sc.name, mesh, om);
            // moments (object/inertia.rs:125-136)
            const auto mgr = VoxelObjectInertialPropertyManager::initialized_from(*obj, dens);
            float m32[10];
            double m64[10];
            orc_inertia(o, dens.data(), m32, m64);
            for (int q = 0; q < 10; ++q)
                EXPECT(std::fabs(mgr.moments().m64[q] - m64[q]) <= 1e-5 * std::fabs(m64[q]) + 1e-6, "%s: moment %d: %.9g vs %.9g", sc.name, q, mgr.moments().m64[q], m64[q]);
            // an absorbing sphere at the +x end of the occupied range, then the incremental remesh of what it invalidated
            int32_t inf[19];
            orc_object_info(o, inf);
            const std::array<float, 3> c = {(float)inf[11], 0.5f * (float)(inf[12] + inf[13]), 0.5f * (float)(inf[14] + inf[15])};
            AbsorptionOutcome a = obj->absorb_sphere(c, 9.0f, 7.0f, dens);
            double removed[10];
            std::vector<uint32_t> by_type(256);
            std::vector<uint8_t> inval(obj->n_chunks());
            uint32_t touched = 0;
            const int removed_chunks = orc_absorb_sphere(o, c.data(), 9.0f, 7.0f, dens.data(), removed, by_type.data(), inval.data(), &touched);
            EXPECT(a.result.touched_chunks == touched && (int)a.result.removed_chunks == removed_chunks, "%s: edit counters differ", sc.name);
            EXPECT(a.emptied_by_type == by_type && a.invalidated_mesh_chunks == inval, "%s: emptied counts / invalidated chunks differ", sc.name);
            EXPECT(a.result.emptied_voxels > 100, "%s: the sphere took only %llu voxels", sc.name, (unsigned long long)a.result.emptied_voxels);
            compare_objects("after the edit", *obj, o);
            mesh.sync_with_voxel_object(a.invalidated_mesh_chunks);
            orc_mesh_sync(om, o, inval.data());
            compare_meshes("after sync_with_voxel_object", mesh, om);
            orc_mesh_free(om);
            orc_object_free(o);
        }
        // a precondition violated: Err where the reference would return one
        {
            SDFGraph g;
            g.add_node(SDFNode::new_sphere(10.0f));
            SDFVoxelGenerator gen(1.0f, SDFGenerator(g), 0);
            auto obj = VoxelObject::generate_without_derived_state(ctx, gen);
            bool threw = false;
            try {
                std::array<float, 256> d1;
                d1.fill(1.0f);
                obj->absorb_sphere({5.0f, 5.0f, 5.0f}, 4.0f, 2.0f, d1);  // derived state missing
            } catch (const Error& e) {
                threw = e.code == IVX_ERR_STATE;
            }
            EXPECT(threw, "absorbing without derived state must fail with IVX_ERR_STATE");
        }
    } catch (const Error& e) {
        std::printf("FAIL: impact_voxel::Error %d: %s\n", e.code, e.what());
        return 2;
    }
    if (failures) {
        std::printf("%d check(s) failed\n", failures);
        return 1;
    }
    std::printf("host mirror check: all equal\n");
    return 0;
}
