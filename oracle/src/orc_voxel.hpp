// ORACLE (test infrastructure only — never linked into the product library).
//
// CPU restatement of the voxel data model and chunked voxel object of the reference:
//   Voxel / VoxelSignedDistance / VoxelFlags  engine/crates/impact_voxel/src/lib.rs:58-101,154-494
//   VoxelObject / VoxelChunk / face distributions / chunk flags
//                                              engine/crates/impact_voxel/src/object.rs:45-221
#pragma once
#include <cstdint>
#include <vector>

#include "orc_math.hpp"

namespace orc {

constexpr int LOG2_CHUNK = 4;
constexpr int CHUNK = 16;
constexpr int CHUNK_VOXELS = 4096;

// lib.rs:75-101
enum : uint8_t {
    F_EMPTY = 1 << 0,
    F_X_DN = 1 << 2,
    F_Y_DN = 1 << 3,
    F_Z_DN = 1 << 4,
    F_X_UP = 1 << 5,
    F_Y_UP = 1 << 6,
    F_Z_UP = 1 << 7,
    F_FULL_ADJ = 0xFC,
};
// lib.rs:317-327: FLAGS[2*dim + side]
inline uint8_t adjacency_flag_for_face(int dim, int side) {
    static const uint8_t f[6] = {F_X_DN, F_X_UP, F_Y_DN, F_Y_UP, F_Z_DN, F_Z_UP};
    return f[2 * dim + side];
}

// lib.rs:154-161 — all constants evaluated in f32 exactly as rustc const-eval does.
constexpr float SD_STEP = 0.02f;
constexpr float SD_INV_STEP = 1.0f / SD_STEP;  // == 50.0f
constexpr float SD_MAX_F32 = SD_STEP * 127.0f;
constexpr float SD_MIN_F32 = SD_STEP * -128.0f;
constexpr int8_t SD_VOID_LIMIT = 100;  // (2.0 * 50.0) as i8

// lib.rs:197-201: `(value * INV) as i8` — Rust `as`: truncate toward zero, saturate, NaN -> 0
inline int8_t sd_from_f32(float v) {
    float s = v * SD_INV_STEP;
    if (s != s) return 0;
    if (s >= 127.0f) return 127;
    if (s <= -128.0f) return -128;
    return (int8_t)(int)s;  // C cast truncates toward zero
}
// lib.rs:220-222
inline float sd_to_f32(int8_t e) { return (float)e * SD_STEP; }
inline bool sd_is_void(int8_t e) { return e > SD_VOID_LIMIT; }

constexpr uint8_t TYPE_DUMMY = 255;  // voxel_types.rs:112-114

// lib.rs:60-66 — #[repr(C)] field order: voxel_type, signed_distance, flags
struct Voxel {
    uint8_t type;
    int8_t sd;
    uint8_t flags;
    bool empty() const { return (flags & F_EMPTY) != 0; }
};
inline Voxel voxel_max_outside() { return {TYPE_DUMMY, 127, F_EMPTY}; }
inline Voxel voxel_max_inside(uint8_t t) { return {t, -128, 0}; }

enum ChunkKind : uint8_t { K_VOID = 0, K_UNIFORM = 1, K_NONUNIFORM = 2 };
enum FaceDist : uint8_t { FD_EMPTY = 0, FD_FULL = 1, FD_MIXED = 2 };  // object.rs:150-161

// object.rs:163-188
enum : uint8_t {
    CF_OBSC_X_DN = 1 << 0,
    CF_OBSC_Y_DN = 1 << 1,
    CF_OBSC_Z_DN = 1 << 2,
    CF_OBSC_X_UP = 1 << 3,
    CF_OBSC_Y_UP = 1 << 4,
    CF_OBSC_Z_UP = 1 << 5,
    CF_ONLY_EMPTY = 1 << 6,
    CF_FULLY_OBSCURED = 0x3F,
};

struct Chunk {
    uint8_t kind = K_VOID;
    uint8_t gen_kind = K_VOID;  // kind right after generation (before uniform demotion)
    Voxel uniform_voxel{};      // Uniform only
    uint32_t data_offset = 0;   // NonUniform only: voxels[data_offset << 12 ..]
    uint8_t face[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    uint8_t flags = 0;
    uint16_t region_count = 0;           // split_detection.rs:82-88
    uint16_t boundary_region_count = 0;
};

struct VoxelObject {
    float extent = 1.0f;
    int cc[3] = {0, 0, 0};  // chunk counts
    int occ_chunk[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int occ_voxel[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    std::vector<Chunk> chunks;
    std::vector<Voxel> voxels;
    std::vector<uint8_t> labels;  // split detector voxel_region_labels, same layout as voxels

    int n_chunks() const { return cc[0] * cc[1] * cc[2]; }
    int cidx(int i, int j, int k) const { return (i * cc[1] + j) * cc[2] + k; }
    // returns chunk or nullptr when outside the chunk grid
    const Chunk* chunk_at(int i, int j, int k) const {
        if (i < 0 || j < 0 || k < 0 || i >= cc[0] || j >= cc[1] || k >= cc[2]) return nullptr;
        return &chunks[cidx(i, j, k)];
    }
    // Voxel at object indices, materialising void/uniform chunks (object.rs get_voxel semantics)
    Voxel voxel_at(int i, int j, int k) const {
        if (i < 0 || j < 0 || k < 0) return voxel_max_outside();
        const Chunk* c = chunk_at(i >> 4, j >> 4, k >> 4);
        if (!c || c->kind == K_VOID) return voxel_max_outside();
        if (c->kind == K_UNIFORM) return c->uniform_voxel;
        return voxels[((size_t)c->data_offset << 12) + (((i & 15) << 8) | ((j & 15) << 4) | (k & 15))];
    }
};

// A ChunkedVoxelGenerator (generation.rs:41-67): fills 4096 voxels for the chunk at `origin`.
struct ChunkSparseness {
    bool only_empty;
    bool is_void;
};
struct Generator {
    virtual ~Generator() {}
    virtual float voxel_extent() const = 0;
    virtual void grid_shape(int out[3]) const = 0;
    virtual ChunkSparseness generate_chunk(Voxel* voxels, const int origin[3]) const = 0;
};

// object.rs:239-244, 307-359
void generate_without_derived_state(VoxelObject& obj, const Generator& gen);
void update_occupied_voxel_ranges(VoxelObject& obj);  // object.rs:1187-1280
void compute_all_derived_state(VoxelObject& obj);     // object.rs:1136-1145
// split_detection.rs:662-891 for one chunk
void update_local_connected_regions_for_chunk(VoxelObject& obj, int chunk_idx);

}  // namespace orc
