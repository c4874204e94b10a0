/*
 * orc.h — CPU ORACLE for the gsx hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain scalar CPU restatement of the reference algorithm
 * (AllIAskOfYou/gaussian-splatting) for the per-frame path
 *   LOD select -> depth key -> view-ordered index sort -> quad raster + blend.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product library (libgsx.so) never links or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Arithmetic is f32 with one rounding per operation
 * (build with -ffp-contract=off) in the operation order of the vendored
 * glm 0.9.9.8, so results are bit-comparable with oracle/_ref (the real
 * reference sources compiled in this container).
 *
 * Pinning status:
 *   data model, keys, sort order, camera matrices, Octree/HC/GridHC cuts,
 *   BB::screen_area  -> pinned against oracle/_ref via tests/golden fixtures.
 *   vs_main / raster / fs_main / blend -> PARITY UNPINNED against the
 *   reference's WebGPU implementation (wgpu-native is not vendored and cannot
 *   run here); the image oracle is this restatement alone.
 */
#ifndef GSX_ORC_H
#define GSX_ORC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- data model (Splat.h:5-29,38-91; ResourceManager.cpp:36-65) ---- */
/* raw32: n records of 32 B {pos f32x3, scale f32x3, rgba u8x4, quat u8x4}.
 * split14: n x 14 floats {pos3, scale3, color4, rot4}. splat20: n x 20 floats
 * {mat4 transform column-major, vec4 color} = the reference's 80-B Splat. */
void orc_raw_to_split(const void* raw32, uint64_t n, float* split14);
void orc_center_split(float* split14, uint64_t n);
void orc_split_to_splat(const float* split14, uint64_t n, float* splat20);

/* ---- sort (SplatMesh.h:84-114; glm/detail/func_geometric.inl:8-23,47-55) ---- */
/* dist_out has n entries (indexed by splat id, like the reference's
 * `distances`); only entries named by idx[0..m) are written. */
void orc_distance_keys(const float* splat20, uint64_t n, const uint32_t* idx,
                       uint64_t m, const float cam[3], float* dist_out);
/* Contract order: distance descending, ties by ascending index. idx may be
 * NULL (= iota over n). out has m entries. */
void orc_sort_contract(const float* splat20, uint64_t n, const uint32_t* idx,
                       uint64_t m, const float cam[3], uint32_t* out);
/* Port of the reference's own routine for timing: per-frame distances alloc,
 * key loop, descending std::sort with the reference's comparator (unstable).
 * Sorts idx_inout in place, returns seconds measured over the region the
 * reference times (SplatMesh.h:88-111). */
double orc_sort_reference_port(const float* splat20, uint64_t n,
                               uint32_t* idx_inout, uint64_t m,
                               const float cam[3]);

/* ---- camera / scene graph (Camera.h:27-34; Node.cpp:10-27;
 *      glm/ext/matrix_clip_space.inl:249-262; glm/detail/func_matrix.inl:294-351) ---- */
void orc_perspective(float fovy_deg, float aspect, float znear, float zfar, float out16[16]);
void orc_mat4_inverse(const float m[16], float out16[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out16[16]);
/* OrbitCamera(distance) child camera under an orbit node rotated by
 * successive Node::rotate(axis, angle) calls: rot3 is the orbit node's
 * accumulated 3x3 rotation (column-major).  Writes the camera world matrix. */
void orc_orbit_world(const float rot3[9], float distance, float world16[16]);
/* Node::rotate (Node.cpp:25-27): rot3 = mat3_cast(angleAxis(angle, axis)) * rot3 */
void orc_node_rotate(float rot3[9], const float axis[3], float angle);

/* ---- LOD cuts ---- */
/* BB::screen_area (BB.hpp:41-77) for a box given as center[3], half[3]. */
float orc_bb_screen_area(const float center[3], const float half[3], const float viewproj[16]);
/* Octree::get_indices (Octree.cpp:189-216) on the BFS-flattened tree: node i
 * owns splat i; first_child[i]/n_children[i] give its contiguous child range
 * (n_children 0 = leaf).  Returns the number of ids written to out (BFS order,
 * which is ascending). */
uint64_t orc_octree_cut(uint32_t n_nodes, const float* center3, const float* half3,
                        const uint32_t* first_child, const uint32_t* n_children,
                        const float viewproj[16], float min_screen_area, uint32_t* out);
/* HC::get_indices_depth (HC.cpp:134-168): roots in list order, child[i*2+k]
 * (UINT32_MAX,UINT32_MAX = leaf).  Output in the reference's queue order. */
uint64_t orc_hc_cut_depth(uint32_t n_roots, const uint32_t* roots, const uint32_t* child2,
                          uint32_t depth, uint32_t* out);
/* HC::get_indices (HC.cpp:100-132): metric = error*weight*(1/dist^2) < thr. */
uint64_t orc_hc_cut_metric(uint32_t n_roots, const uint32_t* roots, const uint32_t* child2,
                           const float* error, const float* weight, const float* pos3,
                           const float cam[3], float threshold, uint32_t* out);
/* Splat::weight (Splat.h:24-28) incl. glm determinant(mat4) operation order. */
float orc_splat_weight(const float* splat20);

/* ---- projection + raster + blend (shader_quads_ordered.wgsl:28-73,120-134,203-210;
 *      main.cpp:324,338-344,359-368; renderer.cpp:162-164) ---- */
typedef struct orc_uniforms {
    float proj[16], view[16], model[16];
    float splatSize, cutOff, fov;
    uint32_t bayerSize;
    float bayerScale;
    float pad[3];
} orc_uniforms; /* 224 B, main.cpp:48-58 */

/* Per-splat vertex stage: writes 4 clip-space vertices (16 floats) per splat
 * in quad-vertex order (-1,-1),(1,-1),(1,1),(-1,1). */
void orc_vs_main(const float* splat20, const orc_uniforms* u, float clip16[16]);
/* Full frame: draws order[0..m) (far->near) into a w x h float RGBA
 * framebuffer cleared to (0.05,0.05,0.05,1), then quantizes to rgba8
 * (round-to-nearest).  rows [y0,y1) only. fb_f32 may be NULL. */
void orc_render(const float* splat20, uint64_t n, const uint32_t* order, uint64_t m,
                const orc_uniforms* u, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                uint8_t* rgba8, float* fb_f32);

#ifdef __cplusplus
}
#endif
#endif
