// gsx_lod.hip — per-frame LOD cuts on flattened trees, producing the selection
// (`indices`) as ASCENDING splat ids in HBM with its size in a device word.
//
// Reference path replaced: Octree::get_indices + BB::screen_area
// (Octree.cpp:189-216, BB.hpp:41-77); HC::get_indices / GridHC::get_indices
// (HC.cpp:100-132, GridHC.cpp:94-111).  The reference walks shared_ptr trees
// breadth first on the CPU; here every node evaluates its own stop predicate,
// then checks its ancestor chain, and a scan compacts the survivors.
//
//   emit[i] = stop[i]  &&  no ancestor a of i has stop[a]
//   octree : stop = is_leaf || screen_area <  thr
//   hc     : stop = is_leaf || error*weight*(1/dist^2) < thr
//
// All f32 arithmetic uses the gsx_mul/gsx_add/... helpers (one rounding per op, no
// FMA contraction) in glm's operation order, so the index SETS are bit-identical to
// the reference's.  The reference emits BFS/queue order; the sort's tie rule
// needs ascending ids, which is what compaction yields.
#include "gsx_internal.h"
#include <cmath>
#include "gsx_radix.hpp"

struct GsxMat4 { float m[16]; };

// BB::screen_area (BB.hpp:59-77); clip = (m0*x + m1*y) + (m2*z + m3*1)
// (glm/detail/type_mat4x4.inl:561-572); glm::min/max as ternaries.
__device__ __forceinline__ float gsx_screen_area(float cx, float cy, float cz, float hx, float hy, float hz,
                                                 const GsxMat4& vp) {
    float mnx = 10000000.0f, mny = 10000000.0f, mxx = -10000000.0f, mxy = -10000000.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // corner order BB.hpp:48-55: x sign = {-,+,+,-,-,+,+,-}, y = {-,-,+,+,-,-,+,+}, z = {-,-,-,-,+,+,+,+}
        const bool xp = (i & 3) == 1 || (i & 3) == 2, yp = (i & 2) != 0, zp = (i & 4) != 0;
        const float px = gsx_add(cx, xp ? hx : -hx), py = gsx_add(cy, yp ? hy : -hy), pz = gsx_add(cz, zp ? hz : -hz);
        float clip[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r == 2) continue;  // z is not used
            const float a0 = gsx_add(gsx_mul(vp.m[0 + r], px), gsx_mul(vp.m[4 + r], py));
            const float a1 = gsx_add(gsx_mul(vp.m[8 + r], pz), gsx_mul(vp.m[12 + r], 1.0f));
            clip[r] = gsx_add(a0, a1);
        }
        if (clip[3] == 0.0f) continue;
        const float nx = gsx_div(clip[0], clip[3]), ny = gsx_div(clip[1], clip[3]);
        mnx = (nx < mnx) ? nx : mnx;
        mny = (ny < mny) ? ny : mny;
        mxx = (mxx < nx) ? nx : mxx;
        mxy = (mxy < ny) ? ny : mxy;
    }
    return gsx_mul(gsx_sub(mxx, mnx), gsx_sub(mxy, mny));
}

__global__ __launch_bounds__(256) void k_octree_stop(const float4* __restrict__ a, const float4* __restrict__ b,
                                                     const uint8_t* __restrict__ leaf, uint32_t n, GsxMat4 vp, float thr,
                                                     uint8_t* __restrict__ stop) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 ca = a[i], cb = b[i];
    const float area = gsx_screen_area(ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, vp);
    stop[i] = (leaf[i] || (area < thr)) ? 1 : 0;  // Octree.cpp:203
}

__global__ __launch_bounds__(256) void k_hc_stop(const float4* __restrict__ ew, const float4* __restrict__ pos4,
                                                 const uint8_t* __restrict__ leaf, uint32_t n, float cx, float cy, float cz,
                                                 float thr, uint8_t* __restrict__ stop) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos4[i], e = ew[i];
    // HC.cpp:112-119: length(splat_pos - camera_pos); dist = 1/(d*d); metric = error*weight*dist
    const float dx = gsx_sub(p.x, cx), dy = gsx_sub(p.y, cy), dz = gsx_sub(p.z, cz);
    const float d = gsx_sqrt(gsx_add(gsx_add(gsx_mul(dx, dx), gsx_mul(dy, dy)), gsx_mul(dz, dz)));
    const float dist = gsx_div(1.0f, gsx_mul(d, d));
    const float metric = gsx_mul(gsx_mul(e.x, e.y), dist);
    stop[i] = (metric < thr || leaf[i]) ? 1 : 0;  // HC.cpp:120
}

// ---- octree in BFS order: top-down, level by level -------------------------------------------------------------
// The reference's breadth-first walk only ever LOOKS at a node whose ancestors all failed the stop test.  The stop test
// is the expensive part (8 corners x 3 matrix rows and 16 IEEE divisions: ~430 instructions per node), so evaluating all
// 8.5 M nodes of cfg 3's tree and then walking ancestor chains is VALU-bound at 6x the necessary work (a cut of 1.1 M
// nodes looks at 1.4 M).  In BFS order the children of a node are contiguous ids, so the walk is done as the reference
// does it, with a queue per level: a looked-at node either stops (its emit byte gets the frame's stamp) or appends its
// children to the next level's queue (one atomic per wave).  Queues are dense, so every lane of a wave evaluates a node
// that needs it, and nothing is ever read for a node that is not reached.  The leading levels (<= 4096 nodes) are
// handled by ONE workgroup: it evaluates all of them at once (no dependence between levels there) and then resolves
// reachability in LDS, so a frame pays one launch and one memory round trip for them instead of five.
// emit bytes carry a frame stamp (1..255) and compaction matches it, so the array is cleared every 255 cuts, not per cut.
// `screen_area < thr` decided from a cheap evaluation with an error bound; false = too close to call (or not finite, or a
// corner near the camera plane): the caller then evaluates gsx_screen_area, so the decision is ALWAYS the exactly rounded
// one.  The exact evaluation is ~430 instructions (8 corners x 3 matrix rows in glm's order, 16 IEEE divisions); this one
// ~150: a corner's clip coordinates are base +- ax +- ay +- az per row (fused multiply-adds, shared partial sums), one
// v_rcp_f32 per corner.  Bound: both evaluations of a row are within 6 roundings x 2^-24 x S of the real value, S = the
// sum of the row's absolute terms, so they differ by < 2^-20 S; the quotients then differ by (d0 + |n| d3) / min|w| plus
// 4 ulp for the reciprocal and the product; the area by the usual product rule.  The whole margin is doubled again.
// The exact code clamps at +-1e7 (its min/max start values): coordinates beyond 1e6 go to the exact evaluation.
__device__ __forceinline__ bool gsx_area_below_fast(float cx, float cy, float cz, float hx, float hy, float hz, const GsxMat4& vp,
                                                    float thr, bool* below) {
    float x[8], y[8], w[8], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int r = k == 2 ? 3 : k;                   // rows 0, 1, 3 (z is not used)
        const float m0 = vp.m[r], m1 = vp.m[4 + r], m2 = vp.m[8 + r], m3 = vp.m[12 + r];
        const float base = __builtin_fmaf(m0, cx, __builtin_fmaf(m1, cy, __builtin_fmaf(m2, cz, m3)));
        const float ax = m0 * hx, ay = m1 * hy, az = m2 * hz;
        const float sum = __builtin_fmaf(fabsf(m0), fabsf(cx), __builtin_fmaf(fabsf(m1), fabsf(cy), __builtin_fmaf(fabsf(m2), fabsf(cz), fabsf(m3)))) +
                          (fabsf(ax) + fabsf(ay) + fabsf(az));
        d[k] = sum * 9.5367431640625e-07f;              // 2^-20 S
        float* out = k == 0 ? x : (k == 1 ? y : w);
        const float b0 = base - ax, b1 = base + ax;
        const float c0 = b0 - ay, c1 = b0 + ay, c2 = b1 - ay, c3 = b1 + ay;
        out[0] = c0 - az; out[1] = c0 + az; out[2] = c1 - az; out[3] = c1 + az;
        out[4] = c2 - az; out[5] = c2 + az; out[6] = c3 - az; out[7] = c3 + az;
    }
    float mnx, mxx, mny, mxy, wmin;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float rw = __builtin_amdgcn_rcpf(w[i]);
        const float nx = x[i] * rw, ny = y[i] * rw, aw = fabsf(w[i]);
        if (i == 0) { mnx = mxx = nx; mny = mxy = ny; wmin = aw; }
        else { mnx = fminf(mnx, nx); mxx = fmaxf(mxx, nx); mny = fminf(mny, ny); mxy = fmaxf(mxy, ny); wmin = fminf(wmin, aw); }
    }
    const float wlo = wmin - d[2];
    const float nxm = fmaxf(fabsf(mnx), fabsf(mxx)), nym = fmaxf(fabsf(mny), fabsf(mxy));
    const float iw = __builtin_amdgcn_rcpf(wlo) * 1.0001f;
    const float ex = __builtin_fmaf(nxm, d[2], d[0]) * iw + nxm * 4.76837158203125e-07f;   // + 2^-21 |n|
    const float ey = __builtin_fmaf(nym, d[2], d[1]) * iw + nym * 4.76837158203125e-07f;
    const float dx = mxx - mnx, dy = mxy - mny, area = dx * dy;
    const float margin = 4.0f * (ex * (dy + ey) + ey * (dx + ex)) + area * 4.76837158203125e-07f;
    // (every comparison is false for a NaN: not decided)
    if (!(wlo > 1.0e-30f) || !(nxm < 1.0e6f) || !(nym < 1.0e6f) || !(fabsf(area - thr) > margin)) return false;
    *below = area < thr;
    return true;
}

// One 32-byte record per node for the walk kernels (a node costs one line request instead of four: box centre + half x,
// half y / z, child word, leaf flag); k_octree_stop and the per-level kernels keep the separate arrays.
struct GsxOctNode { float cx, cy, cz, hx, hy, hz; uint32_t child, leaf; };
static_assert(sizeof(GsxOctNode) == 32, "two 16-byte loads");
__device__ __forceinline__ GsxOctNode gsx_load_node(const GsxOctNode* __restrict__ nodes, uint32_t i) {
    const uint4 lo = reinterpret_cast<const uint4*>(nodes)[2u * i], hi = reinterpret_cast<const uint4*>(nodes)[2u * i + 1u];
    GsxOctNode nd;
    nd.cx = __uint_as_float(lo.x); nd.cy = __uint_as_float(lo.y); nd.cz = __uint_as_float(lo.z); nd.hx = __uint_as_float(lo.w);
    nd.hy = __uint_as_float(hi.x); nd.hz = __uint_as_float(hi.y); nd.child = hi.z; nd.leaf = hi.w;
    return nd;
}
__device__ __forceinline__ bool gsx_node_stops(const GsxOctNode& nd, const GsxMat4& vp, float thr) {
    bool below;
    if (!gsx_area_below_fast(nd.cx, nd.cy, nd.cz, nd.hx, nd.hy, nd.hz, vp, thr, &below))
        below = gsx_screen_area(nd.cx, nd.cy, nd.cz, nd.hx, nd.hy, nd.hz, vp) < thr;
    return nd.leaf != 0u || below;    // Octree.cpp:203
}

__device__ __forceinline__ bool gsx_octree_node_stops(const float4* __restrict__ a, const float4* __restrict__ b,
                                                      const uint8_t* __restrict__ leaf, uint32_t i, const GsxMat4& vp, float thr) {
    const float4 ca = a[i], cb = b[i];
    bool below;
    if (!gsx_area_below_fast(ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, vp, thr, &below))
        below = gsx_screen_area(ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, vp) < thr;
    return leaf[i] || below;          // Octree.cpp:203
}

struct GsxLodLevels { uint32_t start[GSX_LOD_MAX_LEVELS + 1]; };

// Leading levels (<= GSX_LOD_TOP_NODES nodes, <= 8 levels) in ONE launch without any dependence between threads: eight
// adjacent lanes serve one top node and test the node itself (slot 0) and each of its ancestors (slots 1..7, ids from a
// table built at set-up) at the same time - redundant work (every ancestor is tested once per descendant), but the whole
// top of the tree costs one memory round trip and one stop test instead of one of each per level.  A node whose ancestors
// all fail the test is reached: it is emitted if it stops, and in the last top level it queues its children otherwise.
#define GSX_TOP_NT 256
__global__ __launch_bounds__(GSX_TOP_NT) void k_octree_top(const float4* __restrict__ a, const float4* __restrict__ b,
                                                           const uint8_t* __restrict__ leaf, const uint32_t* __restrict__ child,
                                                           const uint32_t* __restrict__ top_anc, uint32_t top_end, uint32_t last_begin,
                                                           uint32_t top_levels, GsxMat4 vp, float thr, uint8_t stamp,
                                                           uint8_t* __restrict__ emit, uint32_t* __restrict__ queue_out,
                                                           uint32_t* __restrict__ qcount) {
    __shared__ uint32_t s_tmp[GSX_TOP_NT / 64];
    __shared__ uint32_t s_base;
    const uint32_t gid = blockIdx.x * GSX_TOP_NT + threadIdx.x, node = gid >> 3, slot = gid & 7u, lane = threadIdx.x & 63u;
    const uint32_t id = node < top_end ? top_anc[gid] : 0xFFFFFFFFu;
    const bool stops = id != 0xFFFFFFFFu && gsx_octree_node_stops(a, b, leaf, id, vp, thr);
    const uint32_t group = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(stops) >> (lane & ~7u)) & 0xFFu;   // bit s = slot s stops
    uint32_t fc = 0, nc = 0;
    if (slot == 0u && node < top_end && (group & 0xFEu) == 0u) {       // reached: no ancestor stops
        if (group & 1u) emit[node] = stamp;
        else if (node >= last_begin) { const uint32_t cw = child[node]; fc = cw >> 4; nc = cw & 15u; }
    }
    uint32_t total;
    const uint32_t excl = gsx_block_excl_scan<GSX_TOP_NT>(nc, s_tmp, &total);
    if (total == 0u) return;                            // (uniform)
    if (threadIdx.x == 0) s_base = atomicAdd(&qcount[top_levels], total);
    __syncthreads();
    const uint32_t base = s_base + excl;
    for (uint32_t k = 0; k < nc; ++k) queue_out[base + k] = fc + k;
}

// One level: thread t looks at node queue_in[t].  NT = 256 for narrow levels (more CUs busy), 1024 for wide ones (one slot
// reservation - a same-address atomic, ~10 ns each at the L2 - per 1024 nodes).
template <int NT>
__global__ __launch_bounds__(NT) void k_octree_level(const float4* __restrict__ a, const float4* __restrict__ b,
                                                     const uint8_t* __restrict__ leaf, const uint32_t* __restrict__ child,
                                                     uint32_t level, GsxMat4 vp, float thr, uint8_t stamp,
                                                     uint8_t* __restrict__ emit, const uint32_t* __restrict__ queue_in,
                                                     uint32_t* __restrict__ queue_out, uint32_t* __restrict__ qcount) {
    __shared__ uint32_t s_tmp[NT / 64];
    __shared__ uint32_t s_base;
    const uint32_t cnt = qcount[level];
    if (blockIdx.x * NT >= cnt) return;                 // the whole workgroup is past the queue
    const uint32_t t = blockIdx.x * NT + threadIdx.x;
    uint32_t fc = 0, nc = 0;
    if (t < cnt) {
        const uint32_t i = queue_in[t];
        const uint32_t cw = child[i];
        if (gsx_octree_node_stops(a, b, leaf, i, vp, thr)) emit[i] = stamp;
        else { fc = cw >> 4; nc = cw & 15u; }
    }
    uint32_t total;
    const uint32_t excl = gsx_block_excl_scan<NT>(nc, s_tmp, &total);
    if (total == 0u) return;                            // (uniform)
    if (threadIdx.x == 0) s_base = atomicAdd(&qcount[level + 1], total);
    __syncthreads();
    const uint32_t base = s_base + excl;
    for (uint32_t k = 0; k < nc; ++k) queue_out[base + k] = fc + k;
}

// ---- the same walk in two launches (round 3) -------------------------------------------------------------------
// One launch per level costs four dependent trips to memory (queue length, queue entry, node, slot reservation) plus the
// launch itself: ~9.5 us a level whatever its width.  Inside a workgroup a level costs one trip (the node) and a block scan.
//  k_octree_head: the leading levels as in k_octree_top, eight top nodes per workgroup, and in the same breath the two
//                 levels below them (<= 64 and <= 512 nodes, contiguous id ranges in BFS order, found at set-up): every
//                 thread tests one node at once, who is reached is settled afterwards in LDS (tests of unreached nodes are
//                 wasted, but the levels next to the top are nearly always reached, and three dependent rounds of
//                 load + test become one); the frontier below goes to the global queue, which deals it out evenly
//                 again - that is where the subtrees' sizes start to differ.
//  k_octree_tail: a workgroup takes 256 entries of that queue and walks everything below them from a stack in LDS, deepest
//                 nodes first.  The stack is ordered by level (children are pushed above everything that stays), a level's
//                 entries are only pushed when all deeper ones fit the same pop, so a level never holds more than 8 x 256
//                 entries: capacity = the queue entries + 8 x 256 x (levels below that are pushed), no spill path needed
//                 (the deepest level is not pushed: its nodes have no children, reached means emitted).  The entries are dealt
//                 out in groups of 8 (a cache line of boxes), round-robin over the workgroups: where the cut runs deep,
//                 EVERY node of a region expands eightfold, and a workgroup holding 256 neighbours walked 9 rounds one
//                 after another while 60 % of the others were done after one (58 us at cfg 3); dealt out, every
//                 workgroup walks 2 to 4 rounds (tools/_dbg/lod_tail_sim.py).
#define GSX_HEAD_NT 640          // 64 slots of the top nodes + 64 nodes of the next level + 512 of the one after
#define GSX_HEAD_G 8u            // top nodes per workgroup: x 8 slots = wave 0
__global__ __launch_bounds__(GSX_HEAD_NT) void k_octree_head(const GsxOctNode* __restrict__ nodes,
                                                             const uint32_t* __restrict__ parent, const uint32_t* __restrict__ top_anc,
                                                             const uint4* __restrict__ ranges, uint32_t top_end, uint32_t last_begin,
                                                             uint32_t walk, uint32_t out_level, GsxMat4 vp, float thr, uint8_t stamp,
                                                             uint8_t* __restrict__ emit, uint32_t* __restrict__ queue_out,
                                                             uint32_t* __restrict__ qcount) {
    __shared__ uint32_t s_tmp[GSX_HEAD_NT / 64];
    __shared__ uint32_t s_base;
    __shared__ uint8_t s_open4[8], s_open5[64];
    const uint32_t tid = threadIdx.x, role = tid < 64u ? 0u : (tid < 128u ? 1u : 2u);   // (uniform per wave)
    const uint4 rg = ranges[blockIdx.x];                // descendants one and two levels below this workgroup's top nodes: (first, count) x 2
    // Nothing below waits for a test's outcome before it starts the next test: the ids of the descendants are known from the
    // table, every thread tests its node at once, and who is reached is settled afterwards in LDS.
    uint32_t id = 0xFFFFFFFFu, pl = 0, cw = 0;
    bool reached = false, stops = false;
    if (role == 0u) {                                   // a top node and its ancestors in eight adjacent lanes
        const uint32_t gid = blockIdx.x * 64u + tid, node = gid >> 3, slot = gid & 7u;
        const uint32_t anc = node < top_end ? top_anc[gid] : 0xFFFFFFFFu;
        bool st = false;
        if (anc != 0xFFFFFFFFu) {
            const GsxOctNode nd = gsx_load_node(nodes, anc);
            if (slot == 0u) cw = nd.child;
            st = gsx_node_stops(nd, vp, thr);
        }
        const uint32_t group = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(st) >> (tid & ~7u)) & 0xFFu;
        if (slot == 0u) {
            if (node < top_end) { id = node; reached = (group & 0xFEu) == 0u; stops = (group & 1u) != 0u; }
            s_open4[tid >> 3] = (reached && !stops && node >= last_begin) ? 1 : 0;
        }
    } else {
        const uint32_t j = tid - (role == 1u ? 64u : 128u), cnt = role == 1u ? rg.y : rg.w;
        if (j < cnt) {
            id = (role == 1u ? rg.x : rg.z) + j;
            pl = parent[id] - (role == 1u ? blockIdx.x * GSX_HEAD_G : rg.x);
            const GsxOctNode nd = gsx_load_node(nodes, id);
            cw = nd.child;
            stops = gsx_node_stops(nd, vp, thr);
        }
    }
    __syncthreads();
    if (role == 1u && id != 0xFFFFFFFFu) {
        reached = s_open4[pl & 7u] != 0;
        s_open5[tid - 64u] = (reached && !stops) ? 1 : 0;
    }
    __syncthreads();
    if (role == 2u && id != 0xFFFFFFFFu) reached = s_open5[pl & 63u] != 0;
    uint32_t fc = 0, nc = 0;
    if (id != 0xFFFFFFFFu && reached) {
        if (stops) emit[id] = stamp;
        else if (role == walk && (role != 0u || id >= last_begin)) { fc = cw >> 4; nc = cw & 15u; }   // the frontier: to the global queue
    }
    uint32_t total;
    const uint32_t excl = gsx_block_excl_scan<GSX_HEAD_NT>(nc, s_tmp, &total);
    if (total == 0u) return;                            // (uniform)
    if (tid == 0) s_base = atomicAdd(&qcount[out_level], total);
    __syncthreads();
    const uint32_t base = s_base + excl;
    for (uint32_t k = 0; k < nc; ++k) queue_out[base + k] = fc + k;
}

#define GSX_TAIL_NT 256
#define GSX_TAIL_MAX_DEPTH 6u    // levels below the queue's that get pushed: 8 x 256 x 6 entries = 48 KB of LDS
template <int TNT>
__global__ __launch_bounds__(TNT) void k_octree_tail(const GsxOctNode* __restrict__ nodes, uint32_t n, uint32_t level, uint32_t last_begin /* first id of the deepest level */,
                                                             uint32_t cap, GsxMat4 vp, float thr, uint8_t stamp,
                                                             uint8_t* __restrict__ emit, const uint32_t* __restrict__ queue_in,
                                                             const uint32_t* __restrict__ qcount, GsxDevFlags* __restrict__ flags) {
    extern __shared__ uint32_t s_stack[];
    __shared__ uint32_t s_tmp[TNT / 64];
    const uint32_t cnt = qcount[level] < n ? qcount[level] : n, tid = threadIdx.x;
    const uint32_t groups = (cnt + 7u) / 8u;
    // the grid is what the GPU holds at once (the host knows the level's width, not the queue's length): the workgroups that
    // have work share the groups evenly, round-robin
    const uint32_t want = (cnt + TNT - 1u) / TNT, wgs = want < gridDim.x ? want : gridDim.x;
    if (blockIdx.x >= wgs) return;                      // the whole workgroup is past the queue
    const uint32_t mine = (groups - blockIdx.x + wgs - 1u) / wgs;        // groups blockIdx.x, blockIdx.x + wgs, ...
    uint32_t sp = mine * 8u;                            // entries on the stack (uniform); the queue entries go on it first
    if (sp > cap) {                                     // (cannot happen: the host sizes the stack for the level's width)
        if (tid == 0) atomicOr(&flags->lod_fault, 2u);
        return;
    }
    for (uint32_t e = tid; e < sp; e += TNT) {
        const uint32_t src = ((e >> 3) * wgs + blockIdx.x) * 8u + (e & 7u);
        s_stack[e] = src < cnt ? queue_in[src] : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (;;) {
        const uint32_t np = sp < TNT ? sp : TNT;
        sp -= np;
        uint32_t i = tid < np ? s_stack[sp + tid] : 0xFFFFFFFFu;   // (the scan's barriers stand between this read and the next push)
        uint32_t fc = 0, nc = 0;
        if (i != 0xFFFFFFFFu && i >= n) {               // (a queue entry that is no node: never follow it)
            atomicOr(&flags->lod_fault, 4u);
            i = 0xFFFFFFFFu;
        }
        if (i != 0xFFFFFFFFu) {
            const GsxOctNode nd = gsx_load_node(nodes, i);
            if (gsx_node_stops(nd, vp, thr)) emit[i] = stamp;
            else { fc = nd.child >> 4; nc = nd.child & 15u; }
        }
        if (nc != 0u && fc >= last_begin) {             // children in the deepest level are leaves (checked at set-up): reached means emitted
            for (uint32_t k = 0; k < nc; ++k)
                if (fc + k < n) emit[fc + k] = stamp;
            nc = 0u;
        }
        uint32_t total;
        const uint32_t pos = sp + gsx_block_excl_scan<TNT>(nc, s_tmp, &total);
        for (uint32_t k = 0; k < nc; ++k)
            if (pos + k < cap) s_stack[pos + k] = fc + k;
        sp += total;
        if (sp > cap) {                                 // (cannot happen, see above; never write past the stack)
            if (tid == 0) atomicOr(&flags->lod_fault, 2u);
            sp = cap;
        }
        if (sp == 0u) return;                           // (uniform)
        __syncthreads();
    }
}

// emit[i] = stop[i] && no ancestor stops.  Chains are as deep as the tree; most
// walks end after a step or two because an ancestor near the cut stops.
__global__ __launch_bounds__(256) void k_emit_flags(const uint8_t* __restrict__ stop, const uint32_t* __restrict__ parent,
                                                    uint32_t n, uint8_t* __restrict__ emit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t e = stop[i];
    uint32_t a = parent[i];
    while (e && a != 0xFFFFFFFFu) {
        if (stop[a]) e = 0;
        a = parent[a];
    }
    emit[i] = e;
}

// ---- flag compaction: emit bytes -> ascending id list + count ----
#define GSX_CMP_NT 256
#define GSX_CMP_IPT 16
#define GSX_CMP_TILE (GSX_CMP_NT * GSX_CMP_IPT)

__device__ __forceinline__ uint32_t gsx_load_flags16(const uint8_t* flags, uint32_t base, uint32_t n, uint32_t match, uint32_t w[4]) {
    // 16 flag bytes per thread; flags buffers are padded to a multiple of 16 bytes
    uint32_t cnt = 0;
    if (base < n) {
        const uint4 v = *reinterpret_cast<const uint4*>(flags + base);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool on = ((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) == match && (base + j) < n;
            cnt += on ? 1u : 0u;
        }
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
    }
    return cnt;
}

__global__ __launch_bounds__(GSX_CMP_NT) void k_compact_count(const uint8_t* __restrict__ flags, uint32_t n, uint32_t match,
                                                              uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_tmp[GSX_CMP_NT / 64];
    uint32_t w[4];
    const uint32_t base = blockIdx.x * GSX_CMP_TILE + threadIdx.x * GSX_CMP_IPT;
    const uint32_t cnt = gsx_load_flags16(flags, base, n, match, w);
    uint32_t total;
    gsx_block_excl_scan<GSX_CMP_NT>(cnt, s_tmp, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: in-place exclusive scan of block_sums[0..nb), total -> *m_out
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ block_sums, uint32_t nb,
                                                          uint32_t* __restrict__ m_out, uint32_t* __restrict__ qcount) {
    if (qcount && threadIdx.x <= GSX_LOD_MAX_LEVELS + 1) qcount[threadIdx.x] = 0u;   // the octree cut's level counters, for the next cut
    __shared__ uint32_t s_tmp[1024 / 64];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < nb) ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = gsx_block_excl_scan<1024>(v, s_tmp, &total);
        const uint32_t carry = s_carry;
        if (i < nb) block_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *m_out = s_carry;
}

__global__ __launch_bounds__(GSX_CMP_NT) void k_compact_write(const uint8_t* __restrict__ flags, uint32_t n, uint32_t match,
                                                              const uint32_t* __restrict__ block_offs,
                                                              uint32_t* __restrict__ out) {
    __shared__ uint32_t s_tmp[GSX_CMP_NT / 64];
    uint32_t w[4];
    const uint32_t base = blockIdx.x * GSX_CMP_TILE + threadIdx.x * GSX_CMP_IPT;
    const uint32_t cnt = gsx_load_flags16(flags, base, n, match, w);
    uint32_t pos = block_offs[blockIdx.x] + gsx_block_excl_scan<GSX_CMP_NT>(cnt, s_tmp, nullptr);
    if (cnt == 0) return;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool on = ((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) == match && (base + j) < n;
        if (on) out[pos++] = base + j;
    }
}

// The same compaction in ONE launch (round 3): workgroups of 512 threads, each owning up to 8 tiles of 512 x 16 consecutive
// flags (a thread: one 16-byte unit per tile, kept in registers as match masks).  A workgroup counts its span, publishes the
// count as (cut number << 32 | count) with a device-scope store, sums the counts of the workgroups before it as they
// appear, and writes its ids.  No chain: a workgroup waits for counts only, and every workgroup publishes its count before
// it waits for anything; it waits for LOWER-numbered workgroups only, which the dispatcher starts first, and the wait is
// bounded all the same (on time-out the cut is flagged as failed instead of hanging the grid).  Measured on its own
// (tools/compactbench.hip, 8.48 M flags, 17 % set): 12.7 us against 18.4 us for the three launches; of that a workgroup
// spends ~1.5 us loading, ~1.5 us in its scans, ~3 us until the counts before it are visible and < 1 us writing.  Ids
// staged through LDS for whole-line stores were slower (15-16 us), a thread owning consecutive units (one scan) as well
// (14.5-16 us), 256- and 1024-thread workgroups too.  More than 768 x 512 x 8 x 16 flags (50 M) go through the three launches.
#define GSX_CP1_NT 512
#define GSX_CP1_MAXG 768u
#define GSX_CP1_MAXU 8u
#define GSX_CP1_SPIN_LIMIT (1u << 20)
__global__ __launch_bounds__(GSX_CP1_NT) void k_compact_onepass(const uint8_t* __restrict__ flags, uint32_t n, uint32_t match,
                                                                uint32_t upt /* tiles per workgroup, <= 8 */, uint64_t* __restrict__ agg,
                                                                uint32_t seq, uint32_t* __restrict__ out, uint32_t* __restrict__ m_out,
                                                                uint32_t* __restrict__ qcount, GsxDevFlags* __restrict__ dflags) {
    __shared__ uint32_t s_tmp[GSX_CP1_NT / 64];
    __shared__ uint32_t s_prefix;
    const uint32_t tid = threadIdx.x, units = (n + 15u) / 16u;
    if (blockIdx.x == 0 && qcount && tid <= GSX_LOD_MAX_LEVELS + 1) qcount[tid] = 0u;   // the octree cut's level counters, for the next cut
    // per 4 flag bytes one word t with bit 8 b + 7 set where byte b matches (exact per byte: no borrow between bytes)
    uint32_t t[GSX_CP1_MAXU][4], tile_cnt[GSX_CP1_MAXU];
    const uint32_t match4 = match * 0x01010101u;
#pragma unroll
    for (uint32_t k = 0; k < GSX_CP1_MAXU; ++k) {
        const uint32_t u = (blockIdx.x * upt + k) * GSX_CP1_NT + tid;           // this thread's unit of tile k
        uint4 w = make_uint4(match4 ^ 0x01010101u, match4 ^ 0x01010101u, match4 ^ 0x01010101u, match4 ^ 0x01010101u);   // (matches nowhere)
        if (k < upt && u < units) w = *reinterpret_cast<const uint4*>(flags + static_cast<uint64_t>(u) * 16u);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t y = ww[q] ^ match4;
            t[k][q] = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
        }
        if (k < upt && u * 16u + 16u > n) {             // the unit that holds flag n - 1 (and those behind): bytes from n on never count
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t first = u * 16u + 4u * q, valid = first >= n ? 0u : (n - first >= 4u ? 4u : n - first);
                t[k][q] &= valid >= 4u ? 0xFFFFFFFFu : ((1u << (8u * valid)) - 1u);
            }
        }
        tile_cnt[k] = __popc(t[k][0]) + __popc(t[k][1]) + __popc(t[k][2]) + __popc(t[k][3]);
    }
    uint32_t total = 0, tile_excl[GSX_CP1_MAXU];
#pragma unroll
    for (uint32_t k = 0; k < GSX_CP1_MAXU; ++k) {
        tile_excl[k] = 0;
        if (k < upt) {                                  // (uniform)
            uint32_t tile_total;
            tile_excl[k] = total + gsx_block_excl_scan<GSX_CP1_NT>(tile_cnt[k], s_tmp, &tile_total);
            total += tile_total;
        }
    }
    if (tid == 0) __hip_atomic_store(&agg[blockIdx.x], (static_cast<uint64_t>(seq) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64u) {
        uint32_t sum = 0;
        bool late = false;
        for (uint32_t p = tid; p < blockIdx.x; p += 64u) {
            uint64_t v = __hip_atomic_load(&agg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t spins = 0; static_cast<uint32_t>(v >> 32) != seq && spins < GSX_CP1_SPIN_LIMIT; ++spins) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(&agg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (static_cast<uint32_t>(v >> 32) != seq) late = true;
            else sum += static_cast<uint32_t>(v);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (late) atomicOr(&dflags->lod_fault, 1u);
        if (tid == 0) s_prefix = sum;
    }
    __syncthreads();
    const uint32_t run = s_prefix;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *m_out = run + total;
    if (run + total > n) return;                        // (only after a timed-out wait: never write past the list)
#pragma unroll
    for (uint32_t k = 0; k < GSX_CP1_MAXU; ++k) {
        if (k < upt) {
            const uint32_t u = (blockIdx.x * upt + k) * GSX_CP1_NT + tid;
            uint32_t pos = run + tile_excl[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t bits = t[k][q];
                while (bits) {                          // one turn per set flag (the wave: as many as its fullest lane has)
                    const uint32_t b = static_cast<uint32_t>(__ffs(static_cast<int>(bits))) - 1u;
                    out[pos++] = u * 16u + 4u * q + (b >> 3);
                    bits &= bits - 1u;
                }
            }
        }
    }
}

// ids whose flag byte equals `match`, ascending
static int compact(gsx_ctx* c, uint32_t n, uint32_t match = 1u) {
    const uint32_t units = gsx_div_up(n, 16u);
    const uint32_t upt = units ? gsx_div_up(units, GSX_CP1_NT * GSX_CP1_MAXG) : 1u;      // tiles of 512 units per workgroup
    if (!c->opt_lod_walk && upt <= GSX_CP1_MAXU) {
        const uint32_t g = units ? gsx_div_up(units, GSX_CP1_NT * upt) : 1u;
        if (++c->lod_seq == 0u) {                       // cut numbers wrapped: forget the old counts
            GSX_HIP(c, hipMemsetAsync(c->lod_agg, 0, GSX_CP1_MAXG * sizeof(uint64_t), c->sel_stream));
            c->lod_seq = 1u;
        }
        hipLaunchKernelGGL(k_compact_onepass, dim3(g), dim3(GSX_CP1_NT), 0, c->sel_stream, c->lod_emit, n, match, upt, c->lod_agg,
                           c->lod_seq, c->sel_idx, c->m_dev, c->lod_qcount, c->flags);
        GSX_HIP(c, hipGetLastError());
        return GSX_OK;
    }
    const uint32_t nb = gsx_div_up(n, GSX_CMP_TILE);
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(GSX_CMP_NT), 0, c->sel_stream, c->lod_emit, n, match, c->lod_block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, c->sel_stream, c->lod_block_sums, nb, c->m_dev, c->lod_qcount);
    hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(GSX_CMP_NT), 0, c->sel_stream, c->lod_emit, n, match, c->lod_block_sums,
                       c->sel_idx);
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

int gsx_launch_select_octree(gsx_ctx* c, const float vp[16], float thr) {
    const uint32_t n = c->lod_nodes;
    GsxMat4 m;
    for (int i = 0; i < 16; ++i) m.m[i] = vp[i];
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SEL0], c->sel_stream));
    uint32_t match = 1u;
    if (c->lod_levels && c->opt_lod_walk != 1) {
        GsxLodLevels lv;
        for (uint32_t l = 0; l <= GSX_LOD_MAX_LEVELS; ++l) lv.start[l] = l <= c->lod_levels ? c->lod_level_start[l] : n;
        if (c->lod_stamp >= 255u) {     // stamps are used up (or another path wrote flags into the bytes): start over
            GSX_HIP(c, hipMemsetAsync(c->lod_emit, 0, (static_cast<uint64_t>(n) + 15) / 16 * 16, c->sel_stream));
            c->lod_stamp = 0;
        }
        match = ++c->lod_stamp;
        const uint32_t top = c->lod_top_levels;
        uint32_t l = top;                               // first level that is walked from the global queue
        if (c->opt_lod_walk == 0 && c->lod_max_children <= 8u) {
            const uint32_t walk = c->lod_levels - top < 2u ? c->lod_levels - top : 2u;
            l = top + walk;
            hipLaunchKernelGGL(k_octree_head, dim3(gsx_div_up(lv.start[top], GSX_HEAD_G)), dim3(GSX_HEAD_NT), 0, c->sel_stream,
                               reinterpret_cast<const GsxOctNode*>(c->lod_node), c->lod_parent, c->lod_top_anc, c->lod_head_ranges, lv.start[top], lv.start[top - 1u], walk, l, m, thr,
                               static_cast<uint8_t>(match), c->lod_emit, c->lod_queue[l & 1u], c->lod_qcount);
        } else {
            hipLaunchKernelGGL(k_octree_top, dim3(gsx_div_up(lv.start[top] * 8u, GSX_TOP_NT)), dim3(GSX_TOP_NT), 0, c->sel_stream, c->lod_a, c->lod_b,
                               c->lod_leaf, c->lod_child, c->lod_top_anc, lv.start[top], lv.start[top - 1u], top, m, thr,
                               static_cast<uint8_t>(match), c->lod_emit, c->lod_queue[top & 1u], c->lod_qcount);
        }
        for (; l < c->lod_levels; ++l) {
            // (the queue's length is only known on the device: the grid covers the level, workgroups past the queue leave at once)
            const uint32_t width = lv.start[l + 1] - lv.start[l];
            // levels below this one whose nodes go on the stack: the deepest level's are emitted when their parent is expanded
            const uint32_t below = c->lod_levels - 1u - l, pushed = (below && c->lod_deepest_leaf) ? below - 1u : below;
            if (c->opt_lod_walk == 0 && c->lod_max_children <= 8u && pushed <= GSX_TAIL_MAX_DEPTH) {   // everything from here down in one launch
                // one workgroup per 256 nodes of the level.  (A grid bounded by what the GPU holds at once, every workgroup taking
                // its share of the queue, measured slower: 29.7 against 24.9 us - finished workgroups make room as they go.)
                // (128- and 512-thread workgroups measured the same and slower: 20.8 / 21.1 / 23.4 us)
                const uint32_t grid = gsx_div_up(width, GSX_TAIL_NT);
                const uint32_t cap = std::max<uint32_t>(GSX_TAIL_NT, gsx_div_up(gsx_div_up(width, 8u), grid) * 8u) + 8u * GSX_TAIL_NT * pushed;
                if (cap * sizeof(uint32_t) <= (64u << 10)) {
                    hipLaunchKernelGGL(k_octree_tail<GSX_TAIL_NT>, dim3(grid), dim3(GSX_TAIL_NT), cap * sizeof(uint32_t), c->sel_stream,
                                       reinterpret_cast<const GsxOctNode*>(c->lod_node), n, l, c->lod_deepest_leaf ? lv.start[c->lod_levels - 1u] : n, cap,
                                       m, thr, static_cast<uint8_t>(match), c->lod_emit, c->lod_queue[l & 1u], c->lod_qcount, c->flags);
                    break;
                }
            }
            if (width <= (256u << 10))
                hipLaunchKernelGGL(k_octree_level<256>, dim3(gsx_div_up(width, 256)), dim3(256), 0, c->sel_stream, c->lod_a, c->lod_b,
                                   c->lod_leaf, c->lod_child, l, m, thr, static_cast<uint8_t>(match), c->lod_emit, c->lod_queue[l & 1u],
                                   c->lod_queue[(l + 1u) & 1u], c->lod_qcount);
            else
                hipLaunchKernelGGL(k_octree_level<1024>, dim3(gsx_div_up(width, 1024)), dim3(1024), 0, c->sel_stream, c->lod_a, c->lod_b,
                                   c->lod_leaf, c->lod_child, l, m, thr, static_cast<uint8_t>(match), c->lod_emit, c->lod_queue[l & 1u],
                                   c->lod_queue[(l + 1u) & 1u], c->lod_qcount);
        }
    } else {
        c->lod_stamp = 255u;            // the flags below overwrite the stamps
        hipLaunchKernelGGL(k_octree_stop, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->sel_stream, c->lod_a, c->lod_b, c->lod_leaf, n,
                           m, thr, c->lod_stop);
        hipLaunchKernelGGL(k_emit_flags, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->sel_stream, c->lod_stop, c->lod_parent, n,
                           c->lod_emit);
    }
    const int rc = compact(c, n, match);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SEL1], c->sel_stream));
    return rc;
}

int gsx_launch_select_hc(gsx_ctx* c, const float cam[3], float thr) {
    const uint32_t n = c->lod_nodes;
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SEL0], c->sel_stream));
    c->lod_stamp = 255u;
    hipLaunchKernelGGL(k_hc_stop, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->sel_stream, c->lod_a, c->pos4, c->lod_leaf, n,
                       cam[0], cam[1], cam[2], thr, c->lod_stop);
    hipLaunchKernelGGL(k_emit_flags, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->sel_stream, c->lod_stop, c->lod_parent, n,
                       c->lod_emit);
    const int rc = compact(c, n);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SEL1], c->sel_stream));
    return rc;
}

// Explicit list -> ascending, duplicate-free selection: mark then compact.
__global__ __launch_bounds__(256) void k_mark_list(const uint32_t* __restrict__ list, uint32_t m, uint32_t n,
                                                   uint8_t* __restrict__ emit, GsxDevFlags* __restrict__ flags) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint32_t i = list[k];
    if (i < n) emit[i] = 1;
    else atomicOr(&flags->bad_index, 1u);  // out-of-range id
}

// `list` is a device staging copy of the caller's ids (c->vals[0] is free between sorts).
int gsx_launch_sort_list(gsx_ctx* c, uint32_t m) {
    const uint32_t n = static_cast<uint32_t>(c->n);
    c->lod_stamp = 255u;
    GSX_HIP(c, hipMemsetAsync(c->lod_emit, 0, (static_cast<uint64_t>(n) + 15) / 16 * 16, c->sel_stream));
    if (m > 0)
        hipLaunchKernelGGL(k_mark_list, dim3(gsx_div_up(m, 256)), dim3(256), 0, c->sel_stream, c->vals[0], m, n, c->lod_emit,
                           c->flags);
    return compact(c, n);
}

// ---------------------------------------------------------------------------
// Cull-by-band selection (tile-band sharding): the splats whose quad can reach rows [y0, y1) of the frame.
// Conservative, from the 16-byte SoA record alone (position + radius, coalesced): the quad's half extent in view space is at
// most cutOff * splatSize * ||W||_F * radius per axis vector (W = the 3x3 of view * model), two axis vectors, projected with
// row 1 of the projection.  A splat the rasterizer would draw into the band is never dropped; what is kept and lies
// outside is dropped later by the exact projection (k_chunk_count).  Flags -> ascending ids like the LOD cuts, so the
// band's sorted permutation is the full permutation with the culled entries removed.
// ---------------------------------------------------------------------------
struct GsxBandView { float wt[16], proj[16]; float reach /* cutOff * splatSize * ||W||_2 */, height, y0, y1; };
__global__ __launch_bounds__(256) void k_band_flags(const float4* __restrict__ pos4, uint32_t n, GsxBandView v, uint8_t* __restrict__ emit) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos4[i];
    float c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = (v.wt[r] * p.x + v.wt[4 + r] * p.y) + (v.wt[8 + r] * p.z + v.wt[12 + r]);
    const float w = v.proj[11] * c[2] + v.proj[15] * c[3];
    const float ccy = (v.proj[1] * c[0] + v.proj[5] * c[1]) + (v.proj[9] * c[2] + v.proj[13] * c[3]);
    const float iw = 1.0f / w, hh = 0.5f * v.height;
    const float cy = (0.5f - ccy * iw * 0.5f) * v.height;
    // the quad's two axis vectors are orthogonal in the view plane with lengths <= reach * radius: their y parts sum to <= sqrt(2) x that
    const float ey = 1.41422f * v.reach * p.w * (fabsf(v.proj[1]) + fabsf(v.proj[5])) * iw * hh * 1.001f + 1.0f;
    // (written so that anything not finite, and anything at or behind the camera plane, is KEPT: the exact test decides)
    const bool cull = (w > 1.0e-6f) && (cy + ey < v.y0 || cy - ey >= v.y1);
    emit[i] = cull ? 0 : 1;
}

int gsx_launch_select_band(gsx_ctx* c, const float wt[16], const float proj[16], float cut_s, uint32_t height, uint32_t y0, uint32_t y1) {
    const uint32_t n = static_cast<uint32_t>(c->n);
    GsxBandView v;
    float fro = 0.0f;
    for (int k = 0; k < 16; ++k) { v.wt[k] = wt[k]; v.proj[k] = proj[k]; }
    for (int col = 0; col < 3; ++col)
        for (int r = 0; r < 3; ++r) fro += wt[col * 4 + r] * wt[col * 4 + r];
    // spectral norm of the 3x3 of view * model: power iteration on W^T W (converges from below: 1 % margin, never above Frobenius)
    double g[3][3], x[3] = {0.577, 0.577, 0.577}, lam = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            g[a][b] = 0.0;
            for (int r = 0; r < 3; ++r) g[a][b] += static_cast<double>(wt[a * 4 + r]) * wt[b * 4 + r];
        }
    for (int it = 0; it < 64; ++it) {
        double y[3] = {0, 0, 0}, nrm = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) y[a] += g[a][b] * x[b];
        for (int a = 0; a < 3; ++a) nrm += y[a] * y[a];
        nrm = std::sqrt(nrm);
        if (!(nrm > 0.0)) break;
        lam = nrm;
        for (int a = 0; a < 3; ++a) x[a] = y[a] / nrm;
    }
    double w2 = std::sqrt(lam) * 1.01;
    if (!(w2 > 0.0) || w2 > std::sqrt(static_cast<double>(fro))) w2 = std::sqrt(static_cast<double>(fro));
    v.reach = cut_s * static_cast<float>(w2) * 1.0001f;
    v.height = static_cast<float>(height);
    v.y0 = static_cast<float>(y0);
    v.y1 = static_cast<float>(y1);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SEL0], c->sel_stream));
    c->lod_stamp = 255u;            // the flags below overwrite the octree cut's stamps
    if (n) hipLaunchKernelGGL(k_band_flags, dim3(gsx_div_up(n, 256)), dim3(256), 0, c->sel_stream, c->pos4, n, v, c->lod_emit);
    const int rc = compact(c, n);
    if (c->profiling) GSX_HIP(c, hipEventRecord(c->ev[EV_SEL1], c->sel_stream));
    return rc;
}
