// mi_bvh_build.h — BVH construction on the device.
//
// The reference builds its hierarchy on the host (agglomerative clustering over a
// sorted list, src/scene/bvh.cu:66-172) and flattens it (bvh.cu:239-369); its report
// names BVH quality as "the significant factor".  This builder is not a restatement of
// that algorithm ("not to be replicated") — the closest hit does not depend on
// the tree (canonical tie rule + padded boxes), so any valid tree renders
// the same image bit for bit.  36-bit Morton codes of the primitive centroids (on the box of the centroids), one radix
// sort (hipCUB), primitive records gathered into that order, then either PLOC (default:
// rounds of nearest-neighbour search in a window, mutual pairs merge) or an LBVH (Karras
// 2012: every interior node found independently from the longest common prefix of
// neighbouring keys, boxes fitted bottom-up); both emit the 64-B node format of
// include/mi_tracer.h (both child boxes in the parent), which is then collapsed into the
// 4-wide nodes the traversal kernels walk.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#define MI_BUILD_BLOCK 256

struct BuildBox {
    float lo[3], hi[3];
};

// A word other waves of the launch update with atomics, read at device scope (not from this CU's L1).
__device__ __forceinline__ float build_peek(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t build_peek(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// geometry.h:13-29 — triangle box = min/max of the three vertices, sphere box = c +- r.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_prim_bounds(uint32_t n_tri, uint32_t n_sph, const float *tri_verts, const float4 *sph, BuildBox *pb, float *scene_box) {
    const uint32_t i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    const uint32_t n = n_tri + n_sph;
    BuildBox b;
    for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; }
    if (i < n_tri) {
        const float *v = tri_verts + 9 * (size_t)i;
        for (int k = 0; k < 3; k++) {
            b.lo[k] = fminf(fminf(v[k], v[3 + k]), v[6 + k]);
            b.hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), v[6 + k]);
        }
    } else if (i < n) {
        const float4 cr = sph[i - n_tri];
        const float c[3] = {cr.x, cr.y, cr.z};
        for (int k = 0; k < 3; k++) { b.lo[k] = c[k] - cr.w; b.hi[k] = c[k] + cr.w; }
    }
    if (i < n) pb[i] = b;
    // scene box [0..5] and box of the centroids [8..13] (the Morton grid spans the latter: a ground plane
    // of two triangles does not coarsen the grid the mesh is sorted on): wave reduction, then one float
    // atomic per wave and component (floats of either sign order correctly through the int/uint trick)
    for (int k = 0; k < 3; k++) {
        float lo = b.lo[k], hi = b.hi[k];
        float clo = i < n ? 0.5f * (b.lo[k] + b.hi[k]) : 3.0e38f, chi = i < n ? clo : -3.0e38f;
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
            clo = fminf(clo, __shfl_xor(clo, off));
            chi = fmaxf(chi, __shfl_xor(chi, off));
        }
        if ((threadIdx.x & 63) == 0) {
            // atomicMin/Max on floats via ordered-int encoding.  -0.0 is canonicalised to +0.0 first (x + 0.0f):
            // its bit pattern is INT_MIN, which as an int would replace a stored negative lower bound and, on
            // the upper side, could never win.
            // A look at the value first (past the L1, which may hold what the CU's first wave saw): after the first few
            // waves hardly any wave still moves a bound, and 12 atomics per wave on six words were 2.8 ms for 1.3 M primitives.
            lo += 0.0f; hi += 0.0f; clo += 0.0f; chi += 0.0f;
            if (lo < build_peek(&scene_box[k])) { if (lo >= 0) atomicMin((int *)&scene_box[k], __float_as_int(lo)); else atomicMax((unsigned int *)&scene_box[k], __float_as_uint(lo)); }
            if (hi > build_peek(&scene_box[3 + k])) { if (hi >= 0) atomicMax((int *)&scene_box[3 + k], __float_as_int(hi)); else atomicMin((unsigned int *)&scene_box[3 + k], __float_as_uint(hi)); }
            if (clo < build_peek(&scene_box[8 + k])) { if (clo >= 0) atomicMin((int *)&scene_box[8 + k], __float_as_int(clo)); else atomicMax((unsigned int *)&scene_box[8 + k], __float_as_uint(clo)); }
            if (chi > build_peek(&scene_box[11 + k])) { if (chi >= 0) atomicMax((int *)&scene_box[11 + k], __float_as_int(chi)); else atomicMin((unsigned int *)&scene_box[11 + k], __float_as_uint(chi)); }
        }
    }
}

__device__ __forceinline__ uint64_t expand_bits12(uint32_t v) { // 12 bits -> every third bit of 36
    uint64_t r = 0;
    for (int b = 0; b < 12; b++) r |= (uint64_t)((v >> b) & 1u) << (3 * b);
    return r;
}

// key = (36-bit Morton code of the centroid in the box of all centroids) << 26 | primitive slot (< 2^26):
// unique, so the common-prefix function is well defined for coincident centroids.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_morton(uint32_t n, const BuildBox *pb, const float *scene_box, uint64_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t code = 0;
    for (int k = 0; k < 3; k++) {
        const float lo = scene_box[8 + k], hi = scene_box[11 + k];
        const float c = 0.5f * (pb[i].lo[k] + pb[i].hi[k]);
        const float ext = hi - lo;
        float u = ext > 0 ? (c - lo) / ext : 0.0f;
        u = fminf(fmaxf(u * 4096.0f, 0.0f), 4095.0f);
        code |= expand_bits12((uint32_t)u) << (2 - k);
    }
    keys[i] = (code << 26) | i;
    vals[i] = i;
}

// Longest common prefix of sorted keys i and j (-1 outside the array).
__device__ __forceinline__ int build_delta(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras 2012, one thread per interior node i in [0, n-1): range, split, children.
// child ref >= 0: interior node, < 0: ~(position in sorted order) for now.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_topology(int n, const uint64_t *keys, int2 *children, int *parent_node, int *parent_leaf) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (build_delta(keys, n, i, i + 1) - build_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = build_delta(keys, n, i, i - d);
    int lmax = 2;
    while (build_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (build_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = build_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (build_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break; // (l >= 1 for unique keys; t == 0 cannot loop for ever either)
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    int2 c;
    if (lo == gamma) { c.x = ~gamma; parent_leaf[gamma] = i; } else { c.x = gamma; parent_node[gamma] = i; }
    if (hi == gamma + 1) { c.y = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { c.y = gamma + 1; parent_node[gamma + 1] = i; }
    children[i] = c;
    if (i == 0) parent_node[0] = -1;
}

// Bottom-up fit: one thread per leaf climbs; the second thread to arrive at a node owns
// it (both children are complete), the first one leaves.  Every thread terminates: it
// either leaves at a flag or reaches the root.
__device__ __forceinline__ BuildBox build_load_box(const BuildBox *p) { // written by another thread of this launch
    const volatile float *f = (const volatile float *)p;
    BuildBox b;
    for (int k = 0; k < 3; k++) { b.lo[k] = f[k]; b.hi[k] = f[3 + k]; }
    return b;
}
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_fit(int n, const uint32_t *sorted_slot, const BuildBox *pb, const int2 *children, const int *parent_node,
            const int *parent_leaf, BuildBox *nb, int *height, unsigned int *flags) {
    const int leaf = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (leaf >= n) return;
    int cur = parent_leaf[leaf];
    while (cur >= 0) {
        __threadfence(); // my child's box and height are visible before the flag is raised
        if (atomicAdd(&flags[cur], 1u) == 0u) return;
        __threadfence(); // the sibling's box and height are visible after the flag is seen
        const int2 c = children[cur];
        BuildBox bl, br;
        int hl = 0, hr = 0;
        if (c.x < 0) bl = pb[sorted_slot[~c.x]]; else { bl = build_load_box(&nb[c.x]); hl = *(volatile int *)&height[c.x]; }
        if (c.y < 0) br = pb[sorted_slot[~c.y]]; else { br = build_load_box(&nb[c.y]); hr = *(volatile int *)&height[c.y]; }
        BuildBox b;
        for (int k = 0; k < 3; k++) { b.lo[k] = fminf(bl.lo[k], br.lo[k]); b.hi[k] = fmaxf(bl.hi[k], br.hi[k]); }
        nb[cur] = b;
        height[cur] = 1 + (hl > hr ? hl : hr);
        cur = parent_node[cur];
    }
}

__device__ __forceinline__ float build_area(const BuildBox &b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// Emit the 64-B nodes of include/mi_tracer.h: padded child boxes, child refs with leaves
// as ~slot.  Also sums the surface areas of all child boxes (SAH cost numerator).
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_emit(int n, const uint32_t *sorted_slot, const BuildBox *pb, const int2 *children, const BuildBox *nb, float pad,
             float4 *nodes, double *area_sum) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    float area = 0.0f;
    if (i < n - 1) {
        const int2 c = children[i];
        const BuildBox bl = c.x < 0 ? pb[sorted_slot[~c.x]] : nb[c.x];
        const BuildBox br = c.y < 0 ? pb[sorted_slot[~c.y]] : nb[c.y];
        const int rl = c.x < 0 ? ~(int)sorted_slot[~c.x] : c.x;
        const int rr = c.y < 0 ? ~(int)sorted_slot[~c.y] : c.y;
        nodes[4 * (size_t)i + 0] = make_float4(bl.lo[0] - pad, bl.lo[1] - pad, bl.lo[2] - pad, bl.hi[0] + pad);
        nodes[4 * (size_t)i + 1] = make_float4(bl.hi[1] + pad, bl.hi[2] + pad, br.lo[0] - pad, br.lo[1] - pad);
        nodes[4 * (size_t)i + 2] = make_float4(br.lo[2] - pad, br.hi[0] + pad, br.hi[1] + pad, br.hi[2] + pad);
        nodes[4 * (size_t)i + 3] = make_float4(__int_as_float(rl), __int_as_float(rr), 0.0f, 0.0f);
        area = build_area(bl) + build_area(br);
    }
    // block reduction -> one double atomic per workgroup
    __shared__ float part[MI_BUILD_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) area += __shfl_xor(area, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = area;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < MI_BUILD_BLOCK / 64; w++) s += part[w];
        if (s != 0) atomicAdd(area_sum, s);
    }
}

// Surface-area sum of the child boxes of resident nodes (any tree, e.g. the host SAH one).
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_nodes_area(uint32_t n_nodes, const float4 *nodes, double *area_sum) {
    const uint32_t i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    float area = 0.0f;
    if (i < n_nodes) {
        const float4 a = nodes[4 * (size_t)i], b = nodes[4 * (size_t)i + 1], c = nodes[4 * (size_t)i + 2];
        const BuildBox bl{{a.x, a.y, a.z}, {a.w, b.x, b.y}}, br{{b.z, b.w, c.x}, {c.y, c.z, c.w}};
        area = build_area(bl) + build_area(br);
    }
    __shared__ float part[MI_BUILD_BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) area += __shfl_xor(area, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = area;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int w = 0; w < MI_BUILD_BLOCK / 64; w++) s += part[w];
        if (s != 0) atomicAdd(area_sum, s);
    }
}

// ---- collapse of the binary tree into the 4-wide tree the traversal kernels walk --------
// A 4-wide node takes the place of every interior node at EVEN depth; its seats are that
// node's grandchildren (the children of each interior child) or the child itself when it is
// a leaf — the same rule as collapse_bvh4() on the host (mi_tracer.hip).
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_even_depth(int n_nodes, const int *parent_node, uint32_t *flag, int *max_depth) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n_nodes) return;
    int d = 0;
    for (int p = parent_node[i]; p >= 0; p = parent_node[p]) d++; // ends at the root (parent -1)
    flag[i] = (d & 1) ? 0u : 1u;
    if (max_depth) atomicMax(max_depth, d + 1); // levels of interior nodes above and including this one
}

__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_emit4(int n_nodes, const float4 *nodes2, const uint32_t *flag, const uint32_t *index4, float *nodes4, int empty_ref) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n_nodes || !flag[i]) return;
    float box[4][6];
    int ref[4];
    int n = 0;
    const float *nd = (const float *)(nodes2 + 4 * (size_t)i);
    for (int c = 0; c < 2; c++) {
        const int r = __float_as_int(nd[12 + c]);
        if (r < 0) {
            for (int k = 0; k < 6; k++) box[n][k] = nd[6 * c + k];
            ref[n++] = r;
        } else {
            const float *ch = (const float *)(nodes2 + 4 * (size_t)r);
            for (int g = 0; g < 2; g++) {
                for (int k = 0; k < 6; k++) box[n][k] = ch[6 * g + k];
                const int rr = __float_as_int(ch[12 + g]);
                ref[n++] = rr < 0 ? rr : (int)index4[rr];
            }
        }
    }
    float *o = nodes4 + 32 * (size_t)index4[i];
    for (int k = 0; k < 4; k++) {
        for (int a = 0; a < 3; a++) {
            o[4 * a + k] = k < n ? box[k][a] : 0.0f;
            o[12 + 4 * a + k] = k < n ? box[k][3 + a] : 0.0f;
        }
        o[24 + k] = __int_as_float(k < n ? ref[k] : empty_ref);
        o[28 + k] = 0.0f;
    }
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) -----------------
// Bottom-up agglomeration over the Morton-sorted primitives: in every round each cluster
// looks at its MI_PLOC_RADIUS neighbours on either side for the one whose union box has
// the smallest surface area; mutual nearest neighbours merge into a new interior node; the
// survivors are compacted in order.  A few dozen rounds; tree quality close to a top-down
// SAH build at a build time close to the LBVH's.
#define MI_PLOC_RADIUS 4 // measured with the 36-bit Morton order: 2/4/8/16/32/64 -> C3 7732/7706/7673/7627/7484/7427, C4 6671/6632/6517/6523/6407/6355 Mrays/s (2 loses 3 % on C5)

__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_ploc_init(int n, const uint32_t *sorted_slot, const BuildBox *pb, int *cl_ref, BuildBox *cl_box) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n) return;
    cl_ref[i] = ~(int)sorted_slot[i];
    cl_box[i] = pb[sorted_slot[i]];
}

__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_ploc_nearest(int m, const BuildBox *cl_box, int *nn) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= m) return;
    const BuildBox a = cl_box[i];
    // Pairs are ranked by (union area, parity of the lower index, distance, lower index): a
    // strict total order on unordered pairs, so the best pair overall is always mutual (a round
    // always merges something), and among equal areas — coincident primitives — neighbours pair
    // up (0,1), (2,3), ... instead of forming one chain.
    float best_area = 3.0e38f;
    unsigned best_key = 0xffffffffu;
    int best = -1;
    const int j0 = i - MI_PLOC_RADIUS < 0 ? 0 : i - MI_PLOC_RADIUS, j1 = i + MI_PLOC_RADIUS >= m ? m - 1 : i + MI_PLOC_RADIUS;
    for (int j = j0; j <= j1; j++) {
        if (j == i) continue;
        const BuildBox b = cl_box[j];
        BuildBox u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
        const float area = build_area(u);
        const int lo = i < j ? i : j, dist = i < j ? j - i : i - j;
        const unsigned key = ((unsigned)(lo & 1) << 31) | ((unsigned)dist << 26) | (unsigned)lo; // dist <= 16 < 32, lo < 2^26
        if (area < best_area || (area == best_area && key < best_key)) { best_area = area; best_key = key; best = j; }
    }
    nn[i] = best;
}

// Mutual nearest neighbours (i < j) become interior node k: children in cluster order, boxes
// padded, 64-B node format.  keep[i] says whether position i survives into the next round.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_ploc_merge(int m, const int *nn, const int *cl_ref, const BuildBox *cl_box, int *out_ref, BuildBox *out_box, uint32_t *keep,
             uint32_t *node_counter, float4 *nodes, int *parent_node, float pad) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    if (mutual && i > j) { keep[i] = 0u; return; } // merged into position j
    BuildBox a = cl_box[i];
    int ref = cl_ref[i];
    if (mutual) {
        const BuildBox b = cl_box[j];
        const int rb = cl_ref[j];
        const int k = (int)atomicAdd(node_counter, 1u);
        nodes[4 * (size_t)k + 0] = make_float4(a.lo[0] - pad, a.lo[1] - pad, a.lo[2] - pad, a.hi[0] + pad);
        nodes[4 * (size_t)k + 1] = make_float4(a.hi[1] + pad, a.hi[2] + pad, b.lo[0] - pad, b.lo[1] - pad);
        nodes[4 * (size_t)k + 2] = make_float4(b.lo[2] - pad, b.hi[0] + pad, b.hi[1] + pad, b.hi[2] + pad);
        nodes[4 * (size_t)k + 3] = make_float4(__int_as_float(ref), __int_as_float(rb), 0.0f, 0.0f);
        if (ref >= 0) parent_node[ref] = k;
        if (rb >= 0) parent_node[rb] = k;
        parent_node[k] = -1; // until something adopts it
        for (int c = 0; c < 3; c++) { a.lo[c] = fminf(a.lo[c], b.lo[c]); a.hi[c] = fmaxf(a.hi[c], b.hi[c]); }
        ref = k;
    }
    out_ref[i] = ref;
    out_box[i] = a;
    keep[i] = 1u;
}

__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_ploc_compact(int m, const uint32_t *keep, const uint32_t *pos, const int *in_ref, const BuildBox *in_box, int *cl_ref, BuildBox *cl_box,
               uint32_t *new_count) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= m) return;
    if (keep[i]) {
        cl_ref[pos[i]] = in_ref[i];
        cl_box[pos[i]] = in_box[i];
    }
    if (i == m - 1) *new_count = pos[i] + keep[i];
}

// ---- top-down binned SAH on the device ---------------------------------------------------
// The algorithm of the host builder (host/scene.cpp Builder::build: 16 bins on the box of the centroids, all three
// axes, cost = half area x count of either side, first minimum in (axis, bin) order, primitives with bin <= the split
// on the left, single primitives as leaves), carried out level by level instead of recursively, with the same
// arithmetic (float bin index, double cost), so that it arrives at the tree the host builds.  Nodes of more than
// MI_TD_SMALL primitives are split one tree level per round of launches (the primitives — box and original slot — are
// moved so that every node owns a contiguous range of positions; waves aggregate in LDS, one or two nodes per wave);
// a node of at most MI_TD_SMALL primitives is finished by ONE wave that holds a primitive per lane and splits lane
// masks (k_td_small).  Where the host splits at the object median (no axis with distinct centroids, or 36 levels
// down) the range is halved in position order: any such split is valid, and depth stays below 36 + log2(n).
#define MI_TD_BINS 16
#define MI_TD_SMALL 64
#define MI_TD_MEDIAN_DEPTH 36
#define MI_TD_NONE 0xffffffffu
#define MI_TD_BIN_WORDS (3 * MI_TD_BINS * 7) // per node: [axis][bin]{count, ~ord(lo)[3], ord(hi)[3]}, identity 0 (adds and maxima only)

struct TdNode {         // a node of the top phase (64 B)
    uint32_t begin, count;
    int32_t parent;     // binary node above it (-1: the root), and which of its children this is
    uint32_t depth;
    uint32_t cb[6];     // box of the centroids: ~ord(lo)[3], ord(hi)[3] (maxima; 0 = nothing yet)
    int32_t axis, bin;  // the split: axis (-1: halved in position order), last bin of the left side
    uint32_t nleft;
    uint32_t child[2];  // TdNode index of a child that stays in the top phase, else MI_TD_NONE
    uint32_t which;
};
struct TdSmall { uint32_t begin, count; int32_t parent; uint32_t depth_which; }; // depth | which << 31

// The binary node that splits positions [begin, begin + count) after its first nleft is node begin + nleft - 1: every
// gap between neighbouring positions is split exactly once, so the n - 1 indices are handed out without a counter
// (one shared counter for 40 000 nodes of a level cost 0.4 ms of returning atomics).  A node learns its index when it is
// split; it then enters itself in its parent's record.
__device__ __forceinline__ void td_link(float4 *out_nodes, int *parent_node, uint32_t *counters, int32_t self, int32_t parent, uint32_t which) {
    parent_node[self] = parent;
    if (parent >= 0) ((int32_t *)out_nodes)[16 * (size_t)parent + 12 + which] = self;
    else counters[3] = (uint32_t)self; // the root
}

// float -> uint32, order preserving (and back)
__device__ __forceinline__ uint32_t td_ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float td_unord(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ int td_bin(float cen, float lo, float scale) {
    const int b = (int)((cen - lo) * scale); // scene.cpp:199
    return b > MI_TD_BINS - 1 ? MI_TD_BINS - 1 : (b < 0 ? 0 : b);
}
__device__ __forceinline__ double td_half_area(const float *lo, const float *hi) { // scene.cpp:153-156
    const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
    return x * y + y * z + z * x;
}
// LDS written by some lanes of a wave and read by others: the hardware keeps a wave's LDS operations in order, this keeps the compiler from moving them
__device__ __forceinline__ void td_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ float td_wave_min(float v) { for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off)); return v; }
__device__ __forceinline__ float td_wave_max(float v) { for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off)); return v; }

// Lane (axis = lane / 16, b = lane % 16) prices the split "bins 0..b of that axis | bins b+1..15" from the node's
// bins (global or LDS) and the wave agrees on the cheapest (scene.cpp:203-221: first minimum in (axis, bin) order).
// Returns the winning lane (>= 48: no valid split); that lane's L/R hold the children's boxes, cl its left count.
__device__ __forceinline__ int td_choose(const uint32_t *bins, int lane, const float *cb_lo, const float *cb_hi, bool allowed, float *L_lo, float *L_hi,
                                         float *R_lo, float *R_hi, uint32_t &cl) {
    // (unions kept as floats: with the maxima taken on the bins' own encoding and decoded once, hipcc 7.2 crashes in
    // instruction selection for k_td_split)
    const int axis = lane >> 4, b = lane & 15;
    double cost = __longlong_as_double(0x7ff0000000000000ll);
    uint32_t cr = 0;
    cl = 0;
    for (int k = 0; k < 3; k++) { L_lo[k] = R_lo[k] = 3.0e38f; L_hi[k] = R_hi[k] = -3.0e38f; }
    const float ext = axis == 0 ? cb_hi[0] - cb_lo[0] : axis == 1 ? cb_hi[1] - cb_lo[1] : cb_hi[2] - cb_lo[2];
    if (allowed && lane < 48 && b < MI_TD_BINS - 1 && ext > 0.0f) {
        const uint32_t *w = bins + (size_t)axis * MI_TD_BINS * 7;
        for (int j = 0; j < MI_TD_BINS; j++, w += 7) {
            const uint32_t c = w[0];
            if (!c) continue;
            if (j <= b) {
                cl += c;
                for (int k = 0; k < 3; k++) { L_lo[k] = fminf(L_lo[k], td_unord(~w[1 + k])); L_hi[k] = fmaxf(L_hi[k], td_unord(w[4 + k])); }
            } else {
                cr += c;
                for (int k = 0; k < 3; k++) { R_lo[k] = fminf(R_lo[k], td_unord(~w[1 + k])); R_hi[k] = fmaxf(R_hi[k], td_unord(w[4 + k])); }
            }
        }
        if (cl && cr) cost = td_half_area(L_lo, L_hi) * (double)cl + td_half_area(R_lo, R_hi) * (double)cr;
    }
    int best = cost < __longlong_as_double(0x7ff0000000000000ll) ? lane : 64;
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(cost, off);
        const int ob = __shfl_xor(best, off);
        if (oc < cost || (oc == cost && ob < best)) { cost = oc; best = ob; }
    }
    return best;
}

__device__ __forceinline__ void td_write_node(float4 *nodes, int32_t out, const float *a_lo, const float *a_hi, const float *b_lo, const float *b_hi, int32_t ra,
                                              int32_t rb, float pad) {
    nodes[4 * (size_t)out + 0] = make_float4(a_lo[0] - pad, a_lo[1] - pad, a_lo[2] - pad, a_hi[0] + pad);
    nodes[4 * (size_t)out + 1] = make_float4(a_hi[1] + pad, a_hi[2] + pad, b_lo[0] - pad, b_lo[1] - pad);
    nodes[4 * (size_t)out + 2] = make_float4(b_lo[2] - pad, b_hi[0] + pad, b_hi[1] + pad, b_hi[2] + pad);
    nodes[4 * (size_t)out + 3] = make_float4(__int_as_float(ra), __int_as_float(rb), 0.0f, 0.0f);
}

__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_init(uint32_t n, const BuildBox *pb, BuildBox *box, uint32_t *idx, uint32_t *node_of, uint32_t first) {
    const uint32_t i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i < n) { idx[i] = i; node_of[i] = first; box[i] = pb[i]; }
}

// The one or two top-phase nodes a wave's 64 positions belong to (nodes there own > 64 consecutive positions).
struct TdWave { uint32_t a, b; bool any; };
__device__ __forceinline__ TdWave td_wave_nodes(uint32_t nid) {
    const unsigned long long am = __ballot(nid != MI_TD_NONE);
    TdWave w{MI_TD_NONE, MI_TD_NONE, am != 0ull};
    if (w.any) { w.a = __shfl(nid, __ffsll((long long)am) - 1); w.b = __shfl(nid, 63 - __clzll((long long)am)); }
    return w;
}

// Box of the centroids of every node of the level (scene.cpp:181-185).  A wave walks `chunks` consecutive blocks of 64
// positions and keeps its partial boxes in registers for as long as the blocks belong to the same one or two nodes
// (near the root: always), so a large node gets a handful of atomics per wave, not per 64 primitives.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_bounds(uint32_t n, const BuildBox *box, const uint32_t *node_of, TdNode *nodes, uint32_t chunks) {
    const int lane = threadIdx.x & 63;
    const uint32_t pos0 = ((blockIdx.x * MI_BUILD_BLOCK + threadIdx.x) >> 6) * 64u * chunks;
    uint32_t cur_a = MI_TD_NONE, cur_b = MI_TD_NONE;
    float a_lo[3], a_hi[3], b_lo[3], b_hi[3];
    auto reset = [&]() { for (int k = 0; k < 3; k++) { a_lo[k] = b_lo[k] = 3.0e38f; a_hi[k] = b_hi[k] = -3.0e38f; } };
    auto flush = [&]() {
        for (int pass = 0; pass < 2; pass++) {
            const uint32_t target = pass ? cur_b : cur_a;
            if (target == MI_TD_NONE || (pass && cur_b == cur_a)) continue;
            for (int k = 0; k < 3; k++) {
                const float lo = td_wave_min(pass ? b_lo[k] : a_lo[k]), hi = td_wave_max(pass ? b_hi[k] : a_hi[k]);
                if (lane == 0 && lo <= hi) { // look first: most waves of a large node move nothing
                    if (~td_ord(lo) > build_peek(&nodes[target].cb[k])) atomicMax(&nodes[target].cb[k], ~td_ord(lo));
                    if (td_ord(hi) > build_peek(&nodes[target].cb[3 + k])) atomicMax(&nodes[target].cb[3 + k], td_ord(hi));
                }
            }
        }
    };
    reset();
    uint32_t nid_next = pos0 + lane < n ? node_of[pos0 + lane] : MI_TD_NONE;
    BuildBox b_next = box[pos0 + lane < n ? pos0 + lane : 0u];
    for (uint32_t ch = 0; ch < chunks; ch++) {
        const uint32_t nid = nid_next;
        const BuildBox b = b_next;
        if (ch + 1 < chunks) { // the next block's loads are in flight while this one is reduced
            const uint32_t pn = pos0 + (ch + 1) * 64u + lane;
            nid_next = pn < n ? node_of[pn] : MI_TD_NONE;
            b_next = box[pn < n ? pn : 0u];
        }
        const TdWave w = td_wave_nodes(nid);
        if (!w.any) continue;
        if (w.a != cur_a || w.b != cur_b) { flush(); reset(); cur_a = w.a; cur_b = w.b; }
        if (nid == MI_TD_NONE) continue;
        for (int k = 0; k < 3; k++) {
            const float c = 0.5f * (b.lo[k] + b.hi[k]);
            if (nid == cur_a) { a_lo[k] = fminf(a_lo[k], c); a_hi[k] = fmaxf(a_hi[k], c); }
            else if (nid == cur_b) { b_lo[k] = fminf(b_lo[k], c); b_hi[k] = fmaxf(b_hi[k], c); }
            else { atomicMax(&nodes[nid].cb[k], ~td_ord(c)); atomicMax(&nodes[nid].cb[3 + k], td_ord(c)); } // cannot happen while top-phase nodes own > 64 positions
        }
    }
    flush();
}

// Bins of every node of the level (scene.cpp:191-202), aggregated per wave in LDS over `chunks` blocks of 64 positions
// like the boxes above.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_bin(uint32_t n, const BuildBox *box, const uint32_t *node_of, const TdNode *nodes, uint32_t first, uint32_t *gbins, uint32_t chunks) {
    __shared__ uint32_t lds[MI_BUILD_BLOCK / 64][2 * MI_TD_BIN_WORDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t pos0 = ((blockIdx.x * MI_BUILD_BLOCK + threadIdx.x) >> 6) * 64u * chunks;
    uint32_t cur_a = MI_TD_NONE, cur_b = MI_TD_NONE;
    auto flush = [&]() { // and clear
        td_wave_sync();
        for (int i = lane; i < 2 * MI_TD_BIN_WORDS; i += 64) {
            const uint32_t v = lds[wv][i];
            lds[wv][i] = 0u;
            if (!v) continue;
            const uint32_t target = i < MI_TD_BIN_WORDS ? cur_a : cur_b;
            const int word = i < MI_TD_BIN_WORDS ? i : i - MI_TD_BIN_WORDS;
            uint32_t *q = gbins + (size_t)(target - first) * MI_TD_BIN_WORDS + word;
            if (word % 7 == 0) atomicAdd(q, v); else if (v > build_peek(q)) atomicMax(q, v);
        }
        td_wave_sync();
    };
    for (int i = lane; i < 2 * MI_TD_BIN_WORDS; i += 64) lds[wv][i] = 0u;
    td_wave_sync();
    uint32_t nid_next = pos0 + lane < n ? node_of[pos0 + lane] : MI_TD_NONE;
    BuildBox b_next = box[pos0 + lane < n ? pos0 + lane : 0u];
    for (uint32_t ch = 0; ch < chunks; ch++) {
        const uint32_t nid = nid_next;
        const BuildBox b = b_next;
        if (ch + 1 < chunks) {
            const uint32_t pn = pos0 + (ch + 1) * 64u + lane;
            nid_next = pn < n ? node_of[pn] : MI_TD_NONE;
            b_next = box[pn < n ? pn : 0u];
        }
        const TdWave w = td_wave_nodes(nid);
        if (!w.any) continue;
        if (w.a != cur_a || w.b != cur_b) { if (cur_a != MI_TD_NONE) flush(); cur_a = w.a; cur_b = w.b; }
        if (nid == MI_TD_NONE) continue;
        const TdNode *nd = nodes + nid;
        const bool local = nid == cur_a || nid == cur_b;
        uint32_t *dst = local ? &lds[wv][nid == cur_a ? 0 : MI_TD_BIN_WORDS] : gbins + (size_t)(nid - first) * MI_TD_BIN_WORDS;
        for (int a = 0; a < 3; a++) {
            const float lo = td_unord(~nd->cb[a]), hi = td_unord(nd->cb[3 + a]);
            const float ext = hi - lo;
            if (!(ext > 0.0f)) continue;
            const float scale = (float)MI_TD_BINS / ext;
            uint32_t *q = dst + (a * MI_TD_BINS + td_bin(0.5f * (b.lo[a] + b.hi[a]), lo, scale)) * 7;
            atomicAdd(q, 1u);
            for (int k = 0; k < 3; k++) { atomicMax(q + 1 + k, ~td_ord(b.lo[k])); atomicMax(q + 4 + k, td_ord(b.hi[k])); }
        }
    }
    if (cur_a != MI_TD_NONE) flush();
}

// One wave per node of the level: choose the split, create the children, write the binary node.
// counters: [1] nodes of the next level, [2] small jobs (one atomic each per workgroup of 16 nodes), [3] the root's index.
#define MI_TD_SPLIT_BLOCK 1024
__global__ void __launch_bounds__(MI_TD_SPLIT_BLOCK)
k_td_split(TdNode *nodes, uint32_t first, uint32_t n_level, uint32_t next_first, const uint32_t *gbins, const BuildBox *box, float4 *out_nodes,
           int *parent_node, uint32_t *counters, TdSmall *small, float pad) {
    __shared__ uint32_t sb[MI_TD_SPLIT_BLOCK / 64][MI_TD_BIN_WORDS]; // the node's bins, fetched together
    __shared__ uint32_t want[MI_TD_SPLIT_BLOCK / 64][2];              // children of the wave's node that go on in the top phase / as small jobs
    __shared__ uint32_t base[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t wi = blockIdx.x * (MI_TD_SPLIT_BLOCK / 64) + wv;
    const bool valid = wi < n_level;
    TdNode *nd = nodes + first + (valid ? wi : 0u);
    const uint32_t begin = nd->begin, count = nd->count, depth = nd->depth;
    float L_lo[3], L_hi[3], R_lo[3], R_hi[3];
    int axis = -1, bin = 0;
    uint32_t nleft = 0;
    if (valid) {
        float cb_lo[3], cb_hi[3];
        for (int k = 0; k < 3; k++) { cb_lo[k] = td_unord(~nd->cb[k]); cb_hi[k] = td_unord(nd->cb[3 + k]); }
        uint32_t cl;
        for (int i = lane; i < MI_TD_BIN_WORDS; i += 64) sb[wv][i] = gbins[(size_t)wi * MI_TD_BIN_WORDS + i];
        td_wave_sync();
        const int best = td_choose(sb[wv], lane, cb_lo, cb_hi, depth < MI_TD_MEDIAN_DEPTH, L_lo, L_hi, R_lo, R_hi, cl);
        if (best < 48) {
            axis = best >> 4; bin = best & 15;
            nleft = __shfl(cl, best);
            for (int k = 0; k < 3; k++) { L_lo[k] = __shfl(L_lo[k], best); L_hi[k] = __shfl(L_hi[k], best); R_lo[k] = __shfl(R_lo[k], best); R_hi[k] = __shfl(R_hi[k], best); }
        } else { // scene.cpp:235-243, in position order
            nleft = count / 2;
            for (int k = 0; k < 3; k++) { L_lo[k] = R_lo[k] = 3.0e38f; L_hi[k] = R_hi[k] = -3.0e38f; }
            for (uint32_t i = lane; i < count; i += 64) {
                const BuildBox b = box[begin + i];
                if (i < nleft) for (int k = 0; k < 3; k++) { L_lo[k] = fminf(L_lo[k], b.lo[k]); L_hi[k] = fmaxf(L_hi[k], b.hi[k]); }
                else for (int k = 0; k < 3; k++) { R_lo[k] = fminf(R_lo[k], b.lo[k]); R_hi[k] = fmaxf(R_hi[k], b.hi[k]); }
            }
            for (int k = 0; k < 3; k++) { L_lo[k] = td_wave_min(L_lo[k]); L_hi[k] = td_wave_max(L_hi[k]); R_lo[k] = td_wave_min(R_lo[k]); R_hi[k] = td_wave_max(R_hi[k]); }
        }
    }
    const uint32_t cnt[2] = {nleft, count - nleft};
    if (lane == 0) {
        want[wv][0] = want[wv][1] = 0u;
        if (valid) for (int c = 0; c < 2; c++) { if (cnt[c] > MI_TD_SMALL) want[wv][0]++; else if (cnt[c] > 1) want[wv][1]++; }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t tot = 0;
        for (int w = 0; w < MI_TD_SPLIT_BLOCK / 64; w++) tot += want[w][threadIdx.x];
        base[threadIdx.x] = tot ? atomicAdd(&counters[1 + threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    if (lane != 0 || !valid) return;
    uint32_t at[2] = {base[0], base[1]};
    for (int w = 0; w < wv; w++) { at[0] += want[w][0]; at[1] += want[w][1]; }
    const int32_t self = (int32_t)(begin + nleft - 1);
    int32_t ref[2];
    uint32_t child[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t cb = c ? begin + nleft : begin;
        child[c] = MI_TD_NONE;
        ref[c] = cnt[c] == 1 ? ~(int32_t)cb : 0; // an interior child enters itself when it is split (td_link)
        if (cnt[c] > MI_TD_SMALL) {
            const uint32_t t = next_first + at[0]++;
            TdNode nn{};
            nn.begin = cb; nn.count = cnt[c]; nn.parent = self; nn.which = (uint32_t)c; nn.depth = depth + 1;
            nodes[t] = nn;
            child[c] = t;
        } else if (cnt[c] > 1) {
            small[at[1]++] = TdSmall{cb, cnt[c], self, (depth + 1) | (uint32_t)c << 31};
        }
    }
    td_write_node(out_nodes, self, L_lo, L_hi, R_lo, R_hi, ref[0], ref[1], pad);
    td_link(out_nodes, parent_node, counters, self, nd->parent, nd->which);
    nd->axis = axis; nd->bin = bin; nd->nleft = nleft; nd->child[0] = child[0]; nd->child[1] = child[1];
}

// Which side every primitive of a split node goes to (scene.cpp:227-230); 0 for everything else.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_flag(uint32_t n, const BuildBox *box, const uint32_t *node_of, const TdNode *nodes, uint32_t *flag) {
    const uint32_t pos = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (pos >= n) return;
    const uint32_t nid = node_of[pos];
    uint32_t f = 0u;
    if (nid != MI_TD_NONE) {
        const TdNode *nd = nodes + nid;
        const int a = nd->axis;
        if (a < 0) f = pos - nd->begin < nd->nleft ? 1u : 0u;
        else {
            const float lo = td_unord(~nd->cb[a]), hi = td_unord(nd->cb[3 + a]);
            const float scale = (float)MI_TD_BINS / (hi - lo);
            f = td_bin(0.5f * (box[pos].lo[a] + box[pos].hi[a]), lo, scale) <= nd->bin ? 1u : 0u;
        }
    }
    flag[pos] = f;
}

// Stable partition of every split node's range (left side first) from the exclusive prefix sum of the flags.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_scatter(uint32_t n, const BuildBox *box, const uint32_t *idx, const uint32_t *node_of, const TdNode *nodes, const uint32_t *flag, const uint32_t *scan,
             BuildBox *box_out, uint32_t *idx_out, uint32_t *node_out) {
    const uint32_t pos = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (pos >= n) return;
    const uint32_t nid = node_of[pos];
    uint32_t to = pos, nn = MI_TD_NONE;
    if (nid != MI_TD_NONE) {
        const TdNode *nd = nodes + nid;
        const uint32_t r = scan[pos] - scan[nd->begin]; // flagged primitives of the node before this one
        if (flag[pos]) { to = nd->begin + r; nn = nd->child[0]; }
        else { to = nd->begin + nd->nleft + (pos - nd->begin - r); nn = nd->child[1]; }
    }
    box_out[to] = box[pos];
    idx_out[to] = idx[pos];
    node_out[to] = nn;
}

// A node of at most 64 primitives, finished by one wave: a primitive per lane, nodes are lane masks, the pending ones
// on a stack in LDS.  The subtree's primitives end up in depth-first order within the node's range (idx_out).
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_small(uint32_t n_jobs, const TdSmall *jobs, const BuildBox *box, const uint32_t *idx, float4 *out_nodes, int *parent_node, uint32_t *counters,
           uint32_t *idx_out, float pad) {
    __shared__ uint32_t bins[MI_BUILD_BLOCK / 64][MI_TD_BIN_WORDS];
    __shared__ unsigned long long st_mask[MI_BUILD_BLOCK / 64][MI_TD_SMALL];
    __shared__ uint32_t st_info[MI_BUILD_BLOCK / 64][MI_TD_SMALL][3]; // parent, depth | which << 31, start
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ji = blockIdx.x * (MI_BUILD_BLOCK / 64) + wv;
    if (ji >= n_jobs) return; // no block-wide barrier below
    const TdSmall job = jobs[ji];
    BuildBox mb;
    for (int k = 0; k < 3; k++) { mb.lo[k] = 3.0e38f; mb.hi[k] = -3.0e38f; }
    uint32_t my_idx = 0, my_pos = 0;
    float c[3] = {0, 0, 0};
    if ((uint32_t)lane < job.count) {
        mb = box[job.begin + lane];
        my_idx = idx[job.begin + lane];
        for (int k = 0; k < 3; k++) c[k] = 0.5f * (mb.lo[k] + mb.hi[k]);
    }
    if (lane == 0) {
        st_mask[wv][0] = job.count >= 64 ? ~0ull : (1ull << job.count) - 1ull;
        st_info[wv][0][0] = (uint32_t)job.parent; st_info[wv][0][1] = job.depth_which; st_info[wv][0][2] = 0u;
    }
    int sp = 1;
    while (sp > 0) {
        sp--;
        td_wave_sync();
        const unsigned long long mask = st_mask[wv][sp];
        const int32_t parent = (int32_t)st_info[wv][sp][0];
        const uint32_t depth = st_info[wv][sp][1] & 0x7fffffffu, which = st_info[wv][sp][1] >> 31, start = st_info[wv][sp][2];
        td_wave_sync();
        const bool in = (mask >> lane) & 1ull;
        const uint32_t m = (uint32_t)__popcll(mask);
        float cb_lo[3], cb_hi[3];
        for (int k = 0; k < 3; k++) { cb_lo[k] = td_wave_min(in ? c[k] : 3.0e38f); cb_hi[k] = td_wave_max(in ? c[k] : -3.0e38f); }
        float L_lo[3], L_hi[3], R_lo[3], R_hi[3];
        bool left, by_boxes = true; // by_boxes: the children's boxes are those of the lanes' own primitives
        const int axis2 = cb_hi[0] - cb_lo[0] > 0.0f ? 0 : cb_hi[1] - cb_lo[1] > 0.0f ? 1 : cb_hi[2] - cb_lo[2] > 0.0f ? 2 : -1;
        if (m == 2 && depth < MI_TD_MEDIAN_DEPTH && axis2 >= 0) {
            // two primitives: every split that separates them costs the same, so the first axis on which the centroids
            // differ decides, and the one at the lower end (bin 0; the other lands in bin 15) goes left
            left = in && (axis2 == 0 ? c[0] == cb_lo[0] : axis2 == 1 ? c[1] == cb_lo[1] : c[2] == cb_lo[2]);
        } else {
            int my_bin[3] = {0, 0, 0};
            int best = 64;
            uint32_t cl = 0;
            if (depth < MI_TD_MEDIAN_DEPTH && axis2 >= 0) {
                for (int i = lane; i < MI_TD_BIN_WORDS; i += 64) bins[wv][i] = 0u;
                td_wave_sync();
                if (in)
                    for (int a = 0; a < 3; a++) {
                        const float ext = cb_hi[a] - cb_lo[a];
                        if (!(ext > 0.0f)) continue;
                        my_bin[a] = td_bin(c[a], cb_lo[a], (float)MI_TD_BINS / ext);
                        uint32_t *q = &bins[wv][(a * MI_TD_BINS + my_bin[a]) * 7];
                        atomicAdd(q, 1u);
                        for (int k = 0; k < 3; k++) { atomicMax(q + 1 + k, ~td_ord(mb.lo[k])); atomicMax(q + 4 + k, td_ord(mb.hi[k])); }
                    }
                td_wave_sync();
                best = td_choose(bins[wv], lane, cb_lo, cb_hi, true, L_lo, L_hi, R_lo, R_hi, cl);
                td_wave_sync();
            }
            if (best < 48) {
                const int axis = best >> 4, bin = best & 15;
                left = in && (axis == 0 ? my_bin[0] : axis == 1 ? my_bin[1] : my_bin[2]) <= bin;
                for (int k = 0; k < 3; k++) { L_lo[k] = __shfl(L_lo[k], best); L_hi[k] = __shfl(L_hi[k], best); R_lo[k] = __shfl(R_lo[k], best); R_hi[k] = __shfl(R_hi[k], best); }
                by_boxes = false;
            } else { // scene.cpp:235-243, in lane order
                const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                left = in && rank < m / 2;
            }
        }
        if (by_boxes)
            for (int k = 0; k < 3; k++) {
                L_lo[k] = td_wave_min(left ? mb.lo[k] : 3.0e38f); L_hi[k] = td_wave_max(left ? mb.hi[k] : -3.0e38f);
                R_lo[k] = td_wave_min(in && !left ? mb.lo[k] : 3.0e38f); R_hi[k] = td_wave_max(in && !left ? mb.hi[k] : -3.0e38f);
            }
        const unsigned long long ml = __ballot(left), mr = mask & ~ml;
        const uint32_t nl = (uint32_t)__popcll(ml), nr = m - nl;
        const int32_t self = (int32_t)(job.begin + start + nl - 1);
        int32_t ref[2];
        for (int s = 0; s < 2; s++) {
            const unsigned long long cm = s ? mr : ml;
            const uint32_t cc = s ? nr : nl, cs = s ? start + nl : start;
            ref[s] = 0;
            if (cc == 1) {
                ref[s] = ~(int32_t)(job.begin + cs);
                if ((cm >> lane) & 1ull) my_pos = cs;
                continue;
            }
            if (lane == 0) {
                st_mask[wv][sp] = cm;
                st_info[wv][sp][0] = (uint32_t)self; st_info[wv][sp][1] = (depth + 1) | (uint32_t)s << 31; st_info[wv][sp][2] = cs;
            }
            sp++;
        }
        if (lane == 0) {
            td_write_node(out_nodes, self, L_lo, L_hi, R_lo, R_hi, ref[0], ref[1], pad);
            td_link(out_nodes, parent_node, counters, self, parent, which);
        }
    }
    if ((uint32_t)lane < job.count) idx_out[job.begin + my_pos] = my_idx;
}

// Leaves were written as ~position; the primitives then moved to their slots (k_build_permute: slot_of[position]).
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_td_fix_refs(uint32_t n_nodes, float4 *nodes, const uint32_t *slot_of) {
    const uint32_t i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n_nodes) return;
    float4 r = nodes[4 * (size_t)i + 3];
    int a = __float_as_int(r.x), b = __float_as_int(r.y);
    if (a < 0) a = ~(int)slot_of[~a];
    if (b < 0) b = ~(int)slot_of[~b];
    r.x = __int_as_float(a); r.y = __int_as_float(b);
    nodes[4 * (size_t)i + 3] = r;
}

// ---- primitives into tree order ---------------------------------------------------------
// After the Morton sort the primitive records are gathered into that order (triangles and
// spheres keep their own slot ranges), so that leaves that are neighbours in the tree read
// neighbouring records; prim_id travels along, so ids and tie-breaks are unchanged.
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_is_tri(int n, const uint32_t *sorted_slot, uint32_t n_tri, uint32_t *flag) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i < n) flag[i] = sorted_slot[i] < n_tri ? 1u : 0u;
}
struct PrimArrays {
    float4 *tri, *tri_normal, *sph;
    uint32_t *tri_mat, *sph_mat;
    uint8_t *tri_has_vmat;
    int32_t *prim_id;
    BuildBox *pb;
};
__global__ void __launch_bounds__(MI_BUILD_BLOCK)
k_build_permute(int n, uint32_t n_tri, uint32_t *sorted_slot, const uint32_t *tri_rank, PrimArrays src, PrimArrays dst) {
    const int i = blockIdx.x * MI_BUILD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t old = sorted_slot[i];
    uint32_t now;
    if (old < n_tri) {
        now = tri_rank[i];
        for (int k = 0; k < 3; k++) dst.tri[3 * (size_t)now + k] = src.tri[3 * (size_t)old + k]; // 48-B records (mi_kernels.h)
        dst.tri_normal[now] = src.tri_normal[old];
        for (int k = 0; k < 3; k++) dst.tri_mat[3 * (size_t)now + k] = src.tri_mat[3 * (size_t)old + k];
        dst.tri_has_vmat[now] = src.tri_has_vmat[old];
    } else {
        const uint32_t r = (uint32_t)i - tri_rank[i]; // spheres before position i
        now = n_tri + r;
        dst.sph[r] = src.sph[old - n_tri];
        dst.sph_mat[r] = src.sph_mat[old - n_tri];
    }
    dst.prim_id[now] = src.prim_id[old];
    dst.pb[now] = src.pb[old];
    sorted_slot[i] = now;
}

// ---- treelet: copies of the top of the 4-wide tree, appended behind it -------------------
// One thread: breadth-first from the root, at most `max_nodes` nodes within `max_levels`
// levels below the root; refs among the copied nodes are redirected to the copies.  The
// traversal kernels keep nodes [base, base + count) in LDS (DevScene::treelet_*).  Mirrors
// the host-side construction in install_bvh4() (mi_tracer.hip).  out[0] = count.
__global__ void k_build_treelet(float *nodes4, int root4, int base, int max_nodes, int max_levels, int empty_ref, int *out) {
    __shared__ int top[128];
    __shared__ int n_top_s;
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) { // the walk is a chain of dependent reads: one lane, one 16-B read per node
        int n_top = 1, level_end = 1, lvl = 0;
        top[0] = root4;
        for (int q = 0; q < n_top && lvl < max_levels; q++) {
            const float4 r4 = *(const float4 *)(nodes4 + 32 * (size_t)top[q] + 24);
            const int rr[4] = {__float_as_int(r4.x), __float_as_int(r4.y), __float_as_int(r4.z), __float_as_int(r4.w)};
            for (int k = 0; k < 4; k++)
                if (rr[k] >= 0 && rr[k] != empty_ref && n_top < max_nodes && n_top < 128) top[n_top++] = rr[k];
            if (q + 1 == level_end) { level_end = n_top; lvl++; }
        }
        n_top_s = n_top;
        out[0] = n_top;
    }
    __syncthreads();
    const int n_top = n_top_s;
    for (int e = threadIdx.x; e < n_top * 32; e += blockDim.x) { // the copies: every word on its own
        const int k = e >> 5, j = e & 31;
        float v = nodes4[32 * (size_t)top[k] + j];
        if (j >= 24 && j < 28) {
            const int r = __float_as_int(v);
            if (r >= 0 && r != empty_ref)
                for (int t = 0; t < n_top; t++)
                    if (top[t] == r) { v = __int_as_float(base + t); break; }
        }
        nodes4[32 * (size_t)(base + k) + j] = v;
    }
}

// ------------------------------------------------------------ quantised 4-wide nodes
// fp32 4-wide nodes (128 B: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] refs[4] pad) -> the 64-B
// nodes the traversal kernels read (DevScene::qnodes).  One thread per node.  A plane b on axis a
// becomes n = (b - qb[a]) / qs[a], which the caller's choice of qs, qb keeps inside [0.51, 1.99]
// for every plane inside the scene box; the code is bits 23..8 of the float n rounded DOWN for a
// lower plane and UP for an upper plane after moving it MI_Q_MARGIN (a quarter or an eighth of a code
// step; ~2e-6 of the scene extent, an order of magnitude above the slab test's rounding for ray origins
// within a few extents of the scene) and MI_Q_PAD whole codes outward.  Planes outside
// the scene box are clamped to it (what lies outside the root's box cannot be reached anyway).
// Empty seats get an inverted box.
__device__ __forceinline__ uint32_t q_encode(float b, float qs, float qb, bool upper) {
    double n = ((double)b - (double)qb) / (double)qs + (upper ? MI_Q_MARGIN : -MI_Q_MARGIN);
    n = n < 0.5 ? 0.5 : (n > 1.999 ? 1.999 : n);
    const float f = upper ? __double2float_ru(n) : __double2float_rd(n);
    const uint32_t m = __float_as_uint(f) - 0x3F000000u; // 24 bits: [0.5, 2) is two binades
    uint32_t q = upper ? (m + 0xFFu) >> 8 : m >> 8;
    if (upper) q = q + MI_Q_PAD > 0xFFFFu ? 0xFFFFu : q + MI_Q_PAD;
    else q = q > MI_Q_PAD ? q - MI_Q_PAD : 0u;
    return q;
}
__global__ void __launch_bounds__(256)
k_quantize_nodes4(const float *nodes4, int n_nodes, float qs0, float qs1, float qs2, float qb0, float qb1, float qb2, int empty_ref, uint4 *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    const float *nd = nodes4 + 32 * (size_t)i;
    const float qs[3] = {qs0, qs1, qs2}, qb[3] = {qb0, qb1, qb2};
    uint32_t P[3][4];
    int refs[4];
    for (int c = 0; c < 4; c++) {
        refs[c] = __float_as_int(nd[24 + c]);
        for (int a = 0; a < 3; a++) {
            if (refs[c] == empty_ref) {
                P[a][c] = 0x0000FFFFu; // lower plane at the top code, upper plane at code 0
            } else {
                const uint32_t lo = q_encode(nd[4 * a + c], qs[a], qb[a], false), hi = q_encode(nd[12 + 4 * a + c], qs[a], qb[a], true);
                P[a][c] = lo | hi << 16;
            }
        }
    }
    out[4 * (size_t)i + 0] = make_uint4(P[0][0], P[0][1], P[0][2], P[0][3]);
    out[4 * (size_t)i + 1] = make_uint4(P[1][0], P[1][1], P[1][2], P[1][3]);
    out[4 * (size_t)i + 2] = make_uint4(P[2][0], P[2][1], P[2][2], P[2][3]);
    out[4 * (size_t)i + 3] = make_uint4((uint32_t)refs[0], (uint32_t)refs[1], (uint32_t)refs[2], (uint32_t)refs[3]);
}

