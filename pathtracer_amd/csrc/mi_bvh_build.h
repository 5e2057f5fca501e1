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
            lo += 0.0f; hi += 0.0f; clo += 0.0f; chi += 0.0f;
            if (lo >= 0) atomicMin((int *)&scene_box[k], __float_as_int(lo)); else atomicMax((unsigned int *)&scene_box[k], __float_as_uint(lo));
            if (hi >= 0) atomicMax((int *)&scene_box[3 + k], __float_as_int(hi)); else atomicMin((unsigned int *)&scene_box[3 + k], __float_as_uint(hi));
            if (clo >= 0) atomicMin((int *)&scene_box[8 + k], __float_as_int(clo)); else atomicMax((unsigned int *)&scene_box[8 + k], __float_as_uint(clo));
            if (chi >= 0) atomicMax((int *)&scene_box[11 + k], __float_as_int(chi)); else atomicMin((unsigned int *)&scene_box[11 + k], __float_as_uint(chi));
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
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int top[128];
    int n_top = 1, level_end = 1, lvl = 0;
    top[0] = root4;
    for (int q = 0; q < n_top && lvl < max_levels; q++) {
        for (int k = 0; k < 4; k++) {
            const int r = __float_as_int(nodes4[32 * (size_t)top[q] + 24 + k]);
            if (r >= 0 && r != empty_ref && n_top < max_nodes) top[n_top++] = r;
        }
        if (q + 1 == level_end) { level_end = n_top; lvl++; }
    }
    for (int k = 0; k < n_top; k++) {
        const float *src = nodes4 + 32 * (size_t)top[k];
        float *dst = nodes4 + 32 * (size_t)(base + k);
        for (int j = 0; j < 32; j++) dst[j] = src[j];
        for (int j = 0; j < 4; j++) {
            const int r = __float_as_int(src[24 + j]);
            if (r < 0 || r == empty_ref) continue;
            for (int t = 0; t < n_top; t++)
                if (top[t] == r) { dst[24 + j] = __int_as_float(base + t); break; }
        }
    }
    out[0] = n_top;
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

