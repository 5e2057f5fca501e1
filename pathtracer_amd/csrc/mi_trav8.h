// mi_trav8.h — traversal over the 8-WIDE tree (compile with -DMI_WIDE8; included by mi_kernels.h in place of the 4-wide
// walk).  Same results as any other walk: the closest hit does not depend on the tree (tie rule in mi_kernels.h).
//
// Node, 80 B = five 16-B loads (built and quantised on the host, mi_wide8.h):
//   q0 = origin.x, origin.y, origin.z (fp32), ex | ey << 8 | ez << 16 | imask << 24
//   q1 = child_base, tri_base, sph_base, tmask | smask << 8
//   q2 = lo.x[0..3], lo.x[4..7], lo.y[0..3], lo.y[4..7]      one byte per child seat: the plane's code q on the node's
//   q3 = lo.z[0..3], lo.z[4..7], hi.x[0..3], hi.x[4..7]      own grid, plane = origin + q * 2^e (e: signed byte)
//   q4 = hi.y[0..3], hi.y[4..7], hi.z[0..3], hi.z[4..7]      (lower planes rounded down, upper planes up)
// Seat s (0..7) holds an interior child (bit s of imask), a triangle (tmask), a sphere (smask) or nothing.  The
// interior children of a node are consecutive nodes: child of seat s = child_base + popcount(imask & ((1 << s) - 1));
// the primitives of its leaf seats are consecutive slots of their class (the uploader permutes the primitives so):
// triangle of seat s = tri_base + popcount(tmask & below s), sphere = n_tri + sph_base + popcount(smask & below s).
// Nodes are stored breadth first, so the first `treelet_count` nodes ARE the top of the tree: they are kept in LDS.
//
// Order of traversal without a sort (Ylitie, Karras, Laine 2017): the builder seats children by octant — bit a of the seat
// index set = the child lies towards +a in its node — so for a ray whose direction signs are `oct` (bit a set: d[a] < 0)
// seats are visited in ascending p = s ^ oct.  The eight hit bits of a visit are permuted into that order once (three
// conditional swaps on a word that carries hit, interior, triangle and sphere masks in its four bytes); the interior
// hits become ONE stack entry per node (child_base, imask | hits' << 8), the leaf hits stay with the lane and are tested
// before the lane goes on (nearest-first by octant), one primitive per leaf phase.
#pragma once

#define MI_NODE8_BYTES 80u
#define MI_LDS_STACK 8        // group entries per lane held in LDS (one entry per level of the 8-wide tree)
#define MI_MAX_STACK 32       // a tree of MI_MAX_DEPTH binary levels has at most 22 levels of 8-wide nodes
#define MI_STACK_ELEM_BYTES 8
#ifndef MI_TREELET
#define MI_TREELET 73         // root + 8 + 64: the first nodes in breadth-first order (5.8 KB)
#endif
#define MI_TREELET_U4 (MI_TREELET * 5)

typedef unsigned int mi_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) mi_v2u mi_lds_stack; // LDS-qualified: stack accesses can only be ds_read_b64 / ds_write_b64
typedef mi_v2u mi_stack_elem;
typedef mi_v2u mi_ovf_elem;

struct Lane {
    f3 o, d;
    f3 inv;         // reciprocal direction (clamped, see lane_start)
    uint32_t oct;   // bit a set: d[a] < 0
    Hit h;
    int cur;        // node to visit, MI_DONE: none
    uint32_t g_base, g_bits;        // the group the lane draws its next node from: child_base, imask | hits' << 8 (hits' in visiting order)
    uint32_t l_tri, l_sph, l_bits;  // leaves of the node visited last that are still to be tested: tri_base, sph_base, tmask | smask << 8 | pending' << 16
    int sp;
};
__device__ __forceinline__ void lane_idle(Lane &L) {
    L.cur = MI_DONE; L.sp = 0; L.g_base = 0; L.g_bits = 0; L.l_tri = L.l_sph = L.l_bits = 0; L.oct = 0;
    L.h = Hit{MI_NO_HIT, -1};
    L.o = L.d = L.inv = mk3(0, 0, 0);
}
__device__ __forceinline__ bool lane_wants_leaf(const Lane &L) { return (L.l_bits >> 16) != 0u; }
__device__ __forceinline__ bool lane_wants_node(const Lane &L) { return (L.l_bits >> 16) == 0u && L.cur != MI_DONE; }
__device__ __forceinline__ bool lane_done(const Lane &L) { return (L.l_bits >> 16) == 0u && L.cur == MI_DONE; }
__device__ __forceinline__ void lane_finish(Lane &L) { L.cur = MI_DONE; L.sp = 0; L.g_bits = 0; L.l_bits = 0; } // any-hit: the query is answered

__device__ __forceinline__ void lane_start(const DevScene &sc, Lane &L, f3 o, f3 d, float t0 = MI_NO_HIT) {
    L.o = o;
    L.d = d;
    // 1/d feeds only the slab test, a culling test with margins: hardware reciprocal clamped to +-1e18 (as in the 4-wide
    // walk: an axis-parallel ray is culled like one that drifts 1e-18 per unit of t; no inf - inf, no 0 * inf).
    L.inv = mk3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e18f, 1e18f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e18f, 1e18f),
                __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e18f, 1e18f));
    L.oct = (L.inv.x < 0.0f ? 1u : 0u) | (L.inv.y < 0.0f ? 2u : 0u) | (L.inv.z < 0.0f ? 4u : 0u);
    L.h = Hit{t0, -1};
    L.sp = 0;
    L.g_base = 0; L.g_bits = 0;
    L.l_tri = L.l_sph = 0; L.l_bits = 0;
    const uint32_t n_prim = sc.n_tri + sc.n_sph;
    if (n_prim == 0) {
        L.cur = MI_DONE;                       // bvh.cu:174-176: nothing to hit
    } else if (sc.root_ref < 0) {              // bvh.cu:180-183: a single primitive, no node: one pending leaf in seat 0
        L.cur = MI_DONE;
        L.l_bits = (sc.n_tri ? 1u : 0x100u) | (1u << (16 + L.oct));
    } else {
        L.cur = 0;                             // the root is node 0 (breadth-first order)
    }
}

// Bits of every byte of w permuted by seat -> visiting order: bit p of the result = bit (p ^ oct) of the input.
__device__ __forceinline__ uint32_t oct_permute(uint32_t w, uint32_t oct) {
    const uint32_t w1 = ((w >> 1) & 0x55555555u) | ((w & 0x55555555u) << 1);
    w = (oct & 1u) ? w1 : w;
    const uint32_t w2 = ((w >> 2) & 0x33333333u) | ((w & 0x33333333u) << 2);
    w = (oct & 2u) ? w2 : w;
    const uint32_t w4 = ((w >> 4) & 0x0f0f0f0fu) | ((w & 0x0f0f0f0fu) << 4);
    w = (oct & 4u) ? w4 : w;
    return w;
}

struct NodeBase {
    uint64_t gbase, lbase; // global address of node 0; generic (flat) address of the LDS copy of node 0
};
__device__ __forceinline__ NodeBase node_base(const DevScene &sc, const uint4 *treelet) {
    return NodeBase{(uint64_t)(uintptr_t)sc.qnodes, (uint64_t)(uintptr_t)treelet};
}
__device__ __forceinline__ const char *node_addr(const DevScene &sc, int cur, const NodeBase &nb) {
    return (const char *)(uintptr_t)((cur < sc.treelet_count ? nb.lbase : nb.gbase) + (uint64_t)(uint32_t)cur * MI_NODE8_BYTES);
}
// every thread of the workgroup; the caller synchronises
__device__ __forceinline__ void treelet_load(const DevScene &sc, uint4 *lds) {
    for (int i = threadIdx.x; i < sc.treelet_count * 5; i += MI_BLOCK) lds[i] = sc.qnodes[i];
}

// The lane's next node: from its current group, or from the group on top of the stack when that one is used up.
// `top` is the stack's top entry, read by the caller (unconditionally, clamped index) so that the LDS read is issued
// early; overflow entries (sp > MI_LDS_STACK) live in global memory.
__device__ __forceinline__ void lane_pick(Lane &L, mi_lds_stack *stack, mi_ovf_elem *my_ovf, size_t ovf_stride) {
    if ((L.g_bits >> 8) == 0u) {
        if (L.sp == 0) { L.cur = MI_DONE; return; }
        L.sp--;
        mi_v2u top = stack[(L.sp < MI_LDS_STACK ? L.sp : MI_LDS_STACK - 1) * MI_BLOCK];
        if (L.sp >= MI_LDS_STACK) top = my_ovf[(size_t)(L.sp - MI_LDS_STACK) * ovf_stride];
        L.g_base = top.x; L.g_bits = top.y;
    }
    uint32_t hp = L.g_bits >> 8;
    const uint32_t p = (uint32_t)__builtin_ctz(hp);
    hp &= hp - 1u;
    const uint32_t s = p ^ L.oct;
    const uint32_t imask = L.g_bits & 0xffu;
    L.cur = (int)(L.g_base + (uint32_t)__builtin_popcount(imask & ((1u << s) - 1u)));
    L.g_bits = imask | (hp << 8);
}
__device__ __forceinline__ void lane_push_group(Lane &L, mi_lds_stack *stack, mi_ovf_elem *my_ovf, size_t ovf_stride) {
    mi_v2u e; e.x = L.g_base; e.y = L.g_bits;
    if (L.sp < MI_LDS_STACK) stack[L.sp * MI_BLOCK] = e;
    else my_ovf[(size_t)(L.sp - MI_LDS_STACK) * ovf_stride] = e;
    L.sp++;
}

__device__ __forceinline__ float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); } // v_cvt_f32_ubyte<k>

// Box tests of a node whose 80 B are in registers, the hits' split into the lane's pending leaves and a new group, and
// the choice of the next node.
__device__ __forceinline__ void node_process(const DevScene &sc, Lane &L, mi_lds_stack *stack, mi_ovf_elem *my_ovf, size_t ovf_stride, uint4 q0, uint4 q1, uint4 q2,
                                             uint4 q3, uint4 q4) {
    const uint32_t hw = q0.w;
    // distance to the plane of code q on axis a: q * sa + ob, sa = 2^e / d (exact scaling), ob = (origin - o) / d
    const float sax = __builtin_ldexpf(L.inv.x, (int)(int8_t)(hw & 0xffu)), say = __builtin_ldexpf(L.inv.y, (int)(int8_t)((hw >> 8) & 0xffu)),
                saz = __builtin_ldexpf(L.inv.z, (int)(int8_t)((hw >> 16) & 0xffu));
    const float obx = (__uint_as_float(q0.x) - L.o.x) * L.inv.x, oby = (__uint_as_float(q0.y) - L.o.y) * L.inv.y, obz = (__uint_as_float(q0.z) - L.o.z) * L.inv.z;
    const float ts = L.h.t;
    // near and far planes per axis: the lower ones when d > 0
    const bool nx = (L.oct & 1u) != 0u, ny = (L.oct & 2u) != 0u, nz = (L.oct & 4u) != 0u;
    const uint32_t NX[2] = {nx ? q3.z : q2.x, nx ? q3.w : q2.y}, FX[2] = {nx ? q2.x : q3.z, nx ? q2.y : q3.w};
    const uint32_t NY[2] = {ny ? q4.x : q2.z, ny ? q4.y : q2.w}, FY[2] = {ny ? q2.z : q4.x, ny ? q2.w : q4.y};
    const uint32_t NZ[2] = {nz ? q4.z : q3.x, nz ? q4.w : q3.y}, FZ[2] = {nz ? q3.x : q4.z, nz ? q3.y : q4.w};
    uint32_t m = 0;
#pragma unroll
    for (int c = 7; c >= 0; c--) {
        const int w = c >> 2, k = c & 3;
        const float n0 = __builtin_fmaf(ubyte_f(NX[w], k), sax, obx), n1 = __builtin_fmaf(ubyte_f(NY[w], k), say, oby), n2 = __builtin_fmaf(ubyte_f(NZ[w], k), saz, obz);
        const float f0 = __builtin_fmaf(ubyte_f(FX[w], k), sax, obx), f1 = __builtin_fmaf(ubyte_f(FY[w], k), say, oby), f2 = __builtin_fmaf(ubyte_f(FZ[w], k), saz, obz);
        const float tmin = vmax3(n0, n1, n2);
        const float tmx = vmin(ts, vmin3(f0, f1, f2));
        m = (m << 1) | ((tmin <= tmx && tmx > 0) ? 1u : 0u);
    }
    const uint32_t imask = hw >> 24, leafmasks = q1.w & 0xffffu;
    // an empty seat's box is inverted on every axis and cannot be hit by a ray with usable reciprocals; the mask makes that
    // hold for every input (a seat that is not there must never be followed)
    m &= imask | (leafmasks & 0xffu) | (leafmasks >> 8);
    const uint32_t wp = oct_permute(m | (imask << 8) | (leafmasks << 16), L.oct);
    const uint32_t hp = wp & 0xffu;
    const uint32_t inner = hp & (wp >> 8), leaf = hp & ((wp >> 16) | (wp >> 24));
    L.l_tri = q1.y; L.l_sph = q1.z;
    L.l_bits = leafmasks | (leaf << 16);
    if (inner != 0u) {
        if ((L.g_bits >> 8) != 0u) lane_push_group(L, stack, my_ovf, ovf_stride);
        L.g_base = q1.x;
        L.g_bits = imask | (inner << 8);
    }
    lane_pick(L, stack, my_ovf, ovf_stride);
}
__device__ __forceinline__ void node_phase(const DevScene &sc, Lane &L, mi_lds_stack *stack, mi_ovf_elem *my_ovf, size_t ovf_stride, const NodeBase &nb) {
    const char *nd = node_addr(sc, L.cur, nb);
    const uint4 q0 = *(const uint4 *)(nd), q1 = *(const uint4 *)(nd + 16), q2 = *(const uint4 *)(nd + 32), q3 = *(const uint4 *)(nd + 48), q4 = *(const uint4 *)(nd + 64);
    node_process(sc, L, stack, my_ovf, ovf_stride, q0, q1, q2, q3, q4);
}
// The pending leaf that comes first in visiting order: its primitive slot; the leaf is struck off the lane's list.
__device__ __forceinline__ int lane_take_leaf(const DevScene &sc, Lane &L) {
    uint32_t pend = L.l_bits >> 16;
    const uint32_t p = (uint32_t)__builtin_ctz(pend);
    pend &= pend - 1u;
    const uint32_t s = p ^ L.oct, below = (1u << s) - 1u;
    const uint32_t tmask = L.l_bits & 0xffu, smask = (L.l_bits >> 8) & 0xffu;
    const bool is_sph = ((smask >> s) & 1u) != 0u;
    const int slot = is_sph ? (int)(sc.n_tri + L.l_sph + (uint32_t)__builtin_popcount(smask & below)) : (int)(L.l_tri + (uint32_t)__builtin_popcount(tmask & below));
    L.l_bits = (L.l_bits & 0xffffu) | (pend << 16);
    return slot;
}
__device__ __forceinline__ void leaf_phase(const DevScene &sc, Lane &L, mi_lds_stack *stack, mi_ovf_elem *my_ovf, size_t ovf_stride) { // one primitive
    prim_test(sc, lane_take_leaf(sc, L), L.o, L.d, L.h);
}
// Leaf lanes and node lanes in ONE step (draining waves and k_tail): every lane with work fetches what it needs — a node
// (80 B) or a primitive record (48 B; the sphere array is padded) — with the same loads, the wave waits once.
template <bool ANY>
__device__ __forceinline__ void merged_step(const DevScene &sc, Lane &L, mi_lds_stack *stack, mi_ovf_elem *my_ovf, size_t ovf_stride, const NodeBase &nb, bool want_leaf,
                                            bool want_node) {
    if (!(want_leaf || want_node)) return;
    int slot = 0;
    if (want_leaf) slot = lane_take_leaf(sc, L);
    const bool is_sph = want_leaf && (uint32_t)slot >= sc.n_tri;
    const char *at = want_node ? node_addr(sc, L.cur, nb)
                               : (is_sph ? (const char *)sc.sph + ((uint32_t)(slot - (int)sc.n_tri) << 4) : (const char *)sc.tri + (size_t)(uint32_t)slot * MI_TRI_BYTES);
    const uint4 r0 = *(const uint4 *)(at), r1 = *(const uint4 *)(at + 16), r2 = *(const uint4 *)(at + 32);
    uint4 r3 = make_uint4(0u, 0u, 0u, 0u), r4 = make_uint4(0u, 0u, 0u, 0u);
    if (want_node) { r3 = *(const uint4 *)(at + 48); r4 = *(const uint4 *)(at + 64); }
    auto f4 = [](uint4 r) { return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)); };
    if (want_leaf) {
        if (is_sph) sphere_test_rec(sc, slot, f4(r0), L.o, L.d, L.h);
        else tri_test_rec(sc, slot, f4(r0), f4(r1), f4(r2), L.o, L.d, L.h);
        if (ANY && L.h.slot >= 0) lane_finish(L);
    } else {
        node_process(sc, L, stack, my_ovf, ovf_stride, r0, r1, r2, r3, r4);
    }
}
