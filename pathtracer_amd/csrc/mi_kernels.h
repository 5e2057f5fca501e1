// mi_kernels.h — hand-written HIP kernels of the wavefront path tracer (gfx950).
//
// Stages (one launch each per bounce level; all paths of a batch are at the same
// level k in iteration k):
//   k_generate : camera rays for a batch of (pixel, sample) slots   [camera.cu:6-16, CpuTracer.cu:99-109]
//   k_extend   : BVH closest hit, LDS short stack, per-wave dynamic  [bvh.cu:3-10,178-207, bbox.cu:15-34,
//                work fetch                                           trimesh.cu:87-121, sphere.cu:5-35]
//   k_shade    : material branch, next ray, termination, wave-ballot [CpuTracer.cu:119-186, :37-53]
//                compaction of survivors into the next queue
//   k_resolve  : per-pixel mean in fixed sample order + Hable tonemap [CpuTracer.cu:108, tracer.cu:20-30,95-100]
//
// Numerics contract: fp32, no FMA contraction (-ffp-contract=off),
// IEEE divide/sqrt, operations in the order of the cited reference lines, so the
// results are bit-identical to oracle/oracle.cpp on the same inputs.  The w lane
// of positions/directions/normals is identically +0 in every scene the loader
// can produce (vec.h:18, Parser.cu:339,777-786), so those are carried as 3 lanes;
// throughput, weights and radiance keep all 4 lanes (vec.h:15 broadcasts into w).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_BLOCK 256          // threads per workgroup = 4 waves of 64
#define MI_LDS_STACK 16       // traversal stack entries per lane held in LDS
#define MI_MAX_STACK 64       // deepest tree the uploader accepts
#define MI_NO_HIT 12345678.9f // ray.h:14
#define MI_EPS 0.00001f       // ray.h:12

struct DevScene {
    const float4 *nodes; // 4 x float4 per interior node (layout: include/mi_tracer.h)
    const float4 *tri;   // 4 x float4 per triangle slot: v0, uInv, vInv, nInv
    const float4 *tri_normal;
    const uint32_t *tri_mat; // 3 per triangle
    const uint8_t *tri_has_vmat;
    const float4 *sph; // (cx, cy, cz, r)
    const uint32_t *sph_mat;
    const float4 *mats; // 5 x float4 per material
    const int32_t *prim_id;
    int32_t root_ref;
    uint32_t n_tri, n_sph;
    float4 eye, u, v, look;
};

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 xyz(const float4 &a) { return f3{a.x, a.y, a.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float f) { return f3{a.x * f, a.y * f, a.z * f}; }
__device__ __forceinline__ f3 operator*(float f, f3 a) { return f3{f * a.x, f * a.y, f * a.z}; }
// vec.h:188-190 sums lanes left to right; the w*w product is +0 and adding it
// changes nothing (only the sign of an exact zero sum, which no test observes).
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { // vec.h:204-208
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// vec.h:200-202 with rsqrtf(x) := 1/sqrt(x), both IEEE (oracle deviation D2).
__device__ __forceinline__ f3 normalize3(f3 a) { return a * (1.0f / __builtin_sqrtf(dot3(a, a))); }

__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float f) { return make_float4(a.x * f, a.y * f, a.z * f, a.w * f); }
__device__ __forceinline__ float4 mul4(float f, float4 a) { return make_float4(f * a.x, f * a.y, f * a.z, f * a.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float length2_4(float4 a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG shared with the oracle (deviation D1): counter = (path_lo,
// path_hi, level, 0), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// sin/cos of p in [0, 2*pi]: quadrant reduction with a two-constant pi/2 and
// degree-7/8 polynomials in explicit fmaf form (oracle deviation D3, mode 1).
__device__ __forceinline__ void sincos_poly(float p, float &s_out, float &c_out) {
    float k = __builtin_rintf(p * 0.636619772f);
    int q = (int)k;
    float r = __builtin_fmaf(-k, 1.5707962513e+0f, p);
    r = __builtin_fmaf(-k, 7.5497894159e-8f, r);
    float z = r * r;
    float sp = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    switch (q & 3) {
        case 0: s_out = s; c_out = c; break;
        case 1: s_out = c; c_out = -s; break;
        case 2: s_out = -s; c_out = -c; break;
        default: s_out = -c; c_out = s; break;
    }
}

// ------------------------------------------------------------ shard geometry
struct ShardGeom {
    int32_t width, height, tile_w, tile_h, tiles_x, n_tiles, rank, count;
    int32_t samples; // per dimension
};
// Local pixel q (tile-major, row-major inside the tile; the order documented for
// mi_render_device) -> image pixel.  Returns false for pixels of edge tiles that
// fall outside the image.  32-bit arithmetic: the host rejects images whose padded
// pixel count does not fit (64-bit integer division is ~100 instructions on gfx950).
__device__ __forceinline__ bool local_to_pixel(const ShardGeom &g, uint32_t q, int &px, int &py) {
    uint32_t tpix = (uint32_t)(g.tile_w * g.tile_h);
    uint32_t j = q / tpix;
    uint32_t off = q - j * tpix;
    uint32_t gt = (uint32_t)g.rank + j * (uint32_t)g.count;
    uint32_t ty = gt / (uint32_t)g.tiles_x, tx = gt - ty * (uint32_t)g.tiles_x;
    uint32_t oy = off / (uint32_t)g.tile_w, ox = off - oy * (uint32_t)g.tile_w;
    px = (int)(tx * (uint32_t)g.tile_w + ox);
    py = (int)(ty * (uint32_t)g.tile_h + oy);
    return gt < (uint32_t)g.n_tiles && px < g.width && py < g.height;
}
// Path slots of a batch walk each tile in 8x8 sub-blocks so that one wave (64
// consecutive slots) covers an 8x8 pixel patch: coherent primary rays.  `lp` is
// the position inside the shard; returns the local pixel it renders.
__device__ __forceinline__ uint32_t slot_to_local(const ShardGeom &g, uint32_t lp) {
    if ((g.tile_w & 7) || (g.tile_h & 7)) return lp;
    uint32_t tpix = (uint32_t)(g.tile_w * g.tile_h);
    uint32_t j = lp / tpix;
    uint32_t off = lp - j * tpix;
    uint32_t sub = off >> 6, in = off & 63u;
    uint32_t sbw = (uint32_t)g.tile_w >> 3;
    uint32_t sby = sub / sbw, sbx = sub - sby * sbw;
    uint32_t x = sbx * 8u + (in & 7u), y = sby * 8u + (in >> 3);
    return j * tpix + y * (uint32_t)g.tile_w + x;
}

// Queue append (ray-queue compaction): wave ballot + popcount prefix inside each wave,
// the four wave totals combined through LDS, ONE global atomic per workgroup.  Must be
// called by every thread of the workgroup (it contains barriers).
__device__ __forceinline__ void queue_append(bool alive, uint32_t slot, uint32_t *queue, uint32_t *count) {
    __shared__ uint32_t wave_cnt[MI_BLOCK / 64];
    __shared__ uint32_t block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long mask = __ballot(alive);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < MI_BLOCK / 64; w++) total += wave_cnt[w];
        block_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    uint32_t prefix = block_base;
    for (int w = 0; w < wave; w++) prefix += wave_cnt[w];
    if (alive) queue[prefix + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
    __syncthreads(); // wave_cnt / block_base are reused by the next call
}

// ------------------------------------------------------------------ generate
// Camera ray of batch slot `slot` (= sample * npix + position in batch).  There is no
// generate kernel and no stored primary ray: level 0 of k_extend and of k_shade both
// call this (camera.cu:6-16 is ~30 instructions; a stored ray is 32 B written + 64 B
// read).  Returns false for slots of edge tiles that fall outside the image.
__device__ __forceinline__ bool primary_ray(const DevScene &sc, const ShardGeom &g, uint32_t pix0, uint32_t npix, uint32_t slot,
                                            f3 &o, f3 &dir, int &px, int &py, uint32_t &samp) {
    samp = slot / npix;
    uint32_t lp = slot - samp * npix;
    if (!local_to_pixel(g, slot_to_local(g, pix0 + lp), px, py)) return false;
    int sx = (int)(samp / (uint32_t)g.samples), sy = (int)samp - sx * g.samples;
    // CpuTracer.cu:103-105
    float xx = float(px) - 0.5f + (1 + 2 * sx) / (2.0f * g.samples);
    float yy = float(py) - 0.5f + (1 + 2 * sy) / (2.0f * g.samples);
    float x = xx / float(g.width), y = yy / float(g.height);
    x -= 0.5f; // camera.cu:11-12
    y -= 0.5f;
    dir = normalize3(xyz(sc.look) + x * xyz(sc.u) + y * xyz(sc.v)); // camera.cu:13
    o = xyz(sc.eye);
    return true;
}

// Rays for mi_trace_primary: full image, index (j*W+i)*spp + sx*s+sy.
__global__ void __launch_bounds__(MI_BLOCK)
k_generate_primary(DevScene sc, int width, int height, int samples, int64_t first, int64_t n, float4 *ray_o, float4 *ray_d) {
    int64_t i = (int64_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    if (i >= n) return;
    int64_t gidx = first + i;
    int spp = samples * samples;
    int samp = (int)(gidx % spp);
    int64_t pix = gidx / spp;
    int px = (int)(pix % width), py = (int)(pix / width);
    int sx = samp / samples, sy = samp - sx * samples;
    float xx = float(px) - 0.5f + (1 + 2 * sx) / (2.0f * samples);
    float yy = float(py) - 0.5f + (1 + 2 * sy) / (2.0f * samples);
    float x = xx / float(width), y = yy / float(height);
    x -= 0.5f;
    y -= 0.5f;
    f3 dir = normalize3(xyz(sc.look) + x * xyz(sc.u) + y * xyz(sc.v));
    ray_o[i] = make_float4(sc.eye.x, sc.eye.y, sc.eye.z, 0.0f);
    ray_d[i] = make_float4(dir.x, dir.y, dir.z, 0.0f);
}

// ------------------------------------------------------------------- extend
struct Hit {
    float t, u, v;
    int slot;
};

// Exact-t ties are resolved as a visit in ascending primitive id would resolve
// them (triangle accepts t <= best, trimesh.cu:96; sphere needs t < best,
// sphere.cu:22,28), so the result does not depend on traversal order.
__device__ __forceinline__ bool tie_allows(const DevScene &sc, int best_slot, int cand_slot, bool cand_is_tri) {
    if (best_slot < 0) return cand_is_tri;
    int cid = sc.prim_id[cand_slot], bid = sc.prim_id[best_slot];
    bool best_is_sphere = (uint32_t)best_slot >= sc.n_tri;
    if (cand_is_tri) return best_is_sphere ? true : (cid > bid);
    return best_is_sphere ? (cid < bid) : false;
}

// trimesh.cu:87-121.  Evaluated without early exits: all four 16-B loads of the 64-B
// record are issued together (one memory round trip instead of two) and the reject
// tests of :92, :96 and :104 — each a pure function of the same values — are combined
// at the end.  The accepted (t, u, v) are the reference's.
__device__ __forceinline__ void tri_test(const DevScene &sc, int slot, f3 o, f3 d, Hit &h) {
    const char *rec = (const char *)sc.tri + ((uint32_t)slot << 6);
    float4 v0 = *(const float4 *)(rec), ui = *(const float4 *)(rec + 16), vi = *(const float4 *)(rec + 32), ni = *(const float4 *)(rec + 48);
    f3 p = o - xyz(v0);
    float pz = dot3(xyz(ni), p);
    float dz = dot3(xyz(ni), d);
    float t = -pz / dz;
    f3 hit = p + t * d;
    float u = dot3(xyz(ui), hit);
    float v = dot3(xyz(vi), hit);
    bool ok = ((pz < 0) != (dz < 0)) && !(t > h.t) && !(u < 0 || u > 1 || v < 0 || v > 1 || u + v > 1);
    if (ok && t == h.t) ok = tie_allows(sc, h.slot, slot, true);
    if (ok) { h.t = t; h.u = u; h.v = v; h.slot = slot; }
}

__device__ __forceinline__ void sphere_test(const DevScene &sc, int slot, f3 o, f3 d, Hit &h) { // sphere.cu:5-35
    float4 cr = sc.sph[slot - (int)sc.n_tri];
    f3 dir = normalize3(d);
    f3 to = xyz(cr) - o;
    float dt = dot3(dir, to);
    float disc = dt * dt - dot3(to, to) + cr.w * cr.w;
    if (disc < 0) return;
    disc = __builtin_sqrtf(disc);
    float t2 = dt + disc;
    if (t2 < MI_EPS) return;
    float t1 = dt - disc;
    float t;
    if (t1 > MI_EPS && (t1 < h.t || (t1 == h.t && tie_allows(sc, h.slot, slot, false)))) t = t1;
    else if (t2 < h.t || (t2 == h.t && tie_allows(sc, h.slot, slot, false))) t = t2;
    else return;
    h.t = t; h.u = 0.0f; h.v = 0.0f; h.slot = slot;
}

__device__ __forceinline__ void prim_test(const DevScene &sc, int slot, f3 o, f3 d, Hit &h) { // geometry.h:26-29
    if ((uint32_t)slot >= sc.n_tri) sphere_test(sc, slot, o, d, h);
    else tri_test(sc, slot, o, d, h);
}

// bbox.cu:15-34.  EXACT keeps the reference's `(a<b?a:b)` macros (a NaN operand
// yields the second operand); the fast form uses v_min/v_max, which is the same
// function whenever no operand is NaN — guaranteed when 1/d is finite on all axes.
template <bool EXACT>
__device__ __forceinline__ bool slab(f3 bmin, f3 bmax, f3 o, f3 inv, float tmax, float &tmin_out) {
    float t10 = (bmin.x - o.x) * inv.x, t11 = (bmin.y - o.y) * inv.y, t12 = (bmin.z - o.z) * inv.z;
    float t20 = (bmax.x - o.x) * inv.x, t21 = (bmax.y - o.y) * inv.y, t22 = (bmax.z - o.z) * inv.z;
    float tmin = -1.0e38f;
    if (EXACT) {
#define MI_MIN(a, b) ((a) < (b) ? (a) : (b))
#define MI_MAX(a, b) ((a) < (b) ? (b) : (a))
        tmin = MI_MAX(tmin, MI_MIN(t10, t20));
        tmin = MI_MAX(tmin, MI_MIN(t11, t21));
        tmin = MI_MAX(tmin, MI_MIN(t12, t22));
        tmax = MI_MIN(tmax, MI_MAX(t10, t20));
        tmax = MI_MIN(tmax, MI_MAX(t11, t21));
        tmax = MI_MIN(tmax, MI_MAX(t12, t22));
#undef MI_MIN
#undef MI_MAX
    } else {
        tmin = fmaxf(fmaxf(fmaxf(tmin, fminf(t10, t20)), fminf(t11, t21)), fminf(t12, t22));
        tmax = fminf(fminf(fminf(tmax, fmaxf(t10, t20)), fmaxf(t11, t21)), fmaxf(t12, t22));
    }
    tmin_out = tmin;
    return tmin <= tmax && tmax > 0;
}

// Persistent waves, per-lane ray refill, and leaf tests postponed to their own phase.
//
// Every lane runs a small state machine: `cur` = next interior node (MI_DONE when the
// stack is empty), `pend0/pend1` = leaf primitives found but not yet tested.  In each
// iteration the WAVE executes one of two code blocks:
//   node phase — lanes without pending leaves fetch their node (64 B), slab-test both
//                child boxes, queue leaf children as pending, descend near-first / pop;
//   leaf phase — lanes with a pending leaf test one primitive (the only place the
//                triangle/sphere code exists).
// The leaf phase runs when at least `leaf_min` lanes wait for it (or nobody can do node
// work), so the expensive primitive test executes with many lanes active instead of
// once per node step for the few lanes that happen to have reached a leaf.  Postponing
// a leaf test never changes the result: the closest hit is order independent (tie rule
// above); only the culling distance h.t is momentarily stale.
//
// Work distribution and refill: see the comments in the kernel body.  Every wave leaves
// through the same exit (work exhausted and no lane active).
// queue == nullptr means the identity queue (slot = index).
// COUNT = true is the instrumented build used once per bench run (outside the timed
// region) to measure node visits and primitive tests per ray: work_stats[0] += node
// visits, [1] += primitive tests.
#define MI_DONE 0x7fffffff
#define MI_NONE (-1)
// PRIMARY = true is level 0: the queue is the identity over the batch's slots, rays
// come from primary_ray(), and the number of slots that are real pixels is added to
// *primary_count (it is the level-0 ray count).
template <bool COUNT, bool PRIMARY>
__global__ void __launch_bounds__(MI_BLOCK)
k_extend(DevScene sc, const uint32_t *queue, const uint32_t *count_ptr, uint32_t fixed_count, uint32_t *cursor,
         const float4 *ray_o, const float4 *ray_d, float4 *hits, int *ovf, unsigned long long *work_stats, int refill_min,
         uint32_t static_eighths, int leaf_min, ShardGeom g, uint32_t pix0, uint32_t npix, uint32_t *primary_count) {
    __shared__ int lds_stack[MI_LDS_STACK * MI_BLOCK];
    const uint32_t n = count_ptr ? *count_ptr : fixed_count;
    const int lane = threadIdx.x & 63;
    int *stack = lds_stack + threadIdx.x;
    const size_t ovf_stride = (size_t)gridDim.x * MI_BLOCK;
    int *my_ovf = ovf + (size_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    const bool empty_scene = sc.n_tri + sc.n_sph == 0;
    const char *node_base = (const char *)sc.nodes;
    // Work distribution in 64-ray chunks.  The first static_eighths/8 of the chunks are
    // dealt statically, interleaved over the waves (wave w takes chunks w, w+W, w+2W, ...:
    // no atomics, and every wave sees the same mix of image regions); the rest is pulled
    // dynamically, 4 chunks per returning atomic (a single word sustains only ~90
    // returning atomics/us), to even out the tail.
    const uint32_t n_waves = gridDim.x * (MI_BLOCK / 64);
    const uint32_t wave_id = blockIdx.x * (MI_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t n_chunks = (n + 63u) >> 6;
    uint32_t static_per_wave = (uint32_t)(((uint64_t)n_chunks * static_eighths) / (8u * n_waves));
    if (static_per_wave == 0) static_per_wave = 1;
    const uint32_t dyn_first = static_per_wave * n_waves;           // first dynamically dealt chunk
    uint32_t static_k = 0;                                          // wave-uniform
    uint32_t rng_next = 0, rng_end = 0;                             // wave-uniform, in rays
    bool exhausted = false;                                         // wave-uniform
    uint32_t n_nodes = 0, n_leaves = 0, ray_steps = 0, max_ray_steps = 0, wave_iters = 0;
    uint32_t n_primary = 0; // wave-uniform

    bool active = false, exact = false;
    uint32_t slot = 0;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), inv = mk3(0, 0, 0);
    Hit h{MI_NO_HIT, 0.0f, 0.0f, -1};
    int cur = MI_DONE, sp = 0, pend0 = MI_NONE, pend1 = MI_NONE;
    for (;;) {
        unsigned long long act = __ballot(active);
        int n_idle = 64 - __popcll(act);
        // Refill: whenever at least refill_min lanes are idle, hand them the next rays of
        // this wave's current chunk range, so rays of different traversal length do not
        // leave lanes idle for the rest of a 64-ray packet.
        if (!exhausted && (n_idle >= refill_min || act == 0ull)) {
            if (rng_next >= rng_end) {
                uint32_t c0 = n_chunks, c1 = n_chunks; // next chunk range [c0, c1)
                if (static_k < static_per_wave) {
                    c0 = wave_id + static_k * n_waves;
                    c1 = c0 + 1;
                    static_k++;
                } else if (dyn_first < n_chunks) {
                    // a relaxed load first: waves that arrive after the queue is drained do
                    // not add to the traffic on the cursor word
                    uint32_t seen = 0;
                    if (lane == 0) seen = __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = __builtin_amdgcn_readfirstlane(seen);
                    if (dyn_first + seen < n_chunks) {
                        uint32_t got = 0;
                        if (lane == 0) got = atomicAdd(cursor, 4u);
                        got = __builtin_amdgcn_readfirstlane(got);
                        c0 = dyn_first + got;
                        c1 = c0 + 4;
                    }
                }
                if (c0 >= n_chunks) {
                    if (static_k >= static_per_wave) exhausted = true;
                    rng_next = rng_end = 0;
                } else {
                    rng_next = c0 << 6;
                    rng_end = (c1 << 6) < n ? (c1 << 6) : n;
                }
            }
            if (rng_next < rng_end) {
                uint32_t rank = (uint32_t)__popcll(~act & ((1ull << lane) - 1ull));
                uint32_t idx = rng_next + rank;
                uint32_t avail = rng_end - rng_next;
                bool take = !active && rank < avail;
                if (PRIMARY) {
                    if (take) {
                        slot = idx;
                        int px, py;
                        uint32_t samp;
                        take = primary_ray(sc, g, pix0, npix, slot, o, d, px, py, samp);
                    }
                    n_primary += (uint32_t)__popcll(__ballot(take));
                } else if (take) {
                    slot = queue ? queue[idx] : idx;
                    float4 o4 = ray_o[slot], d4 = ray_d[slot];
                    o = xyz(o4);
                    d = xyz(d4);
                }
                if (take) {
                    inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); // ray.h:53-57,79-82
                    exact = !(__builtin_isfinite(inv.x) && __builtin_isfinite(inv.y) && __builtin_isfinite(inv.z));
                    h = Hit{MI_NO_HIT, 0.0f, 0.0f, -1};
                    sp = 0;
                    pend0 = pend1 = MI_NONE;
                    active = true;
                    if (empty_scene) {                       // bvh.cu:174-176,180-181
                        cur = MI_DONE;
                    } else if (sc.root_ref < 0) {            // bvh.cu:182-183
                        pend0 = ~sc.root_ref;
                        cur = MI_DONE;
                    } else {
                        cur = sc.root_ref;
                    }
                }
                rng_next += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
            }
            act = __ballot(active);
            if (act == 0ull) {
                if (exhausted) break;
                continue;
            }
        } else if (act == 0ull) {
            break; // exhausted and nothing in flight
        }

        if (COUNT) wave_iters++;
        const bool want_leaf = active && pend0 != MI_NONE;
        const bool want_node = active && pend0 == MI_NONE && cur != MI_DONE;
        const int nl = __popcll(__ballot(want_leaf)), nn = __popcll(__ballot(want_node));
        if (nl > 0 && (nl >= leaf_min || nn == 0)) {
            if (want_leaf) { // leaf phase: one primitive per lane
                if (COUNT) n_leaves++;
                prim_test(sc, pend0, o, d, h);
                pend0 = pend1;
                pend1 = MI_NONE;
            }
        } else if (want_node) { // node phase
            if (COUNT) { n_nodes++; ray_steps++; }
            const char *nd = node_base + ((uint32_t)cur << 6);
            float4 n0 = *(const float4 *)(nd), n1 = *(const float4 *)(nd + 16), n2 = *(const float4 *)(nd + 32);
            int2 refs = *(const int2 *)(nd + 48);
            float tminL, tminR;
            bool hitL, hitR;
            if (exact) {
                hitL = slab<true>(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), o, inv, h.t, tminL);
                hitR = slab<true>(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), o, inv, h.t, tminR);
            } else {
                hitL = slab<false>(mk3(n0.x, n0.y, n0.z), mk3(n0.w, n1.x, n1.y), o, inv, h.t, tminL);
                hitR = slab<false>(mk3(n1.z, n1.w, n2.x), mk3(n2.y, n2.z, n2.w), o, inv, h.t, tminR);
            }
            int l = refs.x, r = refs.y;
            if (hitL && l < 0) pend0 = ~l;
            if (hitR && r < 0) { if (pend0 == MI_NONE) pend0 = ~r; else pend1 = ~r; }
            bool goL = hitL && l >= 0, goR = hitR && r >= 0;
            if (goL && goR) {
                if (tminR < tminL) { int tmp = l; l = r; r = tmp; }
                if (sp < MI_LDS_STACK) stack[sp * MI_BLOCK] = r;
                else my_ovf[(size_t)(sp - MI_LDS_STACK) * ovf_stride] = r;
                sp++;
                cur = l;
            } else if (goL) {
                cur = l;
            } else if (goR) {
                cur = r;
            } else if (sp == 0) {
                cur = MI_DONE;
            } else {
                sp--;
                cur = sp < MI_LDS_STACK ? stack[sp * MI_BLOCK] : my_ovf[(size_t)(sp - MI_LDS_STACK) * ovf_stride];
            }
        }
        if (active && cur == MI_DONE && pend0 == MI_NONE) {
            hits[slot] = make_float4(h.t, __int_as_float(h.slot), h.u, h.v);
            active = false;
            if (COUNT) { max_ray_steps = max(max_ray_steps, ray_steps); ray_steps = 0; }
        }
    }
    if (PRIMARY && lane == 0 && n_primary) atomicAdd(primary_count, n_primary);
    if (COUNT) {
        atomicMax(&work_stats[2], (unsigned long long)max_ray_steps);
        if (lane == 0) { atomicAdd(&work_stats[3], (unsigned long long)wave_iters); atomicMax(&work_stats[4], (unsigned long long)wave_iters); }
        atomicAdd(&work_stats[0], (unsigned long long)n_nodes);
        atomicAdd(&work_stats[1], (unsigned long long)n_leaves);
    }
}

// -------------------------------------------------------------------- shade
struct DMat {
    float4 ke, kd, kr, kt;
    float index;
};
__device__ __forceinline__ DMat load_mat(const DevScene &sc, uint32_t id) {
    const float4 *m = sc.mats + 5 * (size_t)id;
    return DMat{m[0], m[1], m[2], m[3], m[4].x};
}

// Geometric normal and material at a hit (trimesh.cu:107-119, sphere.cu:23-32).
__device__ __forceinline__ void hit_surface(const DevScene &sc, int slot, float t, float u, float v, f3 o, f3 d, f3 &N, DMat &m) {
    if ((uint32_t)slot >= sc.n_tri) {
        int k = slot - (int)sc.n_tri;
        float4 cr = sc.sph[k];
        N = normalize3((o + (t * d)) - xyz(cr)); // ray.h:67, sphere.cu:26,31
        m = load_mat(sc, sc.sph_mat[k]);
    } else {
        N = xyz(sc.tri_normal[slot]);
        const uint32_t *mid = sc.tri_mat + 3 * (size_t)slot;
        if (sc.tri_has_vmat[slot]) { // trimesh.cu:110-116 with material.h:91-100,137-146
            float w = 1 - u - v;     // trimesh.cu:102
            DMat a = load_mat(sc, mid[0]), b = load_mat(sc, mid[1]), c = load_mat(sc, mid[2]);
            m.ke = add4(add4(mul4(a.ke, w), mul4(b.ke, u)), mul4(c.ke, v));
            m.kd = add4(add4(mul4(a.kd, w), mul4(b.kd, u)), mul4(c.kd, v));
            m.kr = add4(add4(mul4(a.kr, w), mul4(b.kr, u)), mul4(c.kr, v));
            m.kt = add4(add4(mul4(a.kt, w), mul4(b.kt, u)), mul4(c.kt, v));
            m.index = a.index * w + b.index * u + c.index * v;
        } else {
            m = load_mat(sc, mid[0]);
        }
    }
}

// CpuTracer.cu:37-53
__device__ __forceinline__ f3 hemisphere(f3 N, float u1, float u2) {
    f3 a = __builtin_fabsf(N.x) > MI_EPS ? normalize3(cross3(mk3(0, 1, 0), N)) : normalize3(cross3(mk3(1, 0, 0), N));
    f3 b = cross3(N, a);
    float p = 2 * 3.141592f * u1;
    float sin_p, cos_p;
    sincos_poly(p, sin_p, cos_p);
    float cos_t = u2 * u2; // std::pow(u, 2) in double then rounded to float == RN(u*u)
    float sin_t = __builtin_sqrtf(1.0f - cos_t * cos_t);
    return sin_t * cos_p * a + sin_t * sin_p * b + cos_t * N;
}

// The reference evaluates radiance by recursion: value_k = ke_k*32 + w_k (.) value_{k+1}
// (CpuTracer.cu:152,166,174,185).  To round exactly like it, every level stores
// (e_k, w_k) and the terminating thread folds them from the deepest level back.
__device__ __forceinline__ float4 fold_history(const float4 *hist_e, const float4 *hist_w, size_t cap, uint32_t slot, int levels, float4 val) {
    for (int j = levels - 1; j >= 0; j--) {
        float4 e = hist_e[(size_t)j * cap + slot], w = hist_w[(size_t)j * cap + slot];
        val = add4(e, mul4(w, val));
    }
    return val;
}
__device__ __forceinline__ float4 clamp01_4(float4 a) { // vec.h:184-186, CpuTracer.cu:115
    return make_float4(fminf(fmaxf(a.x, 0.0f), 1.0f), fminf(fmaxf(a.y, 0.0f), 1.0f), fminf(fmaxf(a.z, 0.0f), 1.0f),
                       fminf(fmaxf(a.w, 0.0f), 1.0f));
}

// One thread per queue entry of level `level`; depth_left = D - level.  PRIMARY = true
// is level 0: identity queue over the batch's n_fixed slots, ray from primary_ray(),
// throughput (1,1,1,1) (traceRay(r, 1, depth), CpuTracer.cu:114); slots that are not
// pixels get a zero path value.
template <bool PRIMARY>
__global__ void __launch_bounds__(MI_BLOCK)
k_shade(DevScene sc, ShardGeom g, uint32_t pix0, uint32_t npix, int level, int depth_left, uint64_t seed,
        const uint32_t *queue, const uint32_t *count_ptr, float4 *ray_o, float4 *ray_d, float4 *thr, const float4 *hits,
        float4 *hist_e, float4 *hist_w, size_t cap, float4 *path_out, uint32_t *next_queue, uint32_t *next_count, uint32_t n_fixed) {
    const uint32_t n = PRIMARY ? n_fixed : *count_ptr;
    const uint32_t stride = gridDim.x * MI_BLOCK;
    // every thread of a workgroup runs the same number of iterations (queue_append has barriers)
    const uint32_t n_round = (n + (MI_BLOCK - 1u)) & ~(MI_BLOCK - 1u);
    for (uint32_t i = blockIdx.x * MI_BLOCK + threadIdx.x; i < n_round; i += stride) {
        bool alive = false;
        uint32_t slot = 0;
        bool live = i < n;
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
        int px = 0, py = 0;
        uint32_t samp = 0;
        if (live) {
            slot = PRIMARY ? i : queue[i];
            if (PRIMARY) {
                live = primary_ray(sc, g, pix0, npix, slot, o, d, px, py, samp);
                if (!live) path_out[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        if (live) {
            float4 h4 = hits[slot];
            float t = h4.x;
            int pslot = __float_as_int(h4.y);
            float4 T = PRIMARY ? make_float4(1.0f, 1.0f, 1.0f, 1.0f) : thr[slot];
            if (pslot < 0) { // CpuTracer.cu:124-125: miss -> (1,1,1,1)
                path_out[slot] = clamp01_4(fold_history(hist_e, hist_w, cap, slot, level, make_float4(1.0f, 1.0f, 1.0f, 1.0f)));
            } else {
                if (!PRIMARY) {
                    o = xyz(ray_o[slot]);
                    d = xyz(ray_d[slot]);
                }
                f3 N;
                DMat m;
                hit_surface(sc, pslot, t, h4.z, h4.w, o, d, N, m);
                float4 e = mul4(m.ke, 32.0f); // CpuTracer.cu:131,185
                if (depth_left < 0) {         // CpuTracer.cu:130-131
                    path_out[slot] = clamp01_4(fold_history(hist_e, hist_w, cap, slot, level, e));
                } else {
                    f3 hitInner = o + (t + MI_EPS) * d; // :127
                    f3 hitOuter = o + (t - MI_EPS) * d; // :128
                    float reflect = m.kr.x, refract = m.kt.x; // :136-137
                    // path id = (j*W + i)*spp + sx*s + sy, the oracle's key
                    if (!PRIMARY) {
                        samp = slot / npix;
                        uint32_t lp = slot - samp * npix;
                        local_to_pixel(g, slot_to_local(g, pix0 + lp), px, py);
                    }
                    uint64_t path = ((uint64_t)py * g.width + px) * (uint64_t)(g.samples * g.samples) + (uint64_t)samp;
                    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)level, 0u};
                    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
                    float rnd = u01(ctr[0]); // :138
                    f3 refl = d - (2.0f * N) * dot3(N, d); // :140
                    f3 no, nd;
                    float4 w;
                    if (rnd < refract) { // :141
                        bool into = dot3(d, N) < 0; // :142
                        f3 nn = into ? N : -N;
                        float n1 = 1.0f, n2 = m.index;
                        float ratio = into ? n1 / n2 : n2 / n1;
                        float dt = dot3(d, nn);
                        float cos2t = 1 - ratio * ratio * (1 - dt * dt); // :148
                        if (cos2t < 0) { // :149-152 total internal reflection
                            no = hitOuter; nd = refl; w = m.kr;
                        } else { // :154-166
                            nd = normalize3(d * ratio - nn * (dt * ratio + __builtin_sqrtf(cos2t)));
                            no = hitInner;
                            float a = n2 - n1, b = n2 + n1;
                            float R0 = (a * a) / (b * b);
                            float cc = 1.0f - (into ? -dt : dot3(nd, -nn));
                            float Re = R0 + (1.0f - R0) * cc * cc * cc * cc * cc;
                            float ratio2 = ratio * ratio;
                            float Tr = (1.0f - Re) * ratio2;
                            w = make_float4(Tr, Tr, Tr, Tr);
                        }
                    } else { // :168-174 glossy/diffuse bounce
                        f3 diff = hemisphere(N, u01(ctr[1]), u01(ctr[2]));
                        nd = normalize3(reflect * refl + (1 - reflect) * diff);
                        no = hitOuter; w = m.kd;
                    }
                    float4 Tn = mul4(w, T); // `w * thresh`, :152,166,174
                    hist_e[(size_t)level * cap + slot] = e;
                    hist_w[(size_t)level * cap + slot] = w;
                    // entry test of the next level, CpuTracer.cu:121-122
                    if ((double)length2_4(Tn) < 3.0 * 12.0 / 255.0 / 255.0) {
                        path_out[slot] = clamp01_4(fold_history(hist_e, hist_w, cap, slot, level + 1, make_float4(0.0f, 0.0f, 0.0f, 0.0f)));
                    } else {
                        ray_o[slot] = make_float4(no.x, no.y, no.z, 0.0f);
                        ray_d[slot] = make_float4(nd.x, nd.y, nd.z, 0.0f);
                        thr[slot] = Tn;
                        alive = true;
                    }
                }
            }
        }
        queue_append(alive, slot, next_queue, next_count);
    }
}

// ------------------------------------------------------------------ resolve
__device__ __forceinline__ double hable(double x) { // tracer.cu:22-30
    double a = 0.15, b = 0.50, c = 0.10, d = 0.20, e = 0.02, f = 0.30;
    return ((x * (a * x + c * b) + d * e) / (x * (a * x + b) + d * f)) - e / f;
}
__device__ __forceinline__ uint8_t tonemap_byte(float c) { // tracer.cu:97
    return (uint8_t)(int)(255.0 * (1.0 / hable(2.0)) * hable(2.0 * (double)c));
}

// One thread per local pixel of the batch, in OUTPUT order (row-major in tile):
// mean over samples in the reference's order (sx outer, sy inner, CpuTracer.cu:101-108).
__global__ void __launch_bounds__(MI_BLOCK)
k_resolve(ShardGeom g, uint32_t pix0, uint32_t npix, const float4 *path_out, float4 *mean_out, uchar4 *rgba8_out) {
    uint32_t lp = blockIdx.x * MI_BLOCK + threadIdx.x;
    if (lp >= npix) return;
    int spp = g.samples * g.samples;
    uint32_t q = slot_to_local(g, pix0 + lp); // the local pixel this batch position rendered
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int s = 0; s < spp; s++) sum = add4(sum, path_out[(size_t)s * npix + lp]);
    float div = (float)g.samples * g.samples; // CpuTracer.cu:108
    float4 mean = make_float4(sum.x / div, sum.y / div, sum.z / div, sum.w / div);
    if (mean_out) mean_out[q] = mean;
    if (rgba8_out) {
        int px, py;
        bool inside = local_to_pixel(g, q, px, py);
        rgba8_out[q] = inside ? make_uchar4(tonemap_byte(mean.x), tonemap_byte(mean.y), tonemap_byte(mean.z), 255)
                              : make_uchar4(0, 0, 0, 0);
    }
}

// Probe output for mi_trace_primary / mi_trace_rays.
__global__ void __launch_bounds__(MI_BLOCK)
k_probe_out(DevScene sc, int64_t n, const float4 *ray_o, const float4 *ray_d, const float4 *hits, int32_t *prim_id, float *t_out, float *n_out) {
    int64_t i = (int64_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 h = hits[i];
    int slot = __float_as_int(h.y);
    prim_id[i] = slot < 0 ? -1 : sc.prim_id[slot];
    t_out[i] = h.x;
    if (n_out) {
        f3 N = mk3(0, 0, 0);
        if (slot >= 0) {
            DMat m;
            hit_surface(sc, slot, h.x, h.z, h.w, xyz(ray_o[i]), xyz(ray_d[i]), N, m);
        }
        n_out[3 * i] = N.x; n_out[3 * i + 1] = N.y; n_out[3 * i + 2] = N.z;
    }
}

// Gathered shard-local RGBA8 -> bottom-up RGB8 frame (tracer.cu:96 layout).
__global__ void __launch_bounds__(MI_BLOCK)
k_assemble(ShardGeom g, const uchar4 *gathered, int64_t shard_stride, int64_t shard_pixels, uint8_t *rgb8) {
    int64_t i = (int64_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    if (i >= shard_pixels * g.count) return;
    int r = (int)(i / shard_pixels);
    uint32_t q = (uint32_t)(i - (int64_t)r * shard_pixels);
    ShardGeom gr = g;
    gr.rank = r;
    int px, py;
    if (!local_to_pixel(gr, q, px, py)) return;
    uchar4 c = gathered[(size_t)r * shard_stride + q];
    size_t o = ((size_t)px + (size_t)py * g.width) * 3;
    rgb8[o] = c.x; rgb8[o + 1] = c.y; rgb8[o + 2] = c.z;
}
