// mi_kernels.h — hand-written HIP kernels of the wavefront path tracer (gfx950).
//
// Kernels (per bounce level k of a batch: k_tail | k_extend | k_shade; all paths of a batch are at level k):
//   k_extend   : BVH closest hit (any-hit variant for occlusion queries) of every ray of   [bvh.cu:3-10,178-207, bbox.cu:15-34,
//                the level's stream: persistent waves, per-lane state machine over a       trimesh.cu:87-121, sphere.cu:5-35]
//                quantised 4-wide tree, LDS short stack + LDS treelet, static chunk deal;
//                level 0 computes the camera rays itself (there is no generate kernel)     [camera.cu:6-16, CpuTracer.cu:99-109]
//   k_shade    : material branch, next ray, termination + history fold, wave-ballot        [CpuTracer.cu:119-186, :37-53]
//                compaction of the survivors into the next level's ray stream
//   k_tail     : once a level holds few rays, ONE launch carries every remaining path
//                through all its remaining levels (traverse -> shade -> traverse ... per lane)
//   k_resolve  : per-pixel mean in fixed sample order + Hable tonemap                       [CpuTracer.cu:108, tracer.cu:20-30,95-100]
//   k_assemble, k_unpermute_mean : shard-local outputs -> frame order
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
#define MI_MAX_DEPTH 64       // deepest (binary) tree the uploader accepts
#ifndef MI_WIDE8
#define MI_LDS_STACK 12       // traversal stack entries per lane held in LDS
#define MI_MAX_STACK 192      // up to three pushes per level of the 4-wide tree collapsed from it
#define MI_STACK_ELEM_BYTES 4
#endif
#define MI_NO_HIT 12345678.9f // ray.h:14
#define MI_EPS 0.00001f       // ray.h:12

struct DevScene {
    // The tree the kernels walk: 64-B quantised 4-wide nodes = px[4] py[4] pz[4] refs[4].  p?[c] packs
    // the two planes of seat c on that axis as 16-bit codes (low half: lower plane, high half: upper
    // plane).  A code q stands for the float F with bits 0x3F000000 | q << 8, a number in [0.5, 2)
    // (two binades, 15 mantissa bits), and for the plane qs[a] * F + qb[a].  Codes are rounded outward
    // (q_encode in mi_bvh_build.h), so a quantised box encloses the fp32 box it was made from.
    const uint4 *qnodes;
    float qs[3], qb[3];
    const float4 *tri;   // 48 B per triangle slot: v0.xyz, uInv.xyz, vInv.xyz, nInv.xyz (the w lanes are +0 and unused), + 16 B behind the last
    const float4 *tri_normal;
    const uint32_t *tri_mat; // 3 per triangle
    const uint8_t *tri_has_vmat;
    // What shading reads of a triangle in ONE 32-B record (k_pack_shade makes it from the three arrays above, which
    // stay the source for the device builder's permutation): (normal.xyz, hasMaterials), (material id a, b, c, 0)
    const uint4 *tri_shade;
    uint32_t n_mat;
    const float4 *sph; // (cx, cy, cz, r)
    const uint32_t *sph_mat;
    const float4 *mats; // 5 x float4 per material
    const int32_t *prim_id;
    int32_t root_ref;
    uint32_t n_tri, n_sph;
    float4 eye, u, v, look;
    // Cubemap background or nullptr (white): 6 x (width, height, byte offset) as int32, then
    // at byte 128 the RGB8 texels of the six faces (+x,-x,+y,-y,+z,-z; row 0 = bottom row).
    const int32_t *cube;
    // Treelet: nodes with index >= treelet_base are copies of the top of the tree (breadth
    // first from the root, at most MI_TREELET nodes) that every workgroup keeps in LDS.
    int32_t treelet_base, treelet_count;
};
#if !defined(MI_TREELET) && !defined(MI_WIDE8)
#define MI_TREELET 69        // nodes (4.4 KB; 138 measured no better): with the 12-KB stack, seven workgroups of k_extend fit the CU's 160 KB of LDS
#endif
#define MI_Q_PAD 0           // whole codes a quantised plane is moved outward beyond its rounding
#define MI_Q_MARGIN 3.814697265625e-06 // 2^-18: what a plane is moved outward (in F) before it is rounded outward
#define MI_Q_SEL_LO 0x0701000cu // v_perm_b32 selectors {0x3F, code byte 1, code byte 0, 0x00} of the low / high half
#define MI_Q_SEL_HI 0x0703020cu
#define MI_Q_SEL_FLIP 0x00020200u
#ifndef MI_TREELET_LEVELS
#define MI_TREELET_LEVELS 4  // root + up to four levels below it, breadth first
#endif

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
// vec.h:188-190 sums four lanes left to right; positions and directions carry w = 0, so the
// fourth product is a zero and adding it changes nothing but the sign of an exact zero sum
// (-0 + +0 = +0).  dot3 leaves it out; dot3w adds the +0 where that sign is observable: the
// plane distances of the triangle test, whose quotient is -0 / +0 / +-inf for a ray that
// starts on, or lies in, the triangle's plane.
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float dot3w(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z + 0.0f; }
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
        // one 64-bit product per multiplier (v_mad_u64_u32 yields both halves; __umulhi and * compile to two quarter-rate multiplies)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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
// Division by a launch-invariant divisor d >= 1 in four VALU instructions: q = (t + ((n - t) >> s1)) >> s2 with
// t = mulhi(m, n).  d a power of two: m = 0, s1 = 0, s2 = log2 d.  Otherwise the 33-bit magic number of
// Granlund & Montgomery in its add-and-shift form: m = floor(2^(33+L) / d) - 2^32 + 1, s1 = 1, s2 = L = floor(log2 d).
// (hipcc expands a 32-bit division by a run-time value into ~30 instructions; the slot -> pixel mapping below has
// seven of them and runs for every ray in k_shade: 40 % of that kernel's VALU instructions.)  The host builds
// and checks the constants (fastdiv_make in mi_tracer.hip).
struct FastDiv {
    uint32_t m, s1, s2;
};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &d) {
    const uint32_t t = __umulhi(d.m, n);
    return (t + ((n - t) >> d.s1)) >> d.s2;
}
struct ShardGeom {
    int32_t width, height, tile_w, tile_h, tiles_x, n_tiles, rank, count;
    int32_t samples; // per dimension
    FastDiv d_tpix, d_tiles_x, d_tile_w, d_sbw, d_samples; // by tile_w * tile_h, tiles_x, tile_w, tile_w / 8 (or 1), samples
    // Camera-plane coordinates of every (sample column, pixel column) and (sample row, pixel row), or nullptr:
    // cam_x[sx * width + px] = the x of camera.cu:11 for pixel column px and sample column sx, computed by k_camera_tables with
    // the operations of CpuTracer.cu:103-105 (so the bits are the reference's); likewise cam_y[sy * height + py].  With them a
    // primary ray costs two loads instead of four IEEE divisions, and — tiles being multiples of 8x8 — a 64-slot chunk's
    // pixel block is found once per wave with scalar arithmetic (chunk_pixels) instead of seven divisions per lane.
    const float *cam_x, *cam_y;
};
// Local pixel q (tile-major, row-major inside the tile; the order documented for
// mi_render_device) -> image pixel.  Returns false for pixels of edge tiles that
// fall outside the image.  32-bit arithmetic: the host rejects images whose padded
// pixel count does not fit (64-bit integer division is ~100 instructions on gfx950).
__device__ __forceinline__ bool local_to_pixel(const ShardGeom &g, uint32_t q, int &px, int &py) {
    uint32_t tpix = (uint32_t)(g.tile_w * g.tile_h);
    uint32_t j = fdiv(q, g.d_tpix);
    uint32_t off = q - j * tpix;
    uint32_t gt = (uint32_t)g.rank + j * (uint32_t)g.count;
    uint32_t ty = fdiv(gt, g.d_tiles_x), tx = gt - ty * (uint32_t)g.tiles_x;
    uint32_t oy = fdiv(off, g.d_tile_w), ox = off - oy * (uint32_t)g.tile_w;
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
    uint32_t j = fdiv(lp, g.d_tpix);
    uint32_t off = lp - j * tpix;
    uint32_t sub = off >> 6, in = off & 63u;
    uint32_t sbw = (uint32_t)g.tile_w >> 3;
    uint32_t sby = fdiv(sub, g.d_sbw), sbx = sub - sby * sbw;
    uint32_t x = sbx * 8u + (in & 7u), y = sby * 8u + (in >> 3);
    return j * tpix + y * (uint32_t)g.tile_w + x;
}

// Ray-queue compaction: wave ballot + popcount prefix inside each wave, the four wave
// totals combined through LDS, ONE global atomic per workgroup.  Returns the position
// reserved for this thread's ray in the next level's ray stream (meaningful only when
// `alive`).  Must be called by every thread of the workgroup (it contains barriers).
__device__ __forceinline__ uint32_t queue_reserve(bool alive, uint32_t *count) {
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
    const uint32_t pos = prefix + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads(); // wave_cnt / block_base are reused by the next call
    return pos;
}

// The rays of one level form a dense STREAM of records at positions [0, count): the
// payload travels with the queue instead of being looked up through a slot index, so
// every kernel reads and writes it with unit stride (at deeper levels the survivors are
// scattered over the slot range: one 16-B read per 128-B line otherwise).
//   o[i] = (origin, path slot as int bits)   d[i] = (direction, 0)   thr[i] = throughput
// Only the per-path history and the final path value stay indexed by slot.
//
// A stream is split into MI_SUB sub-streams, each with its own counter word:
// sub-stream k holds positions [k*segcap, k*segcap + cnt[k]).  A single counter word
// sustains only ~88 returning atomics/us on MI355X (one per 256 appended rays = 130k
// atomics for the 33M primary rays of the 1080p workload: the shade kernel was bound by
// them); the rate scales linearly with the number of words.  Workgroup b appends to
// sub-stream b % MI_SUB.  Consumers walk a VIRTUAL sequence of 64-ray chunks, the chunks
// of sub-stream 0 first, then 1, ...; stream_chunk() maps a virtual chunk to its rays.
#ifndef MI_SUB
#define MI_SUB 8
#endif
#define MI_CNT_STRIDE 32 // words between counters: every counter word has a 128-B line of its own
struct RayStream {
    float4 *o, *d, *thr;
    uint32_t segcap; // positions reserved per sub-stream
};
// Number of virtual chunks of a stream.  cnt == nullptr: one identity sub-stream of
// `fixed` rays (level 0, probes).  All arguments wave-uniform (scalar code).
__device__ __forceinline__ uint32_t stream_chunks(const uint32_t *cnt, uint32_t fixed) {
    if (!cnt) return (fixed + 63u) >> 6;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < MI_SUB; k++) n += (cnt[k * MI_CNT_STRIDE] + 63u) >> 6;
    return n;
}
__device__ __forceinline__ uint32_t stream_total(const uint32_t *cnt) {
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < MI_SUB; k++) n += cnt[k * MI_CNT_STRIDE];
    return n;
}
// Static deal of the virtual chunks over the W persistent waves of a launch: in round k (chunks
// [kW, (k+1)W)) wave w takes position (w + k*rot) mod W.  rot = 0 is the plain stride-W deal (wave w:
// w, w+W, ...), which is what every launch uses unless it ALIASES with the period of the slot order:
// slot = sample * npix + pixel, so with 4096 chunks per sample and W = 6144 every wave of a level-0
// launch saw only two 8x8 pixel patches, for all of its 171 chunks — waves on the object worked, waves
// on the sky idled (C4 level 0: VALU 30 % busy, 13.5 ms; 4.8 ms with a rotation).  The host picks rot
// (deal_rotation in mi_tracer.hip); a rotation costs the 1080p workload 4 % at level 0, hence only then.
struct ChunkDeal {
    uint32_t base, pos, rot, w; // all wave-uniform
};
__device__ __forceinline__ ChunkDeal deal_start(uint32_t wave_id, uint32_t n_waves, uint32_t rot) {
    return ChunkDeal{0u, wave_id, rot < n_waves ? rot : 0u, n_waves};
}
__device__ __forceinline__ uint32_t deal_chunk(const ChunkDeal &d) { return d.base + d.pos; }
__device__ __forceinline__ void deal_next(ChunkDeal &d) {
    d.base += d.w;
    d.pos += d.rot;
    if (d.pos >= d.w) d.pos -= d.w;
}
// Virtual chunk v (< stream_chunks) -> ray positions [first, end), end - first <= 64.
__device__ __forceinline__ void stream_chunk(const uint32_t *cnt, uint32_t fixed, uint32_t segcap, uint32_t v, uint32_t &first, uint32_t &end) {
    if (!cnt) {
        first = v << 6;
        end = first + 64u < fixed ? first + 64u : fixed;
        return;
    }
    uint32_t pre = 0;
    first = end = 0;
#pragma unroll
    for (int k = 0; k < MI_SUB; k++) {
        const uint32_t c = cnt[k * MI_CNT_STRIDE], nc = (c + 63u) >> 6;
        if (v >= pre && v < pre + nc) {
            const uint32_t off = (v - pre) << 6;
            first = (uint32_t)k * segcap + off;
            end = (uint32_t)k * segcap + (off + 64u < c ? off + 64u : c);
        }
        pre += nc;
    }
}

// ------------------------------------------------------------------ generate
// Camera ray of batch slot `slot` (= sample * npix + position in batch).  There is no
// generate kernel and no stored primary ray: level 0 of k_extend and of k_shade both
// call this (camera.cu:6-16 is ~30 instructions; a stored ray is 32 B written + 64 B
// read).  Returns false for slots of edge tiles that fall outside the image.
__device__ __forceinline__ bool primary_ray(const DevScene &sc, const ShardGeom &g, uint32_t pix0, uint32_t npix, const FastDiv &d_npix, uint32_t slot,
                                            f3 &o, f3 &dir, int &px, int &py, uint32_t &samp) {
    samp = fdiv(slot, d_npix);
    uint32_t lp = slot - samp * npix;
    if (!local_to_pixel(g, slot_to_local(g, pix0 + lp), px, py)) return false;
    int sx = (int)fdiv(samp, g.d_samples), sy = (int)samp - sx * g.samples;
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

// The fast form of the same for a whole 64-slot chunk (g.cam_x != nullptr: tiles are multiples of 8x8 and batches of 64
// pixels, so the chunk is one 8x8 pixel block of one sample).  `first` is wave-uniform and a multiple of 64: everything
// here is scalar code.
struct ChunkPix {
    uint32_t samp;
    int px0, py0;
    bool tile_ok;
};
__device__ __forceinline__ ChunkPix chunk_pixels(const ShardGeom &g, uint32_t pix0, uint32_t npix, const FastDiv &d_npix, uint32_t first) {
    ChunkPix c;
    c.samp = fdiv(first, d_npix);
    const uint32_t q0 = pix0 + (first - c.samp * npix);
    const uint32_t tpix = (uint32_t)(g.tile_w * g.tile_h);
    const uint32_t j = fdiv(q0, g.d_tpix), sub = (q0 - j * tpix) >> 6;
    const uint32_t sbw = (uint32_t)g.tile_w >> 3;
    const uint32_t sby = fdiv(sub, g.d_sbw), sbx = sub - sby * sbw;
    const uint32_t gt = (uint32_t)g.rank + j * (uint32_t)g.count;
    const uint32_t ty = fdiv(gt, g.d_tiles_x), tx = gt - ty * (uint32_t)g.tiles_x;
    c.px0 = (int)(tx * (uint32_t)g.tile_w + sbx * 8u);
    c.py0 = (int)(ty * (uint32_t)g.tile_h + sby * 8u);
    c.tile_ok = gt < (uint32_t)g.n_tiles;
    return c;
}
// Lane `in` (0..63) of the chunk: its pixel, and its camera ray from the tables.  False outside the image.
__device__ __forceinline__ bool chunk_ray(const DevScene &sc, const ShardGeom &g, const ChunkPix &c, uint32_t in, f3 &o, f3 &dir, int &px, int &py) {
    px = c.px0 + (int)(in & 7u);
    py = c.py0 + (int)(in >> 3);
    if (!(c.tile_ok && px < g.width && py < g.height)) return false;
    const uint32_t sx = fdiv(c.samp, g.d_samples), sy = c.samp - sx * (uint32_t)g.samples;
    const float x = g.cam_x[sx * (uint32_t)g.width + (uint32_t)px], y = g.cam_y[sy * (uint32_t)g.height + (uint32_t)py];
    dir = normalize3(xyz(sc.look) + x * xyz(sc.u) + y * xyz(sc.v)); // camera.cu:13
    o = xyz(sc.eye);
    return true;
}
// The tables: one thread per entry, the arithmetic of primary_ray.
__global__ void __launch_bounds__(MI_BLOCK)
k_camera_tables(int width, int height, int samples, float *cam_x, float *cam_y) {
    const uint32_t i = blockIdx.x * MI_BLOCK + threadIdx.x;
    const uint32_t nx = (uint32_t)samples * (uint32_t)width, ny = (uint32_t)samples * (uint32_t)height;
    if (i < nx) {
        const int sx = (int)(i / (uint32_t)width), px = (int)(i - (uint32_t)sx * (uint32_t)width);
        float xx = float(px) - 0.5f + (1 + 2 * sx) / (2.0f * samples); // CpuTracer.cu:103
        float x = xx / float(width);                                   // CpuTracer.cu:105
        x -= 0.5f;                                                     // camera.cu:11
        cam_x[i] = x;
    } else if (i < nx + ny) {
        const uint32_t k = i - nx;
        const int sy = (int)(k / (uint32_t)height), py = (int)(k - (uint32_t)sy * (uint32_t)height);
        float yy = float(py) - 0.5f + (1 + 2 * sy) / (2.0f * samples);
        float y = yy / float(height);
        y -= 0.5f;
        cam_y[k] = y;
    }
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
// Running closest hit of a ray: 8 B.  The barycentrics of trimesh.cu:102-103 are needed only where a mesh has
// per-vertex materials (trimesh.cu:110-116); hit_surface recomputes them there from (o, d, t) with the
// arithmetic of tri_test — the same bits — so they cost neither two registers in every traversal lane nor
// 16 of the 32 B a hit record used to take in the stream.
struct Hit {
    float t;
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

// trimesh.cu:87-121.  Evaluated without early exits: the three 16-B loads of the 48-B
// record are issued together (one memory round trip instead of two) and the reject
// tests of :92, :96 and :104 — each a pure function of the same values — are combined
// at the end.  The accepted (t, u, v) are the reference's.
// The record is three float4: (v0.xyz, uInv.x) (uInv.yz, vInv.xy) (vInv.z, nInv.xyz) — 48 B, three loads; the fourth
// lanes of the reference's vectors are +0 in every loadable scene and enter no result.
#define MI_TRI_BYTES 48u
__device__ __forceinline__ void tri_test_rec(const DevScene &sc, int slot, float4 ra, float4 rb, float4 rc, f3 o, f3 d, Hit &h) {
    const f3 v0 = mk3(ra.x, ra.y, ra.z), ui = mk3(ra.w, rb.x, rb.y), vi = mk3(rb.z, rb.w, rc.x), ni = mk3(rc.y, rc.z, rc.w);
    f3 p = o - v0;
    float pz = dot3w(ni, p);
    float dz = dot3w(ni, d);
    float t = -pz / dz;
    f3 hit = p + t * d;
    float u = dot3(ui, hit);
    float v = dot3(vi, hit);
    bool ok = ((pz < 0) != (dz < 0)) && !(t > h.t) && !(u < 0 || u > 1 || v < 0 || v > 1 || u + v > 1);
    if (ok && t == h.t) ok = tie_allows(sc, h.slot, slot, true);
    if (ok) { h.t = t; h.slot = slot; }
}
__device__ __forceinline__ void tri_test(const DevScene &sc, int slot, f3 o, f3 d, Hit &h) {
    const char *rec = (const char *)sc.tri + (size_t)(uint32_t)slot * MI_TRI_BYTES;
    float4 ra = *(const float4 *)(rec), rb = *(const float4 *)(rec + 16), rc = *(const float4 *)(rec + 32);
    // Keeps the three loads whole and together (one round trip).  Left alone, the compiler splits the 48 B by use — v0 and
    // nInv first, uInv and vInv behind the sign test, in pieces of 12, 12, 8 and 16 B — and waits twice: C3 +2.6 %, C4 +3 %
    // with the barrier.  (The same barrier in merged_step, whose fetch it splits likewise, measured no better.)
    asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w), "+v"(rb.x), "+v"(rb.y), "+v"(rb.z), "+v"(rb.w), "+v"(rc.x), "+v"(rc.y), "+v"(rc.z), "+v"(rc.w));
    tri_test_rec(sc, slot, ra, rb, rc, o, d, h);
}

__device__ __forceinline__ void sphere_test_rec(const DevScene &sc, int slot, float4 cr, f3 o, f3 d, Hit &h) { // sphere.cu:5-35
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
    h.t = t; h.slot = slot;
}
__device__ __forceinline__ void sphere_test(const DevScene &sc, int slot, f3 o, f3 d, Hit &h) { sphere_test_rec(sc, slot, sc.sph[slot - (int)sc.n_tri], o, d, h); }

__device__ __forceinline__ void prim_test(const DevScene &sc, int slot, f3 o, f3 d, Hit &h) { // geometry.h:26-29
    if ((uint32_t)slot >= sc.n_tri) sphere_test(sc, slot, o, d, h);
    else tri_test(sc, slot, o, d, h);
}

// The slab test (bbox.cu:15-34 in the reference: (b - p) * (1/d) per plane, `(a<b?a:b)` min/max)
// is a culling test: any test that accepts every box the closest primitive lies in leads to the
// same hit.  The kernels run slab_fast_q below on quantised, outward-rounded planes.

// Raw VALU min/max (no sNaN-canonicalising v_max x,x in front of every operand, which
// hipcc adds to fminf/fmaxf).  Only used where no operand can be NaN.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Plane code -> its float F (one v_perm_b32: byte 3 from the constant, bytes 2..1 the code, byte 0 zero).
__device__ __forceinline__ float q_code(uint32_t packed, uint32_t sel) {
    return __uint_as_float(__builtin_amdgcn_perm(0x3F000000u, packed, sel));
}
// Slab test on the quantised planes.  The ray carries, per axis, ia = qs/d and ic = (qb - o)/d, so the distance to the
// plane of code F is ONE FMA, F*ia + ic; and it carries the v_perm selector of its NEAR
// plane (the lower one when d > 0, the upper one when d < 0), so near and far planes are
// decoded directly and the per-axis min/max of bbox.cu:21-32 disappears: six v_perm take
// the place of six v_min/v_max, and a node is 64 B instead of 128.  The interval is the
// reference's up to rounding of a few ulp of the scene extent, far inside the outward
// rounding of the codes (steps of extent/98304 or /49152) and the 1e-5 x
// scene-diagonal padding of the boxes they were made from: a conservative culling test,
// so the closest hit it leads to is the one a test of every primitive finds (checked
// bit-for-bit against the brute-force oracle).
__device__ __forceinline__ bool slab_fast_q(uint32_t px, uint32_t py, uint32_t pz, const uint32_t seln[3], const uint32_t self[3], f3 ia,
                                            f3 ic, float tmax, float &tmin_out) {
    const float n0 = __builtin_fmaf(q_code(px, seln[0]), ia.x, ic.x), n1 = __builtin_fmaf(q_code(py, seln[1]), ia.y, ic.y),
                n2 = __builtin_fmaf(q_code(pz, seln[2]), ia.z, ic.z);
    const float f0 = __builtin_fmaf(q_code(px, self[0]), ia.x, ic.x), f1 = __builtin_fmaf(q_code(py, self[1]), ia.y, ic.y),
                f2 = __builtin_fmaf(q_code(pz, self[2]), ia.z, ic.z);
    const float tmin = vmax3(n0, n1, n2);
    const float tmx = vmin(tmax, vmin3(f0, f1, f2));
    tmin_out = tmin;
    return tmin <= tmx && tmx > 0;
}

// Set by k_tail when it takes over the remaining levels of a batch: 0 = wavefront
// kernels run, k > 0 = every kernel of level >= k returns at once.
__device__ __forceinline__ bool tail_took_over(const uint32_t *tail_level, int level) {
    uint32_t tl = *tail_level;
    return tl != 0 && (int)tl <= level;
}

#define MI_DONE 0x7fffffff
#define MI_NONE 0x7ffffffe // "pop next": no valid ref (node indices and leaf refs are below 2^26 in magnitude)
#ifdef MI_WIDE8
#include "mi_trav8.h"
#else
// Per-lane traversal state of the persistent kernels.
struct Lane {
    f3 o, d;
    float2 ax, ay, az; // fast slab test: (ia, ic) per axis, the distance to the plane of code F on that axis = F * ia + ic
    uint32_t seln[3]; // v_perm selector of the near plane on each axis
    Hit h;
    int cur, sp; // cur: >= 0 node to visit, < 0 leaf ~cur to test, MI_DONE finished
};
typedef int mi_stack_elem;
typedef int mi_ovf_elem;
#define MI_TREELET_U4 (MI_TREELET * 4)
__device__ __forceinline__ void lane_idle(Lane &L) {
    L.cur = MI_DONE; L.sp = 0;
    L.h = Hit{MI_NO_HIT, -1};
    L.o = L.d = mk3(0, 0, 0);
    L.ax = L.ay = L.az = make_float2(0.0f, 0.0f);
    L.seln[0] = L.seln[1] = L.seln[2] = MI_Q_SEL_LO;
}
__device__ __forceinline__ bool lane_wants_leaf(const Lane &L) { return L.cur < 0; }
__device__ __forceinline__ bool lane_wants_node(const Lane &L) { return L.cur >= 0 && L.cur != MI_DONE; }
__device__ __forceinline__ bool lane_done(const Lane &L) { return L.cur == MI_DONE; }
__device__ __forceinline__ void lane_finish(Lane &L) { L.cur = MI_DONE; L.sp = 0; } // any-hit: the query is answered
__device__ __forceinline__ void lane_start(const DevScene &sc, Lane &L, f3 o, f3 d, float t0 = MI_NO_HIT) {
    L.o = o;
    L.d = d;
    // 1/d (ray.h:53-57,79-82) feeds only the slab test, a culling test with margins (slab_fast_q), so
    // the hardware reciprocal (v_rcp_f32, 1 ulp) replaces three IEEE divisions: it scales every plane
    // distance of the ray by the same 1 +- 1e-7.  Clamped to +-1e18: an axis-parallel ray (d = +-0,
    // 1/d = +-inf) is culled like one that drifts by 1e-18 per unit of t — less than a billionth of
    // the box padding over any distance a hit can have — and every product below stays finite
    // (coordinates are < 1e15), so no inf - inf or 0 * inf can arise in the slab test.
    const f3 inv = mk3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e18f, 1e18f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e18f, 1e18f),
                       __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e18f, 1e18f));
    L.ax = make_float2(sc.qs[0] * inv.x, (sc.qb[0] - o.x) * inv.x);
    L.ay = make_float2(sc.qs[1] * inv.y, (sc.qb[1] - o.y) * inv.y);
    L.az = make_float2(sc.qs[2] * inv.z, (sc.qb[2] - o.z) * inv.z);
    L.seln[0] = inv.x < 0.0f ? MI_Q_SEL_HI : MI_Q_SEL_LO;
    L.seln[1] = inv.y < 0.0f ? MI_Q_SEL_HI : MI_Q_SEL_LO;
    L.seln[2] = inv.z < 0.0f ? MI_Q_SEL_HI : MI_Q_SEL_LO;
    L.h = Hit{t0, -1};
    L.sp = 0;
    // bvh.cu:174-176,180-183: nothing to hit / a single primitive / a tree
    L.cur = sc.n_tri + sc.n_sph == 0 ? MI_DONE : sc.root_ref;
}
typedef __attribute__((address_space(3))) int lds_int; // LDS-qualified, so stack accesses can only be ds_read/ds_write
typedef lds_int mi_lds_stack;
// Pop the next node / leaf (MI_DONE when the stack is empty).  The LDS read is
// unconditional (clamped index) and the rare overflow entry overrides it: a `?:` over the
// two address spaces compiles to one flat_load, which is slower than ds_read and makes
// every later wait cover both counters (hence also the LDS-qualified pointer type).
__device__ __forceinline__ void lane_pop(Lane &L, lds_int *stack, int *my_ovf, size_t ovf_stride) {
    if (L.sp == 0) {
        L.cur = MI_DONE;
    } else {
        L.sp--;
        int top = stack[(L.sp < MI_LDS_STACK ? L.sp : MI_LDS_STACK - 1) * MI_BLOCK];
        if (L.sp >= MI_LDS_STACK) top = my_ovf[(size_t)(L.sp - MI_LDS_STACK) * ovf_stride];
        L.cur = top;
    }
}
__device__ __forceinline__ void lane_push(Lane &L, lds_int *stack, int *my_ovf, size_t ovf_stride, int ref) {
    if (L.sp < MI_LDS_STACK) stack[L.sp * MI_BLOCK] = ref;
    else my_ovf[(size_t)(L.sp - MI_LDS_STACK) * ovf_stride] = ref;
    L.sp++;
}
// Node phase of one lane: fetch the quantised 4-wide node L.cur (64 B: px[4], py[4], pz[4], refs[4];
// DevScene::qnodes), slab-test the four child boxes, continue with the nearest child that is hit and
// push the others, farthest first.  A child ref >= 0 is a node, < 0 a leaf (~slot), MI_DONE an empty
// seat.  (bvh.cu:178-207 restructured; the 4-wide nodes are collapsed from the caller's binary tree
// and quantised at upload, mi_tracer.hip.)
__device__ __forceinline__ void cswap(bool sw, float &ka, float &kb, int &ra, int &rb) {
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}
// Where node i lives, as one generic address: the treelet copy in LDS for i >= treelet_base, global memory otherwise.
// NodeBase holds what does not depend on i: the global base, and for the treelet the 64-bit difference between the
// flat address of the LDS copy of node treelet_base and the global address that index would have — so the address
// is base + (i << 6) (+ delta in the treelet): a compare, two selects and two 64-bit adds instead of the dozen
// instructions of a select between two pointers.
struct NodeBase {
    uint64_t base, delta;
};
// every thread of the workgroup; the caller synchronises
__device__ __forceinline__ void treelet_load(const DevScene &sc, uint4 *lds) {
    for (int i = threadIdx.x; i < sc.treelet_count * 4; i += MI_BLOCK) lds[i] = sc.qnodes[(size_t)sc.treelet_base * 4 + i];
}
__device__ __forceinline__ NodeBase node_base(const DevScene &sc, const uint4 *treelet) {
    const uint64_t g = (uint64_t)(uintptr_t)sc.qnodes;
    const uint64_t l = (uint64_t)(uintptr_t)treelet; // generic (flat) address of the LDS array
    return NodeBase{g, l - g - ((uint64_t)(uint32_t)sc.treelet_base << 6)};
}
__device__ __forceinline__ const char *node_addr(const DevScene &sc, int cur, const NodeBase &nb) {
    return (const char *)(uintptr_t)(nb.base + ((uint64_t)(uint32_t)cur << 6) + (cur >= sc.treelet_base ? nb.delta : 0ull));
}
// Box tests, ordering and pushes of a node whose 64 B are in registers.  POP = false: when no child is hit the lane
// is left with L.cur = MI_NONE and the caller pops (merged_step, which pops once for leaf and node lanes).
template <bool POP>
__device__ __forceinline__ void node_process(const DevScene &sc, Lane &L, lds_int *stack, int *my_ovf, size_t ovf_stride, uint4 px, uint4 py, uint4 pz, int4 refs) {
    const uint32_t PX[4] = {px.x, px.y, px.z, px.w}, PY[4] = {py.x, py.y, py.z, py.w}, PZ[4] = {pz.x, pz.y, pz.z, pz.w};
    int R[4] = {refs.x, refs.y, refs.z, refs.w};
    float K[4]; // entry distance of a child that is hit, +inf otherwise
    const uint32_t self[3] = {L.seln[0] ^ MI_Q_SEL_FLIP, L.seln[1] ^ MI_Q_SEL_FLIP, L.seln[2] ^ MI_Q_SEL_FLIP};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float tmin;
        const bool hit = slab_fast_q(PX[c], PY[c], PZ[c], L.seln, self, mk3(L.ax.x, L.ay.x, L.az.x), mk3(L.ax.y, L.ay.y, L.az.y), L.h.t, tmin);
        K[c] = hit ? tmin : __builtin_inff(); // an empty seat's box is inverted on every axis: never hit
    }
#ifdef MI_DUMMY_VALU
    { uint32_t dz = (uint32_t)L.sp;
#pragma unroll
      for (int q = 0; q < MI_DUMMY_VALU; q++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(dz) : "v"(PX[q & 3]));
      asm volatile("" :: "v"(dz)); }
#endif
    // sorting network, descending by entry distance: K[0] farthest ... K[3] nearest
    cswap(K[0] < K[1], K[0], K[1], R[0], R[1]);
    cswap(K[2] < K[3], K[2], K[3], R[2], R[3]);
    cswap(K[0] < K[2], K[0], K[2], R[0], R[2]);
    cswap(K[1] < K[3], K[1], K[3], R[1], R[3]);
    cswap(K[1] < K[2], K[1], K[2], R[1], R[2]);
    // children that are not hit have K = +inf and sort to the front; push the hit ones of
    // K[0..2] (farthest first), continue with K[3] if it is hit, else pop
    const float inf = __builtin_inff();
    if (__builtin_amdgcn_ballot_w64(L.sp > MI_LDS_STACK - 3) == 0ull) {
        // Nobody in the wave can run out of LDS stack here: the three candidates are STORED
        // unconditionally — the slot above the top of a stack is free, so a ref that is not
        // pushed does no harm there — and only the stack pointer moves conditionally.  No
        // branch per push.  (If K[3] is not hit, nothing is: the pointer stays.)
        int sp = L.sp;
        stack[sp * MI_BLOCK] = R[0];
        sp += K[0] < inf ? 1 : 0;
        stack[sp * MI_BLOCK] = R[1];
        sp += K[1] < inf ? 1 : 0;
        stack[sp * MI_BLOCK] = R[2];
        sp += K[2] < inf ? 1 : 0;
        L.sp = sp;
        if (K[3] < inf) L.cur = R[3];
        else if (POP) lane_pop(L, stack, my_ovf, ovf_stride);
        else L.cur = MI_NONE;
    } else if (K[3] < inf) {
        if (K[0] < inf) lane_push(L, stack, my_ovf, ovf_stride, R[0]);
        if (K[1] < inf) lane_push(L, stack, my_ovf, ovf_stride, R[1]);
        if (K[2] < inf) lane_push(L, stack, my_ovf, ovf_stride, R[2]);
        L.cur = R[3];
    } else if (POP) {
        lane_pop(L, stack, my_ovf, ovf_stride);
    } else {
        L.cur = MI_NONE;
    }
}
__device__ __forceinline__ void node_phase(const DevScene &sc, Lane &L, lds_int *stack, int *my_ovf, size_t ovf_stride, const NodeBase &nb) {
    // one generic (flat) load per 16 B serves both: lanes at the top of the tree read the LDS
    // copy, the others global memory
    const char *nd = node_addr(sc, L.cur, nb);
    const uint4 px = *(const uint4 *)(nd), py = *(const uint4 *)(nd + 16), pz = *(const uint4 *)(nd + 32);
    const int4 refs = *(const int4 *)(nd + 48);
    node_process<true>(sc, L, stack, my_ovf, ovf_stride, px, py, pz, refs);
}
// Leaf lanes and node lanes in ONE step (draining waves and k_tail, whose time is a chain of latencies): every lane
// with work fetches what its ref names — a quantised node (64 B), a triangle record (48 B), or a sphere (16 B used; the
// sphere array is padded so that the 48-B read stays inside it) — with the same loads (a node four of 16 B, a primitive three), the wave waits once,
// both kinds compute, and one stack read serves every lane that needs its next ref.  (Two separate phases per
// iteration paid two memory round trips and two stack round trips.)  ANY: the first accepted primitive ends the query.
template <bool ANY>
__device__ __forceinline__ void merged_step(const DevScene &sc, Lane &L, lds_int *stack, int *my_ovf, size_t ovf_stride, const NodeBase &nb, bool want_leaf,
                                            bool want_node) {
    if (!(want_leaf || want_node)) return;
    const int slot = ~L.cur;
    const bool is_sph = want_leaf && (uint32_t)slot >= sc.n_tri;
    const char *at = want_node ? node_addr(sc, L.cur, nb)
                               : (is_sph ? (const char *)sc.sph + ((uint32_t)(slot - (int)sc.n_tri) << 4) : (const char *)sc.tri + (size_t)(uint32_t)slot * MI_TRI_BYTES);
    const uint4 r0 = *(const uint4 *)(at), r1 = *(const uint4 *)(at + 16), r2 = *(const uint4 *)(at + 32);
    uint4 r3 = make_uint4(0u, 0u, 0u, 0u);
    if (want_node) r3 = *(const uint4 *)(at + 48); // the refs: only a node has a fourth quarter
    auto f4 = [](uint4 r) { return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)); };
    if (want_leaf) {
        if (is_sph) sphere_test_rec(sc, slot, f4(r0), L.o, L.d, L.h);
        else tri_test_rec(sc, slot, f4(r0), f4(r1), f4(r2), L.o, L.d, L.h);
        L.cur = MI_NONE;
        if (ANY && L.h.slot >= 0) lane_finish(L);
    } else {
        node_process<false>(sc, L, stack, my_ovf, ovf_stride, r0, r1, r2, make_int4((int)r3.x, (int)r3.y, (int)r3.z, (int)r3.w));
    }
    if (L.cur == MI_NONE) lane_pop(L, stack, my_ovf, ovf_stride);
}
__device__ __forceinline__ void leaf_phase(const DevScene &sc, Lane &L, lds_int *stack, int *my_ovf, size_t ovf_stride) { // one primitive
    prim_test(sc, ~L.cur, L.o, L.d, L.h);
    lane_pop(L, stack, my_ovf, ovf_stride);
}

#endif // MI_WIDE8

// Persistent waves, per-lane ray refill, and leaf tests postponed to their own phase.
//
// Every lane runs a small state machine over ONE ref, `cur`: >= 0 a 4-wide node to visit, < 0 a leaf (~slot) to
// test, MI_DONE when its stack is empty.  Leaves travel through the same LDS stack as nodes.  In each iteration
// the WAVE executes one of two code blocks:
//   node phase — lanes whose ref is a node fetch it (64 B: four child boxes as 16-bit plane codes + four refs),
//                slab-test the four boxes, continue with the nearest child that is hit and push the others;
//   leaf phase — lanes whose ref is a leaf test that one primitive (the only place the triangle/sphere code
//                exists) and pop.
// The leaf phase runs when at least `leaf_min` lanes wait for it (or nobody can do node work), so the expensive
// primitive test executes with many lanes active instead of once per node step for the few lanes that happen to
// have reached a leaf.  A wave that has nothing left to refill with (draining) runs merged_step instead: both
// kinds in one step with one fetch.  Postponing a leaf test never changes the result: the closest hit is order
// independent (tie rule above); only the culling distance h.t is momentarily stale.
//
// Work distribution and refill: see the comments in the kernel body.  Every wave leaves
// through the same exit (work exhausted and no lane active).
// Ray i of the level's stream is (ray_o[i], ray_d[i]); its hit goes to hits[i].
// COUNT = true is the instrumented build used once per bench run (outside the timed
// region) to measure node visits and primitive tests per ray: work_stats[0] += node
// visits, [1] += primitive tests.
// PRIMARY = true is level 0: the stream is the identity over the batch's slots, rays
// come from primary_ray(), and the number of slots that are real pixels is added to
// *primary_count (it is the level-0 ray count).
// ANY = true is the any-hit (occlusion) query: ray_d[i].w is the ray's maximum distance,
// the lane stops at the first accepted primitive, and hits[i].y >= 0 means "occluded".
#ifndef MI_EXTEND_WG
#define MI_EXTEND_WG 6
#endif
template <bool COUNT, bool PRIMARY, bool ANY = false>
__global__ void __launch_bounds__(MI_BLOCK, PRIMARY ? 6 : MI_EXTEND_WG)
k_extend(DevScene sc, const uint32_t *count_ptr, uint32_t fixed_count, uint32_t segcap,
         const float4 *ray_o, const float4 *ray_d, float2 *hits, int *ovf, unsigned long long *work_stats, int refill_min,
         int leaf_min, ShardGeom g, uint32_t pix0, uint32_t npix, FastDiv d_npix, uint32_t *primary_count,
         const uint32_t *tail_level, int level, uint32_t deal_rot) {
    __shared__ mi_stack_elem lds_stack[MI_LDS_STACK * MI_BLOCK];
    if (!PRIMARY && tail_level && tail_took_over(tail_level, level)) return;
    __shared__ uint4 lds_treelet[MI_TREELET_U4];
    treelet_load(sc, lds_treelet);
    __syncthreads();
    const NodeBase nbase = node_base(sc, lds_treelet);

    const int lane = threadIdx.x & 63;
    mi_lds_stack *stack = (mi_lds_stack *)lds_stack + threadIdx.x;
    const size_t ovf_stride = (size_t)gridDim.x * MI_BLOCK;
    mi_ovf_elem *my_ovf = (mi_ovf_elem *)ovf + (size_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    // Work distribution in 64-ray chunks, dealt statically and interleaved over the waves
    // (wave w takes virtual chunks w, w+W, w+2W, ...): no atomics (a work cursor on one
    // word cost 6 of 7 ms at level 0), and every wave sees the same mix of image regions.
    // Wave-uniform bookkeeping is kept in SGPRs (readfirstlane).  The shape of the loops matters as much as what is
    // in them: three plain loops, each with one back edge and no `continue` — the steady loop (refill, then an inner
    // loop of phases that ends when the next refill is due) and the drain loop behind it.  As ONE loop with the drain
    // as a third branch and the refill test inside, the frame was 2 % slower; with a more tangled control flow the
    // compiler copies the whole per-lane state (≈45 v_mov) on every iteration.
    const uint32_t n_waves = gridDim.x * (MI_BLOCK / 64);
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (MI_BLOCK / 64) + (threadIdx.x >> 6));
    const uint32_t n_chunks = stream_chunks(count_ptr, fixed_count);
#ifdef MI_WAVE_TIMES
    const unsigned long long wt_start = wall_clock64();
    uint32_t wt_iters = 0;
#endif
    ChunkDeal deal = deal_start(wave_id, n_waves, deal_rot);
    uint32_t chunk = deal_chunk(deal);  // next virtual chunk of this wave (wave-uniform)
    uint32_t rng_next = 0, rng_end = 0; // wave-uniform, ray positions
    bool more = chunk < n_chunks;       // wave-uniform: rays may remain for this wave
    uint32_t n_nodes = 0, n_leaves = 0, ray_steps = 0, max_ray_steps = 0, wave_iters = 0;
    uint32_t n_primary = 0; // wave-uniform
    uint32_t dbg_cnt[7] = {0, 0, 0, 0, 0, 0, 0};

    bool active = false;
    uint32_t slot = 0;
    Lane L;
    lane_idle(L);
    unsigned long long act = 0ull; // ballot of `active`
    // Level 0: the camera rays of a chunk are computed by all 64 lanes at once when the wave
    // takes the chunk, and parked in LDS until the refills hand them out.  (Computed inside
    // the refill, the ~200 instructions of the pixel and camera arithmetic ran about three
    // times per chunk, each time for the third of the lanes that was idle: 38 % of the
    // kernel's VALU instructions.)
    __shared__ float lds_dir[PRIMARY ? 3 * MI_BLOCK : 1];
    unsigned long long chunk_valid = 0ull; // wave-uniform: slots of the chunk that are real pixels
    uint32_t chunk_first = 0;              // wave-uniform
    do {
        // Refill: whenever at least refill_min lanes are idle, hand them the next rays of
        // this wave's current chunk range, so rays of different traversal length do not
        // leave lanes idle for the rest of a 64-ray packet.
        const int n_idle = 64 - __popcll(act);
        if (more) { // (the loop around the phases below ends when a refill is due)
            if (rng_next >= rng_end && chunk < n_chunks) {
                stream_chunk(count_ptr, fixed_count, segcap, chunk, rng_next, rng_end);
                rng_next = __builtin_amdgcn_readfirstlane(rng_next);
                rng_end = __builtin_amdgcn_readfirstlane(rng_end);
                deal_next(deal);
                chunk = deal_chunk(deal);
                if (PRIMARY) {
                    chunk_first = rng_next;
                    f3 co = mk3(0, 0, 0), cd = mk3(0, 0, 0);
                    int px, py;
                    uint32_t samp;
                    bool real = rng_next + (uint32_t)lane < rng_end;
                    if (g.cam_x) real = real && chunk_ray(sc, g, chunk_pixels(g, pix0, npix, d_npix, rng_next), (uint32_t)lane, co, cd, px, py);
                    else real = real && primary_ray(sc, g, pix0, npix, d_npix, rng_next + (uint32_t)lane, co, cd, px, py, samp);
                    chunk_valid = __ballot(real);
                    n_primary += (uint32_t)__popcll(chunk_valid);
                    lds_dir[threadIdx.x] = cd.x;
                    lds_dir[MI_BLOCK + threadIdx.x] = cd.y;
                    lds_dir[2 * MI_BLOCK + threadIdx.x] = cd.z;
                    __builtin_amdgcn_wave_barrier(); // same wave reads below; LDS operations of a wave complete in order
                }
            }
            if (rng_next < rng_end) {
                const uint32_t rank = (uint32_t)__popcll(~act & ((1ull << lane) - 1ull));
                const uint32_t idx = rng_next + rank;
                const uint32_t avail = rng_end - rng_next;
                bool take = !active && rank < avail;
                f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
                float t0 = MI_NO_HIT;
                if (PRIMARY) {
                    const uint32_t src = (idx - chunk_first) & 63u;
                    take = take && ((chunk_valid >> src) & 1ull);
                    if (take) {
                        slot = idx;
                        const uint32_t at = (threadIdx.x & ~63u) + src;
                        o = xyz(sc.eye); // camera.cu:9
                        d = mk3(lds_dir[at], lds_dir[MI_BLOCK + at], lds_dir[2 * MI_BLOCK + at]);
                    }
                } else if (take) {
                    slot = idx;
                    o = xyz(ray_o[idx]);
                    const float4 d4 = ray_d[idx];
                    d = xyz(d4);
                    if (ANY) t0 = d4.w;
                }
                if (take) {
                    lane_start(sc, L, o, d, t0);
                    active = true;
                }
                rng_next += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
            }
            more = rng_next < rng_end || chunk < n_chunks;
        }
        if (!more) break; // nothing left to refill with: the drain loop below takes over
        do { // phases until at least refill_min lanes are idle: the refill code stays out of the hot loop (C5 -1.2 %, C3 -0.4 %)
        if (COUNT) wave_iters++;
#ifdef MI_WAVE_TIMES
        wt_iters++;
#endif
        const bool want_leaf = active && lane_wants_leaf(L);
        const bool want_node = active && lane_wants_node(L);
        const int nl = __popcll(__ballot(want_leaf)), nn = __popcll(__ballot(want_node));
        if (COUNT && lane == 0) { // lane slots of this iteration by state (instrumented variant only)
            const bool leaf_it = nl > 0 && (nl >= leaf_min || nn == 0);
            if (leaf_it) { dbg_cnt[0] += (uint32_t)(64 - nl - nn); dbg_cnt[1] += (uint32_t)nn; dbg_cnt[5] += (uint32_t)nl; }  // leaf phase: empty / waiting for node / working
            else { dbg_cnt[0] += (uint32_t)(64 - nl - nn); dbg_cnt[2] += (uint32_t)nl; dbg_cnt[6] += (uint32_t)nn; }              // node phase: empty / waiting for leaf / working
        }
        if (nl > 0 && (nl >= leaf_min || nn == 0)) {
            if (want_leaf) {
                if (COUNT) n_leaves++;
                leaf_phase(sc, L, stack, my_ovf, ovf_stride);
                if (ANY && L.h.slot >= 0) lane_finish(L); // first accepted hit ends the query
            }
        } else if (want_node) {
            if (COUNT) {
                n_nodes++;
                ray_steps++;
            }
            node_phase(sc, L, stack, my_ovf, ovf_stride, nbase);
        }
        if (active && lane_done(L)) {
            hits[slot] = make_float2(L.h.t, __int_as_float(L.h.slot));
            active = false;
            if (COUNT) { max_ray_steps = max(max_ray_steps, ray_steps); ray_steps = 0; }
        }
        act = __ballot(active);
        } while (act != 0ull && 64 - __popcll(act) < refill_min);
    } while (true);
    // Draining (no rays left to refill with): the wave's remaining time is a latency chain, so every lane advances
    // every iteration, leaves and nodes both (merged_step).  A loop of its own: with the drain as a third branch of
    // the loop above, C3 ran 1.4 % and C5 1.7 % slower (the steady loop carries no test of `more`, the drain loop no
    // refill; each has one back edge).
    act = __ballot(active); // (`act` predates the last refill)
    while (act != 0ull) {
        const bool want_leaf = active && lane_wants_leaf(L);
        const bool want_node = active && lane_wants_node(L);
#ifdef MI_WAVE_TIMES
        wt_iters++;
#endif
        if (COUNT) {
            wave_iters++;
            const int nl = __popcll(__ballot(want_leaf)), nn = __popcll(__ballot(want_node));
            if (lane == 0) { dbg_cnt[3] += (uint32_t)(64 - nl - nn); dbg_cnt[4] += (uint32_t)(nl + nn); }
            if (want_leaf) n_leaves++;
            if (want_node) { n_nodes++; ray_steps++; }
        }
        merged_step<ANY>(sc, L, stack, my_ovf, ovf_stride, nbase, want_leaf, want_node);
        if (active && lane_done(L)) {
            hits[slot] = make_float2(L.h.t, __int_as_float(L.h.slot));
            active = false;
            if (COUNT) { max_ray_steps = max(max_ray_steps, ray_steps); ray_steps = 0; }
        }
        act = __ballot(active);
    }
    if (PRIMARY && lane == 0 && n_primary) atomicAdd(primary_count, n_primary);
#ifdef MI_WAVE_TIMES
    // Diagnostic build (-DMI_WAVE_TIMES, MI_TRACER_WAVE_TIMES=<file>; scratch/wt.sh, scratch/wt.py): when each wave
    // started and left (100 MHz clock), where it ran (HW_ID, XCC_ID), how many iterations it made.
    if (!COUNT && work_stats && lane == 0 && level < 8 && wave_id < 8192) {
        unsigned long long *w = work_stats + ((size_t)level * 8192 + wave_id) * 4;
        w[0] = wt_start;
        w[1] = wall_clock64();
        w[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(63492);
        w[3] = wt_iters;
    }
#endif
    if (COUNT) {
        atomicMax(&work_stats[2], (unsigned long long)max_ray_steps);
        if (lane == 0) { atomicAdd(&work_stats[3], (unsigned long long)wave_iters); atomicMax(&work_stats[4], (unsigned long long)wave_iters); }
        if (lane == 0) for (int k = 0; k < 7; k++) atomicAdd(&work_stats[8 + k], (unsigned long long)dbg_cnt[k]);
        atomicAdd(&work_stats[0], (unsigned long long)n_nodes);
        atomicAdd(&work_stats[1], (unsigned long long)n_leaves);
    }
}

// -------------------------------------------------------------------- shade
struct DMat {
    float4 ke, kd, kr, kt;
    float index;
};
// The material table in LDS when it is small (it nearly always is: a handful of materials, 80 B each): the fetch
// of a hit's material is then no second dependent trip to global memory behind the triangle's record.
#define MI_LDS_MATS 48
typedef float mi_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) mi_v4f lds_float4; // a plain vector type: HIP's float4 class does not copy out of an address space
struct MatTable {
    const lds_float4 *lds; // the table's copy in LDS, used when `in_lds`
    bool in_lds;           // wave-uniform
};
__device__ __forceinline__ MatTable mat_table_load(const DevScene &sc, float4 *lds_mats) { // every thread of the workgroup; contains a barrier
    const bool in_lds = sc.n_mat <= MI_LDS_MATS;
    if (in_lds) for (uint32_t i = threadIdx.x; i < 5u * sc.n_mat; i += MI_BLOCK) lds_mats[i] = sc.mats[i];
    __syncthreads();
    return MatTable{(const lds_float4 *)lds_mats, in_lds};
}
__device__ __forceinline__ DMat load_mat(const DevScene &sc, const MatTable &mt, uint32_t id) {
    if (mt.in_lds) {
        const lds_float4 *m = mt.lds + 5 * id;
        const mi_v4f a = m[0], b = m[1], c = m[2], d = m[3], e = m[4];
        return DMat{make_float4(a.x, a.y, a.z, a.w), make_float4(b.x, b.y, b.z, b.w), make_float4(c.x, c.y, c.z, c.w), make_float4(d.x, d.y, d.z, d.w), e.x};
    }
    const float4 *m = sc.mats + 5 * (size_t)id;
    return DMat{m[0], m[1], m[2], m[3], m[4].x};
}

// Geometric normal and material at a hit (trimesh.cu:107-119, sphere.cu:23-32).
__device__ __forceinline__ void hit_surface(const DevScene &sc, const MatTable &mt, int slot, float t, f3 o, f3 d, f3 &N, DMat &m) {
    if ((uint32_t)slot >= sc.n_tri) {
        int k = slot - (int)sc.n_tri;
        float4 cr = sc.sph[k];
        N = normalize3((o + (t * d)) - xyz(cr)); // ray.h:67, sphere.cu:26,31
        m = load_mat(sc, mt, sc.sph_mat[k]);
    } else {
        const uint4 s0 = sc.tri_shade[2 * (size_t)slot], s1 = sc.tri_shade[2 * (size_t)slot + 1];
        N = mk3(__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z));
        const uint32_t mid[3] = {s1.x, s1.y, s1.z};
        if (s0.w) { // trimesh.cu:110-116 with material.h:91-100,137-146
            // u, v of trimesh.cu:98-101 again, from the accepted t: the operations of tri_test on the same inputs
            const char *rec = (const char *)sc.tri + (size_t)(uint32_t)slot * MI_TRI_BYTES;
            const float4 ra = *(const float4 *)(rec), rb = *(const float4 *)(rec + 16);
            const float vz = *(const float *)(rec + 32);
            const f3 hit = (o - mk3(ra.x, ra.y, ra.z)) + t * d;
            const float u = dot3(mk3(ra.w, rb.x, rb.y), hit), v = dot3(mk3(rb.z, rb.w, vz), hit);
            float w = 1 - u - v;     // trimesh.cu:102
            // (a*w + b*u) + c*v, accumulated one vertex material at a time (same
            // association, a third of the registers of loading all three first)
            const float wt[3] = {w, u, v};
            DMat a = load_mat(sc, mt, mid[0]);
            m.ke = mul4(a.ke, w); m.kd = mul4(a.kd, w); m.kr = mul4(a.kr, w); m.kt = mul4(a.kt, w);
            m.index = a.index * w;
#pragma unroll 1
            for (int k = 1; k < 3; k++) {
                DMat b = load_mat(sc, mt, mid[k]);
                m.ke = add4(m.ke, mul4(b.ke, wt[k])); m.kd = add4(m.kd, mul4(b.kd, wt[k]));
                m.kr = add4(m.kr, mul4(b.kr, wt[k])); m.kt = add4(m.kt, mul4(b.kt, wt[k]));
                m.index = m.index + b.index * wt[k];
            }
        } else {
            m = load_mat(sc, mt, mid[0]);
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

// Background of a ray that hits nothing when a cubemap is loaded: CubeMap::getColor
// (cubeMap.cu:6-31) + TextureMap::getMappedValue / getPixelAt (material.cu:20-37).  The
// reference's f4 algebra is reduced to what is not identically zero: `forward`, `up`,
// `right` are signed unit axes, so each dot product is +-one component of dir, exactly.
__device__ __forceinline__ float4 cube_pixel(const int32_t *cube, const uint8_t *texels, int w, int h, int x, int y) {
    x = max(0, min(w - 1, x));
    y = max(0, min(h - 1, y));
    const uint8_t *p = texels + ((size_t)x + (size_t)y * w) * 3;
    return make_float4(p[0] / 256.0f, p[1] / 256.0f, p[2] / 256.0f, 0.0f);
}
__device__ __forceinline__ float4 cubemap_color(const int32_t *cube, f3 raydir) {
    f3 dir = normalize3(raydir);                                     // :10
    int ax = 2;                                                      // :11-15
    if (__builtin_fabsf(dir.y) > __builtin_fabsf(dir.z)) ax = 1;
    if (__builtin_fabsf(dir.x) > __builtin_fabsf(dir.z) && __builtin_fabsf(dir.x) > __builtin_fabsf(dir.y)) ax = 0;
    const float major = ax == 0 ? dir.x : (ax == 1 ? dir.y : dir.z);
    const float sgn = major < 0 ? -1.0f : 1.0f;
    const float dn = sgn * major;                                    // dot(dir, forward), :20
    dir = mk3(dir.x / dn, dir.y / dn, dir.z / dn);
    const float m2 = ax == 0 ? dir.x : (ax == 1 ? dir.y : dir.z);
    int face = 2 * ax;                                               // :23-25
    if ((m2 < 0) != (ax == 2)) face++;
    // :27-29 with forward = sgn*e_ax:  ax 0: up = +y, right = sgn*z;  ax 1: up = q*z (q = sign of
    // dir[1] after the division = sgn), right = +x;  ax 2: up = +y, right = -sgn*x
    float du, dr;
    if (ax == 0) { du = dir.y; dr = sgn * dir.z; }
    else if (ax == 1) { du = (m2 < 0 ? -1.0f : 1.0f) * dir.z; dr = dir.x; }
    else { du = dir.y; dr = -sgn * dir.x; }
    const float u = 0.5f + dr / 2.0f, v = 0.5f + du / 2.0f;
    const int w = cube[3 * face], h = cube[3 * face + 1];
    const uint8_t *texels = (const uint8_t *)cube + 128 + (uint32_t)cube[3 * face + 2];
    const float x = u * (float)w, y = v * (float)h;                  // material.cu:21-29
    const int xi = (int)x, yi = (int)y;
    const float lx = x - (float)xi, ly = y - (float)yi;
    const float4 xl = add4(mul4(lx, cube_pixel(cube, texels, w, h, xi + 1, yi)), mul4(1 - lx, cube_pixel(cube, texels, w, h, xi, yi)));
    const float4 xr = add4(mul4(lx, cube_pixel(cube, texels, w, h, xi + 1, yi + 1)), mul4(1 - lx, cube_pixel(cube, texels, w, h, xi, yi + 1)));
    return add4(mul4(ly, xr), mul4(1 - ly, xl));
}

// The reference evaluates radiance by recursion: value_k = ke_k*32 + w_k (.) value_{k+1}
// (CpuTracer.cu:152,166,174,185).  To round exactly like it, every level stores
// (e_k, w_k) and the terminating thread folds them from the deepest level back.
// History of a path: w_k at hist[k*cap + slot], e_k at hist[e_off + k*cap + slot] (16 B each).  Most
// bounces are off surfaces that emit nothing: e_k is then +0 in all four lanes, is neither written nor
// read, and bit k of the path's `emask` (which travels with the ray, in the unused w lane of its
// direction record) stays clear; the fold still ADDS that zero, so the arithmetic is the reference's.
// Levels >= 31 have no mask bit: their e is always stored.  (One 32-B (e, w) record per level at first:
// the fold at the end of a Cornell-box path read seven of them, 64 B fetched for each.)
__device__ __forceinline__ float4 fold_history(const float4 *hist, size_t cap, size_t e_off, uint32_t slot, int levels, uint32_t emask, float4 val) {
    for (int j = levels - 1; j >= 0; j--) {
        const size_t at = (size_t)j * cap + slot;
        const float4 w = hist[at];
        float4 e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (j >= 31 || ((emask >> j) & 1u)) e = hist[e_off + at];
        val = add4(e, mul4(w, val));
    }
    return val;
}
__device__ __forceinline__ float4 clamp01_4(float4 a) { // vec.h:184-186, CpuTracer.cu:115
    return make_float4(fminf(fmaxf(a.x, 0.0f), 1.0f), fminf(fmaxf(a.y, 0.0f), 1.0f), fminf(fmaxf(a.z, 0.0f), 1.0f),
                       fminf(fmaxf(a.w, 0.0f), 1.0f));
}

// Global path id = (j*W + i)*spp + sx*s + sy: the RNG key the oracle uses.
__device__ __forceinline__ uint64_t path_id(const ShardGeom &g, int px, int py, uint32_t samp) {
    return ((uint64_t)py * g.width + px) * (uint64_t)(g.samples * g.samples) + (uint64_t)samp;
}

// Shading of one path at one level: CpuTracer.cu:124-185 for a ray (o,d) with
// throughput T whose closest hit is (t, pslot, u, v).  Either the path ends — its value
// is folded from the stored history and written to path_out — and false is returned,
// or (e, w) of this level are stored and the continuation ray (no, nd) with throughput
// Tn is returned with true.
__device__ __forceinline__ bool shade_path(const DevScene &sc, const MatTable &mt, uint64_t seed, uint64_t path, int level, int depth_left, uint32_t slot,
                                           f3 o, f3 d, float4 T, float t, int pslot, float4 *hist,
                                           size_t cap, size_t e_off, uint32_t &emask, float4 *path_out, f3 &no, f3 &nd, float4 &Tn) {
    if (pslot < 0) { // CpuTracer.cu:124-125: miss -> cubemap colour, or (1,1,1,1) without a cubemap
        const float4 bg = sc.cube ? cubemap_color(sc.cube, d) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        path_out[slot] = clamp01_4(fold_history(hist, cap, e_off, slot, level, emask, bg));
        return false;
    }
    f3 N;
    DMat m;
    hit_surface(sc, mt, pslot, t, o, d, N, m);
    float4 e = mul4(m.ke, 32.0f); // CpuTracer.cu:131,185
    if (depth_left < 0) {         // CpuTracer.cu:130-131
        path_out[slot] = clamp01_4(fold_history(hist, cap, e_off, slot, level, emask, e));
        return false;
    }
    f3 hitInner = o + (t + MI_EPS) * d; // :127
    f3 hitOuter = o + (t - MI_EPS) * d; // :128
    float reflect = m.kr.x, refract = m.kt.x; // :136-137
    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), (uint32_t)level, 0u};
    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    float rnd = u01(ctr[0]); // :138
    f3 refl = d - (2.0f * N) * dot3(N, d); // :140
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
    Tn = mul4(w, T); // `w * thresh`, :152,166,174
    const size_t at = (size_t)level * cap + slot;
    hist[at] = w;
    if (level >= 31 || ((__float_as_uint(e.x) | __float_as_uint(e.y) | __float_as_uint(e.z) | __float_as_uint(e.w)) != 0u)) {
        hist[e_off + at] = e;
        if (level < 31) emask |= 1u << level;
    }
    // entry test of the next level, CpuTracer.cu:121-122
    if ((double)length2_4(Tn) < 3.0 * 12.0 / 255.0 / 255.0) {
        path_out[slot] = clamp01_4(fold_history(hist, cap, e_off, slot, level + 1, emask, make_float4(0.0f, 0.0f, 0.0f, 0.0f)));
        return false;
    }
    return true;
}

// One thread per ray of level `level` (stream position i); depth_left = D - level.
// PRIMARY = true is level 0: the stream is the identity over the batch's n_fixed slots,
// the ray comes from primary_ray(), throughput (1,1,1,1) (traceRay(r, 1, depth),
// CpuTracer.cu:114); slots that are not pixels get a zero path value.  Continuation rays
// are appended to the next level's stream `out`.
#ifndef MI_SHADE_WG
#define MI_SHADE_WG 6 // workgroups per CU the compiler must leave room for (80 VGPRs): level-1 shade 0.52 -> 0.49 ms against no bound (85 VGPRs); 7 and 8 spill and are slower
#endif
template <bool PRIMARY>
__global__ void __launch_bounds__(MI_BLOCK, MI_SHADE_WG)
k_shade(DevScene sc, ShardGeom g, uint32_t pix0, uint32_t npix, FastDiv d_npix, int level, int depth_left, uint64_t seed, RayStream in,
        const uint32_t *count_ptr, const float2 *hits, float4 *hist, size_t cap, size_t e_off, float4 *path_out, RayStream out,
        uint32_t *next_count, uint32_t n_fixed, const uint32_t *tail_level) {
    if (!PRIMARY && tail_took_over(tail_level, level)) return;
    __shared__ float4 lds_mats[5 * MI_LDS_MATS];
    const MatTable mt = mat_table_load(sc, lds_mats);
    // i runs over the virtual positions (64 per virtual chunk) of the input stream
    const uint32_t n_virtual = stream_chunks(PRIMARY ? nullptr : count_ptr, n_fixed) << 6;
    const uint32_t stride = gridDim.x * MI_BLOCK;
    uint32_t *my_count = next_count + (blockIdx.x % MI_SUB) * MI_CNT_STRIDE;
    const uint32_t out_base = (blockIdx.x % MI_SUB) * out.segcap;
    // every thread of a workgroup runs the same number of iterations (queue_reserve has barriers)
    const uint32_t n_round = (n_virtual + (MI_BLOCK - 1u)) & ~(MI_BLOCK - 1u);
    // The stream records of the NEXT iteration are fetched before this iteration's are used
    // (software pipelining): the kernel is bound by memory latency x resident waves, and this
    // doubles what each wave has in flight on its longest (HBM) leg.
    struct In { bool live; uint32_t pos; float4 o4, d4, T; float2 h2; };
    auto fetch = [&](uint32_t i) {
        In r;
        r.live = false; r.pos = 0;
        r.o4 = r.d4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        r.h2 = make_float2(0.0f, 0.0f);
        r.T = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (i < n_virtual) {
            uint32_t first, end;
            stream_chunk(PRIMARY ? nullptr : count_ptr, n_fixed, in.segcap, __builtin_amdgcn_readfirstlane(i >> 6), first, end);
            r.pos = first + (i & 63u);
            r.live = r.pos < end;
        }
        if (r.live) {
            if (!PRIMARY) {
                r.o4 = in.o[r.pos];
                r.d4 = in.d[r.pos];
                // level 0 stores no throughput: T_1 = w_0 * (1,1,1,1) = w_0 bit for bit, which the history holds already
                r.T = level == 1 ? hist[(uint32_t)__float_as_int(r.o4.w)] : in.thr[r.pos];
            }
            r.h2 = hits[r.pos];
        }
        return r;
    };
    const uint32_t i_first = blockIdx.x * MI_BLOCK + threadIdx.x;
    In nxt = fetch(i_first);
    for (uint32_t i = i_first; i < n_round; i += stride) {
        const In cur = nxt;
        nxt = fetch(i + stride); // beyond n_round: nothing is loaded
        bool alive = false;
        uint32_t slot = 0;
        const uint32_t pos_in = cur.pos;
        bool live = cur.live;
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 0);
        float4 T = cur.T;
        int px = 0, py = 0;
        uint32_t samp = 0;
        if (live) {
            if (PRIMARY) {
                slot = pos_in;
                // (the camera tables of k_extend's level 0 are not used here: two more dependent loads per path cost this
                // latency-bound kernel what the saved divisions gain, measured 0.445 against 0.444 ms, 0.474 when prefetched)
                live = primary_ray(sc, g, pix0, npix, d_npix, slot, o, d, px, py, samp);
                if (!live) path_out[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                slot = (uint32_t)__float_as_int(cur.o4.w);
                o = xyz(cur.o4);
                d = xyz(cur.d4);
                samp = fdiv(slot, d_npix);
                uint32_t lp = slot - samp * npix;
                local_to_pixel(g, slot_to_local(g, pix0 + lp), px, py);
            }
        }
        f3 no = mk3(0, 0, 0), nd = mk3(0, 0, 0);
        float4 Tn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t emask = PRIMARY ? 0u : __float_as_uint(cur.d4.w); // which earlier levels stored an emission term
        if (live) {
            alive = shade_path(sc, mt, seed, path_id(g, px, py, samp), level, depth_left, slot, o, d, T, cur.h2.x, __float_as_int(cur.h2.y),
                               hist, cap, e_off, emask, path_out, no, nd, Tn);
        }
        const uint32_t pos = out_base + queue_reserve(alive, my_count);
        if (alive) {
            out.o[pos] = make_float4(no.x, no.y, no.z, __int_as_float((int)slot));
            out.d[pos] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(emask));
            if (!PRIMARY) out.thr[pos] = Tn; // level 1 reads w_0 from the history instead (above)
        }
    }
}

// Tail of a batch.  When only a few paths are still alive, one launch per level is
// bound by the longest single ray of each level (a few hundred dependent node fetches)
// while most of the chip idles.  k_tail is launched in front of every level k >= 1;
// it does nothing unless the queue of that level holds at most `threshold` rays, and
// then carries every remaining path through ALL its remaining levels in one launch:
// each lane loops traverse -> shade -> traverse ..., with the same node / leaf phases as
// k_extend plus a shade phase, and is refilled from the queue when its path ends.  It
// then publishes its level in *tail_level so the per-level kernels behind it return
// immediately.  Rays traced beyond the first level are added to *extra_rays.
__global__ void __launch_bounds__(MI_BLOCK, 5)
k_tail(DevScene sc, ShardGeom g, uint32_t pix0, uint32_t npix, FastDiv d_npix, int level0, int depth_total, uint64_t seed, RayStream in,
       const uint32_t *count_ptr, uint32_t threshold, uint32_t *tail_level, float4 *hist, size_t cap, size_t e_off, float4 *path_out, int *ovf,
       unsigned long long *extra_rays, int shade_min, int refill_min) {
    __shared__ mi_stack_elem lds_stack[MI_LDS_STACK * MI_BLOCK];
    const uint32_t n = stream_total(count_ptr);
    const uint32_t tl = *tail_level;
    // same decision in every wave: tl is only ever written with level0 by this launch
    if (n == 0 || n > threshold || (tl != 0 && tl != (uint32_t)level0)) return;
    __shared__ uint4 lds_treelet[MI_TREELET_U4];
    treelet_load(sc, lds_treelet);
    __syncthreads();
    const NodeBase nbase = node_base(sc, lds_treelet);
    __shared__ float4 lds_mats[5 * MI_LDS_MATS];
    const MatTable mt = mat_table_load(sc, lds_mats);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *tail_level = (uint32_t)level0;
    mi_lds_stack *stack = (mi_lds_stack *)lds_stack + threadIdx.x;
    const size_t ovf_stride = (size_t)gridDim.x * MI_BLOCK;
    mi_ovf_elem *my_ovf = (mi_ovf_elem *)ovf + (size_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    const uint32_t n_waves = gridDim.x * (MI_BLOCK / 64);
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (MI_BLOCK / 64) + (threadIdx.x >> 6));
    const uint32_t n_chunks = stream_chunks(count_ptr, 0u);
    ChunkDeal deal = deal_start(wave_id, n_waves, 0u);
    uint32_t chunk = deal_chunk(deal);  // static deal of the virtual chunks (wave-uniform)
    uint32_t rng_next = 0, rng_end = 0; // wave-uniform, in rays
    uint32_t n_extra = 0;               // wave-uniform

    bool active = false;
    uint32_t slot = 0, emask = 0;
    int level = level0;
    float4 T = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    Lane L;
    lane_idle(L);
    unsigned long long act = 0ull;
    bool more = true; // wave-uniform
#ifdef MI_TAIL_STATS
    // Diagnostic build (-DMI_TAIL_STATS, printed with MI_TRACER_DEBUG_LEVELS): is the tail bound by its sparsest waves or by
    // its longest path?  Per wave: iterations.  Per path (from the moment a lane takes it to its end): iterations that passed,
    // and iterations in which the path itself advanced (its own dependent chain).
    uint32_t ts_iters = 0, ts_path_t0 = 0, ts_own = 0, ts_lane_steps = 0, ts_max_elapsed = 0, ts_max_own = 0;
#endif
    do {              // single back edge, no `continue` (see k_extend)
        const int n_idle = 64 - __popcll(act);
        if (more && (n_idle >= refill_min || act == 0ull)) {
            if (rng_next >= rng_end && chunk < n_chunks) {
                stream_chunk(count_ptr, 0u, in.segcap, chunk, rng_next, rng_end);
                rng_next = __builtin_amdgcn_readfirstlane(rng_next);
                rng_end = __builtin_amdgcn_readfirstlane(rng_end);
                deal_next(deal);
                chunk = deal_chunk(deal);
            }
            if (rng_next < rng_end) {
                const uint32_t rank = (uint32_t)__popcll(~act & ((1ull << lane) - 1ull));
                const uint32_t avail = rng_end - rng_next;
                if (!active && rank < avail) {
                    const float4 o4 = in.o[rng_next + rank];
                    slot = (uint32_t)__float_as_int(o4.w);
                    T = level0 == 1 ? hist[slot] : in.thr[rng_next + rank]; // level 0 stores no throughput (k_shade)
                    level = level0;
                    const float4 d4 = in.d[rng_next + rank];
                    emask = __float_as_uint(d4.w);
                    lane_start(sc, L, xyz(o4), xyz(d4));
                    active = true;
#ifdef MI_TAIL_STATS
                    ts_path_t0 = ts_iters; ts_own = 0;
#endif
                }
                rng_next += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
            }
            more = rng_next < rng_end || chunk < n_chunks;
        }
        const bool want_leaf = active && lane_wants_leaf(L);
        const bool want_node = active && lane_wants_node(L);
        const bool want_shade = active && lane_done(L);
#ifdef MI_TAIL_STATS
        ts_iters++;
#endif
        const int nl = __popcll(__ballot(want_leaf)), nn = __popcll(__ballot(want_node)), ns = __popcll(__ballot(want_shade));
        if (ns > 0 && (ns >= shade_min || (nl == 0 && nn == 0))) {
            bool cont = false;
            if (want_shade) {
                uint32_t samp = fdiv(slot, d_npix);
                uint32_t lp = slot - samp * npix;
                int px, py;
                local_to_pixel(g, slot_to_local(g, pix0 + lp), px, py);
                f3 no, nd;
                float4 Tn;
                cont = shade_path(sc, mt, seed, path_id(g, px, py, samp), level, depth_total - level, slot, L.o, L.d, T, L.h.t, L.h.slot,
                                  hist, cap, e_off, emask, path_out, no, nd, Tn);
                if (cont) {
                    T = Tn;
                    level++;
                    lane_start(sc, L, no, nd);
                } else {
                    active = false;
                }
#ifdef MI_TAIL_STATS
                ts_own++; ts_lane_steps++;
                if (!cont) { ts_max_elapsed = max(ts_max_elapsed, ts_iters - ts_path_t0); ts_max_own = max(ts_max_own, ts_own); }
#endif
            }
            n_extra += (uint32_t)__popcll(__ballot(cont));
        } else {
#ifdef MI_TAIL_STATS
            if (want_leaf || want_node) { ts_own++; ts_lane_steps++; }
#endif
            // The tail is a latency chain (few rays, whole paths per lane): leaves and nodes
            // both advance in every iteration (1.19 -> 1.04 ms on C3).
            merged_step<false>(sc, L, stack, my_ovf, ovf_stride, nbase, want_leaf, want_node);
        }
        act = __ballot(active);
    } while (more || act != 0ull);
    if (lane == 0 && n_extra) atomicAdd(extra_rays, (unsigned long long)n_extra);
#ifdef MI_TAIL_STATS
    unsigned long long *ts = extra_rays + 41; // totals[48..]
    if (lane == 0) { atomicMax(&ts[0], (unsigned long long)ts_iters); atomicAdd(&ts[1], (unsigned long long)ts_iters); atomicAdd(&ts[5], 1ull); }
    atomicMax(&ts[2], (unsigned long long)ts_max_own);
    atomicMax(&ts[3], (unsigned long long)ts_max_elapsed);
    atomicAdd(&ts[4], (unsigned long long)ts_lane_steps);
#endif
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

// DevScene::tri_shade from the three per-triangle arrays shading used to gather separately.
__global__ void __launch_bounds__(MI_BLOCK)
k_pack_shade(uint32_t n_tri, const float4 *tri_normal, const uint32_t *tri_mat, const uint8_t *tri_has_vmat, uint4 *out) {
    const uint32_t i = blockIdx.x * MI_BLOCK + threadIdx.x;
    if (i >= n_tri) return;
    const float4 n = tri_normal[i];
    out[2 * (size_t)i] = make_uint4(__float_as_uint(n.x), __float_as_uint(n.y), __float_as_uint(n.z), tri_has_vmat[i] ? 1u : 0u);
    out[2 * (size_t)i + 1] = make_uint4(tri_mat[3 * (size_t)i], tri_mat[3 * (size_t)i + 1], tri_mat[3 * (size_t)i + 2], 0u);
}

// Probe output for mi_trace_primary / mi_trace_rays.
__global__ void __launch_bounds__(MI_BLOCK)
k_probe_out(DevScene sc, int64_t n, const float4 *ray_o, const float4 *ray_d, const float2 *hits, int32_t *prim_id, float *t_out, float *n_out) {
    int64_t i = (int64_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    if (i >= n) return;
    float2 h = hits[i];
    int slot = __float_as_int(h.y);
    prim_id[i] = slot < 0 ? -1 : sc.prim_id[slot];
    t_out[i] = h.x;
    if (n_out) {
        f3 N = mk3(0, 0, 0);
        if (slot >= 0) {
            DMat m;
            hit_surface(sc, MatTable{nullptr, false}, slot, h.x, xyz(ray_o[i]), xyz(ray_d[i]), N, m);
        }
        n_out[3 * i] = N.x; n_out[3 * i + 1] = N.y; n_out[3 * i + 2] = N.z;
    }
}

__global__ void __launch_bounds__(MI_BLOCK)
k_probe_occluded(int64_t n, const float2 *hits, uint8_t *out) {
    int64_t i = (int64_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    if (i < n) out[i] = __float_as_int(hits[i].y) >= 0 ? 1 : 0;
}

// Shard-local mean radiance -> frame order (i + j*W), for mi_render's host output.
__global__ void __launch_bounds__(MI_BLOCK)
k_unpermute_mean(ShardGeom g, int64_t local, const float4 *mean_local, float4 *mean_frame) {
    int64_t q = (int64_t)blockIdx.x * MI_BLOCK + threadIdx.x;
    if (q >= local) return;
    int px, py;
    if (!local_to_pixel(g, (uint32_t)q, px, py)) return;
    mean_frame[(size_t)px + (size_t)py * g.width] = mean_local[q];
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
