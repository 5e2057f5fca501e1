/*
 * oracle.cpp — CPU restatement of the reference's path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pathtracer_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (/root/reference) ships no golden vectors, tests
 * or scenes for this path, and it cannot be compiled in this image without
 * writing stand-ins for headers the image lacks (CUDA runtime headers, glm
 * 0.9.9.7 — cmake/glm.cmake:9-16 git-clones it), which the build rules forbid.
 * So this file is pinned only by (i) analytic known answers derived from the
 * cited formulas (tests/test_oracle_known_answers.py), (ii) published
 * known-answer vectors of Philox4x32-10, (iii) self-consistency (BVH order vs
 * brute force, libm vs polynomial sin/cos).  Every function cites the reference
 * file:line it restates.
 *
 * Semantics follow CpuTracer (src/CpuTracer.cu), NOT the CUDA kernel
 * (src/GpuTracer.cu:70-73 differs at the depth cut-off).
 *
 * Deliberate, documented deviations from the reference (all forced by
 * reproducibility; see DESIGN.md "Oracle"):
 *   D1. RNG: counter-based Philox4x32-10 keyed by (seed; path id, recursion
 *       level) instead of a thread-local std::mt19937 seeded from
 *       std::random_device (CpuTracer.cu:24-35), which is unseedable.
 *   D2. rsqrtf(x) (vec.h:201) is evaluated as 1.0f/sqrtf(x) (two correctly
 *       rounded IEEE ops).  The reference's host rsqrtf comes from CUDA headers
 *       that are not in this image.
 *   D3. sin/cos of CpuTracer.cu:48-49: mode 0 uses libm (as the reference does),
 *       mode 1 uses a fixed fmaf polynomial that the HIP kernels evaluate
 *       bit-identically.
 *   D4. Exact-t ties between primitives: the reference resolves them in the
 *       visiting order of ITS BVH (trimesh.cu:96 accepts t<=best, sphere.cu:22,28
 *       needs t<best).  "Canonical" mode resolves them as a visit in ascending
 *       primitive id would, which is BVH independent.
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../include/mi_tracer.h" /* data layout of the flattened scene only */

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

/* ---------------------------------------------------------------- vec.h:8-209 */
struct f4 {
    float v[4];
    f4() : v{0, 0, 0, 0} {}                                  /* vec.h:12 */
    f4(float s) : v{s, s, s, s} {}                           /* vec.h:15 broadcasts into w */
    f4(float x, float y, float z) : v{x, y, z, 0} {}         /* vec.h:18 */
    f4(float x, float y, float z, float w) : v{x, y, z, w} {}
    float &operator[](int i) { return v[i]; }
    const float &operator[](int i) const { return v[i]; }
};
static inline f4 ld4(const float *p) { return f4(p[0], p[1], p[2], p[3]); }
static inline f4 operator-(const f4 &a) { return f4(-a[0], -a[1], -a[2], -a[3]); }
static inline f4 operator+(const f4 &a, const f4 &b) { return f4(a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]); }
static inline f4 operator-(const f4 &a, const f4 &b) { return f4(a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]); }
static inline f4 operator*(const f4 &a, const f4 &b) { return f4(a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]); }
static inline f4 operator*(const f4 &a, float f) { return f4(a[0] * f, a[1] * f, a[2] * f, a[3] * f); }
static inline f4 operator*(float f, const f4 &a) { return f4(f * a[0], f * a[1], f * a[2], f * a[3]); }
static inline f4 operator/(const f4 &a, float f) { return f4(a[0] / f, a[1] / f, a[2] / f, a[3] / f); }
static inline f4 operator/(float f, const f4 &a) { return f4(f / a[0], f / a[1], f / a[2], f / a[3]); }
static inline f4 &operator+=(f4 &a, const f4 &b) { a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; return a; }

static inline float dot(const f4 &a, const f4 &b) {          /* vec.h:188-190, 4 lanes, left to right */
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}
static inline float length2(const f4 &a) { return dot(a, a); }          /* vec.h:196 */
static inline float length(const f4 &a) { return sqrtf(dot(a, a)); }    /* vec.h:192 */
static inline float rsqrt_ref(float x) { return 1.0f / sqrtf(x); }      /* deviation D2 */
static inline f4 normalize(const f4 &a) { return a * rsqrt_ref(dot(a, a)); } /* vec.h:200-202 */
static inline f4 cross(const f4 &a, const f4 &b) {           /* vec.h:204-208, w = 0 */
    return f4(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
static inline f4 clamp01(const f4 &a) {                      /* vec.h:184-186 with lo=0, hi=1 */
    return f4(fminf(fmaxf(a[0], 0.0f), 1.0f), fminf(fmaxf(a[1], 0.0f), 1.0f),
              fminf(fmaxf(a[2], 0.0f), 1.0f), fminf(fmaxf(a[3], 0.0f), 1.0f));
}

constexpr float RAY_EPSILON = 0.00001f;      /* ray.h:12 */
constexpr float ISECT_NO_HIT = 12345678.9f;  /* ray.h:14 */

/* ------------------------------------------------------------ ray.h:51-88 */
struct Ray {
    f4 pos, dir, invdir;
    Ray() {}
    Ray(const f4 &p, const f4 &d) : pos(p), dir(d), invdir(1.0f / d) {}  /* ray.h:53-57: all 4 lanes */
    void setDirection(const f4 &d) {                                      /* ray.h:79-82: invert() -> w = 0 */
        dir = d;
        invdir = f4(1.0f / d[0], 1.0f / d[1], 1.0f / d[2]);
    }
    f4 at(float t) const { return pos + (t * dir); }                      /* ray.h:67 */
};

/* material.h:80-146: five f4 */
struct Mat {
    f4 ke, kd, kr, kt, index;
};
static inline Mat mat_scale(float d, Mat m) {                /* material.h:137-146 */
    m.ke = m.ke * d; m.kd = m.kd * d; m.kr = m.kr * d; m.kt = m.kt * d; m.index = m.index * d;
    return m;
}
static inline void mat_add(Mat &a, const Mat &b) {           /* material.h:91-100 */
    a.ke += b.ke; a.kd += b.kd; a.kr += b.kr; a.kt += b.kt; a.index += b.index;
}

/* ray.h:16-49 (plus the primitive slot, which the reference's isect lacks) */
struct Isect {
    float t = ISECT_NO_HIT;
    f4 N = f4(0.0f);
    float u = 0, v = 0;
    Mat m;
    int slot = -1;
};

struct Counters {
    uint64_t node_visits = 0, leaf_tests = 0, rays = 0;
};

struct SceneView {
    const mi_scene_soa *s;
    Mat material(uint32_t id) const {
        const float *p = s->materials + (size_t)id * 20;
        Mat m; m.ke = ld4(p); m.kd = ld4(p + 4); m.kr = ld4(p + 8); m.kt = ld4(p + 12); m.index = ld4(p + 16);
        return m;
    }
    bool slot_is_sphere(int slot) const { return (uint32_t)slot >= s->n_tri; }
};

/* Deviation D4.  In canonical mode a candidate with t == best is accepted as an
 * ascending-id visit would accept it: a triangle replaces an equal-t earlier
 * primitive (trimesh.cu:96 `t > best` rejects only strictly greater), a sphere
 * never replaces an equal-t one (sphere.cu:22,28 need `<`). */
static inline bool tie_allows(const SceneView &sc, const Isect &cur, int cand_slot, bool cand_is_tri) {
    if (cur.slot < 0) return cand_is_tri; /* t == ISECT_NO_HIT exactly: tri accepts, sphere not */
    int cid = sc.s->prim_id[cand_slot], bid = sc.s->prim_id[cur.slot];
    bool best_is_sphere = sc.slot_is_sphere(cur.slot);
    if (cand_is_tri) return best_is_sphere ? true : (cid > bid);
    /* sphere candidate at equal t: wins only over an equal-t sphere of higher id
     * (it would have been visited first and kept).  Over a triangle: if the
     * sphere came first the triangle replaced it; if later it is rejected. */
    return best_is_sphere ? (cid < bid) : false;
}

/* ------------------------------------------------------- trimesh.cu:87-121 */
static bool tri_intersect(const SceneView &sc, int slot, const Ray &r, Isect &i, bool canonical) {
    const mi_scene_soa *s = sc.s;
    f4 v0 = ld4(s->tri_v0 + 4 * (size_t)slot), uInv = ld4(s->tri_uinv + 4 * (size_t)slot);
    f4 vInv = ld4(s->tri_vinv + 4 * (size_t)slot), nInv = ld4(s->tri_ninv + 4 * (size_t)slot);
    f4 p = r.pos - v0;                                /* :88 */
    f4 d = r.dir;                                     /* :89 */
    float p_z = dot(nInv, p);                         /* :90 */
    float d_z = dot(nInv, d);                         /* :91 */
    if ((p_z < 0) == (d_z < 0)) return false;         /* :92 */
    float t = -p_z / d_z;                             /* :95 */
    if (t > i.t) return false;                        /* :96 */
    if (canonical && t == i.t && !tie_allows(sc, i, slot, true)) return false; /* D4 */
    f4 hit = p + t * d;                               /* :99 */
    float u = dot(uInv, hit);                         /* :100 */
    float v = dot(vInv, hit);                         /* :101 */
    float w = 1 - u - v;                              /* :102 */
    if (u < 0 || u > 1 || v < 0 || v > 1 || u + v > 1) return false; /* :104 */
    i.t = t;                                          /* :107 */
    i.N = ld4(s->tri_normal + 4 * (size_t)slot);      /* :108 */
    i.u = u; i.v = v;                                 /* :109 */
    const uint32_t *mid = s->tri_mat + 3 * (size_t)slot;
    if (s->tri_has_vmat[slot]) {                      /* :110-116 */
        Mat m0 = mat_scale(w, sc.material(mid[0]));
        Mat m1 = mat_scale(u, sc.material(mid[1]));
        Mat m2 = mat_scale(v, sc.material(mid[2]));
        mat_add(m0, m1);
        mat_add(m0, m2);
        i.m = m0;
    } else {
        i.m = sc.material(mid[0]);                    /* :118 */
    }
    i.slot = slot;
    return true;
}

/* --------------------------------------------------------- sphere.cu:5-35 */
static bool sphere_intersect(const SceneView &sc, int slot, const Ray &r, Isect &i, bool canonical) {
    const mi_scene_soa *s = sc.s;
    size_t k = (size_t)slot - s->n_tri;
    const float *cr = s->sph_center_radius + 4 * k;
    f4 position(cr[0], cr[1], cr[2]);                 /* w = 0: built from glm::vec3, Parser.cu:339 */
    float radius = cr[3];
    f4 dir = normalize(r.dir);                        /* :7 */
    f4 toSphere = position - r.pos;                   /* :8 */
    float dt = dot(dir, toSphere);                    /* :9 */
    float discriminant = dt * dt - length2(toSphere) + radius * radius; /* :11 */
    if (discriminant < 0) return false;               /* :12 */
    discriminant = sqrtf(discriminant);               /* :15 */
    float t2 = dt + discriminant;                     /* :16 */
    if (t2 < RAY_EPSILON) return false;               /* :17 */
    float t1 = dt - discriminant;                     /* :20 */
    float t;
    if (t1 > RAY_EPSILON && (t1 < i.t || (canonical && t1 == i.t && tie_allows(sc, i, slot, false)))) {
        t = t1;                                       /* :22-27 */
    } else if (t2 < i.t || (canonical && t2 == i.t && tie_allows(sc, i, slot, false))) {
        t = t2;                                       /* :28-32 */
    } else {
        return false;
    }
    i.t = t;
    i.m = sc.material(s->sph_mat[k]);
    i.N = normalize(r.at(t) - position);              /* :26,31 — uses the UNnormalised dir */
    i.slot = slot;
    return true;
}

static inline bool prim_intersect(const SceneView &sc, int slot, const Ray &r, Isect &i, bool canonical, Counters &c) {
    c.leaf_tests++;                                   /* geometry.h:26-29 dispatch */
    return sc.slot_is_sphere(slot) ? sphere_intersect(sc, slot, r, i, canonical)
                                   : tri_intersect(sc, slot, r, i, canonical);
}

/* ---------------------------------------------------------- bbox.cu:15-34 */
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))        /* bbox.cu:4 — NaN -> second operand */
#define ORC_MAX(a, b) ((a) < (b) ? (b) : (a))        /* bbox.cu:5 */
static inline bool box_intersect(const float *bmin, const float *bmax, const Ray &r, float tMax) {
    const f4 &p = r.pos;
    const f4 &n = r.invdir;
    float tMin = -1.0e38f;                            /* :19 */
    float t10 = (bmin[0] - p[0]) * n[0];
    float t11 = (bmin[1] - p[1]) * n[1];
    float t12 = (bmin[2] - p[2]) * n[2];
    float t20 = (bmax[0] - p[0]) * n[0];
    float t21 = (bmax[1] - p[1]) * n[1];
    float t22 = (bmax[2] - p[2]) * n[2];
    tMin = ORC_MAX(tMin, ORC_MIN(t10, t20));          /* :27-32 */
    tMin = ORC_MAX(tMin, ORC_MIN(t11, t21));
    tMin = ORC_MAX(tMin, ORC_MIN(t12, t22));
    tMax = ORC_MIN(tMax, ORC_MAX(t10, t20));
    tMax = ORC_MIN(tMax, ORC_MAX(t11, t21));
    tMax = ORC_MIN(tMax, ORC_MAX(t12, t22));
    return tMin <= tMax && tMax > 0;                  /* :33 */
}

/* ------------------------------------------------------------ bvh.cu:3-10
 * Recursive, left subtree completely before the right box is even tested, both
 * sides always evaluated (non-short-circuit `|`). */
static bool node_intersect(const SceneView &sc, int32_t ref, const Ray &r, Isect &i, bool canonical, Counters &c) {
    if (ref < 0) return prim_intersect(sc, ~ref, r, i, canonical, c);     /* :4-5 */
    c.node_visits++;
    const float *nd = sc.s->nodes + 16 * (size_t)ref;
    int32_t l, rr;
    memcpy(&l, nd + 12, 4);
    memcpy(&rr, nd + 13, 4);
    bool a = box_intersect(nd + 0, nd + 3, r, i.t) && node_intersect(sc, l, r, i, canonical, c);  /* :7 */
    bool b = box_intersect(nd + 6, nd + 9, r, i.t) && node_intersect(sc, rr, r, i, canonical, c); /* :8 */
    return a | b;
}

/* scene.h:70-72 -> bvh.cu:174-176 (root == nullptr -> miss) */
static bool scene_intersect_bvh(const SceneView &sc, const Ray &r, Isect &i, bool canonical, Counters &c) {
    c.rays++;
    if (sc.s->n_tri + sc.s->n_sph == 0) return false;
    return node_intersect(sc, sc.s->root_ref, r, i, canonical, c);
}

/* BVH-independent closest hit: every primitive, ascending stable id, reference
 * accept rules.  This is what "canonical" BVH traversal must reproduce. */
static bool scene_intersect_brute(const SceneView &sc, const std::vector<int> &slots_by_id, const Ray &r, Isect &i, Counters &c) {
    c.rays++;
    bool any = false;
    for (int slot : slots_by_id) any |= prim_intersect(sc, slot, r, i, false, c);
    return any;
}

/* -------------------------------------------------------- camera.cu:6-16 */
static void ray_through(const mi_scene_soa *s, float x, float y, Ray &r) {
    x -= 0.5;                                         /* :11 (double literal; exact either way) */
    y -= 0.5;
    f4 look = ld4(s->cam_look), u = ld4(s->cam_u), v = ld4(s->cam_v);
    f4 dir = normalize(look + x * u + y * v);         /* :13 */
    r.pos = ld4(s->cam_eye);                          /* :14 */
    r.setDirection(dir);                              /* :15 */
}

/* ------------------------------------------------- RNG (deviation D1)
 * Philox4x32-10 (Salmon et al., SC'11; Random123 1.09).  counter =
 * (path_lo, path_hi, level, 0), key = (seed_lo, seed_hi). */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
/* uniform in [0,1) with 24 random bits (domain of CpuTracer.cu:24-35 random<float>(0,1)) */
static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* ------------------------------------------------- sin/cos (deviation D3) */
static inline void sincos_poly(float p, float *s_out, float *c_out) {
    float k = rintf(p * 0.636619772f);
    int q = (int)k;
    float r = fmaf(-k, 1.5707962513e+0f, p);
    r = fmaf(-k, 7.5497894159e-8f, r);
    float z = r * r;
    float sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fmaf(sp * z, r, r);
    float cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    switch (q & 3) {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

/* ---------------------------------------------------- CpuTracer.cu:37-53 */
static void get_basis(const f4 &normal, f4 &a, f4 &b) {
    a = std::fabs(normal[0]) > RAY_EPSILON ? normalize(cross(f4(0, 1, 0), normal))   /* :38-40 */
                                           : normalize(cross(f4(1, 0, 0), normal));
    b = cross(normal, a);                                                             /* :41 */
}
static f4 random_vec_from_hemisphere(const f4 &normal, float u1, float u2, int sincos_mode) {
    f4 a, b;
    get_basis(normal, a, b);
    float p = 2 * 3.141592f * u1;                     /* :47 */
    float cos_p, sin_p;
    if (sincos_mode == 0) { cos_p = std::cos(p); sin_p = std::sin(p); }   /* :48-49 */
    else sincos_poly(p, &sin_p, &cos_p);
    float cos_t = (float)std::pow((double)u2, 2);     /* :50 std::pow(float,int) promotes to double */
    float sin_t = std::sqrt(1.0f - cos_t * cos_t);    /* :51 */
    return sin_t * cos_p * a + sin_t * sin_p * b + cos_t * normal; /* :52 */
}

/* ------------------------------------------- cubeMap.cu:6-31, material.cu:20-37
 * Background of rays that miss everything when a cubemap is loaded (-c, CpuTracer.cu:125).
 * Faces in the order of TraceUI.cu:76-83: +x, -x, +y, -y, +z, -z; each is RGB8, row 0 =
 * bottom row of the image file (pngimage.cu:210-213, bitmap.cu).  Set with orc_set_cubemap
 * (test infrastructure: one process-wide cubemap, as the reference's TraceUI holds one). */
struct CubeFace {
    int width = 0, height = 0;
    std::vector<uint8_t> data;
};
static CubeFace g_cube[6];
static bool g_cube_set = false;

static f4 tex_pixel_at(const CubeFace &t, int x, int y) {   /* material.cu:32-37 */
    x = std::max(0, std::min(t.width - 1, x));
    y = std::max(0, std::min(t.height - 1, y));
    size_t idx = ((size_t)x + (size_t)y * t.width) * 3;
    return f4(t.data[idx] / 256.0f, t.data[idx + 1] / 256.0f, t.data[idx + 2] / 256.0f);
}
static f4 tex_mapped_value(const CubeFace &t, float u, float v) { /* material.cu:20-30 */
    float x = u * t.width;
    float y = v * t.height;
    float lerp_x = x - (int)x;
    float lerp_y = y - (int)y;
    f4 vlerp_xl = lerp_x * tex_pixel_at(t, (int)x + 1, (int)y) + (1 - lerp_x) * tex_pixel_at(t, (int)x, (int)y);
    f4 vlerp_xr = lerp_x * tex_pixel_at(t, (int)x + 1, (int)y + 1) + (1 - lerp_x) * tex_pixel_at(t, (int)x, (int)y + 1);
    return lerp_y * vlerp_xr + (1 - lerp_y) * vlerp_xl;
}
static f4 cubemap_color(const f4 &raydir, int *face_out = nullptr, float *uv_out = nullptr) { /* cubeMap.cu:6-31 */
    f4 dir = normalize(raydir);                                            /* :10 */
    int idx = 2;                                                           /* :11-15 */
    if (std::abs(dir[1]) > std::abs(dir[2])) idx = 1;
    if (std::abs(dir[0]) > std::abs(dir[2]) && std::abs(dir[0]) > std::abs(dir[1])) idx = 0;
    int sgn = dir[idx] < 0 ? -1 : 1;                                       /* int arithmetic: unselected axes are +0 */
    f4 forward((float)(sgn * (idx == 0)), (float)(sgn * (idx == 1)), (float)(sgn * (idx == 2))); /* :17-19 (glm::vec3 -> f4, w = 0) */
    float dn = dot(dir, forward);
    dir = dir / dn;                                                        /* :20 */
    idx *= 2;                                                              /* :23-25 */
    if ((dir[idx / 2] < 0) != (idx == 4)) idx++;
    f4 up = idx / 2 == 1 ? f4(0, 0, (dir[idx / 2] < 0 ? -1 : 1)) : f4(0, 1, 0); /* :27 */
    f4 right = normalize(cross(forward, up));                              /* :28 */
    float u = 0.5f + dot(dir, right) / 2.0f, v = 0.5f + dot(dir, up) / 2.0f; /* :29 */
    if (face_out) *face_out = idx;
    if (uv_out) { uv_out[0] = u; uv_out[1] = v; }
    return tex_mapped_value(g_cube[idx], u, v);                            /* :30 */
}

struct TraceCfg {
    SceneView sc;
    uint64_t seed;
    int sincos_mode;
    int isect_mode; /* 0 brute, 1 bvh reference order, 2 bvh canonical */
    const std::vector<int> *slots_by_id;
};

static inline bool scene_intersect(const TraceCfg &cfg, const Ray &r, Isect &i, Counters &c) {
    if (cfg.isect_mode == 0) return scene_intersect_brute(cfg.sc, *cfg.slots_by_id, r, i, c);
    return scene_intersect_bvh(cfg.sc, r, i, cfg.isect_mode == 2, c);
}

/* --------------------------------------------------- CpuTracer.cu:119-186 */
static f4 trace_ray(const TraceCfg &cfg, Ray &r, const f4 &thresh, int depth, uint64_t path, uint32_t level, Counters &c) {
    Isect i;
    if ((double)length2(thresh) < 3.0 * 12.0 / 255.0 / 255.0)             /* :121 */
        return f4(0.0f);
    if (!scene_intersect(cfg, r, i, c))                                    /* :124 */
        return g_cube_set ? cubemap_color(r.dir) : f4(1.0f);               /* :125 */
    f4 hitInner = r.pos + (i.t + RAY_EPSILON) * r.dir;                     /* :127 */
    f4 hitOuter = r.pos + (i.t - RAY_EPSILON) * r.dir;                     /* :128 */
    if (depth < 0)                                                         /* :130 */
        return i.m.ke * 32.0f;                                             /* :131 */
    f4 rad;                                                                /* :133 */
    float reflect = i.m.kr[0];                                             /* :136 */
    float refract = i.m.kt[0];                                             /* :137 */
    uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), level, 0};
    philox4x32_10(ctr, (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
    float rnd = u01(ctr[0]);                                               /* :138 */
    f4 refl = r.dir - 2.0f * i.N * dot(i.N, r.dir);                        /* :140 */
    if (rnd < refract) {                                                   /* :141 */
        bool into = dot(r.dir, i.N) < 0;                                   /* :142 */
        f4 n = into ? i.N : -i.N;                                          /* :143 */
        float n1 = 1.0f;
        float n2 = i.m.index[0];                                           /* :145 */
        float ratio = into ? n1 / n2 : n2 / n1;                            /* :146 */
        float dt = dot(r.dir, n);                                          /* :147 */
        float cos2t = 1 - ratio * ratio * (1 - dt * dt);                   /* :148 */
        if (cos2t < 0) {                                                   /* :149 */
            Ray rr(hitOuter, refl);                                        /* :150 */
            f4 w = i.m.kr / 1.0f;                                          /* :151 */
            rad += w * trace_ray(cfg, rr, w * thresh, depth - 1, path, level + 1, c); /* :152 */
        } else {
            f4 dir = normalize(r.dir * ratio - n * (dt * ratio + std::sqrt(cos2t))); /* :154 */
            Ray rr(hitInner, dir);                                         /* :155 */
            float a = n2 - n1;
            float b = n2 + n1;
            float R0 = (a * a) / (b * b);                                  /* :158 */
            float cc = 1.0f - (into ? -dt : dot(rr.dir, -n));              /* :159 */
            float Re = R0 + (1.0f - R0) * cc * cc * cc * cc * cc;          /* :160 */
            float ratio2 = ratio * ratio;
            float Tr = (1.0f - Re) * ratio2;                               /* :162 */
            rad = Tr * trace_ray(cfg, rr, Tr * thresh, depth - 1, path, level + 1, c); /* :166 */
        }
    } else {
        f4 diff = random_vec_from_hemisphere(i.N, u01(ctr[1]), u01(ctr[2]), cfg.sincos_mode); /* :169 */
        f4 dir = normalize(reflect * refl + (1 - reflect) * diff);         /* :170 */
        Ray rr(hitOuter, dir);                                             /* :171 */
        f4 w = i.m.kd / 1.0f;                                              /* :173 */
        rad = w * (trace_ray(cfg, rr, w * thresh, depth - 1, path, level + 1, c)); /* :174 */
    }
    return i.m.ke * 32.0f + rad;                                           /* :185 */
}

/* CpuTracer.cu:111-117 */
static f4 trace(const TraceCfg &cfg, float x, float y, int depth, uint64_t path, Counters &c) {
    Ray r;
    ray_through(cfg.sc.s, x, y, r);
    f4 ret = trace_ray(cfg, r, f4(1.0f), depth, path, 0, c);
    return clamp01(ret);
}

/* CpuTracer.cu:99-109 */
static f4 trace_pixel(const TraceCfg &cfg, int i, int j, int W, int H, int samples, int depth, Counters &c) {
    f4 sum;
    for (int x = 0; x < samples; x++) {
        for (int y = 0; y < samples; y++) {
            float xx = float(i) - 0.5f + (1 + 2 * x) / (2.0f * samples);
            float yy = float(j) - 0.5f + (1 + 2 * y) / (2.0f * samples);
            uint64_t path = ((uint64_t)j * W + i) * (uint64_t)(samples * samples) + (uint64_t)(x * samples + y);
            sum += trace(cfg, xx / float(W), yy / float(H), depth, path, c);
        }
    }
    return sum / ((float)samples * samples);
}

/* tracer.cu:20-30 */
static double hable(double x) {
    double a = 0.15, b = 0.50, c = 0.10, d = 0.20, e = 0.02, f = 0.30;
    return ((x * (a * x + c * b) + d * e) / (x * (a * x + b) + d * f)) - e / f;
}
/* tracer.cu:95-100 */
static inline unsigned char tonemap_byte(float c) {
    return (unsigned char)(int)(255.0 * (1.0 / hable(2.0)) * hable(2.0 * c));
}

static std::vector<int> slots_by_id(const mi_scene_soa *s) {
    size_t n = (size_t)s->n_tri + s->n_sph;
    std::vector<int> order(n, -1);
    for (size_t slot = 0; slot < n; slot++) {
        int id = s->prim_id[slot];
        if (id >= 0 && (size_t)id < n) order[id] = (int)slot;
    }
    return order;
}

} // namespace

/* ============================================================ C interface */

struct orc_counters {
    uint64_t rays, node_visits, leaf_tests;
    double ms;
};

ORC_API int orc_trace_primary(const mi_scene_soa *s, int W, int H, int samples, int isect_mode,
                              int32_t *prim_id_out, float *t_out, float *n_out, orc_counters *cnt) {
    if (!s || W <= 0 || H <= 0 || samples <= 0) return 1;
    std::vector<int> order = slots_by_id(s);
    TraceCfg cfg{SceneView{s}, 0, 0, isect_mode, &order};
    Counters c;
    auto t0 = std::chrono::steady_clock::now();
    size_t spp = (size_t)samples * samples;
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++)
            for (int x = 0; x < samples; x++)
                for (int y = 0; y < samples; y++) {
                    float xx = float(i) - 0.5f + (1 + 2 * x) / (2.0f * samples);
                    float yy = float(j) - 0.5f + (1 + 2 * y) / (2.0f * samples);
                    Ray r;
                    ray_through(s, xx / float(W), yy / float(H), r);
                    Isect is;
                    bool hit = scene_intersect(cfg, r, is, c);
                    size_t k = ((size_t)j * W + i) * spp + (size_t)(x * samples + y);
                    if (prim_id_out) prim_id_out[k] = hit ? s->prim_id[is.slot] : -1;
                    if (t_out) t_out[k] = is.t;
                    if (n_out) { n_out[3 * k] = is.N[0]; n_out[3 * k + 1] = is.N[1]; n_out[3 * k + 2] = is.N[2]; }
                }
    if (cnt) {
        cnt->rays = c.rays; cnt->node_visits = c.node_visits; cnt->leaf_tests = c.leaf_tests;
        cnt->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

ORC_API int orc_trace_rays(const mi_scene_soa *s, int64_t n, const float *rays, int isect_mode,
                           int32_t *prim_id_out, float *t_out, float *n_out, orc_counters *cnt) {
    if (!s || n < 0 || !rays) return 1;
    std::vector<int> order = slots_by_id(s);
    TraceCfg cfg{SceneView{s}, 0, 0, isect_mode, &order};
    Counters c;
    for (int64_t k = 0; k < n; k++) {
        const float *q = rays + 6 * k;
        Ray r(f4(q[0], q[1], q[2]), f4(q[3], q[4], q[5]));   /* bounce rays use the 2-arg ctor, CpuTracer.cu:150,155,171 */
        Isect is;
        bool hit = scene_intersect(cfg, r, is, c);
        if (prim_id_out) prim_id_out[k] = hit ? s->prim_id[is.slot] : -1;
        if (t_out) t_out[k] = is.t;
        if (n_out) { n_out[3 * k] = is.N[0]; n_out[3 * k + 1] = is.N[1]; n_out[3 * k + 2] = is.N[2]; }
    }
    if (cnt) { cnt->rays = c.rays; cnt->node_visits = c.node_visits; cnt->leaf_tests = c.leaf_tests; cnt->ms = 0; }
    return 0;
}

/* Cubemap of the following orc_render calls: faces +x,-x,+y,-y,+z,-z, RGB8, row 0 = bottom
 * row; data == NULL clears it (white background, CpuTracer.cu:125). */
ORC_API int orc_set_cubemap(const uint8_t *const *data, const int *width, const int *height) {
    g_cube_set = false;
    if (!data) return 0;
    for (int f = 0; f < 6; f++) {
        if (!data[f] || width[f] <= 0 || height[f] <= 0) return 1;
        g_cube[f].width = width[f];
        g_cube[f].height = height[f];
        g_cube[f].data.assign(data[f], data[f] + (size_t)width[f] * height[f] * 3);
    }
    g_cube_set = true;
    return 0;
}
/* cubeMap.cu:6-31 for one direction: rgba[4], face index, (u, v). */
ORC_API int orc_cubemap_color(const float *dir3, float *rgba, int *face, float *uv) {
    if (!g_cube_set) return 1;
    f4 c = cubemap_color(f4(dir3[0], dir3[1], dir3[2]), face, uv);
    for (int k = 0; k < 4; k++) rgba[k] = c[k];
    return 0;
}

/* Any-hit: does ANY primitive's intersect() accept the ray when the running closest
 * distance starts at tmax (trimesh.cu:92-96 accepts t <= best, sphere.cu:22-28 t < best)?
 * No counterpart in the reference (it casts no shadow rays); checker for
 * mi_occluded_rays.  Every primitive is tested with a fresh isect. */
ORC_API int orc_occluded(const mi_scene_soa *s, int64_t n, const float *rays, const float *tmax, uint8_t *out) {
    if (!s || n < 0 || !rays || !tmax || !out) return 1;
    SceneView sc{s};
    Counters c;
    const int64_t n_prim = (int64_t)s->n_tri + s->n_sph;
    for (int64_t k = 0; k < n; k++) {
        const float *q = rays + 6 * k;
        Ray r(f4(q[0], q[1], q[2]), f4(q[3], q[4], q[5]));
        uint8_t occ = 0;
        for (int64_t slot = 0; slot < n_prim && !occ; slot++) {
            Isect is;
            is.t = tmax[k] < ISECT_NO_HIT ? tmax[k] : ISECT_NO_HIT; /* clamped to ray.h:14, as mi_occluded_rays documents */
            if (prim_intersect(sc, (int)slot, r, is, false, c)) occ = 1;
        }
        out[k] = occ;
    }
    return 0;
}

/* Full render with CpuTracer's scheduling shape (CpuTracer.cu:57-97: LIFO list of
 * 32x32 tiles behind a mutex, `threads` workers; :188-191 join).  Only tiles with
 * (row-major tile index) % shard_count == shard_rank are rendered; other pixels
 * are left untouched.  y0/y1 optionally restrict to rows [y0,y1) (bounded samples
 * for the CPU baseline); pass 0,height for everything. */
ORC_API int orc_render(const mi_scene_soa *s, const mi_render_params *p, int sincos_mode, int isect_mode,
                       int threads, int y0, int y1, float *rgba_mean_out, uint8_t *rgb8_out, orc_counters *cnt) {
    if (!s || !p || p->width <= 0 || p->height <= 0 || p->samples_per_dim <= 0) return 1;
    const int W = p->width, H = p->height;
    const int tw = p->tile_w > 0 ? p->tile_w : 32, th = p->tile_h > 0 ? p->tile_h : 32;   /* :55 granularity */
    const int sc_n = p->shard_count > 1 ? p->shard_count : 1, sc_r = p->shard_count > 1 ? p->shard_rank : 0;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();                  /* main.cu:5 */
    if (threads <= 0) threads = 1;
    std::vector<int> order = slots_by_id(s);
    TraceCfg cfg{SceneView{s}, p->seed, sincos_mode, isect_mode, &order};

    struct Work { int minX, maxX, minY, maxY; };
    std::vector<Work> work;                                                                /* :58-69 */
    const int tiles_x = (W + tw - 1) / tw;
    for (int y = 0; y < H; y += th)
        for (int x = 0; x < W; x += tw) {
            int tid = (y / th) * tiles_x + (x / tw);
            if (tid % sc_n != sc_r) continue;
            Work w{x, std::min(x + tw, W), std::max(y, y0), std::min(std::min(y + th, H), y1)};
            if (w.minY < w.maxY) work.push_back(w);
        }
    std::mutex lock;
    std::vector<Counters> per_thread(threads);
    auto t0 = std::chrono::steady_clock::now();
    auto body = [&](int tidx) {
        Counters &c = per_thread[tidx];
        for (;;) {
            Work w;
            {
                std::lock_guard<std::mutex> g(lock);                                       /* :75-83 */
                if (work.empty()) break;
                w = work.back();                                                           /* LIFO */
                work.pop_back();
            }
            for (int y = w.minY; y < w.maxY; y++)
                for (int x = w.minX; x < w.maxX; x++) {                                    /* :84-86 */
                    f4 col = trace_pixel(cfg, x, y, W, H, p->samples_per_dim, p->depth, c);
                    size_t k = (size_t)x + (size_t)y * W;
                    if (rgba_mean_out) memcpy(rgba_mean_out + 4 * k, col.v, 16);
                    if (rgb8_out) {
                        rgb8_out[3 * k + 0] = tonemap_byte(col[0]);
                        rgb8_out[3 * k + 1] = tonemap_byte(col[1]);
                        rgb8_out[3 * k + 2] = tonemap_byte(col[2]);
                    }
                }
        }
    };
    if (threads == 1) body(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(body, t);
        for (auto &t : pool) t.join();
    }
    if (cnt) {
        cnt->rays = cnt->node_visits = cnt->leaf_tests = 0;
        for (auto &c : per_thread) { cnt->rays += c.rays; cnt->node_visits += c.node_visits; cnt->leaf_tests += c.leaf_tests; }
        cnt->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

/* trimesh.cu:7-47.  verts: a,b,c as 3 float4.  Returns 0 and leaves outputs
 * untouched for a degenerate face (:18-21), else 1.  The 3x3 inverse restates
 * glm 0.9.9.7 compute_inverse<3,3> (cofactors times one reciprocal determinant;
 * glm is a third-party dependency pinned by cmake/glm.cmake:11,15, not vendored). */
ORC_API int orc_tri_precompute(const float *verts, float *normal, float *uinv, float *vinv, float *ninv) {
    f4 a = ld4(verts), b = ld4(verts + 4), c = ld4(verts + 8);
    f4 vab = b - a, vac = c - a, vcb = b - c;                               /* :13-15 */
    if (length2(vab) == 0.0f || length2(vac) == 0.0f || length2(vcb) == 0.0f) return 0; /* :18-21 */
    f4 n = normalize(cross(vab, vac));                                      /* :24 */
    /* column-major m[col][row]: columns vab, vac, normal (:27-30) */
    float m[3][3] = {{vab[0], vab[1], vab[2]}, {vac[0], vac[1], vac[2]}, {n[0], n[1], n[2]}};
    float ood = 1.0f / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
                        - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2])
                        + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
    float inv[3][3];
    inv[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
    inv[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
    inv[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
    inv[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
    inv[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
    inv[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
    inv[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
    inv[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
    inv[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
    for (int k = 0; k < 4; k++) normal[k] = n[k];                           /* :36 */
    uinv[0] = inv[0][0]; uinv[1] = inv[1][0]; uinv[2] = inv[2][0]; uinv[3] = 0; /* :37 */
    vinv[0] = inv[0][1]; vinv[1] = inv[1][1]; vinv[2] = inv[2][1]; vinv[3] = 0; /* :38 */
    ninv[0] = inv[0][2]; ninv[1] = inv[1][2]; ninv[2] = inv[2][2]; ninv[3] = 0; /* :39 */
    return 1;
}

/* camera.cu:35-65 with viewdir/updir (Parser.cu:116-119), setFOV (:47-52),
 * setAspectRatio (:54-58).  Outputs u, v, look as float4. */
ORC_API void orc_camera_from_look(const float *viewdir, const float *updir, float fov_deg, float aspect,
                                  float *u, float *v, float *look) {
    f4 z = -f4(viewdir[0], viewdir[1], viewdir[2]);                         /* :37 */
    f4 y(updir[0], updir[1], updir[2]);                                     /* :38 */
    f4 x = cross(y, z);                                                     /* :39 */
    float fov = fov_deg;
    fov /= (180.0f / 3.141592f);                                            /* :49 */
    float nh = 2 * std::tan(fov / 2);                                       /* :50 */
    /* m = columns x,y,z (:40-43); m*(1,0,0) = x etc. (glm mat3*vec3 =
     * col0*v.x + col1*v.y + col2*v.z) */
    auto mul = [&](float a, float b, float c) {
        return f4(x[0] * a + y[0] * b + z[0] * c, x[1] * a + y[1] * b + z[1] * c, x[2] * a + y[2] * b + z[2] * c);
    };
    f4 uu = mul(1, 0, 0) * nh * aspect;                                     /* :62 */
    f4 vv = mul(0, 1, 0) * nh;                                              /* :63 */
    f4 ll = mul(0, 0, -1);                                                  /* :64 */
    for (int k = 0; k < 4; k++) { u[k] = uu[k]; v[k] = vv[k]; look[k] = ll[k]; }
}

ORC_API unsigned char orc_tonemap_byte(float c) { return tonemap_byte(c); }

ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, 16);
}

ORC_API void orc_sincos(float p, int mode, float *s, float *c) {
    if (mode == 0) { *s = std::sin(p); *c = std::cos(p); }
    else sincos_poly(p, s, c);
}

ORC_API void orc_hemisphere(const float *normal, float u1, float u2, int sincos_mode, float *out) {
    f4 d = random_vec_from_hemisphere(f4(normal[0], normal[1], normal[2]), u1, u2, sincos_mode);
    memcpy(out, d.v, 16);
}

ORC_API int orc_box_intersect(const float *bmin, const float *bmax, const float *ray6, float tmax) {
    Ray r(f4(ray6[0], ray6[1], ray6[2]), f4(ray6[3], ray6[4], ray6[5]));
    return box_intersect(bmin, bmax, r, tmax) ? 1 : 0;
}

/* raycheck.py:22-32 — RMS = ||a-b||_2 / sqrt(n) over uint8 RGB, but subtracting in
 * double (the original subtracts uint8 arrays, which wraps modulo 256). */
ORC_API double orc_raycheck_rms(const uint8_t *a, const uint8_t *b, int64_t n) {
    double acc = 0;
    for (int64_t k = 0; k < n; k++) { double d = (double)a[k] - (double)b[k]; acc += d * d; }
    return n > 0 ? std::sqrt(acc) / std::sqrt((double)n) : 0.0;
}
