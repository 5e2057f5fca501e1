// scene.cpp — primitive preprocessing, camera set-up, BVH construction and SoA
// flattening on the host.
//
// Preprocessing restates what the reference does at load time
// (TrimeshFace::create, src/scene/trimesh.cu:7-47; Geometry bounds,
// src/scene/geometry.h:13-22; Camera setters, src/scene/camera.cu:18-65).
// The BVH builder is this project's own design (binned SAH, top-down, 16 bins):
// the reference's genCluster (src/scene/bvh.cu:66-145) has known defects (SURVEY
// S11) and only its traversal RESULTS are a parity target.
#include "scene.h"

#include <algorithm>
#include <cstring>
#include <limits>
#include <numeric>

namespace pt {

// ------------------------------------------------------------------ camera
void Camera::setLookQuat(float r, float i, float j, float k) { // camera.cu:18-33
    float q[3][3];
    q[0][0] = 1.0 - 2.0 * (i * i + j * j);
    q[0][1] = 2.0 * (r * i - j * k);
    q[0][2] = 2.0 * (j * r + i * k);
    q[1][0] = 2.0 * (r * i + j * k);
    q[1][1] = 1.0 - 2.0 * (j * j + r * r);
    q[1][2] = 2.0 * (i * j - r * k);
    q[2][0] = 2.0 * (j * r - i * k);
    q[2][1] = 2.0 * (i * j + r * k);
    q[2][2] = 1.0 - 2.0 * (i * i + r * r);
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) m[c][rr] = q[rr][c]; // transpose (:31)
    update();
}

void Camera::setLook(const V4 &viewDir, const V4 &upDir) { // camera.cu:35-45
    V4 z = -viewDir;
    const V4 &y = upDir;
    V4 x = cross(y, z);
    for (int r = 0; r < 3; r++) { m[0][r] = x[r]; m[1][r] = y[r]; m[2][r] = z[r]; }
    update();
}

void Camera::setFOV(float fov) { // camera.cu:47-52
    fov /= (180.0f / 3.141592f);
    normalizedHeight = 2 * std::tan(fov / 2);
    update();
}

void Camera::setAspectRatio(float ar) { // camera.cu:54-58
    aspectRatio = ar;
    update();
}

void Camera::update() { // camera.cu:60-65
    auto mul = [&](float a, float b, float c) {
        return V4(m[0][0] * a + m[1][0] * b + m[2][0] * c, m[0][1] * a + m[1][1] * b + m[2][1] * c,
                  m[0][2] * a + m[1][2] * b + m[2][2] * c);
    };
    u = mul(1, 0, 0) * normalizedHeight * aspectRatio;
    v = mul(0, 1, 0) * normalizedHeight;
    look = mul(0, 0, -1);
}

// ------------------------------------------------------------------ scene
uint32_t Scene::addMaterial(const Material &mat) {
    for (size_t i = materials.size(); i-- > 0;) // recent materials first: scenes reuse the last one
        if (std::memcmp(&materials[i], &mat, sizeof(Material)) == 0) return (uint32_t)i;
    materials.push_back(mat);
    finalized_ = false;
    return (uint32_t)materials.size() - 1;
}

bool Scene::addTriangle(const V4 &a, const V4 &b, const V4 &c, const uint32_t mat[3], bool has_vmat) {
    V4 vab = b - a, vac = c - a, vcb = b - c;                                  // trimesh.cu:13-15
    if (length2(vab) == 0.0f || length2(vac) == 0.0f || length2(vcb) == 0.0f) // :18-21
        return false;
    Triangle t;
    t.a = a; t.b = b; t.c = c;
    t.normal = normalize(cross(vab, vac));                                     // :24
    // inverse of the 3x3 with columns (vab, vac, normal): cofactors times one
    // reciprocal determinant, as glm::inverse(mat3) evaluates it (:27-30).
    const float m[3][3] = {{vab.x, vab.y, vab.z}, {vac.x, vac.y, vac.z}, {t.normal.x, t.normal.y, t.normal.z}};
    const float ood = 1.0f / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2])
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
    t.uinv = {inv[0][0], inv[1][0], inv[2][0]};                                // :37
    t.vinv = {inv[0][1], inv[1][1], inv[2][1]};                                // :38
    t.ninv = {inv[0][2], inv[1][2], inv[2][2]};                                // :39
    t.mat[0] = mat[0]; t.mat[1] = mat[1]; t.mat[2] = mat[2];
    t.has_vmat = has_vmat;
    prims.push_back({false, (uint32_t)tris.size()});
    tris.push_back(t);
    finalized_ = false;
    return true;
}

void Scene::addSphere(const V4 &center, float radius, uint32_t mat) {
    Sphere s;
    s.center = V4(center.x, center.y, center.z);
    s.radius = radius;
    s.mat = mat;
    prims.push_back({true, (uint32_t)spheres.size()});
    spheres.push_back(s);
    finalized_ = false;
}

Aabb Scene::primBounds(const PrimRef &p) const { // geometry.h:13-22
    Aabb b;
    if (p.is_sphere) {
        const Sphere &s = spheres[p.index];
        for (int k = 0; k < 3; k++) { b.lo[k] = s.center[k] - s.radius; b.hi[k] = s.center[k] + s.radius; }
    } else {
        const Triangle &t = tris[p.index];
        for (int k = 0; k < 3; k++) {
            b.lo[k] = std::fmin(std::fmin(t.a[k], t.b[k]), t.c[k]);
            b.hi[k] = std::fmax(std::fmax(t.a[k], t.b[k]), t.c[k]);
        }
    }
    // A sphere of negative radius (mirrored scale) would give lo > hi.
    for (int k = 0; k < 3; k++)
        if (b.lo[k] > b.hi[k]) std::swap(b.lo[k], b.hi[k]);
    return b;
}

// ------------------------------------------------------------------ BVH build
namespace {

struct BuildPrim {
    Aabb box;
    float cen[3];
    uint32_t id; // stable prim id
};

inline void grow(Aabb &a, const Aabb &b) {
    for (int k = 0; k < 3; k++) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
}
inline Aabb emptyBox() {
    Aabb b;
    for (int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<float>::max(); b.hi[k] = -std::numeric_limits<float>::max(); }
    return b;
}
inline double halfArea(const Aabb &b) {
    double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}

struct BuildNode {
    Aabb box[2];
    int32_t child[2]; // >=0 interior node, <0 leaf: ~prim position in `order`
};

constexpr int kBins = 16;
constexpr int kMedianDepth = 36; // beyond this depth split at the object median so depth stays < 64

struct Builder {
    std::vector<BuildPrim> &p;
    std::vector<BuildNode> nodes;
    int maxDepth = 0;
    double sah = 0;

    explicit Builder(std::vector<BuildPrim> &prims) : p(prims) { nodes.reserve(prims.size()); }

    // Builds the subtree over p[lo,hi), returns its ref and box.
    int32_t build(size_t lo, size_t hi, int depth, Aabb &outBox) {
        maxDepth = std::max(maxDepth, depth);
        if (hi - lo == 1) {
            outBox = p[lo].box;
            return ~(int32_t)lo;
        }
        Aabb cb = emptyBox(), bb = emptyBox();
        for (size_t i = lo; i < hi; i++) {
            grow(bb, p[i].box);
            for (int k = 0; k < 3; k++) { cb.lo[k] = std::min(cb.lo[k], p[i].cen[k]); cb.hi[k] = std::max(cb.hi[k], p[i].cen[k]); }
        }
        size_t mid = lo;
        bool split = false;
        if (depth < kMedianDepth) {
            double bestCost = std::numeric_limits<double>::max();
            int bestAxis = -1, bestBin = -1;
            for (int axis = 0; axis < 3; axis++) {
                float ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0)) continue;
                Aabb binBox[kBins];
                size_t binCount[kBins] = {};
                for (auto &b : binBox) b = emptyBox();
                float scale = kBins / ext;
                for (size_t i = lo; i < hi; i++) {
                    int b = std::min(kBins - 1, std::max(0, (int)((p[i].cen[axis] - cb.lo[axis]) * scale)));
                    binCount[b]++;
                    grow(binBox[b], p[i].box);
                }
                double rightArea[kBins];
                size_t rightCount[kBins];
                Aabb acc = emptyBox();
                size_t cnt = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    if (binCount[b]) grow(acc, binBox[b]);
                    cnt += binCount[b];
                    rightArea[b] = cnt ? halfArea(acc) : 0;
                    rightCount[b] = cnt;
                }
                acc = emptyBox();
                cnt = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    if (binCount[b]) grow(acc, binBox[b]);
                    cnt += binCount[b];
                    if (cnt == 0 || rightCount[b + 1] == 0) continue;
                    double cost = halfArea(acc) * cnt + rightArea[b + 1] * rightCount[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
                }
            }
            if (bestAxis >= 0) {
                float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
                float scale = kBins / ext;
                float base = cb.lo[bestAxis];
                auto it = std::partition(p.begin() + lo, p.begin() + hi, [&](const BuildPrim &q) {
                    int b = std::min(kBins - 1, std::max(0, (int)((q.cen[bestAxis] - base) * scale)));
                    return b <= bestBin;
                });
                mid = (size_t)(it - p.begin());
                split = mid > lo && mid < hi;
            }
        }
        if (!split) { // coincident centroids or depth guard: object median on the widest axis
            int axis = 0;
            for (int k = 1; k < 3; k++)
                if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            mid = lo + (hi - lo) / 2;
            std::nth_element(p.begin() + lo, p.begin() + mid, p.begin() + hi, [axis](const BuildPrim &a, const BuildPrim &b) {
                return a.cen[axis] < b.cen[axis] || (a.cen[axis] == b.cen[axis] && a.id < b.id);
            });
        }
        size_t self = nodes.size();
        nodes.emplace_back();
        Aabb lb, rb;
        int32_t l = build(lo, mid, depth + 1, lb);
        int32_t r = build(mid, hi, depth + 1, rb);
        nodes[self].box[0] = lb; nodes[self].box[1] = rb;
        nodes[self].child[0] = l; nodes[self].child[1] = r;
        outBox = lb;
        grow(outBox, rb);
        sah += halfArea(outBox);
        return (int32_t)self;
    }
};

inline void put4(std::vector<float> &dst, const V4 &v) { dst.push_back(v.x); dst.push_back(v.y); dst.push_back(v.z); dst.push_back(v.w); }

} // namespace

void Scene::finalize(bool host_tree) {
    if (finalized_ && (has_host_tree_ || !host_tree)) return;
    has_host_tree_ = host_tree;
    const size_t n = prims.size();
    nodes_.clear(); tri_v0_.clear(); tri_uinv_.clear(); tri_vinv_.clear(); tri_ninv_.clear(); tri_normal_.clear();
    sph_cr_.clear(); mats_.clear(); slot_verts_.clear(); tri_mat_.clear(); sph_mat_.clear(); tri_has_vmat_.clear();
    prim_id_.clear();

    std::vector<BuildPrim> bp(n);
    Aabb sceneBox = emptyBox();
    for (size_t i = 0; i < n; i++) {
        bp[i].box = primBounds(prims[i]);
        bp[i].id = (uint32_t)i;
        for (int k = 0; k < 3; k++) bp[i].cen[k] = 0.5f * (bp[i].box.lo[k] + bp[i].box.hi[k]);
        grow(sceneBox, bp[i].box);
    }
    Builder builder(bp);
    int32_t rootRef = 0;
    if (n > 0 && host_tree) {
        Aabb rb;
        rootRef = builder.build(0, n, 0, rb);
    } else if (n == 1) {
        rootRef = ~0; // the tree will be built on the device (mi_scene_upload_build): primitives stay in parse order
    }
    bvh_depth_ = builder.maxDepth;

    // Slot assignment: triangles first, spheres after, each in BVH leaf (DFS) order
    // so that neighbouring leaves read neighbouring records.
    std::vector<int32_t> slotOfPos(n);
    uint32_t nt = 0, ns = 0;
    for (size_t pos = 0; pos < n; pos++)
        if (!prims[bp[pos].id].is_sphere) slotOfPos[pos] = (int32_t)nt++;
    for (size_t pos = 0; pos < n; pos++)
        if (prims[bp[pos].id].is_sphere) slotOfPos[pos] = (int32_t)(nt + ns++);
    prim_id_.assign(n, -1);
    tri_mat_.resize(3 * (size_t)nt);
    tri_has_vmat_.resize(nt);
    tri_v0_.reserve(4 * (size_t)nt);
    for (size_t pos = 0; pos < n; pos++) {
        const PrimRef &pr = prims[bp[pos].id];
        if (pr.is_sphere) continue;
        const Triangle &t = tris[pr.index];
        size_t slot = (size_t)slotOfPos[pos];
        prim_id_[slot] = (int32_t)bp[pos].id;
        put4(tri_v0_, t.a); put4(tri_uinv_, t.uinv); put4(tri_vinv_, t.vinv); put4(tri_ninv_, t.ninv); put4(tri_normal_, t.normal);
        put4(slot_verts_, t.a); put4(slot_verts_, t.b); put4(slot_verts_, t.c);
        tri_mat_[3 * slot] = t.mat[0]; tri_mat_[3 * slot + 1] = t.mat[1]; tri_mat_[3 * slot + 2] = t.mat[2];
        tri_has_vmat_[slot] = t.has_vmat ? 1 : 0;
    }
    for (size_t pos = 0; pos < n; pos++) {
        const PrimRef &pr = prims[bp[pos].id];
        if (!pr.is_sphere) continue;
        const Sphere &s = spheres[pr.index];
        prim_id_[(size_t)slotOfPos[pos]] = (int32_t)bp[pos].id;
        sph_cr_.push_back(s.center.x); sph_cr_.push_back(s.center.y); sph_cr_.push_back(s.center.z); sph_cr_.push_back(s.radius);
        sph_mat_.push_back(s.mat);
    }

    // Child boxes are padded by 1e-5 x the scene diagonal so that the slab test
    // (which has no tolerance of its own, bbox.cu:15-34) never culls a primitive
    // whose computed hit lies a rounding error outside its exact bounds: BVH
    // traversal then returns exactly what a test of every primitive returns.
    float diag = 0;
    if (n > 0) {
        double d2 = 0;
        for (int k = 0; k < 3; k++) { double e = (double)sceneBox.hi[k] - sceneBox.lo[k]; d2 += e * e; }
        diag = (float)std::sqrt(d2);
    }
    const float pad = 1e-5f * diag + 1e-30f;
    nodes_.resize(16 * builder.nodes.size());
    for (size_t i = 0; i < builder.nodes.size(); i++) {
        const BuildNode &bn = builder.nodes[i];
        float *o = nodes_.data() + 16 * i;
        for (int c = 0; c < 2; c++)
            for (int k = 0; k < 3; k++) {
                o[6 * c + k] = bn.box[c].lo[k] - pad;
                o[6 * c + 3 + k] = bn.box[c].hi[k] + pad;
            }
        for (int c = 0; c < 2; c++) {
            int32_t ref = bn.child[c] >= 0 ? bn.child[c] : ~slotOfPos[(size_t)~bn.child[c]];
            std::memcpy(o + 12 + c, &ref, 4);
        }
        o[14] = 0; o[15] = 0;
    }
    // depth of every node (root = 0) in the first spare word: read only by the
    // instrumented extend kernel (visit histogram per tree level)
    if (n > 1 && host_tree) {
        std::vector<std::pair<int32_t, int32_t>> st{{rootRef, 0}};
        while (!st.empty()) {
            auto [ref, depth] = st.back();
            st.pop_back();
            std::memcpy(nodes_.data() + 16 * (size_t)ref + 14, &depth, 4);
            for (int c = 0; c < 2; c++)
                if (builder.nodes[ref].child[c] >= 0) st.push_back({builder.nodes[ref].child[c], depth + 1});
        }
    }
    if (n > 0 && rootRef < 0) rootRef = ~slotOfPos[(size_t)~rootRef];
    sah_cost_ = (n > 1 && host_tree && halfArea(sceneBox) > 0) ? builder.sah / halfArea(sceneBox) : 0;

    for (const Material &m : materials) { put4(mats_, m.ke); put4(mats_, m.kd); put4(mats_, m.kr); put4(mats_, m.kt); put4(mats_, m.index); }

    soa_ = mi_scene_soa{};
    soa_.n_nodes = (uint32_t)builder.nodes.size();
    soa_.nodes = nodes_.data();
    soa_.root_ref = rootRef;
    soa_.n_tri = nt;
    soa_.tri_v0 = tri_v0_.data(); soa_.tri_uinv = tri_uinv_.data(); soa_.tri_vinv = tri_vinv_.data();
    soa_.tri_ninv = tri_ninv_.data(); soa_.tri_normal = tri_normal_.data();
    soa_.tri_mat = tri_mat_.data(); soa_.tri_has_vmat = tri_has_vmat_.data();
    soa_.n_sph = ns;
    soa_.sph_center_radius = sph_cr_.data(); soa_.sph_mat = sph_mat_.data();
    soa_.n_mat = (uint32_t)materials.size();
    soa_.materials = mats_.data();
    soa_.prim_id = prim_id_.data();
    const V4 *cv[4] = {&camera.eye, &camera.u, &camera.v, &camera.look};
    float *dst[4] = {soa_.cam_eye, soa_.cam_u, soa_.cam_v, soa_.cam_look};
    for (int q = 0; q < 4; q++) { dst[q][0] = cv[q]->x; dst[q][1] = cv[q]->y; dst[q][2] = cv[q]->z; dst[q][3] = cv[q]->w; }
    finalized_ = true;
}

} // namespace pt
