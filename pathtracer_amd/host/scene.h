// scene.h — host scene container: what the reference keeps in Scene / Geometry /
// Material / Camera (src/scene/scene.h:33-84, geometry.h:11-48, material.h:80-134,
// camera.h:9-37), held here as flat arrays ready to be flattened to the SoA layout
// of include/mi_tracer.h.  Primitive ids are STABLE = order of addition (parse
// order); the reference's Geometry::index depends on two unstable sorts
// (scene.h:52-61), which is not reproduced.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi_tracer.h"
#include "vec4.h"

namespace pt {

// material.h:129-133 — ke, kd, kr, kt zero; index (1,1,1,1).
struct Material {
    V4 ke, kd, kr, kt;
    V4 index{1.0f};
};

struct Triangle {
    V4 a, b, c;                 // world-space vertices (trimesh.cu:33-35)
    V4 normal, uinv, vinv, ninv; // trimesh.cu:24,36-39
    uint32_t mat[3];            // material ids of a, b, c (all equal unless per-vertex)
    bool has_vmat = false;      // trimesh.cu:41
};

struct Sphere {
    V4 center;                  // w = 0 (built from a glm::vec3, Parser.cu:339)
    float radius = 1;
    uint32_t mat = 0;
};

struct PrimRef {
    bool is_sphere;
    uint32_t index;             // into tris / spheres
};

// camera.h:30-36 defaults; camera.cu:18-65 setters.
class Camera {
public:
    void setEye(const V4 &e) { eye = e; }
    void setLookQuat(float r, float i, float j, float k);     // camera.cu:18-33
    void setLook(const V4 &viewDir, const V4 &upDir);          // camera.cu:35-45
    void setFOV(float fovDeg);                                 // camera.cu:47-52
    void setAspectRatio(float ar);                             // camera.cu:54-58
    double getAspectRatio() const { return aspectRatio; }
    V4 eye, u{1, 0, 0}, v{0, 1, 0}, look{0, 1, -1};

private:
    void update();                                             // camera.cu:60-65
    float normalizedHeight = 1, aspectRatio = 1;
    float m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};         // m[col][row]
};

struct Aabb {
    float lo[3], hi[3];
};

class Scene {
public:
    uint32_t addMaterial(const Material &m);   // de-duplicated by bit pattern
    // TrimeshFace::create (trimesh.cu:7-47): returns false for a degenerate face.
    bool addTriangle(const V4 &a, const V4 &b, const V4 &c, const uint32_t mat[3], bool has_vmat);
    void addSphere(const V4 &center, float radius, uint32_t mat);
    size_t primCount() const { return prims.size(); }

    // Build the BVH (own binned-SAH builder, one primitive per leaf, child boxes
    // stored in the parent) and flatten everything to SoA.  Idempotent.
    // host_tree = false: flatten the primitives only (parse order, no nodes) for a tree built on the device
    void finalize(bool host_tree = true);
    // View of the flattened arrays; valid until the scene is modified/destroyed.
    const mi_scene_soa &soa() const { return soa_; }
    // Raw triangle vertices in SLOT order, 3 float4 per triangle (for checkers).
    const std::vector<float> &slotVertices() const { return slot_verts_; }
    int bvhDepth() const { return bvh_depth_; }
    double sahCost() const { return sah_cost_; }

    Camera camera;
    std::vector<Material> materials;
    std::vector<Triangle> tris;
    std::vector<Sphere> spheres;
    std::vector<PrimRef> prims; // stable id = position

private:
    Aabb primBounds(const PrimRef &p) const;   // geometry.h:13-22
    std::vector<float> nodes_, tri_v0_, tri_uinv_, tri_vinv_, tri_ninv_, tri_normal_, sph_cr_, mats_, slot_verts_;
    std::vector<uint32_t> tri_mat_, sph_mat_;
    std::vector<uint8_t> tri_has_vmat_;
    std::vector<int32_t> prim_id_;
    mi_scene_soa soa_{};
    int bvh_depth_ = 0;
    double sah_cost_ = 0;
    bool finalized_ = false, has_host_tree_ = false;
};

// .ray text -> Scene (ray_parser.cpp).  Throws ParseError.
struct ParseError {
    std::string message;
};
Scene *parseRayFile(const std::string &path);
Scene *parseRayText(const std::string &text, const std::string &basePath);

} // namespace pt
