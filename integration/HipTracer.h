// integration/HipTracer.h — LISTING: the third Tracer backend a maintainer of the reference adds.
//
// This file and HipTracer.cu are written against the REFERENCE's classes (src/tracer.h, src/scene/*.h); they are
// not compiled in this repository (the reference needs nvcc and glm, absent here) and are not part of the
// library.  This repository ships the same subclass against its own host scene classes
// (pathtracer_amd/host/tracer.h, pt::HipTracer), and tests/test_gpu_reference_shaped_tree.py feeds
// mi_scene_upload a scene flattened exactly the way flatten() below does it.
//
// Edits to existing reference files (all one-liners):
//   src/tracer.h          after the GpuTracer declaration:   #include "../integration/HipTracer.h"
//   src/main.cu:10-11     ui->m_hip ? ui->setRayTracer(new HipTracer(ui)) : ui->m_gpu ? ... GpuTracer ... : ... CpuTracer ...
//   src/ui/TraceUI.h:47   bool m_hip = false, m_hipOwnTree = false; int m_hipDevice = 0; unsigned long long m_seed = 0;
//   src/ui/CommandLineUI.cu:27   getopt string "t:r:w:hj:c:s:gq:G:S:B";  case 'G': m_hip = true; m_hipDevice = atoi(optarg); break;
//                                case 'S': m_seed = strtoull(optarg, nullptr, 10); break;
//                                case 'B': m_hipOwnTree = true; break;   // the library builds its own BVH on the GPU
//   src/scene/bvh.h:44    class BoundedVolumeHierarchy { friend class HipTracer; ...   (root is private)
//   src/scene/geometry.h:40   friend class HipTracer;                                   (the Sphere|TrimeshFace union is private)
//   src/scene/trimesh.h:22, sphere.h:16, camera.h:28, material.h:121, cubeMap.h:11, material.h:52 (TextureMap)
//                         friend class HipTracer;                                       (fields read by flatten())
//   src/CMakeLists.txt    target_link_libraries(ray mi_tracer)   — host compiler only, no device code in these files
#pragma once
#include <thread>
#include <vector>

#include "mi_tracer.h"   // include/mi_tracer.h of this repository
#include "tracer.h"

class HipTracer : public Tracer {
public:
    explicit HipTracer(TraceUI *ui);
    ~HipTracer();

    void traceImage(int w, int h) override;   // src/tracer.h:35 — starts the render, may return before it is done
    void waitRender() override;               // src/tracer.h:37 — joins it

    bool ok() const { return status == MI_OK; }

private:
    bool flatten();                           // Scene::actual + Scene::bvh -> mi_scene_soa (this->soa)
    bool uploadCubeMap();

    mi_ctx *ctx = nullptr;
    mi_status status = MI_OK;
    mi_scene_soa soa{};
    mi_stats stats{};
    unsigned long long seed = 0;
    const Scene *uploaded = nullptr;          // the scene resident on the device, with what else identifies it
    const void *uploaded_root = nullptr;
    size_t uploaded_prims = 0;
    const void *uploaded_cube = nullptr;      // the CubeMap resident on the device (nullptr + cube_valid: white background)
    bool cube_valid = false;
    std::thread worker;
    // storage behind the borrowed pointers of `soa`
    std::vector<float> nodes, tri_v0, tri_uinv, tri_vinv, tri_ninv, tri_normal, sph, mats, tri_verts;
    bool own_tree = false;                    // -B: mi_scene_upload_build instead of the reference's Cluster tree
    std::vector<uint32_t> tri_mat, sph_mat;
    std::vector<uint8_t> tri_has_vmat;
    std::vector<int32_t> prim_id;
};
