// integration/HipTracer.cu — LISTING (see HipTracer.h): reference-side binding of libmi_tracer.so.
// Host code only; named .cu because the reference's build globs *.cu (src/CMakeLists.txt:19-21).
#include "HipTracer.h"

#include <cstring>
#include <iostream>
#include <map>

#include "scene/scene.h"
#include "scene/cubeMap.h"
#include "scene/material.h"

HipTracer::HipTracer(TraceUI *ui) : Tracer(ui), seed(ui->m_seed), own_tree(ui->m_hipOwnTree) {
    status = mi_ctx_create(ui->m_hipDevice, &ctx);          // replaces GpuTracer's implicit device 0 (GpuTracer.cu:165-181)
    if (status != MI_OK) std::cerr << "HipTracer: " << mi_last_error() << std::endl;
}

HipTracer::~HipTracer() {
    if (worker.joinable()) worker.join();
    mi_ctx_destroy(ctx);                                     // the reference never frees its device memory
}

static void put4(std::vector<float> &v, const f4 &a) { v.push_back(a[0]); v.push_back(a[1]); v.push_back(a[2]); v.push_back(a[3]); }

// Materials are stored by value in every primitive (material.h:129-133, 80 B each); the device wants a table.
// Equal bit patterns share an entry.
struct MatTable {
    std::map<std::vector<uint32_t>, uint32_t> seen;
    std::vector<float> *out;
    uint32_t add(const Material &m) {
        float rec[20];
        const f4 *k[5] = {&m._ke, &m._kd, &m._kr, &m._kt, &m._index};
        for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) rec[4 * i + j] = (*k[i])[j];
        std::vector<uint32_t> key(20);
        std::memcpy(key.data(), rec, sizeof rec);
        auto it = seen.find(key);
        if (it != seen.end()) return it->second;
        const uint32_t id = (uint32_t)seen.size();
        seen.emplace(std::move(key), id);
        out->insert(out->end(), rec, rec + 20);
        return id;
    }
};

// Scene::actual (sorted by updateObjects, scene.h:52-61) and the flattened Cluster tree (bvh.cu:239-252; after
// update_children the left/right pointers point into one array, bvh.h:19-34) -> mi_scene_soa.
//   * primitive slot: triangles first, then spheres, each class in Scene::actual order;
//   * one 64-B node per INTERIOR Cluster: the boxes of its two children as the reference stores them (unpadded;
//     the library rounds its 16-bit plane codes outward, which is all the margin the slab test needs — measured
//     100 % primary-hit agreement with the test of every primitive, tests/test_gpu_reference_shaped_tree.py),
//     child ref = node index of an interior child, ~slot of a leaf child (Cluster::obj != nullptr, bvh.cu:72-76);
//   * prim_id[slot] = position in Scene::actual: any fixed total order serves the tie rule of mi_scene_soa.prim_id.
bool HipTracer::flatten() {
    const std::vector<Geometry> &actual = scene->actual;
    const size_t n = actual.size();
    std::vector<int32_t> slot_of(n);
    uint32_t n_tri = 0, n_sph = 0;
    for (size_t i = 0; i < n; i++) (actual[i].isSphere ? n_sph : n_tri)++;
    uint32_t t = 0, s = 0;
    for (size_t i = 0; i < n; i++) slot_of[i] = actual[i].isSphere ? (int32_t)(n_tri + s++) : (int32_t)t++;

    for (auto *v : {&nodes, &tri_v0, &tri_uinv, &tri_vinv, &tri_ninv, &tri_normal, &sph, &mats, &tri_verts}) v->clear();
    tri_mat.clear(); sph_mat.clear(); tri_has_vmat.clear();
    prim_id.assign(n, 0);
    MatTable table{{}, &mats};
    for (size_t i = 0; i < n; i++) {
        const Geometry &g = actual[i];
        prim_id[slot_of[i]] = (int32_t)i;
        if (g.isSphere) {
            const Sphere &sp = g.obj.sphere;                                   // sphere.h:13-15
            sph.push_back(sp.position[0]); sph.push_back(sp.position[1]); sph.push_back(sp.position[2]); sph.push_back(sp.radius);
            sph_mat.push_back(table.add(sp.mat));
        } else {
            const TrimeshFace &f = g.obj.face;                                 // trimesh.h:17-23
            put4(tri_v0, f.vertices[0]);                                       // trimesh.cu:88 reads vertices[0]
            put4(tri_uinv, f.uInv); put4(tri_vinv, f.vInv); put4(tri_ninv, f.nInv); put4(tri_normal, f.normal);
            for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) tri_verts.push_back(f.vertices[k][a]);   // for -B: the boxes of geometry.h:13-29
            tri_has_vmat.push_back(f.hasMaterials ? 1 : 0);
            for (int k = 0; k < 3; k++) tri_mat.push_back(table.add(f.materials[f.hasMaterials ? k : 0]));   // trimesh.cu:40-45
        }
    }

    // interior Clusters in the order a walk from the root meets them
    const Cluster *root = scene->bvh.root;
    int32_t root_ref = ~0;
    if (root && !root->obj) {
        std::map<const Cluster *, int32_t> index;
        std::vector<const Cluster *> order{root};
        index[root] = 0;
        for (size_t q = 0; q < order.size(); q++)
            for (const Cluster *c : {order[q]->left, order[q]->right})
                if (!c->obj) { index[c] = (int32_t)order.size(); order.push_back(c); }
        nodes.assign(order.size() * 16, 0.0f);
        for (size_t q = 0; q < order.size(); q++) {
            float *nd = &nodes[16 * q];
            const Cluster *ch[2] = {order[q]->left, order[q]->right};
            for (int c = 0; c < 2; c++) {
                const f4 lo = ch[c]->bbox.getMin(), hi = ch[c]->bbox.getMax();  // bbox.h:15-17
                for (int a = 0; a < 3; a++) { nd[6 * c + a] = lo[a]; nd[6 * c + 3 + a] = hi[a]; }
                const int32_t ref = ch[c]->obj ? ~slot_of[ch[c]->obj - actual.data()] : index[ch[c]];
                std::memcpy(&nd[12 + c], &ref, 4);
            }
        }
        root_ref = 0;
    }

    soa = mi_scene_soa{};
    soa.n_nodes = (uint32_t)(nodes.size() / 16); soa.nodes = nodes.data(); soa.root_ref = root_ref;
    soa.n_tri = n_tri; soa.tri_v0 = tri_v0.data(); soa.tri_uinv = tri_uinv.data(); soa.tri_vinv = tri_vinv.data();
    soa.tri_ninv = tri_ninv.data(); soa.tri_normal = tri_normal.data(); soa.tri_mat = tri_mat.data(); soa.tri_has_vmat = tri_has_vmat.data();
    soa.n_sph = n_sph; soa.sph_center_radius = sph.data(); soa.sph_mat = sph_mat.data();
    soa.n_mat = (uint32_t)(mats.size() / 20); soa.materials = mats.data();
    soa.prim_id = prim_id.data();
    const Camera &cam = scene->getCamera();                                    // camera.h:33-36
    for (int k = 0; k < 4; k++) { soa.cam_eye[k] = cam.eye[k]; soa.cam_u[k] = cam.u[k]; soa.cam_v[k] = cam.v[k]; soa.cam_look[k] = cam.look[k]; }
    // Either the reference's own hierarchy, validated on the host (refs, coverage, depth), or (-B) a tree the library
    // builds on the GPU from the primitives alone — binned SAH, 2 ms for 82 k triangles — in which case `nodes` is
    // ignored.  The image is the same either way (the closest hit does not depend on the tree: mi_scene_soa.prim_id);
    // the frame rate is not: the kernels' rate on a mesh follows the tree's quality (report §5.2).
    status = own_tree ? mi_scene_upload_build(ctx, &soa, tri_verts.data(), nullptr) : mi_scene_upload(ctx, &soa);
    if (status != MI_OK) std::cerr << "HipTracer: " << mi_last_error() << std::endl;
    return status == MI_OK;
}

// The UI owns the cubemap (TraceUI.h:78); its six TextureMaps hold tightly packed RGB8 with row 0 at the bottom
// (material.h:52-54, pngimage.cu:210-213) — the layout mi_cubemap expects.
bool HipTracer::uploadCubeMap() {
    if (!cubeMap) return mi_cubemap_upload(ctx, nullptr) == MI_OK;
    mi_cubemap cm{};
    for (int f = 0; f < 6; f++) {
        const TextureMap *m = cubeMap->tMap[f].get();
        if (!m) return mi_cubemap_upload(ctx, nullptr) == MI_OK;
        cm.width[f] = m->width; cm.height[f] = m->height; cm.rgb[f] = m->data.data();
    }
    return mi_cubemap_upload(ctx, &cm) == MI_OK;
}

void HipTracer::traceImage(int w, int h) {
    if (!ctx) return;
    if (worker.joinable()) worker.join();         // a second traceImage without waitRender: assigning over a joinable thread terminates
    // Once per scene; the reference re-clones on every call (GpuTracer.cu:173).  Tracer::loadScene is not virtual, so a
    // reload is recognised here: the cache is keyed on the Scene, its BVH root and its primitive count (loadScene leaks
    // the old Scene, tracer.cu:62, so its address is not reused; the other two keys guard a build that frees it).
    const void *root_now = scene ? (const void *)scene->bvh.root : nullptr;
    const size_t prims_now = scene ? scene->actual.size() : 0;
    if (uploaded != scene || uploaded_root != root_now || uploaded_prims != prims_now) {
        if (!flatten()) return;
        uploaded = scene; uploaded_root = root_now; uploaded_prims = prims_now;
        uploaded_cube = nullptr; cube_valid = false;
    }
    if (!cube_valid || uploaded_cube != (const void *)cubeMap) {   // the cubemap has its own key: the UI may set it after the scene
        if (!uploadCubeMap()) return;
        uploaded_cube = (const void *)cubeMap; cube_valid = true;
    }
    worker = std::thread([this, w, h] {
        mi_render_params p{};
        p.width = w; p.height = h;
        p.samples_per_dim = (int32_t)samples;     // -s N: N*N paths per pixel (CpuTracer.cu:99-109)
        p.depth = (int32_t)depth;                 // -r D (CpuTracer.cu:119-131)
        p.seed = seed;
        // `buffer` is the RGB8 frame of traceSetup (tracer.cu:87-93), (i + j*w)*3, row 0 at the bottom: what
        // mi_render writes, tonemapped as setPixel does (tracer.cu:95-100).  `buffer` is a std::vector in pageable
        // memory (tracer.h:47): the frame reaches it through the runtime's staging copies (0.93 of the in-HBM frame
        // rate at 1080p).  A maintainer who wants the DMA path (0.97) renders into a frame from mi_host_alloc(w*h*3)
        // kept beside the context and copies it into `buffer` in waitRender, or gives `buffer` an allocator over
        // mi_host_alloc / mi_host_free — what pt::HipTracer::frameAlloc does in this repository.
        status = mi_render(ctx, &p, nullptr, buffer.data(), &stats);
        if (status != MI_OK) std::cerr << "HipTracer: " << mi_last_error() << std::endl;
    });
}

void HipTracer::waitRender() {
    if (worker.joinable()) worker.join();
}
