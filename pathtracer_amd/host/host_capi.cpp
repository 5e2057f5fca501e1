// host_capi.cpp — C entry points over the host side (scene loading, BVH build,
// flattening, HipTracer) for non-C++ callers (the Python tests and bench.py bind
// these with ctypes).  Nothing here computes pixels.
#include <cstring>
#include <string>

#include "tracer.h"

#define PT_API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_err;

PT_API const char *pt_last_error() { return g_err.c_str(); }

// Parse .ray text (is_path = 0) or a file (is_path = 1), build the BVH, flatten.
PT_API pt::Scene *pt_scene_load(const char *text_or_path, int is_path) {
    try {
        pt::Scene *s = is_path ? pt::parseRayFile(text_or_path) : pt::parseRayText(text_or_path, ".");
        s->finalize();
        return s;
    } catch (pt::ParseError &e) {
        g_err = e.message;
        return nullptr;
    }
}
PT_API void pt_scene_free(pt::Scene *s) { delete s; }
PT_API const mi_scene_soa *pt_scene_soa(pt::Scene *s) { return &s->soa(); }
PT_API const float *pt_scene_slot_vertices(pt::Scene *s) { return s->slotVertices().data(); }
PT_API double pt_scene_aspect(pt::Scene *s) { return s->camera.getAspectRatio(); }
PT_API int pt_scene_bvh_depth(pt::Scene *s) { return s->bvhDepth(); }
PT_API double pt_scene_sah_cost(pt::Scene *s) { return s->sahCost(); }

// Programmatic scene construction (procedural meshes too large to round-trip as text).
PT_API pt::Scene *pt_scene_new() { return new pt::Scene; }
PT_API uint32_t pt_scene_add_material(pt::Scene *s, const float *m20) {
    pt::Material m;
    pt::V4 *f[5] = {&m.ke, &m.kd, &m.kr, &m.kt, &m.index};
    for (int k = 0; k < 5; k++) *f[k] = pt::V4(m20[4 * k], m20[4 * k + 1], m20[4 * k + 2], m20[4 * k + 3]);
    return s->addMaterial(m);
}
PT_API int64_t pt_scene_add_mesh(pt::Scene *s, const float *verts3, int64_t n_verts, const int32_t *faces3,
                                 int64_t n_faces, uint32_t mat) {
    int64_t added = 0;
    uint32_t mm[3] = {mat, mat, mat};
    for (int64_t f = 0; f < n_faces; f++) {
        const int32_t *i = faces3 + 3 * f;
        if (i[0] < 0 || i[1] < 0 || i[2] < 0 || i[0] >= n_verts || i[1] >= n_verts || i[2] >= n_verts) return -1;
        const float *a = verts3 + 3 * i[0], *b = verts3 + 3 * i[1], *c = verts3 + 3 * i[2];
        added += s->addTriangle(pt::V4(a[0], a[1], a[2]), pt::V4(b[0], b[1], b[2]), pt::V4(c[0], c[1], c[2]), mm, false) ? 1 : 0;
    }
    return added;
}
PT_API void pt_scene_add_sphere(pt::Scene *s, const float *center3, float radius, uint32_t mat) {
    s->addSphere(pt::V4(center3[0], center3[1], center3[2]), radius, mat);
}
PT_API void pt_scene_set_camera(pt::Scene *s, const float *eye3, const float *viewdir3, const float *updir3,
                                float fov_deg, float aspect) {
    s->camera.setEye(pt::V4(eye3[0], eye3[1], eye3[2]));
    s->camera.setFOV(fov_deg);
    s->camera.setAspectRatio(aspect);
    s->camera.setLook(pt::V4(viewdir3[0], viewdir3[1], viewdir3[2]), pt::V4(updir3[0], updir3[1], updir3[2]));
}
PT_API void pt_scene_finalize(pt::Scene *s) { s->finalize(); }

// The Tracer-shaped object (tracer.h) for callers without C++.
PT_API pt::HipTracer *pt_hip_tracer_new(int depth, int samples, int device, unsigned long long seed) {
    pt::TraceOptions o;
    o.depth = depth; o.samples = samples; o.device = device; o.seed = seed;
    auto *t = new pt::HipTracer(o);
    if (!t->ok()) { g_err = t->error(); delete t; return nullptr; }
    return t;
}
PT_API pt::HipTracer *pt_hip_tracer_new_group(int depth, int samples, int first_device, int n_devices, unsigned long long seed) {
    pt::TraceOptions o;
    o.depth = depth; o.samples = samples; o.device = first_device; o.n_devices = n_devices; o.seed = seed;
    auto *t = new pt::HipTracer(o);
    if (!t->ok()) { g_err = t->error(); delete t; return nullptr; }
    return t;
}
PT_API void pt_tracer_free(pt::HipTracer *t) { delete t; }
PT_API int pt_tracer_load_cubemap(pt::HipTracer *t, const char *one_face_file) { return t->loadCubemap(one_face_file) ? 1 : 0; }
// matchCubemapFiles for tests: 1 and the six names joined by '\n' in out (NUL terminated) on success
PT_API int pt_match_cubemap(const char *one, char *out, size_t cap) {
    std::string names[6], dir, joined;
    if (!pt::matchCubemapFiles(one, names, dir)) return 0;
    for (int i = 0; i < 6; i++) joined += names[i] + (i < 5 ? "\n" : "");
    if (joined.size() + 1 > cap) return 0;
    memcpy(out, joined.c_str(), joined.size() + 1);
    return 1;
}
// readImage for tests: returns 1 and fills w,h; rgb (w*h*3, row 0 = bottom) is copied into out when out != NULL and cap is large enough
PT_API int pt_read_image(const char *path, int *w, int *h, unsigned char *out, size_t cap) {
    std::vector<unsigned char> rgb;
    if (!pt::readImage(path, *w, *h, rgb)) return 0;
    if (out && cap >= rgb.size()) memcpy(out, rgb.data(), rgb.size());
    return 1;
}
PT_API int pt_tracer_load_scene(pt::HipTracer *t, const char *text_or_path, int is_path) {
    bool ok = is_path ? t->loadScene(text_or_path) : t->loadSceneText(text_or_path);
    if (!ok) g_err = t->error();
    return ok ? 1 : 0;
}
PT_API double pt_tracer_aspect(pt::HipTracer *t) { return t->aspectRatio(); }
PT_API void pt_tracer_trace_setup(pt::HipTracer *t, int w, int h) { t->traceSetup(w, h); }
PT_API void pt_tracer_trace_image(pt::HipTracer *t, int w, int h) { t->traceImage(w, h); }
PT_API int pt_tracer_wait_render(pt::HipTracer *t) {
    t->waitRender();
    if (!t->ok()) g_err = t->error();
    return t->ok() ? 1 : 0;
}
PT_API void pt_tracer_get_buffer(pt::HipTracer *t, unsigned char **buf, int *w, int *h) { t->getBuffer(*buf, *w, *h); }
PT_API void pt_tracer_stats(pt::HipTracer *t, mi_stats *out) { *out = t->stats(); }
PT_API int pt_write_image(const char *path, int w, int h, const unsigned char *rgb) { return pt::writeImage(path, w, h, rgb) ? 1 : 0; }
