// tracer.cpp — Tracer base class (counterpart of src/tracer.cu) and the HipTracer
// backend that drives libmi_tracer.so through its C-ABI.
#include "tracer.h"

#include <dirent.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>

namespace pt {

Tracer::~Tracer() {
    releaseFrame();
    delete scene;
}

void Tracer::releaseFrame() {
    if (buffer.p) frameFree(buffer.p);
    buffer.p = nullptr;
    buffer.n = 0;
}

void Tracer::getBuffer(unsigned char *&buf, int &w, int &h) { // tracer.cu:32-36
    buf = buffer.data();
    w = buffer_width;
    h = buffer_height;
}

double Tracer::aspectRatio() { return scene->camera.getAspectRatio(); } // tracer.cu:38-40

bool Tracer::loadScene(const char *fn) { // tracer.cu:42-85
    try {
        delete scene;
        scene = nullptr;
        scene = parseRayFile(fn);
    } catch (ParseError &e) {
        std::cerr << e.message << std::endl;
        return false;
    }
    // Where the reference sorts primitives and builds + flattens its BVH
    // (tracer.cu:80-83), build this project's BVH (or leave that to the backend) and flatten to SoA.
    scene->finalize(wantsHostTree(scene->prims.size()));
    return sceneLoaded();
}

bool Tracer::loadSceneText(const std::string &text) {
    try {
        delete scene;
        scene = nullptr;
        scene = parseRayText(text, ".");
    } catch (ParseError &e) {
        std::cerr << e.message << std::endl;
        return false;
    }
    scene->finalize(wantsHostTree(scene->prims.size()));
    return sceneLoaded();
}

void Tracer::traceSetup(int w, int h) { // tracer.cu:87-93
    waitRender();                                 // a render in flight still writes the old frame
    const size_t bytes = (size_t)w * h * 3;
    if (bytes != buffer.n) {
        releaseFrame();
        if (bytes) { buffer.p = frameAlloc(bytes); buffer.n = bytes; }
    }
    buffer_width = w;
    buffer_height = h;
    if (bytes) std::memset(buffer.p, 0, bytes);
}

// TraceUI.cu:70-148: the six names are recognised by "pos"/"neg" followed (anywhere later)
// by the axis letter; only names containing ".bmp" or ".png" are looked at; two files for
// one face stop the search.
bool matchCubemapFiles(const std::string &one, std::string matched[6], std::string &dir) {
    static const char *matcher[6][2] = {{"pos", "x"}, {"neg", "x"}, {"pos", "y"}, {"neg", "y"}, {"pos", "z"}, {"neg", "z"}};
    static const char *exts[2] = {".bmp", ".png"};
    dir = one.substr(0, one.find_last_of("/"));
    DIR *dp = opendir(dir.c_str());
    if (!dp) {
        std::cerr << "Couldn't open the directory " << dir << std::endl;
        return false;
    }
    for (int i = 0; i < 6; i++) matched[i].clear();
    int n = 0;
    while (struct dirent *ep = readdir(dp)) {
        std::string fn(ep->d_name);
        bool ext_ok = false;
        for (const char *e : exts) ext_ok = ext_ok || fn.find(e) != std::string::npos;
        if (!ext_ok) continue;
        for (int i = 0; i < 6; i++) {
            size_t p0 = fn.find(matcher[i][0]);
            if (p0 == std::string::npos || fn.find(matcher[i][1], p0) == std::string::npos) continue;
            if (!matched[i].empty()) {
                closedir(dp);
                std::cerr << matcher[i][0] << matcher[i][1] << " matches " << matched[i] << " and " << fn
                          << ", stop smartload to avoid confliction" << std::endl;
                return false;
            }
            matched[i] = fn;
            n++;
            break;
        }
        if (n == 6) break;
    }
    closedir(dp);
    if (n != 6) {
        std::cerr << "Cannot locate all six cubemap files" << std::endl;
        return false;
    }
    return true;
}

bool Tracer::loadCubemap(const char *oneFaceFile) { // TraceUI.cu:150-169
    std::string names[6], dir;
    has_cube_ = false;
    if (!matchCubemapFiles(oneFaceFile, names, dir)) return false;
    for (int i = 0; i < 6; i++) {
        const std::string path = dir + "/" + names[i];
        if (!readImage(path.c_str(), cube_[i].w, cube_[i].h, cube_[i].rgb)) {
            std::cerr << "Unable to load texture map '" << path << "'." << std::endl; // material.cu:13-16
            return false;
        }
    }
    has_cube_ = true;
    return cubemapLoaded();
}

// ------------------------------------------------------------------ HipTracer
HipTracer::HipTracer(const TraceOptions &o) : Tracer(o), opts_(o) {
    if (o.n_devices > 0) {
        std::vector<int> ids;
        for (int k = 0; k < o.n_devices; k++) ids.push_back(o.device + k);
        status_ = mi_group_create(ids.data(), o.n_devices, &group_);
    } else {
        status_ = mi_ctx_create(o.device, &ctx_);
    }
    if (status_ != MI_OK) {
        error_ = mi_last_error();
        std::cerr << "HipTracer: " << error_ << std::endl;
    }
}

// The RGB8 frame is written by every traceImage: in pinned memory, mi_render's copy is one DMA into it.
unsigned char *HipTracer::frameAlloc(size_t bytes) {
    void *p = nullptr;
    if ((ctx_ || group_) && mi_host_alloc(bytes, &p) == MI_OK && p) { pinned_ = (unsigned char *)p; return pinned_; }
    return Tracer::frameAlloc(bytes);             // a refusal only means the slower pageable copy
}
void HipTracer::frameFree(unsigned char *p) {
    if (p && p == pinned_) { mi_host_free(p); pinned_ = nullptr; }
    else Tracer::frameFree(p);
}

HipTracer::~HipTracer() {
    if (worker_.joinable()) worker_.join();
    releaseFrame();                               // while frameFree still dispatches here and the runtime is up
    if (ctx_) mi_ctx_destroy(ctx_);
    if (group_) mi_group_destroy(group_);
}

bool HipTracer::cubemapLoaded() {
    if (!ctx_ && !group_) return false;
    mi_cubemap cm;
    for (int i = 0; i < 6; i++) { cm.width[i] = cube_[i].w; cm.height[i] = cube_[i].h; cm.rgb[i] = cube_[i].rgb.data(); }
    status_ = group_ ? mi_group_cubemap_upload(group_, &cm) : mi_cubemap_upload(ctx_, &cm);
    if (status_ != MI_OK) {
        error_ = mi_last_error();
        std::cerr << "HipTracer: " << error_ << std::endl;
        return false;
    }
    return true;
}

bool HipTracer::wantsHostTree(size_t n_prims) {
    (void)n_prims;
    device_tree_ = opts_.device_bvh != 0;
    return !device_tree_;
}

bool HipTracer::sceneLoaded() { // upload replaces Scene::clone, GpuTracer.cu:172-175
    if (!ctx_ && !group_) return false;
    if (device_tree_) { // slot vertices are kept as 3 float4 per triangle; the ABI takes 9 floats
        const std::vector<float> &v4 = scene->slotVertices();
        std::vector<float> v3(v4.size() / 12 * 9);
        for (size_t t = 0; t < v4.size() / 12; t++)
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) v3[9 * t + 3 * k + a] = v4[12 * t + 4 * k + a];
        mi_build_stats bs;
        status_ = group_ ? mi_group_scene_upload_build(group_, &scene->soa(), v3.empty() ? nullptr : v3.data(), &bs)
                         : mi_scene_upload_build(ctx_, &scene->soa(), v3.empty() ? nullptr : v3.data(), &bs);
    } else {
        status_ = group_ ? mi_group_scene_upload(group_, &scene->soa()) : mi_scene_upload(ctx_, &scene->soa());
    }
    if (status_ != MI_OK) {
        error_ = mi_last_error();
        std::cerr << "HipTracer: " << error_ << std::endl;
        return false;
    }
    return true;
}

void HipTracer::traceImage(int w, int h) { // GpuTracer.cu:161-189
    if (worker_.joinable()) worker_.join();
    worker_ = std::thread([this, w, h] {
        if (!ctx_ && !group_) { status_ = MI_ERR_NO_DEVICE; return; }
        mi_render_params p{};
        p.width = w; p.height = h;
        p.samples_per_dim = (int)samples;
        p.depth = (int)depth;
        p.seed = opts_.seed;
        status_ = group_ ? mi_group_render(group_, &p, buffer.data(), &stats_) : mi_render(ctx_, &p, nullptr, buffer.data(), &stats_);
        if (status_ != MI_OK) {
            error_ = mi_last_error();
            std::cerr << "HipTracer: " << error_ << std::endl;
        }
    });
}

void HipTracer::waitRender() { // GpuTracer.cu:191-193 is a no-op; here it is the join
    if (worker_.joinable()) worker_.join();
}

} // namespace pt
