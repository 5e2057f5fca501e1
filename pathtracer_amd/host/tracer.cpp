// tracer.cpp — Tracer base class (counterpart of src/tracer.cu) and the HipTracer
// backend that drives libmi_tracer.so through its C-ABI.
#include "tracer.h"

#include <algorithm>
#include <cstring>
#include <iostream>

namespace pt {

Tracer::~Tracer() { delete scene; }

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
    // (tracer.cu:80-83), build this project's BVH and flatten to SoA.
    scene->finalize();
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
    scene->finalize();
    return sceneLoaded();
}

void Tracer::traceSetup(int w, int h) { // tracer.cu:87-93
    buffer.resize((size_t)w * h * 3);
    buffer_width = w;
    buffer_height = h;
    std::fill(buffer.begin(), buffer.end(), 0);
}

// ------------------------------------------------------------------ HipTracer
HipTracer::HipTracer(const TraceOptions &o) : Tracer(o), opts_(o) {
    status_ = mi_ctx_create(o.device, &ctx_);
    if (status_ != MI_OK) {
        error_ = mi_last_error();
        std::cerr << "HipTracer: " << error_ << std::endl;
    }
}

HipTracer::~HipTracer() {
    if (worker_.joinable()) worker_.join();
    if (ctx_) mi_ctx_destroy(ctx_);
}

bool HipTracer::sceneLoaded() { // upload replaces Scene::clone, GpuTracer.cu:172-175
    if (!ctx_) return false;
    if (opts_.device_bvh) { // slot vertices are kept as 3 float4 per triangle; the ABI takes 9 floats
        const std::vector<float> &v4 = scene->slotVertices();
        std::vector<float> v3(v4.size() / 12 * 9);
        for (size_t t = 0; t < v4.size() / 12; t++)
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) v3[9 * t + 3 * k + a] = v4[12 * t + 4 * k + a];
        mi_build_stats bs;
        status_ = mi_scene_upload_build(ctx_, &scene->soa(), v3.empty() ? nullptr : v3.data(), &bs);
    } else {
        status_ = mi_scene_upload(ctx_, &scene->soa());
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
        if (!ctx_) { status_ = MI_ERR_NO_DEVICE; return; }
        mi_render_params p{};
        p.width = w; p.height = h;
        p.samples_per_dim = (int)samples;
        p.depth = (int)depth;
        p.seed = opts_.seed;
        status_ = mi_render(ctx_, &p, nullptr, buffer.data(), &stats_);
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
