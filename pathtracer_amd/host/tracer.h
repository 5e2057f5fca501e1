// tracer.h — the Tracer interface of the reference (src/tracer.h:19-52) and the
// MI355X backend that plugs in behind it.
//
// Same call sequence as CommandLineUI::run() drives (src/ui/CommandLineUI.cu:82-94):
//   loadScene -> traceSetup -> traceImage -> waitRender -> getBuffer.
// traceImage may return before the image is done; waitRender is the completion
// point (CpuTracer spawns threads, src/CpuTracer.cu:57-97,188-191).
#pragma once
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi_tracer.h"
#include "scene.h"

namespace pt {

// The subset of TraceUI (src/ui/TraceUI.h:21-83) the tracers read
// (src/tracer.h:21-25,56-59,80-82).
struct TraceOptions {
    int depth = 0;          // -r, TraceUI.h:56 default 0
    int samples = 2;        // -s, per dimension, TraceUI.h:59 default 2
    int threads = 0;        // -t (CPU backends only)
    int device = 0;         // HIP device of the MI355X backend (the first one of a group)
    int n_devices = 0;      // > 0: devices device .. device + n_devices - 1 as one mi_group (frame sharded by tile, RGBA8
                            // tiles gathered over RCCL); 0: one context on `device`
    unsigned long long seed = 0;
    // Where the BVH is built: 1 (default) on the GPU (mi_scene_upload_build: top-down binned SAH, 2 ms for 82 k
    // primitives, 9 ms for 1.3 M), 0 on the host (Scene::finalize: the same algorithm and — node for node — the same
    // tree, at ~2.4 us per primitive: 0.2 s and 3.1 s).
    int device_bvh = 1;
};

class Tracer {
public:
    explicit Tracer(const TraceOptions &o) : depth(o.depth), samples(o.samples) {}
    virtual ~Tracer();

    double aspectRatio();                         // tracer.cu:38-40
    bool loadScene(const char *fn);               // tracer.cu:42-85 (errors -> stderr, false)
    bool loadSceneText(const std::string &text);  // same, from memory
    // TraceUI::smartLoadCubemap (TraceUI.cu:150-169): given ONE of the six face files, finds the
    // others in its directory by name ("pos"/"neg" followed by "x"/"y"/"z", .bmp/.png) and loads
    // them.  Errors go to stderr and leave the tracer without a cubemap, as in the reference.
    bool loadCubemap(const char *oneFaceFile);
    void traceSetup(int w, int h);                // tracer.cu:87-93
    virtual void traceImage(int w, int h) = 0;    // tracer.h:35
    virtual void waitRender() = 0;                // tracer.h:37
    void getBuffer(unsigned char *&buf, int &w, int &h); // tracer.cu:32-36
    Scene *getScene() { return scene; }

protected:
    // Storage of the RGB8 frame (the reference's `std::vector<uchar> buffer`, tracer.h:47): a backend may supply
    // memory of its own kind — HipTracer hands out pinned host memory (mi_host_alloc) so that a frame arrives by DMA.
    virtual unsigned char *frameAlloc(size_t bytes) { return new unsigned char[bytes]; }
    virtual void frameFree(unsigned char *p) { delete[] p; }
    void releaseFrame();                          // derived destructors call it while their frameFree is still reachable
    virtual bool wantsHostTree(size_t /*n_prims*/) { return true; } // false: the backend builds the tree itself
    virtual bool sceneLoaded() { return true; }
    virtual bool cubemapLoaded() { return true; }
    struct Face { int w = 0, h = 0; std::vector<unsigned char> rgb; };
    Face cube_[6];
    bool has_cube_ = false;
    unsigned depth;
    unsigned samples;
    Scene *scene = nullptr;
    struct Frame {                                // RGB8, row 0 = bottom row
        unsigned char *p = nullptr;
        size_t n = 0;
        unsigned char *data() const { return p; }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
    } buffer;
    int buffer_width = 0, buffer_height = 0;
};

// Third Tracer backend: hand-written HIP wavefront pipeline behind the C-ABI of
// include/mi_tracer.h.  Takes the place of GpuTracer (src/tracer.h:78-90,
// src/GpuTracer.cu:161-193).  Fails loudly (loadScene returns false / traceImage
// reports on stderr and leaves the buffer zeroed) when no HIP device is present;
// there is no CPU fallback.
class HipTracer : public Tracer {
public:
    explicit HipTracer(const TraceOptions &o);
    ~HipTracer() override;
    void traceImage(int w, int h) override;
    void waitRender() override;
    const mi_stats &stats() const { return stats_; }
    bool ok() const { return status_ == MI_OK; }
    std::string error() const { return error_; }

protected:
    bool wantsHostTree(size_t n_prims) override;
    bool sceneLoaded() override;
    bool cubemapLoaded() override;
    unsigned char *frameAlloc(size_t bytes) override;   // pinned host memory; ordinary memory if the runtime has none
    void frameFree(unsigned char *p) override;

private:
    TraceOptions opts_;
    mi_ctx *ctx_ = nullptr;
    mi_group *group_ = nullptr;   // instead of ctx_ when opts_.n_devices > 0
    std::thread worker_;
    mi_stats stats_{};
    mi_status status_ = MI_OK;
    std::string error_;
    bool device_tree_ = false;                    // the resident scene's tree was built on the GPU
    unsigned char *pinned_ = nullptr;             // the frame, when it came from mi_host_alloc
};

// Image output (counterpart of src/fileio/images.cu:59-68): ".png" -> PNG RGB8,
// anything else -> BMP; rows are flipped so that buffer row 0 is the bottom row
// (src/fileio/pngimage.cu:258-268, src/fileio/bitmap.cu:95-149).
bool writeImage(const char *path, int w, int h, const unsigned char *rgb);
// Counterpart of readImage (src/fileio/images.cu:51-57): ".bmp" (24 bit) or ".png" (8 bit) ->
// RGB8, row 0 = bottom row.  false on anything else.
bool readImage(const char *path, int &w, int &h, std::vector<unsigned char> &rgb);
// TraceUI::matchCubemapFiles (TraceUI.cu:86-148).
bool matchCubemapFiles(const std::string &one, std::string matched[6], std::string &dir);

} // namespace pt
