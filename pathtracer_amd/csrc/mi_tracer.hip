// mi_tracer.hip — C-ABI of include/mi_tracer.h: context, scene upload, wavefront
// render loop.  Every hip* call is checked; errors are returned as status codes
// with a thread-local message (the reference checks nothing, GpuTracer.cu:161-189).
#include "mi_kernels.h"
#include "mi_bvh_build.h"
#ifdef MI_WIDE8
#include "mi_wide8.h"
#endif

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi_tracer.h"

#define MI_API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_err;
static mi_status fail(mi_status st, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}
#define HIPCHK(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                        \
    } while (0)

// ---- profiler stage ranges (the reference brackets its render with cudaProfilerStart/Stop, GpuTracer.cu:162,183) ----
// With MI_TRACER_ROCTX (or MI_TRACER_DEBUG_LEVELS) set, every frame, batch and level of the wavefront loop is a
// roctx range ("mi_render" > "batch b" > "level k"), so `rocprofv3 --marker-trace --kernel-trace` attributes the
// launches (k_tail | k_extend | k_shade per level) to their batch and bounce level.  The library is loaded on first
// use (librocprofiler-sdk-roctx, else the older libroctx64); without it, or without the variable, nothing happens.
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
};
Roctx g_roctx;
std::mutex g_roctx_mutex;
bool roctx_on() {
    if (!getenv("MI_TRACER_ROCTX") && !getenv("MI_TRACER_DEBUG_LEVELS")) return false;
    std::lock_guard<std::mutex> lock(g_roctx_mutex);
    if (!g_roctx.tried) {
        g_roctx.tried = true;
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            g_roctx.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop");
            if (g_roctx.push && g_roctx.pop) break;
            g_roctx.push = nullptr; g_roctx.pop = nullptr;
            dlclose(h);
        }
    }
    return g_roctx.push != nullptr;
}
struct RoctxRange { // RAII: a range around a scope, when ranges are on
    bool on;
    RoctxRange(bool enabled, const char *fmt, long long a = 0) : on(enabled) {
        if (!on) return;
        char name[64];
        snprintf(name, sizeof name, fmt, a);
        g_roctx.push(name);
    }
    ~RoctxRange() { if (on) g_roctx.pop(); }
};
} // namespace

// ---- workspace of the entry point that allocates nothing (mi_render_device_noalloc) ---------------------
// All per-batch path state (ray streams, hits, history, path values, counters, overflow stacks, totals) in ONE
// file-scope device array, here called `scratch`: MI_NOALLOC_BYTES on every GPU this library's code object is
// loaded on, until the process ends.  In host code the array's NAME is only a stand-in: the address the kernels
// need is asked for per device (hipGetSymbolAddress, noalloc_workspace() below) and passed to them as an
// argument, like the grown buffers of mi_render_device.
#define MI_NOALLOC_BYTES ((size_t)64 << 20)
static __device__ __align__(256) unsigned char scratch[MI_NOALLOC_BYTES];
static std::mutex g_ws_mutex;
static unsigned char *g_ws_ptr[64]; // per device, filled on the first mi_ctx_create on that device

// Called with `device` current (mi_ctx_create).  Each GPU holds its own copy of the array at its own address.
static mi_status noalloc_workspace(int device, unsigned char **out) {
    if (device < 0 || device >= 64) return fail(MI_ERR_NO_DEVICE, "device %d beyond the workspace table", device);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (!g_ws_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(MI_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
        g_ws_ptr[device] = (unsigned char *)ptr;
    }
    *out = g_ws_ptr[device];
    return MI_OK;
}

namespace {

// Grow-only device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T *as() const { return (T *)p; }
};

} // namespace

struct mi_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr, stream2 = nullptr; // stream2: the second of the two streams the batches of a frame alternate between
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int n_streams = 2;          // MI_TRACER_STREAMS: 1 = every batch on the caller's stream
    int64_t split_min = 0;      // MI_TRACER_SPLIT_MIN: path slots each half must have for a one-batch frame to be cut in two (default 4 x the tail threshold)
    bool has_scene = false;
    DevScene sc{};
    uint32_t n_nodes = 0;   // of the binary tree kept in `nodes` (download, quality); the kernels walk `nodes4`
    uint32_t n_nodes8 = 0;  // -DMI_WIDE8: nodes of the 8-wide tree in `qnodes`, and its levels
    int levels8 = 0;
    int32_t root_ref2 = ~0;
    DevBuf nodes, nodes4, qnodes, tri, tri_normal, tri_mat, tri_has_vmat, tri_shade, sph, sph_mat, mats, prim_id, cube;
    // per-batch path state (sized by slot capacity / depth): two ray streams (level k reads
    // stream k&1 and writes stream (k+1)&1), hits by stream position, history and path value by slot
    // Two sets: the batches of a frame alternate between two streams (`streams` below), each with its own state.
    struct PathBufs { DevBuf ray_o[2], ray_d[2], thr[2], hits, path_out, hist, counters, ovf, totals; } pb[2];
    // outputs for the host-pointer entry points
    DevBuf out_mean, out_rgba8, out_rgb8, out_mean_frame, probe_i, probe_t, probe_n;
    DevBuf cam_tab;                      // camera-plane tables of the last (width, height, samples) rendered (ShardGeom::cam_x / cam_y)
    int cam_w = 0, cam_h = 0, cam_s = 0;
    std::vector<hipEvent_t> ev;
    uint64_t *h_totals = nullptr; // pinned
    unsigned char *ws = nullptr;  // this device's copy of the static workspace of mi_render_device_noalloc
    int max_stack = MI_MAX_STACK; // traversal stack entries a ray can need in the resident tree (3 per level of the 4-wide tree)
    // tuning knobs (defaults measured on MI355X, overridable through the environment)
    int extend_wg_per_cu = 7;   // MI_TRACER_EXTEND_WG_PER_CU: persistent workgroups of k_extend per CU (clamped by the occupancy query)
    int extend_wg_primary = 6;  // the same for the level-0 variant, which holds more registers
    int extend_wg_two = 6;      // MI_TRACER_EXTEND_WG_TWO_STREAMS: workgroups of the secondary k_extend per CU when the batches run on two
                                // streams — one resident slot left to the other stream's kernels: C3 +3 %, C4 +2 % against 7; 5 loses
    int refill_min = 20;        // MI_TRACER_REFILL: idle lanes that trigger a refill
    int refill_primary = 12;    // MI_TRACER_REFILL_PRIMARY: the same at level 0, where a refill only reads parked camera rays
    int leaf_min = 16;          // MI_TRACER_LEAF_MIN: lanes with a pending leaf that trigger a leaf phase
    int tail_refill = 16;       // MI_TRACER_TAIL_REFILL: idle lanes that trigger a refill in k_tail
    int64_t batch_budget = (int64_t)32 << 30; // bytes of path state per batch and stream: min(total / 4, 2/5 of the free) of the device's memory at mi_ctx_create (MI_TRACER_BATCH_GB)
    int shade_min = 24;         // MI_TRACER_SHADE_MIN: lanes waiting to be shaded that trigger a shade phase of k_tail
    int shade_wg_per_cu = 8;    // MI_TRACER_SHADE_WG_PER_CU: workgroups of k_shade per CU (grid-stride loop)
    int tail_wg_per_cu = 4;     // workgroups of k_tail resident per CU (occupancy query at context creation)
    bool tail_auto = true;        // no MI_TRACER_TAIL given: small batches hand over later, at 1/40 of their paths (at least 200 k rays; a 1/4 shard of C3: 2.29 -> 2.13 ms, 1/8 and the whole frame unchanged)
    int tail_threshold = 700000;  // MI_TRACER_TAIL: k_tail takes over a batch once a level has at most this many rays (0 = off); with the batches on two streams 0.5-1 M measure alike on C3, 1.5 M 2 % slower (8 % on C5)
};

static const int kMaxLevels = 64; // counters reserved per batch (depth + 2 must fit)
// totals[]: 0 rays, 1 paths, 2.. = work_stats of the instrumented extend kernel
// (+0 node visits, +1 primitive tests, +2 max steps of a ray, +3/+4 wave iterations
// total/max, +5 rays k_tail traced beyond its first level, +8..+47 visits per tree depth)
static const int kTotals = 64;
// counters: cnt[level][MI_SUB] ray counts of the sub-streams of each level, then the tail_level word
static const int kLevelWords = MI_SUB * MI_CNT_STRIDE; // counter j of a level at word j * MI_CNT_STRIDE
static const int kCounterWords = (kMaxLevels + 1) * kLevelWords + MI_CNT_STRIDE;

// Stack entries a ray can need in a tree of `depth2` levels of binary interior nodes: the 4-wide tree has
// ceil(depth2 / 2) levels and a node visit pushes at most three refs (+ 3 of margin).
#ifdef MI_WIDE8
static int stack_need(int depth2) { return std::min(MI_MAX_STACK, (depth2 + 2) / 3 + 2); } // one group entry per level of the 8-wide tree
#else
static int stack_need(int depth2) { return std::min(MI_MAX_STACK, 3 * ((depth2 + 1) / 2) + 3); }
#endif
static int extend_grid(const mi_ctx *c) { return c->n_cu * c->extend_wg_per_cu; }
// Workgroups of the largest persistent launch (k_extend, its level-0 variant, k_tail): every one of their
// threads owns a column of the global overflow stack, indexed by blockIdx.x * MI_BLOCK + threadIdx.x.
static int stack_grid(const mi_ctx *c) { return c->n_cu * std::max(c->extend_wg_per_cu, std::max(c->extend_wg_primary, c->tail_wg_per_cu)); }
static size_t stack_bytes(const mi_ctx *c) { return (size_t)(MI_MAX_STACK - MI_LDS_STACK) * stack_grid(c) * MI_BLOCK * MI_STACK_ELEM_BYTES; }

MI_API int mi_abi_version(void) { return 13; }
MI_API const char *mi_last_error(void) { return g_err.c_str(); }

// Pinned host memory for the caller's frame buffer (hipHostMalloc: DMA-able from every device).  The library
// hands it out instead of pinning the caller's own pages: hipHostRegister is refused on some deployments
// (containers without user-pointer pinning; the MI355X pool this was developed on is one).
MI_API mi_status mi_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(MI_ERR_INVALID, "mi_host_alloc: out is NULL");
    *out = nullptr;
    if (bytes == 0) return fail(MI_ERR_INVALID, "mi_host_alloc: zero bytes");
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return MI_OK;
}
MI_API void mi_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

MI_API mi_status mi_ctx_create(int device_id, mi_ctx **out) {
    if (!out) return fail(MI_ERR_INVALID, "mi_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MI_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(MI_ERR_NO_DEVICE, "device %d out of range (0..%d)", device_id, n - 1);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    mi_ctx *c = new mi_ctx;
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto env_int = [](const char *name, int lo, int hi, int dflt) {
        const char *v = getenv(name);
        if (!v) return dflt;
        int x = atoi(v);
        return x < lo ? lo : (x > hi ? hi : x);
    };
    c->extend_wg_per_cu = env_int("MI_TRACER_EXTEND_WG_PER_CU", 1, 16, c->extend_wg_per_cu);
    c->refill_min = env_int("MI_TRACER_REFILL", 1, 64, c->refill_min);
    c->refill_primary = env_int("MI_TRACER_REFILL_PRIMARY", 1, 64, c->refill_primary);
    c->leaf_min = env_int("MI_TRACER_LEAF_MIN", 1, 64, c->leaf_min);
    c->shade_min = env_int("MI_TRACER_SHADE_MIN", 1, 64, c->shade_min);
    c->tail_refill = env_int("MI_TRACER_TAIL_REFILL", 1, 64, c->tail_refill);
    {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_total > 0)
            c->batch_budget = std::max<int64_t>((int64_t)1 << 30, (int64_t)std::min(mem_total / 4, mem_free * 2 / 5));
        const int gb = env_int("MI_TRACER_BATCH_GB", 0, 1 << 16, 0);
        if (gb > 0) c->batch_budget = (int64_t)gb << 30;
    }
    c->tail_auto = getenv("MI_TRACER_TAIL") == nullptr;
    c->tail_threshold = env_int("MI_TRACER_TAIL", 0, 1 << 30, c->tail_threshold);
    // k_shade walks its stream with a grid-stride loop: a grid that is a multiple of what is resident
    // at once keeps the last round full (8 per CU against 5 resident measured 9 % slower than 5 or 10)
    int occ_s = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, k_shade<false>, MI_BLOCK, 0) == hipSuccess && occ_s > 0)
        c->shade_wg_per_cu = 2 * occ_s;
    c->shade_wg_per_cu = env_int("MI_TRACER_SHADE_WG_PER_CU", 1, 32, c->shade_wg_per_cu);
    // the same for k_extend: never more persistent workgroups than are resident at once (LDS: 12-KB stack +
    // treelet allow 7; the level-0 variant's registers allow 6)
    int occ_e = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_e, k_extend<false, false, false>, MI_BLOCK, 0) == hipSuccess && occ_e > 0)
        c->extend_wg_per_cu = std::min(c->extend_wg_per_cu, occ_e);
    c->extend_wg_primary = std::min(c->extend_wg_per_cu, 6); // measured: level 0 of C3 1.77 ms with 6 workgroups per CU, 1.87 with 5 or 7
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_e, k_extend<false, true, false>, MI_BLOCK, 0) == hipSuccess && occ_e > 0)
        c->extend_wg_primary = std::min(c->extend_wg_primary, occ_e);
    c->extend_wg_primary = env_int("MI_TRACER_EXTEND_WG_PRIMARY", 1, 16, c->extend_wg_primary); // stack_grid() covers whatever is chosen
    c->extend_wg_two = std::min(c->extend_wg_per_cu, env_int("MI_TRACER_EXTEND_WG_TWO_STREAMS", 1, 16, c->extend_wg_two));
    // k_tail is persistent and deals its work statically over the grid: launch exactly as many
    // workgroups as are resident at once (a second round would start after the first has finished).
    // Computed after extend_wg_per_cu is final; the overflow stacks are sized for the largest of the
    // persistent grids (stack_grid), so an override above k_extend's count stays in bounds.
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_tail, MI_BLOCK, 0) == hipSuccess && occ > 0)
        c->tail_wg_per_cu = std::min(occ, c->extend_wg_per_cu);
    c->tail_wg_per_cu = env_int("MI_TRACER_TAIL_WG_PER_CU", 1, 16, c->tail_wg_per_cu);
    {   // this device's address of the static workspace, and the two events that bracket a render: nothing is
        // left to create or allocate when mi_render_device_noalloc runs
        mi_status wsr = noalloc_workspace(device_id, &c->ws);
        if (wsr != MI_OK) { delete c; return wsr; }
        for (int i = 0; i < 2; i++) {
            hipEvent_t ev;
            hipError_t ee = hipEventCreate(&ev);
            if (ee != hipSuccess) { for (hipEvent_t x : c->ev) (void)hipEventDestroy(x); delete c; return fail(MI_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(ee)); }
            c->ev.push_back(ev);
        }
    }
    c->n_streams = env_int("MI_TRACER_STREAMS", 1, 2, c->n_streams);
    c->split_min = env_int("MI_TRACER_SPLIT_MIN", 64, 1 << 30, 4 * std::max(c->tail_threshold, 250000));
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        for (hipEvent_t x : c->ev) (void)hipEventDestroy(x);
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->ev_join) (void)hipEventDestroy(c->ev_join);
        if (c->stream2) (void)hipStreamDestroy(c->stream2);
        delete c;
        return fail(MI_ERR_HIP, "creating the second stream failed");
    }
    hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    auto drop_second = [&]() { (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); (void)hipStreamDestroy(c->stream2); };
    if (es != hipSuccess) { for (hipEvent_t x : c->ev) (void)hipEventDestroy(x); drop_second(); delete c; return fail(MI_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(es)); }
    es = hipHostMalloc((void **)&c->h_totals, 2 * kTotals * sizeof(uint64_t), hipHostMallocDefault);
    if (es != hipSuccess) { for (hipEvent_t x : c->ev) (void)hipEventDestroy(x); drop_second(); (void)hipStreamDestroy(c->stream); delete c; return fail(MI_ERR_HIP, "hipHostMalloc failed: %s", hipGetErrorString(es)); }
    *out = c;
    return MI_OK;
}

MI_API void mi_ctx_destroy(mi_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // Everything that may still read this context's buffers must be done before they are released: its own two
    // streams, and the caller's stream of an enqueue-only mi_render_device_noalloc (include/mi_tracer.h: lifetime).
    (void)hipDeviceSynchronize();
    DevBuf *all[] = {&c->nodes, &c->nodes4, &c->qnodes, &c->tri, &c->tri_normal, &c->tri_mat, &c->tri_has_vmat, &c->tri_shade, &c->sph, &c->sph_mat, &c->mats,
                     &c->prim_id, &c->cube, &c->out_mean, &c->out_rgba8,
                     &c->out_rgb8, &c->out_mean_frame, &c->probe_i, &c->probe_t, &c->probe_n, &c->cam_tab};
    for (auto &pb : c->pb) {
        DevBuf *set[] = {&pb.ray_o[0], &pb.ray_o[1], &pb.ray_d[0], &pb.ray_d[1], &pb.thr[0], &pb.thr[1], &pb.hits, &pb.path_out, &pb.hist, &pb.counters, &pb.ovf, &pb.totals};
        for (DevBuf *b : set) b->release();
    }
    for (DevBuf *b : all) b->release();
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->stream2);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    delete c;
}

// Host-side validation of the flattened BVH before any kernel may walk it: refs in
// range, every primitive slot referenced exactly once, depth <= MI_MAX_STACK.
static mi_status validate_prims(const mi_scene_soa *s) {
    const uint64_t n_prim = (uint64_t)s->n_tri + s->n_sph;
    for (int k = 0; k < 3; k++) // the slab test's products stay finite for coordinates below 1e15 (lane_start)
        if (!(s->cam_eye[k] > -1e15f && s->cam_eye[k] < 1e15f)) return fail(MI_ERR_INVALID, "scene: camera position not finite or huge");
    if (n_prim == 0) return MI_OK;
    if (!s->prim_id || !s->materials || s->n_mat == 0) return fail(MI_ERR_INVALID, "scene: prim_id/materials missing");
    if (s->n_tri && !(s->tri_v0 && s->tri_uinv && s->tri_vinv && s->tri_ninv && s->tri_normal && s->tri_mat && s->tri_has_vmat))
        return fail(MI_ERR_INVALID, "scene: triangle arrays missing");
    if (s->n_sph && !(s->sph_center_radius && s->sph_mat)) return fail(MI_ERR_INVALID, "scene: sphere arrays missing");
    for (uint64_t i = 0; i < 3ull * s->n_tri; i++)
        if (s->tri_mat[i] >= s->n_mat) return fail(MI_ERR_INVALID, "scene: triangle material id out of range");
    for (uint32_t i = 0; i < s->n_sph; i++)
        if (s->sph_mat[i] >= s->n_mat) return fail(MI_ERR_INVALID, "scene: sphere material id out of range");
    // node and triangle records are addressed with 32-bit byte offsets (64 B each)
    if (n_prim >= (1u << 26)) return fail(MI_ERR_INVALID, "scene: more than 2^26 primitives");
    return MI_OK;
}

static mi_status validate_scene(const mi_scene_soa *s, int *depth_out) {
    *depth_out = 0;
    const uint64_t n_prim = (uint64_t)s->n_tri + s->n_sph;
    mi_status ps = validate_prims(s);
    if (ps != MI_OK) return ps;
    if (n_prim == 0) return s->n_nodes == 0 ? MI_OK : fail(MI_ERR_INVALID, "scene: no primitives, n_nodes must be 0");
    if (s->n_nodes >= (1u << 26)) return fail(MI_ERR_INVALID, "scene: more than 2^26 nodes");
    if (n_prim == 1) {
        if (s->root_ref != ~0) return fail(MI_ERR_INVALID, "scene: one primitive needs root_ref == ~0");
        if (s->n_nodes != 0) return fail(MI_ERR_INVALID, "scene: one primitive has no interior node, n_nodes must be 0");
        return MI_OK;
    }
    if (s->n_nodes != n_prim - 1 || !s->nodes) return fail(MI_ERR_INVALID, "scene: %u nodes for %llu primitives (want n-1)", s->n_nodes, (unsigned long long)n_prim);
    if (s->root_ref < 0 || (uint32_t)s->root_ref >= s->n_nodes) return fail(MI_ERR_INVALID, "scene: bad root_ref");
    for (size_t i = 0; i < (size_t)s->n_nodes; i++)
        for (int k = 0; k < 12; k++) {
            float v = s->nodes[16 * i + k];
            if (!(v > -1e15f && v < 1e15f)) return fail(MI_ERR_INVALID, "scene: node %zu has a non-finite or huge box", i);
        }
    std::vector<uint8_t> seen_prim(n_prim, 0), seen_node(s->n_nodes, 0);
    std::vector<std::pair<int32_t, int>> st;
    st.push_back({s->root_ref, 1});
    while (!st.empty()) {
        auto [ref, depth] = st.back();
        st.pop_back();
        if (depth > MI_MAX_DEPTH) return fail(MI_ERR_INVALID, "scene: BVH deeper than %d", MI_MAX_DEPTH);
        *depth_out = std::max(*depth_out, depth);
        if (seen_node[ref]++) return fail(MI_ERR_INVALID, "scene: node %d referenced twice", ref);
        for (int c = 0; c < 2; c++) {
            int32_t ch;
            memcpy(&ch, s->nodes + 16 * (size_t)ref + 12 + c, 4);
            if (ch < 0) {
                uint32_t slot = (uint32_t)~ch;
                if (slot >= n_prim) return fail(MI_ERR_INVALID, "scene: leaf slot out of range");
                if (seen_prim[slot]++) return fail(MI_ERR_INVALID, "scene: primitive slot %u referenced twice", slot);
            } else {
                if ((uint32_t)ch >= s->n_nodes) return fail(MI_ERR_INVALID, "scene: child node out of range");
                st.push_back({ch, depth + 1});
            }
        }
    }
    return MI_OK;
}

static mi_status get_events(mi_ctx *c, size_t n) {
    while (c->ev.size() < n) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    return MI_OK;
}

// The kernels walk a 4-wide tree collapsed from the caller's binary one: a node's seats are
// its grandchildren (the children of each interior child) or the child itself when that is a
// leaf.  This intermediate fp32 form is 128 B per node: lo.x[4], lo.y[4], lo.z[4], hi.x[4], hi.y[4],
// hi.z[4], refs[4], pad; quantize_nodes4 turns it into the 64-B nodes the kernels read.
// Refs: >= 0 index of a 4-wide node, < 0 leaf ~slot, MI_DONE empty seat.  `nodes2` has been
// validated (validate_scene) or was built by this library.  Returns the root ref.
static int32_t collapse_bvh4(const float *nodes2, int32_t root2, std::vector<float> &out) {
    out.clear();
    if (root2 < 0) return root2; // a single primitive: no node at all
    struct Todo { int32_t n2; size_t n4; };
    std::vector<Todo> todo;
    auto new_node = [&](int32_t n2) {
        size_t idx = out.size() / 32;
        out.resize(out.size() + 32, 0.0f);
        todo.push_back({n2, idx});
        return (int32_t)idx;
    };
    const int32_t root4 = new_node(root2);
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        float box[4][6];
        int32_t ref[4];
        int n = 0;
        auto child_of = [&](const float *nd, int c, float *b, int32_t &r) {
            memcpy(b, nd + 6 * c, 24);
            memcpy(&r, nd + 12 + c, 4);
        };
        const float *nd = nodes2 + 16 * (size_t)t.n2;
        for (int c = 0; c < 2; c++) {
            float b[6];
            int32_t r;
            child_of(nd, c, b, r);
            if (r < 0) {
                memcpy(box[n], b, 24);
                ref[n++] = r;
            } else {
                const float *ch = nodes2 + 16 * (size_t)r;
                for (int g = 0; g < 2; g++) child_of(ch, g, box[n], ref[n]), n++;
            }
        }
        int32_t out_ref[4];
        for (int k = 0; k < 4; k++) out_ref[k] = MI_DONE;
        for (int k = 0; k < n; k++) out_ref[k] = ref[k] < 0 ? ref[k] : new_node(ref[k]); // may reallocate `out`
        float *o = out.data() + 32 * t.n4;
        for (int k = 0; k < 4; k++) {
            for (int a = 0; a < 3; a++) {
                o[4 * a + k] = k < n ? box[k][a] : 0.0f;
                o[12 + 4 * a + k] = k < n ? box[k][3 + a] : 0.0f;
            }
            memcpy(o + 24 + k, &out_ref[k], 4);
        }
    }
    return root4;
}

// The same collapse with the seats chosen by dynamic programming instead of the fixed "grandchildren" rule (SAH-optimal
// widening, Ylitie, Karras, Laine 2017, section 3.1, with the kernels' cost model: a visit costs the same however many of
// its four seats are filled, so the cost of a tree is the sum of its 4-wide nodes' surface areas).  C[n][i] = least cost of
// subtree n when it may occupy i + 1 seats of its parent's node: with one seat it must be a 4-wide node of its own
// (area + best split of FOUR seats among its two children), with more it may hand the seats on to its children.
static int32_t collapse_bvh4_dp(const float *nodes2, int32_t root2, uint32_t n_nodes2, std::vector<float> &out) {
    out.clear();
    if (root2 < 0) return root2;
    const int W = 4;
    auto ref_of = [&](int32_t n, int c) { int32_t r; memcpy(&r, nodes2 + 16 * (size_t)n + 12 + c, 4); return r; };
    auto area_of = [&](int32_t n) {
        const float *nd = nodes2 + 16 * (size_t)n;
        float e[3];
        for (int a = 0; a < 3; a++) e[a] = std::max(nd[3 + a], nd[9 + a]) - std::min(nd[a], nd[6 + a]);
        return (double)e[0] * e[1] + (double)e[1] * e[2] + (double)e[2] * e[0];
    };
    std::vector<double> C((size_t)n_nodes2 * (W - 1), 0.0);
    std::vector<uint8_t> K((size_t)n_nodes2 * W, 0); // K[n][j]: seats the left child gets when n distributes j seats (j = 2..W)
    std::vector<int32_t> order, st{root2};
    order.reserve(n_nodes2);
    while (!st.empty()) {
        const int32_t n = st.back(); st.pop_back();
        order.push_back(n);
        for (int c = 0; c < 2; c++) if (ref_of(n, c) >= 0) st.push_back(ref_of(n, c));
    }
    auto cost = [&](int32_t ref, int seats) { return ref < 0 ? 0.0 : C[(size_t)ref * (W - 1) + (size_t)(std::min(seats, W - 1) - 1)]; };
    for (size_t q = order.size(); q-- > 0;) { // children before parents
        const int32_t n = order[q], l = ref_of(n, 0), r = ref_of(n, 1);
        double dist[W + 1];
        for (int j = 2; j <= W; j++) {
            double best = INFINITY; int bk = 1;
            for (int k = 1; k < j; k++) { const double v = cost(l, k) + cost(r, j - k); if (v < best) { best = v; bk = k; } }
            dist[j] = best; K[(size_t)n * W + (size_t)(j - 1)] = (uint8_t)bk;
        }
        double *Cn = &C[(size_t)n * (W - 1)];
        Cn[0] = dist[W] + area_of(n);
        for (int i = 2; i <= W - 1; i++) Cn[i - 1] = std::min(dist[i], Cn[i - 2]);
    }
    // emit: a 4-wide node per binary node that ends up with one seat
    struct Todo { int32_t n2; size_t n4; };
    std::vector<Todo> todo;
    auto new_node = [&](int32_t n2) {
        size_t idx = out.size() / 32;
        out.resize(out.size() + 32, 0.0f);
        todo.push_back({n2, idx});
        return (int32_t)idx;
    };
    const int32_t root4 = new_node(root2);
    float box[4][6]; int32_t ref[4]; int n;
    // seats of child c of n2 when it is given `seats` seats
    std::vector<std::pair<std::pair<int32_t, int>, int>> work;
    while (!todo.empty()) {
        const Todo t = todo.back(); todo.pop_back();
        n = 0;
        work.clear();
        const int kroot = K[(size_t)t.n2 * W + (W - 1)];
        work.push_back({{t.n2, 1}, W - kroot});
        work.push_back({{t.n2, 0}, kroot});
        while (!work.empty()) {
            const auto w = work.back(); work.pop_back();
            const int32_t p = w.first.first; const int c = w.first.second; const int seats = w.second;
            const float *nd = nodes2 + 16 * (size_t)p;
            const int32_t r = ref_of(p, c);
            bool keep = r < 0 || seats == 1;
            int i = seats;
            if (!keep) { // which i <= seats defined C[r][seats - 1]: the smallest with the same cost
                const double *Cr = &C[(size_t)r * (W - 1)];
                i = std::min(seats, W - 1);
                while (i > 1 && Cr[i - 1] == Cr[i - 2]) i--;
                keep = i == 1;
            }
            if (keep) { memcpy(box[n], nd + 6 * c, 24); ref[n++] = r; continue; }
            const int k = K[(size_t)r * W + (size_t)(i - 1)];
            work.push_back({{r, 1}, i - k});
            work.push_back({{r, 0}, k});
        }
        int32_t out_ref[4];
        for (int k = 0; k < 4; k++) out_ref[k] = MI_DONE;
        for (int k = 0; k < n; k++) out_ref[k] = ref[k] < 0 ? ref[k] : new_node(ref[k]);
        float *o = out.data() + 32 * t.n4;
        for (int k = 0; k < 4; k++) {
            for (int a = 0; a < 3; a++) {
                o[4 * a + k] = k < n ? box[k][a] : 0.0f;
                o[12 + 4 * a + k] = k < n ? box[k][3 + a] : 0.0f;
            }
            memcpy(o + 24 + k, &out_ref[k], 4);
        }
    }
    return root4;
}

// Copies primitives, materials, ids and camera (everything but the nodes) to the device.
static mi_status upload_prims(mi_ctx *c, const mi_scene_soa *s) {
    const size_t nt = s->n_tri, ns = s->n_sph;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(std::max<size_t>(bytes, 16));
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    };
    // v0, uInv, vInv, nInv (xyz of each: the fourth lanes are unused) into one 48-B record per triangle, + 16 B
    // behind the last: merged_step's node-shaped fetch never reads them, a record's three loads stay inside
    std::vector<float> rec(12 * nt + 4, 0.0f);
    for (size_t i = 0; i < nt; i++) {
        memcpy(&rec[12 * i + 0], s->tri_v0 + 4 * i, 12);
        memcpy(&rec[12 * i + 3], s->tri_uinv + 4 * i, 12);
        memcpy(&rec[12 * i + 6], s->tri_vinv + 4 * i, 12);
        memcpy(&rec[12 * i + 9], s->tri_ninv + 4 * i, 12);
    }
    HIPCHK(up(c->tri, rec.data(), nt * 48 + 16));
    HIPCHK(up(c->tri_normal, s->tri_normal, nt * 16));
    HIPCHK(up(c->tri_mat, s->tri_mat, nt * 12));
    HIPCHK(up(c->tri_has_vmat, s->tri_has_vmat, nt));
    // + 64 B: merged_step reads 64 B at a sphere's record like at a triangle's or a node's and uses the first 16
    HIPCHK(c->sph.ensure(ns * 16 + 64));
    HIPCHK(up(c->sph, s->sph_center_radius, ns * 16));
    HIPCHK(up(c->sph_mat, s->sph_mat, ns * 4));
    HIPCHK(up(c->mats, s->materials, (size_t)s->n_mat * 80));
    HIPCHK(up(c->prim_id, s->prim_id, (nt + ns) * 4));
    DevScene &d = c->sc;
    d.tri = c->tri.as<float4>();
    d.tri_normal = c->tri_normal.as<float4>();
    d.tri_mat = c->tri_mat.as<uint32_t>();
    d.tri_has_vmat = c->tri_has_vmat.as<uint8_t>();
    d.sph = c->sph.as<float4>();
    d.sph_mat = c->sph_mat.as<uint32_t>();
    d.mats = c->mats.as<float4>();
    d.prim_id = c->prim_id.as<int32_t>();
    d.n_tri = s->n_tri;
    d.n_sph = s->n_sph;
    d.n_mat = s->n_mat;
    d.eye = make_float4(s->cam_eye[0], s->cam_eye[1], s->cam_eye[2], s->cam_eye[3]);
    d.u = make_float4(s->cam_u[0], s->cam_u[1], s->cam_u[2], s->cam_u[3]);
    d.v = make_float4(s->cam_v[0], s->cam_v[1], s->cam_v[2], s->cam_v[3]);
    d.look = make_float4(s->cam_look[0], s->cam_look[1], s->cam_look[2], s->cam_look[3]);
    return MI_OK;
}

// The 32-B shading records (DevScene::tri_shade) from the per-triangle arrays as they stand on the device — after
// the device builder's permutation when there was one.
static mi_status pack_shade(mi_ctx *c) {
    const uint32_t nt = c->sc.n_tri;
    HIPCHK(c->tri_shade.ensure(std::max<size_t>((size_t)nt * 32, 16)));
    if (nt)
        hipLaunchKernelGGL(k_pack_shade, dim3((nt + MI_BLOCK - 1) / MI_BLOCK), dim3(MI_BLOCK), 0, c->stream, nt, c->tri_normal.as<float4>(),
                           c->tri_mat.as<uint32_t>(), c->tri_has_vmat.as<uint8_t>(), c->tri_shade.as<uint4>());
    HIPCHK(hipGetLastError());
    c->sc.tri_shade = c->tri_shade.as<uint4>();
    return MI_OK; // quantize_nodes4 synchronises the stream
}

// Last step of every upload: the fp32 4-wide nodes in c->nodes4 (the tree and the treelet copies behind
// it, n_total nodes, root c->sc.root_ref) -> the 64-B quantised nodes the kernels read.  The code grid
// spans the root's box (the union of its seats): F = 0.51 at its lower face, 1.99 at its upper face.
static mi_status quantize_nodes4(mi_ctx *c, uint32_t n_total) {
    mi_status pk = pack_shade(c);
    if (pk != MI_OK) return pk;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->sc.qnodes = nullptr;
    for (int a = 0; a < 3; a++) { c->sc.qs[a] = 1.0f; c->sc.qb[a] = 0.0f; }
    const int32_t root = c->sc.root_ref;
    if (n_total == 0 || root < 0 || root == MI_DONE || (uint32_t)root >= n_total) return MI_OK; // no interior node: nothing walks qnodes
    float rn[32];
    HIPCHK(hipMemcpy(rn, c->nodes4.as<float>() + 32 * (size_t)root, 128, hipMemcpyDeviceToHost));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 4; k++) {
        int32_t r;
        memcpy(&r, &rn[24 + k], 4);
        if (r == MI_DONE) continue;
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], rn[4 * a + k]); hi[a] = std::max(hi[a], rn[12 + 4 * a + k]); }
    }
    for (int a = 0; a < 3; a++) {
        float ext = hi[a] - lo[a];
        if (!(ext > 0.0f) || !std::isfinite(ext)) { ext = 1.0f; if (!std::isfinite(lo[a])) lo[a] = 0.0f; }
        c->sc.qs[a] = ext / 1.48f;
        c->sc.qb[a] = lo[a] - 0.51f * c->sc.qs[a];
    }
    HIPCHK(c->qnodes.ensure((size_t)n_total * 64));
    hipLaunchKernelGGL(k_quantize_nodes4, dim3((n_total + 255) / 256), dim3(256), 0, c->stream, c->nodes4.as<float>(), (int)n_total, c->sc.qs[0],
                       c->sc.qs[1], c->sc.qs[2], c->sc.qb[0], c->sc.qb[1], c->sc.qb[2], (int)MI_DONE, c->qnodes.as<uint4>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    c->sc.qnodes = c->qnodes.as<uint4>();
    return MI_OK;
}

// Collapses the binary tree (host copy) into the 4-wide tree the kernels walk and uploads it.
static mi_status install_bvh4(mi_ctx *c, const float *nodes2, int32_t root2) {
    std::vector<float> n4;
    const char *how = getenv("MI_TRACER_COLLAPSE");
    int32_t root4 = (how && !strcmp(how, "dp")) ? collapse_bvh4_dp(nodes2, root2, c->n_nodes, n4) : collapse_bvh4(nodes2, root2, n4);
    // treelet: append copies of the root, its children and grandchildren (breadth first); refs
    // among them point at the copies, the root ref at the root's copy
    c->sc.treelet_base = (int32_t)(n4.size() / 32);
    c->sc.treelet_count = 0;
    if (root4 >= 0 && !getenv("MI_TRACER_NO_TREELET")) {
        const int32_t base = c->sc.treelet_base;
        std::vector<int32_t> top{root4};
        for (size_t q = 0, level_end = 1, lvl = 0; q < top.size() && lvl < MI_TREELET_LEVELS; q++) {
            for (int k = 0; k < 4; k++) {
                int32_t r;
                memcpy(&r, &n4[32 * (size_t)top[q] + 24 + k], 4);
                if (r >= 0 && r != MI_DONE && top.size() < MI_TREELET) top.push_back(r);
            }
            if (q + 1 == level_end) { level_end = top.size(); lvl++; }
        }
        std::vector<int32_t> copy_of(n4.size() / 32, -1);
        for (size_t k = 0; k < top.size(); k++) copy_of[top[k]] = base + (int32_t)k;
        for (size_t k = 0; k < top.size(); k++) {
            std::vector<float> node(n4.begin() + 32 * (size_t)top[k], n4.begin() + 32 * (size_t)top[k] + 32);
            for (int j = 0; j < 4; j++) {
                int32_t r;
                memcpy(&r, &node[24 + j], 4);
                if (r >= 0 && r != MI_DONE && copy_of[r] >= 0) { r = copy_of[r]; memcpy(&node[24 + j], &r, 4); }
            }
            n4.insert(n4.end(), node.begin(), node.end());
        }
        c->sc.treelet_count = (int32_t)top.size();
        root4 = base;
    }
    HIPCHK(c->nodes4.ensure(std::max<size_t>(n4.size() * 4, 16)));
    if (!n4.empty()) HIPCHK(hipMemcpy(c->nodes4.p, n4.data(), n4.size() * 4, hipMemcpyHostToDevice));
    c->sc.root_ref = root4;
    return quantize_nodes4(c, (uint32_t)(n4.size() / 32));
}

#ifdef MI_WIDE8
// 8-wide build of the library: the binary tree (host copy) -> 8-wide nodes (mi_wide8.h), the primitive arrays on the
// device gathered into the order its leaf seats name, nodes uploaded.  Stands where install_bvh4 stands.
static mi_status install_bvh8(mi_ctx *c, const float *nodes2, int32_t root2) {
    Wide8Tree t;
    build_wide8(nodes2, root2, c->sc.n_tri, c->sc.n_sph, t);
    const uint32_t nt = c->sc.n_tri, ns = c->sc.n_sph, n = nt + ns;
    hipStream_t st = c->stream;
    if (root2 >= 0 && n > 1) {
        if (t.tri_src.size() != nt || t.sph_src.size() != ns) return fail(MI_ERR_INVALID, "internal: 8-wide collapse lost primitives (%zu + %zu of %u + %u)", t.tri_src.size(), t.sph_src.size(), nt, ns);
        DevBuf d_ts, d_ss, r_tri, r_nrm, r_mat, r_vm, r_sph, r_smat, r_id;
        DevBuf *tmp[] = {&d_ts, &d_ss, &r_tri, &r_nrm, &r_mat, &r_vm, &r_sph, &r_smat, &r_id};
        auto body = [&]() -> mi_status {
            HIPCHK(d_ts.ensure(std::max<size_t>((size_t)nt * 4, 16)));
            HIPCHK(d_ss.ensure(std::max<size_t>((size_t)ns * 4, 16)));
            if (nt) HIPCHK(hipMemcpyAsync(d_ts.p, t.tri_src.data(), (size_t)nt * 4, hipMemcpyHostToDevice, st));
            if (ns) HIPCHK(hipMemcpyAsync(d_ss.p, t.sph_src.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
            HIPCHK(r_tri.ensure((size_t)nt * 48 + 16));
            HIPCHK(r_nrm.ensure(std::max<size_t>((size_t)nt * 16, 16)));
            HIPCHK(r_mat.ensure(std::max<size_t>((size_t)nt * 12, 16)));
            HIPCHK(r_vm.ensure(std::max<size_t>((size_t)nt, 16)));
            HIPCHK(r_sph.ensure((size_t)ns * 16 + 64));
            HIPCHK(r_smat.ensure(std::max<size_t>((size_t)ns * 4, 16)));
            HIPCHK(r_id.ensure((size_t)n * 4));
            HIPCHK(hipMemsetAsync(r_tri.p, 0, (size_t)nt * 48 + 16, st));
            HIPCHK(hipMemsetAsync(r_sph.p, 0, (size_t)ns * 16 + 64, st));
            const Wide8Prims in{c->tri.as<float4>(), c->tri_normal.as<float4>(), c->sph.as<float4>(), c->tri_mat.as<uint32_t>(), c->sph_mat.as<uint32_t>(),
                                c->tri_has_vmat.as<uint8_t>(), c->prim_id.as<int32_t>()};
            const Wide8PrimsOut out{r_tri.as<float4>(), r_nrm.as<float4>(), r_sph.as<float4>(), r_mat.as<uint32_t>(), r_smat.as<uint32_t>(), r_vm.as<uint8_t>(),
                                    r_id.as<int32_t>()};
            hipLaunchKernelGGL(k_wide8_permute, dim3((n + 255) / 256), dim3(256), 0, st, nt, ns, d_ts.as<uint32_t>(), d_ss.as<uint32_t>(), in, out);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
            std::swap(c->tri, r_tri); std::swap(c->tri_normal, r_nrm); std::swap(c->tri_mat, r_mat); std::swap(c->tri_has_vmat, r_vm);
            std::swap(c->sph, r_sph); std::swap(c->sph_mat, r_smat); std::swap(c->prim_id, r_id);
            DevScene &d = c->sc;
            d.tri = c->tri.as<float4>(); d.tri_normal = c->tri_normal.as<float4>(); d.tri_mat = c->tri_mat.as<uint32_t>();
            d.tri_has_vmat = c->tri_has_vmat.as<uint8_t>(); d.sph = c->sph.as<float4>(); d.sph_mat = c->sph_mat.as<uint32_t>();
            d.prim_id = c->prim_id.as<int32_t>();
            return MI_OK;
        };
        const mi_status rs = body();
        for (DevBuf *b : tmp) b->release();
        if (rs != MI_OK) return rs;
    }
    const size_t n8 = t.words.size() / 20;
    for (size_t i = 0; i < n8; i++) { // nothing the kernels follow may point outside the arrays
        const uint32_t *w = &t.words[20 * i];
        const uint32_t imask = w[3] >> 24, tmask = w[7] & 0xffu, smask = (w[7] >> 8) & 0xffu;
        if ((imask & tmask) || (imask & smask) || (tmask & smask) || (w[7] >> 16) || (size_t)w[4] + __builtin_popcount(imask) > n8 || (imask && w[4] <= i) ||
            (size_t)w[5] + __builtin_popcount(tmask) > nt || (size_t)w[6] + __builtin_popcount(smask) > ns)
            return fail(MI_ERR_INVALID, "internal: 8-wide node %zu is inconsistent", i);
    }
    HIPCHK(c->qnodes.ensure(std::max<size_t>(t.words.size() * 4, 16) + 64));
    if (n8) HIPCHK(hipMemcpy(c->qnodes.p, t.words.data(), t.words.size() * 4, hipMemcpyHostToDevice));
    c->sc.qnodes = c->qnodes.as<uint4>();
    c->sc.treelet_base = 0;
    c->sc.treelet_count = getenv("MI_TRACER_NO_TREELET") ? 0 : (int32_t)std::min<size_t>(n8, MI_TREELET);
    c->sc.root_ref = root2 < 0 ? root2 : 0;
    for (int a = 0; a < 3; a++) { c->sc.qs[a] = 1.0f; c->sc.qb[a] = 0.0f; }
    c->n_nodes8 = (uint32_t)n8;
    c->levels8 = t.levels;
    mi_status pk = pack_shade(c);
    if (pk != MI_OK) return pk;
    HIPCHK(hipStreamSynchronize(st));
    return MI_OK;
}
#endif

MI_API mi_status mi_scene_upload(mi_ctx *c, const mi_scene_soa *s) {
    if (!c || !s) return fail(MI_ERR_INVALID, "mi_scene_upload: NULL argument");
    c->has_scene = false; // a rejected or failed upload leaves the context without a scene
    int depth2 = 0;
    mi_status vs = validate_scene(s, &depth2);
    if (vs != MI_OK) return vs;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize()); // kernels an enqueue-only render left on a caller's stream may still read the scene about to be replaced
    HIPCHK(c->nodes.ensure(std::max<size_t>((size_t)s->n_nodes * 64, 16)));
    if (s->n_nodes) HIPCHK(hipMemcpy(c->nodes.p, s->nodes, (size_t)s->n_nodes * 64, hipMemcpyHostToDevice));
    mi_status us = upload_prims(c, s);
    if (us != MI_OK) return us;
    c->root_ref2 = s->root_ref;
    c->n_nodes = s->n_nodes;
#ifdef MI_WIDE8
    mi_status cs = install_bvh8(c, s->nodes, (uint64_t)s->n_tri + s->n_sph > 1 ? s->root_ref : ~0);
#else
    mi_status cs = install_bvh4(c, s->nodes, (uint64_t)s->n_tri + s->n_sph > 0 ? s->root_ref : ~0);
#endif
    if (cs != MI_OK) return cs;
    c->max_stack = stack_need(depth2);
#ifndef MI_WIDE8
    // the dynamic-programming collapse may spend a 4-wide level on a single binary level: no better bound than the tree's depth
    if (const char *how = getenv("MI_TRACER_COLLAPSE")) if (!strcmp(how, "dp")) c->max_stack = std::min(MI_MAX_STACK, 3 * depth2 + 3);
#endif
    c->has_scene = true;
    return MI_OK;
}

// Sum of the surface areas of all child boxes of the resident tree / area of the root's
// box, + 1 for the root: the SAH cost with unit node and leaf costs.
static mi_status resident_sah_cost(mi_ctx *c, double *out) {
    *out = 0.0;
    if (c->n_nodes == 0 || c->root_ref2 < 0 || (uint32_t)c->root_ref2 >= c->n_nodes) return MI_OK; // no interior node
    DevBuf acc;
    HIPCHK(acc.ensure(8));
    HIPCHK(hipMemsetAsync(acc.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_nodes_area, dim3((c->n_nodes + MI_BUILD_BLOCK - 1) / MI_BUILD_BLOCK), dim3(MI_BUILD_BLOCK), 0, c->stream, c->n_nodes,
                       c->nodes.as<float4>(), acc.as<double>());
    double sum = 0;
    float root[16];
    hipError_t e = hipMemcpyAsync(&sum, acc.p, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(root, (const char *)c->nodes.p + (size_t)c->root_ref2 * 64, 64, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    acc.release();
    HIPCHK(e);
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = std::min(root[k], root[6 + k]); hi[k] = std::max(root[3 + k], root[9 + k]); }
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const double a = 2.0 * (dx * dy + dy * dz + dz * dx);
    *out = a > 0 ? 1.0 + sum / a : 0.0;
    return MI_OK;
}

MI_API mi_status mi_scene_quality(mi_ctx *c, double *sah_cost_out) {
    if (!c || !sah_cost_out) return fail(MI_ERR_INVALID, "mi_scene_quality: NULL argument");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_scene_quality: no scene uploaded");
    HIPCHK(hipSetDevice(c->device));
    return resident_sah_cost(c, sah_cost_out);
}

MI_API mi_status mi_scene_download_nodes(mi_ctx *c, float *nodes_out, uint32_t capacity_nodes, uint32_t *n_nodes_out, int32_t *root_ref_out) {
    if (!c || !n_nodes_out || !root_ref_out) return fail(MI_ERR_INVALID, "mi_scene_download_nodes: NULL argument");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_scene_download_nodes: no scene uploaded");
    *n_nodes_out = c->n_nodes;
    *root_ref_out = c->root_ref2;
    if (!nodes_out) return MI_OK; // size query
    if (capacity_nodes < c->n_nodes) return fail(MI_ERR_INVALID, "mi_scene_download_nodes: room for %u nodes, tree has %u", capacity_nodes, c->n_nodes);
    HIPCHK(hipSetDevice(c->device));
    if (c->n_nodes) HIPCHK(hipMemcpy(nodes_out, c->nodes.p, (size_t)c->n_nodes * 64, hipMemcpyDeviceToHost));
    return MI_OK;
}

MI_API mi_status mi_scene_download_nodes4(mi_ctx *c, float *nodes4_out, uint32_t *qnodes_out, uint32_t capacity_nodes, uint32_t *n_nodes_out,
                                          int32_t *root_ref_out, int32_t *treelet_base_out, float grid_out[6]) {
    if (!c) return fail(MI_ERR_INVALID, "mi_scene_download_nodes4: ctx is NULL");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_scene_download_nodes4: no scene uploaded");
#ifdef MI_WIDE8
    return fail(MI_ERR_INVALID, "mi_scene_download_nodes4: this library was built with the 8-wide node format");
#endif
    const uint32_t n = c->sc.qnodes ? (uint32_t)(c->sc.treelet_base + c->sc.treelet_count) : 0u;
    if (n_nodes_out) *n_nodes_out = n;
    if (root_ref_out) *root_ref_out = c->sc.root_ref;
    if (treelet_base_out) *treelet_base_out = c->sc.treelet_base;
    if (grid_out) for (int a = 0; a < 3; a++) { grid_out[a] = c->sc.qs[a]; grid_out[3 + a] = c->sc.qb[a]; }
    if (!nodes4_out && !qnodes_out) return MI_OK;
    if (capacity_nodes < n) return fail(MI_ERR_INVALID, "mi_scene_download_nodes4: room for %u nodes, tree has %u", capacity_nodes, n);
    HIPCHK(hipSetDevice(c->device));
    if (n && nodes4_out) HIPCHK(hipMemcpy(nodes4_out, c->nodes4.p, (size_t)n * 128, hipMemcpyDeviceToHost));
    if (n && qnodes_out) HIPCHK(hipMemcpy(qnodes_out, c->qnodes.p, (size_t)n * 64, hipMemcpyDeviceToHost));
    return MI_OK;
}

MI_API mi_status mi_scene_download_prim_ids(mi_ctx *c, int32_t *prim_id_out, uint32_t capacity) {
    if (!c || !prim_id_out) return fail(MI_ERR_INVALID, "mi_scene_download_prim_ids: NULL argument");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_scene_download_prim_ids: no scene uploaded");
    const uint32_t n = c->sc.n_tri + c->sc.n_sph;
    if (capacity < n) return fail(MI_ERR_INVALID, "mi_scene_download_prim_ids: room for %u ids, scene has %u primitives", capacity, n);
    HIPCHK(hipSetDevice(c->device));
    if (n) HIPCHK(hipMemcpy(prim_id_out, c->prim_id.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return MI_OK;
}

MI_API mi_status mi_scene_upload_build(mi_ctx *c, const mi_scene_soa *s, const float *tri_verts, mi_build_stats *stats) {
    if (!c || !s) return fail(MI_ERR_INVALID, "mi_scene_upload_build: NULL argument");
    c->has_scene = false;
    if (stats) memset(stats, 0, sizeof *stats);
    mi_status vs = validate_prims(s);
    if (vs != MI_OK) return vs;
    if (s->n_tri && !tri_verts) return fail(MI_ERR_INVALID, "mi_scene_upload_build: tri_verts is NULL");
    const uint32_t n = s->n_tri + s->n_sph;
    for (size_t i = 0; i < 9 * (size_t)s->n_tri; i++)
        if (!(tri_verts[i] > -1e15f && tri_verts[i] < 1e15f)) return fail(MI_ERR_INVALID, "scene: non-finite or huge vertex");
    for (size_t i = 0; i < 4 * (size_t)s->n_sph; i++)
        if (!(s->sph_center_radius[i] > -1e15f && s->sph_center_radius[i] < 1e15f)) return fail(MI_ERR_INVALID, "scene: non-finite or huge sphere");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize()); // as in mi_scene_upload
    mi_status us = upload_prims(c, s);
    if (us != MI_OK) return us;
    c->sc.qnodes = nullptr;
    c->n_nodes = 0;
    if (n <= 1) { // no interior node: bvh.cu:174-183
        HIPCHK(c->nodes.ensure(16));
        c->root_ref2 = ~0;
#ifdef MI_WIDE8
        mi_status cs = install_bvh8(c, nullptr, ~0);
#else
        mi_status cs = install_bvh4(c, nullptr, ~0);
#endif
        if (cs != MI_OK) return cs;
        c->max_stack = stack_need(0);
        c->has_scene = true;
        return MI_OK;
    }
    hipStream_t st = c->stream;
    // build-time scratch, released when the build is done (the build is not a hot path)
    DevBuf td_box_a, td_box_b, td_idx_a, td_idx_b, td_node_a, td_node_b, td_nodes, td_small, td_bins, td_cnt;
    DevBuf verts, pb, sbox, keys_a, keys_b, vals_a, vals_b, children, par_n, par_l, nb, height, flags, area, cub_tmp, ref_a, ref_t, box_a, box_t, r_tri, r_nrm, r_mat, r_vm, r_sph, r_smat, r_id;
    DevBuf *tmp[] = {&td_box_a, &td_box_b, &td_idx_a, &td_idx_b, &td_node_a, &td_node_b, &td_nodes, &td_small, &td_bins, &td_cnt, &r_tri, &r_nrm, &r_mat, &r_vm, &r_sph, &r_smat, &r_id, &verts, &pb, &sbox, &keys_a, &keys_b, &vals_a, &vals_b, &children, &par_n, &par_l, &nb, &height, &flags, &area, &cub_tmp, &ref_a, &ref_t, &box_a, &box_t};
    mi_status rs = MI_OK;
    // MI_TRACER_BUILDER=lbvh selects the plain Karras LBVH; the default is PLOC over the same Morton order
    // MI_TRACER_BUILDER: sah (default) = top-down binned SAH, the host builder's tree; ploc = parallel locally-ordered
    // clustering over the Morton order (half the build time, trees that render 1-4 % slower); lbvh = plain Karras LBVH
    const char *builder = getenv("MI_TRACER_BUILDER");
    const bool use_sah = !builder || !*builder || !strcmp(builder, "sah");
    const bool use_ploc = !use_sah && !strcmp(builder, "ploc");
    if (!use_sah && !use_ploc && strcmp(builder, "lbvh")) return fail(MI_ERR_INVALID, "MI_TRACER_BUILDER=%s: expected sah, ploc or lbvh", builder);
    int32_t root2 = 0, root4 = 0;
    uint32_t n4 = 0;
    int treelet_n = 0, built_depth = 0;
    auto body = [&]() -> mi_status {
        HIPCHK(verts.ensure(std::max<size_t>((size_t)s->n_tri * 36, 16)));
        if (s->n_tri) HIPCHK(hipMemcpyAsync(verts.p, tri_verts, (size_t)s->n_tri * 36, hipMemcpyHostToDevice, st));
        HIPCHK(pb.ensure((size_t)n * sizeof(BuildBox)));
        HIPCHK(sbox.ensure(64));
        HIPCHK(keys_a.ensure((size_t)n * 8));
        HIPCHK(keys_b.ensure((size_t)n * 8));
        HIPCHK(vals_a.ensure((size_t)n * 4));
        HIPCHK(vals_b.ensure((size_t)n * 4));
        HIPCHK(children.ensure((size_t)n * 8));
        HIPCHK(par_n.ensure((size_t)n * 4));
        HIPCHK(par_l.ensure((size_t)n * 4));
        HIPCHK(nb.ensure((size_t)n * sizeof(BuildBox)));
        HIPCHK(height.ensure((size_t)n * 4));
        HIPCHK(flags.ensure((size_t)n * 4));
        HIPCHK(area.ensure(8));
        HIPCHK(c->nodes.ensure((size_t)(n - 1) * 64));
        size_t cub_bytes = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(),
                                                  vals_b.as<uint32_t>(), (int)n, 0, 62, st));
        size_t scan_bytes = 0; // the prefix sum of the 4-wide collapse shares the scratch
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags.as<uint32_t>(), vals_a.as<uint32_t>(), (int)n, st));
        HIPCHK(cub_tmp.ensure(std::max<size_t>(std::max(cub_bytes, scan_bytes), 16)));
        mi_status evs = get_events(c, 2); // the context's first two events bracket the build
        if (evs != MI_OK) return evs;
        hipEvent_t e0 = c->ev[0], e1 = c->ev[1];
        const float init_box[16] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, 0, 0, // scene box, two counters
                                    3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, 0, 0}; // box of the centroids
        HIPCHK(hipMemcpyAsync(sbox.p, init_box, 64, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st)); // host sources above are stack/borrowed memory
        HIPCHK(hipEventRecord(e0, st));
        const dim3 grid((n + MI_BUILD_BLOCK - 1) / MI_BUILD_BLOCK), block(MI_BUILD_BLOCK);
        HIPCHK(hipMemsetAsync(flags.p, 0, (size_t)n * 4, st));
        HIPCHK(hipMemsetAsync(height.p, 0, (size_t)n * 4, st));
        HIPCHK(hipMemsetAsync(area.p, 0, 8, st));
        hipLaunchKernelGGL(k_build_prim_bounds, grid, block, 0, st, s->n_tri, s->n_sph, verts.as<float>(), c->sph.as<float4>(), pb.as<BuildBox>(),
                           sbox.as<float>());
        if (!use_sah) {
            hipLaunchKernelGGL(k_build_morton, grid, block, 0, st, n, pb.as<BuildBox>(), sbox.as<float>(), keys_a.as<uint64_t>(), vals_a.as<uint32_t>());
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(cub_tmp.p, cub_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(),
                                                      vals_b.as<uint32_t>(), (int)n, 0, 62, st));
        }
        // primitives into the order vals_b names (position i takes slot vals_b[i]; per class), prim_id and boxes along with
        // them; vals_b[i] becomes the new slot.  Morton order before PLOC / LBVH, tree order after the top-down build.
        auto permute_prims = [&]() -> mi_status {
            const size_t nt = s->n_tri, ns = s->n_sph;
            HIPCHK(r_tri.ensure(nt * 48 + 16));
            HIPCHK(r_nrm.ensure(std::max<size_t>(nt * 16, 16)));
            HIPCHK(r_mat.ensure(std::max<size_t>(nt * 12, 16)));
            HIPCHK(r_vm.ensure(std::max<size_t>(nt, 16)));
            HIPCHK(r_sph.ensure(ns * 16 + 64)); // padded like c->sph (upload_prims)
            HIPCHK(r_smat.ensure(std::max<size_t>(ns * 4, 16)));
            HIPCHK(r_id.ensure((size_t)n * 4));
            HIPCHK(nb.ensure((size_t)n * sizeof(BuildBox))); // permuted boxes (nb is free until the fit)
            hipLaunchKernelGGL(k_build_is_tri, grid, block, 0, st, (int)n, vals_b.as<uint32_t>(), s->n_tri, flags.as<uint32_t>());
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(cub_tmp.p, scan_bytes, flags.as<uint32_t>(), vals_a.as<uint32_t>(), (int)n, st));
            const PrimArrays src{c->tri.as<float4>(), c->tri_normal.as<float4>(), c->sph.as<float4>(), c->tri_mat.as<uint32_t>(), c->sph_mat.as<uint32_t>(),
                                 c->tri_has_vmat.as<uint8_t>(), c->prim_id.as<int32_t>(), pb.as<BuildBox>()};
            const PrimArrays dst{r_tri.as<float4>(), r_nrm.as<float4>(), r_sph.as<float4>(), r_mat.as<uint32_t>(), r_smat.as<uint32_t>(),
                                 r_vm.as<uint8_t>(), r_id.as<int32_t>(), nb.as<BuildBox>()};
            hipLaunchKernelGGL(k_build_permute, grid, block, 0, st, (int)n, s->n_tri, vals_b.as<uint32_t>(), vals_a.as<uint32_t>(), src, dst);
            HIPCHK(hipGetLastError());
            std::swap(c->tri, r_tri); std::swap(c->tri_normal, r_nrm); std::swap(c->tri_mat, r_mat); std::swap(c->tri_has_vmat, r_vm);
            std::swap(c->sph, r_sph); std::swap(c->sph_mat, r_smat); std::swap(c->prim_id, r_id); std::swap(pb, nb);
            DevScene &d = c->sc;
            d.tri = c->tri.as<float4>(); d.tri_normal = c->tri_normal.as<float4>(); d.tri_mat = c->tri_mat.as<uint32_t>();
            d.tri_has_vmat = c->tri_has_vmat.as<uint8_t>(); d.sph = c->sph.as<float4>(); d.sph_mat = c->sph_mat.as<uint32_t>();
            d.prim_id = c->prim_id.as<int32_t>();
            HIPCHK(hipMemsetAsync(flags.p, 0, (size_t)n * 4, st)); // the fit's arrival flags start at zero
            return MI_OK;
        };
        if (!use_sah) { mi_status ps = permute_prims(); if (ps != MI_OK) return ps; }
        // padding needs the scene diagonal: one small read-back in the middle of the build
        float hb[8];
        HIPCHK(hipMemcpyAsync(hb, sbox.p, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double dx = (double)hb[3] - hb[0], dy = (double)hb[4] - hb[1], dz = (double)hb[5] - hb[2];
        const float pad = 1e-5f * (float)std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-30f; // as the host flattener
        const dim3 ngrid((n - 1 + MI_BUILD_BLOCK - 1) / MI_BUILD_BLOCK);
        if (use_sah) {
            // top-down binned SAH (mi_bvh_build.h): one tree level per round while nodes hold more than MI_TD_SMALL
            // primitives, then one wave per remaining node
            // Bounds that hold for every input: a level's nodes own disjoint ranges of more than 64 positions, small jobs
            // disjoint ranges of at least two; all top-phase nodes together are fewer than n (a chain that peels one
            // primitive per level is possible: 36 levels of SAH splits, then halving).
            const size_t cap_top = (size_t)n, cap_small = (size_t)n / 2 + 1, cap_level = (size_t)n / 64 + 2;
            HIPCHK(td_box_a.ensure((size_t)n * sizeof(BuildBox)));
            HIPCHK(td_box_b.ensure((size_t)n * sizeof(BuildBox)));
            HIPCHK(td_idx_a.ensure((size_t)n * 4));
            HIPCHK(td_idx_b.ensure((size_t)n * 4));
            HIPCHK(td_node_a.ensure((size_t)n * 4));
            HIPCHK(td_node_b.ensure((size_t)n * 4));
            HIPCHK(td_nodes.ensure(cap_top * sizeof(TdNode)));
            HIPCHK(td_small.ensure(cap_small * sizeof(TdSmall)));
            HIPCHK(td_bins.ensure(cap_level * MI_TD_BIN_WORDS * 4));
            HIPCHK(td_cnt.ensure(64));
            uint32_t cnt_h[4] = {0u, 0u, 0u, 0u}; // [1] nodes of the next level, [2] small jobs, [3] the root's index
            uint32_t level_first = 0, level_n = 0;
            if (n > MI_TD_SMALL) {
                TdNode root{};
                root.begin = 0; root.count = n; root.parent = -1; root.depth = 0;
                HIPCHK(hipMemcpyAsync(td_nodes.p, &root, sizeof root, hipMemcpyHostToDevice, st));
                level_n = 1;
            } else {
                const TdSmall job{0u, n, -1, 0u};
                HIPCHK(hipMemcpyAsync(td_small.p, &job, sizeof job, hipMemcpyHostToDevice, st));
                cnt_h[2] = 1u;
            }
            HIPCHK(hipMemcpyAsync(td_cnt.p, cnt_h, 16, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_td_init, grid, block, 0, st, n, pb.as<BuildBox>(), td_box_a.as<BuildBox>(), td_idx_a.as<uint32_t>(), td_node_a.as<uint32_t>(),
                               level_n ? 0u : MI_TD_NONE);
            HIPCHK(hipStreamSynchronize(st)); // root / job / cnt_h are stack memory
            DevBuf *bx = &td_box_a, *bx2 = &td_box_b, *ix = &td_idx_a, *ix2 = &td_idx_b, *nx = &td_node_a, *nx2 = &td_node_b;
            for (int level = 0; level_n > 0; level++) {
                if (level > 36 + 32) return fail(MI_ERR_HIP, "top-down build did not end");
                if (level_n > cap_level || (size_t)level_first + level_n > cap_top) return fail(MI_ERR_HIP, "top-down build: %u nodes on level %d", level_n, level);
                const uint32_t next_first = level_first + level_n;
                HIPCHK(hipMemsetAsync(td_bins.p, 0, (size_t)level_n * MI_TD_BIN_WORDS * 4, st));
                HIPCHK(hipMemsetAsync(td_cnt.as<uint32_t>() + 1, 0, 4, st));
                // positions per wave: 64 where nodes are small, up to 1024 where they are large (fewer, fuller flushes of the partial sums)
                uint32_t chunks = 1;
                while (chunks < 16u && (uint64_t)level_n * 4096u * chunks < n) chunks *= 2;
                const dim3 cgrid((n + MI_BUILD_BLOCK * chunks - 1) / (MI_BUILD_BLOCK * chunks));
                hipLaunchKernelGGL(k_td_bounds, cgrid, block, 0, st, n, bx->as<BuildBox>(), nx->as<uint32_t>(), td_nodes.as<TdNode>(), chunks);
                hipLaunchKernelGGL(k_td_bin, cgrid, block, 0, st, n, bx->as<BuildBox>(), nx->as<uint32_t>(), td_nodes.as<TdNode>(), level_first, td_bins.as<uint32_t>(),
                                   chunks);
                hipLaunchKernelGGL(k_td_split, dim3((level_n + MI_TD_SPLIT_BLOCK / 64 - 1) / (MI_TD_SPLIT_BLOCK / 64)), dim3(MI_TD_SPLIT_BLOCK), 0, st, td_nodes.as<TdNode>(), level_first, level_n, next_first, td_bins.as<uint32_t>(),
                                   bx->as<BuildBox>(), c->nodes.as<float4>(), par_n.as<int>(), td_cnt.as<uint32_t>(), td_small.as<TdSmall>(), pad);
                hipLaunchKernelGGL(k_td_flag, grid, block, 0, st, n, bx->as<BuildBox>(), nx->as<uint32_t>(), td_nodes.as<TdNode>(), flags.as<uint32_t>());
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(cub_tmp.p, scan_bytes, flags.as<uint32_t>(), vals_a.as<uint32_t>(), (int)n, st));
                hipLaunchKernelGGL(k_td_scatter, grid, block, 0, st, n, bx->as<BuildBox>(), ix->as<uint32_t>(), nx->as<uint32_t>(), td_nodes.as<TdNode>(),
                                   flags.as<uint32_t>(), vals_a.as<uint32_t>(), bx2->as<BuildBox>(), ix2->as<uint32_t>(), nx2->as<uint32_t>());
                HIPCHK(hipGetLastError());
                std::swap(bx, bx2); std::swap(ix, ix2); std::swap(nx, nx2);
                HIPCHK(hipMemcpyAsync(cnt_h, td_cnt.p, 16, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if ((size_t)next_first + cnt_h[1] > cap_top || cnt_h[2] > cap_small) return fail(MI_ERR_HIP, "top-down build: node lists overflow");
                level_first = next_first;
                level_n = cnt_h[1];
            }
            // every remaining node: one wave each; the primitives of a job end in depth-first order
            HIPCHK(hipMemcpyAsync(vals_b.p, ix->p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            if (cnt_h[2])
                hipLaunchKernelGGL(k_td_small, dim3((cnt_h[2] + 3) / 4), block, 0, st, cnt_h[2], td_small.as<TdSmall>(), bx->as<BuildBox>(), ix->as<uint32_t>(),
                                   c->nodes.as<float4>(), par_n.as<int>(), td_cnt.as<uint32_t>(), vals_b.as<uint32_t>(), pad);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(cnt_h, td_cnt.p, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (cnt_h[3] >= n - 1) return fail(MI_ERR_HIP, "top-down build: root %u of %u nodes", cnt_h[3], n - 1);
            mi_status ps = permute_prims();
            if (ps != MI_OK) return ps;
            hipLaunchKernelGGL(k_td_fix_refs, ngrid, block, 0, st, n - 1, c->nodes.as<float4>(), vals_b.as<uint32_t>());
            hipLaunchKernelGGL(k_nodes_area, ngrid, block, 0, st, n - 1, c->nodes.as<float4>(), area.as<double>());
            root2 = (int32_t)cnt_h[3];
        } else if (use_ploc) {
            // PLOC rounds: nearest neighbour in the window, merge mutual pairs, compact
            HIPCHK(ref_a.ensure((size_t)n * 4));
            HIPCHK(ref_t.ensure((size_t)n * 4));
            HIPCHK(box_a.ensure((size_t)n * sizeof(BuildBox)));
            HIPCHK(box_t.ensure((size_t)n * sizeof(BuildBox)));
            HIPCHK(hipMemsetAsync(sbox.as<uint32_t>() + 6, 0, 8, st)); // [6] node counter, [7] cluster count
            hipLaunchKernelGGL(k_ploc_init, grid, block, 0, st, (int)n, vals_b.as<uint32_t>(), pb.as<BuildBox>(), ref_a.as<int>(), box_a.as<BuildBox>());
            uint32_t m = n;
            for (int round = 0; m > 1; round++) {
                if (round > 4096) return fail(MI_ERR_HIP, "PLOC did not converge");
                const dim3 mg((m + MI_BUILD_BLOCK - 1) / MI_BUILD_BLOCK);
                hipLaunchKernelGGL(k_ploc_nearest, mg, block, 0, st, (int)m, box_a.as<BuildBox>(), par_l.as<int>());
                hipLaunchKernelGGL(k_ploc_merge, mg, block, 0, st, (int)m, par_l.as<int>(), ref_a.as<int>(), box_a.as<BuildBox>(), ref_t.as<int>(),
                                   box_t.as<BuildBox>(), flags.as<uint32_t>(), sbox.as<uint32_t>() + 6, c->nodes.as<float4>(), par_n.as<int>(), pad);
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(cub_tmp.p, scan_bytes, flags.as<uint32_t>(), vals_a.as<uint32_t>(), (int)m, st));
                hipLaunchKernelGGL(k_ploc_compact, mg, block, 0, st, (int)m, flags.as<uint32_t>(), vals_a.as<uint32_t>(), ref_t.as<int>(), box_t.as<BuildBox>(),
                                   ref_a.as<int>(), box_a.as<BuildBox>(), sbox.as<uint32_t>() + 7);
                HIPCHK(hipGetLastError());
                uint32_t m_new = 0;
                HIPCHK(hipMemcpyAsync(&m_new, sbox.as<uint32_t>() + 7, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (m_new == 0 || m_new >= m) return fail(MI_ERR_HIP, "PLOC round made no progress (%u -> %u clusters)", m, m_new);
                m = m_new;
            }
            HIPCHK(hipMemcpyAsync(&root2, ref_a.p, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (root2 < 0 || (uint32_t)root2 >= n - 1) return fail(MI_ERR_HIP, "PLOC produced a bad root");
            hipLaunchKernelGGL(k_nodes_area, ngrid, block, 0, st, n - 1, c->nodes.as<float4>(), area.as<double>());
        } else {
            hipLaunchKernelGGL(k_build_topology, grid, block, 0, st, (int)n, keys_b.as<uint64_t>(), children.as<int2>(), par_n.as<int>(), par_l.as<int>());
            hipLaunchKernelGGL(k_build_fit, grid, block, 0, st, (int)n, vals_b.as<uint32_t>(), pb.as<BuildBox>(), children.as<int2>(), par_n.as<int>(),
                               par_l.as<int>(), nb.as<BuildBox>(), height.as<int>(), flags.as<unsigned int>());
            hipLaunchKernelGGL(k_build_emit, grid, block, 0, st, (int)n, vals_b.as<uint32_t>(), pb.as<BuildBox>(), children.as<int2>(), nb.as<BuildBox>(), pad,
                               c->nodes.as<float4>(), area.as<double>());
            root2 = 0; // Karras: interior node 0 is the root
        }
        // 4-wide collapse on the device: even-depth flags (+ tree depth), prefix sum -> new indices, emit
        hipLaunchKernelGGL(k_build_even_depth, ngrid, block, 0, st, (int)(n - 1), par_n.as<int>(), flags.as<uint32_t>(), height.as<int>());
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(cub_tmp.p, scan_bytes, flags.as<uint32_t>(), vals_a.as<uint32_t>(), (int)(n - 1), st));
        HIPCHK(c->nodes4.ensure((size_t)(n - 1 + MI_TREELET) * 128)); // upper bound: every binary node survives; + the treelet copies
        hipLaunchKernelGGL(k_build_emit4, ngrid, block, 0, st, (int)(n - 1), c->nodes.as<float4>(), flags.as<uint32_t>(), vals_a.as<uint32_t>(),
                           c->nodes4.as<float>(), (int)MI_DONE);
        HIPCHK(hipGetLastError());
        // treelet: where the 4-wide nodes end and which of them is the root, then the copies behind them
        uint32_t root4u = 0, last_idx = 0, last_flag = 0;
        HIPCHK(hipMemcpyAsync(&root4u, vals_a.as<uint32_t>() + root2, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&last_idx, vals_a.as<uint32_t>() + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&last_flag, flags.as<uint32_t>() + (n - 2), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        n4 = last_idx + last_flag;
        if (n4 == 0 || n4 > n - 1 || root4u >= n4) return fail(MI_ERR_HIP, "4-wide collapse produced %u nodes, root %u", n4, root4u);
        hipLaunchKernelGGL(k_build_treelet, dim3(1), dim3(64), 0, st, c->nodes4.as<float>(), (int)root4u, (int)n4, MI_TREELET, MI_TREELET_LEVELS,
                           (int)MI_DONE, height.as<int>() + 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, st));
        int h_root = 0;
        double area_sum = 0;
        HIPCHK(hipMemcpyAsync(&h_root, height.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&treelet_n, height.as<int>() + 1, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&area_sum, area.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        root4 = (int32_t)n4; // the root's copy is the first node of the treelet
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (h_root > MI_MAX_DEPTH) return fail(MI_ERR_INVALID, "device-built BVH is %d levels deep (limit %d)", h_root, MI_MAX_DEPTH);
        built_depth = h_root;
        if (stats) {
            double a = 2.0 * (dx * dy + dy * dz + dz * dx);
            if (use_ploc || use_sah) { const double px = dx + 2.0 * pad, py = dy + 2.0 * pad, pz = dz + 2.0 * pad; a = 2.0 * (px * py + py * pz + pz * px); } // its boxes are summed padded
            stats->ms_build = ms;
            stats->sah_cost = a > 0 ? 1.0 + area_sum / a : 0.0;
            stats->depth = h_root;
            stats->n_nodes = n - 1;
        }
        return MI_OK;
    };
    rs = body();
    for (DevBuf *b : tmp) b->release();
    if (rs != MI_OK) return rs;
    c->root_ref2 = root2;
    c->n_nodes = n - 1;
#ifdef MI_WIDE8
    {   // experiment stage of the 8-wide format: the device-built binary tree is collapsed on the host
        std::vector<float> h2((size_t)(n - 1) * 16);
        HIPCHK(hipMemcpy(h2.data(), c->nodes.p, h2.size() * 4, hipMemcpyDeviceToHost));
        mi_status cs8 = install_bvh8(c, h2.data(), root2);
        if (cs8 != MI_OK) return cs8;
        c->max_stack = stack_need(built_depth);
        c->has_scene = true;
        return MI_OK;
    }
#endif
    c->sc.root_ref = root4; // collapsed on the device above
    c->sc.treelet_base = (int32_t)n4;
    c->sc.treelet_count = treelet_n;
    mi_status qs = quantize_nodes4(c, n4 + (uint32_t)treelet_n);
    if (qs != MI_OK) return qs;
    c->max_stack = stack_need(built_depth);
    c->has_scene = true;
    return MI_OK;
}

// Rotation of the level-0 chunk deal (ChunkDeal in mi_kernels.h).  The slots of a batch are sample-major
// (slot = sample * npix + pixel), so the virtual chunks repeat with period P = npix / 64 per sample; wave
// w's k-th chunk sits at pixel position (w + k * (W + rot)) mod P and visits P / gcd(W + rot, P) distinct
// positions.  0 (the plain deal) unless that count is small; then the first rotation near 0.618 W whose
// count is not.
static uint32_t deal_rotation(uint32_t n_waves, uint32_t npix, int spp) {
    if (spp <= 1 || npix % 64u != 0u || n_waves < 8u) return 0u;
    const uint32_t P = npix / 64u;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };
    auto distinct = [&](uint32_t rot) { return P / gcd((n_waves + rot) % P == 0u ? P : (n_waves + rot) % P, P); };
    const uint32_t want = std::min<uint32_t>(P, 64u); // a wave deals with ~100 chunks per launch: 64 positions average the image well
    if (distinct(0u) >= want) return 0u;
    uint32_t rot = (uint32_t)((unsigned long long)n_waves * 2654435769ull >> 32) | 1u;
    for (int tries = 0; tries < 64 && rot + 2u < n_waves && distinct(rot) < want; tries++) rot += 2u;
    return rot < n_waves ? rot : 0u;
}

// Constants of fdiv() (mi_kernels.h) for the divisor d >= 1, checked against the machine's division on the edges
// of the range and on a pseudo-random sample before any kernel uses them.
static bool fastdiv_make(uint32_t d, FastDiv &out) {
    if (d == 0) return false;
    int L = 31;
    while (!((d >> L) & 1u)) L--;
    if ((d & (d - 1u)) == 0u) {
        out = FastDiv{0u, 0u, (uint32_t)L};
    } else {
        const unsigned __int128 num = (unsigned __int128)1 << (33 + L);
        out = FastDiv{(uint32_t)(uint64_t)(num / d - ((unsigned __int128)1 << 32) + 1), 1u, (uint32_t)L};
    }
    auto q = [&](uint32_t n) { const uint32_t t = (uint32_t)(((uint64_t)out.m * n) >> 32); return (t + ((n - t) >> out.s1)) >> out.s2; };
    uint32_t x = 0x9E3779B9u;
    for (int i = 0; i < 4096; i++) {
        uint32_t n;
        switch (i) {
            case 0: n = 0u; break;
            case 1: n = d - 1u; break;
            case 2: n = d; break;
            case 3: n = 0xFFFFFFFFu; break;
            case 4: n = 0xFFFFFFFFu / d * d; break;
            case 5: n = 0xFFFFFFFFu / d * d - 1u; break;
            default: x = x * 1664525u + 1013904223u; n = (i & 1) ? x : x / d * d + ((i & 2) ? d - 1u : 0u); break; // random, and the two sides of a multiple
        }
        if (q(n) != n / d) return false;
    }
    return true;
}

static mi_status make_geom(const mi_render_params *p, ShardGeom &g) {
    if (!p) return fail(MI_ERR_INVALID, "params is NULL");
    if (p->width <= 0 || p->height <= 0 || p->samples_per_dim <= 0 || p->samples_per_dim > 1024)
        return fail(MI_ERR_INVALID, "bad image size or sample count");
    if (p->depth < -1 || p->depth + 2 > kMaxLevels) return fail(MI_ERR_INVALID, "depth %d out of range", p->depth);
    g.width = p->width; g.height = p->height;
    g.tile_w = p->tile_w > 0 ? p->tile_w : 32;
    g.tile_h = p->tile_h > 0 ? p->tile_h : 32;
    if (g.tile_w > 32768 || g.tile_h > 32768) return fail(MI_ERR_INVALID, "tile larger than 32768 pixels on a side");
    g.tiles_x = (g.width + g.tile_w - 1) / g.tile_w;
    int tiles_y = (g.height + g.tile_h - 1) / g.tile_h;
    int64_t nt = (int64_t)g.tiles_x * tiles_y;
    if (nt * g.tile_w * g.tile_h >= ((int64_t)1 << 31)) return fail(MI_ERR_INVALID, "image too large (padded pixel count must fit 31 bits)");
    g.n_tiles = (int32_t)nt;
    g.count = p->shard_count > 1 ? p->shard_count : 1;
    g.rank = p->shard_count > 1 ? p->shard_rank : 0;
    if (g.rank < 0 || g.rank >= g.count) return fail(MI_ERR_INVALID, "shard_rank %d not in [0,%d)", g.rank, g.count);
    g.samples = p->samples_per_dim;
    g.cam_x = g.cam_y = nullptr; // camera_tables() below attaches them
    const bool sub8 = !(g.tile_w & 7) && !(g.tile_h & 7); // slot_to_local walks 8x8 sub-blocks only then
    if (!fastdiv_make((uint32_t)(g.tile_w * g.tile_h), g.d_tpix) || !fastdiv_make((uint32_t)g.tiles_x, g.d_tiles_x) ||
        !fastdiv_make((uint32_t)g.tile_w, g.d_tile_w) || !fastdiv_make(sub8 ? (uint32_t)g.tile_w >> 3 : 1u, g.d_sbw) ||
        !fastdiv_make((uint32_t)g.samples, g.d_samples))
        return fail(MI_ERR_INVALID, "internal: division constants failed their check");
    return MI_OK;
}

// Attaches the camera-plane tables for g's image and sample grid (ShardGeom::cam_x): rebuilt on `st` when the frame's
// (width, height, samples) differ from the cached ones.  may_build = false (the entry point that allocates nothing): only
// tables that are there already are used.  Without tables — tiles not multiples of 8, or a table beyond 64 MiB — the
// kernels compute every primary ray from its slot as before.  Used by level 0 of k_extend: C3 +0.8 % (five alternations on one
// box: 9783 against 9704 Mrays/s; level-0 extend 1.80 -> 1.71 ms); k_shade's level 0 keeps the plain form (mi_kernels.h).
static mi_status camera_tables(mi_ctx *c, ShardGeom &g, hipStream_t st, bool may_build) {
    g.cam_x = g.cam_y = nullptr;
    if (getenv("MI_TRACER_NO_CAMERA_TABLES")) return MI_OK;
    if ((g.tile_w & 7) || (g.tile_h & 7)) return MI_OK;
    const size_t nx = (size_t)g.samples * g.width, ny = (size_t)g.samples * g.height;
    if ((nx + ny) * 4 > ((size_t)64 << 20)) return MI_OK;
    if (c->cam_w != g.width || c->cam_h != g.height || c->cam_s != g.samples || !c->cam_tab.p) {
        if (!may_build) return MI_OK;
        HIPCHK(hipDeviceSynchronize()); // frames an enqueue-only call left running may still read the tables about to be replaced
        HIPCHK(c->cam_tab.ensure((nx + ny) * 4));
        c->cam_w = c->cam_h = c->cam_s = 0;
        hipLaunchKernelGGL(k_camera_tables, dim3((unsigned)((nx + ny + MI_BLOCK - 1) / MI_BLOCK)), dim3(MI_BLOCK), 0, st, g.width, g.height, g.samples,
                           c->cam_tab.as<float>(), c->cam_tab.as<float>() + nx);
        HIPCHK(hipGetLastError());
        c->cam_w = g.width; c->cam_h = g.height; c->cam_s = g.samples;
    }
    g.cam_x = c->cam_tab.as<float>();
    g.cam_y = c->cam_tab.as<float>() + nx;
    return MI_OK;
}

static int64_t shard_pixels(const ShardGeom &g) {
    int64_t owned = g.rank < g.n_tiles ? ((int64_t)g.n_tiles - g.rank + g.count - 1) / g.count : 0;
    return owned * g.tile_w * g.tile_h;
}

MI_API int64_t mi_shard_pixels(const mi_render_params *p) {
    ShardGeom g;
    if (make_geom(p, g) != MI_OK) return -1;
    return shard_pixels(g);
}


__global__ void k_sum_counts(const uint32_t *counts, int n, uint64_t *totals, const uint32_t *tail_level) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t s = 0, ext = 0;
        const uint32_t tl = *tail_level; // 0, or the level k_tail took over at
        for (int i = 0; i < n; i++) {
            uint64_t c = 0;
            for (int j = 0; j < MI_SUB; j++) c += counts[(i * MI_SUB + j) * MI_CNT_STRIDE];
            s += c;
            if (tl == 0 || (uint32_t)i < tl) ext += c;
        }
        totals[8] += ext; // rays traced by k_extend launches
        totals[0] += s + totals[7]; // rays cast = sum of extend queue lengths + rays k_tail traced beyond its first level
        totals[7] = 0;
        totals[1] += counts[0];     // paths started
    }
}

// Sizes of one wavefront batch of npix_b pixels (cap = npix_b * spp path slots).
struct BatchPlan {
    int64_t npix_b;
    size_t cap;        // path slots
    int sgrid;         // workgroups of k_shade
    uint32_t segcap;   // stream positions per sub-stream
    size_t stream_cap; // stream positions in all
    int hist_levels;
};
// Ray streams: MI_SUB sub-streams of `segcap` positions each.  Shade workgroup b appends to sub-stream
// b % MI_SUB and handles at most ceil((n_virtual + 255) / sgrid) + 256 rays (grid-stride loop over 256-ray
// groups), n_virtual <= cap + 64*MI_SUB.
static mi_status plan_batch(const mi_ctx *c, int64_t npix_b, int spp, int depth, BatchPlan &bp) {
    const int64_t cap64 = npix_b * spp;
    if (cap64 >= ((int64_t)1 << 32) - 64) return fail(MI_ERR_INVALID, "batch too large (samples_per_dim too high)");
    bp.npix_b = npix_b;
    bp.cap = (size_t)cap64;
    bp.hist_levels = std::max(1, depth + 1);
    bp.sgrid = (int)std::min<int64_t>((int64_t)c->n_cu * c->shade_wg_per_cu, (int64_t)(bp.cap + MI_BLOCK - 1) / MI_BLOCK);
    const size_t wg_per_sub = ((size_t)bp.sgrid + MI_SUB - 1) / MI_SUB;
    const size_t segcap64 = (wg_per_sub * ((bp.cap + 64 * MI_SUB + MI_BLOCK + (size_t)bp.sgrid - 1) / (size_t)bp.sgrid + MI_BLOCK) + 63) / 64 * 64;
    if (segcap64 * MI_SUB >= ((size_t)1 << 32)) return fail(MI_ERR_INVALID, "batch too large for 32-bit stream positions");
    bp.segcap = (uint32_t)segcap64;
    bp.stream_cap = segcap64 * MI_SUB;
    return MI_OK;
}
// Device memory of a batch and the grids of the persistent kernels (the overflow stacks in `ovf` have one
// column per thread of the largest of them).
struct PathMem {
    float4 *ray_o[2], *ray_d[2], *thr[2], *path_out, *hist;
    float2 *hits; // (t, primitive slot) by stream position
    uint32_t *counters;
    int *ovf;
    uint64_t *totals;
    int egrid, egrid0, tgrid; // workgroups of k_extend (levels >= 1), of its level-0 variant, of k_tail
};

// The wavefront loop over all batches of a shard: shared by mi_render_device (buffers grown on demand) and
// mi_render_device_noalloc (buffers carved out of the static workspace).  timed: per-launch events may be
// created; sync: wait for the stream and fill *stats (otherwise the work is only enqueued).
// n_sets = 2: the batches alternate between the caller's stream (set 0) and the context's second stream (set 1),
// each with its own path state, so that the sparse end of one batch — the last waves of every launch and the
// latency-bound k_tail — runs beside the dense start of the next.  Every batch writes its own pixels.
#ifdef MI_WAVE_TIMES
static unsigned long long *g_wt_buf = nullptr; // diagnostic build only: per-wave time stamps of the k_extend launches (mi_kernels.h)
#endif
// What mi_render appends to a frame BEFORE the host waits for it: un-permute kernels and the copies to the caller's
// host buffers, enqueued on the frame's stream behind the join, so that the frame costs one host wait, not two.
struct PostWork {
    virtual hipError_t enqueue(hipStream_t st) = 0;
    virtual ~PostWork() {}
};
static mi_status render_batches(mi_ctx *c, const mi_render_params *p, const ShardGeom &g, const BatchPlan &bp, const PathMem *sets, int n_sets,
                                void *d_rgba_mean, void *d_rgba8, hipStream_t st, mi_stats *stats, bool timed, bool sync, PostWork *post = nullptr) {
    const int64_t local = shard_pixels(g);
    const int spp = g.samples * g.samples;
    const int levels = p->depth + 2; // rays per path <= D+2 (CpuTracer.cu:119-131)
    const int64_t npix_b = bp.npix_b;
    const size_t cap = bp.cap;
    const int hist_levels = bp.hist_levels;
    const bool count_work = p->count_work != 0;
    const int sgrid = bp.sgrid;
    const uint32_t segcap = bp.segcap;
    const int64_t n_batches = local > 0 ? (local + npix_b - 1) / npix_b : 0;
    if (n_batches < 2) n_sets = 1;
    const size_t n_ev = 2 + (size_t)n_batches * levels * 4;
    const bool per_launch = timed && n_ev <= 4096 && ((p->flags & MI_RENDER_LAUNCH_TIMES) || getenv("MI_TRACER_DEBUG_LEVELS")); // on request, and only while the event count stays modest
    if (timed) {
        mi_status es = get_events(c, per_launch ? n_ev : 2);
        if (es != MI_OK) return es;
    }
    hipStream_t streams[2] = {st, c->stream2};
    for (int s = 0; s < n_sets; s++) HIPCHK(hipMemsetAsync(sets[s].totals, 0, kTotals * 8, st));
    if (sync) HIPCHK(hipEventRecord(c->ev[0], st));
    if (n_sets > 1) { // fork: the second stream starts behind everything the caller's stream holds so far
        HIPCHK(hipEventRecord(c->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    // One batch = begin (counters cleared), `levels` x (k_tail | k_extend | k_shade), end (resolve, totals); every
    // batch on the stream of its set.  (Enqueuing the launches of two neighbouring batches level by level instead of
    // batch by batch — so that the second stream starts at once instead of ~0.8 ms of launch calls later — measured
    // the same step time at every stagger.)
    struct Batch { uint32_t pix0, np, rot0; int64_t nslots; FastDiv d_np; };
    auto batch_of = [&](int64_t b, Batch &B) -> bool {
        B.pix0 = (uint32_t)(b * npix_b);
        B.np = (uint32_t)std::min<int64_t>(npix_b, local - (int64_t)B.pix0);
        B.nslots = (int64_t)B.np * spp;
        B.rot0 = deal_rotation((uint32_t)sets[b % n_sets].egrid0 * (MI_BLOCK / 64), B.np, spp);
        return fastdiv_make(B.np, B.d_np);
    };
    auto begin = [&](int64_t b) -> hipError_t { return hipMemsetAsync(sets[b % n_sets].counters, 0, kCounterWords * 4, streams[b % n_sets]); };
    auto level = [&](int64_t b, const Batch &B, int k) -> hipError_t {
        const PathMem &m = sets[b % n_sets];
        hipStream_t bs = streams[b % n_sets];
        uint32_t *cnt = m.counters;                                   // cnt[k*kLevelWords + j*MI_CNT_STRIDE] = rays in sub-stream j of level k
        uint32_t *tail_level = cnt + (kMaxLevels + 1) * kLevelWords;  // 0, or the level at which k_tail took over
        const RayStream rs[2] = {{m.ray_o[0], m.ray_d[0], m.thr[0], segcap}, {m.ray_o[1], m.ray_d[1], m.thr[1], segcap}};
        const uint32_t pix0 = B.pix0, np = B.np, rot0 = B.rot0;
        const int64_t nslots = B.nslots;
        const FastDiv d_np = B.d_np;
        const uint32_t tail_thr = c->tail_auto ? (uint32_t)std::min<int64_t>(c->tail_threshold, std::max<int64_t>(200000, nslots / 40)) : (uint32_t)c->tail_threshold;
        size_t evi = 2 + ((size_t)b * levels + k) * 4;
        hipError_t e = hipSuccess;
#define MI_EV() if (per_launch && (e = hipEventRecord(c->ev[evi++], bs)) != hipSuccess) return e
        const RayStream &sin = rs[k & 1], &sout = rs[(k + 1) & 1];
        unsigned long long *ws = count_work ? (unsigned long long *)m.totals + 2 : nullptr;
#ifdef MI_WAVE_TIMES
        if (!g_wt_buf && getenv("MI_TRACER_WAVE_TIMES") && (e = hipMalloc((void **)&g_wt_buf, 8 * 8192 * 32)) != hipSuccess) return e;
        if (!count_work && g_wt_buf) { ws = g_wt_buf; if (k == 0 && (e = hipMemsetAsync(g_wt_buf, 0, 8 * 8192 * 32, bs)) != hipSuccess) return e; }
#endif
        MI_EV(); // 4 events per level: tail | extend | shade
        if (k >= 1 && c->tail_threshold > 0 && !count_work)
            hipLaunchKernelGGL(k_tail, dim3(m.tgrid), dim3(MI_BLOCK), 0, bs, c->sc, g, pix0, np, d_np, k, p->depth, p->seed, sin, cnt + k * kLevelWords,
                               tail_thr, tail_level, m.hist, cap, (size_t)hist_levels * cap, m.path_out, m.ovf,
                               (unsigned long long *)m.totals + 7, c->shade_min, c->tail_refill);
        MI_EV();
#define MI_EXTEND_ARGS(PRIM)                                                                                                   \
    dim3((PRIM) ? m.egrid0 : m.egrid), dim3(MI_BLOCK), 0, bs, c->sc, (PRIM) ? (const uint32_t *)nullptr : cnt + k * kLevelWords,                   \
        (PRIM) ? (uint32_t)nslots : 0u, segcap, sin.o, sin.d, m.hits, m.ovf, ws,                          \
        (PRIM) ? c->refill_primary : c->refill_min, c->leaf_min, g, pix0, np, d_np, cnt, tail_level, k, (PRIM) ? rot0 : 0u
        if (k == 0) {
            if (count_work) hipLaunchKernelGGL((k_extend<true, true>), MI_EXTEND_ARGS(true));
            else hipLaunchKernelGGL((k_extend<false, true>), MI_EXTEND_ARGS(true));
        } else {
            if (count_work) hipLaunchKernelGGL((k_extend<true, false>), MI_EXTEND_ARGS(false));
            else hipLaunchKernelGGL((k_extend<false, false>), MI_EXTEND_ARGS(false));
        }
#undef MI_EXTEND_ARGS
        MI_EV();
#define MI_SHADE_ARGS                                                                                                          \
    dim3(sgrid), dim3(MI_BLOCK), 0, bs, c->sc, g, pix0, np, d_np, k, p->depth - k, p->seed, sin, cnt + k * kLevelWords, m.hits,         \
        m.hist, cap, (size_t)hist_levels * cap, m.path_out, sout, cnt + (k + 1) * kLevelWords, (uint32_t)nslots, tail_level
        if (k == 0) hipLaunchKernelGGL(k_shade<true>, MI_SHADE_ARGS);
        else hipLaunchKernelGGL(k_shade<false>, MI_SHADE_ARGS);
#undef MI_SHADE_ARGS
        MI_EV();
#undef MI_EV
        return hipSuccess;
    };
    auto end = [&](int64_t b, const Batch &B) {
        const PathMem &m = sets[b % n_sets];
        hipStream_t bs = streams[b % n_sets];
        hipLaunchKernelGGL(k_resolve, dim3((B.np + MI_BLOCK - 1) / MI_BLOCK), dim3(MI_BLOCK), 0, bs, g, B.pix0, B.np, m.path_out,
                           (float4 *)d_rgba_mean, (uchar4 *)d_rgba8);
        hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(64), 0, bs, m.counters, levels, m.totals, m.counters + (kMaxLevels + 1) * kLevelWords);
    };
    const size_t evi = 2 + (size_t)n_batches * levels * 4;
    const bool ranges = roctx_on();
    RoctxRange frame_range(ranges, "mi_render");
    for (int64_t b = 0; b < n_batches; b++) {
        Batch B;
        if (!batch_of(b, B)) return fail(MI_ERR_INVALID, "internal: division constants failed their check");
        RoctxRange batch_range(ranges, "batch %lld", (long long)b);
        HIPCHK(begin(b));
        for (int k = 0; k < levels; k++) {
            RoctxRange level_range(ranges, "level %lld", (long long)k);
            HIPCHK(level(b, B, k));
        }
        end(b, B);
    }
    HIPCHK(hipGetLastError());
    if (n_sets > 1) { // join: the caller's stream continues when the second stream has finished its batches
        HIPCHK(hipEventRecord(c->ev_join, c->stream2));
        HIPCHK(hipStreamWaitEvent(st, c->ev_join, 0));
    }
    if (post) HIPCHK(post->enqueue(st));
    if (!sync) return MI_OK;
    for (int s = 0; s < n_sets; s++) HIPCHK(hipMemcpyAsync(c->h_totals + s * kTotals, sets[s].totals, kTotals * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(c->ev[1], st));
    HIPCHK(hipStreamSynchronize(st));
    for (int s = 1; s < n_sets; s++) // the two sets' totals are sums, except two maxima (slots 4 and 6)
        for (int k = 0; k < kTotals; k++) c->h_totals[k] = (k == 4 || k == 6 || k == 48 || k == 50 || k == 51) ? std::max(c->h_totals[k], c->h_totals[s * kTotals + k]) : c->h_totals[k] + c->h_totals[s * kTotals + k];
#ifdef MI_WAVE_TIMES
    if (const char *wt_path = getenv("MI_TRACER_WAVE_TIMES")) { // the last batch's launches
        std::vector<unsigned long long> h(8 * 8192 * 4);
        if (g_wt_buf) {
            HIPCHK(hipMemcpy(h.data(), g_wt_buf, h.size() * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(wt_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
    }
#endif
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->rays_cast = c->h_totals[0];
        stats->paths = c->h_totals[1];
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        stats->ms_total = ms;
        stats->batches = (uint32_t)n_batches;
        stats->node_visits = c->h_totals[2];
        stats->prim_tests = c->h_totals[3];
        if (count_work && getenv("MI_TRACER_DEBUG_LEVELS"))
            fprintf(stderr, "max node steps of one ray %llu, wave iterations total %llu max %llu\n", (unsigned long long)c->h_totals[4],
                    (unsigned long long)c->h_totals[5], (unsigned long long)c->h_totals[6]);
        if (count_work && getenv("MI_TRACER_DEBUG_LEVELS")) {
            const char *what[7] = {"no ray (steady)", "node lanes idle in leaf phases", "leaf lanes idle in node phases", "idle while draining", "working while draining",
                                   "working in leaf phases", "working in node phases"};
            double tot = 0;
            for (int dd = 0; dd < 7; dd++) tot += (double)c->h_totals[10 + dd];
            for (int dd = 0; dd < 7; dd++) fprintf(stderr, "lane slots %-32s %5.1f %%\n", what[dd], 100.0 * (double)c->h_totals[10 + dd] / std::max(tot, 1.0));
        }
#ifdef MI_TAIL_STATS
        if (getenv("MI_TRACER_DEBUG_LEVELS"))
            fprintf(stderr, "k_tail: %llu waves, iterations max %llu mean %.1f; longest path: %llu own steps, %llu iterations from start to end; lane steps %llu = %.1f of 64 lanes\n",
                    (unsigned long long)c->h_totals[53], (unsigned long long)c->h_totals[48], (double)c->h_totals[49] / std::max<double>(1.0, (double)c->h_totals[53]),
                    (unsigned long long)c->h_totals[50], (unsigned long long)c->h_totals[51], (unsigned long long)c->h_totals[52],
                    (double)c->h_totals[52] / std::max<double>(1.0, (double)c->h_totals[49]));
#endif
        stats->rays_extend = c->h_totals[8];
        if (per_launch) {
            // With two streams the launches of neighbouring batches overlap: these are the launches' own durations
            // (what rocprofv3 --kernel-trace reports), and their sum can exceed ms_total.
            double me = 0, msh = 0, mt = 0;
            for (size_t i = 2; i + 4 <= evi; i += 4) {
                float t0 = 0, a = 0, b2 = 0;
                HIPCHK(hipEventElapsedTime(&t0, c->ev[i], c->ev[i + 1]));
                HIPCHK(hipEventElapsedTime(&a, c->ev[i + 1], c->ev[i + 2]));
                HIPCHK(hipEventElapsedTime(&b2, c->ev[i + 2], c->ev[i + 3]));
                mt += t0; me += a; msh += b2;
                stats->extend_launches++;
            }
            if (getenv("MI_TRACER_DEBUG_LEVELS")) { // per-launch timings, for tuning only
                size_t j = 2;
                for (int64_t b = 0; b < n_batches; b++)
                    for (int k = 0; k < levels; k++, j += 4) {
                        float t0 = 0, a = 0, b2 = 0;
                        (void)hipEventElapsedTime(&t0, c->ev[j], c->ev[j + 1]);
                        (void)hipEventElapsedTime(&a, c->ev[j + 1], c->ev[j + 2]);
                        (void)hipEventElapsedTime(&b2, c->ev[j + 2], c->ev[j + 3]);
                        fprintf(stderr, "batch %lld level %d tail %.3f ms extend %.3f ms shade %.3f ms\n", (long long)b, k, t0, a, b2);
                    }
            }
            stats->ms_tail = mt;
            stats->ms_extend = me;
            stats->ms_shade = msh;
            stats->ms_other = std::max(0.0, stats->ms_total - me - msh - mt);
        } else {
            stats->ms_other = stats->ms_total; // no per-kernel times without MI_RENDER_LAUNCH_TIMES
        }
    }
    return MI_OK;
}

static mi_status render_device_impl(mi_ctx *c, const mi_render_params *p, void *d_rgba_mean, void *d_rgba8, void *stream_v, mi_stats *stats, PostWork *post);
MI_API mi_status mi_render_device(mi_ctx *c, const mi_render_params *p, void *d_rgba_mean, void *d_rgba8, void *stream_v,
                                  mi_stats *stats) {
    return render_device_impl(c, p, d_rgba_mean, d_rgba8, stream_v, stats, nullptr);
}
static mi_status render_device_impl(mi_ctx *c, const mi_render_params *p, void *d_rgba_mean, void *d_rgba8, void *stream_v, mi_stats *stats, PostWork *post) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_render: no scene uploaded");
    ShardGeom g;
    mi_status gs = make_geom(p, g);
    if (gs != MI_OK) return gs;
    const bool enqueue_only = (p->flags & MI_RENDER_ENQUEUE_ONLY) != 0;
    if (enqueue_only && (stats || p->count_work || post)) return fail(MI_ERR_INVALID, "MI_RENDER_ENQUEUE_ONLY needs stats == NULL, no count_work, and mi_render_device");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = stream_v ? (hipStream_t)stream_v : c->stream;
    mi_status cts = camera_tables(c, g, st, true);
    if (cts != MI_OK) return cts;
    const int64_t local = shard_pixels(g);
    const int spp = g.samples * g.samples;
    // default batch: as many path slots as a quarter of the GPU's memory holds (there are two sets of path state, one
    // per stream), at most 256 Mi: per slot two ray streams, hits, a path value (~128 B) and 32 B of history per level.
    // 288 GB of HBM are there to be used: the fewer batches a frame has, the fewer sparse tails it pays (C4, 2.1 G
    // paths: 64 Mi slots per batch 8990 Mrays/s, 128 Mi 9265, 256 Mi 9511).
    const int64_t per_slot = 128 + 32 * (int64_t)std::max(1, p->depth + 1);
    int64_t want = p->batch_paths > 0 ? p->batch_paths : std::min<int64_t>((int64_t)256 << 20, std::max<int64_t>((int64_t)1 << 20, c->batch_budget / per_slot));
    want = std::min<int64_t>(want, (int64_t)1 << 30);
    int64_t npix_b = std::max<int64_t>(64, want / spp / 64 * 64);
    npix_b = std::min(npix_b, std::max<int64_t>(local, 1));
    // Two streams: a frame that fits one batch is cut into two (each must still be large against the tail
    // threshold: the point is to overlap one half's sparse end with the other half's dense start), multi-batch frames
    // alternate as they are.  MI_TRACER_STREAMS=1 keeps everything on the caller's stream.
    int n_sets = (c->n_streams > 1 && !p->count_work && !(p->flags & MI_RENDER_ONE_STREAM)) ? 2 : 1;
    if (local > npix_b) {
        // several batches: equal ones, not n - 1 full ones and a remainder, so that both streams carry the same load
        // (a 1/8 shard of C4 is 1.2 default batches: 61.8 ms as 225 M + 40 M paths, 52.1 ms as 2 x 132 M); three
        // become four (1/4 shard of C4: 106.4 -> 102.9 ms); from five on an odd count stays, fewer and larger batches
        // being worth more than the last one's idle stream (1/2 shard: 5 batches 199 ms, 6 batches 208 ms)
        int64_t n = (local + npix_b - 1) / npix_b;
        if (n_sets > 1 && n == 3 && p->batch_paths <= 0) n = 4;
        npix_b = std::max<int64_t>(64, ((local + n - 1) / n + 63) / 64 * 64);
    }
    if (n_sets > 1 && local <= npix_b) {
        const int64_t half = ((local + 1) / 2 + 63) / 64 * 64;
        if (half * spp >= c->split_min && half < local) npix_b = half;
        else n_sets = 1;
    }
    // The batch size above comes from the memory that was free when the context was created (batch_budget); other
    // contexts, or a host framework that allocated since, may have taken it.  A hipErrorOutOfMemory while the path
    // state grows is therefore not an error yet: both sets are released, the batch is halved and planned again, down
    // to 64 pixels; only then MI_ERR_OOM.
    BatchPlan bp;
    PathMem sets[2];
    for (;;) {
        mi_status ps = plan_batch(c, npix_b, spp, p->depth, bp);
        if (ps != MI_OK) return ps;
        hipError_t ae = hipSuccess;
        bool drained = false; // a buffer that must grow is freed first: frames an enqueue-only call left running may still use it
        auto grow = [&](DevBuf &b, size_t bytes) {
            if (ae == hipSuccess && bytes > b.bytes && !drained) { ae = hipDeviceSynchronize(); drained = true; }
            if (ae == hipSuccess) ae = b.ensure(bytes);
        };
        for (int s = 0; s < n_sets; s++) {
            auto &pb = c->pb[s];
            for (int i = 0; i < 2; i++) { grow(pb.ray_o[i], bp.stream_cap * 16); grow(pb.ray_d[i], bp.stream_cap * 16); grow(pb.thr[i], bp.stream_cap * 16); }
            grow(pb.hits, bp.stream_cap * 8);
            grow(pb.path_out, bp.cap * 16);
            grow(pb.hist, bp.cap * 32 * bp.hist_levels);
            grow(pb.counters, kCounterWords * 4);
            grow(pb.ovf, stack_bytes(c));
            grow(pb.totals, kTotals * 8);
        }
        if (ae == hipSuccess) break;
        if (ae != hipErrorOutOfMemory) HIPCHK(ae);
        (void)hipGetLastError(); // the failed hipMalloc's sticky error
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(c->stream2));
        for (auto &pb : c->pb) {
            DevBuf *set[] = {&pb.ray_o[0], &pb.ray_o[1], &pb.ray_d[0], &pb.ray_d[1], &pb.thr[0], &pb.thr[1], &pb.hits, &pb.path_out, &pb.hist};
            for (DevBuf *b : set) b->release();
        }
        if (npix_b <= 64) return fail(MI_ERR_OOM, "mi_render_device: no device memory for the path state of a 64-pixel batch");
        npix_b = std::max<int64_t>(64, npix_b / 2 / 64 * 64);
    }
    for (int s = 0; s < n_sets; s++) {
        auto &pb = c->pb[s];
        PathMem &m = sets[s];
        for (int i = 0; i < 2; i++) { m.ray_o[i] = pb.ray_o[i].as<float4>(); m.ray_d[i] = pb.ray_d[i].as<float4>(); m.thr[i] = pb.thr[i].as<float4>(); }
        m.hits = pb.hits.as<float2>(); m.path_out = pb.path_out.as<float4>(); m.hist = pb.hist.as<float4>();
        m.counters = pb.counters.as<uint32_t>(); m.ovf = pb.ovf.as<int>(); m.totals = pb.totals.as<uint64_t>();
        m.egrid = n_sets > 1 ? c->n_cu * c->extend_wg_two : extend_grid(c);
        m.egrid0 = c->n_cu * c->extend_wg_primary; m.tgrid = c->n_cu * c->tail_wg_per_cu;
    }
    return render_batches(c, p, g, bp, sets, n_sets, d_rgba_mean, d_rgba8, st, stats, !enqueue_only, !enqueue_only, post);
}

// ---- the entry point that allocates nothing (workspace: see `scratch` at the top of this file) ----------
MI_API int64_t mi_noalloc_workspace_bytes(void) { return (int64_t)MI_NOALLOC_BYTES; }

MI_API mi_status mi_render_device_noalloc(mi_ctx *c, const mi_render_params *p, void *d_rgba_mean, void *d_rgba8, void *stream_v,
                                          mi_stats *stats) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_render_device_noalloc: no scene uploaded");
    if (!c->ws) return fail(MI_ERR_HIP, "mi_render_device_noalloc: the context has no workspace address");
    if (p && p->count_work) return fail(MI_ERR_INVALID, "mi_render_device_noalloc: count_work is not supported");
    ShardGeom g;
    mi_status gs = make_geom(p, g);
    if (gs != MI_OK) return gs;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = stream_v ? (hipStream_t)stream_v : c->stream;
    mi_status cts = camera_tables(c, g, st, false); // no allocation here: only tables an earlier mi_render_device left for this frame size
    if (cts != MI_OK) return cts;
    const int64_t local = shard_pixels(g);
    const int spp = g.samples * g.samples;
    // Fixed part: totals, counters, overflow stacks.  The batches are small, so two persistent workgroups per CU
    // are plenty; the stack columns are only as long as the resident tree can make a stack (3 pushes per level of
    // the 4-wide tree, MI_LDS_STACK of them in LDS).
    auto up256 = [](size_t x) { return (x + 255) / 256 * 256; };
    PathMem m;
    const int wg = std::min(2, std::min(c->extend_wg_per_cu, std::min(c->extend_wg_primary, c->tail_wg_per_cu)));
    m.egrid = m.egrid0 = m.tgrid = c->n_cu * wg;
    const size_t ovf_entries = (size_t)std::max(0, c->max_stack - MI_LDS_STACK);
    size_t off = 0;
    auto carve = [&](size_t bytes) { unsigned char *q = c->ws + off; off += up256(bytes); return q; };
    m.totals = (uint64_t *)carve(kTotals * 8);
    m.counters = (uint32_t *)carve(kCounterWords * 4);
    m.ovf = (int *)carve(std::max<size_t>(ovf_entries * (size_t)m.egrid * MI_BLOCK * MI_STACK_ELEM_BYTES, 256));
    const size_t fixed = off;
    // The rest holds one batch: the largest multiple of 64 pixels that fits (per slot: two ray streams of three
    // float4 records, hits, path value, 32 B of history per level; plan_batch has the exact stream sizes).
    auto batch_bytes = [&](const BatchPlan &b) { return 6 * up256(b.stream_cap * 16) + up256(b.stream_cap * 8) + up256(b.cap * 16) + up256(b.cap * 32 * (size_t)b.hist_levels); };
    BatchPlan bp;
    int64_t lo = 0, hi = (std::max<int64_t>(local, 1) + 63) / 64; // in units of 64 pixels; lo fits (0 = nothing), find the largest that fits
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) / 2;
        BatchPlan t;
        const bool fits = plan_batch(c, mid * 64, spp, p->depth, t) == MI_OK && fixed + batch_bytes(t) <= MI_NOALLOC_BYTES;
        if (fits) lo = mid; else hi = mid - 1;
    }
    if (lo == 0) {
        BatchPlan t;
        size_t need = 0;
        if (plan_batch(c, 64, spp, p->depth, t) == MI_OK) need = fixed + batch_bytes(t);
        return fail(MI_ERR_INVALID, "mi_render_device_noalloc: the smallest batch (64 pixels x %d samples, depth %d, tree stack %d) needs %zu bytes, "
                    "the static workspace has %zu: use mi_render_device", spp, p->depth, c->max_stack, need, (size_t)MI_NOALLOC_BYTES);
    }
    mi_status ps = plan_batch(c, std::min<int64_t>(lo * 64, std::max<int64_t>(local, 1)), spp, p->depth, bp);
    if (ps != MI_OK) return ps;
    for (int i = 0; i < 2; i++) {
        m.ray_o[i] = (float4 *)carve(bp.stream_cap * 16);
        m.ray_d[i] = (float4 *)carve(bp.stream_cap * 16);
        m.thr[i] = (float4 *)carve(bp.stream_cap * 16);
    }
    m.hits = (float2 *)carve(bp.stream_cap * 8);
    m.path_out = (float4 *)carve(bp.cap * 16);
    m.hist = (float4 *)carve(bp.cap * 32 * (size_t)bp.hist_levels);
    if (off > MI_NOALLOC_BYTES) return fail(MI_ERR_INVALID, "mi_render_device_noalloc: internal sizing error (%zu > %zu)", off, (size_t)MI_NOALLOC_BYTES);
    // Every call on this GPU shares the array and finds the last call's contents in it: what must start at zero
    // (totals, counters) is cleared by render_batches on the stream; everything else is written before it is read.
    return render_batches(c, p, g, bp, &m, 1, d_rgba_mean, d_rgba8, st, stats, false, stats != nullptr);
}

MI_API mi_status mi_render(mi_ctx *c, const mi_render_params *p, float *rgba_mean_out, uint8_t *rgb8_out, mi_stats *stats) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_render: no scene uploaded");
    if (p && (p->flags & MI_RENDER_ENQUEUE_ONLY)) return fail(MI_ERR_INVALID, "mi_render: outputs in host memory need the wait (MI_RENDER_ENQUEUE_ONLY is for mi_render_device)");
    ShardGeom g;
    mi_status gs = make_geom(p, g);
    if (gs != MI_OK) return gs;
    HIPCHK(hipSetDevice(c->device));
    const int64_t local = shard_pixels(g);
    HIPCHK(c->out_mean.ensure(std::max<size_t>((size_t)local * 16, 16)));
    HIPCHK(c->out_rgba8.ensure(std::max<size_t>((size_t)local * 4, 16)));
    if (g.count == 1) {
        // The whole frame is ours: un-permute on the device and copy each output out, all enqueued behind the frame's
        // kernels before the one host wait (GpuTracer.cu:181-188: D2H of w*h float4 + a host setPixel loop).  A buffer
        // the caller pinned with mi_host_register is written by DMA directly; a pageable one goes through the
        // runtime's staging.
        const size_t npx = (size_t)g.width * g.height;
        if (rgb8_out) HIPCHK(c->out_rgb8.ensure(npx * 3));
        if (rgba_mean_out) HIPCHK(c->out_mean_frame.ensure(npx * 16));
        struct Post : PostWork {
            mi_ctx *c; ShardGeom g; int64_t local; size_t npx; float *mean_out; uint8_t *rgb_out;
            hipError_t enqueue(hipStream_t st) override {
                hipError_t e = hipSuccess;
                const dim3 grid((unsigned)((local + MI_BLOCK - 1) / MI_BLOCK));
                if (rgb_out) {
                    hipLaunchKernelGGL(k_assemble, grid, dim3(MI_BLOCK), 0, st, g, c->out_rgba8.as<uchar4>(), (int64_t)local, (int64_t)local, c->out_rgb8.as<uint8_t>());
                    if ((e = hipGetLastError()) != hipSuccess) return e;
                    if ((e = hipMemcpyAsync(rgb_out, c->out_rgb8.p, npx * 3, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
                }
                if (mean_out) {
                    hipLaunchKernelGGL(k_unpermute_mean, grid, dim3(MI_BLOCK), 0, st, g, (int64_t)local, c->out_mean.as<float4>(), c->out_mean_frame.as<float4>());
                    if ((e = hipGetLastError()) != hipSuccess) return e;
                    if ((e = hipMemcpyAsync(mean_out, c->out_mean_frame.p, npx * 16, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
                }
                return e;
            }
        } post;
        post.c = c; post.g = g; post.local = local; post.npx = npx; post.mean_out = rgba_mean_out; post.rgb_out = rgb8_out;
        // stats->ms_total then includes the copies: "first launch -> outputs complete"
        return render_device_impl(c, p, rgba_mean_out ? c->out_mean.p : nullptr, rgb8_out ? c->out_rgba8.p : nullptr, nullptr, stats, &post);
    }
    mi_status rs = mi_render_device(c, p, rgba_mean_out ? c->out_mean.p : nullptr, rgb8_out ? c->out_rgba8.p : nullptr, nullptr, stats);
    if (rs != MI_OK) return rs;
    // a shard of the frame: pixels of tiles it does not own must stay untouched in the caller's
    // buffers, so its tiles are scattered on the host
    std::vector<float> mean;
    std::vector<uint8_t> rgba;
    if (rgba_mean_out) { mean.resize((size_t)local * 4); HIPCHK(hipMemcpy(mean.data(), c->out_mean.p, (size_t)local * 16, hipMemcpyDeviceToHost)); }
    if (rgb8_out) { rgba.resize((size_t)local * 4); HIPCHK(hipMemcpy(rgba.data(), c->out_rgba8.p, (size_t)local * 4, hipMemcpyDeviceToHost)); }
    const int tpix = g.tile_w * g.tile_h;
    for (int64_t q = 0; q < local; q++) {
        int64_t j = q / tpix;
        int off = (int)(q - j * tpix);
        int64_t gt = g.rank + j * (int64_t)g.count;
        int px = (int)(gt % g.tiles_x) * g.tile_w + off % g.tile_w, py = (int)(gt / g.tiles_x) * g.tile_h + off / g.tile_w;
        if (px >= g.width || py >= g.height) continue;
        size_t o = (size_t)px + (size_t)py * g.width;
        if (rgba_mean_out) memcpy(rgba_mean_out + 4 * o, &mean[4 * (size_t)q], 16);
        if (rgb8_out) { rgb8_out[3 * o] = rgba[4 * (size_t)q]; rgb8_out[3 * o + 1] = rgba[4 * (size_t)q + 1]; rgb8_out[3 * o + 2] = rgba[4 * (size_t)q + 2]; }
    }
    return MI_OK;
}

static mi_status assemble_common(mi_ctx *c, const mi_render_params *p, const void *d_gathered, int64_t shard_stride_pixels, uint8_t *d_out,
                                  hipStream_t st) {
    ShardGeom g;
    mi_status gs = make_geom(p, g);
    if (gs != MI_OK) return gs;
    ShardGeom g0 = g;
    g0.rank = 0;
    const int64_t sp = shard_pixels(g0); // rank 0 owns the most tiles
    if (shard_stride_pixels < sp) return fail(MI_ERR_INVALID, "shard stride %lld < shard pixels %lld", (long long)shard_stride_pixels, (long long)sp);
    const int64_t total = sp * g.count; // every pixel of the frame belongs to exactly one shard: no clearing needed
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((total + MI_BLOCK - 1) / MI_BLOCK)), dim3(MI_BLOCK), 0, st, g, (const uchar4 *)d_gathered,
                       shard_stride_pixels, sp, d_out);
    HIPCHK(hipGetLastError());
    return MI_OK;
}

MI_API mi_status mi_assemble_rgb8(mi_ctx *c, const mi_render_params *p, const void *d_gathered, int64_t shard_stride_pixels, uint8_t *rgb8_out) {
    if (!c || !p || !d_gathered || !rgb8_out) return fail(MI_ERR_INVALID, "mi_assemble_rgb8: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = (size_t)p->width * p->height * 3;
    HIPCHK(c->out_rgb8.ensure(std::max<size_t>(bytes, 16)));
    mi_status as = assemble_common(c, p, d_gathered, shard_stride_pixels, c->out_rgb8.as<uint8_t>(), c->stream);
    if (as != MI_OK) return as;
    HIPCHK(hipMemcpyAsync(rgb8_out, c->out_rgb8.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MI_OK;
}

MI_API mi_status mi_assemble_rgb8_device(mi_ctx *c, const mi_render_params *p, const void *d_gathered, int64_t shard_stride_pixels,
                                         void *d_rgb8_out, void *stream) {
    if (!c || !p || !d_gathered || !d_rgb8_out) return fail(MI_ERR_INVALID, "mi_assemble_rgb8_device: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    mi_status as = assemble_common(c, p, d_gathered, shard_stride_pixels, (uint8_t *)d_rgb8_out, st);
    if (as != MI_OK) return as;
    HIPCHK(hipStreamSynchronize(st));
    return MI_OK;
}

// Shared tail of the two probes: extend over n rays already in ray_o/ray_d, copy out.
static mi_status probe_chunk(mi_ctx *c, int64_t n, int32_t *prim_id_out, float *t_out, float *n_out) {
    hipStream_t st = c->stream;
    hipLaunchKernelGGL((k_extend<false, false>), dim3(extend_grid(c)), dim3(MI_BLOCK), 0, st, c->sc,
                       (const uint32_t *)nullptr, (uint32_t)n, 0u, c->pb[0].ray_o[0].as<float4>(), c->pb[0].ray_d[0].as<float4>(), c->pb[0].hits.as<float2>(),
                       c->pb[0].ovf.as<int>(), (unsigned long long *)nullptr, c->refill_min, c->leaf_min, ShardGeom{}, 0u, 1u, FastDiv{0u, 0u, 0u},
                       (uint32_t *)nullptr, (const uint32_t *)nullptr, 0, 0u);
    hipLaunchKernelGGL(k_probe_out, dim3((unsigned)((n + MI_BLOCK - 1) / MI_BLOCK)), dim3(MI_BLOCK), 0, st, c->sc, n, c->pb[0].ray_o[0].as<float4>(),
                       c->pb[0].ray_d[0].as<float4>(), c->pb[0].hits.as<float2>(), c->probe_i.as<int32_t>(), c->probe_t.as<float>(),
                       n_out ? c->probe_n.as<float>() : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(prim_id_out, c->probe_i.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(t_out, c->probe_t.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (n_out) HIPCHK(hipMemcpyAsync(n_out, c->probe_n.p, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return MI_OK;
}

static mi_status probe_alloc(mi_ctx *c, size_t chunk) {
    HIPCHK(c->pb[0].ray_o[0].ensure(chunk * 16));
    HIPCHK(c->pb[0].ray_d[0].ensure(chunk * 16));
    HIPCHK(c->pb[0].hits.ensure(chunk * 8));
    HIPCHK(c->probe_i.ensure(chunk * 4));
    HIPCHK(c->probe_t.ensure(chunk * 4));
    HIPCHK(c->probe_n.ensure(chunk * 12));
    HIPCHK(c->pb[0].ovf.ensure(stack_bytes(c)));
    return MI_OK;
}

MI_API mi_status mi_trace_primary(mi_ctx *c, int32_t width, int32_t height, int32_t samples, int32_t *prim_id_out, float *t_out, float *n_out) {
    if (!c || !prim_id_out || !t_out) return fail(MI_ERR_INVALID, "mi_trace_primary: NULL argument");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_trace_primary: no scene uploaded");
    if (width <= 0 || height <= 0 || samples <= 0 || samples > 1024) return fail(MI_ERR_INVALID, "bad image size or sample count");
    HIPCHK(hipSetDevice(c->device));
    const int64_t total = (int64_t)width * height * samples * samples;
    const int64_t chunk = std::min<int64_t>(total, (int64_t)4 << 20);
    mi_status as = probe_alloc(c, (size_t)chunk);
    if (as != MI_OK) return as;
    for (int64_t first = 0; first < total; first += chunk) {
        int64_t n = std::min(chunk, total - first);
        hipLaunchKernelGGL(k_generate_primary, dim3((unsigned)((n + MI_BLOCK - 1) / MI_BLOCK)), dim3(MI_BLOCK), 0, c->stream, c->sc, width, height,
                           samples, first, n, c->pb[0].ray_o[0].as<float4>(), c->pb[0].ray_d[0].as<float4>());
        mi_status ps = probe_chunk(c, n, prim_id_out + first, t_out + first, n_out ? n_out + 3 * first : nullptr);
        if (ps != MI_OK) return ps;
    }
    return MI_OK;
}

MI_API mi_status mi_trace_rays(mi_ctx *c, int64_t total, const float *rays, int32_t *prim_id_out, float *t_out, float *n_out) {
    if (!c || !rays || !prim_id_out || !t_out || total < 0) return fail(MI_ERR_INVALID, "mi_trace_rays: bad argument");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_trace_rays: no scene uploaded");
    if (total == 0) return MI_OK;
    HIPCHK(hipSetDevice(c->device));
    const int64_t chunk = std::min<int64_t>(total, (int64_t)4 << 20);
    mi_status as = probe_alloc(c, (size_t)chunk);
    if (as != MI_OK) return as;
    std::vector<float> o4((size_t)chunk * 4), d4((size_t)chunk * 4);
    for (int64_t first = 0; first < total; first += chunk) {
        int64_t n = std::min(chunk, total - first);
        for (int64_t i = 0; i < n; i++) {
            const float *r = rays + 6 * (first + i);
            o4[4 * i] = r[0]; o4[4 * i + 1] = r[1]; o4[4 * i + 2] = r[2]; o4[4 * i + 3] = 0;
            d4[4 * i] = r[3]; d4[4 * i + 1] = r[4]; d4[4 * i + 2] = r[5]; d4[4 * i + 3] = 0;
        }
        HIPCHK(hipMemcpyAsync(c->pb[0].ray_o[0].p, o4.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->pb[0].ray_d[0].p, d4.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
        mi_status ps = probe_chunk(c, n, prim_id_out + first, t_out + first, n_out ? n_out + 3 * first : nullptr);
        if (ps != MI_OK) return ps;
    }
    return MI_OK;
}

MI_API mi_status mi_occluded_rays(mi_ctx *c, int64_t total, const float *rays, const float *tmax, uint8_t *occluded_out) {
    if (!c || !rays || !tmax || !occluded_out || total < 0) return fail(MI_ERR_INVALID, "mi_occluded_rays: bad argument");
    if (!c->has_scene) return fail(MI_ERR_NO_SCENE, "mi_occluded_rays: no scene uploaded");
    if (total == 0) return MI_OK;
    HIPCHK(hipSetDevice(c->device));
    const int64_t chunk = std::min<int64_t>(total, (int64_t)4 << 20);
    mi_status as = probe_alloc(c, (size_t)chunk);
    if (as != MI_OK) return as;
    hipStream_t st = c->stream;
    std::vector<float> o4((size_t)chunk * 4), d4((size_t)chunk * 4);
    for (int64_t first = 0; first < total; first += chunk) {
        const int64_t n = std::min(chunk, total - first);
        for (int64_t i = 0; i < n; i++) {
            const float *r = rays + 6 * (first + i);
            o4[4 * i] = r[0]; o4[4 * i + 1] = r[1]; o4[4 * i + 2] = r[2]; o4[4 * i + 3] = 0;
            d4[4 * i] = r[3]; d4[4 * i + 1] = r[4]; d4[4 * i + 2] = r[5]; d4[4 * i + 3] = tmax[first + i] < MI_NO_HIT ? tmax[first + i] : MI_NO_HIT; // clamp (also NaN) to the reference's "no hit yet" distance
        }
        HIPCHK(hipMemcpyAsync(c->pb[0].ray_o[0].p, o4.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c->pb[0].ray_d[0].p, d4.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL((k_extend<false, false, true>), dim3(extend_grid(c)), dim3(MI_BLOCK), 0, st, c->sc, (const uint32_t *)nullptr, (uint32_t)n,
                           0u, c->pb[0].ray_o[0].as<float4>(), c->pb[0].ray_d[0].as<float4>(), c->pb[0].hits.as<float2>(), c->pb[0].ovf.as<int>(),
                           (unsigned long long *)nullptr, c->refill_min, c->leaf_min, ShardGeom{}, 0u, 1u, FastDiv{0u, 0u, 0u}, (uint32_t *)nullptr,
                           (const uint32_t *)nullptr, 0, 0u);
        hipLaunchKernelGGL(k_probe_occluded, dim3((unsigned)((n + MI_BLOCK - 1) / MI_BLOCK)), dim3(MI_BLOCK), 0, st, n, c->pb[0].hits.as<float2>(),
                           c->probe_i.as<uint8_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(occluded_out + first, c->probe_i.p, (size_t)n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return MI_OK;
}

MI_API mi_status mi_cubemap_upload(mi_ctx *c, const mi_cubemap *cm) {
    if (!c) return fail(MI_ERR_INVALID, "mi_cubemap_upload: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize()); // as in mi_scene_upload
    if (!cm) { // back to the white background of CpuTracer.cu:125
        c->sc.cube = nullptr;
        return MI_OK;
    }
    size_t total = 0;
    int32_t header[32] = {0};
    for (int f = 0; f < 6; f++) {
        if (!cm->rgb[f] || cm->width[f] <= 0 || cm->height[f] <= 0 || cm->width[f] > 32768 || cm->height[f] > 32768)
            return fail(MI_ERR_INVALID, "mi_cubemap_upload: face %d missing or of bad size", f);
        header[3 * f] = cm->width[f];
        header[3 * f + 1] = cm->height[f];
        header[3 * f + 2] = (int32_t)total;
        total += ((size_t)cm->width[f] * cm->height[f] * 3 + 15) / 16 * 16;
        if (total >= ((size_t)1 << 31)) return fail(MI_ERR_INVALID, "mi_cubemap_upload: cubemap larger than 2 GiB");
    }
    c->sc.cube = nullptr;
    HIPCHK(c->cube.ensure(128 + total));
    HIPCHK(hipMemcpy(c->cube.p, header, 128, hipMemcpyHostToDevice));
    for (int f = 0; f < 6; f++)
        HIPCHK(hipMemcpy((char *)c->cube.p + 128 + header[3 * f + 2], cm->rgb[f], (size_t)cm->width[f] * cm->height[f] * 3, hipMemcpyHostToDevice));
    c->sc.cube = c->cube.as<int32_t>();
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// mi_group: ONE process driving several GPUs behind the Tracer interface.  One mi_ctx per device,
// scene replicated, device k renders shard k of n (tile k mod n); the tonemapped RGBA8 tiles travel to the first
// device over RCCL (grouped ncclSend / ncclRecv on a communicator from ncclCommInitAll: device-to-device over
// xGMI, no host staging and no fallback), where k_assemble scatters them into the frame.  RCCL is loaded at run
// time (dlopen) when the first group is created: single-GPU users never load it, and a host process that already
// carries an RCCL of its own (PyTorch does) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>
#include <thread>

namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

mi_status rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return MI_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(MI_ERR_HIP, "mi_group: librccl.so not found (%s)", dlerror());
    RcclApi a;
    a.lib = h;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv || !a.GetErrorString) {
        dlclose(h);
        return fail(MI_ERR_HIP, "mi_group: librccl.so lacks a required symbol");
    }
    g_rccl = a;
    return MI_OK;
}
} // namespace

#define NCCLCHK(call)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (call);                                                                                  \
        if (r_ != ncclSuccess) return fail(MI_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct mi_group {
    std::vector<int> devices;
    std::vector<mi_ctx *> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<DevBuf> rgba8;   // per device: its shard's RGBA8 tiles
    DevBuf gathered;             // first device: every shard's tiles
};

MI_API void mi_group_destroy(mi_group *g) {
    if (!g) return;
    for (size_t k = 0; k < g->ctx.size(); k++) {
        (void)hipSetDevice(g->devices[k]);
        if (k < g->comms.size() && g->comms[k]) (void)g_rccl.CommDestroy(g->comms[k]);
        if (k < g->rgba8.size()) g->rgba8[k].release();
        if (k == 0) g->gathered.release();
        mi_ctx_destroy(g->ctx[k]);
    }
    delete g;
}

MI_API mi_status mi_group_create(const int *device_ids, int n_dev, mi_group **out) {
    if (!out) return fail(MI_ERR_INVALID, "mi_group_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n_dev <= 0 || n_dev > 64) return fail(MI_ERR_INVALID, "mi_group_create: need 1..64 device ids");
    for (int a = 0; a < n_dev; a++)
        for (int b = a + 1; b < n_dev; b++)
            if (device_ids[a] == device_ids[b]) return fail(MI_ERR_INVALID, "mi_group_create: device %d listed twice", device_ids[a]);
    mi_group *g = new mi_group;
    g->devices.assign(device_ids, device_ids + n_dev);
    g->rgba8.resize(n_dev);
    for (int k = 0; k < n_dev; k++) {
        mi_ctx *c = nullptr;
        mi_status st = mi_ctx_create(device_ids[k], &c);
        if (st != MI_OK) { mi_group_destroy(g); return st; }
        g->ctx.push_back(c);
    }
    mi_status ls = rccl_load();
    if (ls != MI_OK) { mi_group_destroy(g); return ls; }
    g->comms.assign(n_dev, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_dev, g->devices.data());
    if (r != ncclSuccess) {
        g->comms.clear();
        mi_group_destroy(g);
        return fail(MI_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s", n_dev, g_rccl.GetErrorString(r));
    }
    *out = g;
    return MI_OK;
}

MI_API int mi_group_size(const mi_group *g) { return g ? (int)g->ctx.size() : 0; }

MI_API mi_status mi_group_scene_upload(mi_group *g, const mi_scene_soa *scene) {
    if (!g) return fail(MI_ERR_INVALID, "mi_group_scene_upload: group is NULL");
    for (mi_ctx *c : g->ctx) { // replicated: the scene is read-only and small against 288 GB
        mi_status st = mi_scene_upload(c, scene);
        if (st != MI_OK) return st;
    }
    return MI_OK;
}

MI_API mi_status mi_group_scene_upload_build(mi_group *g, const mi_scene_soa *scene, const float *tri_verts, mi_build_stats *stats) {
    if (!g) return fail(MI_ERR_INVALID, "mi_group_scene_upload_build: group is NULL");
    for (size_t k = 0; k < g->ctx.size(); k++) { // every device builds its own copy: the builders are deterministic
        mi_status st = mi_scene_upload_build(g->ctx[k], scene, tri_verts, k == 0 ? stats : nullptr);
        if (st != MI_OK) return st;
    }
    return MI_OK;
}

MI_API mi_status mi_group_cubemap_upload(mi_group *g, const mi_cubemap *cm) {
    if (!g) return fail(MI_ERR_INVALID, "mi_group_cubemap_upload: group is NULL");
    for (mi_ctx *c : g->ctx) {
        mi_status st = mi_cubemap_upload(c, cm);
        if (st != MI_OK) return st;
    }
    return MI_OK;
}

MI_API mi_status mi_group_render(mi_group *g, const mi_render_params *p, uint8_t *rgb8_out, mi_stats *stats) {
    if (!g || !p || !rgb8_out) return fail(MI_ERR_INVALID, "mi_group_render: NULL argument");
    if (p->shard_count > 1) return fail(MI_ERR_INVALID, "mi_group_render: the group shards the frame itself (shard_count must be 0 or 1)");
    const int n = (int)g->ctx.size();
    const auto t0 = std::chrono::steady_clock::now();
    mi_render_params p0 = *p;
    p0.shard_rank = 0;
    p0.shard_count = n;
    const int64_t stride = mi_shard_pixels(&p0); // shard 0 owns the most tiles
    if (stride < 0) return MI_ERR_INVALID;      // make_geom has set the message
    // every device renders its shard on a host thread of its own (mi_render_device blocks until its stream is done)
    std::vector<mi_status> st(n, MI_OK);
    std::vector<std::string> err(n);
    std::vector<mi_stats> ds(n);
    for (int k = 0; k < n; k++) {
        HIPCHK(hipSetDevice(g->devices[k]));
        HIPCHK(g->rgba8[k].ensure(std::max<size_t>((size_t)stride * 4, 16)));
    }
    HIPCHK(hipSetDevice(g->devices[0]));
    HIPCHK(g->gathered.ensure(std::max<size_t>((size_t)stride * 4 * n, 16)));
    {
        std::vector<std::thread> th;
        for (int k = 0; k < n; k++)
            th.emplace_back([&, k] {
                mi_render_params pk = *p;
                pk.shard_rank = k;
                pk.shard_count = n;
                st[k] = mi_render_device(g->ctx[k], &pk, nullptr, g->rgba8[k].p, nullptr, &ds[k]);
                if (st[k] != MI_OK) err[k] = mi_last_error(); // the message is per thread
            });
        for (auto &t : th) t.join();
    }
    for (int k = 0; k < n; k++)
        if (st[k] != MI_OK) return fail(st[k], "mi_group_render: device %d: %s", g->devices[k], err[k].c_str());
    // exchange: RGBA8 tiles of shards 1..n-1 -> first device, one grouped send/recv pair per shard; shard 0 is already there
    const size_t bytes = (size_t)stride * 4;
    const auto t1 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(g->devices[0]));
    HIPCHK(hipMemcpyAsync(g->gathered.p, g->rgba8[0].p, bytes, hipMemcpyDeviceToDevice, g->ctx[0]->stream));
    NCCLCHK(g_rccl.GroupStart());
    ncclResult_t xr = ncclSuccess; // a call that fails inside the group must not leave the group open
    hipError_t xe = hipSuccess;
    for (int k = 1; k < n && xr == ncclSuccess && xe == hipSuccess; k++) {
        if ((xe = hipSetDevice(g->devices[k])) != hipSuccess) break;
        if ((xr = g_rccl.Send(g->rgba8[k].p, bytes, ncclUint8, 0, g->comms[k], g->ctx[k]->stream)) != ncclSuccess) break;
        if ((xe = hipSetDevice(g->devices[0])) != hipSuccess) break;
        xr = g_rccl.Recv((char *)g->gathered.p + (size_t)k * bytes, bytes, ncclUint8, k, g->comms[0], g->ctx[0]->stream);
    }
    const ncclResult_t ge = g_rccl.GroupEnd();
    if (xe != hipSuccess) return fail(MI_ERR_HIP, "mi_group_render: hipSetDevice failed in the exchange: %s", hipGetErrorString(xe));
    if (xr != ncclSuccess) return fail(MI_ERR_HIP, "mi_group_render: ncclSend/ncclRecv failed: %s", g_rccl.GetErrorString(xr));
    NCCLCHK(ge);
    for (int k = 1; k < n; k++) { // the senders' buffers are free again when their streams are done
        HIPCHK(hipSetDevice(g->devices[k]));
        HIPCHK(hipStreamSynchronize(g->ctx[k]->stream));
    }
    HIPCHK(hipSetDevice(g->devices[0]));
    HIPCHK(hipStreamSynchronize(g->ctx[0]->stream));
    const auto t2 = std::chrono::steady_clock::now();
    mi_status as = mi_assemble_rgb8(g->ctx[0], &p0, g->gathered.p, stride, rgb8_out);
    if (as != MI_OK) return as;
    const auto t3 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int k = 0; k < n; k++) {
            stats->rays_cast += ds[k].rays_cast; stats->paths += ds[k].paths; stats->rays_extend += ds[k].rays_extend;
            stats->extend_launches += ds[k].extend_launches; stats->batches += ds[k].batches;
            stats->ms_extend = std::max(stats->ms_extend, ds[k].ms_extend); // of the slowest device
            stats->ms_shade = std::max(stats->ms_shade, ds[k].ms_shade);
            stats->ms_tail = std::max(stats->ms_tail, ds[k].ms_tail);
        }
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        stats->ms_total = ms(t0, t3);                      // wall clock of the whole call, exchange and copy to the host included
        stats->ms_other = ms(t1, t2) + ms(t2, t3);         // exchange + assembly
    }
    return MI_OK;
}
