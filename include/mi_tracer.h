/*
 * mi_tracer.h — C-ABI of the MI355X-native tracer backend (libmi_tracer.so).
 *
 * This is the drop-in boundary that sits UNDER a `Tracer` subclass (reference
 * interface: src/tracer.h:19-52, pure virtuals traceImage/waitRender at :35-37).
 * The reference's GPU backend crosses its device boundary inside
 * GpuTracer::traceImage (src/GpuTracer.cu:161-189) with C++ objects deep-copied
 * by Scene::clone (src/scene/scene.cu:3-13) and
 * BoundedVolumeHierarchy::flatten_clone (src/scene/bvh.cu:395-415).  Here that
 * boundary is a plain C interface: POD structs, borrowed host pointers, sizes,
 * status codes.  No C++ types, no exceptions, no torch types cross it.
 *
 * All colour/vector data is fp32.  "float4" below means 4 consecutive floats.
 *
 * Threading: a context owns its device buffers and is used by one call at a time (the
 * reference's caller is single-threaded, CommandLineUI.cu:82-94); different contexts,
 * also on the same GPU, are independent.  Error strings are per thread.
 */
#ifndef MI_TRACER_H
#define MI_TRACER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx mi_ctx;

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_INVALID = 1,   /* bad argument (NULL, zero size, inconsistent scene) */
    MI_ERR_NO_DEVICE = 2, /* no usable HIP device / bad device id              */
    MI_ERR_HIP = 3,       /* a hip* call failed; see mi_last_error()           */
    MI_ERR_NO_SCENE = 4,  /* render called before mi_scene_upload              */
    MI_ERR_OOM = 5        /* device allocation failed                          */
} mi_status;

/*
 * Flattened scene, structure-of-arrays.  Replaces what the reference uploads
 * with Scene::clone / flatten_clone (src/scene/scene.cu:3-13,
 * src/scene/bvh.cu:395-415): Cluster[2N-1] AoS nodes of 64 B
 * (src/scene/bvh.h:19-34) and Geometry[N] AoS records of 480 B
 * (src/scene/geometry.h:11-48).
 *
 * BVH: binary, one primitive per leaf like the reference's (bvh.cu:72-76), but
 * only INTERIOR nodes are stored; a node holds the boxes of both children (the
 * reference also tests both child boxes at the parent, bvh.cu:7-8,190-191).
 *   node = 16 floats (64 B):
 *     [0..2] left  bmin   [3..5]  left  bmax
 *     [6..8] right bmin   [9..11] right bmax
 *     [12] left  child ref (int32 bits)   [13] right child ref (int32 bits)
 *     [14],[15] reserved (0)
 *   child ref >= 0 : index of an interior node
 *   child ref <  0 : leaf holding primitive slot ~ref
 * Primitive slots: [0, n_tri) are triangles, [n_tri, n_tri+n_sph) are spheres
 * (slot - n_tri indexes the sphere arrays).
 * The library collapses this binary tree into the 4-wide tree its kernels walk (a node's
 * seats are its grandchildren; 64-B nodes, child boxes as 16-bit plane codes rounded outward)
 * when the scene is uploaded; the caller never sees that form.  Coordinates must be finite
 * and smaller than 1e15 in magnitude.
 * root_ref: ref of the root (an interior node index, a leaf ref for a
 * one-primitive scene — bvh.cu:182-183 — or ignored when n_tri+n_sph == 0).
 */
typedef struct mi_scene_soa {
    uint32_t n_nodes;
    const float *nodes;           /* n_nodes * 16 floats                            */
    int32_t root_ref;

    uint32_t n_tri;
    const float *tri_v0;          /* float4[n_tri]  vertex 0   (trimesh.cu:33,88)   */
    const float *tri_uinv;        /* float4[n_tri]  uInv       (trimesh.cu:37)      */
    const float *tri_vinv;        /* float4[n_tri]  vInv       (trimesh.cu:38)      */
    const float *tri_ninv;        /* float4[n_tri]  nInv       (trimesh.cu:39)      */
    const float *tri_normal;      /* float4[n_tri]  face normal (trimesh.cu:24,36)  */
    const uint32_t *tri_mat;      /* uint32[3*n_tri] material ids of vertex a,b,c;
                                     all three equal when the mesh has no per-vertex
                                     materials (trimesh.cu:32,40-45)                */
    const uint8_t *tri_has_vmat;  /* uint8[n_tri]   hasMaterials (trimesh.cu:41)    */

    uint32_t n_sph;
    const float *sph_center_radius; /* float4[n_sph] (cx,cy,cz,radius) sphere.h:13-15 */
    const uint32_t *sph_mat;        /* uint32[n_sph]                                 */

    uint32_t n_mat;
    const float *materials;       /* n_mat * 20 floats: ke,kd,kr,kt,index as float4
                                     each (material.h:129-133)                      */

    const int32_t *prim_id;       /* int32[n_tri+n_sph] stable primitive id (parse
                                     order) of each slot; used to break exact-t ties
                                     independently of BVH order                     */

    float cam_eye[4];             /* camera.h:33-36 / camera.cu:60-65               */
    float cam_u[4];
    float cam_v[4];
    float cam_look[4];
} mi_scene_soa;

/* Render parameters.  samples_per_dim and depth have the reference's meaning:
 * -s N -> N*N paths per pixel on a regular grid (CpuTracer.cu:99-109);
 * -r D -> a path casts up to D+2 rays (CpuTracer.cu:119-131).                      */
typedef struct mi_render_params {
    int32_t width;
    int32_t height;
    int32_t samples_per_dim;
    int32_t depth;
    uint64_t seed;        /* key of the counter-based RNG (replaces the unseedable
                             mt19937 of CpuTracer.cu:24-35)                         */
    int32_t tile_w;       /* shard tile size in pixels; 0 -> 32 (CpuTracer.cu:55)   */
    int32_t tile_h;
    int32_t shard_rank;   /* this context renders tiles k with k % shard_count ==   */
    int32_t shard_count;  /*   shard_rank (row-major tile index); 0/1 -> everything */
    int64_t batch_paths;  /* max paths in flight per wavefront batch; 0 -> default:
                             what a quarter of the GPU's memory holds, at most 256 Mi;
                             the batches of a frame are made equal in size          */
    int32_t count_work;   /* != 0: run the instrumented extend kernel that counts node
                             visits and primitive tests (slower; for roofline
                             accounting, never inside a timed region)                */
    int32_t flags;        /* MI_RENDER_* bits, 0 = defaults                            */
} mi_render_params;
/* The batches of a frame normally alternate between two streams, each with its own path state (a frame that
 * fits one batch is cut in two), so that the sparse end of one batch runs beside the dense start of the next;
 * the image does not depend on it.  This bit keeps every batch on the one stream of the call: each kernel then
 * has the chip to itself, which is what a per-kernel measurement wants. */
#define MI_RENDER_ONE_STREAM 1
/* Per-kernel times (mi_stats.ms_extend / ms_shade / ms_tail / extend_launches) need an event around every launch:
 * 4 per level and batch, ~0.15 ms of a 6.5 ms frame.  They are recorded only on request (or with
 * MI_TRACER_DEBUG_LEVELS in the environment); without this bit those fields are 0 and ms_other = ms_total. */
#define MI_RENDER_LAUNCH_TIMES 2
/* mi_render_device only, with stats == NULL: the frame is ENQUEUED on `stream` and the call returns without waiting for it
 * (a caller that renders frame after frame — an animation, a progressive viewer, bench.py's timed loop — keeps the GPU
 * busy between frames instead of paying a host round trip per frame: ~0.16 ms of a 6-ms frame).  The outputs are complete
 * when `stream` has run dry (hipStreamSynchronize / an event of the caller's); frames enqueued on one stream run in order.
 * No counterpart in the reference (GpuTracer.cu:180 synchronises the device after its one kernel).  MI_ERR_INVALID with
 * stats != NULL or with count_work. */
#define MI_RENDER_ENQUEUE_ONLY 4

typedef struct mi_stats {
    uint64_t rays_cast;      /* closest-hit queries issued (primary + bounce)       */
    uint64_t paths;          /* paths started                                       */
    double ms_total;         /* first launch -> outputs complete (HIP events)       */
    double ms_extend;        /* sum over launches of the BVH extend kernel (MI_RENDER_LAUNCH_TIMES) */
    double ms_shade;         /* sum over launches of the shade+compact kernel       */
    double ms_other;         /* everything else (resolve, memsets, gaps)             */
    uint32_t extend_launches;
    uint32_t batches;
    uint64_t node_visits;    /* only with count_work: interior nodes fetched          */
    uint64_t prim_tests;     /* only with count_work: primitive records tested        */
    uint64_t rays_extend;    /* rays traced by the per-level extend kernel (the rest,
                                rays_cast - rays_extend, by the tail kernel)           */
    double ms_tail;          /* sum over launches of the tail kernel                   */
} mi_stats;

/* Create a context on HIP device `device_id`.  One context = one GPU; multi-GPU
 * jobs use one process (and one context) per GPU.  Replaces the implicit device-0
 * state of GpuTracer (GpuTracer.cu:165-181). */
mi_status mi_ctx_create(int device_id, mi_ctx **out);

/* Free all device memory of the context (the reference never frees; GpuTracer.cu). */
void mi_ctx_destroy(mi_ctx *ctx);

/* Copy the flattened scene to the device.  Host pointers are borrowed for the call
 * only.  Replaces Scene::clone (scene.cu:3-13) + flatten_clone (bvh.cu:395-415). */
mi_status mi_scene_upload(mi_ctx *ctx, const mi_scene_soa *scene);

/* Same as mi_scene_upload, but the BVH is BUILT ON THE DEVICE: scene->nodes / n_nodes /
 * root_ref are ignored, and tri_verts gives the three vertices of every triangle slot (float[9*n_tri], HOST; the boxes of
 * geometry.h:13-29 need them — the precomputed uInv/vInv/nInv do not contain them).
 * Replaces the host-side BoundedVolumeHierarchy::construct + flatten (bvh.cu:66-172,
 * 239-369) with a top-down binned-SAH build (16 bins, three axes: the algorithm of this project's host
 * builder carried out level by level on the GPU; it arrives at the same tree, node for node.  MI_TRACER_BUILDER=ploc
 * in the environment selects parallel locally-ordered clustering over the Morton order instead — half the build time,
 * trees that render 1-4 % slower — and =lbvh a plain Karras LBVH).  The
 * primitive records are gathered into tree order on the device: the slots the resident tree
 * refers to are NOT the caller's (mi_scene_download_prim_ids maps them to primitive ids).  The rendered image does not depend on which tree is used (exact-t ties
 * are resolved by primitive id and child boxes are padded, see mi_scene_soa.prim_id), so
 * renders with this tree are bit-identical to renders with an uploaded one.  A tree
 * deeper than 64 levels is rejected (MI_ERR_INVALID).  stats may be NULL. */
typedef struct mi_build_stats {
    double ms_build;     /* device time of the build kernels (HIP events)               */
    double sah_cost;     /* 1 + sum of child-box areas / root area (unit costs)          */
    int32_t depth;       /* levels of interior nodes                                     */
    uint32_t n_nodes;
} mi_build_stats;
mi_status mi_scene_upload_build(mi_ctx *ctx, const mi_scene_soa *scene, const float *tri_verts,
                                mi_build_stats *stats);

/* SAH cost (same definition as mi_build_stats.sah_cost, on the padded boxes) of the tree
 * resident in the context, whichever way it got there.  The reference reports BVH quality
 * only as render time (report 5.2); no counterpart. */
mi_status mi_scene_quality(mi_ctx *ctx, double *sah_cost_out);

/* Copy the resident tree back (n_nodes*16 floats in the mi_scene_soa node format), e.g.
 * to validate or to hand a device-built tree to the CPU checker.  nodes_out == NULL only
 * queries n_nodes / root_ref.  No counterpart in the reference. */
mi_status mi_scene_download_nodes(mi_ctx *ctx, float *nodes_out, uint32_t capacity_nodes,
                                  uint32_t *n_nodes_out, int32_t *root_ref_out);

/* Copy the tree the kernels actually walk, for inspection and tests: per 4-wide node (the tree, then the
 * treelet copies) its 32 fp32 words `lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] refs[4] pad[4]` into
 * nodes4_out and its 16 quantised words `px[4] py[4] pz[4] refs[4]` into qnodes_out (either may be NULL);
 * grid_out[0..2] = qs, grid_out[3..5] = qb: the plane of a 16-bit code q on axis a is
 * qs[a] * F + qb[a] with F the float whose bits are 0x3F000000 | q << 8 (low half of a packed word: lower
 * plane, high half: upper plane).  n_nodes_out / root_ref_out / treelet_base_out may be NULL.  No
 * counterpart in the reference. */
mi_status mi_scene_download_nodes4(mi_ctx *ctx, float *nodes4_out, uint32_t *qnodes_out, uint32_t capacity_nodes,
                                   uint32_t *n_nodes_out, int32_t *root_ref_out, int32_t *treelet_base_out, float grid_out[6]);

/* prim_id of every slot of the resident scene (the caller's array after mi_scene_upload, the
 * permuted one after mi_scene_upload_build).  No counterpart in the reference. */
mi_status mi_scene_download_prim_ids(mi_ctx *ctx, int32_t *prim_id_out, uint32_t capacity);

/* Render (this shard of) the image: the whole of GpuTracer::traceImage
 * (GpuTracer.cu:161-189) incl. the host setPixel loop (:186-188, tracer.cu:95-100).
 *   rgba_mean_out : float4[width*height] mean clamped radiance per pixel (what
 *                   setPixel receives), row 0 = bottom row; may be NULL.
 *   rgb8_out      : uint8[width*height*3] tonemapped bytes at (i + j*width)*3
 *                   (tracer.cu:96); may be NULL.
 * Pixels of tiles this shard does not own are left untouched in both buffers.
 * Synchronous.  Both outputs are HOST pointers. */
mi_status mi_render(mi_ctx *ctx, const mi_render_params *params,
                    float *rgba_mean_out, uint8_t *rgb8_out, mi_stats *stats);

/* Same render, outputs left in DEVICE memory in shard-local order (for an RCCL
 * gather without a host round-trip).  Local pixel q of the shard = owned tile
 * q / (tile_w*tile_h) (j-th owned tile is global tile shard_rank + j*shard_count),
 * offset q % (tile_w*tile_h) inside the tile, row-major.
 *   d_rgba_mean : device float4[mi_shard_pixels()] or NULL
 *   d_rgba8     : device uint8[4*mi_shard_pixels()] (r,g,b,255) or NULL
 * Pixels of edge tiles that fall outside the image are written as zeros.
 * `stream` is a hipStream_t; NULL selects the context's OWN stream (created with
 * hipStreamNonBlocking: it is ordered with no other stream, not with the legacy default
 * stream either).  The call enqueues its work on that stream and synchronises it before
 * returning, so the outputs are complete on return.  The context's buffers and the caller's
 * outputs are ordered only with the stream passed in: work of the caller that still reads
 * d_rgba8 / d_rgba_mean on another stream (e.g. a collective) must have finished, or be
 * ordered before this stream, when the next call is made. */
mi_status mi_render_device(mi_ctx *ctx, const mi_render_params *params,
                           void *d_rgba_mean, void *d_rgba8, void *stream,
                           mi_stats *stats);

/* mi_render_device for callers that CANNOT ALLOCATE: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  NO counterpart in the reference (its GpuTracer::traceImage
 * allocates on every call, GpuTracer.cu:165-176).  Same arguments, same outputs, bit for bit; the call
 * makes no hipMalloc / hipFree / event or stream creation: all per-batch path state (ray streams, hits,
 * history, path values, counters, traversal overflow stacks) lives in one static device array of
 * mi_noalloc_workspace_bytes() = 64 MiB that every GPU running this library holds until the process ends.
 * The frame is rendered in as many batches as that takes (C3 at 1080p: 252 batches, 87 ms against
 * 6 ms through mi_render_device: it is the entry point for those who cannot call the other).
 * Limits: MI_ERR_INVALID with a message naming the sizes when even the smallest batch (64 pixels x
 * samples_per_dim^2 paths at `depth`, plus the overflow stacks the resident tree needs) does not fit —
 * e.g. depth 5 allows samples_per_dim up to ~40, depth 62 up to ~16; count_work is not supported.
 * stats != NULL: the stream is synchronised and *stats filled (ms_total, rays_cast, paths, batches; no
 * per-kernel times).  stats == NULL: the work is only ENQUEUED on `stream` (kernel launches and
 * hipMemsetAsync, nothing else) and the call returns without synchronising.
 * Every context on a GPU shares that GPU's array, and a call finds the previous call's contents in it
 * (what must start at zero is cleared on the stream): calls that overlap on one GPU — two contexts, two
 * streams, or a call made while an un-synchronised earlier one is still running on another stream —
 * RACE.  Calls on the same stream are ordered and safe.
 * Lifetime: the enqueued kernels read the context's scene until they have run.  mi_ctx_destroy, mi_scene_upload,
 * mi_scene_upload_build and mi_cubemap_upload therefore wait for the whole DEVICE (hipDeviceSynchronize) before
 * they release or replace anything — never call them between hipStreamBeginCapture and hipStreamEndCapture, and
 * do not launch a captured graph of this call after its context was destroyed or its scene replaced.
 * Verified under capture: tests/test_zz_no_alloc.py captures the stats == NULL mode with hipStreamBeginCapture
 * (through the host framework's graph API), replays the graph twice and compares with mi_render_device; C3 at 1080p through this entry
 * point: see DESIGN.md ("The entry point that allocates nothing"). */
mi_status mi_render_device_noalloc(mi_ctx *ctx, const mi_render_params *params,
                                   void *d_rgba_mean, void *d_rgba8, void *stream,
                                   mi_stats *stats);

/* Size in bytes of the static workspace of mi_render_device_noalloc (per GPU).  No counterpart
 * in the reference. */
int64_t mi_noalloc_workspace_bytes(void);

/* Pinned (page-locked, DMA-able) host memory for the buffers the caller hands to mi_render / mi_group_render again
 * and again — the Tracer's RGB8 frame `buffer` (tracer.h:47, tracer.cu:87-93).  A frame rendered into such a buffer
 * arrives by one DMA behind the last kernel; into ordinary pageable memory it goes through the runtime's staging
 * copies (C3 at 1080p, 6.2 MB: DESIGN.md section 6 has both rates).  Replaces nothing in the reference
 * (GpuTracer.cu:181 copies w*h float4 to pageable memory and converts on the host, :186-188).  The memory is plain
 * host memory to the CPU; release it with mi_host_free (NULL is ignored) after the last render into it has
 * returned.  MI_ERR_OOM / MI_ERR_HIP if the runtime cannot provide it: fall back to an ordinary buffer. */
mi_status mi_host_alloc(size_t bytes, void **out);
void mi_host_free(void *ptr);

/* Number of local pixel slots (owned tiles * tile_w * tile_h) of a shard. */
int64_t mi_shard_pixels(const mi_render_params *params);

/* Scatter gathered shard-local RGBA8 buffers (shard r at d_gathered +
 * r*4*shard_pixels_max bytes, each laid out as mi_render_device writes them) into a
 * bottom-up RGB8 frame on the HOST (tracer.cu:96 layout).  Used by rank 0 after the
 * gather.  d_gathered is a DEVICE pointer. */
mi_status mi_assemble_rgb8(mi_ctx *ctx, const mi_render_params *params,
                           const void *d_gathered, int64_t shard_stride_pixels,
                           uint8_t *rgb8_out);

/* Same scatter with the frame left in DEVICE memory (d_rgb8_out: uint8[width*height*3]),
 * so that a multi-GPU render ends like a single-GPU mi_render_device: everything in HBM.
 * Enqueued on `stream` (NULL = the context's stream) and synchronised before returning. */
mi_status mi_assemble_rgb8_device(mi_ctx *ctx, const mi_render_params *params,
                                  const void *d_gathered, int64_t shard_stride_pixels,
                                  void *d_rgb8_out, void *stream);

/* Primary-ray closest hit only (parity probe; the reference has no such entry —
 * it corresponds to Camera::rayThrough (camera.cu:6-16) + Scene::intersect
 * (scene.h:70-72) per sample of CpuTracer::tracePixel (CpuTracer.cu:99-109)).
 * Outputs are HOST arrays of width*height*samples_per_dim^2 entries, index
 * (j*width + i)*spp + (sx*samples_per_dim + sy):
 *   prim_id_out : stable primitive id, -1 on miss
 *   t_out       : hit distance (ISECT_NO_HIT 12345678.9f on miss, ray.h:14)
 *   n_out       : float3 geometric normal (0 on miss); may be NULL              */
mi_status mi_trace_primary(mi_ctx *ctx, int32_t width, int32_t height,
                           int32_t samples_per_dim, int32_t *prim_id_out,
                           float *t_out, float *n_out);

/* Closest hit for caller-supplied rays (parity probe for bounce rays).
 * rays: float[n*6] (ox,oy,oz,dx,dy,dz), HOST.  Outputs as mi_trace_primary. */
mi_status mi_trace_rays(mi_ctx *ctx, int64_t n, const float *rays,
                        int32_t *prim_id_out, float *t_out, float *n_out);

/* Any-hit (occlusion) query: occluded_out[i] = 1 iff some primitive's intersect() accepts
 * ray i when the running closest distance starts at tmax[i] (triangle: 0 < t <= tmax,
 * trimesh.cu:92-96; sphere: the accepted root < tmax, sphere.cu:17-28), 0 otherwise.  The
 * traversal stops at the first accepted primitive.  tmax above ISECT_NO_HIT (12345678.9f,
 * ray.h:14 — the distance every reference query starts with) or NaN is clamped to it: with an
 * infinite bound the triangle test would accept t = inf of plane-parallel rays.  NO counterpart in the reference, which
 * has no shadow rays (lights are emissive geometry, Parser.cu:650-651); it is the "shadow
 * any-hit" of the north star, outside the parity renderer, checked against the oracle's
 * test of every primitive.  rays: float[n*6] HOST, tmax: float[n] HOST, out: uint8[n] HOST. */
mi_status mi_occluded_rays(mi_ctx *ctx, int64_t n, const float *rays, const float *tmax,
                           uint8_t *occluded_out);

/* Cubemap background: the colour of a ray that hits nothing (CpuTracer.cu:125,
 * `cubeMap ? cubeMap->getColor(r) : 1`; CubeMap::getColor cubeMap.cu:6-31 with
 * TextureMap::getMappedValue / getPixelAt, material.cu:20-37: major-axis face, bilinear
 * fetch, texel / 256, clamp to edge).  CPU-only in the reference; on the device here
 * (SURVEY 8 f4).  Faces in the order of TraceUI.cu:76-83 (+x, -x, +y, -y, +z, -z), tightly
 * packed RGB8, row 0 = BOTTOM row of the image file (what readImage returns,
 * pngimage.cu:210-213, bitmap.cu:70-93).  HOST pointers, copied.  The cubemap belongs to
 * the context, not to the scene (the reference's UI owns it, TraceUI.h:78): it stays until
 * replaced; cm == NULL removes it (white background again). */
typedef struct mi_cubemap {
    int32_t width[6];
    int32_t height[6];
    const uint8_t *rgb[6];
} mi_cubemap;
mi_status mi_cubemap_upload(mi_ctx *ctx, const mi_cubemap *cm);

/* Several GPUs in ONE process behind the Tracer interface ("one ctx = one process, N devices"; no
 * counterpart in the reference, which drives one device, GpuTracer.cu:165-181).  A group holds one context per
 * device, the scene replicated on each; mi_group_render makes device k render shard k of n (tiles k mod n, as
 * mi_render_params.shard_* describe), moves the tonemapped RGBA8 tiles to the first device with RCCL (grouped
 * ncclSend / ncclRecv on an ncclCommInitAll communicator: device to device over xGMI, no host staging), assembles
 * the frame there and copies it to rgb8_out (HOST, uint8[width*height*3], tracer.cu:96 layout).  The image is
 * bit-identical to a one-device render for any n.  librccl.so is loaded when the first group is created.
 * stats: rays / paths summed over the devices, kernel times of the slowest device, ms_total = wall clock of the
 * call, ms_other = exchange + assembly.  params->shard_count must be 0 or 1.
 * Verified on hardware with n = 1 only (one GPU per lease): the n > 1 exchange is the same code, unverified. */
typedef struct mi_group mi_group;
mi_status mi_group_create(const int *device_ids, int n_dev, mi_group **out);
void mi_group_destroy(mi_group *group);
int mi_group_size(const mi_group *group);
mi_status mi_group_scene_upload(mi_group *group, const mi_scene_soa *scene);
mi_status mi_group_scene_upload_build(mi_group *group, const mi_scene_soa *scene, const float *tri_verts,
                                      mi_build_stats *stats);
mi_status mi_group_cubemap_upload(mi_group *group, const mi_cubemap *cm);
mi_status mi_group_render(mi_group *group, const mi_render_params *params, uint8_t *rgb8_out, mi_stats *stats);

/* Thread-local description of the last error returned on this thread. */
const char *mi_last_error(void);

/* Library/ABI version (bumped on any struct change). */
int mi_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_TRACER_H */
