/*
 * vkr_rt.h -- C ABI of the MI355X-native RTAO / shadow-ray path.
 *
 * This is the drop-in boundary for ONE hot path of chenjunjia-java/VulkanRenderer:
 *   RtaoComputePass (trace+temporal -> A-Trous xN -> upsample) + the pbr.frag shadow rays,
 *   with the Vulkan TLAS/BLAS + rayQueryEXT replaced by a HIP LBVH.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository root).  The reference-side binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures: a stream is passed as `void*`
 *     (a hipStream_t; NULL = the default stream).  Work is enqueued asynchronously on that
 *     stream, like commands recorded into the reference's single command buffer
 *     (Renderer.cpp:330-359); vkr_sync() is the fence.
 *   - every function returns a vkr_status (0 = OK) and never throws; the reference throws
 *     std::runtime_error (RayTracingContext.cpp:49-53,59,77; Rendergraph.cpp:95,120) -- the C++
 *     adapter (include/vkr_pass.hpp) turns a non-zero status back into std::runtime_error.
 *   - a context is NOT thread-safe (reference: single render thread, VulkanApplication.cpp:56-69).
 *   - images are tightly packed row-major arrays in DEVICE memory:
 *       R16F    -> uint16_t[W*H]      (IEEE binary16 bits)
 *       RGBA16F -> uint16_t[W*H*4]
 *   - "host" pointers are read during the call and may be freed when it returns.
 */
#ifndef VKR_RT_H
#define VKR_RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define VKR_API
#else
#define VKR_API __attribute__((visibility("default")))
#endif

typedef enum vkr_status {
    VKR_OK = 0,
    VKR_ERR_INVALID_ARG = 1, /* null pointer, bad size, count mismatch            */
    VKR_ERR_STATE = 2,       /* call order (e.g. trace before bvh_build)          */
    VKR_ERR_HIP = 3,         /* a HIP runtime call failed; see vkr_last_error()   */
    VKR_ERR_OOM = 4,         /* device allocation failed                          */
    VKR_ERR_LIMIT = 5,       /* a fixed capacity was exceeded (stack depth, ...)  */
    VKR_ERR_RCCL = 6         /* the attached communicator / all-gather reported an error */
} vkr_status;

typedef struct vkr_ctx vkr_ctx;

/* AppConfig::MAX_REFLECTION_MATERIAL_COUNT (rtao_trace_half.comp:44, FrameManager.cpp:1355-1357). */
#define VKR_MAX_MATERIALS 256u
/* Floats in PBRUniformBufferObject (VulkanTypes.h:47-62): 3 mat4 + 11 vec4 + 1 mat4 = 480 B. */
#define VKR_UBO_FLOATS 120u

/* Offsets (in floats) of the UBO fields the path reads -- layout of VulkanTypes.h:47-62. */
enum {
    VKR_UBO_MODEL = 0,
    VKR_UBO_VIEW = 16,
    VKR_UBO_PROJ = 32,
    VKR_UBO_DIR_LIGHT_DIR = 48,
    VKR_UBO_DIR_LIGHT_COLOR = 52,
    VKR_UBO_DIR_LIGHT_PARAMS = 56, /* x = sun angular radius (rad), y = soft shadow sample count */
    VKR_UBO_LIGHT_POSITIONS = 60,  /* 3 x vec4 */
    VKR_UBO_LIGHT_COLORS = 72,     /* 3 x vec4 */
    VKR_UBO_CAM_POS = 84,
    VKR_UBO_PARAMS = 88,      /* w = pointLightCount */
    VKR_UBO_IBL_PARAMS = 92,  /* w = debug view mode (9..12 change RTAO outputs) */
    VKR_UBO_RTAO_PARAMS0 = 96,  /* x enable, y rayCount, z radius, w bias */
    VKR_UBO_RTAO_PARAMS1 = 100, /* x strength, y temporalAlpha, z disocclusionThreshold, w frameIndex */
    VKR_UBO_PREV_VIEW_PROJ = 104
};

/* ---------------------------------------------------------------- context -------------- */

typedef struct vkr_create_desc {
    int32_t device;   /* HIP device ordinal                                              */
    uint32_t width;   /* swapchain extent: all RTAO images are full resolution          */
    uint32_t height;  /*   (FrameManager.cpp:996-1002, RtaoComputePass.cpp:505-507)     */
} vkr_create_desc;

/* Replaces: construction of RtaoComputePass + RtaoComputePipeline + the RTAO slice of FrameManager
 * (RtaoComputePass.cpp:16-25, RtaoComputePipeline.cpp:30-75, FrameManager::createRtaoComputeTextures
 * FrameManager.cpp:989-1092).  Allocates history x2, ping-pong x2 (R16F, cleared to 1.0 like
 * FrameManager.cpp:1057) and frame index 0. */
VKR_API int vkr_create(const vkr_create_desc* desc, vkr_ctx** out);
/* Replaces: Renderer::cleanup release order (Renderer.cpp:224-253). Safe on NULL. */
VKR_API void vkr_destroy(vkr_ctx* ctx);
/* Text of the last failure on this context ("" if none). With ctx == NULL: the last vkr_create failure. */
VKR_API const char* vkr_last_error(const vkr_ctx* ctx);
/* The fence: hipStreamSynchronize(stream) (reference: waitForFences, Renderer.cpp:265-266). */
VKR_API int vkr_sync(vkr_ctx* ctx, void* stream);

/* ---------------------------------------------------------------- scene ---------------- */

/* One reference Mesh (= one glTF primitive, GltfModelLoader.cpp:630-717).  HOST pointers. */
typedef struct vkr_mesh_desc {
    const void* positions;     /* float3 at offset 0 of each vertex (Vertex.h: pos first)        */
    uint32_t position_stride;  /* bytes between vertices; 60 = sizeof(Vertex), RayTracingContext.cpp:213-219 */
    uint32_t vertex_count;
    const uint32_t* indices;   /* mesh-local vertex ids; primitiveCount = index_count/3 (RayTracingContext.cpp:230) */
    uint32_t index_count;
    const void* uvs;           /* float2 texCoord per vertex (V already flipped, GltfModelLoader.cpp:649); may be NULL */
    uint32_t uv_stride;        /* bytes between texCoords                                         */
    int32_t material_index;    /* Mesh::materialIndex; out of range => opaque, LUT material 0
                                  (Renderer.cpp:83-86, FrameManager.cpp:1280-1283)                */
    uint32_t opaque;           /* meshOpaqueFlags[i] (Renderer.cpp:80-87): BLAS geometry eOpaque flag
                                  (RayTracingContext.cpp:226-227)                                 */
    const void* normals;       /* float3 vertex normal (Vertex.h: offset 12); may be NULL: the G-buffer producer then
                                  falls back to the triangle's geometric normal.  Only vkr_gbuffer_primary reads it */
    uint32_t normal_stride;    /* bytes between normals                                           */
} vkr_mesh_desc;

/* RayTracingInstanceDesc (RayTracingContext.h:10-13) after toVkTransformMatrix (RayTracingContext.cpp:28-44):
 * 3x4 ROW-major world-from-mesh; customIndex = mesh_index, mask 0xFF, cull disabled (:62-68). */
typedef struct vkr_instance {
    uint32_t mesh_index;
    float transform[12];
} vkr_instance;

/* LOD-0 alpha plane of material m's baseColor texture (rtao_trace_half.comp:170 reads .a only). */
typedef struct vkr_alpha_texture {
    const uint8_t* alpha; /* HOST, width*height unorm8, row 0 = v 0; NULL => alpha 1.0 everywhere */
    uint32_t width;
    uint32_t height;
} vkr_alpha_texture;

typedef struct vkr_scene_desc {
    const vkr_mesh_desc* meshes;
    uint32_t mesh_count;
    const vkr_instance* instances;   /* DFS order of Renderer::rebuildRayTracingInstances (Renderer.cpp:503-527) */
    uint32_t instance_count;
    /* materialParams[256] vec4: x = alphaCutoff, y = alphaMode 0/1/2 (FrameManager.cpp:1354-1371). HOST, 1024 floats. */
    const float* material_params;
    /* baseColorArray[256] (rtao_trace_half.comp:45). HOST array of VKR_MAX_MATERIALS entries, or NULL. */
    const vkr_alpha_texture* alpha_textures;
} vkr_scene_desc;

/* Replaces: RayTracingContext::init inputs (RayTracingContext.cpp:124-137) + FrameManager::createReflectionBuffers
 * (FrameManager.cpp:1229-1381): uploads vertices/indices, builds InstanceLUT {materialID, indexBufferOffset},
 * merged global-id index buffer, merged UV buffer, materialParams, alpha planes. Synchronous. */
VKR_API int vkr_scene_upload(vkr_ctx* ctx, const vkr_scene_desc* scene);

/* Replaces: buildBottomLevelASes + buildTopLevelAS + buildOrUpdateTopLevelAS(eBuild)
 * (RayTracingContext.cpp:199-345,110-122).  Flattens instances to world-space triangles, Morton codes,
 * LDS radix sort, Karras hierarchy, bottom-up AABB fit.  Empty meshes are skipped (:206-208,231-233). */
VKR_API int vkr_bvh_build(vkr_ctx* ctx, void* stream);

/* Replaces: RayTracingContext::updateTopLevelAS (RayTracingContext.cpp:165-197): same topology, new
 * transforms -> re-transform triangles + refit.  `count` must equal the build-time instance count and
 * mesh indices must be unchanged, else VKR_ERR_INVALID_ARG (reference throws "instance count mismatch", :51-53). */
VKR_API int vkr_bvh_refit(vkr_ctx* ctx, const vkr_instance* instances, uint32_t count, void* stream);

/* One Node of the model's scene graph (Node::getLocalMatrix / children / meshIndices), in DFS pre-order -- the order
 * Renderer::rebuildRayTracingInstances visits them (Renderer.cpp:510-524), so parents precede their children. */
typedef struct vkr_scene_node {
    int32_t parent;          /* index of the parent node, -1 for a root node (parent < own index)                 */
    float local[16];         /* Node::getLocalMatrix(), GLM column-major                                          */
    uint32_t first_instance; /* the node's meshIndices became instances [first_instance, first_instance + count)  */
    uint32_t instance_count; /*   of the vkr_scene_desc instance list (one instance per (node, mesh), DFS order)  */
} vkr_scene_node;

/* SURVEY 8(f) row 1.  Replaces the CPU walk of Renderer::rebuildRayTracingInstances (Renderer.cpp:503-527, the
 * reference's documented CPU hot spot): upload the hierarchy once (HOST array, copied) ... */
VKR_API int vkr_hierarchy_upload(vkr_ctx* ctx, const vkr_scene_node* nodes, uint32_t node_count);
/* ... then per frame hand over only the model matrix (GLM column-major).  Mirrors the TLAS cache of
 * Renderer::recordCommandBuffer (Renderer.cpp:470-479): if the matrix equals the one of the last call nothing is
 * done and *refitted = 0; otherwise a kernel recomputes worldFromNode = parentWorld * local down the hierarchy,
 * writes the instance transforms (toVkTransformMatrix, RayTracingContext.cpp:28-44) on the device and refits the
 * BVH like vkr_bvh_refit (RayTracingContext::updateTopLevelAS), *refitted = 1.  `refitted` may be NULL. */
VKR_API int vkr_bvh_refit_model(vkr_ctx* ctx, const float model_matrix[16], int* refitted, void* stream);
/* Copies the current device-side instance transforms (instance_count x 12 floats, 3x4 row-major) to HOST memory:
 * inspection/tests only (synchronises). */
VKR_API int vkr_instances_download(vkr_ctx* ctx, float* out_transforms);

typedef struct vkr_bvh_info {
    uint32_t triangle_count; /* flattened world-space triangles                      */
    uint32_t node_count;     /* wide (4-ary) nodes                                   */
    uint32_t height;         /* levels of wide nodes, root = 1                       */
    uint32_t morton_bits;    /* bits per axis used by the last build (21 or 10)      */
    uint32_t binary_height;  /* height of the intermediate binary LBVH               */
    float bounds_min[3];
    float bounds_max[3];
} vkr_bvh_info;
/* Synchronises the build stream. No reference counterpart (driver-internal AS is opaque). */
VKR_API int vkr_bvh_info_get(vkr_ctx* ctx, vkr_bvh_info* out);

/* Debug/inspection copies of the BVH to HOST memory for structural tests (no reference counterpart:
 * the Vulkan AS is opaque).  nodes: node_count x 16 words (64 B, the record the traversal reads):
 *   w0-2  origin xyz (float)        w3  ex | ey<<8 | ez<<16 | child_count<<24   (biased exponents)
 *   w4-6  qmin x/y/z, one byte per child slot (slot c in bits 8c..8c+7)
 *   w7-9  qmax x/y/z, same packing   w10-11 zero
 *   w12-15 child refs: >= 0 wide node; < 0 leaf, x = ~ref, first = x & 0x0fffffff, count = ((x>>28)&7)+1;
 *          0x7fffffff unused slot (stored with qmin = 255, qmax = 0 so it can never be hit).
 *   slot box: min = origin + qmin * 2^(e-127), max = origin + qmax * 2^(e-127) (conservative).  Root = node 0.
 * tris: triangle_count x 12 floats {v0 xyz, mesh|class<<28 (bits), e1 xyz, prim(bits), e2 xyz, flat id(bits)},
 *   class 0 = commits, 1 = BLEND (never occludes AO), 2 = MASK (alpha test); flat id = position in
 *   instance-major, mesh-triangle-minor order. */
VKR_API int vkr_bvh_download(vkr_ctx* ctx, float* nodes, float* tris);

/* ---------------------------------------------------------------- RTAO pass ------------- */

/* Arguments of one RtaoComputePass::execute = the descriptor set of RtaoComputePipeline.cpp:34-53.
 * Device pointers unless stated.  Binding 1 (HW depth) is bound but never sampled by the shaders: dropped. */
typedef struct vkr_rtao_frame_io {
    float ubo[VKR_UBO_FLOATS];         /* binding 0: PBRUniformBufferObject verbatim (HOST values)          */
    uint32_t atrous_iterations;        /* max(1, denoise ? RTAO_ATROUS_ITERATIONS : 1), RtaoComputePass.cpp:529 */
    const uint16_t* linear_depth_r16f; /* binding 17: linearDepthResolve, 0 = sky (DepthPrepass.cpp:48)      */
    const uint16_t* normal_rgba16f;    /* binding 2: normalResolve, n*0.5+0.5 (depth_only.frag:31-33)         */
    uint16_t* rtao_full_r16f;          /* binding 10: rtaoFull output (FrameManager.cpp:1089-1091)            */
    uint32_t* ray_hit_mask;            /* optional debug tap (may be NULL): bit i = AO ray i ended occluded   */
} vkr_rtao_frame_io;

/* Replaces: RtaoComputePass::render (RtaoComputePass.cpp:102-156): dispatchTrace -> dispatchAtrous (N) ->
 * dispatchUpsample, then flips the history ping-pong index like FrameManager.cpp:1155-1165.
 * UBO semantics are the caller's, exactly as in the reference (FrameManager.cpp:448-462,501-510):
 * prevViewProj = last frame's proj*view, rtaoParams1.w = frames since (re)create. */
VKR_API int vkr_rtao_execute(vkr_ctx* ctx, const vkr_rtao_frame_io* io, void* stream);

/* Row-band variant for multi-GPU (no reference counterpart: the reference is single-GPU).  Produces rows
 * [y0, y1) of rtao_full and of history[curr]; rows outside are untouched.  Internally traces and filters the
 * A-Trous halo (2*(2^N - 1) rows each side) so the band is bit-identical to the same rows of a full-frame
 * vkr_rtao_execute.  The caller all-gathers the bands (history via vkr_rtao_image(ctx, 0)) before the next frame. */
VKR_API int vkr_rtao_execute_band(vkr_ctx* ctx, const vkr_rtao_frame_io* io, uint32_t y0, uint32_t y1, void* stream);

/* Two-phase sharding for multi-GPU (no reference counterpart).  Phase 1: dispatchTrace on the 8-row stripes
 * s = rank, rank + nranks, rank + 2*nranks, ... (stripe s = rows [8s, 8s+8)): interleaving evens out the per-row cost
 * of the traversal.  The caller then all-gathers history[curr] (vkr_rtao_image after the frame, or which = 0 of the
 * ping-pong BEFORE end-of-frame: see vkr_rtao_history_curr) so that every rank holds all rows.  Phase 2:
 * vkr_rtao_denoise_band runs A-Trous x N + upsample for rows [y0, y1) -- its halo rows are read from the gathered
 * history instead of being re-traced -- and ends the frame like vkr_rtao_execute.  y0 == y1 is allowed (a rank
 * without rows still flips its frame index).  Bit-identical to vkr_rtao_execute on the same inputs. */
VKR_API int vkr_rtao_trace_stripes(vkr_ctx* ctx, const vkr_rtao_frame_io* io, uint32_t rank, uint32_t nranks, void* stream);
VKR_API int vkr_rtao_denoise_band(vkr_ctx* ctx, const vkr_rtao_frame_io* io, uint32_t y0, uint32_t y1, void* stream);

/* The collective of that schedule, behind the boundary (no reference counterpart; SURVEY 8b proposed ABI, 8e "one collective
 * per exchange").  A C++ host attaches its communicator once per context:
 *   vkr_comm_attach         an RCCL communicator (ncclComm_t passed as void*; borrowed, must outlive the context or be
 *                           detached with nccl_comm = NULL).  librccl.so is resolved with dlopen at this call, the library
 *                           does not link it.  rank / n_ranks are checked against ncclCommUserRank / ncclCommCount.
 *   vkr_comm_attach_custom  the host's own transport (MPI, gloo, a test harness): fn(user, send, recv, bytes, stream) must
 *                           leave rank r's `bytes` bytes at recv + r * bytes on every rank, ordered on `stream`
 *                           (device pointers; return 0 for success).  fn = NULL detaches.
 * and then calls, per frame, on the stream of the frame:
 *   vkr_rtao_trace_stripes(rank, n) -> vkr_rtao_gather_history -> vkr_rtao_denoise_band(rows of rank) -> vkr_gather_bands
 * vkr_rtao_gather_history: all ranks end up with every row of history[curr] of the frame in flight (one fused pack kernel
 *   of this rank's stripes, ONE all-gather of raw bytes, one fused unpack kernel).
 * vkr_gather_bands: planes[k] are full-frame R16F (or any 16-bit) images on the device of which this rank has produced the
 *   rows [rank*Hb, min(H,(rank+1)*Hb)), Hb = ceil(H / n_ranks) -- rtao_full, and e.g. a 16-bit sun-visibility plane; after
 *   the call every rank holds all rows of all planes (pack, ONE all-gather, unpack).  1 <= n_planes <= VKR_GATHER_MAX_PLANES.
 * Errors: VKR_ERR_STATE without an attached communicator, VKR_ERR_RCCL when the transport fails. */
#define VKR_GATHER_MAX_PLANES 4
typedef int (*vkr_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank, void* stream);
VKR_API int vkr_comm_attach(vkr_ctx* ctx, void* nccl_comm, uint32_t rank, uint32_t n_ranks);
VKR_API int vkr_comm_attach_custom(vkr_ctx* ctx, vkr_allgather_fn fn, void* user, uint32_t rank, uint32_t n_ranks);
VKR_API int vkr_rtao_gather_history(vkr_ctx* ctx, void* stream);
VKR_API int vkr_gather_bands(vkr_ctx* ctx, uint16_t* const* planes_r16f, uint32_t n_planes, void* stream);
/* history[curr] of the frame in flight (the image dispatchTrace writes; device pointer, R16F W*H). */
VKR_API int vkr_rtao_history_curr(vkr_ctx* ctx, uint16_t** out);

/* The three dispatches, individually callable (RtaoComputePass.cpp:503-517, 519-554, 556-571).
 * Push constants {width,height,step,iteration} (RtaoComputePass.h:21-26) become arguments. */
VKR_API int vkr_rtao_trace(vkr_ctx* ctx, const vkr_rtao_frame_io* io, void* stream);
VKR_API int vkr_rtao_atrous(vkr_ctx* ctx, const vkr_rtao_frame_io* io, uint32_t step, uint32_t iteration, void* stream);
VKR_API int vkr_rtao_upsample(vkr_ctx* ctx, const vkr_rtao_frame_io* io, uint32_t final_index, void* stream);
/* Advance the frame-in-flight index (history curr/prev swap), FrameManager.cpp:1155-1165. vkr_rtao_execute
 * calls this itself; callers of the three separate dispatches call it once per frame. */
VKR_API int vkr_rtao_end_frame(vkr_ctx* ctx);

/* Replaces: swapchain-recreate reset (FrameManager.cpp:28,57-58,1057): frame-in-flight index 0, history and
 * ping-pong cleared to 1.0. */
VKR_API int vkr_rtao_reset(vkr_ctx* ctx, void* stream);

/* Context-owned images (device pointers, R16F W*H): which = 0 history[curr of the LAST executed frame],
 * 1 history[prev], 2 ping0, 3 ping1.  Reference: FrameManager getters used at RtaoComputePass.cpp:289-368. */
VKR_API int vkr_rtao_image(vkr_ctx* ctx, uint32_t which, uint16_t** out);

/* ---------------------------------------------------------------- shadow rays ----------- */

/* pbr.frag lifted from per-fragment to per-pixel.  world_pos / geom_normal: float3 per pixel (device),
 * valid: optional uint8 per pixel (0 => pixel skipped, output 1.0), out_vis: float per pixel (device). */

/* Replaces: ComputeShadowVisibilityDirectional (pbr.frag:217-260) with L = -normalize(directionalLightDir.xyz)
 * (pbr.frag:379-380) taken from `light_dir_to_light` (already negated+normalised by the caller),
 * sun_radius / samples = directionalLightParams.xy.
 * Concurrency: this call shares no scratch with the vkr_rtao_* calls, so it may be issued on a second stream while an
 * RTAO frame of the same context runs (its persistent waves then fill the drain of the trace kernel).  No other two
 * calls on one context may overlap. */
VKR_API int vkr_shadow_sun(vkr_ctx* ctx, const float* world_pos, const float* geom_normal, const uint8_t* valid,
                           uint32_t pixel_count, const float light_dir_to_light[3], float sun_radius,
                           float sample_count, float* out_vis, void* stream);
/* No counterpart in the reference: vkr_shadow_sun for callers that cannot allocate -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same inputs, same outputs, same bits as vkr_shadow_sun (its packet
 * kernels), but the scratch those kernels need (the chunk counter of the persistent waves and the frame's sample-direction
 * table, which vkr_shadow_sun keeps in buffers it allocates on first use) lives in a fixed device array of
 * VKR_NOALLOC_SCRATCH_BYTES inside the library (each GPU the library runs on holds it until the process ends); the call
 * performs no hipMalloc / hipFree and calls no allocating routine.
 * Limits (VKR_ERR_LIMIT, nothing is launched): pixel_count <= VKR_NOALLOC_MAX_PIXELS; the BVH must fit the packet
 * kernel's queue items (<= 2^22 wide nodes; larger trees: vkr_shadow_sun).
 * Every call on a GPU -- from any context of that device -- shares that GPU's array: calls that overlap on one GPU race;
 * issue them on one stream or order them with events. */
#define VKR_NOALLOC_SCRATCH_BYTES 1024
#define VKR_NOALLOC_MAX_PIXELS (1u << 28)
#define VKR_NOALLOC_MAX_DEVICES 64
VKR_API int vkr_shadow_sun_noalloc(vkr_ctx* ctx, const float* world_pos, const float* geom_normal, const uint8_t* valid,
                                   uint32_t pixel_count, const float light_dir_to_light[3], float sun_radius,
                                   float sample_count, float* out_vis, void* stream);
/* Replaces: ComputeShadowVisibility (pbr.frag:169-196), one light. */
VKR_API int vkr_shadow_point(vkr_ctx* ctx, const float* world_pos, const float* geom_normal, const uint8_t* valid,
                             uint32_t pixel_count, const float light_pos[3], float* out_vis, void* stream);

/* ---------------------------------------------------------------- G-buffer producer ----- */

/* SURVEY 8(f) row 2 (input side of the path): stands in for DepthPrepass (depth_prepass.vert:42-52,
 * depth_only.frag:20-35) by casting one primary ray per pixel centre through the same BVH.
 * Outputs (device, any may be NULL): linear view-Z R16F (0 = sky); normal*0.5+0.5 RGBA16F with
 * normal = normalize(interpolated normalize(transpose(inverse(mat3(model))) * vertex normal)) as depth_prepass.vert:47-48
 * and depth_only.frag:31-33 produce it (clear value (0.5,0.5,1,1), DepthPrepass.cpp:36); world position float3;
 * Ng float3 = the same interpolated normal, negated on back faces (pbr.frag:339-346); valid uint8.
 * MASK texels below the cutoff are holes (depth_only.frag:22-29), BLEND geometry is not in the prepass. */
VKR_API int vkr_gbuffer_primary(vkr_ctx* ctx, const float ubo[VKR_UBO_FLOATS], uint16_t* linear_depth_r16f,
                                uint16_t* normal_rgba16f, float* world_pos, float* geom_normal, uint8_t* valid,
                                void* stream);

/* The same with the prepass's MSAA (DepthPrepass.cpp:31-65: both attachments are multisampled and resolve with
 * VK_RESOLVE_MODE_AVERAGE; no sample shading, DepthPrepassPipeline.cpp:96).  The reference takes the device's maximum
 * sample count, the driver's sample positions and its resolve arithmetic; this entry point fixes one definition:
 * `samples` in {1, 2, 4, 8} at the Vulkan standard sample locations; coverage and depth test per sample; the fragment
 * outputs of the winning primitive evaluated once at the pixel centre (extrapolated on its plane when the centre is not
 * covered) -- the MASK discard too; fp16 per sample; resolve = float32 mean in sample order, rounded to fp16; uncovered
 * samples hold the clear values (0 and (0.5,0.5,1,1)).  valid = resolved depth > 1e-6.  world_pos / geom_normal keep the
 * pixel-centre semantics of vkr_gbuffer_primary.  samples = 1 is vkr_gbuffer_primary. */
VKR_API int vkr_gbuffer_primary_msaa(vkr_ctx* ctx, const float ubo[VKR_UBO_FLOATS], uint32_t samples,
                                     uint16_t* linear_depth_r16f, uint16_t* normal_rgba16f, float* world_pos,
                                     float* geom_normal, uint8_t* valid, void* stream);

/* ---------------------------------------------------------------- GPU-driven indirect build */

/* SURVEY 8(f) row 4 (BASELINE configs[2]).  Replaces the per-frame CPU loop FrameManager::prepareSharedOpaqueIndirect
 * (FrameManager.cpp:118-175) that fills the mapped drawData / indirect-command buffers ScenePass draws from
 * (vkCmdDrawIndexedIndirect per bucket span).  Independent of the ray-tracing scene: needs no vkr_scene_upload. */
typedef struct vkr_draw_slot {            /* SharedOpaqueDrawSlot (FrameManager.cpp:99-104)                        */
    uint32_t node_linear_index, mesh_index, mat_index, double_sided;
} vkr_draw_slot;
typedef struct vkr_mesh_draw_info {       /* MeshDrawInfo (GlobalMeshBuffer.h:12-16)                               */
    uint32_t vertex_offset, first_index, index_count;
} vkr_mesh_draw_info;
typedef struct vkr_draw_indexed_indirect_command {   /* VkDrawIndexedIndirectCommand, 20 bytes                    */
    uint32_t index_count, instance_count, first_index;
    int32_t vertex_offset;
    uint32_t first_instance;
} vkr_draw_indexed_indirect_command;
typedef struct vkr_bucket_span {          /* SharedOpaqueBucketSpan (FrameManager.cpp:163-171)                     */
    uint32_t double_sided, mat_index, first_command, draw_count;
} vkr_bucket_span;
typedef struct vkr_indirect_desc {        /* all HOST arrays, copied                                               */
    const int32_t* node_parent;           /* Model::getLinearNodes(): parent's linearIndex or -1; parents first     */
    const float* node_local;              /* Node::getLocalMatrix() per linear node, 16 floats, GLM column-major    */
    uint32_t node_count;
    const vkr_draw_slot* slots;           /* in the collection order of buildSharedDrawSlots (:89-109), BLEND        */
    uint32_t slot_count;                  /*   meshes already left out (:98); at most 2^24                           */
    const vkr_mesh_draw_info* mesh_infos; /* GlobalMeshBuffer::getMeshInfos()                                       */
    uint32_t mesh_count;
    const float* mesh_bounds;             /* optional: object-space AABB per mesh {min xyz, max xyz}; enables the    */
                                          /*   frustum test of docs/GPU_DRIVEN_RENDERING_DESIGN.md                   */
    uint32_t max_draws;                   /* AppConfig::MAX_DRAWS; clamped to >= 1 (FrameManager.cpp:24)            */
} vkr_indirect_desc;
/* Replaces FrameManager::buildSharedDrawSlots (FrameManager.cpp:81-116): keeps the slots and applies its
 * std::stable_sort by (doubleSided, matIndex, meshIndex) (:110-115), once per model. */
VKR_API int vkr_indirect_upload(vkr_ctx* ctx, const vkr_indirect_desc* desc);
/* The sorted slots (slot_count entries, HOST): inspection/tests. */
VKR_API int vkr_indirect_slots_download(vkr_ctx* ctx, vkr_draw_slot* out_slots);
/* Replaces FrameManager::prepareSharedOpaqueIndirect (:118-175).  model_matrix: GLM column-major.  view_proj: NULL =
 * the reference as it is (every valid slot draws); non-NULL = additionally drop slots whose mesh AABB, transformed by
 * the node's world matrix (BoundingBox::Transform, BoundingBox.h:14-39), lies outside the frustum of view_proj
 * (Frustum::extractPlanes / Intersects, Frustum.h:18-60).  draw_data (DEVICE, capacity x 16 floats) = drawDataMapped,
 * commands (DEVICE, capacity entries) = indirectMapped; entries past the draw count are left untouched.  capacity is
 * what the two arrays hold and must be >= min(max_draws, slot_count) (the reference sizes both maxDraws,
 * FrameManager.cpp:24), else VKR_ERR_INVALID_ARG and nothing is launched. */
VKR_API int vkr_indirect_prepare(vkr_ctx* ctx, const float model_matrix[16], const float* view_proj, float* draw_data,
                                 vkr_draw_indexed_indirect_command* commands, uint32_t capacity, void* stream);
/* sharedOpaqueDrawCount and sharedOpaqueBucketSpans of the last prepare (HOST outputs; synchronises `stream`).
 * *span_count is the number of spans produced; at most span_capacity are written. */
VKR_API int vkr_indirect_result(vkr_ctx* ctx, uint32_t* draw_count, vkr_bucket_span* spans, uint32_t span_capacity,
                                uint32_t* span_count, void* stream);

/* ---------------------------------------------------------------- IBL precompute --------- */

/* SURVEY 8(f) row 3 (BASELINE configs[1]).  All images are tightly packed DEVICE arrays:
 *   equirect   RGBA32F [height][width][4]           (HdrTextureLoader: stbi_loadf RGBA, HdrTextureLoader.cpp:20-28)
 *   env cube   RGBA32F [6][S][S][4]                 (EquirectToCubemap.cpp:75), faces +X -X +Y -Y +Z -Z, Vulkan layout
 *   irradiance RGBA16F [6][S][S][4]                 (IblPrecompute.cpp:115-120)
 *   prefilter  RGBA16F, mip-major: mip m = [6][S>>m][S>>m][4], m < floor(log2 S) + 1   (IblPrecompute.cpp:285,376-387)
 *   BRDF LUT   RG16F   [S][S][2], column = NdotV, row = roughness                        (IblPrecompute.cpp:472-546)
 * Cube sizes must be powers of two. */

/* Replaces: EquirectToCubemap::convert (EquirectToCubemap.cpp:66-398, equirect_to_cubemap.frag); the equirect
 * sampler is LINEAR, U repeat, V clamp-to-edge (Renderer.cpp:113-115).  Reference cube_size: 512 (Renderer.cpp:117). */
VKR_API int vkr_ibl_equirect_to_cube(vkr_ctx* ctx, const float* equirect, uint32_t width, uint32_t height,
                                     float* env_cube, uint32_t cube_size, void* stream);
/* Replaces: IblPrecompute::compute step 1 (IblPrecompute.cpp:107-283, irradiance_convolution.frag). Reference size: 32. */
VKR_API int vkr_ibl_irradiance(vkr_ctx* ctx, const float* env_cube, uint32_t env_size, uint16_t* irradiance,
                               uint32_t size, void* stream);
/* Replaces: step 2 (IblPrecompute.cpp:285-470, prefilter.frag), every mip, roughness = mip / (mips - 1) (:383).
 * Reference size: 128 (8 mips); `prefilter` holds vkr_ibl_prefilter_texels(size) * 4 halves. */
VKR_API int vkr_ibl_prefilter(vkr_ctx* ctx, const float* env_cube, uint32_t env_size, uint16_t* prefilter,
                              uint32_t size, void* stream);
/* Replaces: step 3 (IblPrecompute.cpp:472-546, brdf_integrate.frag).  Reference size: 512. */
VKR_API int vkr_ibl_brdf_lut(vkr_ctx* ctx, uint16_t* brdf_lut, uint32_t size, void* stream);
/* Replaces: IblPrecompute::compute (IblPrecompute.cpp:75-561): the three steps above on one stream. */
VKR_API int vkr_ibl_compute(vkr_ctx* ctx, const float* env_cube, uint32_t env_size, uint16_t* irradiance,
                            uint32_t irradiance_size, uint16_t* prefilter, uint32_t prefilter_size,
                            uint16_t* brdf_lut, uint32_t brdf_lut_size, void* stream);
/* mipCount() of IblPrecompute.cpp:68-71 and the number of texels (all faces, all mips) of a prefilter map. */
VKR_API uint32_t vkr_ibl_prefilter_mip_count(uint32_t size);
VKR_API size_t vkr_ibl_prefilter_texels(uint32_t size);

/* ---------------------------------------------------------------- counters -------------- */

typedef struct vkr_trace_stats {
    uint64_t rays;          /* ray queries issued (AO rays + retries)      */
    uint64_t nodes_visited; /* internal nodes popped                        */
    uint64_t tris_tested;   /* ray/triangle tests                           */
    uint64_t wave_node_iters; /* node-loop iterations counted once per wave (lane utilisation = nodes_visited / (64*this)) */
    uint64_t wave_tri_iters;  /* triangle-loop iterations counted once per wave */
    uint64_t overflow_tiles;  /* K1 queue items that took the wave's global overflow stack instead of LDS (a performance counter, not an error) */
} vkr_trace_stats;
/* Enable (1) / disable (0) traversal counters for subsequent AO traces (slower kernel variant), and read
 * them back (synchronises).  Feeds bench.py's algorithmic-bytes figure. */
VKR_API int vkr_trace_stats_enable(vkr_ctx* ctx, int enable);
VKR_API int vkr_trace_stats_get(vkr_ctx* ctx, vkr_trace_stats* out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* VKR_RT_H */
