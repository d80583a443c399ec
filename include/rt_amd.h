/*
 * rt_amd.h -- C ABI of librt_amd.so: an MI355X (gfx950) replacement for the reference's
 * path-tracing shader pass.
 *
 * The reference (grayleafy/RayTracing) has no plugin/FFI interface; the seam this library
 * fills is the set of GL objects and uniforms that main.cpp:137-172 binds before
 * RT_Screen::DrawScreen, plus the CPU BVH builder that feeds them.  Each entry point below
 * names the reference code it stands in for.  File names are relative to the reference's
 * source directory; "frag" is RayTracerFragmentShader.glsl.
 *
 * Conventions: every function returns 0 on success and a negative RT_E_* code on failure
 * (the reference only prints errors: Shader.h:45-48, Model.h:60-64); rt_last_error() gives
 * the message.  No exceptions cross this boundary.  Host arrays passed in are copied before
 * the call returns and stay owned by the caller (the reference uploads and delete[]s its
 * staging arrays, ObjectTexture.h:82,133).  A context belongs to one GPU; calls on one
 * context must be serialised by the caller, distinct contexts may be used from distinct
 * threads.  Images are W*H*3 floats, row 0 is the BOTTOM row (TexCoords.y ~ 0,
 * RT_Screen.h:8-20): pixel (px,py) has TexCoords ((px+0.5)/W, (py+0.5)/H).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID (-1)   /* bad argument / size beyond a stated limit */
#define RT_E_STATE (-2)     /* call order: no scene, no camera ... */
#define RT_E_HIP (-3)       /* a HIP runtime call failed */
#define RT_E_IO (-4)        /* file not found / parse error */
#define RT_E_NOMEM (-5)

#define RT_MAX_SPHERES 10   /* uniform Sphere sphere[10], frag:143 */
#define RT_SOBOL_WORDS 512  /* uniform uint V[512], frag:81 */
#define RT_MAX_BOUNCES 8    /* 16 Sobol dimensions / 2 per bounce, frag:100-104 */
#define RT_MAX_TRIANGLES 50000000   /* per scene: every per-triangle record array on the GPU stays below 4 GiB, so the
                                       kernels address records with 32-bit offsets (the reference's own BVH texture
                                       cannot express more than 4 194 304 triangles, ObjectTexture.h:152-154) */

/* class Triangle, BVH.h:26-33 (144 bytes, same field order). */
typedef struct RtTriangle {
    float v[3][3];
    float n[3][3];
    float tex[3][2];
    float ka[3], kd[3], ks[3];
    float center[3];
} RtTriangle;

/* class LinearBVHNode, BVH.h:99-105 (36 bytes, same field order). */
typedef struct RtLinearBVHNode {
    float minBound[3], maxBound[3];
    int32_t triangleNum;
    int32_t offsetOrRc;
    int32_t axis;
} RtLinearBVHNode;

/* class Sphere, Scene.h:27-44; uniform struct Sphere, frag:135-142. */
typedef struct RtSphere {
    float center[3];
    float radius;
    float albedo[3];
    float specularRate;     /* diffuse threshold, frag:586 */
    float refractionRate;   /* mirror threshold, frag:594 */
    float roughness;        /* carried, unused (frag:543,554) */
} RtSphere;

/* uniform struct Camera, frag:115-124, filled at main.cpp:156-163. */
typedef struct RtCamera {
    float camPos[3];
    float front[3];
    float right[3];
    float up[3];
    float halfH;
    float halfW;
    float leftbottom[3];
    int32_t LoopNum;        /* ignored by rt_set_camera; frames are numbered by rt_render_frames */
} RtCamera;

/* Work counters of the render calls since the last rt_reset (all frames, all pixels). */
typedef struct RtStats {
    uint64_t rays;          /* closest-hit queries traced: camera ray + one per scatter; a ray
                               that already missed is not traced again (the reference re-traces
                               it, frag:655-670, with no effect on the image) */
    uint64_t node_tests;    /* bounding boxes tested (IntersectBound, frag:309-331) */
    uint64_t tri_tests;     /* triangles tested (hitTriangle, frag:255-278) */
    uint64_t tri_hits;      /* rays whose closest BVH hit was a triangle (frag:457) */
    uint64_t frames;        /* frames rendered */
    uint64_t retraced;      /* rays whose search-index result could not be proven equal to the
                               reference's and that were traced again in the reference's order */
} RtStats;

/* How the library runs the uploaded scene (diagnostics; no counterpart in the reference). */
typedef struct RtSceneInfo {
    int64_t triangles;
    int64_t index_nodes;            /* 4-wide nodes of the search index; 0 when the scene has none */
    int32_t tree_height;            /* levels below the root of the reference tree */
    int32_t index_stack_need;       /* bound on the deferred nodes of one search through the index */
    int32_t index_stack_lds;        /* how many of them are kept in LDS; deeper ones use global memory */
    int32_t index_enabled;          /* 0: the tree's boxes are not the exact nested min/max boxes the
                                       proof step relies on (or the index was switched off), every
                                       ray is traced in the reference's order */
    int32_t paths_waves_per_simd;   /* variant of the persistent path kernel used for this scene: 5..8 */
    int32_t frames_per_launch;      /* most frames one launch of the path kernel renders at this image size */
    int32_t index_built_on_device;  /* 1: the search index of this scene was built on the GPU (RtOptions.index_build) */
    int32_t index_build_fell_back;  /* 1: the device build of the index failed (no working memory, or a tree it
                                       gives up on) and the host built the index instead */
} RtSceneInfo;

/* Render options; rt_default_options() gives the reference's behaviour. */
typedef struct RtOptions {
    int32_t max_bounces;    /* loop bound of shading(), 4 in the reference (frag:624); <= 8 */
    int32_t pipeline;       /* RT_PIPELINE_* */
    int32_t cull;           /* 1: skip boxes that provably cannot hold a closer hit (same image);
                               0: visit exactly the boxes the reference visits */
    int32_t count_work;     /* 1: maintain node_tests/tri_tests/tri_hits (slower kernels) */
    int32_t accel;          /* 1 (persistent and wavefront pipelines): find the closest accepted triangle through
                               a SAH search index built at upload, prove with one exact box test that
                               the reference's traversal reaches it, and re-trace in the reference's
                               order the rays for which the proof fails (same image);
                               0: always traverse the reference's tree in the reference's order */
    int32_t history8;       /* 0 (default): the running mean is kept in fp32;
                               1: emulate the reference's history texture, an 8-bit GL_RGB attachment
                               (RT_Screen.h:61): after every frame the mean is clamped to [0,1] and
                               rounded to a multiple of 1/255 (NaN becomes 0).  The plain sum is
                               not affected. */
    int32_t overlap;        /* persistent pipeline only.  1 (default): consecutive batches of frames (calls
                               of rt_render_frames, or the 64-frame batches of one call) alternate between
                               two internal streams (three for batches of up to ~5M pixel samples, e.g. one
                               or two 1080p frames per call), so that a batch starts while the last paths
                               of the previous ones are still running; accumulation stays in frame order, and the
                               context's stream waits for each batch, so reads and the caller's later work
                               on that stream see the result.  Work the CALLER enqueues on the context's
                               stream before a render call (a collective on the sum buffer, a memset of a
                               bound buffer) is waited for by the kernels that add the new frames to the
                               accumulators, not by the path tracing itself, which reads nothing the caller
                               can write.  0: everything runs on the context's stream. */
    int32_t index_build;    /* where rt_scene_upload builds the search index of `accel` (read at upload):
                               RT_INDEX_BUILD_HOST: binned SAH on the host (upload of 1.3M triangles in
                               0.3 s, of 10.5M in 2.2 s), the index that traces fastest (by 5 % on those
                               two scenes, by 15 % on a 262k-triangle interior); RT_INDEX_BUILD_GPU:
                               tree clustered in Morton order with size classes on the device (0.1 s / 0.6 s);
                               RT_INDEX_BUILD_AUTO (default): on the host below
                               RT_INDEX_AUTO_GPU_TRIANGLES triangles, on the device from there on, where
                               the host build takes several seconds.  Same images whichever builds it. */
    int32_t shading_normals; /* 0 (default): the reference's behaviour, the face normal of getTriangleNormal
                               (frag:296-300).  1: an extension the reference only prepares - its Triangle
                               carries n0 n1 n2 (frag:131, Model.h:101-112) and the comment on
                               getTriangleNormal marks it "to be changed to interpolation": shade with
                               the vertex normals RtTriangle.n interpolated at the hit point (weights =
                               the three edge functions of hitTriangle, frag:265-275), normalised and
                               turned to the side of the face normal; a triangle whose normals are zero
                               keeps its face normal.  The scene must have been uploaded with this
                               option already set (the normals are only copied to the GPU then). */
    int32_t albedo_texture; /* 0 (default): the reference's behaviour.  1: a second extension the reference
                               only prepares - its Triangle carries uv0 uv1 uv2 (frag:132, Model.h:117-124)
                               and its loader reads a material's diffuse maps (Model.h:167-169), but the
                               shader never samples them: the albedo of a diffuse draw (frag:466-468)
                               becomes kd * texel of the image given to rt_set_albedo_texture, at the
                               texture coordinates RtTriangle.tex interpolated at the hit point (weights
                               as for shading_normals, divided by their sum), wrapped as GL_REPEAT
                               (Model.h:244-245), nearest texel.  Lights, mirrors, glass and the spheres
                               are not affected.  The scene must have been uploaded with this option
                               already set (the coordinates are only copied to the GPU then). */
    int32_t coalesce;       /* persistent pipeline with overlap only.  1 (default): a call of rt_render_frames that
                               asks for fewer frames than a launch holds is not always launched at once.  While
                               two earlier batches are still running on the GPU, the frames of such calls
                               (their randOrigin values are copied) are collected and launched together -
                               when a launch is full, when fewer than two batches are in flight at a later
                               call, or at the next call of any other function of this context (reads,
                               rt_synchronize, setters, ...), whichever comes first.  Same images, same
                               counters; only the moment the work starts changes, and the GPU is busy in the
                               meantime.  Why: the waves of a one-frame launch spend a fifth of their paths
                               draining with nothing to refill their lanes, those of an eight-frame launch a
                               fortieth (one 1080p frame per call: 9.5 -> ~12.5 Grays/s on a 1.3M-triangle scene).
                               Never applied where the caller could tell: with a stream of its own
                               (rt_set_stream), a bound sum buffer (rt_bind_sum_buffer) or after
                               rt_sum_device_ptr, a call's frames are enqueued before the call returns, as
                               stream ordering demands.  0: every call launches its own frames. */
} RtOptions;
#define RT_INDEX_BUILD_HOST 0
#define RT_INDEX_BUILD_GPU 1
#define RT_INDEX_BUILD_AUTO 2
#define RT_INDEX_AUTO_GPU_TRIANGLES 16000000
#define RT_PIPELINE_WAVEFRONT 0   /* generate / extend / shade / accumulate kernels over ray queues */
#define RT_PIPELINE_MEGAKERNEL 1  /* one thread per pixel runs the whole path */
#define RT_PIPELINE_PERSISTENT 2  /* persistent waves; each lane runs whole paths and takes the next (frame, pixel)
                                     when one ends; up to 64 frames of an rt_render_frames call share a launch */

typedef struct RtContext RtContext;
typedef struct RtScene RtScene;     /* host-side scene assembly (Scene.h) */

/* ---------------------------------------------------------------- host side, no GPU */

/* BVHTree::build + split + buildLinerTree + addLinearNode, BVH.h:117-287.
 * tris[0..n) hold v/n/tex/ka/kd/ks in scene order (mesh order, then index order,
 * BVH.h:120-135); `center` is computed here (BVH.h:36-39).  On return tris is permuted in
 * place exactly as the reference's nth_element recursion permutes it and nodes[0..2n-1)
 * holds the pre-order tree: leaves have triangleNum 1, offsetOrRc = triangle index, axis -1;
 * interior nodes have offsetOrRc = right child, left child = own index + 1.  The selection
 * step reproduces libstdc++'s std::nth_element permutation without calling it.
 * node_capacity must be >= 2n-1.  *node_count receives 2n-1.  n == 0 is RT_E_INVALID (the
 * reference dereferences triangles[0], BVH.h:141). */
int rt_bvh_build(RtTriangle *tris, int64_t n, RtLinearBVHNode *nodes, int64_t node_capacity,
                 int64_t *node_count);

/* The same builder with the split rule the reference's author left commented out at
 * BVH.h:225-256 in place of the median (BVH.h:219-222): the range is sorted along each axis
 * (std::sort; the permutation of libstdc++'s introsort is reproduced), prefix and suffix boxes
 * are swept, and the split with the smallest Bound::getArea(left) + Bound::getArea(right) wins
 * -- getArea as written at BVH.h:181-184, `dif.y + dif.z` included, and no triangle counts in
 * the cost -- then the range is sorted along the winning axis, which becomes the node's `axis`.
 * The node ORDER differs from rt_bvh_build's, so this is a separate mode, never a default.  The
 * cost tends to split off one triangle at a time: trees get as deep as they are large, the build
 * is quadratic, and rt_scene_upload refuses trees deeper than 60 levels (the shader's own stack
 * holds 256 entries, frag:381).  n <= RT_SWEEP_MAX_TRIANGLES. */
#define RT_SWEEP_MAX_TRIANGLES 16384
int rt_bvh_build_sweep(RtTriangle *tris, int64_t n, RtLinearBVHNode *nodes, int64_t node_capacity,
                       int64_t *node_count);

/* Texture geometry of getTexture(), ObjectTexture.h:79-80 and 138-139: a count*per float
 * array stored in a (x*per) x y R32F texture. */
int rt_texture_dims(int64_t count, int per, int *x, int *y);
/* ObjectTexture.h:82-115: out has x*y*33 floats, 33 per triangle (v,n,uv per vertex, ka, kd,
 * ks); texels after the last triangle are zero (uninitialised in the reference). */
int rt_pack_triangle_texture(const RtTriangle *tris, int64_t n, float *out, int x, int y);
/* ObjectTexture.h:140-161: out has x*y*9 floats per node: max xyz, min xyz, triangleNum,
 * offsetOrRc, axis (integers stored as floats). */
int rt_pack_bvh_texture(const RtLinearBVHNode *nodes, int64_t m, float *out, int x, int y);

/* uint V[16*32], sobol.h:8-25 (= v_matrix.txt): regenerated from the Joe-Kuo primitive
 * polynomials and initial direction numbers of dimensions 1-16. */
int rt_sobol_default_table(uint32_t v[RT_SOBOL_WORDS]);
/* grayCode / sobol, sobol.h:37-50 (CPU twins of frag:84-97). */
uint32_t rt_gray_code(uint32_t i);
float rt_sobol(const uint32_t v[RT_SOBOL_WORDS], uint32_t d, uint32_t i);

/* Camera::Camera + updateCameraVectors, Camera.h:18-46: yaw/pitch/fov in degrees,
 * halfH = tan(radians(fov)), halfW = halfH*W/H, leftbottom = front - halfW*right - halfH*up.
 * rt_camera_default = position (0,2,5), yaw -90, pitch 0, fov 45. */
int rt_camera_from_angles(RtCamera *out, const float pos[3], float yaw_deg, float pitch_deg,
                          float fov_deg, int width, int height);
int rt_camera_default(RtCamera *out, int width, int height);

/* class Camera, Camera.h:16-139: the interactive camera with its state.  Every change resets
 * LoopNum to 0 (Camera.h:45,59,94,109): the caller then calls rt_reset() and numbers frames
 * from 1 again, as main.cpp does through cam.LoopIncrease(). */
typedef struct RtCameraRig {
    float cameraPos[3], cameraFront[3], cameraUp[3], cameraRight[3], worldUp[3];
    float fov, Pitch, Yaw;          /* degrees */
    float cameraSpeed;
    int32_t firstMouse;
    float lastX, lastY;
    float ScreenRatio, halfH, halfW;
    float LeftBottomCorner[3];
    int32_t LoopNum;
} RtCameraRig;
#define RT_CAMERA_FORWARD 0     /* enum Camera_Movement, Camera.h:9-14 */
#define RT_CAMERA_BACKWARD 1
#define RT_CAMERA_LEFT 2
#define RT_CAMERA_RIGHT 3
int rt_camera_rig_init(RtCameraRig *rig, int width, int height);                  /* Camera.h:18-31 */
int rt_camera_rig_update_vectors(RtCameraRig *rig);                               /* Camera.h:33-46 */
int rt_camera_rig_update_fov(RtCameraRig *rig, float offset);                     /* Camera.h:47-60: fov clamped to 1..45 */
int rt_camera_rig_update_front(RtCameraRig *rig, float xpos, float ypos);         /* Camera.h:61-87: 0.03 deg per pixel, pitch clamped to +-89 */
int rt_camera_rig_update_screen_ratio(RtCameraRig *rig, int width, int height);   /* Camera.h:88-95 */
int rt_camera_rig_process_keyboard(RtCameraRig *rig, int direction, float delta_time);   /* Camera.h:96-110 */
int rt_camera_rig_loop_increase(RtCameraRig *rig);                                /* Camera.h:111-113 */
int rt_camera_rig_uniforms(const RtCameraRig *rig, RtCamera *out);                /* main.cpp:156-163 */

/* Replaces CPURandomInit/GetCPURandom, Tool.h:8-14, which seed from time(): an explicit
 * 32-bit generator.  *state is advanced; returns 874264.0f * (u + 1.0f), u in [0,1), the
 * value main.cpp:149 feeds to the `randOrigin` uniform. */
float rt_rand_origin(uint32_t *state);

/* Scene assembly, Scene.h:46-90 + Model.h:54-182 + BVH.h:120-135. */
int rt_scene_create(RtScene **out);
void rt_scene_destroy(RtScene *scene);
/* Model(path) then Scene::pushModel(model, scale, translation), main.cpp:103-107: Wavefront
 * OBJ + MTL; one mesh per object/material group in file order, polygons fan-triangulated,
 * per-mesh Ka/Kd/Ks; Position = Position*scale + translation (Scene.h:57-60). */
int rt_scene_push_obj(RtScene *scene, const char *obj_path, const float scale[3],
                      const float translation[3]);
/* Same transform applied to caller-provided triangles (v/n/tex/ka/kd/ks used). */
int rt_scene_push_triangles(RtScene *scene, const RtTriangle *tris, int64_t n,
                            const float scale[3], const float translation[3]);
/* Scene::pushSphere, Scene.h:67-69. */
int rt_scene_push_sphere(RtScene *scene, const RtSphere *sphere);
/* The scene of main.cpp:103-110 built procedurally (room + cube + two spheres): the same 32
 * triangles, in the same order, that loading objects/room.obj and objects/cube.obj gives. */
int rt_scene_push_demo(RtScene *scene);
/* Synthetic benchmark geometry.  kind 0: the demo room (20 triangles, light included) around
 * `copies` displaced icospheres of 20*4^level triangles (seeded radial sinusoid);
 * kind 1: room scaled x3 with a grid of cols x rows tessellated columns of 2*level*level
 * triangles cycling diffuse / mirror / glass, plus the two demo spheres. */
int rt_scene_push_synthetic(RtScene *scene, int kind, int level, int copies, uint32_t seed);
int64_t rt_scene_triangle_count(const RtScene *scene);
int rt_scene_sphere_count(const RtScene *scene);
/* The vector<Triangle> as it stands before BVHTree::split (BVH.h:120-135). */
int rt_scene_get_triangles(const RtScene *scene, RtTriangle *out, int64_t capacity);
int rt_scene_get_spheres(const RtScene *scene, RtSphere *out, int capacity);

/* ---------------------------------------------------------------- device side */

/* Creates a context on HIP device `device` for a W x H image (the FBO pair of
 * RT_Screen.h:53-125, created once at a fixed size, main.cpp:91).  The Sobol table defaults
 * to rt_sobol_default_table (setSobol, main.cpp:98). */
int rt_create(RtContext **out, int width, int height, int device);
void rt_destroy(RtContext *ctx);
/* HIP stream (hipStream_t) the context launches on; NULL = the default stream. */
int rt_set_stream(RtContext *ctx, void *hip_stream);
int rt_default_options(RtOptions *out);
int rt_set_options(RtContext *ctx, const RtOptions *opt);

/* getTexture(tree, shader, objTex), main.cpp:120 / ObjectTexture.h:69-182: takes the
 * builder's output in its in-memory layouts and converts it to the device layout.  The tree
 * must be the reference's shape (pre-order, left child = index+1, one triangle per leaf). */
int rt_scene_upload(RtContext *ctx, const RtTriangle *tris, int64_t n,
                    const RtLinearBVHNode *nodes, int64_t m);
/* Same, from the flat arrays the reference puts into texTriangle / texBVH (33 resp. 9
 * floats per record, records contiguous). */
int rt_scene_upload_textures(RtContext *ctx, const float *tex_triangle, int64_t n,
                             const float *tex_bvh, int64_t m);
/* The image of RtOptions.albedo_texture: width x height texels of 3 bytes (GL_RGB,
 * GL_UNSIGNED_BYTE, tightly packed; row 0 is the row at v = 0, the order glTexImage2D takes them in,
 * Model.h:241), copied to the GPU.  width, height in 1..16384.  rgb == NULL removes the image.
 * Stands where TextureFromFile (Model.h:220-258) stands in the reference, without the file
 * decoding (stb_image there): the caller passes decoded texels. */
int rt_set_albedo_texture(RtContext *ctx, const uint8_t *rgb, int width, int height);
/* setSobol, sobol.h:27-34. */
int rt_set_sobol(RtContext *ctx, const uint32_t v[RT_SOBOL_WORDS]);
/* Scene::setSphere, Scene.h:71-85 (count <= 10). */
int rt_set_spheres(RtContext *ctx, const RtSphere *spheres, int count);
/* The camera.* uniforms of main.cpp:156-162. */
int rt_set_camera(RtContext *ctx, const RtCamera *cam);

/* The per-frame part of the main loop, main.cpp:137-172: renders `count` frames with
 * camera.LoopNum = first_loopnum .. first_loopnum+count-1 and randOrigin[i] for frame i,
 * adding each frame's sample to the accumulators.  Asynchronous on the context's stream. */
int rt_render_frames(RtContext *ctx, int first_loopnum, int count, const float *rand_origin);
/* Camera::LoopNum = 0 (Camera.h:45): clears the accumulators and the counters. */
int rt_reset(RtContext *ctx);
int rt_synchronize(RtContext *ctx);

/* The running mean the reference keeps in its FBO (frag:697, fp32 here instead of the
 * reference's 8-bit texture): mean_N = sample/N + mean_{N-1}*(N-1)/N, valid when frames were
 * rendered in order from LoopNum 1. */
int rt_read_mean(RtContext *ctx, float *rgb);
/* The same image without stopping the pipeline, for a caller that displays after every frame as the
 * reference's loop does (main.cpp:178-197) and can live with showing frame N while frame N+1 is being
 * traced: rt_read_mean_async enqueues the copy of the running mean as it stands after the frames of
 * all calls so far and returns; rt_read_wait blocks until that copy has arrived (rt_synchronize
 * would also wait for frames requested since).  The path tracing of later calls runs beside the
 * copy; their frames are added to the image only after it.  rgb should be page-locked host memory
 * (rt_host_alloc), or the copy is staged and slow.  No counterpart in the reference, whose image
 * never leaves the GPU. */
int rt_read_mean_async(RtContext *ctx, float *rgb);
int rt_read_wait(RtContext *ctx);
/* Page-locked host memory for image reads (hipHostMalloc); NULL when it cannot be had. */
void *rt_host_alloc(size_t bytes);
void rt_host_free(void *p);
/* Device address of the W*H*3 fp32 running mean, for a consumer on the same GPU (what the
 * reference's display pass does with its FBO texture, main.cpp:187-190).  Valid after the work on
 * the context's stream that precedes the read. */
int rt_mean_device_ptr(RtContext *ctx, void **dev_ptr);
/* Plain fp32 sum of all samples since rt_reset (what a multi-GPU reduce adds up). */
int rt_read_sum(RtContext *ctx, float *rgb);
/* Resume an interrupted render: loads a running mean and a sum read earlier with rt_read_mean /
 * rt_read_sum (W*H*3 floats each, NULL keeps the current content) and sets the frame count.
 * The caller continues with first_loopnum = frames + 1.  No counterpart in the reference, whose
 * accumulation lives only in its two FBO textures and is lost on exit. */
int rt_restore(RtContext *ctx, const float *mean_rgb, const float *sum_rgb, uint64_t frames);
/* Device address of the W*H*3 fp32 sum buffer (for a caller-side collective). */
int rt_sum_device_ptr(RtContext *ctx, void **dev_ptr);
/* Accumulate sums into caller-owned device memory (W*H*3 floats) instead; NULL restores the
 * internal buffer.  The buffer is not cleared by this call. */
int rt_bind_sum_buffer(RtContext *ctx, float *dev_ptr);

int rt_get_stats(RtContext *ctx, RtStats *out);
int rt_get_scene_info(RtContext *ctx, RtSceneInfo *out);
/* Device time of the traversal ("extend") kernel launches since the last rt_reset, measured
 * with HIP events on the context's stream.  Synchronises. */
int rt_get_kernel_time(RtContext *ctx, double *total_ms, int64_t *launches);

/* The traversal loop of hitBVH alone (frag:373-456) for caller-supplied rays: origins/dirs are
 * n*3 floats; t_out[i] = ray.hitMin of the kept triangle (its t - 1e-5) or -1 for a miss;
 * id_out[i] = that triangle's index in the uploaded order or -1.  Synchronous; for tests and
 * tools. */
int rt_trace_rays(RtContext *ctx, const float *origins, const float *dirs, int64_t n,
                  float *t_out, int32_t *id_out);

/* rt_trace_rays for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  d_origins/d_dirs
 * (n*3 floats) and d_t_out/d_id_out (n values) are DEVICE pointers on the context's GPU.  The
 * call allocates and frees nothing, on the host or on the device: its working memory (ray queue,
 * hit records, queue counters; 8 words per ray) is a fixed array inside the library, one copy
 * per GPU (8.4 MB, held until the process ends), which limits a call to RT_NOALLOC_MAX_RAYS
 * rays; more is RT_E_INVALID.  Asynchronous on the context's stream (rt_synchronize waits).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two contexts on
 * different streams or threads) race and must be serialised by the caller. */
#define RT_NOALLOC_MAX_RAYS 262144
int rt_trace_rays_noalloc(RtContext *ctx, const float *d_origins, const float *d_dirs, int64_t n,
                          float *d_t_out, int32_t *d_id_out);

/* Diagnostics with no counterpart in the reference, used by the parity tests.
 * rt_selftest_nth_element: the builder's selection routine alone (keys with ids carried
 * along), to compare its permutation with std::nth_element's (BVH.h:222).
 * rt_selftest_arith: the device's a/b, sqrt|a|, fma(a,b,a), and sin/cos of |b| (0 outside [0, 2*pi]), five
 * floats per element, to compare with the host's arithmetic.
 * rt_selftest_exact_math: runs EVERY fp32 bit pattern through the device's 1/x, sqrt(x) and
 * 1/sqrt(x) routines (a Newton step on the hardware estimate where that is exact, the full
 * IEEE expansion elsewhere) and counts the inputs whose result differs from IEEE division /
 * square root, and the pixel coordinate (X + 0.5) / W of every pixel of every width up to 16384
 * (mismatches[3]): all four counts must be 0. */
int rt_selftest_nth_element(float *keys, int32_t *ids, int64_t n, int64_t nth);
/* rt_selftest_index_build: the host build of the library's own search index (RtOptions.accel,
 * RT_INDEX_BUILD_HOST) for tris[0..n) without a GPU: *node_count 4-wide nodes of 64 bytes in the
 * layout the kernels read (origin xyz, scale x, four child refs - interior: node index, leaf:
 * 0x80000000 | triangle number, empty: 0xFFFFFFFF -, 8-bit box bounds qlo xyz / qhi xyz with child c
 * in bits 8c..8c+7, scale y, scale z) copied to nodes_out when it is not NULL; stats = build
 * time in ms, levels of wide nodes, summed surface area of the nodes over the root's (wide nodes a
 * random ray through the root box visits), bound on the deferred-node stack.  scene_max_abs: largest
 * |coordinate| of the scene. */
int rt_selftest_index_build(const RtTriangle *tris, int64_t n, float scene_max_abs, void *nodes_out,
                            int64_t node_capacity, int64_t *node_count, double stats[4]);
int rt_selftest_sort(float *keys, int32_t *ids, int64_t n);     /* the builder's sort against std::sort */
int rt_selftest_arith(RtContext *ctx, const float *a, const float *b, int64_t n, float *out);
/* The two device-wide primitives of the index build on the GPU, on host arrays: a stable radix sort of
 * n (64-bit key, 32-bit value) pairs by key, in place; the exclusive prefix sums of n 64-bit values, in place. */
int rt_selftest_device_sort(RtContext *ctx, uint64_t *keys, uint32_t *vals, int64_t n);
int rt_selftest_device_scan(RtContext *ctx, uint64_t *values, int64_t n);
int rt_selftest_exact_math(RtContext *ctx, uint64_t mismatches[4]);

const char *rt_last_error(const RtContext *ctx);    /* ctx may be NULL: last error of a call without a context */
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
