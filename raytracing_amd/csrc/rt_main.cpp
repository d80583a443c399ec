// rt_render: the reference's main loop (main.cpp:50-208) without the window, on top of the C ABI.
//
//   rt_render [--demo | --obj FILE SX SY SZ TX TY TZ ... | --synthetic KIND LEVEL COPIES]
//             [--sphere CX CY CZ R AR AG AB DIFFUSE_RATE MIRROR_RATE] [--size W H] [--spp N]
//             [--bounces B] [--seed S] [--camera PX PY PZ YAW PITCH FOV] [--exact] [--megakernel | --wavefront] [--history8]
//             [--shading-normals] [--texture image.ppm] [--gpu-index | --host-index]
//             [--frames-per-call N] [--show] [--out image.pfm] [--ppm image.ppm]
//
// --frames-per-call 1 is the reference's own loop (one draw per iteration, main.cpp:171); the
// default hands 64 frames to a call.  --show fetches the image after every call the way a window
// would show it (main.cpp:178-197), with rt_read_mean_async into page-locked memory: the image of
// one call arrives while the next call is traced.
//
// Loads the scene (Model + Scene::pushModel), builds the BVH in the reference's node order
// (BVHTree::build / buildLinerTree), uploads it (getTexture), then renders N frames with
// LoopNum = 1..N (cam.LoopIncrease) and an explicit randOrigin stream, and writes the running
// mean.  The PPM is what the reference's window would show: clamped to [0,1], 8 bits, no gamma
// (ScreenFragmentShader.glsl:10-14).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
#include "../../include/rt_amd.h"
}

static int die(const char *what, RtContext *ctx) {
    std::fprintf(stderr, "rt_render: %s: %s\n", what, rt_last_error(ctx));
    return 1;
}

int main(int argc, char **argv) {
    int W = 1200, H = 800, spp = 16, bounces = 4;     // SCR_WIDTH/SCR_HEIGHT of main.cpp:29-30
    uint32_t seed = 20221003u;
    bool exact = false, mega = false, wavefront = false, history8 = false, shadingNormals = false, gpuIndex = false, hostIndex = false, customCam = false, haveGeometry = false;
    int framesPerCall = 64;       // the path kernel renders up to 64 frames per launch
    bool show = false;
    float camArgs[6] = {0, 2, 5, -90, 0, 45};
    std::string outPfm, outPpm, texturePpm;
    RtScene *scene = nullptr;
    if (rt_scene_create(&scene) != RT_OK) return die("rt_scene_create", nullptr);
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { std::fprintf(stderr, "rt_render: %s needs %d values\n", a.c_str(), n); std::exit(2); } };
        if (a == "--demo") { if (rt_scene_push_demo(scene) != RT_OK) return die("demo scene", nullptr); haveGeometry = true; }
        else if (a == "--obj") {
            need(7);
            const float s[3] = {(float)atof(argv[i + 2]), (float)atof(argv[i + 3]), (float)atof(argv[i + 4])};
            const float t[3] = {(float)atof(argv[i + 5]), (float)atof(argv[i + 6]), (float)atof(argv[i + 7])};
            if (rt_scene_push_obj(scene, argv[i + 1], s, t) != RT_OK) return die("load OBJ", nullptr);
            haveGeometry = true; i += 7;
        } else if (a == "--synthetic") {
            need(3);
            if (rt_scene_push_synthetic(scene, atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), 12345u) != RT_OK) return die("synthetic scene", nullptr);
            haveGeometry = true; i += 3;
        } else if (a == "--sphere") {
            need(9);
            RtSphere s;
            for (int k = 0; k < 3; k++) s.center[k] = (float)atof(argv[i + 1 + k]);
            s.radius = (float)atof(argv[i + 4]);
            for (int k = 0; k < 3; k++) s.albedo[k] = (float)atof(argv[i + 5 + k]);
            s.specularRate = (float)atof(argv[i + 8]); s.refractionRate = (float)atof(argv[i + 9]); s.roughness = 0.0f;
            if (rt_scene_push_sphere(scene, &s) != RT_OK) return die("sphere", nullptr);
            i += 9;
        } else if (a == "--size") { need(2); W = atoi(argv[i + 1]); H = atoi(argv[i + 2]); i += 2; }
        else if (a == "--spp") { need(1); spp = atoi(argv[++i]); }
        else if (a == "--bounces") { need(1); bounces = atoi(argv[++i]); }
        else if (a == "--seed") { need(1); seed = (uint32_t)strtoul(argv[++i], nullptr, 10); }
        else if (a == "--camera") { need(6); for (int k = 0; k < 6; k++) camArgs[k] = (float)atof(argv[i + 1 + k]); customCam = true; i += 6; }
        else if (a == "--exact") exact = true;
        else if (a == "--megakernel") mega = true;
        else if (a == "--wavefront") wavefront = true;
        else if (a == "--history8") history8 = true;
        else if (a == "--shading-normals") shadingNormals = true;     // interpolated vertex normals (not in the reference)
        else if (a == "--texture") { need(1); texturePpm = argv[++i]; }  // diffuse albedo map at the models' vt coordinates (not in the reference)
        else if (a == "--gpu-index") gpuIndex = true;                 // search index built on the device
        else if (a == "--host-index") hostIndex = true;               // ... on the host (default: by scene size)
        else if (a == "--frames-per-call") { need(1); framesPerCall = atoi(argv[++i]); }
        else if (a == "--show") show = true;
        else if (a == "--out") { need(1); outPfm = argv[++i]; }
        else if (a == "--ppm") { need(1); outPpm = argv[++i]; }
        else { std::fprintf(stderr, "rt_render: unknown option %s\n", a.c_str()); return 2; }
    }
    if (!haveGeometry && rt_scene_push_demo(scene) != RT_OK) return die("demo scene", nullptr);
    if (spp < 1) { std::fprintf(stderr, "rt_render: --spp must be >= 1\n"); return 2; }
    if (framesPerCall < 1) { std::fprintf(stderr, "rt_render: --frames-per-call must be >= 1\n"); return 2; }

    // tree.build(scene.meshes); tree.buildLinerTree();   main.cpp:113-115
    const int64_t n = rt_scene_triangle_count(scene);
    std::vector<RtTriangle> tris((size_t)n);
    std::vector<RtLinearBVHNode> nodes((size_t)(2 * n - 1));
    if (rt_scene_get_triangles(scene, tris.data(), n) != RT_OK) return die("scene triangles", nullptr);
    int64_t m = 0;
    auto t0 = std::chrono::steady_clock::now();
    if (rt_bvh_build(tris.data(), n, nodes.data(), (int64_t)nodes.size(), &m) != RT_OK) return die("rt_bvh_build", nullptr);
    const double bvhMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("triangles: %lld\nbvh nodes: %lld (%.1f ms)\n", (long long)n, (long long)m, bvhMs);   // ObjectTexture.h:73-74

    RtContext *ctx = nullptr;
    if (rt_create(&ctx, W, H, 0) != RT_OK) return die("rt_create", nullptr);
    RtOptions opt;
    rt_default_options(&opt);
    opt.max_bounces = bounces;
    opt.accel = exact ? 0 : 1;
    opt.history8 = history8 ? 1 : 0;      // the reference keeps its running mean in an 8-bit texture (RT_Screen.h:61)
    opt.shading_normals = shadingNormals ? 1 : 0;
    opt.albedo_texture = texturePpm.empty() ? 0 : 1;
    opt.index_build = gpuIndex ? RT_INDEX_BUILD_GPU : (hostIndex ? RT_INDEX_BUILD_HOST : RT_INDEX_BUILD_AUTO);
    opt.pipeline = mega ? RT_PIPELINE_MEGAKERNEL : (wavefront ? RT_PIPELINE_WAVEFRONT : RT_PIPELINE_PERSISTENT);
    // before the upload: index_build and shading_normals are read there
    if (rt_set_options(ctx, &opt) != RT_OK) return die("rt_set_options", ctx);
    if (rt_scene_upload(ctx, tris.data(), n, nodes.data(), m) != RT_OK) return die("rt_scene_upload", ctx);   // getTexture, main.cpp:120
    if (!texturePpm.empty()) {
        // binary PPM, 8 bits per channel.  Rows go to the library in file order (top row first), as
        // stbi_load hands them to glTexImage2D in TextureFromFile (Model.h:229-241); the loader has
        // flipped v like aiProcess_FlipUVs (Model.h:58), so the image appears upright.
        FILE *f = std::fopen(texturePpm.c_str(), "rb");
        if (!f) { std::perror(texturePpm.c_str()); return 1; }
        int tw = 0, th = 0, maxv = 0;
        auto next_int = [&](int &v) {
            int c = std::fgetc(f);
            while (c == '#' || c == ' ' || c == '\n' || c == '\r' || c == '\t') {
                if (c == '#') while (c != '\n' && c != EOF) c = std::fgetc(f);
                else c = std::fgetc(f);
            }
            v = 0;
            bool any = false;
            while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); any = true; if (v > 1 << 20) return false; c = std::fgetc(f); }
            return any;     // the one whitespace character after the number is consumed
        };
        const bool magic = std::fgetc(f) == 'P' && std::fgetc(f) == '6';
        if (!magic || !next_int(tw) || !next_int(th) || !next_int(maxv) || maxv != 255 || tw < 1 || th < 1 || tw > 16384 || th > 16384) {
            std::fprintf(stderr, "rt_render: %s: not a binary PPM (P6) with 8 bits per channel, at most 16384 x 16384\n", texturePpm.c_str());
            std::fclose(f);
            return 1;
        }
        std::vector<uint8_t> texels((size_t)tw * th * 3);
        const size_t got = std::fread(texels.data(), 1, texels.size(), f);
        std::fclose(f);
        if (got != texels.size()) { std::fprintf(stderr, "rt_render: %s: file ends early\n", texturePpm.c_str()); return 1; }
        if (rt_set_albedo_texture(ctx, texels.data(), tw, th) != RT_OK) return die("rt_set_albedo_texture", ctx);
    }
    std::vector<RtSphere> spheres((size_t)rt_scene_sphere_count(scene));
    if (!spheres.empty() && rt_scene_get_spheres(scene, spheres.data(), (int)spheres.size()) != RT_OK) return die("scene spheres", nullptr);
    if (rt_set_spheres(ctx, spheres.data(), (int)spheres.size()) != RT_OK) return die("rt_set_spheres", ctx);     // scene.setSphere, main.cpp:169
    RtCamera cam;
    if (customCam) { if (rt_camera_from_angles(&cam, camArgs, camArgs[3], camArgs[4], camArgs[5], W, H) != RT_OK) return die("camera", nullptr); }
    else if (rt_camera_default(&cam, W, H) != RT_OK) return die("camera", nullptr);
    if (rt_set_camera(ctx, &cam) != RT_OK) return die("rt_set_camera", ctx);                                      // main.cpp:156-163

    // the render loop, main.cpp:124-176, in batches of frames
    std::vector<float> ro((size_t)spp);
    uint32_t state = seed;
    for (int f = 0; f < spp; f++) ro[(size_t)f] = rt_rand_origin(&state);                                         // main.cpp:149
    t0 = std::chrono::steady_clock::now();
    const int batch = framesPerCall;
    const size_t imageBytes = (size_t)W * H * 3 * sizeof(float);
    float *shown[2] = {nullptr, nullptr};       // what a window would display: the image of the call before the one being traced
    if (show) {
        for (float *&p : shown) { p = (float *)rt_host_alloc(imageBytes); if (!p) return die("rt_host_alloc", nullptr); }
    }
    int calls = 0;
    double shownSum = 0.0;
    for (int first = 1; first <= spp; first += batch, calls++) {
        const int cnt = (spp - first + 1 < batch) ? spp - first + 1 : batch;
        if (rt_render_frames(ctx, first, cnt, ro.data() + (first - 1)) != RT_OK) return die("rt_render_frames", ctx);
        if (show) {
            if (rt_read_wait(ctx) != RT_OK) return die("rt_read_wait", ctx);            // the previous call's image is there ...
            if (calls > 0) shownSum += shown[(calls - 1) & 1][((size_t)(H / 2) * W + W / 2) * 3];  // ... a window would draw it now
            if (rt_read_mean_async(ctx, shown[calls & 1]) != RT_OK) return die("rt_read_mean_async", ctx);
        }
    }
    if (show && rt_read_wait(ctx) != RT_OK) return die("rt_read_wait", ctx);
    if (rt_synchronize(ctx) != RT_OK) return die("rt_synchronize", ctx);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    RtStats st;
    if (rt_get_stats(ctx, &st) != RT_OK) return die("rt_get_stats", ctx);
    std::printf("LoopNum: %d  frame time: %.3f ms  rays: %llu  %.1f Mrays/s\n", spp, ms / spp, (unsigned long long)st.rays, (double)st.rays / ms * 1e-3);

    if (show) std::printf("shown: %d images, centre pixel of the last one %.6f\n", calls, (double)shown[(calls - 1) & 1][((size_t)(H / 2) * W + W / 2) * 3]);
    (void)shownSum;
    std::vector<float> rgb((size_t)W * H * 3);
    if (rt_read_mean(ctx, rgb.data()) != RT_OK) return die("rt_read_mean", ctx);
    if (show && std::memcmp(rgb.data(), shown[(calls - 1) & 1], imageBytes) != 0) { std::fprintf(stderr, "rt_render: the last image shown differs from the final image\n"); return 1; }
    for (float *p : shown) rt_host_free(p);
    if (!outPfm.empty()) {          // PFM stores rows bottom to top, as the accumulation buffer does
        FILE *f = std::fopen(outPfm.c_str(), "wb");
        if (!f) { std::perror(outPfm.c_str()); return 1; }
        std::fprintf(f, "PF\n%d %d\n-1.0\n", W, H);
        std::fwrite(rgb.data(), sizeof(float), rgb.size(), f);
        std::fclose(f);
    }
    if (!outPpm.empty()) {          // top row first: flip
        FILE *f = std::fopen(outPpm.c_str(), "wb");
        if (!f) { std::perror(outPpm.c_str()); return 1; }
        std::fprintf(f, "P6\n%d %d\n255\n", W, H);
        std::vector<unsigned char> row((size_t)W * 3);
        for (int y = H - 1; y >= 0; y--) {
            for (int i = 0; i < W * 3; i++) {
                float v = rgb[(size_t)y * W * 3 + i];
                v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
                row[(size_t)i] = (unsigned char)(v * 255.0f + 0.5f);
            }
            std::fwrite(row.data(), 1, row.size(), f);
        }
        std::fclose(f);
    }
    rt_destroy(ctx);
    rt_scene_destroy(scene);
    return 0;
}
