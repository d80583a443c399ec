// Host-side pieces of the shader-pass boundary that need no GPU: the Sobol table and its
// CPU twin, the camera uniforms, the explicit randOrigin generator, error strings.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "rt_internal.h"

namespace rt {

static std::mutex g_err_mutex;
static std::string g_err;
static thread_local std::string g_err_copy;

int host_threads() {
    if (const char *env = getenv("RT_AMD_HOST_THREADS")) { const int v = atoi(env); if (v >= 1 && v <= 64) return v; }
    return (int)std::min<unsigned>(std::max(std::thread::hardware_concurrency(), 1u), 16u);
}

int host_fork_levels() {
    int levels = 0;
    for (int t = host_threads(); t > 1; t >>= 1) levels++;
    return levels;
}

void set_global_error(const std::string &msg) {
    std::lock_guard<std::mutex> lock(g_err_mutex);
    g_err = msg;
}
const char *global_error() {
    std::lock_guard<std::mutex> lock(g_err_mutex);
    g_err_copy = g_err;
    return g_err_copy.c_str();
}

}  // namespace rt

namespace {

// Joe & Kuo "new-joe-kuo-6.21201" parameters of Sobol dimensions 2..16: degree s of the
// primitive polynomial, its inner coefficients a, and the initial direction integers m_i.
// Dimension 1 is the van der Corput sequence.  Expanding them with the standard recurrence
// gives exactly the 16 x 32 table the reference hard-codes (sobol.h:8-25, v_matrix.txt).
struct JoeKuo { int s; uint32_t a; uint32_t m[6]; };
const JoeKuo kJoeKuo[15] = {
    {1, 0, {1}},
    {2, 1, {1, 3}},
    {3, 1, {1, 3, 1}},
    {3, 2, {1, 1, 1}},
    {4, 1, {1, 1, 3, 3}},
    {4, 4, {1, 3, 5, 13}},
    {5, 2, {1, 1, 5, 5, 17}},
    {5, 4, {1, 1, 5, 5, 5}},
    {5, 7, {1, 1, 7, 11, 19}},
    {5, 11, {1, 1, 5, 1, 1}},
    {5, 13, {1, 1, 1, 3, 11}},
    {5, 14, {1, 3, 5, 5, 31}},
    {6, 1, {1, 3, 3, 9, 7, 49}},
    {6, 13, {1, 1, 1, 15, 21, 21}},
    {6, 16, {1, 3, 1, 13, 27, 49}},
};

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }  // glm::radians

inline void normalize3(float v[3]) {
    float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);  // glm::normalize = v * inversesqrt(dot)
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}
inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - b[1] * a[2];
    o[1] = a[2] * b[0] - b[2] * a[0];
    o[2] = a[0] * b[1] - b[0] * a[1];
}

}  // namespace

extern "C" {

int rt_sobol_default_table(uint32_t v[RT_SOBOL_WORDS]) {
    if (!v) { rt::set_global_error("rt_sobol_default_table: null pointer"); return RT_E_INVALID; }
    for (int j = 0; j < 32; j++) v[j] = 1u << (31 - j);
    for (int d = 1; d < 16; d++) {
        const JoeKuo &p = kJoeKuo[d - 1];
        uint32_t *o = v + 32 * d;
        for (int i = 0; i < p.s; i++) o[i] = p.m[i] << (31 - i);
        for (int i = p.s; i < 32; i++) {
            uint32_t x = o[i - p.s] ^ (o[i - p.s] >> p.s);
            for (int k = 1; k < p.s; k++) x ^= ((p.a >> (p.s - 1 - k)) & 1u) * o[i - k];
            o[i] = x;
        }
    }
    return RT_OK;
}

uint32_t rt_gray_code(uint32_t i) { return i ^ (i >> 1); }     // sobol.h:37-39

float rt_sobol(const uint32_t v[RT_SOBOL_WORDS], uint32_t d, uint32_t i) {   // sobol.h:42-50
    uint32_t result = 0, offset = d * 32;
    for (uint32_t j = 0; i; i >>= 1, j++)
        if (i & 1) result ^= v[j + offset];
    return (float)result * (1.0f / (float)0xFFFFFFFFu);
}

int rt_camera_from_angles(RtCamera *out, const float pos[3], float yaw, float pitch, float fov,
                          int width, int height) {
    if (!out || !pos || width <= 0 || height <= 0) { rt::set_global_error("rt_camera_from_angles: bad argument"); return RT_E_INVALID; }
    std::memset(out, 0, sizeof *out);
    const float ratio = (float)width / (float)height;               // Camera.h:26
    out->halfH = std::tan(radians(fov));                            // Camera.h:27
    out->halfW = out->halfH * ratio;                                // Camera.h:28
    float front[3] = {std::cos(radians(yaw)) * std::cos(radians(pitch)),     // Camera.h:34-37
                      std::sin(radians(pitch)),
                      std::sin(radians(yaw)) * std::cos(radians(pitch))};
    normalize3(front);
    const float worldUp[3] = {0.0f, 1.0f, 0.0f};
    float right[3], up[3];
    cross3(front, worldUp, right); normalize3(right);               // Camera.h:39
    cross3(right, front, up); normalize3(up);                       // Camera.h:41
    for (int c = 0; c < 3; c++) {
        out->camPos[c] = pos[c];
        out->front[c] = front[c];
        out->right[c] = right[c];
        out->up[c] = up[c];
        out->leftbottom[c] = front[c] - out->halfW * right[c] - out->halfH * up[c];   // Camera.h:43
    }
    out->LoopNum = 0;                                               // Camera.h:45
    return RT_OK;
}

int rt_camera_default(RtCamera *out, int width, int height) {
    const float pos[3] = {0.0f, 2.0f, 5.0f};                        // Camera.h:20-24
    int rc = rt_camera_from_angles(out, pos, -90.0f, 0.0f, 45.0f, width, height);
    if (rc != RT_OK) return rc;
    // With yaw -90 and pitch 0 the trigonometric values are 0, -1 and 1 up to rounding
    // (cos(-pi/2) is 4e-8 in fp32); the camera this library calls "default" uses the exact
    // axis-aligned frame so that results do not depend on the host libm.
    const float front[3] = {0.0f, 0.0f, -1.0f}, right[3] = {1.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    out->halfH = 1.0f;
    out->halfW = out->halfH * ((float)width / (float)height);
    for (int c = 0; c < 3; c++) {
        out->front[c] = front[c]; out->right[c] = right[c]; out->up[c] = up[c];
        out->leftbottom[c] = front[c] - out->halfW * right[c] - out->halfH * up[c];
    }
    return RT_OK;
}

// ---- class Camera (Camera.h:16-139), field for field

static void rig_corner(RtCameraRig *r) {                            // Camera.h:43,57,92,107
    for (int c = 0; c < 3; c++)
        r->LeftBottomCorner[c] = r->cameraFront[c] - r->halfW * r->cameraRight[c] - r->halfH * r->cameraUp[c];
}

int rt_camera_rig_update_vectors(RtCameraRig *r) {
    if (!r) { rt::set_global_error("rt_camera_rig_update_vectors: null pointer"); return RT_E_INVALID; }
    r->cameraFront[0] = std::cos(radians(r->Yaw)) * std::cos(radians(r->Pitch));
    r->cameraFront[1] = std::sin(radians(r->Pitch));
    r->cameraFront[2] = std::sin(radians(r->Yaw)) * std::cos(radians(r->Pitch));
    normalize3(r->cameraFront);
    cross3(r->cameraFront, r->worldUp, r->cameraRight); normalize3(r->cameraRight);
    cross3(r->cameraRight, r->cameraFront, r->cameraUp); normalize3(r->cameraUp);
    rig_corner(r);
    r->LoopNum = 0;
    return RT_OK;
}

int rt_camera_rig_init(RtCameraRig *r, int width, int height) {
    if (!r || width <= 0 || height <= 0) { rt::set_global_error("rt_camera_rig_init: bad argument"); return RT_E_INVALID; }
    std::memset(r, 0, sizeof *r);
    r->LoopNum = 0;
    r->Yaw = -90.0f; r->Pitch = 0.0f; r->fov = 45.0f;
    r->firstMouse = 1;
    r->cameraPos[0] = 0.0f; r->cameraPos[1] = 2.0f; r->cameraPos[2] = 5.0f;
    r->worldUp[0] = 0.0f; r->worldUp[1] = 1.0f; r->worldUp[2] = 0.0f;
    r->ScreenRatio = (float)width / (float)height;
    r->halfH = std::tan(radians(r->fov));
    r->halfW = r->halfH * r->ScreenRatio;
    r->cameraSpeed = 1.0f;
    return rt_camera_rig_update_vectors(r);
}

int rt_camera_rig_update_fov(RtCameraRig *r, float offset) {
    if (!r) { rt::set_global_error("rt_camera_rig_update_fov: null pointer"); return RT_E_INVALID; }
    r->fov -= offset;
    if (r->fov < 1.0f) r->fov = 1.0f;
    if (r->fov > 45.0f) r->fov = 45.0f;
    r->halfH = std::tan(radians(r->fov));
    r->halfW = r->halfH * r->ScreenRatio;
    rig_corner(r);
    r->LoopNum = 0;
    return RT_OK;
}

int rt_camera_rig_update_front(RtCameraRig *r, float xpos, float ypos) {
    if (!r) { rt::set_global_error("rt_camera_rig_update_front: null pointer"); return RT_E_INVALID; }
    if (r->firstMouse) { r->lastX = xpos; r->lastY = ypos; r->firstMouse = 0; }
    float xoffset = xpos - r->lastX, yoffset = r->lastY - ypos;
    r->lastX = xpos; r->lastY = ypos;
    const float sensitivity = 0.03f;
    xoffset *= sensitivity; yoffset *= sensitivity;
    r->Yaw += xoffset; r->Pitch += yoffset;
    if (r->Pitch > 89.0f) r->Pitch = 89.0f;
    if (r->Pitch < -89.0f) r->Pitch = -89.0f;
    return rt_camera_rig_update_vectors(r);
}

int rt_camera_rig_update_screen_ratio(RtCameraRig *r, int width, int height) {
    if (!r || width <= 0 || height <= 0) { rt::set_global_error("rt_camera_rig_update_screen_ratio: bad argument"); return RT_E_INVALID; }
    r->ScreenRatio = (float)width / (float)height;
    r->halfW = r->halfH * r->ScreenRatio;
    rig_corner(r);
    r->LoopNum = 0;
    return RT_OK;
}

int rt_camera_rig_process_keyboard(RtCameraRig *r, int direction, float delta_time) {
    if (!r || direction < RT_CAMERA_FORWARD || direction > RT_CAMERA_RIGHT) { rt::set_global_error("rt_camera_rig_process_keyboard: bad argument"); return RT_E_INVALID; }
    const float velocity = r->cameraSpeed * delta_time;
    for (int c = 0; c < 3; c++) {
        if (direction == RT_CAMERA_FORWARD) r->cameraPos[c] += r->cameraFront[c] * velocity;
        if (direction == RT_CAMERA_BACKWARD) r->cameraPos[c] -= r->cameraFront[c] * velocity;
        if (direction == RT_CAMERA_LEFT) r->cameraPos[c] -= r->cameraRight[c] * velocity;
        if (direction == RT_CAMERA_RIGHT) r->cameraPos[c] += r->cameraRight[c] * velocity;
    }
    rig_corner(r);
    r->LoopNum = 0;
    return RT_OK;
}

int rt_camera_rig_loop_increase(RtCameraRig *r) {
    if (!r) { rt::set_global_error("rt_camera_rig_loop_increase: null pointer"); return RT_E_INVALID; }
    r->LoopNum++;
    return RT_OK;
}

int rt_camera_rig_uniforms(const RtCameraRig *r, RtCamera *out) {
    if (!r || !out) { rt::set_global_error("rt_camera_rig_uniforms: null pointer"); return RT_E_INVALID; }
    for (int c = 0; c < 3; c++) {
        out->camPos[c] = r->cameraPos[c]; out->front[c] = r->cameraFront[c];
        out->right[c] = r->cameraRight[c]; out->up[c] = r->cameraUp[c];
        out->leftbottom[c] = r->LeftBottomCorner[c];
    }
    out->halfH = r->halfH; out->halfW = r->halfW;
    out->LoopNum = r->LoopNum;
    return RT_OK;
}

float rt_rand_origin(uint32_t *state) {
    // xorshift32 (Marsaglia 13/17/5); the top 24 bits give u in [0,1) exactly in fp32.
    uint32_t x = *state ? *state : 0x9E3779B9u;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *state = x;
    float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return 874264.0f * (u + 1.0f);                                  // main.cpp:149
}

const char *rt_version(void) { return "rt_amd 0.1 (gfx950)"; }

}  // extern "C"
