// What the shader does around the BVH loop (device code, gfx950): random numbers, the camera
// ray, material draw, spheres, scattering, Russian roulette (frag:36-104, 281-300, 457-689).
// Included by rt_device.hip only; compile with -ffp-contract=off (rt_math.h).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rt_kernels.h"
#include "rt_traverse.h"

namespace rtd {

__device__ __forceinline__ float rand01(uint32_t &wseed) {     // frag:36-47
    wseed = rtm::wang_step(wseed);
    return (float)wseed * (1.0f / 4294967296.0f);
}

// frag:84-104 for one bounce: dims 2b and 2b+1 of frame `loopNum`.  V is the context's copy of
// the direction numbers (uniform uint V[512], frag:81), read-only device memory; loopNum and b
// are wave uniform, so the words come through the scalar unit's constant cache.
__device__ __forceinline__ void sobol_pair(const uint32_t *__restrict__ V, uint32_t loopNum, uint32_t b, float &u, float &v) {
    uint32_t g = loopNum ^ (loopNum >> 1);
    uint32_t r0 = 0u, r1 = 0u;
    const uint32_t base = b * 64u;
    for (uint32_t j = 0u; g != 0u; g >>= 1, j++)
        if (g & 1u) {
            r0 ^= V[base + j];
            r1 ^= V[base + 32u + j];
        }
    u = (float)r0 * (1.0f / 4294967296.0f);
    v = (float)r1 * (1.0f / 4294967296.0f);
}

// frag:59-77.  The shader recovers the pixel from the interpolated coordinate,
// x = uint(((uv.x * 2 - 1) * 0.5 + 0.5) * width) with uv.x = (X + 0.5) / width; in fp32 that IS X for
// every X < width <= 16384 (the rounding errors add up to < 0.01 of a pixel against a margin of
// 0.5; tests/test_sobol.py checks every pair), so the pixel is passed in directly.
__device__ __forceinline__ void cranley_patterson(uint32_t X, uint32_t Y, float &px, float &py) {
    uint32_t pseed = (X * 1973u + Y * 9277u + (uint32_t)(114514 / 1919) * 26699u) | 1u;
    pseed = rtm::wang_step(pseed);
    const float u = (float)pseed / 4294967296.0f;
    pseed = rtm::wang_step(pseed);
    const float v = (float)pseed / 4294967296.0f;
    float x = px + u;
    if (x > 1.0f) x -= 1.0f;
    if (x < 0.0f) x += 1.0f;
    float y = py + v;
    if (y > 1.0f) y -= 1.0f;
    if (y < 0.0f) y += 1.0f;
    px = x; py = y;
}

struct Path {
    vec3 o, d, thr;
    uint32_t seed;
};

// hitSphere, frag:281-293
template <typename SphereT>
__device__ __forceinline__ float sphere_test(const SphereT &s, const vec3 o, const vec3 d) {
    const vec3 oc = rtm::sub(o, rtm::mk(s.center[0], s.center[1], s.center[2]));
    const float a = rtm::dot(d, d);
    const float b = 2.0f * rtm::dot(oc, d);
    const float c = rtm::madd(-s.radius, s.radius, rtm::dot(oc, oc));
    const float disc = rtm::madd(b, b, -((4.0f * a) * c));
    if (disc > 0.0f) {
        const float dis = (-b - __builtin_sqrtf(disc)) / (2.0f * a);
        return (dis > 0.0f) ? dis : -1.0f;
    }
    return -1.0f;
}

// frag:494-500: tangent and bitangent of a normal
__host__ __device__ __forceinline__ void tangent_frame(const vec3 nrm, vec3 &tangent, vec3 &bitangent) {
    const vec3 helper = (fabsf(nrm.x) > 0.999f) ? rtm::mk(0.0f, 0.0f, 1.0f) : rtm::mk(1.0f, 0.0f, 0.0f);
    tangent = rtm::normalize(rtm::cross(nrm, helper));
    bitangent = rtm::normalize(rtm::cross(nrm, tangent));
}

// Everything the shader does with a ray after the BVH loop: the rest of hitBVH
// (frag:457-488), the sphere loop of hitWorld (frag:570-614), the material switch of
// shading (frag:631-670), the 4*pi scaling (frag:672-673) and the Russian-roulette draw that
// opens the next iteration (frag:625-630).  sobU/sobV are the frame's Sobol numbers for this
// bounce (sobol_pair).  Returns true when the path goes on (p holds the next ray); otherwise
// `sample` is the frame's radiance for the pixel.
// EXT: the extensions RtOptions.shading_normals (S.vnrm set: shade with interpolated vertex
// normals) and RtOptions.albedo_texture (S.texels set: diffuse albedo times a texel); kernels
// that offer them are separate instantiations, so the default ones pay neither registers nor
// branches for them.
// FrameT / SceneT: FrameD / SceneD, possibly qualified with an address space (the path kernel reads
// them from the kernel-argument segment at the point of use instead of holding them in registers).
template <bool EXT = false, typename FrameT = FrameD, typename SceneT = SceneD>
__device__ __forceinline__ bool shade_hit(const FrameT &F, const SceneT &S, Path &p, float tTri, uint32_t tri,
                                          int bounce, float sobU, float sobV, uint32_t pixX, uint32_t pixY, vec3 &sample) {
    sample = rtm::mk(0.0f, 0.0f, 0.0f);
    // The triangle part runs for every lane (a lane without a triangle reads record 0 and drops the
    // result): no lane-dependent branch, and every value below is defined on one path only.
    const bool triHit = tri != REF_NONE;
    const float4 *m = RT_REC(S.mats, triHit ? tri : 0u, 80u);
    const float4 m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4];
    vec3 pos = rtm::along(p.o, tTri, p.d);                                      // frag:458
    vec3 nrm = rtm::mk(m2.y, m2.z, m2.w);                                       // frag:296-300, precomputed
    // tangent frame of the hit: taken from the triangle's record, computed for a sphere
    vec3 tangent = rtm::mk(m3.x, m3.y, m3.z);
    vec3 bitangent = rtm::mk(m3.w, m4.x, m4.y);
    bool recFrame = true;
    if (rtm::dot(nrm, p.d) > 0.0f) { nrm = rtm::neg(nrm); tangent = rtm::neg(tangent); }
    vec3 texel = rtm::mk(1.0f, 1.0f, 1.0f);
    if constexpr (EXT) if (triHit) {
        // both extensions weigh the vertex attributes with the edge functions of hitTriangle
        // (frag:265-275) at the hit position: w0 belongs to v0 (edge v1->v2), w1 to v1, w2 to v2
        const float4 *g = S.tris + (size_t)tri * TRI_RECORD;
        const float4 g0 = g[0], g1 = g[1], g2 = g[2];
        const vec3 v0 = rtm::mk(g0.x, g0.y, g0.z), v1 = rtm::mk(g0.w, g1.x, g1.y), v2 = rtm::mk(g1.z, g1.w, g2.x);
        const vec3 N = rtm::mk(g2.y, g2.z, g2.w);
        const float w2 = rtm::dot(N, rtm::cross(rtm::sub(v1, v0), rtm::sub(pos, v0)));
        const float w0 = rtm::dot(N, rtm::cross(rtm::sub(v2, v1), rtm::sub(pos, v1)));
        const float w1 = rtm::dot(N, rtm::cross(rtm::sub(v0, v2), rtm::sub(pos, v2)));
        if (S.vnrm) {
            // RtOptions.shading_normals: the interpolated vertex normal, normalised, turned to the
            // side of the face normal; a zero, infinite or NaN sum keeps the face normal
            const float4 *vn = S.vnrm + (size_t)tri * 3;
            const float4 a0 = vn[0], a1 = vn[1], a2 = vn[2];
            vec3 n = rtm::mk(rtm::madd(w2, a1.z, rtm::madd(w1, a0.w, w0 * a0.x)),
                             rtm::madd(w2, a1.w, rtm::madd(w1, a1.x, w0 * a0.y)),
                             rtm::madd(w2, a2.x, rtm::madd(w1, a1.y, w0 * a0.z)));
            const float l2 = rtm::dot(n, n);
            if ((l2 > 0.0f) && (l2 < INFINITY)) {
                n = rtm::scale(n, rtm::rsqrt_exact(l2));
                if (rtm::dot(n, nrm) < 0.0f) n = rtm::neg(n);
                nrm = n;
                recFrame = false;
            }
        }
        if (S.texels) {
            // RtOptions.albedo_texture: texture coordinate = weighted mean of uv0 uv1 uv2, wrapped
            // as GL_REPEAT (fractional part), nearest texel; a coordinate that is not finite reads
            // texel 0; a byte b is the colour b / 255
            const float4 *uq = S.uvs + (size_t)tri * 2;
            const float4 u0 = uq[0], u1 = uq[1];
            const float sum = (w0 + w1) + w2;
            float cu = rtm::madd(w2, u1.x, rtm::madd(w1, u0.z, w0 * u0.x)) / sum;
            float cv = rtm::madd(w2, u1.y, rtm::madd(w1, u0.w, w0 * u0.y)) / sum;
            if (!(fabsf(cu) < INFINITY)) cu = 0.0f;
            if (!(fabsf(cv) < INFINITY)) cv = 0.0f;
            float fu = cu - floorf(cu), fv = cv - floorf(cv);
            if (fu >= 1.0f) fu = 0.0f;          // a tiny negative coordinate has the fractional part 1 in fp32: GL_REPEAT wraps it to texel 0
            if (fv >= 1.0f) fv = 0.0f;
            const int tx = min((int)(fu * (float)S.texW), S.texW - 1);
            const int ty = min((int)(fv * (float)S.texH), S.texH - 1);
            const uint8_t *px = S.texels + ((size_t)ty * (size_t)S.texW + (size_t)tx) * 3;
            texel = rtm::mk((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
        }
    }
    uint32_t seedT = p.seed;
    const float tempT = rand01(seedT);                                          // frag:462
    if (triHit) p.seed = seedT;
    const vec3 ka = rtm::mk(m0.x, m0.y, m0.z), kd = rtm::mk(m0.w, m1.x, m1.y);
    const float ksx = m1.z, ksy = m1.w;
    int material;
    vec3 albedo = rtm::mk(1.0f, 1.0f, 1.0f);
    float refraction = 1.0f;
    if (rtm::dot(ka, ka) > 0.0f) { material = 0; albedo = ka; }                 // length(ka) > 0
    else if (tempT < ksx) { material = 1; albedo = EXT ? rtm::mul(kd, texel) : kd; }
    else if (tempT < ksy) { material = 2; }
    else { material = 3; refraction = 1.5f; }
    bool hit = triHit && tTri > 0.0f && tTri < HIT_MIN_INIT;                    // frag:562-565
    float hitMin = hit ? tTri : HIT_MIN_INIT;
    for (int i = 0; i < F.sphereNum; i++) {                                     // frag:570-614
        const auto &s = F.sph[i];
        const float dis = sphere_test(s, p.o, p.d);
        if (dis > 0.0f && dis < hitMin) {
            hitMin = dis;
            hit = true;
            const vec3 centre = rtm::mk(s.center[0], s.center[1], s.center[2]);
            const vec3 n0 = rtm::sub(rtm::along(p.o, hitMin, p.d), centre);
            const bool internal = rtm::dot(n0, p.d) > 0.0f;
            const float temp = rand01(p.seed);
            if (temp < s.specularRate) {
                hitMin -= 0.00001f; material = 1; albedo = rtm::mk(s.albedo[0], s.albedo[1], s.albedo[2]);
            } else if (temp < s.refractionRate) {
                hitMin -= 0.00001f; material = 2; albedo = rtm::mk(1.0f, 1.0f, 1.0f);
            } else {
                hitMin += 0.00001f; material = 3; albedo = rtm::mk(1.0f, 1.0f, 1.0f);
                refraction = internal ? 1.0f / 1.3f : 1.3f;
            }
            pos = rtm::along(p.o, hitMin, p.d);
            nrm = rtm::sub(pos, centre);
            recFrame = false;
        }
    }
    if (!hit) return false;         // a miss can never reach a light afterwards (frag:655-672)
    p.o = pos;
    p.thr = rtm::mul(p.thr, albedo);                                            // frag:635
    if (material == 0) {                                                        // frag:642-645
        sample = rtm::scale(p.thr, 4.0f * 3.1415926f);                          // frag:673
        return false;
    }
    if (material == 1) {                                                        // frag:637-641
        float u = sobU, v = sobV;                                               // sobol(2b), sobol(2b+1) of this frame
        cranley_patterson(pixX, pixY, u, v);
        // SampleCosineHemisphere, frag:504-515; PI of frag:4 is 3.14159274f in fp32
        const float r = rtm::sqrt_exact(u);
        const float theta = (v * 2.0f) * 3.14159274101257324f;
        float sn, cs;
        rtm::sincos_2pi_range(theta, sn, cs);
        const float x = r * cs, y = r * sn;
        const float z = rtm::sqrt_exact(rtm::madd(-y, y, rtm::madd(-x, x, 1.0f)));
        // toNormalHemisphere, frag:494-500.  For a triangle the frame depends on the triangle
        // only and comes from its record (tangent_frame below, same arithmetic; negating the
        // normal negates the tangent and leaves the bitangent, exactly).
        if (!recFrame) tangent_frame(nrm, tangent, bitangent);
        const vec3 L = rtm::mk(rtm::madd(z, nrm.x, rtm::madd(y, bitangent.x, x * tangent.x)),
                               rtm::madd(z, nrm.y, rtm::madd(y, bitangent.y, x * tangent.y)),
                               rtm::madd(z, nrm.z, rtm::madd(y, bitangent.z, x * tangent.z)));
        p.d = rtm::normalize(L);
        p.thr = rtm::scale(p.thr, 0.5f);
    } else if (material == 2) {                                                 // frag:542-545
        const float k = 2.0f * rtm::dot(nrm, p.d);
        p.d = rtm::mk(rtm::madd(-k, nrm.x, p.d.x), rtm::madd(-k, nrm.y, p.d.y), rtm::madd(-k, nrm.z, p.d.z));
    } else {                                                                    // frag:548-556
        const vec3 n2 = rtm::scale(nrm, -rtm::dot(p.d, nrm));
        vec3 t = rtm::add(p.d, n2);
        t = rtm::mk(t.x / refraction, t.y / refraction, t.z / refraction);
        p.d = rtm::normalize(rtm::sub(t, n2));
    }
    const int next = bounce + 1;
    if (next >= F.maxBounces) return false;                                     // loop bound, frag:624
    if (next > 1) {                                                             // frag:625-630
        if (rand01(p.seed) > 0.8f) return false;
    }
    return true;
}

// frag:677-689: seed and camera ray of pixel (px,py)
template <typename FrameT>
__device__ __forceinline__ void generate_path(const FrameT &F, float randOrigin, uint32_t px, uint32_t py,
                                              Path &p, float &uvx, float &uvy) {
    uvx = rtm::pixel_coord(px, (float)F.width);
    uvy = rtm::pixel_coord(py, (float)F.height);
    p.seed = (uint32_t)((randOrigin * 6.95857f) * (uvx * uvy));
    p.o = rtm::mk(F.cam.camPos[0], F.cam.camPos[1], F.cam.camPos[2]);
    const float sx = (uvx * 2.0f) * F.cam.halfW;
    const float sy = (uvy * 2.0f) * F.cam.halfH;
    const vec3 dir = rtm::mk(rtm::madd(sy, F.cam.up[0], rtm::madd(sx, F.cam.right[0], F.cam.leftbottom[0])),
                             rtm::madd(sy, F.cam.up[1], rtm::madd(sx, F.cam.right[1], F.cam.leftbottom[1])),
                             rtm::madd(sy, F.cam.up[2], rtm::madd(sx, F.cam.right[2], F.cam.leftbottom[2])));
    p.d = rtm::normalize(dir);
    p.thr = rtm::mk(1.0f, 1.0f, 1.0f);
}

// Pixel of lane `lane` in 8x8 tile `tile` (tiles numbered row-major over the image).
__device__ __forceinline__ bool tile_pixel(int tile, int lane, int width, int height, uint32_t &px, uint32_t &py) {
    const int tilesX = (width + 7) >> 3;
    const int ty = tile / tilesX, tx = tile - ty * tilesX;
    px = (uint32_t)(tx * 8 + (lane & 7));
    py = (uint32_t)(ty * 8 + (lane >> 3));
    return px < (uint32_t)width && py < (uint32_t)height;
}

// What an 8-bit unsigned-normalised colour attachment keeps of a value (RT_Screen.h:61).
__device__ __forceinline__ float store_unorm8(float v) {
    const float c = (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f;      // NaN -> 0
    return floorf(c * 255.0f + 0.5f) / 255.0f;
}

}  // namespace rtd
