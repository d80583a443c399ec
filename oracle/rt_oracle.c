/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp32 CPU restatement of the reference's path-tracing fragment shader
 * (RayTracerFragmentShader.glsl, cited below as frag:LINE) operating on the
 * reference's own wire format: the two R32F "textures" written by
 * ObjectTexture.h:69-182 (33 floats per triangle, 9 floats per BVH node), the
 * 512-entry Sobol table (sobol.h:8-25) and the per-frame uniforms set at
 * main.cpp:144-169.  Only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py may build, load or call this file.  The shipped library never
 * links it.
 *
 * PARITY STATUS: "parity unpinned" for shader output.  The reference has no tests,
 * golden images or recorded outputs, its GLSL cannot be executed in the build
 * container (no GL context), and its C++ cannot be compiled without third-party
 * headers that are absent.  What IS pinned by reference data: the Sobol table
 * (v_matrix.txt == sobol.h V[]) and the sobol()/grayCode() values; see
 * tests/test_oracle_sobol.py.  Everything else rests on this file following the
 * GLSL statement by statement.
 *
 * ARITHMETIC CONTRACT.  GLSL leaves the precision of sin/cos/normalize/sqrt/division
 * and the contraction of a*b+c to the implementation.  This oracle fixes one
 * concrete fp32 evaluation so that results are reproducible to the bit:
 *   - every +,-,*,/ and sqrt is IEEE-754 binary32, round-to-nearest-even,
 *     subnormals kept; build with -ffp-contract=off so the compiler fuses nothing;
 *   - MADD(a,b,c) marks the places where a product feeds an add: with
 *     RT_ORACLE_FUSED=1 (default, what GPU shader compilers do) it is one fused
 *     multiply-add, with RT_ORACLE_FUSED=0 it is a rounded product then an add;
 *   - dot(a,b)   = MADD(a.z,b.z, MADD(a.y,b.y, a.x*b.x));
 *   - cross(a,b) = (MADD(a.y,b.z,-(b.y*a.z)), MADD(a.z,b.x,-(b.z*a.x)), MADD(a.x,b.y,-(b.x*a.y)))
 *     following the GLSL definition x[1]*y[2]-y[1]*x[2], ...;
 *   - normalize(v) = v * (1.0f / sqrtf(dot(v,v)));
 *   - sin/cos(theta): trig_mode 0 = libm sinf/cosf; trig_mode 1 = or_sincos() below,
 *     a fixed Cody-Waite + polynomial evaluation (<= 2 ulp from libm on [0, 2*pi]).
 *     Mode 1 exists so that a second implementation can match bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef RT_ORACLE_FUSED
#define RT_ORACLE_FUSED 1
#endif

#if RT_ORACLE_FUSED
#define MADD(a, b, c) __builtin_fmaf((a), (b), (c))
#else
static inline float or_madd_plain(float a, float b, float c) {
    volatile float p = a * b;
    return p + c;
}
#define MADD(a, b, c) or_madd_plain((a), (b), (c))
#endif

/* ------------------------------------------------------------------ types */

typedef struct { float x, y, z; } v3;

typedef struct {            /* uniform struct Camera, frag:115-125 */
    float camPos[3];
    float front[3];
    float right[3];
    float up[3];
    float halfH;
    float halfW;
    float leftbottom[3];
    int LoopNum;
} OrCamera;

typedef struct {            /* uniform Sphere, frag:135-142; Scene.h:27-44 */
    float center[3];
    float radius;
    float albedo[3];
    float specularRate;
    float refractionRate;
    float roughness;
} OrSphere;

typedef struct {
    const float *texTriangle;   /* R32F texture rows, ObjectTexture.h:82-122 */
    int texTriangleWidth;       /* textureSize(texTriangle,0).x = x*33      */
    const float *texBVH;        /* ObjectTexture.h:140-168                  */
    int texBVHWidth;            /* x*9                                      */
    int triangleNum;
    int bvhNum;
    int width;                  /* uniform int width/height, main.cpp:165-166 */
    int height;
    const uint32_t *V;          /* uniform uint V[512], frag:81 */
    OrSphere sphere[10];        /* frag:143 */
    int sphereNum;              /* frag:144 */
    int maxBounces;             /* 4 in the reference (frag:624) */
    int trigMode;               /* see ARITHMETIC CONTRACT */
    const int32_t *bvhInts;     /* NULL, or 3 ints per node (triangleNum, offsetOrRc, axis) that replace
                                   the float-encoded ones of texBVH.  The reference stores node indices
                                   as floats (ObjectTexture.h:152-154) and recovers them with
                                   int(f + 0.5) (frag:180-182), exact only below 2^23: it cannot express
                                   a tree of more than 4 194 304 triangles.  Callers set this for such
                                   trees only; everything else is read from the textures as usual. */
    int shadingNormals;         /* 0: the reference's behaviour (the face normal of getTriangleNormal).
                                   1: an extension the reference only prepares (the Triangle struct carries
                                   n0 n1 n2, frag:131, and the comment above getTriangleNormal, frag:295,
                                   marks it as to be changed to interpolation): shade with the
                                   interpolated vertex normals, see shading_normal() below. */
    const uint8_t *albedoTex;   /* NULL: the reference's behaviour.  Otherwise a second extension of the same
                                   kind: the Triangle struct carries uv0 uv1 uv2 (frag:132) and the loader
                                   reads the diffuse maps of a material (Model.h:167-169), but the shader
                                   never samples them.  albedoTexW x albedoTexH texels of 3 bytes (GL_RGB,
                                   GL_UNSIGNED_BYTE, row 0 at v = 0, as glTexImage2D takes them,
                                   Model.h:241); the albedo of a diffuse draw becomes kd * texel, see
                                   albedo_texel() below. */
    int albedoTexW, albedoTexH;
} OrScene;

typedef struct {
    uint64_t rays;          /* closest-hit queries excluding re-traces of a ray that already missed */
    uint64_t raysRef;       /* hitWorld() calls exactly as the reference makes them (frag:631) */
    uint64_t nodeTests;     /* IntersectBound() calls (frag:389), re-traces excluded */
    uint64_t triTests;      /* hitTriangle() calls (frag:395), re-traces excluded */
    uint64_t triHits;       /* hitBVH() returning true (frag:457), re-traces excluded */
} OrStats;

/* ------------------------------------------------------------ vector math */

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(v3 a, v3 b) { return MADD(a.z, b.z, MADD(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(MADD(a.y, b.z, -(b.y * a.z)),
              MADD(a.z, b.x, -(b.z * a.x)),
              MADD(a.x, b.y, -(b.x * a.y)));
}
static inline v3 normalize3(v3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return vscale(a, inv);
}
/* o + t*d, each component one MADD */
static inline v3 along(v3 o, float t, v3 d) {
    return V3(MADD(t, d.x, o.x), MADD(t, d.y, o.y), MADD(t, d.z, o.z));
}

/* Fixed-sequence sin/cos for theta in [0, 2*pi] (trig_mode 1). */
void or_sincos(float x, float *s_out, float *c_out) {
    const float TWO_OVER_PI = 0.63661977236758134f;
    const float PIO2_HI = 1.57079625129699707031f;      /* 0x3FC90FDA */
    const float PIO2_LO = 7.54978941586159635335e-08f;  /* pi/2 - PIO2_HI */
    int k = (int)(MADD(x, TWO_OVER_PI, 0.5f));
    float fk = (float)k;
    float r = MADD(-fk, PIO2_HI, x);
    r = MADD(-fk, PIO2_LO, r);
    float r2 = r * r;
    /* sin(r) ~ r + r^3 * (S1 + r2*(S2 + r2*S3)) */
    float ps = MADD(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = MADD(r2, ps, -1.6666654611e-1f);
    float s = MADD(r * r2, ps, r);
    /* cos(r) ~ 1 - r2/2 + r2^2 * (C1 + r2*(C2 + r2*C3)) */
    float pc = MADD(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = MADD(r2, pc, 4.166664568298827e-2f);
    float c = MADD(r2 * r2, pc, MADD(r2, -0.5f, 1.0f));
    switch (k & 3) {
    case 0: *s_out = s;  *c_out = c;  break;
    case 1: *s_out = c;  *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

/* --------------------------------------------------------- per-pixel state */

typedef struct { v3 origin, direction; float hitMin; } Ray;      /* frag:236-240 */

typedef struct {                                                 /* frag:242-250 */
    v3 Normal, Pos, albedo;
    int materialIndex;
    float roughness, refraction, rayHitMin;
} HitRecord;

typedef struct { v3 v0, v1, v2, ka, kd, ks, n0, n1, n2; float uv[3][2]; } Tri;   /* used fields of frag:128-133; n0..n2 only with shadingNormals, uv only with albedoTex */

typedef struct {
    const OrScene *sc;
    const OrCamera *cam;
    float uvx, uvy;         /* TexCoords */
    uint32_t wseed;         /* frag:35 */
    HitRecord rec;          /* frag:251 (global) */
    OrStats *st;
    int hitTriIndex;        /* instrumentation: index of the triangle hitBVH kept */
    int retrace;            /* instrumentation: this hitWorld call repeats a ray that already missed */
} Px;

/* frag:36-47 */
static float or_rand(Px *p) {
    uint32_t seed = p->wseed;
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    p->wseed = seed ^ (seed >> 15);
    return (float)p->wseed * (1.0f / 4294967296.0f);
}

/* frag:50-57 */
static uint32_t wang_hash(uint32_t *seed) {
    uint32_t s = *seed;
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    *seed = s;
    return s;
}

/* frag:59-77 */
static void cranley_patterson(const Px *p, float *px, float *py) {
    float w = (float)p->sc->width, h = (float)p->sc->height;
    uint32_t X = (uint32_t)(MADD(MADD(p->uvx, 2.0f, -1.0f), 0.5f, 0.5f) * w);
    uint32_t Y = (uint32_t)(MADD(MADD(p->uvy, 2.0f, -1.0f), 0.5f, 0.5f) * h);
    uint32_t pseed = (X * 1973u + Y * 9277u + (uint32_t)(114514 / 1919) * 26699u) | 1u;
    float u = (float)wang_hash(&pseed) / 4294967296.0f;
    float v = (float)wang_hash(&pseed) / 4294967296.0f;
    float x = *px + u;
    if (x > 1.0f) x -= 1.0f;
    if (x < 0.0f) x += 1.0f;
    float y = *py + v;
    if (y > 1.0f) y -= 1.0f;
    if (y < 0.0f) y += 1.0f;
    *px = x; *py = y;
}

/* frag:84-97; host twin sobol.h:37-50 */
uint32_t or_gray_code(uint32_t i) { return i ^ (i >> 1); }
float or_sobol(const uint32_t *V, uint32_t d, uint32_t i) {
    uint32_t result = 0u, offset = d * 32u;
    for (uint32_t j = 0u; i != 0u; i >>= 1, j++)
        if ((i & 1u) != 0u) result ^= V[j + offset];
    return (float)result * (1.0f / (float)0xFFFFFFFFu);
}

/* frag:147-159: flat index -> (x,y) -> texel */
static inline float At(const float *tex, int texW, int index) {
    int y = index / texW;
    int x = index - y * texW;
    return tex[(size_t)y * (size_t)texW + (size_t)x];
}

typedef struct { v3 maxB, minB; int triangleNum, offsetOrRc, axis; } Node;

/* frag:169-185 */
static Node get_node(const OrScene *sc, int offset) {
    Node n;
    int o = offset * 9;
    n.maxB = V3(At(sc->texBVH, sc->texBVHWidth, o + 0), At(sc->texBVH, sc->texBVHWidth, o + 1),
                At(sc->texBVH, sc->texBVHWidth, o + 2));
    n.minB = V3(At(sc->texBVH, sc->texBVHWidth, o + 3), At(sc->texBVH, sc->texBVHWidth, o + 4),
                At(sc->texBVH, sc->texBVHWidth, o + 5));
    n.triangleNum = (int)(At(sc->texBVH, sc->texBVHWidth, o + 6) + 0.5f);
    n.offsetOrRc = (int)(At(sc->texBVH, sc->texBVHWidth, o + 7) + 0.5f);
    n.axis = (int)(At(sc->texBVH, sc->texBVHWidth, o + 8) + 0.5f);
    if (sc->bvhInts) {          /* tree beyond the float encoding's range, see OrScene */
        n.triangleNum = sc->bvhInts[3 * (size_t)offset];
        n.offsetOrRc = sc->bvhInts[3 * (size_t)offset + 1];
        n.axis = sc->bvhInts[3 * (size_t)offset + 2];
    }
    return n;
}

/* frag:190-233; normals (3-5,11-13,19-21) and uvs (6-7,14-15,22-23) are fetched by the
 * shader but never used, so they are not read here. */
static Tri get_triangle(const OrScene *sc, int offset) {
    Tri t;
    const float *T = sc->texTriangle; int W = sc->texTriangleWidth; int o = offset * 33;
    t.v0 = V3(At(T, W, o + 0), At(T, W, o + 1), At(T, W, o + 2));
    t.v1 = V3(At(T, W, o + 8), At(T, W, o + 9), At(T, W, o + 10));
    t.v2 = V3(At(T, W, o + 16), At(T, W, o + 17), At(T, W, o + 18));
    t.ka = V3(At(T, W, o + 24), At(T, W, o + 25), At(T, W, o + 26));
    t.kd = V3(At(T, W, o + 27), At(T, W, o + 28), At(T, W, o + 29));
    t.ks = V3(At(T, W, o + 30), At(T, W, o + 31), At(T, W, o + 32));
    t.n0 = t.n1 = t.n2 = V3(0.0f, 0.0f, 0.0f);
    if (sc->shadingNormals) {
        t.n0 = V3(At(T, W, o + 3), At(T, W, o + 4), At(T, W, o + 5));
        t.n1 = V3(At(T, W, o + 11), At(T, W, o + 12), At(T, W, o + 13));
        t.n2 = V3(At(T, W, o + 19), At(T, W, o + 20), At(T, W, o + 21));
    }
    memset(t.uv, 0, sizeof t.uv);
    if (sc->albedoTex)
        for (int k = 0; k < 3; k++) { t.uv[k][0] = At(T, W, o + 8 * k + 6); t.uv[k][1] = At(T, W, o + 8 * k + 7); }
    return t;
}

/* frag:255-278 */
static float hit_triangle(const Tri *tri, const Ray *r) {
    v3 A = vsub(tri->v1, tri->v0);
    v3 B = vsub(tri->v2, tri->v0);
    v3 N = normalize3(cross3(A, B));
    if (dot3(N, r->direction) == 0.0f) return -1.0f;
    float D = -dot3(N, tri->v0);
    float t = -(dot3(N, r->origin) + D) / dot3(N, r->direction);
    if (t < 0.0f) return -1.0f;
    v3 pHit = along(r->origin, t, r->direction);
    v3 edge0 = vsub(tri->v1, tri->v0);
    v3 C0 = vsub(pHit, tri->v0);
    if (dot3(N, cross3(edge0, C0)) < 0.0f) return -1.0f;
    v3 edge1 = vsub(tri->v2, tri->v1);
    v3 C1 = vsub(pHit, tri->v1);
    if (dot3(N, cross3(edge1, C1)) < 0.0f) return -1.0f;
    v3 edge2 = vsub(tri->v0, tri->v2);
    v3 C2 = vsub(pHit, tri->v2);
    if (dot3(N, cross3(edge2, C2)) < 0.0f) return -1.0f;
    return t - 0.00001f;
}

/* frag:281-293 */
static float hit_sphere(const OrSphere *s, const Ray *r) {
    v3 oc = vsub(r->origin, V3(s->center[0], s->center[1], s->center[2]));
    float a = dot3(r->direction, r->direction);
    float b = 2.0f * dot3(oc, r->direction);
    float c = MADD(-s->radius, s->radius, dot3(oc, oc));
    float discriminant = MADD(b, b, -((4.0f * a) * c));
    if (discriminant > 0.0f) {
        float dis = (-b - sqrtf(discriminant)) / (2.0f * a);
        if (dis > 0.0f) return dis;
        else return -1.0f;
    }
    else return -1.0f;
}

/* frag:296-300 */
static v3 triangle_normal(const Tri *tri, const Ray *r) {
    v3 res = normalize3(cross3(vsub(tri->v2, tri->v0), vsub(tri->v1, tri->v0)));
    if (dot3(res, r->direction) > 0.0f) res = vneg(res);
    return res;
}

/* shadingNormals = 1 (not in the reference, see OrScene).  The weights are the three edge
 * functions hitTriangle evaluates (frag:265-275) taken at the hit position P: w0 belongs to v0
 * (edge v1->v2), w1 to v1 (edge v2->v0), w2 to v2 (edge v0->v1); their common factor drops out
 * in the normalisation.  The result is turned to the side of the (already flipped) face normal.
 * A sum that is zero, infinite or NaN (a model without normals stores zeros) keeps the face normal. */
static v3 shading_normal(const Tri *tri, v3 P, v3 faceNormal) {
    v3 N = normalize3(cross3(vsub(tri->v1, tri->v0), vsub(tri->v2, tri->v0)));
    float w2 = dot3(N, cross3(vsub(tri->v1, tri->v0), vsub(P, tri->v0)));
    float w0 = dot3(N, cross3(vsub(tri->v2, tri->v1), vsub(P, tri->v1)));
    float w1 = dot3(N, cross3(vsub(tri->v0, tri->v2), vsub(P, tri->v2)));
    v3 n = V3(MADD(w2, tri->n2.x, MADD(w1, tri->n1.x, w0 * tri->n0.x)),
              MADD(w2, tri->n2.y, MADD(w1, tri->n1.y, w0 * tri->n0.y)),
              MADD(w2, tri->n2.z, MADD(w1, tri->n1.z, w0 * tri->n0.z)));
    float l2 = dot3(n, n);
    if (!(l2 > 0.0f) || !(l2 < INFINITY)) return faceNormal;
    n = vscale(n, 1.0f / sqrtf(l2));
    if (dot3(n, faceNormal) < 0.0f) n = vneg(n);
    return n;
}

/* albedoTex != NULL (not in the reference, see OrScene).  The texture coordinate is the
 * barycentric mean of uv0 uv1 uv2 with the weights of shading_normal() (the edge functions of
 * hitTriangle at the hit position), divided by their sum.  Wrapping is GL_REPEAT
 * (Model.h:244-245): the fractional part c - floor(c); the filter is GL_NEAREST like the
 * tracer's own data textures (ObjectTexture.h:127-128; the trilinear filter of Model.h:246-247
 * belongs to a rasteriser with screen-space derivatives): texel floor(frac * size), clamped to
 * size - 1 (the fraction of a tiny negative coordinate rounds to 1).  A coordinate that is not
 * finite (degenerate weights) reads texel (0,0).  A byte b is the colour b / 255. */
static v3 albedo_texel(const OrScene *sc, const Tri *tri, v3 P) {
    v3 N = normalize3(cross3(vsub(tri->v1, tri->v0), vsub(tri->v2, tri->v0)));
    float w2 = dot3(N, cross3(vsub(tri->v1, tri->v0), vsub(P, tri->v0)));
    float w0 = dot3(N, cross3(vsub(tri->v2, tri->v1), vsub(P, tri->v1)));
    float w1 = dot3(N, cross3(vsub(tri->v0, tri->v2), vsub(P, tri->v2)));
    float sum = (w0 + w1) + w2;
    int xy[2];
    for (int k = 0; k < 2; k++) {
        float c = MADD(w2, tri->uv[2][k], MADD(w1, tri->uv[1][k], w0 * tri->uv[0][k])) / sum;
        int size = k ? sc->albedoTexH : sc->albedoTexW;
        if (!(fabsf(c) < INFINITY)) c = 0.0f;
        float frac = c - floorf(c);             /* GL_REPEAT: the fractional part ... */
        if (frac >= 1.0f) frac = 0.0f;          /* ... which rounds to 1 for a tiny negative c: that wraps to texel 0 */
        int i = (int)(frac * (float)size);
        xy[k] = i > size - 1 ? size - 1 : i;
    }
    const uint8_t *px = sc->albedoTex + ((size_t)xy[1] * (size_t)sc->albedoTexW + (size_t)xy[0]) * 3;
    return V3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

/* frag:306-331.  getBoundp(b,i) = i==0 ? pMin : pMax. */
static int intersect_bound(v3 pMin, v3 pMax, const Ray *ray, v3 invDir, const int neg[3]) {
    float tMin = ((neg[0] ? pMax.x : pMin.x) - ray->origin.x) * invDir.x;
    float tMax = ((neg[0] ? pMin.x : pMax.x) - ray->origin.x) * invDir.x;
    float tyMin = ((neg[1] ? pMax.y : pMin.y) - ray->origin.y) * invDir.y;
    float tyMax = ((neg[1] ? pMin.y : pMax.y) - ray->origin.y) * invDir.y;
    if (tMin > tyMax || tyMin > tMax) return 0;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((neg[2] ? pMax.z : pMin.z) - ray->origin.z) * invDir.z;
    float tzMax = ((neg[2] ? pMin.z : pMax.z) - ray->origin.z) * invDir.z;
    if (tMin > tzMax || tzMin > tMax) return 0;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return tMax > 0.0f;
}

/* frag:373-490.  `ray` is a by-value copy exactly as in GLSL. */
static int hit_bvh(Px *p, Ray ray) {
    const OrScene *sc = p->sc;
    int hit = 0;
    v3 invDir = V3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    int neg[3] = { invDir.x < 0.0f, invDir.y < 0.0f, invDir.z < 0.0f };
    int toVisitOffset = 0, currentNodeIndex = 0;
    int nodesToVisit[256];
    Tri tri; memset(&tri, 0, sizeof tri);
    if (sc->bvhNum <= 0) return 0;   /* the reference would read an empty texture; no geometry = no hit */
    while (1) {
        Node node = get_node(sc, currentNodeIndex);
        if (p->st && !p->retrace) p->st->nodeTests++;
        if (intersect_bound(node.minB, node.maxB, &ray, invDir, neg)) {
            if (node.triangleNum == 1) {
                for (int i = 0; i < node.triangleNum; ++i) {
                    int offset = node.offsetOrRc + i;
                    Tri tri_t = get_triangle(sc, offset);
                    if (p->st && !p->retrace) p->st->triTests++;
                    float dis_t = hit_triangle(&tri_t, &ray);
                    if (dis_t > 0.0f && dis_t < ray.hitMin) {
                        ray.hitMin = dis_t;
                        tri = tri_t;
                        hit = 1;
                        p->hitTriIndex = offset;
                    }
                }
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            } else {
                int ax = node.axis;
                if (neg[ax]) {
                    nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                    currentNodeIndex = node.offsetOrRc;
                } else {
                    nodesToVisit[toVisitOffset++] = node.offsetOrRc;
                    currentNodeIndex = currentNodeIndex + 1;
                }
            }
        } else {
            if (toVisitOffset == 0) break;
            currentNodeIndex = nodesToVisit[--toVisitOffset];
        }
    }
    if (hit) {
        if (p->st && !p->retrace) p->st->triHits++;
        HitRecord *rec = &p->rec;
        rec->Pos = along(ray.origin, ray.hitMin, ray.direction);
        rec->Normal = triangle_normal(&tri, &ray);
        if (sc->shadingNormals) rec->Normal = shading_normal(&tri, rec->Pos, rec->Normal);
        float temp = or_rand(p);
        if (sqrtf(dot3(tri.ka, tri.ka)) > 0.0f) {           /* length(tri.ka) > 0.0 */
            rec->materialIndex = 0;
            rec->albedo = tri.ka;
        } else if (temp < tri.ks.x) {
            rec->materialIndex = 1;
            rec->albedo = tri.kd;
            if (sc->albedoTex) rec->albedo = vmul(tri.kd, albedo_texel(sc, &tri, rec->Pos));
        } else if (temp < tri.ks.y) {
            rec->materialIndex = 2;
            rec->albedo = V3(1.0f, 1.0f, 1.0f);
            rec->roughness = tri.ks.z;
        } else {
            rec->materialIndex = 3;
            rec->albedo = V3(1.0f, 1.0f, 1.0f);
            rec->refraction = 1.5f;
            rec->roughness = tri.ks.z;
        }
        rec->rayHitMin = ray.hitMin;
    }
    return hit;
}

/* frag:494-500 */
static v3 to_normal_hemisphere(v3 v, v3 N) {
    v3 helper = V3(1.0f, 0.0f, 0.0f);
    if (fabsf(N.x) > 0.999f) helper = V3(0.0f, 0.0f, 1.0f);
    v3 tangent = normalize3(cross3(N, helper));
    v3 bitangent = normalize3(cross3(N, tangent));
    return V3(MADD(v.z, N.x, MADD(v.y, bitangent.x, v.x * tangent.x)),
              MADD(v.z, N.y, MADD(v.y, bitangent.y, v.x * tangent.y)),
              MADD(v.z, N.z, MADD(v.y, bitangent.z, v.x * tangent.z)));
}

/* frag:504-515; PI at frag:4 rounds to 3.14159274f */
static v3 sample_cosine_hemisphere(const Px *p, float xi_1, float xi_2, v3 N) {
    const float PI = 3.14159274101257324f;
    float r = sqrtf(xi_1);
    float theta = (xi_2 * 2.0f) * PI;
    float cs, sn;
    if (p->sc->trigMode == 1) or_sincos(theta, &sn, &cs);
    else { cs = cosf(theta); sn = sinf(theta); }
    float x = r * cs;
    float y = r * sn;
    float z = sqrtf(MADD(-y, y, MADD(-x, x, 1.0f)));
    return normalize3(to_normal_hemisphere(V3(x, y, z), N));
}

/* frag:100-104 + frag:536-539 */
static v3 diffuse_reflection(const Px *p, v3 Normal, int hitTimes) {
    uint32_t i = (uint32_t)p->cam->LoopNum, b = (uint32_t)hitTimes;
    float u = or_sobol(p->sc->V, b * 2u, or_gray_code(i));
    float v = or_sobol(p->sc->V, b * 2u + 1u, or_gray_code(i));
    cranley_patterson(p, &u, &v);
    return sample_cosine_hemisphere(p, u, v, Normal);
}

/* frag:542-545 */
static v3 specular_reflection(v3 Normal, v3 inDirection) {
    float k = 2.0f * dot3(Normal, inDirection);
    return V3(MADD(-k, Normal.x, inDirection.x), MADD(-k, Normal.y, inDirection.y),
              MADD(-k, Normal.z, inDirection.z));
}

/* frag:548-556 */
static v3 specular_refraction(v3 Normal, v3 inDirection, float refraction) {
    Normal = vscale(Normal, -dot3(inDirection, Normal));
    v3 temp = vadd(inDirection, Normal);
    temp = V3(temp.x / refraction, temp.y / refraction, temp.z / refraction);
    temp = vsub(temp, Normal);
    return normalize3(temp);
}

/* frag:558-618.  `r` by value; rec is the global. */
static int hit_world(Px *p, Ray r) {
    const OrScene *sc = p->sc;
    HitRecord *rec = &p->rec;
    int hit = 0;
    if (hit_bvh(p, r)) {
        if (rec->rayHitMin > 0.0f && rec->rayHitMin < r.hitMin) {
            r.hitMin = rec->rayHitMin;
            hit = 1;
        }
    }
    for (int i = 0; i < sc->sphereNum; i++) {
        const OrSphere *s = &sc->sphere[i];
        v3 center = V3(s->center[0], s->center[1], s->center[2]);
        float dis_t = hit_sphere(s, &r);
        if (dis_t > 0.0f && dis_t < r.hitMin) {
            r.hitMin = dis_t;
            hit = 1;
            rec->Pos = along(r.origin, r.hitMin, r.direction);
            rec->Normal = vsub(rec->Pos, center);
            int internal = 0;
            if (dot3(rec->Normal, r.direction) > 0.0f) {
                rec->Normal = vneg(rec->Normal);
                internal = 1;
            }
            float temp = or_rand(p);
            if (temp < s->specularRate) {
                r.hitMin -= 0.00001f;
                rec->Pos = along(r.origin, r.hitMin, r.direction);
                rec->Normal = vsub(rec->Pos, center);
                rec->materialIndex = 1;
                rec->albedo = V3(s->albedo[0], s->albedo[1], s->albedo[2]);
            } else if (temp < s->refractionRate) {
                r.hitMin -= 0.00001f;
                rec->Pos = along(r.origin, r.hitMin, r.direction);
                rec->Normal = vsub(rec->Pos, center);
                rec->materialIndex = 2;
                rec->albedo = V3(1.0f, 1.0f, 1.0f);
                rec->roughness = s->roughness;
            } else {
                r.hitMin += 0.00001f;
                rec->Pos = along(r.origin, r.hitMin, r.direction);
                rec->Normal = vsub(rec->Pos, center);
                rec->materialIndex = 3;
                rec->albedo = V3(1.0f, 1.0f, 1.0f);
                rec->refraction = internal ? 1.0f / 1.3f : 1.3f;
                rec->roughness = s->roughness;
            }
            rec->rayHitMin = r.hitMin;
        }
    }
    return hit;
}

/* frag:621-675 */
static v3 shading(Px *p, Ray r) {
    v3 color = V3(1.0f, 1.0f, 1.0f);
    int hitLight = 0;
    int missed = 0;
    for (int i = 0; i < p->sc->maxBounces; i++) {
        if (i > 1) {
            if (or_rand(p) > 0.8f) {
                color = V3(0.0f, 0.0f, 0.0f);
                break;
            }
        }
        if (p->st) { p->st->raysRef++; if (!missed) p->st->rays++; }
        p->retrace = missed;
        if (hit_world(p, r)) {
            const HitRecord *rec = &p->rec;
            r.origin = rec->Pos;
            r.hitMin = 100000.0f;
            color = vmul(color, rec->albedo);
            if (rec->materialIndex == 1) {
                r.direction = diffuse_reflection(p, rec->Normal, i);
                color = vscale(color, 0.5f);
            } else if (rec->materialIndex == 0) {
                hitLight = 1;
                break;
            } else if (rec->materialIndex == 2) {
                r.direction = specular_reflection(rec->Normal, r.direction);
            } else if (rec->materialIndex == 3) {
                r.direction = specular_refraction(rec->Normal, r.direction, rec->refraction);
            }
        } else {
            color = vmul(color, V3(0.5f, 0.5f, 0.5f));
            missed = 1;     /* the same ray is traced again next iteration (frag:655-670) */
        }
    }
    if (hitLight == 0) color = V3(0.0f, 0.0f, 0.0f);
    color = vscale(color, 4.0f * 3.1415926f);       /* 2.0 * 2.0 * 3.1415926, frag:673 */
    return color;
}

/* frag:677-692: returns curColor before blending.  TexCoords = ((px+0.5)/W, (py+0.5)/H). */
void or_shade_pixel(const OrScene *sc, const OrCamera *cam, float randOrigin, int px, int py,
                    float out_rgb[3], OrStats *st) {
    Px p; memset(&p, 0, sizeof p);
    p.sc = sc; p.cam = cam; p.st = st;
    p.uvx = ((float)px + 0.5f) / (float)sc->width;
    p.uvy = ((float)py + 0.5f) / (float)sc->height;
    p.wseed = (uint32_t)((randOrigin * 6.95857f) * (p.uvx * p.uvy));
    Ray cameraRay;
    cameraRay.origin = V3(cam->camPos[0], cam->camPos[1], cam->camPos[2]);
    float sx = (p.uvx * 2.0f) * cam->halfW;
    float sy = (p.uvy * 2.0f) * cam->halfH;
    v3 d = V3(MADD(sy, cam->up[0], MADD(sx, cam->right[0], cam->leftbottom[0])),
              MADD(sy, cam->up[1], MADD(sx, cam->right[1], cam->leftbottom[1])),
              MADD(sy, cam->up[2], MADD(sx, cam->right[2], cam->leftbottom[2])));
    cameraRay.direction = normalize3(d);
    cameraRay.hitMin = 100000.0f;
    v3 c = shading(&p, cameraRay);
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}

/* frag:683,697-698: one frame over the whole image with an fp32 history buffer.
 * hist/out are W*H*3 floats, row 0 = uv.y nearest 0.  cur (optional) receives the
 * pre-blend sample.  Rows y0 <= py < y1 only (lets the caller time or test a band).
 * Parallel over rows with OpenMP when compiled with -fopenmp. */
void or_render_frame(const OrScene *sc, const OrCamera *cam, float randOrigin,
                     const float *hist, float *out, float *cur, int y0, int y1, OrStats *st) {
    int W = sc->width;
    float N = (float)cam->LoopNum;
    float wa = 1.0f / N;
    float wb = (float)(cam->LoopNum - 1) / N;
    OrStats total; memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        OrStats loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 4)
        for (int py = y0; py < y1; py++) {
            for (int px = 0; px < W; px++) {
                float c[3];
                or_shade_pixel(sc, cam, randOrigin, px, py, c, st ? &loc : NULL);
                size_t o = ((size_t)py * (size_t)W + (size_t)px) * 3;
                for (int k = 0; k < 3; k++) {
                    if (cur) cur[o + k] = c[k];
                    if (out) {
                        float h = hist ? hist[o + k] : 0.0f;
                        out[o + k] = MADD(wa, c[k], wb * h);
                    }
                }
            }
        }
#pragma omp critical
        {
            total.rays += loc.rays; total.raysRef += loc.raysRef; total.nodeTests += loc.nodeTests;
            total.triTests += loc.triTests; total.triHits += loc.triHits;
        }
    }
    if (st) {
        st->rays += total.rays; st->raysRef += total.raysRef; st->nodeTests += total.nodeTests;
        st->triTests += total.triTests; st->triHits += total.triHits;
    }
}

/* Frames firstLoopNum .. firstLoopNum+count-1 (randOrigin[f] for frame f) over the pixels of
 * rows y0 <= py < y1: what `count` calls of or_render_frame, each fed the previous one's
 * output as its history, leave in those rows - the running mean of frag:697 continued from
 * the contents of `mean` - and the plain fp32 sum of the samples added to `sum` in frame
 * order.  mean/sum are whole W*H*3 images of which only the band is read and written.
 * A pixel's frames do not depend on any other pixel, so the loop over pixels is the parallel
 * one and every thread runs all frames of its pixels: a 16-row band of a thousand frames
 * keeps all cores busy, which a loop over 16 rows per frame does not. */
void or_render_band(const OrScene *sc, const OrCamera *cam, int firstLoopNum, int count,
                    const float *randOrigin, float *mean, float *sum, int y0, int y1) {
    const int W = sc->width;
    const long long npix = (long long)(y1 - y0) * W;
#pragma omp parallel for schedule(dynamic, 16)
    for (long long i = 0; i < npix; i++) {
        const int py = y0 + (int)(i / W), px = (int)(i % W);
        const size_t o = ((size_t)py * (size_t)W + (size_t)px) * 3;
        OrCamera c = *cam;
        float m[3] = {mean[o], mean[o + 1], mean[o + 2]};
        float s[3] = {sum[o], sum[o + 1], sum[o + 2]};
        for (int f = 0; f < count; f++) {
            c.LoopNum = firstLoopNum + f;
            const float N = (float)c.LoopNum;
            const float wa = 1.0f / N, wb = (float)(c.LoopNum - 1) / N;
            float rgb[3];
            or_shade_pixel(sc, &c, randOrigin[f], px, py, rgb, NULL);
            for (int k = 0; k < 3; k++) {
                m[k] = MADD(wa, rgb[k], wb * m[k]);
                s[k] += rgb[k];
            }
        }
        for (int k = 0; k < 3; k++) { mean[o + k] = m[k]; sum[o + k] = s[k]; }
    }
}

/* hitBVH alone (frag:373-456) for traversal parity tests: returns 1 on a hit and gives
 * ray.hitMin (t - 1e-5 of the kept triangle) and that triangle's index. */
int or_hit_bvh(const OrScene *sc, const float origin[3], const float dir[3], float *t_out, int *tri_out) {
    Px p; memset(&p, 0, sizeof p);
    OrCamera cam; memset(&cam, 0, sizeof cam);
    p.sc = sc; p.cam = &cam; p.hitTriIndex = -1;
    Ray r; r.origin = V3(origin[0], origin[1], origin[2]);
    r.direction = V3(dir[0], dir[1], dir[2]); r.hitMin = 100000.0f;
    int h = hit_bvh(&p, r);
    if (h) { *t_out = p.rec.rayHitMin; *tri_out = p.hitTriIndex; }
    return h;
}

/* The same for n rays (origins/dirs: n*3 floats); t_out[i] = -1 and tri_out[i] = -1 on a miss.
 * Parallel over rays with OpenMP. */
void or_hit_bvh_many(const OrScene *sc, const float *origins, const float *dirs, long long n,
                     float *t_out, int *tri_out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < n; i++) {
        float t = -1.0f; int tri = -1;
        if (!or_hit_bvh(sc, origins + 3 * i, dirs + 3 * i, &t, &tri)) { t = -1.0f; tri = -1; }
        t_out[i] = t; tri_out[i] = tri;
    }
}

int or_is_fused(void) { return RT_ORACLE_FUSED; }
