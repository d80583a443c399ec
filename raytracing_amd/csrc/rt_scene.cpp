// Host-side scene assembly: the reference's Scene (Scene.h:46-90), its model loader
// restricted to what reaches the tracer (Model.h:54-182: positions, normals, uvs, per-mesh
// Ka/Kd/Ks), the flattening into the triangle list that enters the BVH builder
// (BVH.h:120-135), the bundled demo scene of main.cpp:103-110, and synthetic benchmark
// geometry.
#include <algorithm>
#include <array>
#include <new>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "rt_internal.h"

namespace rt {

namespace {

struct Material { float ka[3] = {0, 0, 0}, kd[3] = {0.6f, 0.6f, 0.6f}, ks[3] = {0, 0, 0}; };

std::string trim(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) b++;
    while (e > b && std::isspace((unsigned char)s[e - 1])) e--;
    return s.substr(b, e - b);
}

std::string dir_of(const std::string &path) {
    size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

std::string lower(std::string s) {
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

// Opens dir/name; if that fails, retries with the file name lower-cased and upper/lower
// variants of its first letter (room.obj says "mtllib Room.mtl" while the file on disk is
// room.mtl, which only resolves on case-insensitive file systems).
bool open_relaxed(const std::string &dir, const std::string &name, std::ifstream &f) {
    const std::string cands[3] = {name, lower(name),
                                  name.empty() ? name : std::string(1, (char)std::toupper((unsigned char)name[0])) + lower(name).substr(1)};
    for (const auto &c : cands) {
        f.open(dir + "/" + c);
        if (f.is_open()) return true;
        f.clear();
    }
    return false;
}

void load_mtl(const std::string &dir, const std::string &name, std::map<std::string, Material> &mats) {
    std::ifstream f;
    if (!open_relaxed(dir, name, f)) return;    // missing library: default materials, as Assimp does
    std::string line, cur;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key; ss >> key;
        if (key == "newmtl") { ss >> cur; mats[cur] = Material(); }
        else if (!cur.empty() && (key == "Ka" || key == "Kd" || key == "Ks")) {
            float *dst = key == "Ka" ? mats[cur].ka : key == "Kd" ? mats[cur].kd : mats[cur].ks;
            float a = 0, b = 0, c = 0; ss >> a;
            if (ss >> b) { ss >> c; } else { b = a; c = a; }
            dst[0] = a; dst[1] = b; dst[2] = c;
        }
    }
}

// one "v/vt/vn" corner; indices are 1-based, negative = relative to the end
bool parse_corner(const std::string &tok, int nv, int nt, int nn, int out[3]) {
    out[0] = out[1] = out[2] = 0;
    int field = 0; size_t i = 0;
    while (field < 3 && i <= tok.size()) {
        size_t j = tok.find('/', i);
        if (j == std::string::npos) j = tok.size();
        if (j > i) {
            int v = std::atoi(tok.substr(i, j - i).c_str());
            int count = field == 0 ? nv : field == 1 ? nt : nn;
            if (v < 0) v = count + 1 + v;
            out[field] = v;
        }
        field++; i = j + 1;
    }
    return out[0] >= 1 && out[0] <= nv;
}

}  // namespace

int load_obj(const std::string &path, std::vector<HostMesh> &meshes, std::string &err) {
    std::ifstream f(path);
    if (!f.is_open()) { err = "cannot open " + path; return RT_E_IO; }
    const std::string dir = dir_of(path);
    std::vector<float> V, VT, VN;
    std::map<std::string, Material> mats;
    std::string curMat;
    bool curMatKnown = false;
    // A new mesh starts at every object/group statement and at every change of material,
    // and only meshes that received faces survive - the behaviour of Assimp's OBJ importer
    // that Model::processNode (Model.h:73-89) then walks in file order.
    HostMesh cur; bool haveCur = false; std::string curMeshMat;
    auto flush = [&]() {
        if (haveCur && !cur.indices.empty()) meshes.push_back(cur);
        cur = HostMesh(); haveCur = false;
    };
    auto begin_mesh = [&]() {
        flush();
        haveCur = true; curMeshMat = curMat;
        Material m;
        if (curMatKnown) m = mats[curMat];
        std::memcpy(cur.ambient, m.ka, sizeof m.ka);
        std::memcpy(cur.diffuse, m.kd, sizeof m.kd);
        std::memcpy(cur.specular, m.ks, sizeof m.ks);
    };
    std::string line;
    int lineno = 0;
    while (std::getline(f, line)) {
        lineno++;
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key; ss >> key;
        if (key == "v") { float x = 0, y = 0, z = 0; ss >> x >> y >> z; V.insert(V.end(), {x, y, z}); }
        else if (key == "vt") { float u = 0, v = 0; ss >> u >> v; VT.insert(VT.end(), {u, v}); }
        else if (key == "vn") { float x = 0, y = 0, z = 0; ss >> x >> y >> z; VN.insert(VN.end(), {x, y, z}); }
        else if (key == "mtllib") { std::string name; std::getline(ss, name); load_mtl(dir, trim(name), mats); }
        else if (key == "usemtl") {
            ss >> curMat; curMatKnown = mats.count(curMat) > 0;
            if (haveCur && !cur.indices.empty() && curMat != curMeshMat) begin_mesh();
            else if (haveCur) {     // mesh still empty: it simply takes the new material
                curMeshMat = curMat;
                Material m; if (curMatKnown) m = mats[curMat];
                std::memcpy(cur.ambient, m.ka, sizeof m.ka);
                std::memcpy(cur.diffuse, m.kd, sizeof m.kd);
                std::memcpy(cur.specular, m.ks, sizeof m.ks);
            }
        }
        else if (key == "o" || key == "g") { begin_mesh(); }
        else if (key == "f") {
            if (!haveCur) begin_mesh();
            std::vector<std::array<int, 3>> corners;
            std::string tok;
            const int nv = (int)V.size() / 3, nt = (int)VT.size() / 2, nn = (int)VN.size() / 3;
            while (ss >> tok) {
                int c[3];
                if (!parse_corner(tok, nv, nt, nn, c)) { err = path + ":" + std::to_string(lineno) + ": bad face index"; return RT_E_IO; }
                corners.push_back({c[0], c[1], c[2]});
            }
            if (corners.size() < 3) continue;
            // aiProcess_Triangulate on a convex polygon: fan (0,i,i+1); vertices are not shared
            for (size_t i = 1; i + 1 < corners.size(); i++) {
                const size_t tri[3] = {0, i, i + 1};
                for (size_t k : tri) {
                    const auto &c = corners[k];
                    const uint32_t idx = (uint32_t)(cur.positions.size() / 3);
                    cur.positions.insert(cur.positions.end(), {V[3 * (c[0] - 1)], V[3 * (c[0] - 1) + 1], V[3 * (c[0] - 1) + 2]});
                    if (c[2] >= 1 && c[2] <= nn) cur.normals.insert(cur.normals.end(), {VN[3 * (c[2] - 1)], VN[3 * (c[2] - 1) + 1], VN[3 * (c[2] - 1) + 2]});
                    else cur.normals.insert(cur.normals.end(), {0.0f, 0.0f, 0.0f});
                    // aiProcess_FlipUVs: v -> 1 - v
                    if (c[1] >= 1 && c[1] <= nt) cur.uvs.insert(cur.uvs.end(), {VT[2 * (c[1] - 1)], 1.0f - VT[2 * (c[1] - 1) + 1]});
                    else cur.uvs.insert(cur.uvs.end(), {0.0f, 0.0f});
                    cur.indices.push_back(idx);
                }
            }
        }
    }
    flush();
    return RT_OK;
}

namespace {

// Scene::pushModel (Scene.h:54-65) followed by the flattening of BVHTree::build
// (BVH.h:120-135): Position = Position*scale, then += translation (two roundings).
void append_mesh(RtScene *scene, const HostMesh &m, const float scale[3], const float trans[3]) {
    for (size_t j = 0; j + 2 < m.indices.size(); j += 3) {
        RtTriangle t; std::memset(&t, 0, sizeof t);
        for (int k = 0; k < 3; k++) {
            const uint32_t idx = m.indices[j + k];
            for (int c = 0; c < 3; c++) {
                volatile float p = m.positions[3 * idx + c] * scale[c];
                t.v[k][c] = p + trans[c];
                t.n[k][c] = m.normals[3 * idx + c];
            }
            t.tex[k][0] = m.uvs[2 * idx]; t.tex[k][1] = m.uvs[2 * idx + 1];
        }
        for (int c = 0; c < 3; c++) { t.ka[c] = m.ambient[c]; t.kd[c] = m.diffuse[c]; t.ks[c] = m.specular[c]; }
        scene->triangles.push_back(t);
    }
}

HostMesh quad_mesh(const float (*verts)[3], const int (*faces)[4], int nfaces, const float uv[4][2],
                   const float ka[3], const float kd[3], const float ks[3]) {
    HostMesh m;
    const float nrm[3] = {0.0f, -1.0f, 0.0f};       // every face of both files uses "vn 0 -1 0"
    for (int f = 0; f < nfaces; f++) {
        const int fan[6] = {0, 1, 2, 0, 2, 3};
        for (int k = 0; k < 6; k++) {
            const int corner = fan[k];
            const float *p = verts[faces[f][corner] - 1];
            m.indices.push_back((uint32_t)(m.positions.size() / 3));
            m.positions.insert(m.positions.end(), {p[0], p[1], p[2]});
            m.normals.insert(m.normals.end(), {nrm[0], nrm[1], nrm[2]});
            m.uvs.insert(m.uvs.end(), {uv[corner][0], uv[corner][1]});
        }
    }
    std::memcpy(m.ambient, ka, 12); std::memcpy(m.diffuse, kd, 12); std::memcpy(m.specular, ks, 12);
    return m;
}

// The room of objects/room.obj + room.mtl as the loader above returns it: 7 meshes (the
// eighth object's only face is commented out in the file), 10 quads.
std::vector<HostMesh> room_meshes() {
    static const float v[12][3] = {
        {-0.5f, 1.0f, 0.5f}, {0.5f, 1.0f, 0.5f}, {0.5f, 1.0f, -0.5f}, {-0.5f, 1.0f, -0.5f},
        {-2.0f, -1.0f, 1.0f}, {-2.0f, 1.0f, 1.0f}, {-2.0f, 1.0f, -1.0f}, {-2.0f, -1.0f, -1.0f},
        {1.0f, -1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, -1.0f}, {1.0f, -1.0f, -1.0f}};
    static const float uv[4][2] = {{0.0f, 0.0f}, {0.0f, 1.0f}, {1.0f, 1.0f}, {1.0f, 0.0f}};   // vt 1..4 with v flipped
    static const float ks[3] = {1.0f, 1.0f, 0.009961f};
    static const float zero[3] = {0, 0, 0}, one[3] = {1.0f, 1.0f, 1.0f};
    static const float kdLight[3] = {0.709804f, 0.690196f, 0.627451f};
    static const float kdLeft[3] = {0.824222f, 0.092157f, 0.158824f};
    static const float kdRight[3] = {0.124222f, 0.392157f, 0.858824f};
    static const float kdWall[3] = {0.824222f, 0.824222f, 0.824222f};
    static const float kdFloor[3] = {0.924222f, 0.924222f, 0.924222f};
    static const int light[1][4] = {{1, 2, 3, 4}}, left[1][4] = {{5, 6, 7, 8}}, right[1][4] = {{9, 10, 11, 12}};
    static const int back[1][4] = {{8, 12, 11, 7}}, front[1][4] = {{5, 9, 10, 6}}, floor_[1][4] = {{5, 9, 12, 8}};
    static const int ceiling[4][4] = {{6, 1, 4, 7}, {6, 10, 2, 1}, {10, 11, 3, 2}, {11, 7, 4, 3}};
    std::vector<HostMesh> out;
    out.push_back(quad_mesh(v, light, 1, uv, one, kdLight, ks));
    out.push_back(quad_mesh(v, left, 1, uv, zero, kdLeft, ks));
    out.push_back(quad_mesh(v, right, 1, uv, zero, kdRight, ks));
    out.push_back(quad_mesh(v, back, 1, uv, zero, kdWall, ks));
    out.push_back(quad_mesh(v, front, 1, uv, zero, kdWall, ks));
    out.push_back(quad_mesh(v, floor_, 1, uv, zero, kdFloor, ks));
    out.push_back(quad_mesh(v, ceiling, 4, uv, zero, kdFloor, ks));
    return out;
}

// objects/cube.obj + cube.mtl: one mesh of 6 quads, every corner uses "vt 0 1" (flipped: 0 0).
std::vector<HostMesh> cube_meshes() {
    static const float v[8][3] = {{-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}, {-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}};
    static const int faces[6][4] = {{1, 2, 3, 4}, {5, 6, 7, 8}, {4, 3, 7, 8}, {1, 2, 6, 5}, {1, 4, 8, 5}, {2, 3, 7, 6}};
    static const float uv[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    static const float zero[3] = {0, 0, 0};
    static const float kd[3] = {0.824222f, 0.692157f, 0.158824f};
    static const float ks[3] = {1.0f, 1.0f, 0.009961f};
    return {quad_mesh(v, faces, 6, uv, zero, kd, ks)};
}

// ---- synthetic geometry ----------------------------------------------------------------

struct D3 { double x, y, z; };
inline D3 dnorm(D3 a) { double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return {a.x / l, a.y / l, a.z / l}; }
inline D3 dmid(D3 a, D3 b) { return dnorm({a.x + b.x, a.y + b.y, a.z + b.z}); }

struct Displace {
    double k[4][3], phase[4], freq[4];
    explicit Displace(uint32_t seed) {
        uint32_t s = seed ? seed : 1u;
        auto next = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (double)(s >> 8) / 16777216.0; };
        for (int o = 0; o < 4; o++) {
            D3 d = dnorm({next() * 2 - 1, next() * 2 - 1, next() * 2 - 1 + 1e-3});
            k[o][0] = d.x; k[o][1] = d.y; k[o][2] = d.z;
            phase[o] = next() * 6.283185307179586;
            freq[o] = 3.0 * (double)(1 << o);
        }
    }
    double operator()(D3 p) const {
        double a = 0, amp = 1.0;
        for (int o = 0; o < 4; o++) {
            a += amp * std::sin(freq[o] * (k[o][0] * p.x + k[o][1] * p.y + k[o][2] * p.z) + phase[o]);
            amp *= 0.5;
        }
        return a;
    }
};

struct IcoSink {
    RtScene *scene; const Displace *disp; double radius; D3 centre; float kd[3], ks[3];
    void emit(D3 a, D3 b, D3 c) {
        RtTriangle t; std::memset(&t, 0, sizeof t);
        const D3 p[3] = {a, b, c};
        for (int k = 0; k < 3; k++) {
            const double r = radius * (1.0 + 0.05 * (*disp)(p[k]));
            t.v[k][0] = (float)(centre.x + r * p[k].x);
            t.v[k][1] = (float)(centre.y + r * p[k].y);
            t.v[k][2] = (float)(centre.z + r * p[k].z);
            t.n[k][0] = (float)p[k].x; t.n[k][1] = (float)p[k].y; t.n[k][2] = (float)p[k].z;
        }
        std::memcpy(t.kd, kd, 12); std::memcpy(t.ks, ks, 12);
        scene->triangles.push_back(t);
    }
    void subdivide(D3 a, D3 b, D3 c, int level) {
        if (level == 0) { emit(a, b, c); return; }
        const D3 ab = dmid(a, b), bc = dmid(b, c), ca = dmid(c, a);
        subdivide(a, ab, ca, level - 1);
        subdivide(ab, b, bc, level - 1);
        subdivide(ca, bc, c, level - 1);
        subdivide(ab, bc, ca, level - 1);
    }
};

void push_icosphere(RtScene *scene, int level, double radius, D3 centre, uint32_t seed) {
    const double t = (1.0 + std::sqrt(5.0)) / 2.0;
    const D3 v[12] = {dnorm({-1, t, 0}), dnorm({1, t, 0}), dnorm({-1, -t, 0}), dnorm({1, -t, 0}),
                      dnorm({0, -1, t}), dnorm({0, 1, t}), dnorm({0, -1, -t}), dnorm({0, 1, -t}),
                      dnorm({t, 0, -1}), dnorm({t, 0, 1}), dnorm({-t, 0, -1}), dnorm({-t, 0, 1})};
    static const int f[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                                 {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                                 {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    Displace disp(seed);
    IcoSink sink{scene, &disp, radius, centre, {0.75f, 0.75f, 0.75f}, {1.0f, 1.0f, 0.0f}};
    for (int i = 0; i < 20; i++) sink.subdivide(v[f[i][0]], v[f[i][1]], v[f[i][2]], level);
}

void push_column(RtScene *scene, int level, double cx, double cz, double radius, double height, int material) {
    static const float kds[3][3] = {{0.8f, 0.6f, 0.4f}, {1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}};
    static const float kss[3][3] = {{1.0f, 1.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};   // diffuse / mirror / glass
    const double twoPi = 6.283185307179586;
    auto P = [&](int i, int j, float out[3], float nrm[3]) {
        const double a = twoPi * (double)(i % level) / (double)level;
        const double y = height * (double)j / (double)level;
        const double r = radius * (1.0 + 0.15 * std::sin(6.0 * y / height * twoPi / 4.0));
        out[0] = (float)(cx + r * std::cos(a)); out[1] = (float)y; out[2] = (float)(cz + r * std::sin(a));
        nrm[0] = (float)std::cos(a); nrm[1] = 0.0f; nrm[2] = (float)std::sin(a);
    };
    for (int j = 0; j < level; j++)
        for (int i = 0; i < level; i++) {
            float p[4][3], n[4][3];
            P(i, j, p[0], n[0]); P(i + 1, j, p[1], n[1]); P(i + 1, j + 1, p[2], n[2]); P(i, j + 1, p[3], n[3]);
            const int fan[2][3] = {{0, 1, 2}, {0, 2, 3}};
            for (int f = 0; f < 2; f++) {
                RtTriangle t; std::memset(&t, 0, sizeof t);
                for (int k = 0; k < 3; k++) { std::memcpy(t.v[k], p[fan[f][k]], 12); std::memcpy(t.n[k], n[fan[f][k]], 12); }
                std::memcpy(t.kd, kds[material], 12); std::memcpy(t.ks, kss[material], 12);
                scene->triangles.push_back(t);
            }
        }
}

}  // namespace
}  // namespace rt

extern "C" {

int rt_scene_create(RtScene **out) {
    if (!out) { rt::set_global_error("rt_scene_create: null pointer"); return RT_E_INVALID; }
    *out = new (std::nothrow) RtScene;
    if (!*out) { rt::set_global_error("rt_scene_create: out of memory"); return RT_E_NOMEM; }
    return RT_OK;
}

void rt_scene_destroy(RtScene *scene) { delete scene; }

int rt_scene_push_obj(RtScene *scene, const char *obj_path, const float scale[3], const float translation[3]) {
    if (!scene || !obj_path || !scale || !translation) { rt::set_global_error("rt_scene_push_obj: null pointer"); return RT_E_INVALID; }
    std::vector<rt::HostMesh> meshes; std::string err;
    int rc = rt::load_obj(obj_path, meshes, err);
    if (rc != RT_OK) { rt::set_global_error("rt_scene_push_obj: " + err); return rc; }
    for (const auto &m : meshes) rt::append_mesh(scene, m, scale, translation);
    return RT_OK;
}

int rt_scene_push_triangles(RtScene *scene, const RtTriangle *tris, int64_t n, const float scale[3], const float translation[3]) {
    if (!scene || (!tris && n > 0) || n < 0 || !scale || !translation) { rt::set_global_error("rt_scene_push_triangles: bad argument"); return RT_E_INVALID; }
    for (int64_t i = 0; i < n; i++) {
        RtTriangle t = tris[i];
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < 3; c++) {
                volatile float p = t.v[k][c] * scale[c];
                t.v[k][c] = p + translation[c];
            }
        scene->triangles.push_back(t);
    }
    return RT_OK;
}

int rt_scene_push_sphere(RtScene *scene, const RtSphere *sphere) {
    if (!scene || !sphere) { rt::set_global_error("rt_scene_push_sphere: null pointer"); return RT_E_INVALID; }
    scene->spheres.push_back(*sphere);
    return RT_OK;
}

static void push_demo_spheres(RtScene *scene) {
    // main.cpp:109-110
    const RtSphere glass = {{0.0f, 2.0f, 1.0f}, 1.5f, {0.9f, 0.9f, 0.9f}, 0.0f, 0.0f, 0.0f};
    const RtSphere mirror = {{-3.0f, 2.0f, 1.0f}, 1.5f, {0.2f, 0.2f, 0.9f}, 0.0f, 1.0f, 0.0f};
    scene->spheres.push_back(glass);
    scene->spheres.push_back(mirror);
}

int rt_scene_push_demo(RtScene *scene) {
    if (!scene) { rt::set_global_error("rt_scene_push_demo: null pointer"); return RT_E_INVALID; }
    const float roomScale[3] = {5.0f, 5.0f, 5.0f}, roomTrans[3] = {0.0f, 5.0f, 0.0f};        // main.cpp:106
    const float cubeScale[3] = {0.5f, 5.0f, -1.0f}, cubeTrans[3] = {0.0f, 1.0f, -2.0f};      // main.cpp:107
    for (const auto &m : rt::room_meshes()) rt::append_mesh(scene, m, roomScale, roomTrans);
    for (const auto &m : rt::cube_meshes()) rt::append_mesh(scene, m, cubeScale, cubeTrans);
    push_demo_spheres(scene);
    return RT_OK;
}

int rt_scene_push_synthetic(RtScene *scene, int kind, int level, int copies, uint32_t seed) {
    if (!scene || level < 0 || copies < 1) { rt::set_global_error("rt_scene_push_synthetic: bad argument"); return RT_E_INVALID; }
    if (kind == 0) {
        if (level > 10 || copies > 64) { rt::set_global_error("rt_scene_push_synthetic: level > 10 or copies > 64"); return RT_E_INVALID; }
        const float roomScale[3] = {5.0f, 5.0f, 5.0f}, roomTrans[3] = {0.0f, 5.0f, 0.0f};
        for (const auto &m : rt::room_meshes()) rt::append_mesh(scene, m, roomScale, roomTrans);
        scene->triangles.reserve(scene->triangles.size() + (size_t)copies * 20u * ((size_t)1 << (2 * level)));
        if (copies == 1) {
            rt::push_icosphere(scene, level, 3.0, {-2.5, 4.0, -1.0}, seed);
        } else {
            int g = 1; while (g * g * g < copies) g++;
            const double cell = 6.0 / g;
            for (int i = 0; i < copies; i++) {
                const int ix = i % g, iy = (i / g) % g, iz = i / (g * g);
                const double r = 0.42 * cell * (0.8 + 0.2 * (double)((i * 7) % 5) / 4.0);
                rt::push_icosphere(scene, level, r,
                                   {-2.5 - 3.0 + cell * (ix + 0.5), 4.0 - 3.0 + cell * (iy + 0.5), -1.0 - 3.0 + cell * (iz + 0.5)},
                                   seed);
            }
        }
        return RT_OK;
    }
    if (kind == 1) {
        if (level < 1 || level > 1024) { rt::set_global_error("rt_scene_push_synthetic: level out of range"); return RT_E_INVALID; }
        const float roomScale[3] = {15.0f, 15.0f, 15.0f}, roomTrans[3] = {0.0f, 15.0f, 0.0f};
        for (const auto &m : rt::room_meshes()) rt::append_mesh(scene, m, roomScale, roomTrans);
        const int cols = 8, rows = 4;
        for (int j = 0; j < rows; j++)
            for (int i = 0; i < cols; i++)
                rt::push_column(scene, level, -26.0 + 5.0 * i, -12.0 + 5.0 * j, 0.8, 14.0, (i + j * cols) % 3);
        push_demo_spheres(scene);
        (void)copies; (void)seed;
        return RT_OK;
    }
    rt::set_global_error("rt_scene_push_synthetic: unknown kind");
    return RT_E_INVALID;
}

int64_t rt_scene_triangle_count(const RtScene *scene) { return scene ? (int64_t)scene->triangles.size() : 0; }
int rt_scene_sphere_count(const RtScene *scene) { return scene ? (int)scene->spheres.size() : 0; }

int rt_scene_get_triangles(const RtScene *scene, RtTriangle *out, int64_t capacity) {
    if (!scene || !out || capacity < (int64_t)scene->triangles.size()) { rt::set_global_error("rt_scene_get_triangles: bad argument"); return RT_E_INVALID; }
    std::memcpy(out, scene->triangles.data(), scene->triangles.size() * sizeof(RtTriangle));
    return RT_OK;
}

int rt_scene_get_spheres(const RtScene *scene, RtSphere *out, int capacity) {
    if (!scene || !out || capacity < (int)scene->spheres.size()) { rt::set_global_error("rt_scene_get_spheres: bad argument"); return RT_E_INVALID; }
    std::memcpy(out, scene->spheres.data(), scene->spheres.size() * sizeof(RtSphere));
    return RT_OK;
}

}  // extern "C"
