// Internal declarations shared by the host-side sources of librt_amd.so.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace rt {

// Last error of calls that have no context (host-side entry points).
void set_global_error(const std::string &msg);
const char *global_error();

// Worker threads the host-side builders may use at once: RT_AMD_HOST_THREADS (1..64) when set,
// else the machine's hardware threads, at most 16.  A job that runs one process per GPU sets the
// variable so that the ranks together do not oversubscribe the host.
int host_threads();
// Levels of two-way forks that stay within host_threads().
int host_fork_levels();

// Host-side mesh as the reference's loader leaves it (Mesh.h:18-151, data fields only):
// un-indexed triangles plus the per-mesh Ka/Kd/Ks of Model.h:158-165.
struct HostMesh {
    std::vector<float> positions;   // 3 per vertex
    std::vector<float> normals;     // 3 per vertex
    std::vector<float> uvs;         // 2 per vertex
    std::vector<uint32_t> indices;  // 3 per triangle
    float ambient[3] = {0, 0, 0};
    float diffuse[3] = {0, 0, 0};
    float specular[3] = {0, 0, 0};
};

// Loads a Wavefront OBJ the way the reference's Model(path) sees it (Model.h:54-182).
int load_obj(const std::string &path, std::vector<HostMesh> &meshes, std::string &err);

}  // namespace rt

struct RtScene {
    std::vector<RtTriangle> triangles;  // BVH.h:120-135 order
    std::vector<RtSphere> spheres;
};
