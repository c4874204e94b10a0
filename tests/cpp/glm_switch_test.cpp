// compile-only check of GSX_FACADE_WITH_GLM against the reference's vendored glm: the caller-side lines of
// main.cpp:48-58,222-224 and SplatMesh.h:84-85 with glm types.
#define GSX_FACADE_WITH_GLM
#include "gsx_facade.hpp"
#include <cstdio>
struct Uniforms { glm::mat4x4 projectionMatrix, viewMatrix, modelMatrix; float splatSize = 1, cutOff = 3.5f, fov = 45.0f; uint32_t bayerSize = 1; float bayerScale = 1.0f; float padding[3]; };
static_assert(sizeof(Uniforms) == 224, "Uniforms block");
int main() {
    gsx::mat4 v(1.0f);
    Uniforms u;
    u.viewMatrix = v;                         // camera->getViewMatrix() into the glm block
    u.projectionMatrix = gsx::mat4(2.0f);
    glm::vec3 cam(1.0f, 2.0f, 3.0f);
    gsx::vec3 g = cam;                        // sortSplats(indices, glm::vec3)
    glm::vec3 back = g;
    gsx::mat4 m2 = u.projectionMatrix;
    std::printf("%g %g %g %g\n", back.x, back.z, m2[1][1], u.viewMatrix[3][3]);
    return (back.z == 3.0f && m2[1][1] == 2.0f && u.viewMatrix[3][3] == 1.0f) ? 0 : 1;
}
