// gsx_scene.hpp — scene-graph inputs of the path: Node (Node.h:8-36, Node.cpp:10-27),
// Camera (Camera.h:9-35) and OrbitCamera's pose construction (OrbitCamera.h:24-30),
// on gsx_math.  Only what produces the matrices the path consumes; UI handlers
// (mouse/scroll) are out of scope.
#pragma once
#include <memory>
#include <vector>

#include "gsx_math.hpp"

namespace gsx {

class Node {
public:
    std::vector<std::shared_ptr<Node>> children;
    vec3 position{0.0f};
    vec3 scale{1.0f};
    mat3 rotation{1.0f};
    mat4 localMatrix{1.0f};
    mat4 worldMatrix{1.0f};
    virtual ~Node() = default;

    void addChild(std::shared_ptr<Node> child) {          // Node.cpp:5-8
        children.push_back(child);
        child->updateWorldMatrix(worldMatrix);
    }
    void updateWorldMatrix(const mat4& parentWorldMatrix) {  // Node.cpp:10-16
        updateLocalMatrix();
        worldMatrix = parentWorldMatrix * localMatrix;
        for (auto& child : children) child->updateWorldMatrix(worldMatrix);
    }
    void updateLocalMatrix() {                            // Node.cpp:18-23
        localMatrix = mat4(1.0f);
        localMatrix = gsx::scale(localMatrix, scale);
        localMatrix = mat4(rotation) * localMatrix;
        localMatrix[3] = vec4(position, 1.0f);
    }
    void rotate(vec3 axis, float angle) {                 // Node.cpp:25-27
        rotation = rotation_about(axis, angle) * rotation;
    }
};

class Camera : public Node {
public:
    using Ptr = std::shared_ptr<Camera>;
    float fov = 45.0f, aspect = 1.0f, near = 0.1f, far = 100.0f;  // Camera.h:19-24
    mat4 getViewMatrix() { return inverse(worldMatrix); }                                    // Camera.h:27-29
    mat4 getProjectionMatrix() { return perspective(radians(fov), aspect, near, far); }     // Camera.h:32-34
};

class OrbitCamera : public Node {
public:
    std::shared_ptr<Camera> camera = std::make_shared<Camera>();
    OrbitCamera() { children.push_back(camera); }
    explicit OrbitCamera(float distance) : OrbitCamera() { camera->position = vec3(0.0f, 0.0f, distance); }
};

}  // namespace gsx
