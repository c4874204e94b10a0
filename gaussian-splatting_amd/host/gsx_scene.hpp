// gsx_scene.hpp — scene-graph inputs of the path: Node (Node.h:8-36, Node.cpp:10-27),
// Camera (Camera.h:9-35) and OrbitCamera (OrbitCamera.h:8-75: pose construction and the
// mouse / scroll handlers the viewer front end forwards, host/gsx_viewer.hpp), on gsx_math.
#pragma once
#include <algorithm>
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

    float scrollRate = 5.0f, minDistance = 0.1f;      // OrbitCamera.h:18-22
    float orbitRate = 0.3f, maxOrbitRate = 0.5f;

    // OrbitCamera.h:32-37: the dolly runs in double and is rounded once into the float position
    void onScroll(double yoffset, double deltaTime) {
        camera->position.z = static_cast<float>(static_cast<double>(camera->position.z) + scrollRate * yoffset * (-1) * deltaTime);
        if (camera->position.z < minDistance) camera->position.z = minDistance;
    }
    void onMouseButton(int button, int action) {      // OrbitCamera.h:39-43: left button, 1 = pressed
        if (button == 0) pressed = action == 1;
    }
    void onMouseMove(double xpos, double ypos, double deltaTime) {   // OrbitCamera.h:45-49
        orbit(xpos, ypos, deltaTime);
        p_xpos = xpos;
        p_ypos = ypos;
    }
    // OrbitCamera.h:51-74: yaw about world +Y, pitch about the node's own X axis, both clamped per event; a pitch that
    // would carry the camera over the pole (its forward axis turning against the horizontal forward) is taken back.
    void orbit(double xpos, double ypos, double deltaTime) {
        if (!pressed) return;
        float dx = static_cast<float>(xpos - p_xpos);
        float dy = static_cast<float>(ypos - p_ypos);
        const float rate = static_cast<float>(orbitRate * deltaTime);
        dx = std::min(maxOrbitRate, std::max(-maxOrbitRate, dx * rate));
        dy = std::min(maxOrbitRate, std::max(-maxOrbitRate, dy * rate));
        rotate(vec3(0.0f, 1.0f, 0.0f), -dx);
        rotate(rotation[0], -dy);
        const vec3 cameraForward = rotation * vec3(0.0f, 0.0f, -1.0f);
        const vec3 cameraRight = rotation * vec3(1.0f, 0.0f, 0.0f);
        const vec3 forward = cross(cameraRight, vec3(0.0f, 1.0f, 0.0f));
        if (dot(forward, cameraForward) > 0) rotate(rotation[0], dy);
    }

private:
    double p_xpos = 0.0, p_ypos = 0.0;   // (uninitialised in the reference: a press before the first move is undefined there)
    bool pressed = false;
};

}  // namespace gsx
