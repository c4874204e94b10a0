// facade_test.cpp — drives the drop-in classes (host/gsx_facade.hpp) the way the reference's
// Application does (main.cpp:157-290, 415-481): Renderer::init, loadData, initialize, a uniform
// buffer written through queue.writeBuffer, a render pass, splatMesh.render(renderPass, camera,
// gui.params).  Dumps what each class drew so tests/test_gpu_facade.py can check it against the
// CPU oracle.   usage: facade_test <scene.splat> <outdir> <width> <height>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "../../gaussian-splatting_amd/host/gsx_facade.hpp"

using namespace gsx;

struct Uniforms {   // main.cpp:48-58
    mat4 projectionMatrix, viewMatrix, modelMatrix;
    float splatSize = 1, cutOff = 3.5, fov = 45.0;
    uint32_t bayerSize = 1;
    float bayerScale = 1.0f;
    float padding[3];
};
static_assert(sizeof(Uniforms) == sizeof(gsx_uniforms), "uniform block is 224 bytes");

template <typename T>
static void dump(const std::string& path, const T* data, size_t count) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(count * sizeof(T)));
}

template <typename Mesh>
static void run(const char* name, const std::string& scene_path, const std::string& out, int w, int h, GUI& gui, int orbit_steps) {
    GLFWwindow* window = glfwCreateWindow(w, h, "Gaussian Splatting", nullptr, nullptr);   // main.cpp:162-163
    Renderer renderer;
    renderer.init(window);                                         // main.cpp:188, the reference's signature
    if (renderer.window != window || renderer.width != w || renderer.height != h || !renderer.device || !renderer.queue ||
        !renderer.surface || !renderer.uncaptured_error_callback_handle)
        throw std::runtime_error("Renderer::init(GLFWwindow*) did not fill the reference's public members");
    Mesh splatMesh;
    splatMesh.loadData(scene_path, true);                          // main.cpp:428
    splatMesh.initialize(renderer.device, renderer.queue);         // main.cpp:431
    splatMesh.mirrorSortedIndices = true;

    // scene graph exactly as main.cpp:90-94,469-481
    auto scene = std::make_shared<Node>();
    auto orbitCamera = std::make_shared<OrbitCamera>(5.0f);
    auto camera = orbitCamera->camera;
    camera->aspect = static_cast<float>(w) / static_cast<float>(h);
    scene->addChild(orbitCamera);
    auto splatNode = std::make_shared<Node>();
    scene->addChild(splatNode);
    for (int s = 0; s < orbit_steps; ++s) orbitCamera->rotate(vec3(0.0f, 1.0f, 0.0f), 0.21f);

    BufferDescriptor bd;
    bd.size = sizeof(Uniforms);
    bd.usage = BufferUsage::CopyDst | BufferUsage::Uniform;
    Buffer transformBuffer = renderer.device.createBuffer(bd);     // main.cpp:433-438
    BindGroup bindGroup{transformBuffer};

    for (int frame = 0; frame < 2; ++frame) {                      // two frames: the second exercises cached cuts
        scene->updateWorldMatrix(mat4(1.0f));                      // main.cpp:483-486
        Uniforms uniforms;
        uniforms.projectionMatrix = camera->getProjectionMatrix(); // main.cpp:222-231
        uniforms.viewMatrix = camera->getViewMatrix();
        uniforms.modelMatrix = splatNode->worldMatrix;
        uniforms.splatSize = gui.params.splatSize;
        uniforms.cutOff = gui.params.cutOff;
        uniforms.fov = gui.params.fov;
        uniforms.bayerSize = gui.params.bayerSize;
        uniforms.bayerScale = gui.params.bayerScale;
        renderer.queue.writeBuffer(transformBuffer, 0, &uniforms, sizeof(Uniforms));   // main.cpp:239-240
        TextureView targetView = renderer.get_next_surface_texture_view();   // main.cpp:243-253
        if (!targetView) throw std::runtime_error("no surface texture view");
        CommandEncoderDescriptor encoderDesc = {};
        encoderDesc.label = "My command encoder";
        CommandEncoder encoder = wgpuDeviceCreateCommandEncoder(renderer.device, &encoderDesc);
        RenderPassEncoder renderPass = renderer.create_render_pass_encoder(targetView, encoder);
        renderPass.setPipeline(renderer.pipeline);
        renderPass.setBindGroup(0, bindGroup, 0, nullptr);
        splatMesh.render(renderPass, camera, gui.params);          // main.cpp:261
        renderPass.end();
        renderPass.release();                                      // main.cpp:265-289, unedited
        CommandBufferDescriptor cmdBufferDescriptor = {};
        cmdBufferDescriptor.label = "Command buffer";
        CommandBuffer command = encoder.finish(cmdBufferDescriptor);
        encoder.release();
        renderer.queue.submit(1, &command);
        command.release();
        renderer.surface.present();
        renderer.device.poll(false);
        if (frame == 1) {
            const std::string p = out + "/" + name;
            dump(p + ".splats", splatMesh.splatData.data(), splatMesh.splatData.size());
            dump(p + ".indices", splatMesh.indices.data(), splatMesh.indices.size());
            dump(p + ".rgba", targetView.pixels->data(), targetView.pixels->size());
            dump(p + ".uniforms", &uniforms, 1);
            dump(p + ".world", camera->worldMatrix.data(), 16);
            std::printf("%s n=%zu drawn=%u indices=%zu\n", name, splatMesh.splatData.size(), renderPass.drawnInstances,
                        splatMesh.indices.size());
        }
        targetView.release();                                      // main.cpp:280
    }
    glfwDestroyWindow(window);                                     // main.cpp:205
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s scene.splat outdir width height\n", argv[0]); return 2; }
    const std::string scene = argv[1], out = argv[2];
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    try {
        GUI gui;
        run<SplatMesh>("base", scene, out, w, h, gui, 1);
        gui.params.min_screen_area = 1.5f;            // octree threshold = 0.015 (SplatMeshOctree.hpp:44)
        run<SplatMeshOctree>("octree", scene, out, w, h, gui, 2);
        gui.params.depth = 40;
        run<SplatMeshHC>("hc", scene, out, w, h, gui, 3);
        gui.params.depth = 3;
        run<SplatMeshGridHC>("gridhc", scene, out, w, h, gui, 4);
        // loadSplatsRaw on a missing file gives an empty scene (ResourceManager.cpp:38-40): renders the clear colour
        run<SplatMesh>("missing", out + "/does_not_exist.splat", out, w, h, gui, 0);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "facade_test failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
