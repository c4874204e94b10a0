// gsx_viewer.hpp — the optional viewer front end: the reference's Application (main.cpp:23-108,
// 139-294, 415-486) on the drop-in classes of gsx_facade.hpp, for a GPU without a display.
//
// An MI355X has no display engine and the box no window system, so the two ends of the reference's viewer that
// touch a desktop are replaced and everything between them is the reference's frame loop, call for call:
//
//   input   glfwPollEvents + the three GLFW callbacks (main.cpp:168-185,216)  ->  EventScript: the same callbacks
//           (onMouseMove / onMouseButton / onScroll) fed from a text stream — a file, or a pipe a remote front end
//           writes into — one window event per line, `frame` = "the loop comes round"
//   panel   ImGui sliders (gui.cpp:58-122)  ->  `set <name> <value>` lines, clamped to the sliders' ranges;
//           the panel's statistics line (gui.cpp:130-131) goes to the log, from the same 60-frame average ImGui keeps
//   output  surface.present() (main.cpp:282)  ->  Surface::onPresent: every frame as a binary PPM file and / or as
//           raw RGBA8 appended to a stream (a FIFO that ffplay / a socket relay reads)
//
// Script lines (blank lines and `#` comments are skipped):
//   move X Y            cursor position in window pixels            (glfwSetCursorPosCallback)
//   button B ACTION     mouse button B, ACTION 1 = press, 0 = release (glfwSetMouseButtonCallback)
//   scroll DY           wheel                                        (glfwSetScrollCallback)
//   set NAME VALUE      a panel slider: splatSize cutOff bayerSize bayerScale fov orbitRate scrollRate depth
//                       min_screen_area weight_e weight_w weight_d
//   dt SECONDS          fix ImGui's DeltaTime for the frames that follow (0 = measure the wall clock again)
//   frame [N]           run the main loop N times (default 1); events since the last `frame` are delivered by the first
//   quit                glfwWindowShouldClose
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <istream>
#include <sstream>

#include "gsx_facade.hpp"

namespace gsx {

struct ViewerEvent {
    enum Kind { Move, Button, Scroll, Set, DeltaTime, Frame, Quit } kind = Frame;
    double a = 0.0, b = 0.0;
    std::string name;
};

// One line of the script -> one event; throws std::invalid_argument naming the line on anything else.
inline bool parse_viewer_event(const std::string& line, ViewerEvent& ev) {
    std::istringstream in(line);
    std::string word;
    if (!(in >> word) || word[0] == '#') return false;
    auto need = [&](double& v) { if (!(in >> v)) throw std::invalid_argument("viewer script: missing number in `" + line + "`"); };
    ev = ViewerEvent{};
    if (word == "move") { ev.kind = ViewerEvent::Move; need(ev.a); need(ev.b); }
    else if (word == "button") { ev.kind = ViewerEvent::Button; need(ev.a); need(ev.b); }
    else if (word == "scroll") { ev.kind = ViewerEvent::Scroll; need(ev.a); }
    else if (word == "set") { ev.kind = ViewerEvent::Set; if (!(in >> ev.name)) throw std::invalid_argument("viewer script: `set` without a name"); need(ev.a); }
    else if (word == "dt") { ev.kind = ViewerEvent::DeltaTime; need(ev.a); }
    else if (word == "frame") {
        ev.kind = ViewerEvent::Frame;
        ev.a = 1.0;
        std::string count;
        if (in >> count) {
            char* end = nullptr;
            ev.a = std::strtod(count.c_str(), &end);
            if (end == count.c_str() || *end || !(ev.a >= 0.0)) throw std::invalid_argument("viewer script: bad frame count in `" + line + "`");
        }
    }
    else if (word == "quit") { ev.kind = ViewerEvent::Quit; }
    else throw std::invalid_argument("viewer script: unknown line `" + line + "`");
    return true;
}

// The panel of gui.cpp:58-122 without the pixels: name -> slider (field, range).  Returns false for an unknown name.
inline bool set_gui_parameter(GUI::Parameters& p, const std::string& name, double value) {
    auto f = [&](float& field, float lo, float hi) { field = std::min(hi, std::max(lo, static_cast<float>(value))); return true; };
    auto u = [&](uint32_t& field, int lo, int hi) { field = static_cast<uint32_t>(std::min(hi, std::max(lo, static_cast<int>(value)))); return true; };
    if (name == "splatSize") return f(p.splatSize, 0.001f, 5.0f);          // gui.cpp:58
    if (name == "cutOff") return f(p.cutOff, 0.001f, 5.0f);                // gui.cpp:59
    if (name == "bayerSize") return u(p.bayerSize, 0, 6);                  // gui.cpp:60
    if (name == "bayerScale") return f(p.bayerScale, 0.5f, 2.0f);          // gui.cpp:61
    if (name == "fov") return f(p.fov, 0.01f, 100.0f);                     // gui.cpp:75
    if (name == "orbitRate") return f(p.orbitRate, 0.01f, 2.0f);           // gui.cpp:91
    if (name == "scrollRate") return f(p.scrollRate, 0.1f, 20.0f);         // gui.cpp:92
    if (name == "depth") return u(p.depth, 0, 2000);                       // gui.cpp:106
    if (name == "min_screen_area") return f(p.min_screen_area, 0.0f, 1.0f);   // gui.cpp:107
    if (name == "weight_e") return f(p.weight_e, 0.0f, 1.0f);              // gui.cpp:120-122
    if (name == "weight_w") return f(p.weight_w, 0.0f, 1.0f);
    if (name == "weight_d") return f(p.weight_d, 0.0f, 1.0f);
    return false;
}

// What ImGuiIO gives the reference's loop: DeltaTime (main.cpp:211-212) and Framerate (gui.cpp:130-131; ImGui averages
// the last 60 frame times).  `fixed_dt` > 0 replaces the wall clock (scripts that must replay identically).
struct FrameClock {
    double DeltaTime = 1.0 / 60.0;     // ImGui's value before the second frame
    double Framerate = 0.0;
    double fixed_dt = 0.0;
    void new_frame() {
        const auto now = std::chrono::steady_clock::now();
        if (started) {
            const double wall = std::chrono::duration<double>(now - last).count();
            DeltaTime = fixed_dt > 0.0 ? fixed_dt : wall;
            window.push_back(wall);
            sum += wall;
            if (window.size() > 60) { sum -= window.front(); window.pop_front(); }
            Framerate = sum > 0.0 ? static_cast<double>(window.size()) / sum : 0.0;
        } else if (fixed_dt > 0.0) {
            DeltaTime = fixed_dt;
        }
        last = now;
        started = true;
    }
private:
    std::chrono::steady_clock::time_point last;
    std::deque<double> window;
    double sum = 0.0;
    bool started = false;
};

struct ViewerUniforms {   // main.cpp:48-58
    mat4 projectionMatrix, viewMatrix, modelMatrix;
    float splatSize = 1, cutOff = 3.5, fov = 45.0;
    uint32_t bayerSize = 1;
    float bayerScale = 1.0f;
    float padding[3] = {0, 0, 0};
};
static_assert(sizeof(ViewerUniforms) == sizeof(gsx_uniforms), "uniform block is 224 bytes");

// main.cpp:23-108.  Mesh = SplatMesh / SplatMeshOctree / SplatMeshHC / SplatMeshGridHC (main.cpp:96-98 picks one at
// compile time too).  Members and methods keep the reference's names; `events` replaces the GLFW event queue.
template <typename Mesh>
class Application {
public:
    std::istream* events = nullptr;        // the script; nullptr = no input, every MainLoop call renders one frame
    std::string scenePath;                 // main.cpp:428 loads RESOURCE_DIR "/splats/nike.splat"
    bool centerScene = true;
    float width = 1000, height = 800;      // main.cpp:101-102
    int gpu = 0;
    std::function<void(const uint8_t* rgba8, uint32_t w, uint32_t h, uint64_t frame)> onFrame;   // behind surface.present()
    std::FILE* log = stderr;
    uint32_t statsEvery = 60;              // frames between two statistics lines (0 = none)
    bool pipelined = false;                // present every frame two loops late and keep both frame slots busy (Renderer::setDeferredPresent)
    ViewerUniforms presentedUniforms;      // inside onFrame: the uniform block and camera of the frame being presented
    mat4 presentedWorld{1.0f};
    uint64_t framesPresented = 0;
    uint64_t framesRendered = 0;           // (ahead of framesPresented by up to two when pipelined)

    GUI gui;
    FrameClock imGuiIo;                    // (gui.imGuiIo in the reference)
    ViewerUniforms uniforms;
    std::shared_ptr<Node> scene = std::make_shared<Node>();                       // main.cpp:90-94
    std::shared_ptr<OrbitCamera> orbitCamera = std::make_shared<OrbitCamera>(5.0f);
    std::shared_ptr<Camera> camera = std::make_shared<Camera>();
    std::shared_ptr<Node> splatNode = std::make_shared<Node>();
    Mesh splatMesh;
    double deltaTime = 0.0;

    bool Initialize() {                                                           // main.cpp:157-201
        m_window = glfwCreateWindow(static_cast<int>(width), static_cast<int>(height), "Gaussian Splatting", nullptr, nullptr);
        m_renderer.gpu = gpu;
        m_renderer.init(m_window);
        if (!m_renderer.device) return false;
        InitializeBuffers();
        InitializeScene();
        m_renderer.surface.onPresent = [this](const uint8_t* px, uint32_t w, uint32_t h) {
            if (!inFlight.empty()) {
                presentedUniforms = inFlight.front().first;
                presentedWorld = inFlight.front().second;
                inFlight.pop_front();
            }
            if (onFrame) onFrame(px, w, h, framesPresented);
            ++framesPresented;
        };
        if (pipelined) m_renderer.setDeferredPresent(true);
        return true;
    }

    void Terminate() {                                                            // main.cpp:203-207
        m_renderer.finishPresent();                                               // (pipelined: the last two frames)
        pipeline.release();
        glfwDestroyWindow(m_window);
        m_window = nullptr;
    }

    bool IsRunning() { return !shouldClose; }                                      // main.cpp:292-294

    void MainLoop() {                                                             // main.cpp:209-290
        imGuiIo.new_frame();
        deltaTime = imGuiIo.DeltaTime;
        PollEvents();                                                             // main.cpp:216
        if (shouldClose) return;
        UpdateScene();
        uniforms.projectionMatrix = camera->getProjectionMatrix();                // main.cpp:222-236
        uniforms.viewMatrix = camera->getViewMatrix();
        uniforms.modelMatrix = splatNode->worldMatrix;
        uniforms.splatSize = gui.params.splatSize;
        uniforms.cutOff = gui.params.cutOff;
        uniforms.fov = gui.params.fov;
        uniforms.bayerSize = gui.params.bayerSize;
        uniforms.bayerScale = gui.params.bayerScale;
        camera->fov = uniforms.fov;
        orbitCamera->orbitRate = gui.params.orbitRate;
        orbitCamera->scrollRate = gui.params.scrollRate;
        m_renderer.queue.writeBuffer(transformBuffer, 0, &uniforms, sizeof(ViewerUniforms));   // main.cpp:239-240
        TextureView targetView = m_renderer.get_next_surface_texture_view();
        if (!targetView) return;
        CommandEncoderDescriptor encoderDesc = {};
        encoderDesc.label = "My command encoder";
        CommandEncoder encoder = wgpuDeviceCreateCommandEncoder(m_renderer.device, &encoderDesc);
        RenderPassEncoder renderPass = m_renderer.create_render_pass_encoder(targetView, encoder);
        renderPass.setPipeline(pipeline);
        renderPass.setBindGroup(0, bindGroup, 0, nullptr);
        splatMesh.render(renderPass, camera, gui.params);                         // main.cpp:261
        renderPass.end();                                                         // (gui.update's panel has no pixels here)
        renderPass.release();
        CommandBufferDescriptor cmdBufferDescriptor = {};
        cmdBufferDescriptor.label = "Command buffer";
        CommandBuffer command = encoder.finish(cmdBufferDescriptor);
        encoder.release();
        m_renderer.queue.submit(1, &command);
        command.release();
        drawnInstances = renderPass.drawnInstances;
        targetView.release();
        inFlight.emplace_back(uniforms, camera->worldMatrix);
        ++framesRendered;
        m_renderer.surface.present();                                             // main.cpp:282
        m_renderer.device.poll(false);
        if (statsEvery && log && framesRendered % statsEvery == 0 && imGuiIo.Framerate > 0.0) {
            const uint32_t* sorted = nullptr;           // (the draw records the selection's capacity; the cut's size is asked for only here)
            uint32_t drawn = drawnInstances;
            if (gsx_get_sorted_dev(m_renderer.device.ctx.get(), &sorted, &drawn) != GSX_OK) drawn = drawnInstances;
            std::fprintf(log, "%.3f ms/frame (%.1f FPS)  frame %llu  %u splats drawn\n", 1000.0 / imGuiIo.Framerate, imGuiIo.Framerate,   // gui.cpp:130-131
                         static_cast<unsigned long long>(framesRendered), drawn);
        }
    }

    // the event handlers of main.cpp:139-155 (no panel, so the mouse is never captured by one)
    void onMouseMove(double xpos, double ypos) { orbitCamera->onMouseMove(xpos, ypos, deltaTime); }
    void onMouseButton(int button, int action, int /*mods*/) { orbitCamera->onMouseButton(button, action); }
    void onScroll(double /*xoffset*/, double yoffset) { orbitCamera->onScroll(yoffset, deltaTime); }

    uint32_t drawnInstances = 0;
    Renderer& renderer() { return m_renderer; }

private:
    // Delivers the window events queued since the last frame.  A `frame N` line ends the delivery; the N - 1 loops
    // after it run without input, like a loop that finds the GLFW queue empty.  End of the script closes the window.
    void PollEvents() {
        if (idleFrames) { --idleFrames; return; }
        if (!events) return;
        std::string line;
        while (std::getline(*events, line)) {
            ViewerEvent ev;
            if (!parse_viewer_event(line, ev)) continue;
            switch (ev.kind) {
            case ViewerEvent::Move: onMouseMove(ev.a, ev.b); break;
            case ViewerEvent::Button: onMouseButton(static_cast<int>(ev.a), static_cast<int>(ev.b), 0); break;
            case ViewerEvent::Scroll: onScroll(0.0, ev.a); break;
            case ViewerEvent::Set:
                if (!set_gui_parameter(gui.params, ev.name, ev.a)) throw std::invalid_argument("viewer script: no slider named `" + ev.name + "`");
                // the handlers read the rates from the orbit camera, which the loop refreshes once per frame (main.cpp:235-236)
                break;
            case ViewerEvent::DeltaTime:
                imGuiIo.fixed_dt = ev.a;
                if (ev.a > 0.0) deltaTime = imGuiIo.DeltaTime = ev.a;
                break;
            case ViewerEvent::Frame:
                if (ev.a < 1.0) continue;
                idleFrames = static_cast<uint64_t>(ev.a) - 1u;
                return;
            case ViewerEvent::Quit: shouldClose = true; return;
            }
        }
        shouldClose = true;
    }

    void InitializeBuffers() {                                                    // main.cpp:415-439
        splatMesh.loadData(scenePath, centerScene);
        splatMesh.initialize(m_renderer.device, m_renderer.queue);
        BufferDescriptor bufferDesc;
        bufferDesc.size = sizeof(ViewerUniforms);
        bufferDesc.usage = BufferUsage::CopyDst | BufferUsage::Uniform;
        bufferDesc.mappedAtCreation = false;
        transformBuffer = m_renderer.device.createBuffer(bufferDesc);
        bindGroup = BindGroup{transformBuffer};                                   // main.cpp:441-467, binding 0
    }
    void InitializeScene() {                                                      // main.cpp:469-481
        camera = orbitCamera->camera;
        camera->aspect = width / height;
        scene->addChild(orbitCamera);
        splatNode->position = vec3(0.0f, 0.0f, 0.0f);
        splatNode->scale = vec3(1.0f, 1.0f, 1.0f);
        scene->addChild(splatNode);
    }
    void UpdateScene() {                                                          // main.cpp:483-486
        scene->updateWorldMatrix(mat4(1.0f));
        camera->fov = uniforms.fov;
    }

    GLFWwindow* m_window = nullptr;
    Renderer m_renderer;
    RenderPipeline pipeline;
    BindGroup bindGroup;
    Buffer transformBuffer;
    std::deque<std::pair<ViewerUniforms, mat4>> inFlight;   // frames rendered and not presented yet
    uint64_t idleFrames = 0;
    bool shouldClose = false;
};

// Frame sinks for Application::onFrame: binary PPM files (RGB, alpha dropped: the frame is opaque) and / or raw RGBA8.
struct FrameSink {
    std::string ppmDir;          // frame_00000.ppm, ... when not empty
    std::FILE* raw = nullptr;    // width * height * 4 bytes per frame when not null
    uint64_t every = 1;          // keep every n-th frame
    void operator()(const uint8_t* px, uint32_t w, uint32_t h, uint64_t frame) {
        if (every > 1 && frame % every) return;
        if (!ppmDir.empty()) {
            char name[64];
            std::snprintf(name, sizeof name, "/frame_%05llu.ppm", static_cast<unsigned long long>(frame));
            std::ofstream f(ppmDir + name, std::ios::binary);
            if (!f) throw std::runtime_error("viewer: cannot write into " + ppmDir);
            f << "P6\n" << w << " " << h << "\n255\n";
            rgb.resize(static_cast<size_t>(w) * h * 3);
            for (size_t i = 0, n = static_cast<size_t>(w) * h; i < n; ++i) { rgb[3 * i] = px[4 * i]; rgb[3 * i + 1] = px[4 * i + 1]; rgb[3 * i + 2] = px[4 * i + 2]; }
            f.write(reinterpret_cast<const char*>(rgb.data()), static_cast<std::streamsize>(rgb.size()));
        }
        if (raw) {
            if (std::fwrite(px, 4, static_cast<size_t>(w) * h, raw) != static_cast<size_t>(w) * h) throw std::runtime_error("viewer: the raw frame stream was closed");
            std::fflush(raw);
        }
    }
private:
    std::vector<uint8_t> rgb;
};

}  // namespace gsx
