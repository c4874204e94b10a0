// gsx_facade.hpp — the reference's class interfaces for the hot path, implemented on
// the libgsx C ABI (include/gsx.h) and nothing else:
//
//   Renderer                      renderer.hpp:10-35
//   GUI::Parameters               gui.hpp:16-30
//   SplatMesh                     SplatMesh.h:31-200
//   SplatMeshOctree               SplatMeshOctree.hpp:32-80
//   SplatMeshHC                   SplatMeshHC.hpp:33-92
//   SplatMeshGridHC               SplatMeshGridHC.hpp:33-41, SplatMeshGridHC.cpp
//
// Same class names, public members, method names, argument meaning and error
// behaviour, so `Application` code written against the reference (main.cpp:188,
// 239-240, 261, 428-431) compiles against this header after swapping the include.
// The `wgpu` namespace below is NOT WebGPU: it is the handful of handle types that
// code touches (last row), backed by host memory and a gsx context.
//
// Where the frame lands: the reference's render() records a draw into a WebGPU
// render pass; here render() runs select -> sort -> project -> bin -> raster on the
// GPU and leaves RGBA8 pixels in the pass's target view (`TextureView::pixels`),
// which is the "device framebuffer readback" the optional viewer front end shows.
#pragma once
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gsx.h"
#include "gsx_lod_host.hpp"
#include "gsx_ply.hpp"
#include "gsx_scene.hpp"
#include "gsx_splat.hpp"

// ---- the window type Renderer::init takes (renderer.hpp:28, renderer.cpp:3-5) ----
// A viewer build defines GSX_FACADE_WITH_GLFW and gets the real <GLFW/glfw3.h>.  Headless (the MI355X box has no
// display): GLFWwindow is a plain size record and the three GLFW calls the init path makes (main.cpp:162-163,205;
// renderer.cpp:5) exist with the same signatures, so `m_window = glfwCreateWindow(w, h, ...); m_renderer.init(m_window);`
// compiles and runs unedited.
#if defined(GSX_FACADE_WITH_GLFW)
#include <GLFW/glfw3.h>
#else
struct GLFWwindow { int width = 0, height = 0; };
struct GLFWmonitor;
inline GLFWwindow* glfwCreateWindow(int width, int height, const char* /*title*/, GLFWmonitor* /*monitor*/, GLFWwindow* /*share*/) {
    GLFWwindow* w = new GLFWwindow;
    w->width = width;
    w->height = height;
    return w;
}
inline void glfwGetWindowSize(GLFWwindow* window, int* width, int* height) {
    if (width) *width = window ? window->width : 0;
    if (height) *height = window ? window->height : 0;
}
inline void glfwDestroyWindow(GLFWwindow* window) { delete window; }
#endif

namespace gsx {

// Frames are read back from the GPU every render(): their bytes live in page-locked memory (gsx_host_alloc_pinned), which
// the device writes by DMA without a staging copy.  A std::vector with this allocator, so `pixels->data()` / `size()` stay.
template <typename T>
struct PinnedAllocator {
    using value_type = T;
    PinnedAllocator() = default;
    template <typename U> PinnedAllocator(const PinnedAllocator<U>&) {}
    T* allocate(size_t n) {
        void* p = gsx_host_alloc_pinned(static_cast<uint64_t>(n) * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { gsx_host_free_pinned(p); }
    template <typename U> bool operator==(const PinnedAllocator<U>&) const { return true; }
    template <typename U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
using FrameBytes = std::vector<uint8_t, PinnedAllocator<uint8_t>>;

namespace wgpu {

enum class BufferUsage : uint32_t { None = 0, CopyDst = 1, Storage = 2, Index = 4, Vertex = 8, Uniform = 16 };
inline BufferUsage operator|(BufferUsage a, BufferUsage b) { return static_cast<BufferUsage>(static_cast<uint32_t>(a) | static_cast<uint32_t>(b)); }
enum class IndexFormat { Undefined, Uint16, Uint32 };
enum class VertexFormat { Float32x2 };
enum class VertexStepMode { Vertex, Instance };
enum class TextureFormat { Undefined, RGBA8Unorm };

struct BufferDescriptor {
    uint64_t size = 0;
    BufferUsage usage = BufferUsage::None;
    bool mappedAtCreation = false;
};
struct Buffer {
    std::shared_ptr<std::vector<uint8_t>> host;   // host shadow (uniforms, quad geometry, last permutation)
    uint64_t getSize() const { return host ? host->size() : 0; }
    void release() { host.reset(); }
    explicit operator bool() const { return static_cast<bool>(host); }
};
struct VertexAttribute { uint32_t shaderLocation = 0; VertexFormat format = VertexFormat::Float32x2; uint64_t offset = 0; };
struct VertexBufferLayout {
    uint32_t attributeCount = 0;
    const VertexAttribute* attributes = nullptr;
    uint64_t arrayStride = 0;
    VertexStepMode stepMode = VertexStepMode::Vertex;
};

// Device = one gsx context (one GPU).  Copies share the context, like wgpu handles.
struct Device {
    std::shared_ptr<gsx_ctx> ctx;
    Buffer createBuffer(const BufferDescriptor& d) const {
        Buffer b;
        b.host = std::make_shared<std::vector<uint8_t>>(d.size);
        return b;
    }
    explicit operator bool() const { return static_cast<bool>(ctx); }
    void release() { ctx.reset(); }
    void poll(bool /*wait*/) const {}      // main.cpp:288 (wgpu-native); gsx calls complete on their own stream
    void tick() const {}                   // main.cpp:286 (Dawn)
};
struct CommandBuffer { void release() {} };
struct CommandBufferDescriptor { const char* label = nullptr; };
struct CommandEncoderDescriptor { const char* label = nullptr; };
struct Queue {
    std::shared_ptr<gsx_ctx> ctx;
    void writeBuffer(Buffer& b, uint64_t offset, const void* data, uint64_t size) const {
        if (!b.host || offset + size > b.host->size()) throw std::out_of_range("writeBuffer outside the buffer");
        std::memcpy(b.host->data() + offset, data, size);
    }
    void submit(uint32_t /*count*/, const CommandBuffer* /*commands*/) const {}   // main.cpp:275: render() already ran the frame
    explicit operator bool() const { return static_cast<bool>(ctx); }
    void release() { ctx.reset(); }
};
struct BindGroup { Buffer uniforms; };           // binding 0 of main.cpp:441-467 (the 224-byte Uniforms block)
struct TextureView {
    uint32_t width = 0, height = 0;
    std::shared_ptr<FrameBytes> pixels;   // RGBA8, row-major, y down (page-locked)
    explicit operator bool() const { return static_cast<bool>(pixels); }
    void release() { pixels.reset(); }
};
struct CommandEncoder {
    CommandBuffer finish(const CommandBufferDescriptor&) { return CommandBuffer{}; }   // main.cpp:271
    void release() {}
};
inline CommandEncoder wgpuDeviceCreateCommandEncoder(const Device&, const CommandEncoderDescriptor*) { return CommandEncoder{}; }   // main.cpp:249-250
struct RenderPipeline { void release() {} };
enum class ErrorType { NoError, Validation, OutOfMemory, Internal, Unknown, DeviceLost };
using ErrorCallback = std::function<void(ErrorType, const char*)>;   // webgpu.hpp's ErrorCallback
// The swap chain of renderer.cpp:73-95,123-151: one RGBA8 frame in host memory that render() fills by readback.
// present() hands it to `onPresent` (the optional viewer front end blits it); headless it is a no-op.
// With Renderer::setDeferredPresent(true) it is a ring of three such frames, like a swap chain with images in flight:
// render() queues frame k into ring[k % 3] and returns, and the present() of loop k + 2 shows frame k, which the library has
// delivered by then (gsx option deferred_readback); Renderer::finishPresent() shows the last two.
struct Surface {
    std::shared_ptr<FrameBytes> frame;
    uint32_t width = 0, height = 0;
    std::function<void(const uint8_t* rgba8, uint32_t width, uint32_t height)> onPresent;
    std::vector<std::shared_ptr<FrameBytes>> ring;     // deferred presentation: the frames in flight (empty = synchronous)
    uint64_t queued = 0, shown = 0;                    // frames handed to render passes / to onPresent
    void present() {                                   // main.cpp:282
        if (ring.empty()) { if (onPresent && frame) onPresent(frame->data(), width, height); return; }
        ++queued;
        if (queued - shown > 2) show_next();
    }
    void show_next() {
        frame = ring[shown % ring.size()];
        ++shown;
        if (onPresent) onPresent(frame->data(), width, height);
    }
    void unconfigure() {}
    void release() { frame.reset(); }
    explicit operator bool() const { return static_cast<bool>(frame); }
};
struct RenderPassEncoder {
    TextureView target;
    BindGroup bindGroup;
    uint32_t drawnInstances = 0;     // instance count of the last drawIndexed
    uint32_t y0 = 0, y1 = 0;         // tile band (rows); 0,0 = whole frame
    void setPipeline(const RenderPipeline&) {}
    void setBindGroup(uint32_t, const BindGroup& g, uint32_t, const uint32_t*) { bindGroup = g; }
    void setVertexBuffer(uint32_t, const Buffer&, uint64_t, uint64_t) {}
    void setIndexBuffer(const Buffer&, IndexFormat, uint64_t, uint64_t) {}
    void drawIndexed(uint32_t /*indexCount*/, uint32_t instanceCount, uint32_t, int32_t, uint32_t) { drawnInstances = instanceCount; }
    void end() {}
    void release() {}
};

}  // namespace wgpu

using namespace wgpu;

inline void gsx_check(gsx_ctx* ctx, int rc, const char* what) {
    if (rc != GSX_OK) throw std::runtime_error(std::string(what) + ": " + (ctx ? gsx_last_error(ctx) : "no context") + " (" + std::to_string(rc) + ")");
}

// gui.hpp:16-30
struct GUI {
    struct Parameters {
        float splatSize = 1.0f;
        float cutOff = 3.5f;
        float fov = 45.0f;
        uint32_t bayerSize = 1;
        float bayerScale = 1.0f;
        float orbitRate = 1.0f;
        float scrollRate = 1.0f;
        uint32_t depth = 0;
        float min_screen_area = 0.2f;
        float weight_e = 0.0f;
        float weight_w = 0.0f;
        float weight_d = 0.0f;
    } params;
};

// renderer.hpp:10-35: same public members and methods.  `gpu` (the HIP device ordinal, set before init) is the one
// addition; init(width, height, gpu) without a window is kept as an extra for headless callers.
class Renderer {
public:
    GLFWwindow* window = nullptr;
    int width = 0;
    int height = 0;

    Device device;
    Queue queue;
    Surface surface;
    TextureFormat surface_format = TextureFormat::Undefined;
    RenderPipeline pipeline;

    std::unique_ptr<ErrorCallback> uncaptured_error_callback_handle;

    int gpu = 0;

public:
    Renderer() = default;
    ~Renderer() {                                                 // renderer.cpp:178-183
        surface.unconfigure();
        queue.release();
        surface.release();
        device.release();
    }

    void init(GLFWwindow* window) {                               // renderer.cpp:3-100
        this->window = window;
        glfwGetWindowSize(window, &width, &height);
        bring_up();
    }
    void init(int frame_width, int frame_height, int gpu_ordinal = 0) {
        window = nullptr;
        width = frame_width;
        height = frame_height;
        gpu = gpu_ordinal;
        bring_up();
    }
    TextureView get_next_surface_texture_view() {                 // renderer.cpp:123-151
        TextureView v;
        if (!surface) return v;                                   // (the reference returns nullptr when the surface has no texture)
        v.width = surface.width;
        v.height = surface.height;
        v.pixels = surface.ring.empty() ? surface.frame : surface.ring[surface.queued % surface.ring.size()];
        return v;
    }
    // Pipelined presentation for front ends that can show a frame two loops late (see Surface): both of the library's frame
    // slots stay busy although every frame is read back.  Off (the default), present() shows the frame just rendered.
    void setDeferredPresent(bool on) {
        if (!device) throw std::runtime_error("Renderer::setDeferredPresent before init");
        finishPresent();
        gsx_check(device.ctx.get(), gsx_set_option(device.ctx.get(), "deferred_readback", on ? 1 : 0), "gsx_set_option(deferred_readback)");
        surface.ring.clear();
        surface.queued = surface.shown = 0;
        if (on)
            for (int i = 0; i < 3; ++i) surface.ring.push_back(std::make_shared<FrameBytes>(static_cast<size_t>(width) * height * 4));
    }
    // Delivers and shows the frames still in flight (deferred presentation; a no-op otherwise).
    void finishPresent() {
        if (surface.ring.empty() || !device) return;
        gsx_check(device.ctx.get(), gsx_sync(device.ctx.get()), "gsx_sync");
        while (surface.shown < surface.queued) surface.show_next();
    }
    RenderPassEncoder create_render_pass_encoder(TextureView& targetView, CommandEncoder&) {  // renderer.cpp:153-176
        RenderPassEncoder p;
        p.target = targetView;
        return p;   // the clear to (0.05,0.05,0.05,1) is applied by gsx_render (every pixel of the band is written)
    }

private:
    // Like the reference, a failed bring-up prints and returns (renderer.cpp:12-15,23-26,51-54): `device` stays null and the
    // first SplatMesh call on it throws.
    void bring_up() {
        if (width <= 0 || height <= 0) {
            std::cerr << "Failed to create a surface: window size " << width << " x " << height << "." << std::endl;
            return;
        }
        gsx_ctx* raw = nullptr;
        const int rc = gsx_create(gpu, &raw);
        if (rc != GSX_OK) {
            std::cerr << "Failed to get a gsx device on GPU " << gpu << " (" << rc << ")." << std::endl;
            return;
        }
        device.ctx = std::shared_ptr<gsx_ctx>(raw, [](gsx_ctx* c) { gsx_destroy(c); });
        uncaptured_error_callback_handle = std::make_unique<ErrorCallback>([](ErrorType type, const char* message) {   // renderer.cpp:58-63
            std::cout << "Uncaptured device error: type " << static_cast<int>(type);
            if (message) std::cout << " (" << message << ")";
            std::cout << std::endl;
        });
        queue.ctx = device.ctx;
        surface_format = TextureFormat::RGBA8Unorm;
        surface.width = static_cast<uint32_t>(width);
        surface.height = static_cast<uint32_t>(height);
        surface.frame = std::make_shared<FrameBytes>(static_cast<size_t>(width) * height * 4);
    }
};

class SplatMesh {
public:
    SplatVector splatData;
    std::vector<uint32_t> indices;

    Buffer splatBuffer;       // shadow only: the splats live in HBM as SoA once initialize() ran
    Buffer sortIndexBuffer;   // receives the permutation when mirrorSortedIndices is set
    Buffer quadBuffer;
    Buffer indexQuadBuffer;

    std::vector<VertexAttribute> splatAttribs;
    std::vector<VertexAttribute> quadAttribs;
    std::vector<VertexBufferLayout> vertexBufferLayouts;

    Device device;
    Queue queue;

    // The reference rewrites `indices` and `sortIndexBuffer` on the host every frame
    // (SplatMesh.h:55-57).  Here the permutation stays in HBM; set this to copy it back
    // into `indices` / `sortIndexBuffer` after every render() (costs a device->host copy).
    bool mirrorSortedIndices = false;

    virtual ~SplatMesh() {                                         // SplatMesh.h:116-122
        splatBuffer.release();
        sortIndexBuffer.release();
        quadBuffer.release();
        indexQuadBuffer.release();
        vertexBufferLayouts.clear();
    }

    virtual void render(RenderPassEncoder& renderPass, Camera::Ptr camera, GUI::Parameters& params) {  // SplatMesh.h:49-59
        (void)params;
        setBuffers(renderPass);
        const vec3 cameraPos = xyz(camera->worldMatrix[3]);
        ensureUploaded();
        if (!selectionIsAll) { gsx_check(ctx(), gsx_select_all(ctx()), "gsx_select_all"); selectionIsAll = true; }
        sortAndDraw(renderPass, cameraPos, static_cast<uint32_t>(splatData.size()));
    }

    virtual void loadData(const std::string& path, bool center) {  // SplatMesh.h:61-70
        SplatSplitVector s = loadSplatsAny(path, center);   // .splat as the reference; .ply too (gsx_ply.hpp)
        splatData.resize(s.size());
        for (size_t i = 0; i < s.size(); ++i) splatData[i] = split_to_splat(s[i]);
        indices.resize(splatData.size());
        std::iota(indices.begin(), indices.end(), 0);
        uploaded = false;
    }

    void initialize(Device& dev, Queue& q) {                       // SplatMesh.h:72-77
        device = dev;
        queue = q;
        initializeBuffers();
        initializeVertexBufferLayouts();
    }

    void setBuffers(RenderPassEncoder& renderPass) {               // SplatMesh.h:79-82
        renderPass.setVertexBuffer(0, quadBuffer, 0, quadBuffer.getSize());
        renderPass.setIndexBuffer(indexQuadBuffer, IndexFormat::Uint16, 0, indexQuadBuffer.getSize());
    }

    // SplatMesh.h:84-114: sorts `idx` in place, far -> near, ties by ascending index.
    void sortSplats(std::vector<uint32_t>& idx, vec3 cameraPos) {
        ensureUploaded();
        gsx_check(ctx(), gsx_select_list(ctx(), idx.data(), static_cast<uint32_t>(idx.size())), "gsx_select_list");
        selectionIsAll = false;
        cutValid = false;         // the device selection is now `idx`: a cached LOD cut must be re-selected by the next render()
        const float cam[3] = {cameraPos.x, cameraPos.y, cameraPos.z};
        std::vector<uint32_t> out(std::max<size_t>(idx.size(), 1));
        gsx_check(ctx(), gsx_sort(ctx(), cam, out.data()), "gsx_sort");
        const uint32_t* dev = nullptr;
        uint32_t m = 0;
        gsx_check(ctx(), gsx_get_sorted_dev(ctx(), &dev, &m), "gsx_get_sorted_dev");
        out.resize(m);            // duplicates in `idx` collapse (the reference never has any)
        idx.swap(out);
    }

protected:
    gsx_ctx* ctx() const {
        if (!device.ctx) throw std::runtime_error("SplatMesh used before initialize(device, queue)");
        return device.ctx.get();
    }
    virtual void uploadLod() {}
    void ensureUploaded() {
        if (uploaded) return;
        gsx_check(ctx(), gsx_upload_splats(ctx(), splatData.data(), splatData.size()), "gsx_upload_splats");
        uploaded = true;
        selectionIsAll = true;
        uploadLod();
    }
    // sort the current device selection, rasterize into the pass's target, record the draw
    void sortAndDraw(RenderPassEncoder& renderPass, vec3 cameraPos, uint32_t capacity) {
        const float cam[3] = {cameraPos.x, cameraPos.y, cameraPos.z};
        uint32_t m = capacity;
        if (mirrorSortedIndices) {
            const uint32_t* dev = nullptr;
            indices.resize(std::max<uint32_t>(capacity, 1));
            gsx_check(ctx(), gsx_sort(ctx(), cam, indices.data()), "gsx_sort");
            gsx_check(ctx(), gsx_get_sorted_dev(ctx(), &dev, &m), "gsx_get_sorted_dev");
            indices.resize(m);
            if (sortIndexBuffer.host && sortIndexBuffer.host->size() >= static_cast<size_t>(m) * 4)
                queue.writeBuffer(sortIndexBuffer, 0, indices.data(), static_cast<uint64_t>(m) * 4);   // SplatMesh.h:56-57
        } else {
            gsx_check(ctx(), gsx_sort(ctx(), cam, nullptr), "gsx_sort");
        }
        if (renderPass.target && renderPass.bindGroup.uniforms.host && renderPass.bindGroup.uniforms.host->size() >= sizeof(gsx_uniforms)) {
            gsx_uniforms u;
            std::memcpy(&u, renderPass.bindGroup.uniforms.host->data(), sizeof u);
            const uint32_t w = renderPass.target.width, h = renderPass.target.height;
            const uint32_t y0 = renderPass.y1 ? renderPass.y0 : 0, y1 = renderPass.y1 ? renderPass.y1 : h;
            gsx_check(ctx(), gsx_render(ctx(), &u, w, h, y0, y1, renderPass.target.pixels->data()), "gsx_render");
        }
        renderPass.drawIndexed(6, m, 0, 0, 0);                     // SplatMesh.h:58
    }

    bool uploaded = false;
    bool selectionIsAll = false;
    // HC / GridHC: the view-independent depth cut currently held by the device selection (cleared by anything else
    // that replaces the selection, e.g. the public sortSplats)
    bool cutValid = false;
    uint32_t cutDepth = 0;

private:
    void initializeBuffers() {                                     // SplatMesh.h:125-185
        BufferDescriptor d;
        d.size = splatData.size() * sizeof(Splat);
        d.usage = BufferUsage::CopyDst | BufferUsage::Storage;
        splatBuffer = device.createBuffer(BufferDescriptor{0, d.usage, false});   // no host copy kept: HBM SoA is the store
        ensureUploaded();
        d.size = splatData.size() * sizeof(uint32_t);
        sortIndexBuffer = device.createBuffer(d);
        const std::vector<uint16_t> indexQuadData = {0, 1, 2, 0, 2, 3};
        d.size = indexQuadData.size() * sizeof(uint16_t);
        d.usage = BufferUsage::CopyDst | BufferUsage::Index;
        indexQuadBuffer = device.createBuffer(d);
        queue.writeBuffer(indexQuadBuffer, 0, indexQuadData.data(), d.size);
        const float quadData[8] = {-1.0f, -1.0f, 1.0f, -1.0f, 1.0f, 1.0f, -1.0f, 1.0f};
        d.size = sizeof quadData;
        d.usage = BufferUsage::CopyDst | BufferUsage::Vertex;
        quadBuffer = device.createBuffer(d);
        queue.writeBuffer(quadBuffer, 0, quadData, d.size);
    }
    void initializeVertexBufferLayouts() {                         // SplatMesh.h:132-137,187-199
        vertexBufferLayouts.resize(1);
        quadAttribs.resize(1);
        quadAttribs[0].shaderLocation = 0;
        quadAttribs[0].format = VertexFormat::Float32x2;
        quadAttribs[0].offset = 0;
        vertexBufferLayouts[0].attributeCount = static_cast<uint32_t>(quadAttribs.size());
        vertexBufferLayouts[0].attributes = quadAttribs.data();
        vertexBufferLayouts[0].arrayStride = 2 * sizeof(float);
        vertexBufferLayouts[0].stepMode = VertexStepMode::Vertex;
    }
};

class SplatMeshOctree : public SplatMesh {
public:
    Octree octree;

    void render(RenderPassEncoder& renderPass, Camera::Ptr camera, GUI::Parameters& params) override {  // SplatMeshOctree.hpp:36-55
        setBuffers(renderPass);
        ensureUploaded();
        const mat4 vp = camera->getProjectionMatrix() * camera->getViewMatrix();   // BB.hpp:60-61
        gsx_check(ctx(), gsx_select_octree(ctx(), vp.data(), params.min_screen_area * 0.01f, nullptr), "gsx_select_octree");
        selectionIsAll = false;
        sortAndDraw(renderPass, xyz(camera->worldMatrix[3]), static_cast<uint32_t>(splatData.size()));
    }
    void loadData(const std::string& path, bool center) override {  // SplatMeshOctree.hpp:57-79
        SplatSplitVector s = loadSplatsAny(path, center);   // .splat as the reference; .ply too (gsx_ply.hpp)
        octree.build(std::move(s));
        octree.generate();
        splatData = octree.splats;
        uploaded = false;
    }

protected:
    void uploadLod() override {
        const OctreeFlat f = octree.flatten();
        gsx_octree_flat t{static_cast<uint32_t>(f.parent.size()), f.center3.data(), f.half3.data(), f.parent.data(), f.is_leaf.data()};
        gsx_check(ctx(), gsx_set_lod_octree(ctx(), &t), "gsx_set_lod_octree");
    }
};

class SplatMeshHC : public SplatMesh {
public:
    HC hc;

    void render(RenderPassEncoder& renderPass, Camera::Ptr camera, GUI::Parameters& params) override {  // SplatMeshHC.hpp:37-56
        setBuffers(renderPass);
        ensureUploaded();
        if (!cutValid || cutDepth != params.depth) {      // the depth cut is view independent: cache it
            indices = hc.get_indices_depth(params.depth);
            gsx_check(ctx(), gsx_select_list(ctx(), indices.data(), static_cast<uint32_t>(indices.size())), "gsx_select_list");
            cutDepth = params.depth;
            cutValid = true;
            selectionIsAll = false;
        }
        sortAndDraw(renderPass, xyz(camera->worldMatrix[3]), static_cast<uint32_t>(splatData.size()));
    }
    void loadData(const std::string& path, bool center) override {  // SplatMeshHC.hpp:58-91
        SplatSplitVector s = loadSplatsAny(path, center);   // .splat as the reference; .ply too (gsx_ply.hpp)
        SplatVector splats(s.size());
        for (size_t i = 0; i < s.size(); ++i) splats[i] = split_to_splat(s[i]);
        const uint32_t maxSplats = 500;
        if (splats.size() > maxSplats) splats.resize(maxSplats);
        hc.build(splats);
        splatData = hc.splats;
        uploaded = false;
        cutValid = false;
    }

protected:
    void uploadLod() override {
        const HcFlat f = hc.flatten();
        gsx_hc_flat t{static_cast<uint32_t>(f.parent.size()), f.parent.data(), f.is_leaf.data(), f.error.data(), f.weight.data()};
        gsx_check(ctx(), gsx_set_lod_hc(ctx(), &t), "gsx_set_lod_hc");
        cutValid = false;
    }
};

class SplatMeshGridHC : public SplatMesh {
public:
    GridHC gridhc;
    // The reference's alternative (commented out at SplatMeshGridHC.cpp:10-13): camera metric cut.
    bool useCameraMetric = false;

    void render(RenderPassEncoder& renderPass, Camera::Ptr camera, GUI::Parameters& params) override {  // SplatMeshGridHC.cpp:3-24
        setBuffers(renderPass);
        ensureUploaded();
        const vec3 cameraPos = xyz(camera->worldMatrix[3]);
        if (useCameraMetric) {
            const float cam[3] = {cameraPos.x, cameraPos.y, cameraPos.z};
            gsx_check(ctx(), gsx_select_hc(ctx(), cam, params.min_screen_area * 0.1f, nullptr), "gsx_select_hc");
            cutValid = false;
            selectionIsAll = false;
        } else if (!cutValid || cutDepth != params.depth) {
            indices = gridhc.get_indices_error(params.depth, params.min_screen_area);
            gsx_check(ctx(), gsx_select_list(ctx(), indices.data(), static_cast<uint32_t>(indices.size())), "gsx_select_list");
            cutDepth = params.depth;
            cutValid = true;
            selectionIsAll = false;
        }
        sortAndDraw(renderPass, cameraPos, static_cast<uint32_t>(splatData.size()));
    }
    void loadData(const std::string& path, bool center) override {  // SplatMeshGridHC.cpp:26-52
        SplatSplitVector s = loadSplatsAny(path, center);   // .splat as the reference; .ply too (gsx_ply.hpp)
        SplatVector splats(s.size());
        for (size_t i = 0; i < s.size(); ++i) splats[i] = split_to_splat(s[i]);
        const uint32_t maxSplats = 1000000;
        if (splats.size() > maxSplats) splats.resize(maxSplats);
        SplatVector kept;
        for (const auto& sp : splats)
            if (sp.transform[3][0] > 0.0f) kept.push_back(sp);
        gridhc.build(kept);
        splatData = gridhc.splats;
        uploaded = false;
        cutValid = false;
    }

protected:
    void uploadLod() override {
        const HcFlat f = gridhc.flatten();
        gsx_hc_flat t{static_cast<uint32_t>(f.parent.size()), f.parent.data(), f.is_leaf.data(), f.error.data(), f.weight.data()};
        gsx_check(ctx(), gsx_set_lod_hc(ctx(), &t), "gsx_set_lod_hc");
        cutValid = false;
    }
};

}  // namespace gsx
