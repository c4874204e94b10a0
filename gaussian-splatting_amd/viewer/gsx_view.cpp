// gsx_view — the optional viewer front end as a program (reference main.cpp:110-137): loads a scene, runs
// the reference's main loop on the drop-in classes (host/gsx_viewer.hpp) and presents every frame as PPM files and / or
// a raw RGBA8 stream, with the window's events read from a script (a file, or stdin / a FIFO a remote front end feeds).
//
//   gsx_view scene.splat|scene.ply [--mesh base|octree|hc|gridhc] [--size WxH] [--gpu N] [--events FILE|-]
//            [--frames N] [--out DIR] [--raw FILE|-] [--every N] [--uniforms FILE] [--stats N] [--no-center] [--pipelined]
//
//   --events   the event script (see gsx_viewer.hpp); without it the camera rests and --frames frames are rendered
//   --frames   stop after N presented frames (default: until the script ends; 1 without a script)
//   --out      directory that receives frame_00000.ppm, ...
//   --raw      file / FIFO / - (stdout) that receives width*height*4 bytes per frame, e.g.
//              gsx_view s.splat --size 1280x720 --events ev.txt --raw - | ffplay -f rawvideo -pixel_format rgba -video_size 1280x720 -
//   --uniforms appends the 224-byte uniform block and the camera's 64-byte world matrix of every presented frame
//   --pipelined  present every frame two loops late; both of the library's frame slots stay busy (about a third more frames per second)
//   --stats    frames between two "ms/frame (FPS)" lines on stderr (gui.cpp:130-131; default 60, 0 = none)
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>

#include "../host/gsx_viewer.hpp"

using namespace gsx;

struct Options {
    std::string scene, mesh = "base", events, out, raw, uniforms;
    int w = 1000, h = 800, gpu = 0;                  // main.cpp:101-102
    uint64_t frames = 0, every = 1;
    uint32_t stats = 60;
    bool center = true, pipelined = false;
};

template <typename Mesh>
static int run(const Options& o) {
    Application<Mesh> app;
    app.scenePath = o.scene;
    app.centerScene = o.center;
    app.width = static_cast<float>(o.w);
    app.height = static_cast<float>(o.h);
    app.gpu = o.gpu;
    app.statsEvery = o.stats;
    app.pipelined = o.pipelined;
    std::ifstream script;
    if (o.events == "-") app.events = &std::cin;
    else if (!o.events.empty()) {
        script.open(o.events);
        if (!script) { std::fprintf(stderr, "gsx_view: cannot open %s\n", o.events.c_str()); return 2; }
        app.events = &script;
    }
    FrameSink sink;
    sink.ppmDir = o.out;
    sink.every = o.every;
    if (o.raw == "-") sink.raw = stdout;
    else if (!o.raw.empty() && !(sink.raw = std::fopen(o.raw.c_str(), "wb"))) { std::fprintf(stderr, "gsx_view: cannot open %s\n", o.raw.c_str()); return 2; }
    std::ofstream uni;
    if (!o.uniforms.empty()) uni.open(o.uniforms, std::ios::binary);
    app.onFrame = [&](const uint8_t* px, uint32_t w, uint32_t h, uint64_t frame) {
        sink(px, w, h, frame);
        if (uni) {
            uni.write(reinterpret_cast<const char*>(&app.presentedUniforms), sizeof app.presentedUniforms);
            uni.write(reinterpret_cast<const char*>(app.presentedWorld.data()), 64);
        }
    };
    if (!app.Initialize()) return 1;                                              // main.cpp:113-115
    std::fprintf(stderr, "gsx_view: %zu splats, %d x %d, mesh %s\n", app.splatMesh.splatData.size(), o.w, o.h, o.mesh.c_str());
    const uint64_t limit = o.frames ? o.frames : (app.events ? std::numeric_limits<uint64_t>::max() : 1u);
    while (app.IsRunning() && app.framesRendered < limit) app.MainLoop();         // main.cpp:128-131
    app.Terminate();                                                              // main.cpp:134
    if (sink.raw && sink.raw != stdout) std::fclose(sink.raw);
    std::fprintf(stderr, "gsx_view: %llu frames presented\n", static_cast<unsigned long long>(app.framesPresented));
    return 0;
}

int main(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "gsx_view: %s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--mesh") o.mesh = value();
        else if (a == "--size") { if (std::sscanf(value(), "%dx%d", &o.w, &o.h) != 2 || o.w <= 0 || o.h <= 0) { std::fprintf(stderr, "gsx_view: --size WxH\n"); return 2; } }
        else if (a == "--gpu") o.gpu = std::atoi(value());
        else if (a == "--events") o.events = value();
        else if (a == "--frames") o.frames = std::strtoull(value(), nullptr, 10);
        else if (a == "--out") o.out = value();
        else if (a == "--raw") o.raw = value();
        else if (a == "--every") o.every = std::max<uint64_t>(1, std::strtoull(value(), nullptr, 10));
        else if (a == "--uniforms") o.uniforms = value();
        else if (a == "--stats") o.stats = static_cast<uint32_t>(std::strtoul(value(), nullptr, 10));
        else if (a == "--no-center") o.center = false;
        else if (a == "--pipelined") o.pipelined = true;
        else if (a[0] == '-' && a != "-") { std::fprintf(stderr, "gsx_view: unknown option %s\n", a.c_str()); return 2; }
        else o.scene = a;
    }
    if (o.scene.empty()) {
        std::fprintf(stderr, "usage: gsx_view scene.splat|scene.ply [--mesh base|octree|hc|gridhc] [--size WxH] [--gpu N] [--events FILE|-] [--frames N]\n"
                             "                [--out DIR] [--raw FILE|-] [--every N] [--uniforms FILE] [--stats N] [--no-center] [--pipelined]\n");
        return 2;
    }
    try {
        if (o.mesh == "base") return run<SplatMesh>(o);
        if (o.mesh == "octree") return run<SplatMeshOctree>(o);
        if (o.mesh == "hc") return run<SplatMeshHC>(o);
        if (o.mesh == "gridhc") return run<SplatMeshGridHC>(o);                   // (the reference's choice, main.cpp:98)
        std::fprintf(stderr, "gsx_view: --mesh base|octree|hc|gridhc\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "gsx_view: %s\n", e.what());
        return 1;
    }
}
