// viewer_script_test.cpp — host-only checks of the viewer front end's script layer (host/gsx_viewer.hpp): the line
// grammar, the sliders' ranges (gui.cpp:58-122) and the frame clock.  No GPU call is made.  Prints "ok".
#include <cassert>
#include <cmath>
#include <cstdio>

#include "../../gaussian-splatting_amd/host/gsx_viewer.hpp"

using namespace gsx;

#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "viewer_script_test: %s failed (line %d)\n", #x, __LINE__); return 1; } } while (0)

template <typename Fn>
static bool throws(Fn fn) { try { fn(); } catch (const std::invalid_argument&) { return true; } return false; }

int main() {
    ViewerEvent e;
    CHECK(!parse_viewer_event("", e) && !parse_viewer_event("   ", e) && !parse_viewer_event("# move 1 2", e));
    CHECK(parse_viewer_event("move 12.5 -3", e) && e.kind == ViewerEvent::Move && e.a == 12.5 && e.b == -3.0);
    CHECK(parse_viewer_event("button 0 1", e) && e.kind == ViewerEvent::Button && e.a == 0.0 && e.b == 1.0);
    CHECK(parse_viewer_event("scroll -2", e) && e.kind == ViewerEvent::Scroll && e.a == -2.0);
    CHECK(parse_viewer_event("set fov 60", e) && e.kind == ViewerEvent::Set && e.name == "fov" && e.a == 60.0);
    CHECK(parse_viewer_event("dt 0.0125", e) && e.kind == ViewerEvent::DeltaTime && e.a == 0.0125);
    CHECK(parse_viewer_event("frame", e) && e.kind == ViewerEvent::Frame && e.a == 1.0);
    CHECK(parse_viewer_event("frame 7", e) && e.a == 7.0);
    CHECK(parse_viewer_event("quit", e) && e.kind == ViewerEvent::Quit);
    CHECK(throws([&] { parse_viewer_event("move 1", e); }) && throws([&] { parse_viewer_event("jump 1 2", e); }) &&
          throws([&] { parse_viewer_event("set", e); }) && throws([&] { parse_viewer_event("frame -1", e); }) &&
          throws([&] { parse_viewer_event("frame many", e); }) && throws([&] { parse_viewer_event("frame 3x", e); }));

    GUI::Parameters p;                                   // gui.hpp:16-30 defaults
    CHECK(p.splatSize == 1.0f && p.cutOff == 3.5f && p.fov == 45.0f && p.bayerSize == 1u && p.orbitRate == 1.0f && p.min_screen_area == 0.2f);
    CHECK(set_gui_parameter(p, "splatSize", 9.0) && p.splatSize == 5.0f);           // clamped like the sliders
    CHECK(set_gui_parameter(p, "splatSize", 0.0) && p.splatSize == 0.001f);
    CHECK(set_gui_parameter(p, "cutOff", 2.25) && p.cutOff == 2.25f);
    CHECK(set_gui_parameter(p, "bayerSize", 9) && p.bayerSize == 6u && set_gui_parameter(p, "bayerSize", -1) && p.bayerSize == 0u);
    CHECK(set_gui_parameter(p, "bayerScale", 0.1) && p.bayerScale == 0.5f);
    CHECK(set_gui_parameter(p, "fov", 1000) && p.fov == 100.0f);
    CHECK(set_gui_parameter(p, "orbitRate", 5) && p.orbitRate == 2.0f && set_gui_parameter(p, "scrollRate", 0) && p.scrollRate == 0.1f);
    CHECK(set_gui_parameter(p, "depth", 5000) && p.depth == 2000u);
    CHECK(set_gui_parameter(p, "min_screen_area", 1.5) && p.min_screen_area == 1.0f);
    CHECK(set_gui_parameter(p, "weight_e", 0.25) && p.weight_e == 0.25f && set_gui_parameter(p, "weight_w", 2) && p.weight_w == 1.0f &&
          set_gui_parameter(p, "weight_d", -2) && p.weight_d == 0.0f);
    CHECK(!set_gui_parameter(p, "no_such_slider", 1.0));

    FrameClock c;
    c.new_frame();
    CHECK(c.DeltaTime == 1.0 / 60.0 && c.Framerate == 0.0);                          // ImGui's first frame
    c.new_frame();
    CHECK(c.DeltaTime >= 0.0 && c.Framerate > 0.0);
    c.fixed_dt = 0.25;
    for (int i = 0; i < 100; ++i) c.new_frame();
    CHECK(c.DeltaTime == 0.25 && std::isfinite(c.Framerate));
    std::puts("ok");
    return 0;
}
