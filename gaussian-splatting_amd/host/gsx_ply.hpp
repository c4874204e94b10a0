// gsx_ply.hpp — reader for 3D-Gaussian-splatting PLY files (the INRIA training output), producing the
// reference's on-disk record: 32-byte SplatRaw (Splat.h:5-10).  SURVEY 8(f) rank 1: the reference itself
// has NO PLY code (its configs mention ".ply", ResourceManager.cpp:36-65 reads only .splat records), so
// there is nothing to be bit-exact with; the conversion below is the one .splat files are produced with
// in practice, and everything after it is the reference's own raw_to_split / split_to_splat path:
//   position  = (x, y, z)                                         f32
//   scale     = exp(scale_0..2)                                   f32
//   rgb       = (0.5 + C0 * f_dc_0..2) * 255,  C0 = 0.28209479177387814 (degree-0 SH), clamped, rounded   u8
//   alpha     = sigmoid(opacity) * 255, clamped, rounded          u8
//   quaternion= normalize(rot_0..3) * 128 + 128, clamped, rounded u8, order (w, x, y, z) as Splat.h:38-58 reads it
// Higher-order SH (f_rest_*) and any other property are skipped.  Supported: format binary_little_endian 1.0,
// one `vertex` element first, scalar properties of type float/double/(u)char/(u)short/(u)int.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "gsx_splat.hpp"

namespace gsx {

enum class PlyStatus { Ok = 0, CannotOpen, BadHeader, Unsupported, MissingProperty, Truncated };

inline const char* ply_status_text(PlyStatus s) {
    switch (s) {
        case PlyStatus::Ok: return "ok";
        case PlyStatus::CannotOpen: return "cannot open file";
        case PlyStatus::BadHeader: return "not a PLY header";
        case PlyStatus::Unsupported: return "unsupported PLY (need binary_little_endian 1.0, scalar vertex properties)";
        case PlyStatus::MissingProperty: return "vertex element lacks x/y/z, f_dc_0..2, opacity, scale_0..2 or rot_0..3";
        case PlyStatus::Truncated: return "file shorter than its header promises";
    }
    return "?";
}

namespace ply_detail {
inline int type_size(const std::string& t) {
    if (t == "float" || t == "float32" || t == "int" || t == "int32" || t == "uint" || t == "uint32") return 4;
    if (t == "double" || t == "float64") return 8;
    if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
    if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
    return 0;
}
inline double read_scalar(const unsigned char* p, const std::string& t) {
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
    if (t == "double" || t == "float64") { double v; std::memcpy(&v, p, 8); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, p, 1); return v; }
    uint8_t v; std::memcpy(&v, p, 1); return v;
}
inline uint8_t to_u8(double v) {
    if (!(v > 0.0)) return 0;          // also NaN
    if (v > 255.0) return 255;
    return static_cast<uint8_t>(std::nearbyint(v));   // round half to even, like numpy.rint
}
}  // namespace ply_detail

// Reads `path`; on success `out` holds one SplatRaw per vertex, in file order.
inline PlyStatus load_ply_raw(const std::string& path, std::vector<SplatRaw>& out) {
    using namespace ply_detail;
    out.clear();
    std::ifstream f(path, std::ios::binary);
    if (!f) return PlyStatus::CannotOpen;
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") return PlyStatus::BadHeader;
    struct Prop { std::string type, name; int offset; };
    std::vector<Prop> props;
    uint64_t n = 0;
    int stride = 0;
    bool in_vertex = false, seen_vertex = false, format_ok = false, ended = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream is(line);
        std::string word;
        is >> word;
        if (word == "format") {
            std::string kind, ver;
            is >> kind >> ver;
            format_ok = (kind == "binary_little_endian");
        } else if (word == "element") {
            std::string name;
            uint64_t count = 0;
            is >> name >> count;
            if (name == "vertex" && !seen_vertex) { in_vertex = true; seen_vertex = true; n = count; }
            else { if (!seen_vertex) return PlyStatus::Unsupported; in_vertex = false; }   // vertex data must come first
        } else if (word == "property" && in_vertex) {
            std::string type, name;
            is >> type;
            if (type == "list") return PlyStatus::Unsupported;
            is >> name;
            const int sz = type_size(type);
            if (!sz) return PlyStatus::Unsupported;
            props.push_back({type, name, stride});
            stride += sz;
        } else if (word == "end_header") { ended = true; break; }
    }
    if (!ended || !seen_vertex) return PlyStatus::BadHeader;
    if (!format_ok) return PlyStatus::Unsupported;
    static const char* need[14] = {"x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
                                   "rot_0", "rot_1", "rot_2", "rot_3"};
    const Prop* src[14] = {};
    for (int k = 0; k < 14; ++k) {
        for (const Prop& p : props)
            if (p.name == need[k]) { src[k] = &p; break; }
        if (!src[k]) return PlyStatus::MissingProperty;
    }
    {   // the vertex count comes from an untrusted header: it must fit the bytes that are actually left in the file
        const std::streamoff here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streamoff end = f.tellg();
        f.seekg(here, std::ios::beg);
        if (here < 0 || end < here || !f) return PlyStatus::BadHeader;
        const uint64_t left = static_cast<uint64_t>(end - here);
        if (stride <= 0 || n > left / static_cast<uint64_t>(stride)) return PlyStatus::Truncated;
    }
    out.resize(n);
    const uint64_t chunk = 1u << 16;
    std::vector<unsigned char> buf(static_cast<size_t>(chunk) * stride);
    const double C0 = 0.28209479177387814;
    for (uint64_t base = 0; base < n; base += chunk) {
        const uint64_t cnt = (n - base < chunk) ? n - base : chunk;
        f.read(reinterpret_cast<char*>(buf.data()), static_cast<std::streamsize>(cnt * stride));
        if (static_cast<uint64_t>(f.gcount()) != cnt * stride) { out.clear(); return PlyStatus::Truncated; }
        for (uint64_t i = 0; i < cnt; ++i) {
            const unsigned char* rec = buf.data() + i * stride;
            double v[14];
            for (int k = 0; k < 14; ++k) v[k] = read_scalar(rec + src[k]->offset, src[k]->type);
            SplatRaw& r = out[base + i];
            for (int k = 0; k < 3; ++k) {
                r.position[k] = static_cast<float>(v[k]);
                r.scale[k] = static_cast<float>(std::exp(v[7 + k]));
            }
            r.color[0] = to_u8((0.5 + C0 * v[3]) * 255.0);
            r.color[1] = to_u8((0.5 + C0 * v[4]) * 255.0);
            r.color[2] = to_u8((0.5 + C0 * v[5]) * 255.0);
            r.color[3] = to_u8(255.0 / (1.0 + std::exp(-v[6])));
            double qn = std::sqrt(v[10] * v[10] + v[11] * v[11] + v[12] * v[12] + v[13] * v[13]);
            if (!(qn > 0.0)) { v[10] = 1.0; v[11] = v[12] = v[13] = 0.0; qn = 1.0; }
            for (int k = 0; k < 4; ++k) r.rotation[k] = to_u8(v[10 + k] / qn * 128.0 + 128.0);
        }
    }
    return PlyStatus::Ok;
}

// What the facade's loadData() calls: `.ply` goes through the reader above, everything else is the reference's
// loadSplatsRaw.  A PLY that exists but cannot be parsed throws (the reference has no such case; a missing file is
// an empty scene in both).
inline SplatSplitVector loadSplatsAny(const std::string& path, bool center) {
    const bool is_ply = path.size() >= 4 && (path.compare(path.size() - 4, 4, ".ply") == 0 || path.compare(path.size() - 4, 4, ".PLY") == 0);
    if (!is_ply) return loadSplatsRaw(path, center);
    std::vector<SplatRaw> raw;
    const PlyStatus st = load_ply_raw(path, raw);
    if (st == PlyStatus::CannotOpen) return {};
    if (st != PlyStatus::Ok) throw std::runtime_error(path + ": " + ply_status_text(st));
    SplatSplitVector splats;
    splats.reserve(raw.size());
    for (const SplatRaw& r : raw) splats.push_back(raw_to_split(r));
    if (center && !splats.empty()) center_splats(splats);
    return splats;
}

}  // namespace gsx
