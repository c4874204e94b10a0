// orc_raster.cpp — CPU ORACLE (test infrastructure): restatement of the
// reference's GPU program resources/shader_quads_ordered.wgsl plus the
// fixed-function state the pipeline sets (main.cpp:324,338-344,359-368;
// renderer.cpp:153-176).
//
// PARITY UNPINNED against the reference's WebGPU backend: wgpu-native/Dawn are
// fetched at configure time (webgpu/webgpu.cmake:62-87) and cannot run here, so
// this restatement is the image oracle.  Pins chosen where WebGPU leaves the
// result implementation-defined (documented in DESIGN.md):
//   * f32 arithmetic, one rounding per op, glm-style association for mat ops;
//   * pixel centres at (i+0.5, j+0.5), y down, top-left fill rule, edge
//     functions evaluated in f64 on the f32 screen-space vertices;
//   * the colour target is linear RGBA8-unorm; blending accumulates in f32 and
//     is quantized once at the end (round to nearest).
#include "orc.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct V2 { float x, y; };

// wgsl:28-40
bool eigenvectors(float a00, float a01, float a10, float a11_, float E[4]) {
    const float a11 = a00;       // a[0][0]
    const float a22 = a11_;      // a[1][1]
    const float a1221 = a01 * a10;
    const float det = std::sqrt(a11 * a11 + 4 * a1221 - 2 * a11 * a22 + a22 * a22);
    const float lam1 = 0.5f * (a11 + a22 + det);
    const float lam2 = 0.5f * (a11 + a22 - det);
    const float phi = 0.5f * std::atan2(2 * a01, a11 - a22);
    const float c = std::cos(phi), s = std::sin(phi);
    const float s1 = std::sqrt(lam1), s2 = std::sqrt(lam2);
    E[0] = c * s1;  E[1] = s * s1;   // column 0
    E[2] = -s * s2; E[3] = c * s2;   // column 1
    return true;
}

void mat4_mul(const float* a, const float* b, float* out) { orc_mat4_mul(a, b, out); }

void mat4_vec(const float* m, const float v[4], float out[4]) {
    for (int r = 0; r < 4; ++r) {
        const float a0 = m[0 * 4 + r] * v[0] + m[1 * 4 + r] * v[1];
        const float a1 = m[2 * 4 + r] * v[2] + m[3 * 4 + r] * v[3];
        out[r] = a0 + a1;
    }
}

// 3x3 product, columns-major, ((a0*b0 + a1*b1) + a2*b2)
void mat3_mul(const float* a, const float* b, float* out) {
    float r[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            r[j * 3 + i] = a[0 * 3 + i] * b[j * 3 + 0] + a[1 * 3 + i] * b[j * 3 + 1] +
                           a[2 * 3 + i] * b[j * 3 + 2];
    std::memcpy(out, r, sizeof r);
}

const float QUAD[4][2] = {{-1.0f, -1.0f}, {1.0f, -1.0f}, {1.0f, 1.0f}, {-1.0f, 1.0f}};  // SplatMesh.h:171-176

// wgsl:75-91
void generate_permutation(uint32_t state, uint32_t arr[4]) {
    uint32_t s = state, fac[3];
    fac[0] = s / 6u; s -= fac[0] * 6u;
    fac[1] = s / 2u; s -= fac[1] * 2u;
    fac[2] = s;
    arr[0] = 0; arr[1] = 1; arr[2] = 2; arr[3] = 3;
    for (uint32_t i = 0; i < 3; ++i) {
        const uint32_t tmp = arr[i];
        arr[i] = arr[fac[i] + i];
        arr[fac[i] + i] = tmp;
    }
}

// wgsl:93-117
uint32_t generate_bayer(uint32_t x, uint32_t y, uint32_t bayer_size, uint32_t init_state) {
    uint32_t arr[4];
    generate_permutation(init_state % 24u, arr);
    uint32_t value = 0, cur = bayer_size, cx = x % bayer_size, cy = y % bayer_size, k = 1;
    while (cur > 1u) {
        cur = cur / 2u;
        const uint32_t cell_x = cx / cur, cell_y = cy / cur;
        cx = cx % cur; cy = cy % cur;
        value += arr[(cell_x * 2u + cell_y) & 3u] * k;
        k *= 4u;
    }
    return value;
}

}  // namespace

extern "C" {

// wgsl:42-73
void orc_vs_main(const float* sp, const orc_uniforms* u, float clip16[16]) {
    float wt[16];
    mat4_mul(u->view, u->model, wt);
    float W[9], S[9], Wt[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            W[c * 3 + r] = wt[c * 4 + r];
            S[c * 3 + r] = sp[c * 4 + r];
            Wt[r * 3 + c] = wt[c * 4 + r];
        }
    float center[4];
    mat4_vec(wt, sp + 12, center);
    float WS[9], V[9];
    mat3_mul(W, S, WS);
    mat3_mul(WS, Wt, V);
    float E[4];
    eigenvectors(V[0], V[1], V[3], V[4], E);
    for (int k = 0; k < 4; ++k) {
        const float qx = QUAD[k][0], qy = QUAD[k][1];
        // s_var_proj * quadPosition * cut_off * s
        float ox = E[0] * qx + E[2] * qy;
        float oy = E[1] * qx + E[3] * qy;
        ox = ox * u->cutOff * u->splatSize;
        oy = oy * u->cutOff * u->splatSize;
        const float v[4] = {center[0] + ox, center[1] + oy, center[2] + 0.0f, center[3] + 0.0f};
        mat4_vec(u->proj, v, clip16 + 4 * k);
    }
}

void orc_render(const float* splat20, uint64_t n, const uint32_t* order, uint64_t m,
                const orc_uniforms* u, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                uint8_t* rgba8, float* fb_out) {
    (void)n;
    std::vector<float> fb(static_cast<size_t>(w) * h * 4);
    for (size_t p = 0; p < static_cast<size_t>(w) * h; ++p) {  // renderer.cpp:162-164
        fb[4 * p + 0] = 0.05f; fb[4 * p + 1] = 0.05f; fb[4 * p + 2] = 0.05f; fb[4 * p + 3] = 1.0f;
    }
    const uint32_t B = u->bayerSize;
    for (uint64_t rank = 0; rank < m; ++rank) {
        const float* sp = splat20 + static_cast<uint64_t>(order[rank]) * 20;
        float clip[16];
        orc_vs_main(sp, u, clip);
        // clip volume 0 <= z <= w (WebGPU), NaN/inf vertices drop the primitive
        bool ok = true;
        V2 scr[4];
        for (int k = 0; k < 4 && ok; ++k) {
            const float* c = clip + 4 * k;
            for (int j = 0; j < 4; ++j) if (!std::isfinite(c[j])) ok = false;
            if (!ok) break;
            if (!(c[3] > 0.0f) || !(c[2] >= 0.0f) || !(c[2] <= c[3])) { ok = false; break; }
            const float nx = c[0] / c[3], ny = c[1] / c[3];
            scr[k].x = (nx * 0.5f + 0.5f) * static_cast<float>(w);
            scr[k].y = (0.5f - ny * 0.5f) * static_cast<float>(h);
        }
        if (!ok) continue;
        const float col[4] = {sp[16], sp[17], sp[18], sp[19]};
        const float cut = u->cutOff;
        const int tris[2][3] = {{0, 1, 2}, {0, 2, 3}};  // SplatMesh.h:158-160
        for (int t = 0; t < 2; ++t) {
            const V2 a = scr[tris[t][0]], b = scr[tris[t][1]], c = scr[tris[t][2]];
            const double area = (double)(b.x - a.x) * (double)(c.y - a.y) - (double)(b.y - a.y) * (double)(c.x - a.x);
            if (area == 0.0) continue;
            float fminx = std::fmin(a.x, std::fmin(b.x, c.x)), fmaxx = std::fmax(a.x, std::fmax(b.x, c.x));
            float fminy = std::fmin(a.y, std::fmin(b.y, c.y)), fmaxy = std::fmax(a.y, std::fmax(b.y, c.y));
            if (fmaxx < 0.0f || fmaxy < (float)y0 || fminx > (float)w || fminy > (float)y1) continue;
            const int ix0 = (int)std::fmax(0.0f, std::floor(fminx - 0.5f));
            const int ix1 = (int)std::fmin((float)w - 1.0f, std::ceil(fmaxx));
            const int iy0 = (int)std::fmax((float)y0, std::floor(fminy - 0.5f));
            const int iy1 = (int)std::fmin((float)y1 - 1.0f, std::ceil(fmaxy));
            const V2 vs[3] = {a, b, c};
            const float uv[3][2] = {
                {QUAD[tris[t][0]][0] * cut, QUAD[tris[t][0]][1] * cut},
                {QUAD[tris[t][1]][0] * cut, QUAD[tris[t][1]][1] * cut},
                {QUAD[tris[t][2]][0] * cut, QUAD[tris[t][2]][1] * cut}};
            const double sgn = area > 0 ? 1.0 : -1.0;
            for (int py = iy0; py <= iy1; ++py)
                for (int px = ix0; px <= ix1; ++px) {
                    const double sx = px + 0.5, sy = py + 0.5;
                    double bary[3];
                    bool inside = true;
                    for (int e = 0; e < 3 && inside; ++e) {
                        // edge from v[e+1] to v[e+2], weight for vertex e
                        const V2 p0 = vs[(e + 1) % 3], p1 = vs[(e + 2) % 3];
                        const double ex = (double)p1.x - p0.x, ey = (double)p1.y - p0.y;
                        double f = (ex * (sy - p0.y) - ey * (sx - p0.x)) * sgn;
                        if (f < 0) inside = false;
                        else if (f == 0) {
                            // top-left rule in the orientation-normalised frame (y down):
                            // a "top" edge is horizontal with the interior below it,
                            // a "left" edge has the interior to its right.
                            const double nex = ex * sgn, ney = ey * sgn;
                            const bool top = (ney == 0 && nex > 0);
                            const bool left = (ney < 0);
                            if (!(top || left)) inside = false;
                        }
                        bary[e] = f;
                    }
                    if (!inside) continue;
                    const double tot = bary[0] + bary[1] + bary[2];
                    const float uu = (float)((bary[0] * uv[0][0] + bary[1] * uv[1][0] + bary[2] * uv[2][0]) / tot);
                    const float vv = (float)((bary[0] * uv[0][1] + bary[1] * uv[1][1] + bary[2] * uv[2][1]) / tot);
                    // wgsl:120-134, 203-210
                    const float dist = uu * uu + vv * vv;
                    const float falloff = std::exp(-0.5f * dist);
                    float alpha = col[3] * falloff;
                    float src_a;
                    if (B == 1u) src_a = alpha;
                    else if (B == 10u) src_a = (alpha > u->bayerScale) ? 1.0f : 0.0f;
                    else {
                        const float bv = (float)generate_bayer((uint32_t)sx, (uint32_t)sy, B, (uint32_t)rank);
                        if ((alpha * u->bayerScale) * (float)(B * B + 1u) < bv + 1.0f) continue;  // discard
                        src_a = 1.0f;
                    }
                    // main.cpp:338-344: rgb = src*a + dst*(1-a); alpha: src*0 + dst*1
                    float* d = &fb[(static_cast<size_t>(py) * w + px) * 4];
                    for (int k = 0; k < 3; ++k) d[k] = col[k] * src_a + d[k] * (1.0f - src_a);
                }
        }
    }
    for (uint32_t y = y0; y < y1; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const size_t p = static_cast<size_t>(y) * w + x;
            for (int k = 0; k < 4; ++k) {
                float v = fb[4 * p + k];
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                if (rgba8) rgba8[4 * p + k] = (uint8_t)std::floor(v * 255.0f + 0.5f);
            }
        }
    if (fb_out) std::memcpy(fb_out, fb.data(), fb.size() * sizeof(float));
}

}  // extern "C"
