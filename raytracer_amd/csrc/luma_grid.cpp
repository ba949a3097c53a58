#include "luma_grid.h"

#include <algorithm>
#include <cmath>

namespace luma {

namespace {

constexpr uint32_t kMaxDim = 256;             // cells per axis
constexpr uint64_t kMaxCells = 4u << 20;      // 4M cells = 16 MB of offsets
constexpr uint32_t kMaxSpan = 64;             // a sphere listed in more cells than this is "large"

struct Box { float lo[3], hi[3]; };

// The sphere's bounding box, widened by more than any rounding of the ray-sphere arithmetic can
// move a reported hit point: a hit the reference's float test accepts lies within a few ulp of
// the sphere, i.e. well inside this box.
bool sphere_box(const luma_sphere& s, Box& b)
{
    const float c[3] = { s.cx, s.cy, s.cz };
    const float r = std::fabs(s.radius);
    if (!std::isfinite(r) || !std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) { return false; }
    const float reach = std::max({ std::fabs(c[0]), std::fabs(c[1]), std::fabs(c[2]) }) + r;
    const float pad = 1e-4f * reach + 1e-30f;
    for (int a = 0; a < 3; a++) { b.lo[a] = c[a] - r - pad; b.hi[a] = c[a] + r + pad; }
    return std::isfinite(b.lo[0] + b.lo[1] + b.lo[2] + b.hi[0] + b.hi[1] + b.hi[2]);
}

}  // namespace

void build_grid(const luma_sphere* spheres, uint32_t n, HostGrid& g)
{
    g = HostGrid{};
    std::vector<Box> boxes(n);
    std::vector<uint8_t> small(n, 0);
    std::vector<float> radii;
    radii.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (sphere_box(spheres[i], boxes[i])) { small[i] = 1; radii.push_back(std::fabs(spheres[i].radius)); }
    }
    float r_med = 0.f;
    if (!radii.empty()) {
        std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
        r_med = radii[radii.size() / 2];
    }
    // a sphere much larger than the typical one (the ground) would sit in thousands of cells
    uint32_t n_small = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (small[i] && r_med > 0.f && std::fabs(spheres[i].radius) > 8.f * r_med) { small[i] = 0; }
        n_small += small[i];
    }

    float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 0.f, 0.f, 0.f };
    bool first = true;
    for (uint32_t i = 0; i < n; i++) {
        if (!small[i]) { continue; }
        for (int a = 0; a < 3; a++) {
            lo[a] = first ? boxes[i].lo[a] : std::min(lo[a], boxes[i].lo[a]);
            hi[a] = first ? boxes[i].hi[a] : std::max(hi[a], boxes[i].hi[a]);
        }
        first = false;
    }
    if (n_small) {
        // about two cells per sphere, cells no smaller than the median radius
        double ext[3], vol = 1.0;
        for (int a = 0; a < 3; a++) { ext[a] = std::max(double(hi[a]) - lo[a], 1e-20); vol *= ext[a]; }
        // (measured at 4096 and 65536 spheres: 1 / 2 / 4 / 8 cells per sphere -> 3.05 / 2.93 / 2.97 / 3.16 and
        //  3.92 / 3.64 / 3.53 / 3.44 ms per 1080p x 2 spp call)
        double h = std::max(std::cbrt(vol / (2.0 * n_small)), double(r_med));
        const double longest = std::max({ ext[0], ext[1], ext[2] });
        h = std::max(h, longest / kMaxDim);
        for (;;) {
            uint64_t cells = 1;
            for (int a = 0; a < 3; a++) {
                g.dims[a] = static_cast<uint32_t>(std::min<double>(kMaxDim, std::max(1.0, std::ceil(ext[a] / h))));
                cells *= g.dims[a];
            }
            if (cells <= kMaxCells) { break; }
            h *= 1.26;
        }
        g.h = static_cast<float>(h);
        // one cell of slack around the boxes so that no sphere touches the outer faces
        for (int a = 0; a < 3; a++) {
            g.org[a] = lo[a] - g.h;
            g.dims[a] = std::min(kMaxDim + 2u, g.dims[a] + 2u);
        }
    }

    const uint32_t dx = g.dims[0], dy = g.dims[1], dz = g.dims[2];
    const size_t n_cells = static_cast<size_t>(dx) * dy * dz;
    auto cell_range = [&](const Box& b, uint32_t c0[3], uint32_t c1[3]) {
        uint64_t span = 1;
        for (int a = 0; a < 3; a++) {
            const float slack = 1e-3f * g.h;
            const double f0 = std::floor((double(b.lo[a]) - slack - g.org[a]) / g.h);
            const double f1 = std::floor((double(b.hi[a]) + slack - g.org[a]) / g.h);
            c0[a] = static_cast<uint32_t>(std::min<double>(g.dims[a] - 1, std::max(0.0, f0)));
            c1[a] = static_cast<uint32_t>(std::min<double>(g.dims[a] - 1, std::max(0.0, f1)));
            span *= c1[a] - c0[a] + 1;
        }
        return span;
    };
    std::vector<uint32_t> count(n_cells + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (!small[i]) { continue; }
        uint32_t c0[3], c1[3];
        if (cell_range(boxes[i], c0, c1) > kMaxSpan) { small[i] = 0; continue; }
        for (uint32_t z = c0[2]; z <= c1[2]; z++) {
            for (uint32_t y = c0[1]; y <= c1[1]; y++) {
                for (uint32_t x = c0[0]; x <= c1[0]; x++) { count[(static_cast<size_t>(z) * dy + y) * dx + x + 1]++; }
            }
        }
    }
    for (size_t c = 0; c < n_cells; c++) { count[c + 1] += count[c]; }
    g.cell_start = count;
    g.items.assign(count[n_cells], 0);
    std::vector<uint32_t> fill(count.begin(), count.end() - 1);
    for (uint32_t i = 0; i < n; i++) {                     // ascending i => ascending inside every cell
        if (!small[i]) { g.large.push_back(i); continue; }
        uint32_t c0[3], c1[3];
        cell_range(boxes[i], c0, c1);
        for (uint32_t z = c0[2]; z <= c1[2]; z++) {
            for (uint32_t y = c0[1]; y <= c1[1]; y++) {
                for (uint32_t x = c0[0]; x <= c1[0]; x++) { g.items[fill[(static_cast<size_t>(z) * dy + y) * dx + x]++] = i; }
            }
        }
    }
}

namespace {

constexpr uint32_t kViewMaxBins = 2048;       // a sphere listed in more direction bins than this is "large"

// Range of u = tan(angle) on one face, for directions whose projection on the (a, major) plane
// passes through the disc that is the sphere's shadow there: centre angle phi, half-angle theta,
// cut to the face's +-45 degrees.  False when the disc cannot be seen through this face.
bool tangent_range(double qa, double qm, double r, double& lo, double& hi)
{
    const double quarter = 0.78539816339744830962, slack = 1e-6;
    const double rho = std::hypot(qa, qm);
    if (rho <= r) { lo = -1.0; hi = 1.0; return true; }     // the shadow covers the origin: every angle
    const double phi = std::atan2(qa, qm), theta = std::asin(std::min(1.0, r / rho)) + slack;
    bool any = false;
    lo = 1.0; hi = -1.0;
    for (int turn = -1; turn <= 1; turn++) {                  // the interval and its copies a full turn away
        const double a0 = std::max(phi - theta + turn * 6.283185307179586, -quarter);
        const double a1 = std::min(phi + theta + turn * 6.283185307179586, quarter);
        if (a0 <= a1) { lo = std::min(lo, std::tan(a0)); hi = std::max(hi, std::tan(a1)); any = true; }
    }
    return any;
}

}  // namespace

void build_view_grid(const luma_sphere* spheres, uint32_t n, const float origin[3], HostViewGrid& g)
{
    g = HostViewGrid{};
    // bins about half the angular size of a typical sphere: ~4 sqrt(n) per face edge, 16 .. 256.  The kernels pay for
    // the LONGEST list among a wave's 64 rays, so short lists count for more than the table's size (measured with
    // 2 / 3 / 4 / 6 sqrt(n): config #5 band 7.05 / 6.52 / 6.29 / 6.65 ms, 1024 spheres 2.99 / 2.65 / 2.50 / 2.45; an edge
    // beyond 256 thrashes L2 on the offsets: 65536 spheres 10.0 -> 12.2 ms at 512)
    const uint32_t res = static_cast<uint32_t>(std::min(256.0, std::max(16.0, std::ceil(4.0 * std::sqrt(double(n) + 1.0)))));
    g.res = res;
    const size_t n_bins = size_t(6) * res * res;

    struct Rect { uint32_t u0, u1, v0, v1; bool on; };
    std::vector<Rect> rects(size_t(n) * 6, Rect{ 0, 0, 0, 0, false });
    std::vector<uint8_t> small(n, 0);
    std::vector<float> nearest(n, 0.f);
    auto bin_of = [&](double t) {                             // u or v in [-1, 1] -> bin index
        return static_cast<long>(std::floor((t + 1.0) * 0.5 * res));
    };
    for (uint32_t i = 0; i < n; i++) {
        const luma_sphere& s = spheres[i];
        const double q[3] = { double(s.cx) - origin[0], double(s.cy) - origin[1], double(s.cz) - origin[2] };
        const double dist = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        // the float test can accept a hit a few ulp outside the sphere: pad well beyond that
        const double r = std::fabs(double(s.radius)) * (1.0 + 1e-4) + 1e-5 * dist + 1e-30;
        if (!std::isfinite(r) || !std::isfinite(dist) || dist <= r) { continue; }   // non-finite, or the origin is inside: large
        nearest[i] = static_cast<float>(std::max(0.0, (dist - r) * (1.0 - 1e-6)));   // (r is the padded radius: a lower bound)
        uint64_t bins = 0;
        for (int face = 0; face < 6; face++) {
            const int m = face >> 1, a = (m + 1) % 3 < (m + 2) % 3 ? (m + 1) % 3 : (m + 2) % 3, b = 3 - m - a;   // u, v axes in axis order
            const double sg = (face & 1) ? -1.0 : 1.0;
            double ulo, uhi, vlo, vhi;
            if (!tangent_range(q[a], sg * q[m], r, ulo, uhi) || !tangent_range(q[b], sg * q[m], r, vlo, vhi)) { continue; }
            Rect& rc = rects[size_t(i) * 6 + face];
            // one bin of margin each way: the kernel bins in float, with an approximate reciprocal
            rc.u0 = static_cast<uint32_t>(std::max(0L, bin_of(ulo) - 1));
            rc.u1 = static_cast<uint32_t>(std::min<long>(res - 1, bin_of(uhi) + 1));
            rc.v0 = static_cast<uint32_t>(std::max(0L, bin_of(vlo) - 1));
            rc.v1 = static_cast<uint32_t>(std::min<long>(res - 1, bin_of(vhi) + 1));
            rc.on = true;
            bins += uint64_t(rc.u1 - rc.u0 + 1) * (rc.v1 - rc.v0 + 1);
        }
        small[i] = bins <= kViewMaxBins;
    }
    std::vector<uint32_t> count(n_bins + 1, 0);
    auto each_bin = [&](uint32_t i, auto&& fn) {
        for (int face = 0; face < 6; face++) {
            const Rect& rc = rects[size_t(i) * 6 + face];
            if (!rc.on) { continue; }
            for (uint32_t v = rc.v0; v <= rc.v1; v++) {
                for (uint32_t u = rc.u0; u <= rc.u1; u++) { fn((size_t(face) * res + v) * res + u); }
            }
        }
    };
    for (uint32_t i = 0; i < n; i++) { if (small[i]) { each_bin(i, [&](size_t bin) { count[bin + 1]++; }); } }
    for (size_t c = 0; c < n_bins; c++) { count[c + 1] += count[c]; }
    g.cell_start = count;
    g.items.assign(count[n_bins], 0);
    std::vector<uint32_t> fill(count.begin(), count.end() - 1);
    for (uint32_t i = 0; i < n; i++) {
        if (!small[i]) { g.large.push_back(i); continue; }
        each_bin(i, [&](size_t bin) { g.items[fill[bin]++] = i; });
    }
    // nearest first inside every bin (ties by index, so the order is fully determined)
    for (size_t c = 0; c < n_bins; c++) {
        std::sort(g.items.begin() + count[c], g.items.begin() + count[c + 1], [&](uint32_t x, uint32_t y) {
            return nearest[x] < nearest[y] || (nearest[x] == nearest[y] && x < y);
        });
    }
    g.nearest.resize(g.items.size());
    for (size_t k = 0; k < g.items.size(); k++) { g.nearest[k] = nearest[g.items[k]]; }
}

}  // namespace luma
