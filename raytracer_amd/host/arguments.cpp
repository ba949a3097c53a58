#include "arguments.h"

#include <algorithm>

#include <charconv>
#include <cstdlib>
#include <functional>
#include <unordered_map>

#include "log.h"

namespace luma {

Options _options;

namespace {

// Whole-string integer, as parse_integer in the reference (arguments.cpp:52-62).
template <typename T>
bool whole_number(const std::string& text, T& out, int base = 10)
{
    const char* first = text.data();
    const char* last = first + text.size();
    if (base == 16 && text.size() > 2 && text[0] == '0' && (text[1] == 'x' || text[1] == 'X')) { first += 2; }
    const auto [ptr, ec] = std::from_chars(first, last, out, base);
    return ec == std::errc() && ptr == last && first != last;
}

// Whole-string decimal real (strtof accepts more: hex floats, inf, nan; those are refused).
bool real_number(const std::string& text, float& out)
{
    if (text.empty() || text.find_first_not_of("+-0123456789.eE") != std::string::npos) { return false; }
    char* end = nullptr;
    const float v = std::strtof(text.c_str(), &end);
    if (end != text.c_str() + text.size()) { return false; }
    out = v;
    return true;
}

using Setter = std::function<bool(const std::string&)>;   // false: value not understood

template <typename T>
Setter number_into(T& field)
{
    return [&field](const std::string& v) { T parsed{}; if (!whole_number(v, parsed)) { return false; } field = parsed; return true; };
}

}  // namespace

void Arguments::parse(const std::vector<std::string>& arguments)
{
    struct Entry { const char* what; Setter set; };
    const std::unordered_map<std::string, Entry> table{
        { "width",   { "image width", number_into(_options.width) } },
        { "height",  { "image height", number_into(_options.height) } },
        { "samples", { "image samples", number_into(_options.samples) } },
        { "bounces", { "image bounces", number_into(_options.bounces) } },
        { "paths",   { "path count", [](const std::string& v) {
            if (_options.mode != RenderMode::PATHTRACE) {   // arguments.cpp:179-183: warn, keep going
                warning("path count `" + v + "` will be ignored if render mode is not set to \"pathtrace\"");
            }
            return number_into(_options.paths)(v); } } },
        { "mode",    { "mode", [](const std::string& v) {
            if (v == "raytrace") { _options.mode = RenderMode::RAYTRACE; return true; }
            if (v == "pathtrace") { _options.mode = RenderMode::PATHTRACE; return true; }
            return false; } } },
        { "context", { "context", [](const std::string& v) {
            if (v == "interative" || v == "interactive") { _options.context = Context::INTERACTIVE; return true; }   // arguments.h:30 spells it "interative"
            if (v == "directx") { _options.context = Context::DIRECTX; return true; }
            if (v == "headless") { _options.context = Context::HEADLESS; return true; }
            return false; } } },
        { "backend", { "backend", [](const std::string& v) {
            if (v == "cpu") { _options.backend = Backend::CPU; return true; }
            if (v == "hip") { _options.backend = Backend::HIP; return true; }
            return false; } } },
        { "gpus",    { "gpu count", [](const std::string& v) {
            std::uint32_t n = 0;
            if (!whole_number(v, n) || n < 1 || n > 8) { return false; }
            _options.gpus = n; return true; } } },
        { "devices", { "device list", [](const std::string& v) {   // e.g. --devices=0,1,2,3 (overrides --gpus)
            std::vector<int> ids;
            std::size_t at = 0;
            while (at <= v.size()) {
                const std::size_t comma = std::min(v.find(',', at), v.size());
                std::uint32_t id = 0;
                if (!whole_number(v.substr(at, comma - at), id) || id > 1023) { return false; }
                ids.push_back(static_cast<int>(id));
                at = comma + 1;
            }
            if (ids.empty() || ids.size() > 64) { return false; }
            _options.devices = ids;
            _options.gpus = static_cast<std::uint32_t>(ids.size()); return true; } } },
        { "gather",  { "gather", [](const std::string& v) {       // how the bands of --gpus N reach `target`
            if (v == "pcie") { _options.gather_rccl = false; return true; }   // each GPU copies its rows to the host frame
            if (v == "rccl") { _options.gather_rccl = true; return true; }    // bands gathered on GPU 0 over RCCL, one D2H
            return false; } } },
        { "seed",    { "seed", [](const std::string& v) {
            std::uint64_t s = 0;
            const bool hex = v.size() > 2 && v[0] == '0' && (v[1] == 'x' || v[1] == 'X');
            if (!whole_number(v, s, hex ? 16 : 10)) { return false; }
            _options.seed = s; return true; } } },
        { "pos",     { "camera position", [](const std::string& v) {
            float xyz[3];
            std::size_t at = 0;
            for (int i = 0; i < 3; i++) {
                const std::size_t comma = std::min(v.find(',', at), v.size());
                if (!real_number(v.substr(at, comma - at), xyz[i]) || (i < 2 && comma == v.size())) { return false; }
                at = comma + 1;
            }
            if (at <= v.size()) { return false; }                  // a fourth component
            std::copy(xyz, xyz + 3, _options.pos);
            _options.have_pose = true; return true; } } },
        { "pitch",   { "camera pitch", [](const std::string& v) {
            if (!real_number(v, _options.pitch)) { return false; }
            _options.have_pose = true; return true; } } },
        { "yaw",     { "camera yaw", [](const std::string& v) {
            if (!real_number(v, _options.yaw)) { return false; }
            _options.have_pose = true; return true; } } },
        { "frames",  { "frame count", number_into(_options.frames) } },
        { "flags",   { "flags", number_into(_options.flags) } },
        { "shm",     { "shared-memory name", [](const std::string& v) { if (v.empty()) { return false; } _options.shm = v; return true; } } },
        { "output",  { "output path", [](const std::string& v) { if (v.empty()) { return false; } _options.output = v; return true; } } },
    };

    for (std::size_t i = 0; i < arguments.size(); i++) {
        const std::string& argument = arguments[i];
        if (argument.rfind("--", 0) != 0) {
            log("invalid argument `" + argument + "`");              // arguments.cpp:227-230
            continue;
        }
        std::string key = argument.substr(2), value;
        const auto eq = key.find('=');
        if (eq != std::string::npos) {                               // --key=value (arguments.cpp:103)
            value = key.substr(eq + 1);
            key.erase(eq);
        } else if (i + 1 < arguments.size() && arguments[i + 1].rfind("--", 0) != 0) {
            value = arguments[++i];                                  // --key value (BASELINE.json's spelling)
        } else {
            log("unrecognized option format `" + argument + "`");    // arguments.cpp:107
            continue;
        }
        const auto it = table.find(key);
        if (it == table.end()) {
            log("unrecognized option `" + key + "`");                // arguments.cpp:115
            continue;
        }
        if (!it->second.set(value)) {
            log(std::string("unrecognized ") + it->second.what + " `" + value + "`");   // arguments.cpp:131 etc.
        }
    }
}

}  // namespace luma
