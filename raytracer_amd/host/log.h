// log.h -- luma::log / luma::warning with the reference's line prefixes (log.cpp:10-18),
// written with stdio because libstdc++ 11 has no <print>.
#pragma once
#include <string>

namespace luma {
void log(const std::string& message);
void warning(const std::string& message);
}  // namespace luma
