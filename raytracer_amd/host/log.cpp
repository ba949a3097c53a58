#include "log.h"

#include <cstdio>

namespace luma {

void log(const std::string& message) { std::printf("[INFORMATION]: %s\n", message.c_str()); }

void warning(const std::string& message) { std::printf("[WARNING]: %s\n", message.c_str()); }

}  // namespace luma
