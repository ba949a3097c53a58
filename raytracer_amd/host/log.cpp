#include "log.h"

#include <cstdio>

namespace luma {

// Flushed per line: the viewer reads these through a pipe, and a fatal error is followed by abort().
void log(const std::string& message) { std::printf("[INFORMATION]: %s\n", message.c_str()); std::fflush(stdout); }

void warning(const std::string& message) { std::printf("[WARNING]: %s\n", message.c_str()); std::fflush(stdout); }

}  // namespace luma
