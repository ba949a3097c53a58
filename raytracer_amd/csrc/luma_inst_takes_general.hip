// One group of render kernel instances (luma_instances.h), compiled on its own.
#include "luma_device.hpp"
#include "luma_instances.h"

namespace luma {
LUMA_LIST_TAKES_GENERAL(LUMA_KERNEL_DEFINE)
}  // namespace luma
