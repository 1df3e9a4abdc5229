// Sanitizer build only (oracle/Makefile, target `asan`): csrc/pt_assets.cpp reports errors through pt::set_error / pt::last_error,
// which live in pt_scene.cpp together with the HIP-dependent scene flattener. This file provides just those two so that the
// file parsers can be built and run under AddressSanitizer / UBSan on the CPU, without a device.
#include <string>

namespace pt {
static thread_local std::string g_error;
int set_error(const std::string& msg) {
    g_error = msg;
    return -1;
}
const char* last_error() { return g_error.c_str(); }
}  // namespace pt
extern "C" const char* pt_last_error(void) { return pt::last_error(); }
