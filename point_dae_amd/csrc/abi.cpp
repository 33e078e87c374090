// abi.cpp -- library identification and error text of the C ABI (include/pdae.h).
#include <string>

#include "common.h"

namespace pdae {
static thread_local std::string g_last_error;
void set_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace pdae

extern "C" const char* pdae_version(void) { return "pdae-hip gfx950 abi-1"; }
extern "C" const char* pdae_last_error(void) { return pdae::g_last_error.c_str(); }
