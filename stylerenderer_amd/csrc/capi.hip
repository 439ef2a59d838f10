// Library-level entry points of libstylerenderer_hip.so (error strings, ABI version).
#include "common.h"

extern "C" int sr_abi_version(void) { return 2; }

extern "C" const char* sr_error_string(int code) {
    if (code == SR_OK) return "ok";
    if (code == SR_EINVAL) return "stylerenderer_amd: invalid argument (size, null pointer or unsupported combination)";
    if (code == SR_ERANGE) return "stylerenderer_amd: size exceeds the kernel's index range";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "stylerenderer_amd: unknown error";
}
