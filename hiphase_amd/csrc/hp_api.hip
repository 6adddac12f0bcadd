// hp_api.hip — misc entry points of the C ABI (errors, device discovery, version).
#include "hp_common.h"

#include <cstdlib>

namespace hp {
static thread_local std::string g_err;
thread_local double g_last_kernel_ms = 0.0;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace hp

extern "C" {

const char* hp_last_error(void) { return hp::g_err.c_str(); }
const char* hp_version(void) { return "hiphase_gpu 0.1.0 (gfx950)"; }
double hp_last_kernel_ms(void) { return hp::g_last_kernel_ms; }

int hp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// One process per GPU under torch.distributed / torchrun: LOCAL_RANK selects the device unless
// HP_DEVICE overrides it.
int hp_default_device(void) {
    const char* e = std::getenv("HP_DEVICE");
    if (!e) e = std::getenv("LOCAL_RANK");
    int d = e ? std::atoi(e) : 0;
    int n = hp_device_count();
    if (n > 0 && d >= n) d %= n;
    return d < 0 ? 0 : d;
}

}  // extern "C"
