// api.hip -- library-wide pieces of the C ABI: version and thread-local error text.
#include "lvg_common.h"

static thread_local char g_lvg_error[512] = "";

void lvg_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_lvg_error, sizeof(g_lvg_error), fmt, ap);
    va_end(ap);
}

extern "C" int lvg_abi_version(void) { return LVG_ABI_VERSION; }
extern "C" const char* lvg_last_error(void) { return g_lvg_error; }
