// Host-side plumbing shared by every entry point: version + last-error text.
#include "umr_common.h"
#include <string.h>

static thread_local char g_err[256] = "";

int umr_set_error(int code, const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}

extern "C" int umr_version(void) { return 100; }
extern "C" const char* umr_last_error_string(void) { return g_err; }
