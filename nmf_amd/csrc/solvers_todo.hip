// Entry points declared in include/nmfx.h whose device path is not built yet.
#include "nmfx_internal.h"
extern "C" {
int nmfx_anls_run(nmfx_handle_t E, double, double, int64_t, double, double, int64_t, int64_t) {
    if (E) E->err = "ANLS: not built yet"; return NMFX_E_ARG; }
}
